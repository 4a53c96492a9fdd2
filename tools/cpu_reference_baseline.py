"""BASELINE.md support (build container only): time the REFERENCE's own VideoUNet / VideoDecoder modules (imported unmodified from
/root/reference through oracle/ref_import.py) on this container's host cores, next to the fp32 oracle port (oracle/sgm_oracle.py) on the
same inputs and weights - the bench.py `cpu_baseline` leg runs the port on the GPU box, where /root/reference does not exist; this script
shows what the port's number stands for.  fp32, no autocast, attention mode "softmax", torch threads = all cores.
    python tools/cpu_reference_baseline.py [T_frames=6]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.set_grad_enabled(False)
from oracle import ref_import, sgm_oracle as O  # noqa: E402
from v3d_amd import synth  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ref = ref_import.load()
cfg = synth.unet_config(320, attn_type="softmax")
net = ref["video_model"].VideoUNet(**cfg).eval()
sd = synth.seeded_state_dict(net, 3)
net.load_state_dict(sd)
g = torch.Generator().manual_seed(0)
x, ts = torch.randn(T, 8, 64, 64, generator=g), torch.randn(T, generator=g)
ctx, y = torch.randn(T, 1, 1024, generator=g), torch.randn(T, 768, generator=g)
ioi = torch.zeros(1, T)


def timed(fn, n=2):
    fn()
    ts_ = []
    for _ in range(n):
        t0 = time.time()
        r = fn()
        ts_.append(time.time() - t0)
    return r, ts_


r_ref, t_ref = timed(lambda: net(x, ts, context=ctx, y=y, num_video_frames=T, image_only_indicator=ioi))
r_port, t_port = timed(lambda: O.unet_forward(sd, synth.unet_config(320), x, ts, ctx, y, T, ioi))
err = ((r_ref - r_port).abs().max() / r_ref.abs().max()).item()
dec = ref["temporal_ae"].VideoDecoder(**synth.decoder_config(128)).eval()
dsd = synth.seeded_state_dict(dec, 4)
dec.load_state_dict(dsd)
z = torch.randn(2, 4, 64, 64, generator=g)
d_ref, td_ref = timed(lambda: dec(z, timesteps=2), n=1)
d_port, td_port = timed(lambda: O.decoder_forward(dsd, synth.decoder_config(128), z, 2), n=1)
out = {"threads": torch.get_num_threads(), "unet_images": T, "reference_unet_s": [round(t, 1) for t in t_ref], "port_unet_s": [round(t, 1) for t in t_port],
       "port_vs_reference_max_rel": err, "reference_decode_2_frames_s": round(td_ref[0], 1), "port_decode_2_frames_s": round(td_port[0], 1)}
mean = lambda v: sum(v) / len(v)
for k, tu, td in (("reference", mean(t_ref), td_ref[0]), ("port", mean(t_port), td_port[0])):
    t_sample = 25 * tu * (36 / T) + td * 9
    out[k + "_frames_per_s_extrapolated"] = round(18 / t_sample, 6)
print(json.dumps(out))
