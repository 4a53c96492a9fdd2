"""Frame sharding of one sample over the GPUs of a node: one process per GPU, torch.distributed
("nccl" == RCCL over xGMI on the box, "gloo" in the CPU tests).

The image batch (cfg x B x T) is split into contiguous frame ranges per rank; both cfg halves and all B samples of a
frame range stay on one rank, weights are replicated.  Everything 2-D (GroupNorm-2D, conv3x3, linears, spatial attention,
sampler / guider elementwise) is local.  The frame axis couples ranks in exactly three places, each with its own exchange:
  (i)   temporal self-attention          -> all-gather of K|V along frames            (FrameShard.allgather_frames)
  (ii)  (3,1,1) temporal convolution     -> +-1 frame halo with the ring neighbours   (FrameShard.convt3 / tmix_small)
  (iii) 3-D GroupNorm statistics         -> all-reduce of (sum, sumsq) per group      (FrameShard.allreduce_stats)
The reference has no distributed code on this path; this module is new design, verified against the unsharded result.

Data movement is exact-size and copy-free:
  * (i) is a grouped point-to-point exchange (xGMI is a full point-to-point mesh: one direct link per peer), every rank's K|V
    block lands straight in its frame rows of the [B, T_global, S, 2C] buffer the attention kernel reads with strides - uneven
    shards (18 frames over 8 ranks = 3,3,2,...) move exactly their own bytes (no padding to the largest shard, no trim copy).
    It is issued asynchronously right after the temporal to_k / to_v GEMM; the to_q GEMM runs while it is in flight.
  * (ii) uses the split-halo layout of v3d_gemm (ABI 3): GroupNorm writes the local frames into the middle of a persistent
    [B*S | B*T_local*S | B*S]-row buffer, the neighbours' boundary frames are received straight into the two outer slabs and
    ONE 3-tap GEMM covers all B samples (the first version zero-filled and copied a [B, T+2, S, C] buffer per conv and looped
    over samples in Python).
`sharded_sample` runs the whole sampler loop with x sharded (25 steps), decodes the local frames and gathers only the decoded
frames; `bench.py --gpus N` reports it as the frame-sharded (strong-scaling, latency) mode next to the replica mode.
"""
from __future__ import annotations

import contextlib
import copy
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def frame_partition(T: int, world: int) -> List[range]:
    """Contiguous, as-even-as-possible split (18 over 8 -> 3,3,2,2,2,2,2,2)."""
    base, rem = divmod(T, world)
    out, t = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append(range(t, t + n))
        t += n
    return out


_ACTIVE: Optional["FrameShard"] = None


def active_shard() -> Optional["FrameShard"]:
    """The FrameShard the engine executors (run_unet / run_decoder) use when none is passed explicitly."""
    return _ACTIVE


class _Handle:
    """Completion handle of an exchange: wait() orders the CURRENT stream behind it (NCCL) / blocks the host (gloo)."""

    def __init__(self, works=(), after=None):
        self.works = list(works)
        self.after = after

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []
        if self.after is not None:
            self.after()
            self.after = None


class FrameShard:
    def __init__(self, T_global: int, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if self.world > T_global:
            raise ValueError(f"cannot shard {T_global} frames over {self.world} ranks")
        self.T_global = T_global
        self.parts = frame_partition(T_global, self.world)
        self.local_frames = self.parts[self.rank]
        self.T_local = len(self.local_frames)
        self.t0 = self.local_frames.start
        self.first = self.rank == 0
        self.last = self.rank == self.world - 1
        self.context_frame0: Optional[torch.Tensor] = None     # [B, ...] context of every sample's GLOBAL frame 0 (set by activate)
        self._bufs: Dict[Tuple, torch.Tensor] = {}
        self.bytes_sent = 0                                    # payload bytes this rank has sent (bench / tests)

    def describe(self) -> str:
        return "+".join(str(len(p)) for p in self.parts)

    # ---- activation -----------------------------------------------------------------------------------
    @contextlib.contextmanager
    def activate(self, context_frame0: Optional[torch.Tensor] = None):
        """Everything the engine evaluates inside this context runs frame-sharded on LOCAL tensors ([(b T_local), ...])."""
        global _ACTIVE
        prev, prev_ctx = _ACTIVE, self.context_frame0
        _ACTIVE = self
        if context_frame0 is not None:
            self.context_frame0 = context_frame0
        try:
            yield self
        finally:
            _ACTIVE = prev
            self.context_frame0 = prev_ctx

    # ---- communication primitives (a test subclass stages them through the host) -------------------------
    def _peer(self, r: int) -> int:
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def _allreduce_sum(self, t: torch.Tensor) -> None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def _exchange(self, sends: Sequence[Tuple[torch.Tensor, int]], recvs: Sequence[Tuple[torch.Tensor, int]], async_op: bool = False) -> _Handle:
        """Grouped point-to-point exchange: (contiguous tensor, group rank) pairs.  Messages between one pair of ranks match in
        list order on both sides."""
        ops_ = [dist.P2POp(dist.isend, t, self._peer(r), self.group) for t, r in sends]
        ops_ += [dist.P2POp(dist.irecv, t, self._peer(r), self.group) for t, r in recvs]
        self.bytes_sent += sum(t.numel() * t.element_size() for t, _ in sends)
        h = _Handle(dist.batch_isend_irecv(ops_) if ops_ else ())
        if not async_op:
            h.wait()
        return h

    # ---- slicing of replicated inputs ---------------------------------------------------------------
    def take_frames(self, x: torch.Tensor, B: int) -> torch.Tensor:
        """[(b T_global), ...] -> [(b T_local), ...] rows of this rank."""
        shp = x.shape
        x = x.reshape((B, self.T_global) + tuple(shp[1:]))[:, self.t0:self.t0 + self.T_local]
        return x.reshape((B * self.T_local,) + tuple(shp[1:])).contiguous()

    def gather_frames_out(self, x: torch.Tensor, B: int) -> torch.Tensor:
        """Inverse of take_frames for final outputs: [(b T_local), ...] -> [(b T_global), ...] on every rank."""
        shp = x.shape
        g, _ = self.allgather_frames(x.reshape((B, self.T_local) + tuple(shp[1:])).contiguous())
        return g.reshape((B * self.T_global,) + tuple(shp[1:]))

    # ---- (iii) 3-D GroupNorm statistics -------------------------------------------------------------
    def allreduce_stats(self, stats: torch.Tensor) -> torch.Tensor:
        self._allreduce_sum(stats)
        return stats

    # ---- (i) temporal attention ---------------------------------------------------------------------
    def allgather_frames(self, x: torch.Tensor, async_op: bool = False) -> Tuple[torch.Tensor, _Handle]:
        """x [B, T_local, ...] contiguous -> ([B, T_global, ...], handle).  Exact-size grouped send / recv: rank r's frames land
        directly in rows parts[r] of every sample; nothing is padded, concatenated or trimmed."""
        assert x.is_contiguous() and x.shape[1] == self.T_local, (tuple(x.shape), self.T_local)
        B = x.shape[0]
        out = torch.empty((B, self.T_global) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
        out[:, self.t0:self.t0 + self.T_local].copy_(x)
        sends, recvs = [], []
        for r, part in enumerate(self.parts):
            if r == self.rank:
                continue
            for b in range(B):
                sends.append((x[b], r))
                recvs.append((out[b, part.start:part.stop], r))
        return out, self._exchange(sends, recvs, async_op=async_op)

    # ---- (ii) temporal conv halos -------------------------------------------------------------------
    def halo_buffer(self, B: int, S: int, C: int, dtype, device, slot: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
        """Persistent split-halo activation buffer [(B + B*T_local + B) * S, C] and its middle (local frames) view: the producer
        (GroupNorm apply) writes the view, `convt3` receives the neighbours' frames into the outer slabs."""
        key = (B, S, C, dtype, str(device), slot)
        buf = self._bufs.get(key)
        if buf is None:
            buf = self._bufs[key] = torch.empty(((B + B * self.T_local + B) * S, C), dtype=dtype, device=device)
        return buf, buf[B * S:(B + B * self.T_local) * S]

    def _halo_exchange(self, buf: torch.Tensor, B: int, S: int) -> None:
        Tl = self.T_local
        mid0, right0 = B * S, (B + B * Tl) * S
        sends, recvs = [], []
        for b in range(B):
            f0 = mid0 + b * Tl * S
            if not self.first:
                sends.append((buf[f0:f0 + S], self.rank - 1))                              # my first frame -> previous rank
                recvs.append((buf[b * S:(b + 1) * S], self.rank - 1))                      # its last frame -> my frame -1
            if not self.last:
                sends.append((buf[f0 + (Tl - 1) * S:f0 + Tl * S], self.rank + 1))          # my last frame -> next rank
                recvs.append((buf[right0 + b * S:right0 + (b + 1) * S], self.rank + 1))    # its first frame -> my frame T_local
        self._exchange(sends, recvs)

    def convt3(self, ops, buf: torch.Tensor, w: torch.Tensor, b, g, **epi):
        """Frame-sharded (3,1,1) conv over a split-halo buffer (see halo_buffer): exchange the +-1 frame halos, then ONE 3-tap
        GEMM over all B samples.  The halo slabs at the global ends are never read (tmin / tmax mask those taps to zero)."""
        B, S, Tl = g.B, g.S, self.T_local
        self._halo_exchange(buf, B, S)
        return ops.convt3(buf, w, b, Tl, S, tmin=0 if self.first else -1, tmax=Tl - 1 if self.last else Tl, a_row0=B * S,
                          M=B * Tl * S, halo_rows=B * S, **epi)

    def tmix_small(self, ops, y: torch.Tensor, w, b, g, out_ch: int):
        """Frame-sharded AE3DConv.time_mix_conv on the fp32 [rows, 4] map (one sample at a time; the decode has B = 1)."""
        B, S, Tl = g.B, g.S, self.T_local
        C = y.shape[-1]
        yv = y.reshape(B, Tl, S, C)
        buf = torch.empty((B, Tl + 2, S, C), dtype=y.dtype, device=y.device)
        buf[:, 1:Tl + 1] = yv
        if self.first:
            buf[:, 0].zero_()
        if self.last:
            buf[:, Tl + 1].zero_()
        sends, recvs = [], []
        for bi in range(B):
            if not self.first:
                sends.append((buf[bi, 1], self.rank - 1))
                recvs.append((buf[bi, 0], self.rank - 1))
            if not self.last:
                sends.append((buf[bi, Tl], self.rank + 1))
                recvs.append((buf[bi, Tl + 1], self.rank + 1))
        self._exchange(sends, recvs)
        tmin = 0 if self.first else -1
        tmax = Tl - 1 if self.last else Tl
        outs = [ops.tmix_small(buf[bi].reshape((Tl + 2) * S, -1), w, b, 1, Tl, S, out_ch, tmin, tmax, row0=S) for bi in range(B)]
        return torch.cat(outs, dim=0) if B > 1 else outs[0]


def sharded_unet_eval(net, shard: FrameShard, x, scale, concat, timesteps, context, y, image_only_indicator):
    """One frame-sharded U-Net evaluation.  Inputs are the LOCAL rows ([(b T_local), ...]) except `context`, which may be
    the full [(b T_global), ...] tensor: frame 0's context feeds every rank's temporal cross-attention."""
    from .engine.unet import run_unet
    n_loc = x.shape[0]
    B = n_loc // shard.T_local
    if context.shape[0] == B * shard.T_global:
        ctx0 = context.reshape((B, shard.T_global) + tuple(context.shape[1:]))[:, 0]
        context = shard.take_frames(context, B)
    else:
        raise ValueError("sharded_unet_eval needs the full (b T_global) context to find each sample's frame-0 context")
    return run_unet(net.packed(), x, scale, concat, timesteps, context, y, shard.T_local, image_only_indicator, shard=shard,
                    context_frame0=ctx0)


def local_sampler(sampler, shard: FrameShard):
    """Shallow copy of a sampler whose guider applies THIS rank's slice of the per-frame guidance scale (guiders.py:61-86 index the
    scale by global frame id)."""
    s = copy.copy(sampler)
    g = copy.copy(sampler.guider)
    sc = getattr(g, "scale", None)
    if isinstance(sc, torch.Tensor) and sc.numel() == shard.T_global and hasattr(g, "num_frames"):
        g.scale = sc.reshape(1, -1)[:, shard.t0:shard.t0 + shard.T_local].clone()
        g.num_frames = shard.T_local
        if hasattr(g, "_scale_dev"):
            g._scale_dev = None
    s.guider = g
    return s


def sharded_sample(shard: FrameShard, sampler, denoiser, network, decode, noise, c: dict, uc: dict, *, B: int = 1,
                   image_only_indicator: Optional[torch.Tensor] = None, gather: bool = True):
    """One sample (B inputs x T_global frames) with the frame axis sharded for the WHOLE path: every rank keeps its frames of the
    sampler state x for all steps (denoiser evaluations exchange K|V, halo frames and GroupNorm sums inside the U-Net), decodes its
    own frames (VideoDecoder: halos + GroupNorm sums again) and only the decoded frames are gathered.

    noise / c / uc are the FULL tensors [(b T_global), ...], identical on every rank (what scripts/pub/V3D_512.py builds before the
    sampler); `sampler`, `denoiser`, `network` (OpenAIWrapper) are the ordinary plugin objects; decode(z_local) -> frames_local is
    e.g. `lambda z: engine.decode_first_stage(z)`.  Returns frames [(b T_global), 3, H, W] (every rank) or the local frames."""
    T, Tl = shard.T_global, shard.T_local
    dev = noise.device
    take = lambda t: shard.take_frames(t, B)
    x = take(noise)
    c_loc = {k: (take(v) if isinstance(v, torch.Tensor) and v.shape[0] == B * T else v) for k, v in c.items()}
    uc_loc = {k: (take(v) if isinstance(v, torch.Tensor) and v.shape[0] == B * T else v) for k, v in uc.items()}
    # frame-0 context of every sample of the guided batch [uc ; c] (guiders.py:95 order), needed by every rank's temporal blocks
    ctx0 = torch.cat([uc["crossattn"].reshape((B, T) + tuple(uc["crossattn"].shape[1:]))[:, 0],
                      c["crossattn"].reshape((B, T) + tuple(c["crossattn"].shape[1:]))[:, 0]], dim=0)
    if image_only_indicator is None:
        image_only_indicator = torch.zeros(2 * B, T, device=dev)
    ioi_loc = image_only_indicator[:, shard.t0:shard.t0 + Tl].contiguous()
    extra = {"image_only_indicator": ioi_loc, "num_video_frames": Tl}
    smp = local_sampler(sampler, shard)
    with shard.activate(context_frame0=ctx0):
        z = smp(lambda inp, sigma, cc: denoiser(network, inp, sigma, cc, **extra), x, cond=c_loc, uc=uc_loc)
        frames = decode(z)
    if not gather:
        return frames
    return shard.gather_frames_out(frames.contiguous(), B)
