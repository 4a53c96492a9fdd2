"""ctypes binding of libv3d_hip.so (include/v3d_hip.h) — the product operator backend.

Tensors are only used as device-memory handles here (`data_ptr()`), torch provides the stream.  Everything
fails loudly: a missing library, a missing symbol, a non-GPU tensor or a non-zero return code raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from .ops import GemmCall, OpsBase

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("V3D_HIP_LIB") or os.path.join(_HERE, "lib", "libv3d_hip.so")     # (override: A/B runs of two builds on one box)

ABI_VERSION = 5

c_i64, c_i32, c_f32, c_f64, c_vp = C.c_int64, C.c_int32, C.c_float, C.c_double, C.c_void_p


class _GemmArgs(C.Structure):
    _fields_ = [
        ("A", c_vp), ("W", c_vp), ("out", c_vp), ("bias", c_vp), ("add", c_vp), ("res1", c_vp), ("res2", c_vp),
        ("coef", c_vp),
        ("M", c_i64), ("N", c_i64), ("K", c_i64),
        ("lda", c_i64), ("ldw", c_i64), ("ldo", c_i64), ("ldr1", c_i64), ("ldr2", c_i64),
        ("a_rows", c_i64), ("a_row0", c_i64),
        ("add_rpg", c_i64), ("add_ld", c_i64), ("coef_rpg", c_i64),
        ("c_acc", c_f32), ("c_res1", c_f32), ("c_res2", c_f32),
        ("mode", c_i32), ("geglu", c_i32), ("out_fp32", c_i32),
        ("Hin", c_i32), ("Win", c_i32), ("Hout", c_i32), ("Wout", c_i32), ("stride", c_i32), ("up", c_i32),
        ("T", c_i32), ("tmin", c_i32), ("tmax", c_i32),
        ("S", c_i64),
        ("batch", c_i32), ("pad_mode", c_i32),
        ("sA", c_i64), ("sW", c_i64), ("sO", c_i64),
        ("halo_rows", c_i64),
        ("gn_stats", c_vp), ("gn_rps", c_i64), ("gn_cpg", c_i32), ("gn_in_silu", c_i32),
        ("gn_nslots", c_i64),
        ("gn_in_table", c_vp), ("gn_in_rps", c_i64), ("gn_in_rows", c_i64),
        ("A2", c_vp), ("K1", c_i64), ("lda2", c_i64),
    ]


# name -> (restype, argtypes); must list every symbol declared in include/v3d_hip.h
SIGNATURES = {
    "v3d_abi_version": (c_i32, []),
    "v3d_last_error": (C.c_char_p, []),
    "v3d_device_info": (c_i32, [c_vp]),
    "v3d_gemm": (c_i32, [C.POINTER(_GemmArgs), c_vp]),
    "v3d_sizeof_gemm_args": (c_i32, []),
    "v3d_ff_fused": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_f32, c_f32, c_f32,
                             c_vp, c_i64, c_i64, c_i32, c_i32, c_vp]),
    "v3d_ln_ff_fused": (c_i32, [c_vp, c_i64, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_f32, c_f32, c_f32,
                                c_vp, c_i64, c_i64, c_i32, c_i32, c_vp]),
    "v3d_ln_proj": (c_i32, [c_vp, c_i64, c_f32, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_i32, c_i64, c_vp]),
    "v3d_gemm_gn_in_supported": (c_i32, [C.POINTER(_GemmArgs)]),
    "v3d_groupnorm_stats": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i32, c_i64, c_vp]),
    "v3d_groupnorm_stats_table": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i64, c_vp, c_vp, C.c_double, C.c_float, c_vp, c_vp]),
    "v3d_groupnorm_finalize": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_i64, c_f64, c_f32, c_vp, c_vp]),
    "v3d_groupnorm_apply": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "v3d_groupnorm_small_supported": (c_i32, [c_i64, c_i64, c_i64, c_i32, c_i64]),
    "v3d_groupnorm_small": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i64, c_f32, c_i32, c_vp]),
    "v3d_layernorm": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_f32, c_vp]),
    "v3d_attn_spatial": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_f32, c_vp]),
    "v3d_attn_temporal": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64,
                                  c_i64, c_i64, c_i32, c_i32, c_i64, c_i32, c_f32, c_vp]),
    "v3d_attn_vae_d512": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_f32, c_vp]),
    "v3d_quant_fp8_tiles": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_vp]),
    "v3d_quant_fp8_slab": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp]),
    "v3d_attn_spatial_fp8": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_f32, c_vp]),
    "v3d_softmax_rows": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp]),
    "v3d_timestep_embedding": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_f32, c_vp]),
    "v3d_silu_add": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "v3d_edm_scalings": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "v3d_pack_input": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "v3d_pack_input_im2col3x3": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_i64, c_vp]),
    "v3d_tapsum3x3": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "v3d_denoise_combine": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "v3d_cfg_combine": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "v3d_euler_step": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "v3d_heun_step": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "v3d_clip_preprocess": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "v3d_gelu_bf16": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "v3d_frames_to_uint8": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i64, c_vp]),
    "v3d_axpb_f32": (c_i32, [c_vp, c_f32, c_f32, c_vp, c_i64, c_vp]),
    "v3d_blend_coefs": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "v3d_nchw_to_nhwc_bf16": (c_i32, [c_vp, c_f32, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp]),
    "v3d_tmix_small": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_i32, c_i32, c_i32, c_i64, c_vp]),
    "v3d_copy2d_bf16": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp]),
}


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """dlopen the C-ABI library and bind every declared symbol (no GPU needed for this step)."""
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `python -m v3d_amd.build`). There is no CPU fallback for the V3D hot path.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud
        fn.restype = res
        fn.argtypes = args
    v = lib.v3d_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"libv3d_hip.so ABI version {v} != expected {ABI_VERSION}; rebuild the extension")
    if lib.v3d_sizeof_gemm_args() != C.sizeof(_GemmArgs):
        raise RuntimeError(f"v3d_gemm_args layout mismatch: library {lib.v3d_sizeof_gemm_args()} B vs binding {C.sizeof(_GemmArgs)} B")
    return lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class HipOps(OpsBase):
    name = "hip"

    def __init__(self, device: Optional[torch.device] = None, lib_path: Optional[str] = None):
        # lib_path: another build of the library (tools/mainloop_ab.py interleaves several builds in ONE process: same box, same clocks)
        self.lib = load_library(lib_path or LIB_PATH)
        if not torch.cuda.is_available():
            raise RuntimeError("v3d_amd: no HIP device visible (torch.cuda.is_available() is False); the V3D hot path "
                               "has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        info = (c_i32 * 4)()
        self._check(self.lib.v3d_device_info(C.cast(info, c_vp)), "v3d_device_info")
        self.cu_count, self.lds_bytes, self.wave_size, self.arch = info[0], info[1], info[2], info[3]
        if self.wave_size != 64:
            raise RuntimeError(f"v3d_amd kernels are wave64 gfx950 code; device reports wave size {self.wave_size}")

    # ---- health ----------------------------------------------------------------------------------
    def streamk_timeouts(self) -> int:
        """Stream-K hand-offs of this process that gave up waiting for a donor block (gemm_common.h sk_gather: bounded spin).  Never expected:
        every block of a persistent launch is resident.  A non-zero count means a tile was retired without a donor's partial sums - callers that
        care about the numbers (bench.py, the entry script, smoke()) check it after their work and fail loudly.  Synchronises the device."""
        fn = self.lib.v3d_debug_sk_timeouts
        fn.restype, fn.argtypes = C.c_longlong, []
        return int(fn())

    def last_gemm_launch(self) -> dict:
        """What the last `gemm` of this thread actually launched (the library's own record, gemm.hip v3d_debug_last_gemm_launch): kernel family
        (1 = v1, 2 = v2, 3 = persistent v3, 5 = LDS-haloed, 6 = two persistent 4-wave blocks per CU), tile, tile count, co-resident blocks per CU, split-K ways, stream-K tail - and `fill`,
        the fraction of the CU slots the launch keeps busy over its rounds (1.0 with a stream-K tail: the last round is shared out)."""
        fn = self.lib.v3d_debug_last_gemm_launch
        fn.restype, fn.argtypes = c_i32, [c_vp]
        o = (C.c_longlong * 8)()
        fn(C.cast(o, c_vp))
        fam, bm, bn, tiles, bpc, sk, tail, cus = (int(v) for v in o)
        slots = max(1, bpc) * max(1, cus)
        tiles_eff = tiles * max(1, sk)
        fill = 1.0 if tail else tiles_eff / (-(-tiles_eff // slots) * slots) if tiles_eff else 0.0
        return {"family": fam, "bm": bm, "bn": bn, "tiles": tiles, "blocks_per_cu": bpc, "splitk": sk, "streamk_tail": tail, "cus": cus, "fill": fill}

    def check_health(self):
        n = self.streamk_timeouts()
        if n:
            raise RuntimeError(f"v3d_amd: {n} stream-K hand-off(s) timed out in this process (a donor block was not co-resident): results of the "
                               "affected launches are wrong; rerun with V3D_STREAMK=0 and report the configuration")

    # ---- plumbing ---------------------------------------------------------------------------------
    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.v3d_last_error()
            raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    @staticmethod
    def _req(t: torch.Tensor, dtype, what: str, inner_contig: bool = True):
        if t.device.type != "cuda":
            raise RuntimeError(f"{what}: tensor is on {t.device}, HIP ops need device memory")
        if t.dtype != dtype:
            raise RuntimeError(f"{what}: dtype {t.dtype} != {dtype}")
        if inner_contig and t.dim() > 0 and t.stride(-1) != 1 and t.shape[-1] != 1:
            raise RuntimeError(f"{what}: inner stride must be 1")
        return t

    @staticmethod
    def _req_c(t: torch.Tensor, dtype, what: str):
        HipOps._req(t, dtype, what)
        if not t.is_contiguous():
            raise RuntimeError(f"{what}: tensor must be contiguous")
        return t

    # ---- primitives -------------------------------------------------------------------------------
    _MAX_OPERAND_BYTES = 0xFFFFFF00      # one hardware buffer descriptor (v3d_gemm's A / W operands)

    def _gemm_in_row_chunks(self, g: GemmCall) -> bool:
        """An activation operand beyond one buffer descriptor (> 4 GiB: the 24-frame 576 x 1024 scene decode has a 7.2 GB conv input)
        is contracted in chunks of whole images (conv3x3) / rows (linear): every row-indexed argument is sliced consistently."""
        import dataclasses
        import math
        if g.batch != 1 or g.mode == 2 or g.A.dim() != 2:
            return False
        lda = g.A.stride(0)
        rows_in = g.A.shape[0]
        if rows_in * lda * 2 <= self._MAX_OPERAND_BYTES:
            return False
        if g.mode == 1:
            s_out, s_in = g.Hout * g.Wout, g.Hin * g.Win
        else:
            s_out = s_in = 1
        align = 1
        for rpg in (g.add_rpg if g.add is not None else 0, g.coef_rpg if g.coef is not None else 0, g.gn_rps if g.gn_stats is not None else 0):
            if rpg:
                align = align * rpg // math.gcd(align, rpg)
        unit_out = s_out * align // math.gcd(s_out, align)          # output rows per indivisible unit
        unit_in = unit_out // s_out * s_in
        units = g.M // unit_out
        if units * unit_out != g.M:
            raise RuntimeError("gemm: operand larger than 4 GiB and M is not a whole number of images / row groups")
        per = max(1, int((self._MAX_OPERAND_BYTES // 2) // (unit_in * lda * 2)))
        for u0 in range(0, units, per):
            u1 = min(units, u0 + per)
            r0, r1 = u0 * unit_out, u1 * unit_out
            kw = dict(A=g.A[u0 * unit_in:u1 * unit_in], out=g.out[r0:r1], M=r1 - r0, a_rows=0)
            if g.res1 is not None:
                kw["res1"] = g.res1[r0:r1]
            if g.res2 is not None:
                kw["res2"] = g.res2[r0:r1]
            if g.add is not None:
                kw["add"] = g.add[r0 // g.add_rpg:]
            if g.coef is not None:
                kw["coef"] = g.coef.reshape(-1, 3)[r0 // g.coef_rpg:].contiguous()
            if g.gn_stats is not None:
                kw["gn_stats"] = g.gn_stats[r0 // g.gn_rps:]
            if g.gn_in is not None:
                raise RuntimeError("gemm: gn_in is not defined for operands beyond one buffer descriptor (normalise with groupnorm_apply first)")
            self.gemm(dataclasses.replace(g, **kw))
        return True

    def _gemm_args(self, g: GemmCall) -> _GemmArgs:
        bf, f32 = torch.bfloat16, torch.float32
        a = _GemmArgs()
        self._req(g.A, bf, "gemm.A")
        self._req(g.W, bf, "gemm.W")
        ldw = g.W.stride(-2)
        if g.W.dim() == 3 and g.mode != 0 and g.W.stride(0) != g.N * ldw:
            raise RuntimeError("gemm.W: taps must be densely stacked")
        out_fp32 = g.out.dtype == f32
        self._req(g.out, f32 if out_fp32 else bf, "gemm.out")
        a.A, a.W, a.out = g.A.data_ptr(), g.W.data_ptr(), g.out.data_ptr()
        a.bias = _ptr(None if g.bias is None else self._req_c(g.bias, f32, "gemm.bias"))
        a.add = _ptr(None if g.add is None else self._req(g.add, f32, "gemm.add"))
        a.res1 = _ptr(None if g.res1 is None else self._req(g.res1, bf, "gemm.res1"))
        a.res2 = _ptr(None if g.res2 is None else self._req(g.res2, bf, "gemm.res2"))
        a.coef = _ptr(None if g.coef is None else self._req_c(g.coef, f32, "gemm.coef"))
        a.M, a.N, a.K = g.M, g.N, g.K
        a.lda = g.A.stride(-2) if g.A.dim() >= 2 else g.K
        a.ldw = ldw
        a.ldo = g.out.stride(-2)
        a.ldr1 = 0 if g.res1 is None else g.res1.stride(-2)
        a.ldr2 = 0 if g.res2 is None else g.res2.stride(-2)
        a.a_rows = g.a_rows if g.a_rows else g.A.shape[-2]
        a.a_row0 = g.a_row0
        a.add_rpg, a.add_ld, a.coef_rpg = g.add_rpg, g.add_ld, g.coef_rpg
        a.c_acc, a.c_res1, a.c_res2 = g.c_acc, g.c_res1, g.c_res2
        a.mode, a.geglu, a.out_fp32 = g.mode, int(g.geglu), int(out_fp32)
        a.Hin, a.Win, a.Hout, a.Wout, a.stride, a.up = g.Hin, g.Win, g.Hout, g.Wout, g.stride, g.up
        a.T, a.tmin, a.tmax, a.S = g.T, g.tmin, g.tmax, g.S
        a.batch = g.batch
        a.pad_mode = g.pad_mode
        a.halo_rows = g.halo_rows
        if g.gn_stats is not None:
            self._req_c(g.gn_stats, f32, "gemm.gn_stats")
            a.gn_stats, a.gn_rps, a.gn_cpg, a.gn_nslots = g.gn_stats.data_ptr(), g.gn_rps, g.gn_cpg, g.gn_stats.shape[1]
        if g.gn_in is not None:
            self._req_c(g.gn_in, f32, "gemm.gn_in")
            a.gn_in_table, a.gn_in_rps, a.gn_in_rows, a.gn_in_silu = g.gn_in.data_ptr(), g.gn_in_rps, g.gn_in.shape[0], int(g.gn_in_silu)
        if g.A2 is not None:
            self._req(g.A2, bf, "gemm.A2")
            a.A2, a.K1, a.lda2 = g.A2.data_ptr(), g.A.shape[-1], g.A2.stride(-2)
        if g.batch > 1:
            if g.mode != 0:
                raise RuntimeError("gemm: batching is only defined for LINEAR mode")
            a.sA = g.A.stride(0) if g.A.dim() == 3 else 0
            a.sW = g.W.stride(0) if g.W.dim() == 3 else 0
            a.sO = g.out.stride(0) if g.out.dim() == 3 else 0
            if g.out.dim() != 3:
                raise RuntimeError("gemm: batched out must be 3-D")
        return a

    def gemm(self, g: GemmCall):
        if self._gemm_in_row_chunks(g):
            return
        a = self._gemm_args(g)
        self._check(self.lib.v3d_gemm(C.byref(a), self._stream()), "v3d_gemm")

    def gemm_gn_in_supported(self, g: GemmCall) -> bool:
        """Would v3d_gemm run this call (with its gn_in table / second source) on a kernel that normalises the operand in flight?"""
        if g.gn_in is None or g.A.dim() != 2 or g.A.shape[0] * g.A.stride(0) * 2 > self._MAX_OPERAND_BYTES:
            return False
        a = self._gemm_args(g)
        return bool(self.lib.v3d_gemm_gn_in_supported(C.byref(a)))

    def groupnorm_stats(self, x1, x2, stats, n_img, S, groups, imgs_per_stat):
        bf = torch.bfloat16
        self._req_c(x1, bf, "gn.x1")
        if x2 is not None:
            self._req_c(x2, bf, "gn.x2")
        self._req_c(stats, torch.float32, "gn.stats")
        self._check(self.lib.v3d_groupnorm_stats(x1.data_ptr(), x1.shape[-1], _ptr(x2), 0 if x2 is None else x2.shape[-1],
                                                 stats.data_ptr(), stats.shape[1], n_img, S, groups, imgs_per_stat, self._stream()),
                    "v3d_groupnorm_stats")

    def groupnorm_stats_table(self, x1, x2, stats, tickets, n_img, S, groups, imgs_per_stat, gamma, beta, count, eps, table):
        """Statistics + (scale, shift) table in ONE launch (ABI 5): the last block of each statistics group folds the group's slots."""
        bf, f32 = torch.bfloat16, torch.float32
        self._req_c(x1, bf, "gn.x1")
        if x2 is not None:
            self._req_c(x2, bf, "gn.x2")
        self._req_c(stats, f32, "gn.stats"); self._req_c(gamma, f32, "gn.gamma"); self._req_c(beta, f32, "gn.beta"); self._req_c(table, f32, "gn.table")
        if tickets.dtype != torch.int32 or tickets.numel() < n_img // imgs_per_stat or not tickets.is_contiguous():
            raise ValueError("gn.tickets: need a contiguous int32 tensor with one (zero) entry per statistics group")
        self._check(self.lib.v3d_groupnorm_stats_table(x1.data_ptr(), x1.shape[-1], _ptr(x2), 0 if x2 is None else x2.shape[-1], stats.data_ptr(), stats.shape[1],
                                                       tickets.data_ptr(), n_img, S, groups, imgs_per_stat, gamma.data_ptr(), beta.data_ptr(), float(count), float(eps),
                                                       table.data_ptr(), self._stream()), "v3d_groupnorm_stats_table")

    def groupnorm_finalize(self, stats, sums, gamma, beta, count, eps, table):
        """stats [n_stat, nslots, groups, 2] fp32 (or None: read `sums`), sums [n_stat, groups, 2] fp64 (or None), table [n_stat, C, 2] fp32 (or None)."""
        f32 = torch.float32
        if stats is not None:
            self._req_c(stats, f32, "gn.stats")
        if sums is not None:
            self._req_c(sums, torch.float64, "gn.sums")
        ref = stats if stats is not None else sums
        n_stat, groups = ref.shape[0], ref.shape[-2]
        Cc = 0
        if table is not None:
            self._req_c(gamma, f32, "gn.gamma"); self._req_c(beta, f32, "gn.beta"); self._req_c(table, f32, "gn.table")
            Cc = gamma.numel()
        self._check(self.lib.v3d_groupnorm_finalize(_ptr(stats), 0 if stats is None else stats.shape[1], _ptr(sums), n_stat, groups,
                                                    _ptr(gamma) if table is not None else None, _ptr(beta) if table is not None else None, Cc,
                                                    float(count), float(eps), _ptr(table), self._stream()), "v3d_groupnorm_finalize")

    def groupnorm_small_supported(self, C1, C2, S, imgs_per_stat=1, groups=32):
        return bool(self.lib.v3d_groupnorm_small_supported(C1, C2, S, groups, imgs_per_stat))

    def groupnorm_small(self, x1, x2, gamma, beta, out, n_img, S, *, eps, silu, imgs_per_stat=1, groups=32):
        bf = torch.bfloat16
        self._req(x1, bf, "gns.x1"); self._req_c(gamma, torch.float32, "gns.gamma"); self._req_c(beta, torch.float32, "gns.beta"); self._req_c(out, bf, "gns.out")
        C1, C2 = x1.shape[-1], 0
        assert x1.stride(-1) == 1 and x1.stride(0) == C1
        if x2 is not None:
            self._req(x2, bf, "gns.x2")
            C2 = x2.shape[-1]
            assert x2.stride(-1) == 1 and x2.stride(0) == C2
        self._check(self.lib.v3d_groupnorm_small(x1.data_ptr(), C1, _ptr(x2), C2, gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), n_img, S, groups,
                                                 imgs_per_stat, eps, int(silu), self._stream()), "v3d_groupnorm_small")
        return out

    def groupnorm_apply(self, x1, x2, table, out, n_img, S, imgs_per_stat, silu):
        bf, f32 = torch.bfloat16, torch.float32
        self._req_c(x1, bf, "gn.x1")
        if x2 is not None:
            self._req_c(x2, bf, "gn.x2")
        self._req_c(table, f32, "gn.table")
        self._req_c(out, bf, "gn.out")
        self._check(self.lib.v3d_groupnorm_apply(x1.data_ptr(), x1.shape[-1], _ptr(x2), 0 if x2 is None else x2.shape[-1],
                                                 table.data_ptr(), out.data_ptr(), n_img, S, imgs_per_stat, int(silu),
                                                 self._stream()), "v3d_groupnorm_apply")

    def layernorm(self, x, gamma, beta, out, eps, add=None, add_rpg=0, add_ld=0, xsum_out=None):
        bf, f32 = torch.bfloat16, torch.float32
        self._req_c(x, bf, "ln.x"); self._req_c(out, bf, "ln.out")
        self._req_c(gamma, f32, "ln.gamma"); self._req_c(beta, f32, "ln.beta")
        if add is not None:
            self._req(add, f32, "ln.add")
        if xsum_out is not None:
            self._req_c(xsum_out, bf, "ln.xsum_out")
        M, Cc = x.numel() // x.shape[-1], x.shape[-1]
        self._check(self.lib.v3d_layernorm(x.data_ptr(), _ptr(add), add_rpg, add_ld, _ptr(xsum_out), gamma.data_ptr(),
                                           beta.data_ptr(), out.data_ptr(), M, Cc, float(eps), self._stream()), "v3d_layernorm")

    def attn_spatial(self, q, k, vT, out, n_img, S, heads, scale):
        bf = torch.bfloat16
        self._req(q, bf, "attn.q"); self._req(k, bf, "attn.k"); self._req_c(vT, bf, "attn.vT"); self._req(out, bf, "attn.out")
        self._check(self.lib.v3d_attn_spatial(q.data_ptr(), q.stride(-2), k.data_ptr(), k.stride(-2), vT.data_ptr(),
                                              out.data_ptr(), out.stride(-2), n_img, S, heads, float(scale), self._stream()),
                    "v3d_attn_spatial")

    def attn_temporal(self, q, k, v, out, heads, scale):
        """q/out: [B, Tq, S, C] views, k/v: [B, Tk, S, C] views (any strides with unit inner stride)."""
        bf = torch.bfloat16
        for t, nm in ((q, "q"), (k, "k"), (v, "v"), (out, "out")):
            self._req(t, bf, f"tattn.{nm}")
        B, Tq, S, _ = q.shape
        Tk = k.shape[1]
        assert k.stride()[:3] == v.stride()[:3], "k and v must share strides"
        self._check(self.lib.v3d_attn_temporal(q.data_ptr(), q.stride(0), q.stride(1), q.stride(2),
                                               k.data_ptr(), v.data_ptr(), k.stride(0), k.stride(1), k.stride(2),
                                               out.data_ptr(), out.stride(0), out.stride(1), out.stride(2),
                                               B, Tq, Tk, S, heads, float(scale), self._stream()), "v3d_attn_temporal")

    ATTN_VAE_WIDTHS = (128, 256, 512)

    def attn_vae(self, q, k, vT, bias, out, n_img, S, C, scale):
        """Single-head attention of the VAE AttnBlock: q / k [n_img * S, C] views, vT [n_img, C, S], bias [C] fp32 or None."""
        bf = torch.bfloat16
        self._req(q, bf, "attn_vae.q"); self._req(k, bf, "attn_vae.k"); self._req_c(vT, bf, "attn_vae.vT"); self._req(out, bf, "attn_vae.out")
        if bias is not None:
            self._req_c(bias, torch.float32, "attn_vae.bias")
        self._check(self.lib.v3d_attn_vae_d512(q.data_ptr(), q.stride(-2), k.data_ptr(), k.stride(-2), vT.data_ptr(), _ptr(bias), out.data_ptr(),
                                               out.stride(-2), n_img, S, C, float(scale), self._stream()), "v3d_attn_vae_d512")

    # ---- fp8 attention (scene config; never used by the headline benchmark) ------------------------------
    def quant_fp8_tiles(self, x, n_img, S):
        """x [n_img * S, ncols] bf16 view -> (x8 [n_img * S, ncols] uint8 e4m3 bytes, scales [n_img, ceil(S / 64), ncols / 64] fp32)."""
        self._req(x, torch.bfloat16, "quant_fp8_tiles.x")
        ncols = x.shape[-1]
        x8 = torch.empty((n_img * S, ncols), dtype=torch.uint8, device=x.device)
        scales = torch.empty((n_img, (S + 63) // 64, ncols // 64), dtype=torch.float32, device=x.device)
        self._check(self.lib.v3d_quant_fp8_tiles(x.data_ptr(), x.stride(-2), x8.data_ptr(), ncols, scales.data_ptr(), n_img, S, ncols, self._stream()),
                    "v3d_quant_fp8_tiles")
        return x8, scales

    def quant_fp8_slab(self, vT, heads):
        """vT [n_img, heads * 64, S] bf16 -> (v8 uint8 e4m3 bytes of the same shape, vscale [n_img, heads] fp32)."""
        self._req_c(vT, torch.bfloat16, "quant_fp8_slab.vT")
        n_img, _, S = vT.shape
        v8 = torch.empty(vT.shape, dtype=torch.uint8, device=vT.device)
        vscale = torch.empty((n_img, heads), dtype=torch.float32, device=vT.device)
        scratch = torch.empty((n_img, heads), dtype=torch.int32, device=vT.device)
        self._check(self.lib.v3d_quant_fp8_slab(vT.data_ptr(), v8.data_ptr(), vscale.data_ptr(), scratch.data_ptr(), n_img, S, heads, self._stream()),
                    "v3d_quant_fp8_slab")
        return v8, vscale

    def attn_spatial_fp8(self, qk8, scales, v8, vscale, out, n_img, S, heads, scale):
        self._req_c(qk8, torch.uint8, "attn_fp8.qk8"); self._req_c(v8, torch.uint8, "attn_fp8.v8")
        self._req_c(scales, torch.float32, "attn_fp8.scales"); self._req_c(vscale, torch.float32, "attn_fp8.vscale")
        self._req(out, torch.bfloat16, "attn_fp8.out")
        self._check(self.lib.v3d_attn_spatial_fp8(qk8.data_ptr(), qk8.stride(-2), scales.data_ptr(), v8.data_ptr(), vscale.data_ptr(), out.data_ptr(),
                                                  out.stride(-2), n_img, S, heads, float(scale), self._stream()), "v3d_attn_spatial_fp8")

    def softmax_rows(self, inp, out):
        self._req_c(inp, torch.float32, "softmax.in"); self._req_c(out, torch.bfloat16, "softmax.out")
        L = inp.shape[-1]
        self._check(self.lib.v3d_softmax_rows(inp.data_ptr(), out.data_ptr(), inp.numel() // L, L, self._stream()), "v3d_softmax_rows")

    def timestep_embedding(self, t, dim, max_period=10000.0):
        self._req_c(t, torch.float32, "temb.t")
        out = self.empty((t.numel(), dim), torch.bfloat16, t.device)
        self._check(self.lib.v3d_timestep_embedding(t.data_ptr(), out.data_ptr(), t.numel(), dim, float(max_period), self._stream()),
                    "v3d_timestep_embedding")
        return out

    def silu_add(self, a, b=None):
        self._req_c(a, torch.float32, "silu.a")
        if b is not None:
            self._req_c(b, torch.float32, "silu.b")
        out = self.empty(a.shape, torch.bfloat16, a.device)
        self._check(self.lib.v3d_silu_add(a.data_ptr(), _ptr(b), out.data_ptr(), a.numel(), self._stream()), "v3d_silu_add")
        return out

    def edm_scalings(self, sigma):
        self._req_c(sigma, torch.float32, "edm.sigma")
        outs = [torch.empty_like(sigma) for _ in range(4)]
        self._check(self.lib.v3d_edm_scalings(sigma.data_ptr(), *[o.data_ptr() for o in outs], sigma.numel(), self._stream()),
                    "v3d_edm_scalings")
        return tuple(outs)

    def pack_input(self, x, scale, cond, Cpad):
        f32 = torch.float32
        self._req_c(x, f32, "pack.x")
        n, C1 = x.shape[0], x.shape[1]
        S = x.numel() // (n * C1)
        C2 = 0
        if cond is not None:
            self._req_c(cond, f32, "pack.cond")
            C2 = cond.shape[1]
        if scale is not None:
            self._req_c(scale, f32, "pack.scale")
        out = self.empty((n * S, Cpad), torch.bfloat16, x.device)
        self._check(self.lib.v3d_pack_input(x.data_ptr(), _ptr(scale), C1, _ptr(cond), C2, out.data_ptr(), n, S, Cpad, self._stream()),
                    "v3d_pack_input")
        return out

    def pack_input_im2col3x3(self, x, scale, cond, Kpad):
        f32 = torch.float32
        self._req_c(x, f32, "im2col.x")
        n, C1, H, W = x.shape
        C2 = 0
        if cond is not None:
            self._req_c(cond, f32, "im2col.cond")
            C2 = cond.shape[1]
        if scale is not None:
            self._req_c(scale, f32, "im2col.scale")
        out = self.empty((n * H * W, Kpad), torch.bfloat16, x.device)
        self._check(self.lib.v3d_pack_input_im2col3x3(x.data_ptr(), _ptr(scale), C1, _ptr(cond), C2, out.data_ptr(), n, H, W, Kpad, self._stream()),
                    "v3d_pack_input_im2col3x3")
        return out

    def tapsum3x3(self, y, bias, n, H, W, C):
        f32 = torch.float32
        self._req(y, f32, "tapsum.y")
        assert y.shape[0] == n * H * W and y.stride(-1) == 1
        if bias is not None:
            self._req_c(bias, f32, "tapsum.bias")
        out = self.empty((n * H * W, C), f32, y.device)
        self._check(self.lib.v3d_tapsum3x3(y.data_ptr(), y.stride(0), _ptr(bias), out.data_ptr(), n, H, W, C, self._stream()), "v3d_tapsum3x3")
        return out

    def denoise_combine(self, net, x, c_out, c_skip):
        f32 = torch.float32
        self._req(net, f32, "dc.net"); self._req_c(x, f32, "dc.x"); self._req_c(c_out, f32, "dc.c_out"); self._req_c(c_skip, f32, "dc.c_skip")
        n, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (n * Cc)
        out = torch.empty_like(x)
        self._check(self.lib.v3d_denoise_combine(net.data_ptr(), net.stride(-2), x.data_ptr(), c_out.data_ptr(), c_skip.data_ptr(),
                                                 out.data_ptr(), n, Cc, S, self._stream()), "v3d_denoise_combine")
        return out

    def cfg_combine(self, x, scale, T):
        f32 = torch.float32
        self._req_c(x, f32, "cfg.x"); self._req_c(scale, f32, "cfg.scale")
        n = x.shape[0] // 2
        chw = x.numel() // x.shape[0]
        out = torch.empty((n,) + tuple(x.shape[1:]), dtype=f32, device=x.device)
        self._check(self.lib.v3d_cfg_combine(x.data_ptr(), scale.data_ptr(), out.data_ptr(), n, T, chw, self._stream()), "v3d_cfg_combine")
        return out

    def euler_step(self, x, den, sigma, next_sigma):
        f32 = torch.float32
        for t, nm in ((x, "x"), (den, "den"), (sigma, "sigma"), (next_sigma, "next")):
            self._req_c(t, f32, f"euler.{nm}")
        out = torch.empty_like(x)
        n = x.shape[0]
        self._check(self.lib.v3d_euler_step(x.data_ptr(), den.data_ptr(), sigma.data_ptr(), next_sigma.data_ptr(), out.data_ptr(),
                                            n, x.numel() // n, self._stream()), "v3d_euler_step")
        return out

    def ff_fused(self, x, w1p, b1, w2p, b2, out, *, res1=None, res2=None, coef=None, coef_rpg=0, c_acc=1.0, c_res1=1.0, c_res2=1.0):
        bf = torch.bfloat16
        for t, nm in ((x, "x"), (w1p, "w1"), (w2p, "w2"), (out, "out")):
            self._req(t, bf, f"ff.{nm}")
        M, Cc = x.shape
        hidden = w2p.shape[-1]
        self._check(self.lib.v3d_ff_fused(x.data_ptr(), x.stride(0), w1p.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(),
                                          _ptr(res1), res1.stride(0) if res1 is not None else 0, _ptr(res2),
                                          res2.stride(0) if res2 is not None else 0, _ptr(coef), int(coef_rpg), float(c_acc),
                                          float(c_res1), float(c_res2), out.data_ptr(), out.stride(0), M, Cc, hidden, self._stream()),
                    "v3d_ff_fused")
        return out

    def ln_ff_fused(self, x, eps, w1p, b1, w2p, b2, out, *, res1=None, res2=None, coef=None, coef_rpg=0, c_acc=1.0, c_res1=1.0, c_res2=1.0):
        """LayerNorm (affine folded into w1p / b1 at pack time) + the fused feed-forward on the un-normalised rows x."""
        bf = torch.bfloat16
        for t, nm in ((x, "x"), (w1p, "w1"), (w2p, "w2"), (out, "out")):
            self._req(t, bf, f"ln_ff.{nm}")
        M, Cc = x.shape
        hidden = w2p.shape[-1]
        self._check(self.lib.v3d_ln_ff_fused(x.data_ptr(), x.stride(0), float(eps), w1p.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(),
                                             _ptr(res1), res1.stride(0) if res1 is not None else 0, _ptr(res2),
                                             res2.stride(0) if res2 is not None else 0, _ptr(coef), int(coef_rpg), float(c_acc),
                                             float(c_res1), float(c_res2), out.data_ptr(), out.stride(0), M, Cc, hidden, self._stream()),
                    "v3d_ln_ff_fused")
        return out

    LN_PROJ_WIDTHS = (320,)

    def ln_proj(self, x, eps, wp, bias, n_rm, S):
        """(x - mean) * rstd per row, then @ wp^T + bias for the concatenated q | k | v weight with the LayerNorm affine folded in (DMA-tiled,
        packing.ln_proj_pack): returns (out [M, n_rm] or None, outT [M / S, N - n_rm, S] or None)."""
        bf, f32 = torch.bfloat16, torch.float32
        self._req(x, bf, "ln_proj.x"); self._req_c(wp, bf, "ln_proj.w"); self._req_c(bias, f32, "ln_proj.bias")
        M, Cc = x.shape
        N = wp.shape[0]
        out = self.empty((M, n_rm), bf, x.device) if n_rm else None
        outT = self.empty((M // S, N - n_rm, S), bf, x.device) if n_rm < N else None
        self._check(self.lib.v3d_ln_proj(x.data_ptr(), x.stride(0), float(eps), wp.data_ptr(), bias.data_ptr(), _ptr(out),
                                         n_rm if out is not None else 0, _ptr(outT), M, Cc, N, n_rm, S, self._stream()), "v3d_ln_proj")
        return out, outT

    def heun_step(self, x, den, euler, den2, sigma, next_sigma):
        f32 = torch.float32
        for t, nm in ((x, "x"), (den, "den"), (euler, "euler"), (den2, "den2"), (sigma, "sigma"), (next_sigma, "next")):
            self._req_c(t, f32, f"heun.{nm}")
        out = torch.empty_like(x)
        n = x.shape[0]
        self._check(self.lib.v3d_heun_step(x.data_ptr(), den.data_ptr(), euler.data_ptr(), den2.data_ptr(), sigma.data_ptr(),
                                           next_sigma.data_ptr(), out.data_ptr(), n, x.numel() // n, self._stream()), "v3d_heun_step")
        return out

    def clip_preprocess(self, img, size, patch, antialias, mean, std, kpad):
        """img [B, 3, H, W] fp32 in [-1, 1] -> normalised, resized, unfolded patches [B * (size // patch)^2, kpad] bf16."""
        self._req_c(img, torch.float32, "clip_preprocess.img")
        B, Cc, H, W = img.shape
        if Cc != 3:
            raise RuntimeError("clip_preprocess: expected 3 channels")
        g = size // patch
        out = self.empty((B * g * g, kpad), torch.bfloat16, img.device)
        m3 = (C.c_float * 3)(*[float(v) for v in mean])
        s3 = (C.c_float * 3)(*[float(v) for v in std])
        self._check(self.lib.v3d_clip_preprocess(img.data_ptr(), B, H, W, size, patch, int(bool(antialias)), C.cast(m3, C.c_void_p),
                                                 C.cast(s3, C.c_void_p), out.data_ptr(), kpad, self._stream()), "v3d_clip_preprocess")
        return out

    def frames_to_uint8(self, x):
        """x [n, C, H, W] fp32 in [-1, 1] -> [n, H, W, C] uint8 (device tensor)."""
        self._req_c(x, torch.float32, "frames_to_uint8.x")
        n, Cc, H, W = x.shape
        out = torch.empty((n, H, W, Cc), dtype=torch.uint8, device=x.device)
        self._check(self.lib.v3d_frames_to_uint8(x.data_ptr(), out.data_ptr(), n, Cc, H * W, self._stream()), "v3d_frames_to_uint8")
        return out

    def gelu(self, x, out=None):
        self._req_c(x, torch.bfloat16, "gelu.x")
        out = torch.empty_like(x) if out is None else out
        self._check(self.lib.v3d_gelu_bf16(x.data_ptr(), out.data_ptr(), x.numel(), self._stream()), "v3d_gelu_bf16")
        return out

    def axpb_f32(self, x, a, b=0.0, out=None):
        self._req_c(x, torch.float32, "axpb.x")
        if out is None:
            out = torch.empty_like(x)
        self._check(self.lib.v3d_axpb_f32(x.data_ptr(), float(a), float(b), out.data_ptr(), x.numel(), self._stream()), "v3d_axpb_f32")
        return out

    def blend_coefs(self, alpha, kind, ioi, n_img):
        self._req_c(alpha, torch.float32, "blend.alpha"); self._req_c(kind, torch.int32, "blend.kind")
        if ioi is not None:
            self._req_c(ioi, torch.float32, "blend.ioi")
        nm = alpha.numel()
        out = self.empty((nm, n_img, 3), torch.float32, alpha.device)
        self._check(self.lib.v3d_blend_coefs(alpha.data_ptr(), kind.data_ptr(), _ptr(ioi), out.data_ptr(), nm, n_img, self._stream()),
                    "v3d_blend_coefs")
        return out

    def nchw_to_nhwc_bf16(self, x, scale, Cpad):
        self._req_c(x, torch.float32, "nchw.x")
        n, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (n * Cc)
        out = self.empty((n * S, Cpad), torch.bfloat16, x.device)
        self._check(self.lib.v3d_nchw_to_nhwc_bf16(x.data_ptr(), float(scale), out.data_ptr(), n, Cc, S, Cpad, self._stream()),
                    "v3d_nchw_to_nhwc_bf16")
        return out

    def tmix_small(self, x, w, b, B, T, S, Cc, tmin, tmax, row0=0):
        f32 = torch.float32
        self._req(x, f32, "tmix.x"); self._req_c(w, f32, "tmix.w"); self._req_c(b, f32, "tmix.b")
        out = self.empty((B * T, Cc, S), f32, x.device)
        self._check(self.lib.v3d_tmix_small(x.data_ptr(), x.stride(-2), w.data_ptr(), b.data_ptr(), out.data_ptr(), B, T, S, Cc,
                                            tmin, tmax, row0, self._stream()), "v3d_tmix_small")
        return out

    def copy2d_bf16(self, src, dst):
        bf = torch.bfloat16
        self._req(src, bf, "copy.src"); self._req(dst, bf, "copy.dst")
        rows, Cc = src.shape
        self._check(self.lib.v3d_copy2d_bf16(src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), rows, Cc, self._stream()),
                    "v3d_copy2d_bf16")
        return dst
