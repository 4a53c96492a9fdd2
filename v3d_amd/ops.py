"""Operator interface of the V3D hot path.

The engine (v3d_amd/engine/*) is written against the primitive operator set below, whose names and semantics
are exactly the C ABI in include/v3d_hip.h.  The product backend is :class:`v3d_amd.hip.HipOps` (ctypes ->
libv3d_hip.so, hand-written gfx950 kernels).  There is NO CPU/PyTorch fallback in the product: resolving the
default backend without a usable HIP library raises.  Tests inject oracle/ops_emul.py (a torch restatement of
each C-ABI op) through :func:`use_backend` to check the host logic and as the per-op checker on the GPU.
"""
from __future__ import annotations

import os

import contextlib
import dataclasses
from typing import Optional

import torch

GEMM_LINEAR, GEMM_CONV3X3, GEMM_CONVT3 = 0, 1, 2


@dataclasses.dataclass
class GemmCall:
    """Mirror of `v3d_gemm_args` (include/v3d_hip.h) with tensors instead of raw pointers.

    A: 2-D bf16 view [a_rows, >=K] with unit inner stride (row stride = lda).  W: bf16 [taps, N, K] contiguous
    (batched: [batch, taps, N, K] with sW given).  out: 2-D view [M, N_out] (bf16 or fp32, row stride = ldo).
    """
    A: torch.Tensor
    W: torch.Tensor
    out: torch.Tensor
    M: int
    N: int
    K: int
    bias: Optional[torch.Tensor] = None
    add: Optional[torch.Tensor] = None      # fp32, indexed [(m // add_rpg) * add_ld + n]
    add_rpg: int = 0
    add_ld: int = 0
    res1: Optional[torch.Tensor] = None     # bf16 2-D views [M, N_out]
    res2: Optional[torch.Tensor] = None
    coef: Optional[torch.Tensor] = None     # fp32 [groups, 3]
    coef_rpg: int = 0
    c_acc: float = 1.0
    c_res1: float = 1.0
    c_res2: float = 1.0
    mode: int = GEMM_LINEAR
    geglu: bool = False
    # conv3x3 geometry
    Hin: int = 0
    Win: int = 0
    Hout: int = 0
    Wout: int = 0
    stride: int = 1
    up: int = 1
    pad_mode: int = 0        # 0: one zero pixel on every side; 1: right/bottom only (F.pad(0,1,0,1) + padding=0)
    # convt3 geometry
    T: int = 0
    S: int = 0
    tmin: int = 0
    tmax: int = 0
    a_row0: int = 0
    a_rows: int = 0          # 0 -> A.shape[-2]
    batch: int = 1           # batched (LINEAR only): A / W / out may be 3-D [batch, rows, cols]; a 2-D operand is shared
    halo_rows: int = 0       # CONVT3 split-halo layout (frame sharding): B*S halo rows on either side of the M local rows
    gn_stats: Optional[torch.Tensor] = None   # fp32 [M / gn_rps, nslots, 32, 2]: GroupNorm partial sums of the output (epilogue)
    gn_rps: int = 0
    gn_cpg: int = 0
    # GroupNorm (+SiLU) of the INPUT in the operand path: A (| A2, channel-concatenated) hold the raw tensor, gn_in = the (scale, shift)
    # table [n_stat, K, 2] of groupnorm_finalize, one row per gn_in_rps source rows
    gn_in: Optional[torch.Tensor] = None
    gn_in_rps: int = 0
    gn_in_silu: bool = True
    A2: Optional[torch.Tensor] = None


class OpsBase:
    """Primitive op names == C-ABI entry points (without the v3d_ prefix).  Subclasses implement them."""

    name = "base"
    act_dtype = torch.bfloat16   # storage dtype of activations / packed weights (the emulator can run an exact fp32 mode)

    # ---- allocation -------------------------------------------------------------------------------
    def empty(self, shape, dtype=None, device=None):
        return torch.empty(shape, dtype=dtype or self.act_dtype, device=device if device is not None else self.device)

    def zeros(self, shape, dtype=torch.float32, device=None):
        return torch.zeros(shape, dtype=dtype, device=device if device is not None else self.device)

    _ZERO_ARENA_FLOATS = 32 << 20   # 128 MiB: about one network evaluation's worth of GroupNorm partial-sum buffers (slots: one per writer)

    def begin_evaluation(self, device):
        """Called at the top of every network evaluation (U-Net, VAE decoder / encoder, CLIP tower): rewinds the zero-initialised
        scratch arena of `zeros_f32_pooled` and re-zeros the part the previous evaluation dirtied with ONE fill.  The fill is an
        ordinary stream operation, so a HIP-graph capture of the evaluation records it as its first node and every replay starts
        from zeroed GroupNorm partial sums (the first version zeroed a slice once, outside the capture: replays accumulated on top
        of the sums of the capture run).  Slices handed out before the rewind must not be used afterwards - they are
        evaluation-local by construction (statistics live between a stats and an apply kernel)."""
        key = str(device)
        arenas = self.__dict__.setdefault("_zero_arenas", {})
        arena = arenas.get(key)
        if arena is None:
            return
        if torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
            arena[0].zero_()       # the recorded fill must cover whatever the captured evaluation dirties on every replay
        elif arena[1] > 0:
            arena[0][:arena[1]].zero_()
        arena[1] = 0

    def zeros_f32_pooled(self, shape, device):
        """Zero-initialised fp32 scratch carved out of a pre-zeroed arena: one fill kernel per evaluation instead of one per
        request (the per-GroupNorm `torch.zeros` fills were 3000 launches / 14 ms per sample in the rocprof trace).  A slice is
        handed out once per evaluation (`begin_evaluation` rewinds); an exhausted arena is replaced by one at least twice as large (the
        old one is kept alive: slices and captured graphs may still point into it) - except under stream capture, where a new allocation would belong to the
        graph's private pool: there the request falls back to its own `torch.zeros` (a memset node)."""
        n = 1
        for d in shape:
            n *= int(d)
        key = str(device)
        arenas = self.__dict__.setdefault("_zero_arenas", {})
        arena = arenas.get(key)
        if arena is None or arena[1] + n > arena[0].numel():
            if torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
                return torch.zeros(shape, dtype=torch.float32, device=device)
            # grow geometrically, and never drop a retired arena: a HIP graph captured earlier keeps raw pointers into it (its memset
            # node and its kernels would otherwise touch memory the caching allocator has handed to someone else)
            grown = max(n, self._ZERO_ARENA_FLOATS, 0 if arena is None else 2 * arena[0].numel())
            if arena is not None:
                self.__dict__.setdefault("_retired_arenas", []).append(arena[0])
            arena = [torch.zeros(grown, dtype=torch.float32, device=device), 0]
            arenas[key] = arena
        t = arena[0][arena[1]:arena[1] + n].view(shape)
        arena[1] += (n + 63) // 64 * 64
        return t

    @staticmethod
    def gn_nslots(rps, imgs_per_stat=1):
        """Slots per statistics group: one per writer.  Stand-alone statistics kernel: imgs_per_stat x (blocks per image, the library fits
        its grid to what it gets); GEMM epilogues: one per 64+-row wave tile inside the group (+2 for the straddlers)."""
        return max(8 * imgs_per_stat, rps // 64 + 2)

    def gn_stats_buffer(self, n_stat, device, groups=32, *, rps, imgs_per_stat=1):
        """Zeroed partial-sum buffer [n_stat, nslots, groups, 2] of one GroupNorm (v3d_groupnorm_stats / the gn_stats epilogue)."""
        return self.zeros_f32_pooled((n_stat, self.gn_nslots(rps, imgs_per_stat), groups, 2), device)

    # ---- composite helpers shared by every backend ------------------------------------------------
    def linear(self, x2d, w, bias=None, *, out=None, out_dtype=None, geglu=False, **epi):
        """x2d [M, K] (row-strided ok) @ w[N, K]^T with the fused epilogue of v3d_gemm."""
        M, K = x2d.shape
        N = w.shape[-2]
        n_out = N // 2 if geglu else N
        if out is None:
            out = self.empty((M, n_out), out_dtype or self.act_dtype, x2d.device)
        self.gemm(GemmCall(A=x2d, W=w, out=out, M=M, N=N, K=K, bias=bias, geglu=geglu, **epi))
        return out

    def conv3x3(self, x, w, bias, n_img, Hin, Win, *, stride=1, up=1, pad_mode=0, out=None, out_dtype=None, **epi):
        """x [n_img*Hin*Win, Cin] channels-last, w [9, Cout, Cin] -> [n_img*Hout*Wout, Cout]."""
        Hl, Wl = Hin * up, Win * up
        ptot = 1 if pad_mode else 2
        Hout, Wout = (Hl + ptot - 3) // stride + 1, (Wl + ptot - 3) // stride + 1
        K = x.shape[-1]
        N = w.shape[-2]
        M = n_img * Hout * Wout
        if out is None:
            out = self.empty((M, N), out_dtype or self.act_dtype, x.device)
        self.gemm(GemmCall(A=x, W=w, out=out, M=M, N=N, K=K, bias=bias, mode=GEMM_CONV3X3, Hin=Hin, Win=Win,
                           Hout=Hout, Wout=Wout, stride=stride, up=up, pad_mode=pad_mode, **epi))
        return out

    def convt3(self, x, w, bias, T, S, *, tmin=0, tmax=None, a_row0=0, M=None, out=None, out_dtype=None, halo_rows=0, **epi):
        """Temporal 3-tap conv over frames: x [(b t) * S (+halo), C], w [3, Cout, Cin]."""
        K = x.shape[-1]
        N = w.shape[-2]
        if M is None:
            M = x.shape[0]
        if tmax is None:
            tmax = T - 1
        if out is None:
            out = self.empty((M, N), out_dtype or self.act_dtype, x.device)
        self.gemm(GemmCall(A=x, W=w, out=out, M=M, N=N, K=K, bias=bias, mode=GEMM_CONVT3, T=T, S=S, tmin=tmin,
                           tmax=tmax, a_row0=a_row0, halo_rows=halo_rows, **epi))
        return out

    def groupnorm_table(self, x1, x2, gamma, beta, n_img, S, *, eps, imgs_per_stat=1, groups=32, stats_hook=None, count_imgs=None, stats=None):
        """Statistics of GroupNorm over channels-last x1 (| x2) -> the (scale, shift) table [n_stat, C, 2] its normalisation amounts to
        (y = x * scale + shift per (statistics group, channel)): what groupnorm_apply and the operand-path consumers (GemmCall.gn_in) take.

        stats: partial sums already gathered by the epilogue of the GEMM that produced x1 (GemmCall.gn_stats).
        stats_hook(sums fp64 [n_stat, groups, 2]) lets the frame-sharded runtime all-reduce (sum, sumsq) over ranks;
        count_imgs = number of images (global) contributing to one statistics group."""
        C = x1.shape[-1] + (x2.shape[-1] if x2 is not None else 0)
        n_stat = n_img // imgs_per_stat
        if count_imgs is None:
            count_imgs = imgs_per_stat
        count = float(count_imgs) * S * (C // groups)
        table = self.empty((n_stat, C, 2), torch.float32, x1.device)
        if stats is None and stats_hook is None and self._GN_FUSED_TABLE and groups <= 32 and groups % 2 == 0 and hasattr(self, "groupnorm_stats_table"):
            # statistics + table in ONE launch (ABI 5): the last block of a statistics group folds its slots - no finalize launch
            stats = self.gn_stats_buffer(n_stat, x1.device, groups, rps=imgs_per_stat * S, imgs_per_stat=imgs_per_stat)
            tickets = self.zeros_f32_pooled((n_stat,), x1.device).view(torch.int32)
            self.groupnorm_stats_table(x1, x2, stats, tickets, n_img, S, groups, imgs_per_stat, gamma, beta, count, eps, table)
            return table
        if stats is None:
            stats = self.gn_stats_buffer(n_stat, x1.device, groups, rps=imgs_per_stat * S, imgs_per_stat=imgs_per_stat)
            self.groupnorm_stats(x1, x2, stats, n_img, S, groups, imgs_per_stat)
        if stats_hook is None:
            self.groupnorm_finalize(stats, None, gamma, beta, count, eps, table)
        else:
            sums = self.empty((n_stat, groups, 2), torch.float64, x1.device)
            self.groupnorm_finalize(stats, sums, None, None, count, eps, None)
            sums = stats_hook(sums)
            self.groupnorm_finalize(None, sums, gamma, beta, count, eps, table)
        return table

    def groupnorm(self, x1, x2, gamma, beta, n_img, S, *, eps, silu, imgs_per_stat=1, groups=32, stats_hook=None,
                  count_imgs=None, out=None, stats=None, table=None):
        """GroupNorm(+SiLU) over channels-last x1 (and optional channel-concatenated x2): statistics -> table -> apply."""
        C = x1.shape[-1] + (x2.shape[-1] if x2 is not None else 0)
        if out is None:
            out = self.empty((n_img * S, C), self.act_dtype, x1.device)
        if tuple(out.shape) != (n_img * S, C):
            raise ValueError(f"groupnorm: out has shape {tuple(out.shape)}, the normalised tensor is [{n_img * S}, {C}]")
        if (table is None and stats is None and stats_hook is None and (count_imgs is None or count_imgs == imgs_per_stat)
                and self.groupnorm_small_fits(x1, x2, S, imgs_per_stat, groups)):
            # small statistics groups (8 x 8 level, the 16 x 16 level's transformer norms): one launch instead of three launch-bound ones
            return self.groupnorm_small(x1, x2, gamma, beta, out, n_img, S, eps=eps, silu=silu, imgs_per_stat=imgs_per_stat, groups=groups)
        if table is None:
            table = self.groupnorm_table(x1, x2, gamma, beta, n_img, S, eps=eps, imgs_per_stat=imgs_per_stat, groups=groups,
                                         stats_hook=stats_hook, count_imgs=count_imgs, stats=stats)
        self.groupnorm_apply(x1, x2, table, out, n_img, S, imgs_per_stat, silu)
        return out

    _GN_FUSED_TABLE = os.environ.get("V3D_GN_FUSED_TABLE", "1") not in ("", "0")      # A/B knob: 0 = stand-alone statistics pass + finalize launch
    _GN_SMALL = os.environ.get("V3D_GN_SMALL", "1") not in ("", "0")      # A/B knob: 0 = always statistics -> finalize -> apply

    def groupnorm_small_fits(self, x1, x2, S, imgs_per_stat=1, groups=32) -> bool:
        """Would `groupnorm` take the one-launch form for this tensor (given no statistics from a producer and no cross-rank sums)?  Producers
        ask before they spend an epilogue / a statistics pass on sums nobody will read."""
        if not self._GN_SMALL or x1.stride(0) != x1.shape[-1] or (x2 is not None and x2.stride(0) != x2.shape[-1]):
            return False
        return self.groupnorm_small_supported(x1.shape[-1], 0 if x2 is None else x2.shape[-1], S, imgs_per_stat, groups)

    def groupnorm_small_supported(self, C1, C2, S, imgs_per_stat=1, groups=32) -> bool:
        return False

    def gemm_gn_in_supported(self, g: "GemmCall") -> bool:
        """Backends that can normalise a GEMM operand in flight answer per call (HipOps asks the library)."""
        return False


_ACTIVE: Optional[OpsBase] = None
_DEFAULT: Optional[OpsBase] = None


def get_ops() -> OpsBase:
    """Active operator backend.  Default = HIP kernels; raises if libv3d_hip.so / a GPU is unavailable."""
    global _DEFAULT
    if _ACTIVE is not None:
        return _ACTIVE
    if _DEFAULT is None:
        from .hip import HipOps  # raises RuntimeError loudly when the extension cannot be used
        _DEFAULT = HipOps()
    return _DEFAULT


@contextlib.contextmanager
def use_backend(ops: OpsBase):
    """Test hook: temporarily route the engine through another implementation of the op set."""
    global _ACTIVE
    prev = _ACTIVE
    _ACTIVE = ops
    try:
        yield ops
    finally:
        _ACTIVE = prev
