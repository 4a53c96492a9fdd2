"""Frame sharding of one sample over the GPUs of a node: one process per GPU, torch.distributed
("nccl" == RCCL over xGMI on the box, "gloo" in the CPU tests).

The image batch (cfg x B x T) is split into contiguous frame ranges per rank; both cfg halves and all B samples of a
frame range stay on one rank, weights are replicated.  Everything 2-D (GroupNorm-2D, conv3x3, linears, spatial attention,
sampler / guider elementwise) is local.  The frame axis couples ranks in exactly three places:
  (i)   temporal self-attention          -> all-gather of K|V along frames            (FrameShard.allgather_frames)
  (ii)  (3,1,1) temporal convolution     -> +-1 frame halo with the ring neighbours   } ONE grouped point-to-point call per norm + conv
  (iii) 3-D GroupNorm statistics         -> every rank's (sum, sumsq) to every rank   } (FrameShard.exchange_halo_and_sums)
The reference has no distributed code on this path; this module is new design, verified against the unsharded result.

Data movement is exact-size and copy-free, and every exchange is ONE grouped point-to-point launch (xGMI is a full point-to-point mesh:
one direct link per peer; no ring collective on the evaluation path):
  * (i) every rank's K|V block lands straight in its frame rows of the [B, T_global, S, 2C] buffer the attention kernel reads with
    strides - uneven shards (18 frames over 8 ranks = 3,3,2,...) move exactly their own bytes (no padding to the largest shard, no trim
    copy).  It is issued asynchronously right after the temporal to_k / to_v GEMM; the to_q GEMM runs while it is in flight.
  * (ii) + (iii) (round 3): the spatial half of a VideoResBlock writes its output into the middle of a persistent split-halo buffer
    [B*S | B*T_local*S | B*S] (v3d_gemm halo_rows layout); its RAW first / last local frames go to the ring neighbours and this rank's
    fp64 (sum, sumsq) table of the 3-D GroupNorm goes to every rank in the SAME grouped call; the tables are added in rank order (same bits
    on every rank), the local frames and the received halo frames are normalised with the all-rank statistics, ONE 3-tap GEMM covers all B
    samples.  Rounds 1-2 issued a blocking exchange of normalised halos plus a blocking all-reduce per norm: 176 serialized small
    collectives per U-Net evaluation, now 44 (+ 16 K|V exchanges = 60 grouped calls per evaluation; `FrameShard.counters()`).
Budget at V3D_512, 8 ranks (SURVEY 8e derives 832 MB per evaluation over the node): K|V 16 blocks x (2 x 18 x S x 2C) bf16 = 755 MB
leave the ranks in total (7/8 of it crosses links), halos 44 x 2 x 2 x S x C bf16 per interior rank = 69 MB in total, statistics 44 x 56 KB.
Only the K|V exchange is overlapped with compute (the q projection); the 44 halo + statistics calls are latency-exposed (they carry
the statistics the next kernel needs): 44 x ~20-30 us ~ 1 ms per ~8 ms evaluation at 8 ranks, i.e. <= 15 % un-overlapped.
`HybridShard` is the cfg-parallel x frame-shard layout (2 x F ranks): half the K|V volume per rank, F-way instead of 2F-way exchanges,
one 131 KB pair swap per evaluation.
`sharded_sample` runs the whole sampler loop with x sharded (25 steps), decodes the local frames and gathers only the decoded
frames; `bench.py --gpus N` reports it as the frame-sharded (strong-scaling, latency) mode next to the replica mode.
NOTHING here has been timed on RCCL: the build boxes have one GPU (tests: gloo on CPU with the emulator, and two processes on one GPU
with host-staged exchanges on the HIP kernels).
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
    """One rank's view of a frame-sharded evaluation.  `ranks` (global ranks, in frame order) restricts the shard to a subset of the
    job - the cfg-parallel x frame-shard layout runs two such groups side by side (HybridShard); every exchange is a grouped
    point-to-point call between explicit peers, so no sub-communicator is needed."""

    def __init__(self, T_global: int, group: Optional[dist.ProcessGroup] = None, ranks: Optional[Sequence[int]] = None):
        self.group = group
        if ranks is None:
            self.ranks = [r if group is None else dist.get_global_rank(group, r) for r in range(dist.get_world_size(group))]
            self.rank = dist.get_rank(group)
        else:
            self.ranks = list(ranks)
            self.group = None
            self.rank = self.ranks.index(dist.get_rank())
        self._setup(T_global)

    def _setup(self, T_global: int):
        self.world = len(self.ranks)
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
        self.n_exchanges = 0                                   # grouped point-to-point calls (one NCCL group launch each)
        self.n_allreduce = 0                                   # (none on the evaluation path since round 3: sums ride on the halo exchange)

    def describe(self) -> str:
        return "+".join(str(len(p)) for p in self.parts)

    def counters(self) -> Dict[str, int]:
        return {"bytes_sent": self.bytes_sent, "grouped_p2p_calls": self.n_exchanges, "all_reduces": self.n_allreduce}

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
            if prev is not self:
                self._bufs.clear()       # the split-halo buffers (up to ~1 GB at the decoder's 512 x 512 levels) live for one sharded run, not for the process

    # ---- communication primitives (a test subclass stages them through the host) -------------------------
    def _peer(self, r: int) -> int:
        """shard rank -> rank of the process group the exchanges run on (subclasses add addresses outside the shard, e.g. PARTNER)"""
        return self.ranks[r]

    def _allreduce_sum(self, t: torch.Tensor) -> None:
        self.n_allreduce += 1
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def _exchange(self, sends: Sequence[Tuple[torch.Tensor, int]], recvs: Sequence[Tuple[torch.Tensor, int]], async_op: bool = False) -> _Handle:
        """Grouped point-to-point exchange: (contiguous tensor, shard rank) pairs.  Messages between one pair of ranks match in
        list order on both sides.  One call = one NCCL group launch, whatever the number of messages."""
        ops_ = [dist.P2POp(dist.isend, t, self._peer(r)) for t, r in sends]
        ops_ += [dist.P2POp(dist.irecv, t, self._peer(r)) for t, r in recvs]
        self.bytes_sent += sum(t.numel() * t.element_size() for t, _ in sends)
        self.n_exchanges += 1 if ops_ else 0
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
    def allreduce_stats(self, sums: torch.Tensor) -> torch.Tensor:
        """Stand-alone reduction of the fp64 (sum, sumsq) table over the shard's ranks (decoder norms without a halo to ride on)."""
        return self.exchange_halo_and_sums(None, 0, 0, sums)[1]

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
        """Persistent split-halo activation buffer [(B + B*T_local + B) * S, C] and its middle (local frames) view: the producer writes the
        view, the neighbours' boundary frames are received into the outer slabs.  One buffer per (shape, slot): a VideoResBlock uses slot 0
        for the raw block input, slot 1 for normalised operands, slot 2 for the raw intermediate.  The outer slabs start zeroed: at the
        global ends nothing is ever received there, and although the 3-tap GEMM masks those taps (tmin / tmax), no consumer may meet
        uninitialised memory."""
        key = (B, S, C, dtype, str(device), slot)
        buf = self._bufs.get(key)
        if buf is None:
            buf = self._bufs[key] = torch.empty(((B + B * self.T_local + B) * S, C), dtype=dtype, device=device)
            buf[:B * S].zero_()
            buf[(B + B * self.T_local) * S:].zero_()
        return buf, buf[B * S:(B + B * self.T_local) * S]

    def exchange_halo_and_sums(self, buf: Optional[torch.Tensor], B: int, S: int, sums: Optional[torch.Tensor] = None,
                               async_op: bool = False):
        """ONE grouped point-to-point call per temporal norm + convolution:
          * the +-1 frame halos of the split-halo buffer `buf` (raw, un-normalised rows: they leave as soon as their producer has
            written them and are normalised on arrival with the all-rank statistics) - one contiguous message per neighbour: the first /
            last local frame of every sample is packed into a [B, S, C] staging block, the neighbour's block lands directly in the slab;
          * this rank's fp64 (sum, sumsq) table of the 3-D GroupNorm to every other rank of the shard, and theirs back (1 KB each): the
            ranks' tables are then added in RANK ORDER on every rank - the same bits everywhere, no all-reduce launch.
        Rounds 1-2 issued a blocking halo exchange and a blocking all-reduce per norm (176 serialized small collectives per evaluation at
        8 ranks); this is 44 grouped calls per U-Net evaluation, + 16 K|V exchanges = 60.  Returns (handle, summed table or None); with
        async_op the table is only valid after handle.wait()."""
        sends, recvs, after = [], [], []
        if buf is not None and self.world > 1:
            Tl = self.T_local
            mid = buf[B * S:(B + B * Tl) * S].view(B, Tl, S, -1)
            if not self.first:
                sends.append((mid[:, 0].contiguous(), self.rank - 1))                        # my first frames -> previous rank
                recvs.append((buf[:B * S], self.rank - 1))                                   # its last frames -> my frame -1 slab
            if not self.last:
                sends.append((mid[:, Tl - 1].contiguous(), self.rank + 1))                   # my last frames -> next rank
                recvs.append((buf[(B + B * Tl) * S:], self.rank + 1))                        # its first frames -> my frame T_local slab
        total = None
        if sums is not None:
            assert sums.is_contiguous()
            allsums = torch.empty((self.world,) + tuple(sums.shape), dtype=sums.dtype, device=sums.device)
            allsums[self.rank].copy_(sums)
            for r in range(self.world):
                if r != self.rank:
                    sends.append((sums, r))
                    recvs.append((allsums[r], r))
            total = torch.empty_like(sums)

            def reduce_in_rank_order():
                acc = allsums[0].clone()
                for r in range(1, self.world):
                    acc += allsums[r]
                total.copy_(acc)
            after.append(reduce_in_rank_order)
        h = self._exchange(sends, recvs, async_op=True)
        prev_after = h.after

        def done():
            if prev_after is not None:
                prev_after()
            for f in after:
                f()
        h.after = done
        if not async_op:
            h.wait()
        return h, total

    def convt3(self, ops, buf: torch.Tensor, w: torch.Tensor, b, g, **epi):
        """Frame-sharded (3,1,1) conv over a split-halo buffer whose slabs already hold the neighbours' frames: ONE 3-tap GEMM over all B
        samples.  The halo slabs at the global ends are never read (tmin / tmax mask those taps to zero)."""
        B, S, Tl = g.B, g.S, self.T_local
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


class SimFrameShard(FrameShard):
    """MEASUREMENT ONLY (bench.py --shard-sim): one process plays rank `rank` of a `world`-way frame shard with NO communication.  Every
    receive buffer of an exchange is fed from the rank's own data (halo frames <- its own boundary frames, the other ranks' K|V rows <- its own
    rows repeated, the other ranks' fp64 GroupNorm sums <- its own), so the rank launches exactly the kernels, tile counts and buffer sizes
    it would launch inside a real `world`-GPU run - what one GPU of an 8-GPU node COMPUTES per evaluation - while bytes and grouped calls are
    counted as in the real exchange.  The numbers it produces are not a shard of any real sample."""

    def __init__(self, T_global: int, world: int, rank: int):
        self.group = None
        self.ranks = list(range(world))
        self.rank = rank
        self._setup(T_global)

    def _allreduce_sum(self, t: torch.Tensor) -> None:
        self.n_allreduce += 1
        t.mul_(self.world)

    def _exchange(self, sends, recvs, async_op: bool = False) -> _Handle:
        self.bytes_sent += sum(t.numel() * t.element_size() for t, _ in sends)
        self.n_exchanges += 1 if (sends or recvs) else 0
        # every receive buffer is fed from this rank's own send OF THE SAME DTYPE (the bf16 halo from the bf16 halo, the fp64 GroupNorm sums
        # from the fp64 sums: other ranks' statistics = this rank's), repeated / cut to the receive's size
        by_dtype = {}
        for t, _ in sends:
            by_dtype.setdefault(t.dtype, t.reshape(-1))
        for t, _ in recvs:
            src = by_dtype.get(t.dtype)
            if src is None:
                t.zero_()
                continue
            flat = t.reshape(-1) if t.is_contiguous() else None
            n = t.numel()
            rep = src if src.numel() >= n else src.repeat((n + src.numel() - 1) // src.numel())
            if flat is not None:
                flat.copy_(rep[:n])
            else:
                t.copy_(rep[:n].reshape(t.shape))
        return _Handle(())


class HybridShard(FrameShard):
    """cfg-parallel x frame-shard (SURVEY 8e "simpler fall-backs"): the job's 2 F ranks form two frame groups of F ranks - group 0
    evaluates the unconditional half of every guided batch, group 1 the conditional half - so a rank runs B (not 2 B) samples per
    evaluation over an F-way frame partition: half the K|V volume per rank, F- instead of 2F-way halos and statistics exchanges.  The
    price is one pair exchange per evaluation: rank (g, r) swaps its half of the network output (T_local frames x 4 x H x W fp32, 131 KB at
    V3D_512 with F = 4) with rank (1 - g, r), after which both hold [uc ; c] and run the guider / sampler update redundantly.
    Rank layout: global rank = g * F + r."""

    def __init__(self, T_global: int):
        world, rank = dist.get_world_size(), dist.get_rank()
        if world % 2:
            raise ValueError("the cfg-parallel x frame-shard layout needs an even number of ranks")
        F = world // 2
        self.cfg_index = rank // F
        super().__init__(T_global, ranks=list(range(self.cfg_index * F, (self.cfg_index + 1) * F)))
        self.partner = (1 - self.cfg_index) * F + self.rank            # global rank holding the other cfg half of MY frames

    PARTNER = -1      # address of the rank holding the other cfg half of my frames, for the ordinary exchange primitive

    def _peer(self, r: int) -> int:
        return self.partner if r == self.PARTNER else self.ranks[r]

    def describe(self) -> str:
        return "cfg2 x (" + "+".join(str(len(p)) for p in self.parts) + ")"

    def swap_cfg_halves(self, mine: torch.Tensor) -> torch.Tensor:
        """mine [(b T_local), ...] = my half of the guided batch's network output -> [uc rows ; c rows] (guiders.py:95 order)."""
        mine = mine.contiguous()
        other = torch.empty_like(mine)
        self._exchange([(mine, self.PARTNER)], [(other, self.PARTNER)])
        return torch.cat([mine, other] if self.cfg_index == 0 else [other, mine], dim=0)


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
                   image_only_indicator: Optional[torch.Tensor] = None, gather: bool = True, graph: bool = False):
    """One sample (B inputs x T_global frames) with the frame axis sharded for the WHOLE path: every rank keeps its frames of the
    sampler state x for all steps (denoiser evaluations exchange K|V, halo frames and GroupNorm sums inside the U-Net), decodes its
    own frames (VideoDecoder: halos + GroupNorm sums again) and only the decoded frames are gathered.

    noise / c / uc are the FULL tensors [(b T_global), ...], identical on every rank (what scripts/pub/V3D_512.py builds before the
    sampler); `sampler`, `denoiser`, `network` (OpenAIWrapper) are the ordinary plugin objects; decode(z_local) -> frames_local is
    e.g. `lambda z: decoder(z / scale_factor, timesteps=shard.T_local)` (ALL local frames in one call: a chunked decode_first_stage is not
    defined under a FrameShard, run_decoder refuses it).  A HybridShard runs the cfg-parallel x frame-shard layout.  Returns frames [(b T_global), 3, H, W] (every rank) or the local frames."""
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
    smp = local_sampler(sampler, shard)
    if isinstance(shard, HybridShard):
        # cfg-parallel x frame-shard: this rank's frame group evaluates ONE half of the guided batch [uc ; c] (B samples instead of 2 B),
        # the partner rank of the other group the other half; the halves are swapped once per evaluation and the guider / sampler update
        # runs redundantly on both
        h = shard.cfg_index
        extra = {"image_only_indicator": ioi_loc[h * B:(h + 1) * B].contiguous(), "num_video_frames": Tl}
        ctx0 = ctx0[h * B:(h + 1) * B]

        def den(inp, sigma, cc):
            n = inp.shape[0] // 2
            sl = slice(h * n, (h + 1) * n)
            mine = denoiser(network, inp[sl], sigma[sl], {k: (v[sl] if isinstance(v, torch.Tensor) and v.shape[0] == 2 * n else v) for k, v in cc.items()},
                            **extra)
            return shard.swap_cfg_halves(mine)
    else:
        extra = {"image_only_indicator": ioi_loc, "num_video_frames": Tl}

        def den(inp, sigma, cc):
            return denoiser(network, inp, sigma, cc, **extra)
    if graph:
        # HIP-graph replay of the rank's evaluation and decode (engine/graph.py), cached on the shard (one capture per shape).  Measured (bench.py --shard-sim 8
        # --graph, profiles/r06_shard_sim8_graph.log): a rank of an 8-way shard runs ~600 kernels on 4-6 images per evaluation in 20.4 / 18.6 ms (3 / 2 frames)
        # against 20.1 / 21.2 ms eager - the rank is NOT bound by the host's launch rate but by the fixed cost of 600 small kernels on the GPU itself (~33 us each:
        # persistent-kernel prologues, pipeline fill, partial rounds); compute-only ceiling 2.78x vs 2.66x.  SimFrameShard only (self-fed device copies): under a
        # real FrameShard the grouped point-to-point calls would have to be captured too (torch.cuda.graph + NCCL capture) - untested, one GPU per build box.
        from .engine.graph import graphed
        cache = shard.__dict__.setdefault("_graphed", {})
        key = (id(network), id(denoiser), B, Tl, isinstance(shard, HybridShard))
        if key not in cache:
            cache[key] = (graphed(den, enabled=True), graphed(decode, enabled=True))
        den, decode = cache[key]
    with shard.activate(context_frame0=ctx0):
        z = smp(den, x, cond=c_loc, uc=uc_loc)
        frames = decode(z)
    if not gather:
        return frames
    return shard.gather_frames_out(frames.contiguous(), B)
