"""Frame sharding of one sample over the GPUs of a node: one process per GPU, torch.distributed
("nccl" == RCCL over xGMI on the box, "gloo" in the CPU tests).

The image batch (cfg x B x T) is split into contiguous frame ranges per rank; both cfg halves and all B samples of a
frame range stay on one rank, weights are replicated.  Everything 2-D (GroupNorm-2D, conv3x3, linears, spatial attention,
sampler / guider elementwise) is local.  The frame axis couples ranks in exactly three places, each with its own exchange:
  (i)   temporal self-attention          -> all-gather of K|V along frames            (FrameShard.allgather_frames)
  (ii)  (3,1,1) temporal convolution     -> +-1 frame halo with the ring neighbours   (FrameShard.convt3 / tmix_small)
  (iii) 3-D GroupNorm statistics         -> all-reduce of (sum, sumsq) per group      (FrameShard.allreduce_stats)
The reference has no distributed code on this path; this module is new design, verified against the unsharded result.
"""
from __future__ import annotations

from typing import List, Optional

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
        self.T_max = max(len(p) for p in self.parts)
        self.first = self.rank == 0
        self.last = self.rank == self.world - 1

    # ---- slicing of replicated inputs ---------------------------------------------------------------
    def take_frames(self, x: torch.Tensor, B: int) -> torch.Tensor:
        """[(b T_global), ...] -> [(b T_local), ...] rows of this rank."""
        shp = x.shape
        x = x.reshape((B, self.T_global) + tuple(shp[1:]))[:, self.t0:self.t0 + self.T_local]
        return x.reshape((B * self.T_local,) + tuple(shp[1:])).contiguous()

    def gather_frames_out(self, x: torch.Tensor, B: int) -> torch.Tensor:
        """Inverse of take_frames for final outputs: [(b T_local), ...] -> [(b T_global), ...] on every rank."""
        shp = x.shape
        g = self.allgather_frames(x.reshape((B, self.T_local) + tuple(shp[1:])))
        return g.reshape((B * self.T_global,) + tuple(shp[1:]))

    # ---- (iii) 3-D GroupNorm statistics -------------------------------------------------------------
    def allreduce_stats(self, stats: torch.Tensor) -> torch.Tensor:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)
        return stats

    # ---- (i) temporal attention ---------------------------------------------------------------------
    def allgather_frames(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, T_local, ...] -> [B, T_global, ...] (uneven shards: padded all-gather to T_max frames, then trimmed)."""
        B = x.shape[0]
        tail = tuple(x.shape[2:])
        if self.T_local < self.T_max:
            pad = torch.zeros((B, self.T_max - self.T_local) + tail, dtype=x.dtype, device=x.device)
            xs = torch.cat([x, pad], dim=1)
        else:
            xs = x
        xs = xs.contiguous()
        buf = torch.empty((self.world * xs.shape[0],) + tuple(xs.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(buf, xs, group=self.group)      # concatenated along dim 0: [world * B, T_max, ...]
        buf = buf.reshape((self.world,) + tuple(xs.shape))
        return torch.cat([buf[r][:, :len(p)] for r, p in enumerate(self.parts)], dim=1).contiguous()

    # ---- (ii) temporal conv halos -------------------------------------------------------------------
    def _halo_buffer(self, x: torch.Tensor, B: int, S: int) -> torch.Tensor:
        """x [(b T_local) * S, C] -> [B, T_local + 2, S, C] with the neighbours' boundary frames in slots 0 and -1
        (zeros at the global ends)."""
        C = x.shape[-1]
        Tl = self.T_local
        buf = torch.zeros((B, Tl + 2, S, C), dtype=x.dtype, device=x.device)
        xv = x.reshape(B, Tl, S, C)
        buf[:, 1:Tl + 1] = xv
        ops_, recv_l, recv_r = [], None, None
        send_first = xv[:, 0].contiguous()
        send_last = xv[:, Tl - 1].contiguous()
        peer = (lambda r: r) if self.group is None else (lambda r: dist.get_global_rank(self.group, r))
        if not self.first:
            recv_l = torch.empty_like(send_first)
            ops_.append(dist.P2POp(dist.isend, send_first, peer(self.rank - 1), self.group))
            ops_.append(dist.P2POp(dist.irecv, recv_l, peer(self.rank - 1), self.group))
        if not self.last:
            recv_r = torch.empty_like(send_last)
            ops_.append(dist.P2POp(dist.isend, send_last, peer(self.rank + 1), self.group))
            ops_.append(dist.P2POp(dist.irecv, recv_r, peer(self.rank + 1), self.group))
        if ops_:
            for req in dist.batch_isend_irecv(ops_):
                req.wait()
        if recv_l is not None:
            buf[:, 0] = recv_l
        if recv_r is not None:
            buf[:, Tl + 1] = recv_r
        return buf

    def convt3(self, ops, h: torch.Tensor, w: torch.Tensor, b, g, *, add=None, add_rpg=0, add_ld=0, res1=None, coef=None,
               coef_rpg=0, c_acc=1.0, c_res1=1.0):
        """Frame-sharded (3,1,1) conv: exchange +-1 frame halos, then one 3-tap GEMM per sample over its halo'd frames."""
        B, S, Tl = g.B, g.S, self.T_local
        buf = self._halo_buffer(h, B, S)
        N = w.shape[-2]
        out = ops.empty((B * Tl * S, N), None, h.device)
        tmin = 0 if self.first else -1
        tmax = Tl - 1 if self.last else Tl
        for bi in range(B):
            rows = slice(bi * Tl * S, (bi + 1) * Tl * S)
            epi = {}
            if add is not None:
                epi.update(add=add[bi * Tl:], add_rpg=add_rpg, add_ld=add_ld)
            if res1 is not None:
                epi.update(res1=res1[rows])
            if coef is not None:
                epi.update(coef=coef[bi * Tl:(bi + 1) * Tl].contiguous(), coef_rpg=coef_rpg)
            else:
                epi.update(c_acc=c_acc, c_res1=c_res1)
            ops.convt3(buf[bi].reshape((Tl + 2) * S, -1), w, b, Tl, S, tmin=tmin, tmax=tmax, a_row0=S, M=Tl * S, out=out[rows], **epi)
        return out

    def tmix_small(self, ops, y: torch.Tensor, w, b, g, out_ch: int):
        """Frame-sharded AE3DConv.time_mix_conv on the fp32 [rows, 4] map."""
        B, S, Tl = g.B, g.S, self.T_local
        buf = self._halo_buffer(y, B, S)                  # [B, Tl+2, S, 4] fp32
        tmin = 0 if self.first else -1
        tmax = Tl - 1 if self.last else Tl
        outs = []
        for bi in range(B):
            outs.append(ops.tmix_small(buf[bi].reshape((Tl + 2) * S, -1), w, b, 1, Tl, S, out_ch, tmin, tmax, row0=S))
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
