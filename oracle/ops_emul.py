"""TEST INFRASTRUCTURE — torch restatement of every C-ABI operator in include/v3d_hip.h.

Same primitive names, argument meaning and storage dtypes as v3d_amd.hip.HipOps, but computed with plain
fp32 torch ops on whatever device the tensors live on (CPU in the `-m "not gpu"` suite, the GPU when it is the
per-op checker for the HIP kernels).  Used only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline.
Semantics follow the op contracts in include/v3d_hip.h, which cite the reference call sites they replace.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from v3d_amd.ops import GEMM_CONV3X3, GEMM_CONVT3, GEMM_LINEAR, GemmCall, OpsBase


def _flat_from(t: torch.Tensor) -> torch.Tensor:
    """1-D view of the storage starting at t's first element (C pointer semantics for strided `add` tables)."""
    off = t.storage_offset()
    total = t.untyped_storage().nbytes() // t.element_size()
    return t.as_strided((total - off,), (1,), off)


class EmulOps(OpsBase):
    name = "emul"
    always_fuse = True      # the engine's "does the fused kernel pay here" policies are about GPU tile counts; the emulator takes every fused path (host-wiring coverage)

    def __init__(self, device="cpu", exact: bool = False):
        """exact=True keeps activations / packed weights in fp32 (no bf16 rounding points): the engine's wiring can then be
        checked against the fp32 oracle to ~1e-5, separating host-logic errors from precision."""
        self.device = torch.device(device)
        self.act_dtype = torch.float32 if exact else torch.bfloat16

    # ---- v3d_gemm ---------------------------------------------------------------------------------
    def _gemm_acc(self, g: GemmCall, A, W):
        K, N, M = g.K, g.N, g.M
        if g.mode == GEMM_LINEAR:
            return A[:M, :K].float() @ W.float().reshape(N, K).t()
        if g.mode == GEMM_CONV3X3:
            n_img = M // (g.Hout * g.Wout)
            x = A[: n_img * g.Hin * g.Win, :K].float().reshape(n_img, g.Hin, g.Win, K).permute(0, 3, 1, 2)
            if g.up == 2:
                x = F.interpolate(x, scale_factor=2, mode="nearest")
            w = W.float().reshape(3, 3, N, K).permute(2, 3, 0, 1)
            if g.pad_mode:
                y = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, stride=g.stride, padding=0)
            else:
                y = F.conv2d(x, w, stride=g.stride, padding=1)
            assert y.shape[2] == g.Hout and y.shape[3] == g.Wout, (y.shape, g.Hout, g.Wout)
            return y.permute(0, 2, 3, 1).reshape(M, N)
        if g.mode == GEMM_CONVT3:
            Af = A[:, :K].float()
            w = W.float().reshape(3, N, K)
            m = torch.arange(M, device=A.device)
            t = (m // g.S) % g.T
            acc = torch.zeros(M, N, dtype=torch.float32, device=A.device)
            for dt in range(3):
                tt = t + dt - 1
                valid = (tt >= g.tmin) & (tt <= g.tmax)
                src = m + g.a_row0 + (dt - 1) * g.S
                if g.halo_rows:
                    frame = m // g.S
                    hoff = (frame // g.T) * g.S + (m - frame * g.S)
                    src = torch.where(tt < 0, g.a_row0 - g.halo_rows + hoff, src)
                    src = torch.where(tt >= g.T, g.a_row0 + M + hoff, src)
                src = torch.where(valid, src, torch.zeros_like(src))
                rows = torch.where(valid[:, None], Af[src], torch.zeros((), dtype=Af.dtype, device=Af.device))   # (masked rows may be uninitialised halo slabs)
                acc += rows @ w[dt].t()
            return acc
        raise ValueError(g.mode)

    def gemm_gn_in_supported(self, g: GemmCall) -> bool:
        return g.gn_in is not None and g.batch == 1 and not g.geglu      # the emulator normalises any operand

    def _gn_in_operand(self, g: GemmCall):
        """The operand the contraction sees when GemmCall.gn_in is set: act(x * scale + shift) of the raw rows (A | A2), rounded to the storage dtype."""
        x = g.A[:, :g.A.shape[-1]].float()
        if g.A2 is not None:
            x = torch.cat([x, g.A2.float()], dim=-1)
        r = torch.arange(x.shape[0], device=x.device)
        rows = (r - g.a_row0) // g.gn_in_rps
        if g.mode == GEMM_CONVT3 and g.halo_rows:
            # split-halo layout (frame sharding): the slabs in front of / behind the local frames hold frame -1 / T of sample 0 .. B-1
            rows = torch.where(r < g.a_row0, (r - (g.a_row0 - g.halo_rows)) // g.S, rows)
            rows = torch.where(r >= g.a_row0 + g.M, (r - (g.a_row0 + g.M)) // g.S, rows)
        rows = rows.clamp(0, g.gn_in.shape[0] - 1)           # (in-line halo frames of a single sample belong to its one statistics group)
        y = x * g.gn_in[rows, :, 0] + g.gn_in[rows, :, 1]
        if g.gn_in_silu:
            y = y * torch.sigmoid(y)
        return y.to(self.act_dtype)

    def gemm(self, g: GemmCall):
        if g.gn_in is not None:
            import dataclasses
            g = dataclasses.replace(g, A=self._gn_in_operand(g), A2=None, gn_in=None)
        for z in range(g.batch):
            A = g.A[z] if (g.A.dim() == 3 and g.mode == GEMM_LINEAR and (g.batch > 1 or g.out.dim() == 3)) else g.A
            W = g.W[z] if (g.W.dim() == 3 and g.mode == GEMM_LINEAR and (g.batch > 1 or g.out.dim() == 3)) else g.W
            out = g.out[z] if (g.batch > 1 or g.out.dim() == 3) else g.out
            M, N = g.M, g.N
            v = self._gemm_acc(g, A, W)
            if g.bias is not None:
                v = v + g.bias.float()[None, :]
            if g.add is not None:
                rows = torch.arange(M, device=v.device) // g.add_rpg
                idx = rows[:, None] * g.add_ld + torch.arange(N, device=v.device)[None, :]
                v = v + _flat_from(g.add)[idx]
            if g.geglu:
                v4 = v.reshape(M, N // 32, 2, 16)
                v = (v4[:, :, 0, :] * F.gelu(v4[:, :, 1, :])).reshape(M, N // 2)
            ca, c1, c2 = g.c_acc, g.c_res1, g.c_res2
            if g.coef is not None:
                grp = torch.arange(M, device=v.device) // g.coef_rpg
                cf = g.coef.reshape(-1, 3)[grp]
                ca, c1, c2 = cf[:, 0:1], cf[:, 1:2], cf[:, 2:3]
            o = ca * v
            if g.res1 is not None:
                o = o + c1 * g.res1[:M].float()
            if g.res2 is not None:
                o = o + c2 * g.res2[:M].float()
            out[:M].copy_(o.to(out.dtype))
            if g.gn_stats is not None:
                # GroupNorm partial sums of the stored (rounded) output, layout of groupnorm_stats (slot 0 only in the emulator)
                v = out[:M].float().reshape(M // g.gn_rps, g.gn_rps, N // g.gn_cpg, g.gn_cpg)
                g.gn_stats[:, 0, :, 0] += v.sum(dim=(1, 3))
                g.gn_stats[:, 0, :, 1] += (v * v).sum(dim=(1, 3))

    # ---- norms ------------------------------------------------------------------------------------
    @staticmethod
    def _cat(x1, x2):
        return x1 if x2 is None else torch.cat([x1, x2], dim=-1)

    def groupnorm_stats(self, x1, x2, stats, n_img, S, groups, imgs_per_stat):
        x = self._cat(x1, x2).float()
        C = x.shape[-1]
        xg = x.reshape(n_img // imgs_per_stat, imgs_per_stat * S, groups, C // groups)
        stats[:, 0, :, 0] += xg.sum(dim=(1, 3))          # [stat group][slot][group][2]; the emulator uses slot 0 only
        stats[:, 0, :, 1] += (xg * xg).sum(dim=(1, 3))

    def groupnorm_stats_table(self, x1, x2, stats, tickets, n_img, S, groups, imgs_per_stat, gamma, beta, count, eps, table):
        """v3d_groupnorm_stats_table (ABI 5): statistics + table in one launch == stats, then finalize."""
        self.groupnorm_stats(x1, x2, stats, n_img, S, groups, imgs_per_stat)
        self.groupnorm_finalize(stats, None, gamma, beta, count, eps, table)

    def groupnorm_finalize(self, stats, sums, gamma, beta, count, eps, table):
        if stats is not None:
            tot = stats.double().sum(dim=1)              # [n_stat, groups, 2]
            if sums is not None:
                sums.copy_(tot)
        else:
            tot = sums.double()
        if table is None:
            return
        groups = tot.shape[1]
        C = gamma.numel()
        mean = tot[..., 0] / count
        var = (tot[..., 1] / count - mean * mean).clamp_min(0)
        rstd = 1.0 / torch.sqrt(var + eps)
        sc = gamma.double()[None, :] * rstd.repeat_interleave(C // groups, dim=1)
        sh = beta.double()[None, :] - mean.repeat_interleave(C // groups, dim=1) * sc
        table.copy_(torch.stack([sc, sh], dim=-1).float())

    def groupnorm_small_supported(self, C1, C2, S, imgs_per_stat=1, groups=32):
        """Mirror of v3d_groupnorm_small_supported (norm.hip gn_small_groups): so the engine takes the same path on the emulator."""
        C, rows = C1 + C2, imgs_per_stat * S
        if groups != 32 or C % 32 or C1 % 8 or rows <= 0 or (C // 32) % 8:
            return False
        cpg = C // 32
        for gb in (8, 4, 2, 1):
            if C2 and C1 % (gb * cpg):
                continue
            if rows * (gb * cpg // 8) <= 256 * 24 and gb * cpg * 2 >= 64:
                return True
        return False

    def groupnorm_small(self, x1, x2, gamma, beta, out, n_img, S, *, eps, silu, imgs_per_stat=1, groups=32):
        """v3d_groupnorm_small: statistics in fp64 over the statistics group, y = x * (gamma rstd) + (beta - mean gamma rstd), SiLU, bf16."""
        x = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], dim=-1)
        C = x.shape[-1]
        n_stat, rows, cpg = n_img // imgs_per_stat, imgs_per_stat * S, C // groups
        v = x.reshape(n_stat, rows, groups, cpg).double()
        mean = v.mean(dim=(1, 3), keepdim=True)
        var = ((v * v).mean(dim=(1, 3), keepdim=True) - mean * mean).clamp_min(0.0)
        rstd = 1.0 / torch.sqrt(var + eps)
        sc = gamma.float().reshape(1, 1, groups, cpg) * rstd.float()
        sh = beta.float().reshape(1, 1, groups, cpg) - mean.float() * sc
        y = v.float() * sc + sh
        if silu:
            y = y * torch.sigmoid(y)
        out.copy_(y.reshape(n_img * S, C).to(out.dtype))
        return out

    def groupnorm_apply(self, x1, x2, table, out, n_img, S, imgs_per_stat, silu):
        x = self._cat(x1, x2).float()
        C = x.shape[-1]
        xg = x.reshape(n_img // imgs_per_stat, imgs_per_stat * S, C)
        y = (xg * table[:, None, :, 0] + table[:, None, :, 1]).reshape(n_img * S, C)
        if silu:
            y = y * torch.sigmoid(y)
        out.copy_(y.to(out.dtype))

    def layernorm(self, x, gamma, beta, out, eps, add=None, add_rpg=0, add_ld=0, xsum_out=None):
        C = x.shape[-1]
        xf = x.reshape(-1, C).float()
        M = xf.shape[0]
        if add is not None:
            rows = torch.arange(M, device=x.device) // add_rpg
            idx = rows[:, None] * add_ld + torch.arange(C, device=x.device)[None, :]
            xf = xf + _flat_from(add)[idx]
            if xsum_out is not None:
                xs = xf.to(self.act_dtype)
                xsum_out.reshape(-1, C).copy_(xs)
                xf = xs.float()
        y = F.layer_norm(xf, (C,), gamma.float(), beta.float(), eps)
        out.reshape(-1, C).copy_(y.to(out.dtype))

    # ---- attention --------------------------------------------------------------------------------
    def attn_spatial(self, q, k, vT, out, n_img, S, heads, scale):
        C = heads * 64
        qf = q[:, :C].float().reshape(n_img, S, heads, 64).permute(0, 2, 1, 3)
        kf = k[:, :C].float().reshape(n_img, S, heads, 64).permute(0, 2, 1, 3)
        vf = vT.float().reshape(n_img, heads, 64, S).permute(0, 1, 3, 2)
        o = F.scaled_dot_product_attention(qf, kf, vf, scale=scale)
        out[:, :C].copy_(o.permute(0, 2, 1, 3).reshape(n_img * S, C).to(out.dtype))

    def attn_temporal(self, q, k, v, out, heads, scale):
        B, Tq, S, C = q.shape
        Tk = k.shape[1]
        qf = q.float().reshape(B, Tq, S, heads, 64).permute(0, 2, 3, 1, 4)   # b s h t d
        kf = k.float().reshape(B, Tk, S, heads, 64).permute(0, 2, 3, 1, 4)
        vf = v.float().reshape(B, Tk, S, heads, 64).permute(0, 2, 3, 1, 4)
        o = F.scaled_dot_product_attention(qf, kf, vf, scale=scale)           # b s h tq d
        out.copy_(o.permute(0, 3, 1, 2, 4).reshape(B, Tq, S, C).to(out.dtype))

    # ---- fp8 attention: tile-scaled e4m3 quantisation restated with torch.float8_e4m3fn, attention on the dequantised operands ----
    def quant_fp8_tiles(self, x, n_img, S):
        ncols = x.shape[-1]
        nt = (S + 63) // 64
        xf = x.float().reshape(n_img, S, ncols)
        pad = nt * 64 - S
        xp = F.pad(xf, (0, 0, 0, pad)).reshape(n_img, nt, 64, ncols // 64, 64)
        amax = xp.abs().amax(dim=(2, 4))                                             # [n, nt, ncols / 64]
        scales = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
        q = (xp / scales[:, :, None, :, None]).to(torch.float8_e4m3fn)
        x8 = q.view(torch.uint8).reshape(n_img, nt * 64, ncols)[:, :S].reshape(n_img * S, ncols).contiguous()
        return x8, scales.contiguous()

    def quant_fp8_slab(self, vT, heads):
        n_img, C, S = vT.shape
        vf = vT.float().reshape(n_img, heads, 64, S)
        amax = vf.abs().amax(dim=(2, 3))
        vscale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
        v8 = (vf / vscale[:, :, None, None]).to(torch.float8_e4m3fn).view(torch.uint8).reshape(n_img, C, S).contiguous()
        return v8, vscale.contiguous()

    def attn_spatial_fp8(self, qk8, scales, v8, vscale, out, n_img, S, heads, scale):
        C = heads * 64
        nt = scales.shape[1]
        deq = qk8.view(torch.float8_e4m3fn).float().reshape(n_img, S, 2 * heads, 64)
        rows = torch.arange(S, device=qk8.device) // 64
        deq = deq * scales[:, rows][..., None]                                       # [n, S, 2 heads, 64]
        qf, kf = deq[:, :, :heads].permute(0, 2, 1, 3), deq[:, :, heads:].permute(0, 2, 1, 3)
        vf = v8.view(torch.float8_e4m3fn).float().reshape(n_img, heads, 64, S) * vscale[:, :, None, None]
        o = F.scaled_dot_product_attention(qf, kf, vf.permute(0, 1, 3, 2), scale=scale)
        out[:, :C].copy_(o.permute(0, 2, 1, 3).reshape(n_img * S, C).to(out.dtype))

    ATTN_VAE_WIDTHS = (128, 256, 512)

    def attn_vae(self, q, k, vT, bias, out, n_img, S, C, scale):
        qf = q[:, :C].float().reshape(n_img, 1, S, C)
        kf = k[:, :C].float().reshape(n_img, 1, S, C)
        vf = vT.float().reshape(n_img, 1, C, S).permute(0, 1, 3, 2)
        o = F.scaled_dot_product_attention(qf, kf, vf, scale=scale).reshape(n_img * S, C)
        if bias is not None:
            o = o + bias.float()[None, :]
        out[:, :C].copy_(o.to(out.dtype))

    def softmax_rows(self, inp, out):
        out.copy_(torch.softmax(inp.float(), dim=-1).to(out.dtype))

    # ---- small elementwise ------------------------------------------------------------------------
    def timestep_embedding(self, t, dim, max_period=10000.0):
        half = dim // 2
        freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        args = t.float()[:, None] * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        if dim % 2:
            emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
        return emb.to(self.act_dtype)

    def silu_add(self, a, b=None):
        v = a.float() if b is None else a.float() + b.float()
        return (v * torch.sigmoid(v)).to(self.act_dtype)

    def edm_scalings(self, sigma):
        s = sigma.float()
        d = s * s + 1.0
        return 1.0 / d, -s / d.sqrt(), 1.0 / d.sqrt(), 0.25 * s.log()

    def pack_input(self, x, scale, cond, Cpad):
        n, C1 = x.shape[0], x.shape[1]
        xs = x.float().reshape(n, C1, -1)
        S = xs.shape[-1]
        if scale is not None:
            xs = xs * scale.float()[:, None, None]
        parts = [xs]
        if cond is not None:
            parts.append(cond.float().reshape(n, cond.shape[1], S))
        full = torch.cat(parts, dim=1)
        if full.shape[1] < Cpad:
            full = torch.cat([full, torch.zeros(n, Cpad - full.shape[1], S, device=x.device)], dim=1)
        return full.permute(0, 2, 1).reshape(n * S, Cpad).to(self.act_dtype)

    def pack_input_im2col3x3(self, x, scale, cond, Kpad):
        """v3d_pack_input_im2col3x3: the packed 8-channel input, unfolded 3x3 (tap-major columns tap*8 + c), zero-padded to Kpad columns."""
        n, C1, H, W = x.shape
        packed = self.pack_input(x, scale, cond, 8).float().reshape(n, H, W, 8).permute(0, 3, 1, 2)      # bf16-rounded, like the kernel's lanes
        cols = torch.nn.functional.unfold(packed, 3, padding=1).reshape(n, 8, 9, H * W)                  # [n, c, tap, s]
        out = torch.zeros(n, H * W, Kpad, device=x.device)
        out[:, :, :72] = cols.permute(0, 3, 2, 1).reshape(n, H * W, 72)
        return out.reshape(n * H * W, Kpad).to(self.act_dtype)

    def tapsum3x3(self, y, bias, n, H, W, C):
        """v3d_tapsum3x3: gather-sum of the nine taps' products (fp32, fixed tap order)."""
        yv = y.float()[:, :9 * C].reshape(n, H, W, 9, C)
        out = torch.zeros(n, H, W, C, device=y.device) if bias is None else bias.float().reshape(1, 1, 1, C).expand(n, H, W, C).clone()
        for tap in range(9):
            dy, dx = tap // 3 - 1, tap % 3 - 1
            ys, ye = max(0, -dy), min(H, H - dy)
            xs, xe = max(0, -dx), min(W, W - dx)
            out[:, ys:ye, xs:xe] += yv[:, ys + dy:ye + dy, xs + dx:xe + dx, tap]
        return out.reshape(n * H * W, C)

    def denoise_combine(self, net, x, c_out, c_skip):
        n, C = x.shape[0], x.shape[1]
        S = x.numel() // (n * C)
        nf = net[:, :C].float().reshape(n, S, C).permute(0, 2, 1).reshape(x.shape)
        shp = (n,) + (1,) * (x.dim() - 1)
        return nf * c_out.reshape(shp) + x.float() * c_skip.reshape(shp)

    def cfg_combine(self, x, scale, T):
        n = x.shape[0] // 2
        xu, xc = x[:n].float(), x[n:].float()
        sc = scale.float()[torch.arange(n, device=x.device) % T].reshape((n,) + (1,) * (x.dim() - 1))
        return xu + sc * (xc - xu)

    def euler_step(self, x, den, sigma, next_sigma):
        shp = (x.shape[0],) + (1,) * (x.dim() - 1)
        d = (x - den) / sigma.reshape(shp)
        return x + (next_sigma - sigma).reshape(shp) * d

    @staticmethod
    def ff_k_perm(hidden: int, device):
        """Column order of W2p inside v3d_ff_fused: position 32 s + 16 a + 8 h + 4 t + c holds hidden channel 32 s + 16 a + 8 t + 4 h + c."""
        s_, a, h, t, c = torch.meshgrid(torch.arange(hidden // 32), torch.arange(2), torch.arange(2), torch.arange(2), torch.arange(4),
                                        indexing="ij")
        return (32 * s_ + 16 * a + 8 * t + 4 * h + c).reshape(-1).to(device)

    @staticmethod
    def _ff_untile(w, rows, cols, slab_rows, slab_cols):
        """v3d_ff_fused weight stream order -> row-major [rows, cols]: slabs of slab_rows x slab_cols, inside a slab 1-KiB pieces of
        16 rows x 32 columns ordered (column block, row block), inside a piece the 16-byte unit 4 r + p holds columns
        8 (p ^ swz(r)) .. + 7 of row r, swz(r) = {0,2,3,1}[(r >> 2) & 3]."""
        t = w.reshape(rows // slab_rows, cols // slab_cols, slab_cols // 32, slab_rows // 16, 16, 4, 8)
        r = torch.arange(16, device=w.device)
        swz = torch.tensor([0, 2, 3, 1], device=w.device)[(r >> 2) & 3]
        pos = (torch.arange(4, device=w.device)[None, :] ^ swz[:, None])            # [row, logical chunk] -> unit position
        pos = pos[None, None, None, None, :, :, None].expand(*t.shape[:4], 16, 4, 8)
        logical = torch.gather(t, 5, pos)                                            # [.., row, chunk, 8]
        return logical.permute(0, 3, 4, 1, 2, 5, 6).reshape(rows, cols)

    def ln_ff_fused(self, x, eps, w1p, b1, w2p, b2, out, **kw):
        """v3d_ln_ff_fused: rows normalised without affine (it is folded into w1p / b1), rounded where the kernel rounds them, then ff_fused."""
        xh = F.layer_norm(x.float(), (x.shape[-1],), None, None, eps).to(self.act_dtype)
        return self.ff_fused(xh, w1p, b1, w2p, b2, out, **kw)

    def ff_fused(self, x, w1p, b1, w2p, b2, out, *, res1=None, res2=None, coef=None, coef_rpg=0, c_acc=1.0, c_res1=1.0, c_res2=1.0):
        M, Cc = x.shape
        hidden = w2p.shape[-1]
        # the weight buffers arrive in the kernel's LDS-DMA piece order (include/v3d_hip.h): undo it first
        w1p = self._ff_untile(w1p, 2 * hidden, Cc, 64, Cc)
        w2p = self._ff_untile(w2p, Cc, hidden, Cc, 32)
        # W1p rows: 64 s + 32 a + 8 g + 4 h + c = (g odd ? gate : value) of hidden channel 32 s + 16 a + 8 (g >> 1) + 4 h + c
        sraw = (x.float() @ w1p.float().t() + b1.float()).reshape(M, hidden // 16, 2, 2, 8)   # [.., g >> 1, g & 1, 4 h + c]
        h = (sraw[:, :, :, 0] * F.gelu(sraw[:, :, :, 1])).reshape(M, hidden)                   # natural channel order
        h = h.to(self.act_dtype).float()                                      # the kernel feeds bf16 hidden values to the 2nd MFMA
        y = h[:, self.ff_k_perm(hidden, x.device)] @ w2p.float().t() + b2.float()
        if coef is not None:
            cf = coef[(torch.arange(M, device=x.device) // coef_rpg)]
            ca, c1, c2 = cf[:, 0:1], cf[:, 1:2], cf[:, 2:3]
        else:
            ca, c1, c2 = c_acc, c_res1, c_res2
        y = ca * y
        if res1 is not None:
            y = y + c1 * res1.float()
        if res2 is not None:
            y = y + c2 * res2.float()
        out.copy_(y.to(out.dtype))
        return out

    LN_PROJ_WIDTHS = (320,)

    def ln_proj(self, x, eps, wp, bias, n_rm, S):
        M, C = x.shape
        N = wp.shape[0]
        w = self._ff_untile(wp, N, C, 64, C).float()                        # undo the LDS-DMA piece order
        xh = F.layer_norm(x.float(), (C,), None, None, eps).to(self.act_dtype).float()    # affine folded into w / bias at pack time
        y = xh @ w.t() + bias.float()[None, :]
        out = y[:, :n_rm].to(self.act_dtype).contiguous() if n_rm else None
        outT = y[:, n_rm:].reshape(M // S, S, N - n_rm).permute(0, 2, 1).to(self.act_dtype).contiguous() if n_rm < N else None
        return out, outT

    def clip_preprocess(self, img, size, patch, antialias, mean, std, kpad):
        """kornia.geometry.resize(bicubic, align_corners=True, antialias) restated with torch ops + normalise + patch unfold."""
        B, _, H, W = img.shape
        x = img.float()
        fy, fx = H / size, W / size
        if antialias and max(fy, fx) > 1:
            sg = (max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001))
            ks = [int(max(4.0 * s_, 3)) for s_ in sg]
            ks = [k + 1 if k % 2 == 0 else k for k in ks]

            def g1(k, s_):
                t = torch.arange(k, dtype=torch.float32, device=x.device) - k // 2
                w = torch.exp(-t * t / (2.0 * s_ * s_))
                return w / w.sum()
            ky, kx = g1(ks[0], sg[0]), g1(ks[1], sg[1])
            xp = F.pad(x, (ks[1] // 2, ks[1] // 2, ks[0] // 2, ks[0] // 2), mode="reflect")
            w2 = (ky[:, None] * kx[None, :])[None, None].repeat(3, 1, 1, 1)
            x = F.conv2d(xp, w2, groups=3)
        x = F.interpolate(x, size=(size, size), mode="bicubic", align_corners=True)
        x = (x + 1.0) / 2.0
        x = (x - torch.tensor(mean, device=x.device).view(1, 3, 1, 1)) / torch.tensor(std, device=x.device).view(1, 3, 1, 1)
        g = size // patch
        pt = x.reshape(B, 3, g, patch, g, patch).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, 3 * patch * patch)
        out = torch.zeros(B * g * g, kpad, dtype=self.act_dtype, device=img.device)
        out[:, :3 * patch * patch] = pt.to(self.act_dtype)
        return out

    def frames_to_uint8(self, x):
        samples = torch.clamp((x.float() + 1.0) / 2.0, min=0.0, max=1.0)          # V3D_512.py:286-303, verbatim order of operations
        return (samples.permute(0, 2, 3, 1) * 255).to(torch.uint8).contiguous()

    def gelu(self, x, out=None):
        y = F.gelu(x.float()).to(x.dtype)
        if out is not None:
            out.copy_(y)
            return out
        return y

    def heun_step(self, x, den, euler, den2, sigma, next_sigma):
        shp = (x.shape[0],) + (1,) * (x.dim() - 1)
        sg, nx = sigma.reshape(shp), next_sigma.reshape(shp)
        d = (x - den) / sg
        dn = (euler - den2) / torch.where(nx > 0, nx, torch.ones_like(nx))
        return torch.where(nx > 0, x + (nx - sg) * ((d + dn) * 0.5), euler)

    def axpb_f32(self, x, a, b=0.0, out=None):
        r = x * a + b
        if out is not None:
            out.copy_(r)
            return out
        return r

    def blend_coefs(self, alpha, kind, ioi, n_img):
        nm = alpha.numel()
        a = alpha.float()[:, None].expand(nm, n_img).clone()
        if ioi is not None:
            a = torch.where(ioi.reshape(1, n_img) != 0, torch.ones_like(a), a)
        out = torch.empty(nm, n_img, 3, dtype=torch.float32, device=alpha.device)
        k0 = (kind == 0)[:, None]
        out[..., 0] = 1 - a
        out[..., 1] = torch.where(k0, torch.ones_like(a), 1 - a)
        out[..., 2] = torch.where(k0, torch.zeros_like(a), a)
        return out

    def nchw_to_nhwc_bf16(self, x, scale, Cpad):
        return self.pack_input(x.float() * scale, None, None, Cpad)

    def tmix_small(self, x, w, b, B, T, S, Cc, tmin, tmax, row0=0):
        nfr = x.shape[0] // S
        xf = x[:, :Cc].float().reshape(nfr, S, Cc)
        f0 = row0 // S
        out = torch.zeros(B * T, Cc, S, dtype=torch.float32, device=x.device)
        f = torch.arange(B * T, device=x.device)
        t = f % T
        for dt in range(3):
            tt = t + dt - 1
            valid = ((tt >= tmin) & (tt <= tmax)).float()
            src = (f + f0 + dt - 1).clamp(0, nfr - 1)
            xs = torch.where(valid[:, None, None] > 0, xf[src], torch.zeros((), dtype=xf.dtype, device=xf.device))   # [f, s, ci]
            out += torch.einsum("fsi,oi->fos", xs, w[:, :, dt].float())
        return out + b.float()[None, :, None]

    def copy2d_bf16(self, src, dst):
        dst.copy_(src)
        return dst
