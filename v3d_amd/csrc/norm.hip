// GroupNorm (stats / apply), LayerNorm and row softmax for channels-last bf16 activations (gfx950).
// All of these are HBM-bound: every access is a 16-byte (8 x bf16) per-lane vector, coalesced along channels.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int NVMAX = 2;       // vector columns per thread -> supports C <= 2*256*8 = 4096
constexpr int CMAX = 4096;

struct GNGeom {
    int VC;    // 16-B vectors per row
    int TPR;   // threads per row
    int RPP;   // rows per pass of a 256-thread block
    int NV;    // vector columns per thread
};

__host__ __device__ inline GNGeom gn_geom(long long C) {
    GNGeom g;
    g.VC = (int)(C / 8);
    g.TPR = g.VC < 256 ? g.VC : 256;
    g.RPP = 256 / g.TPR;
    g.NV = (g.VC + g.TPR - 1) / g.TPR;
    return g;
}

__device__ __forceinline__ uint4 load_vec2(const bf16_t* x1, long long C1, const bf16_t* x2, long long C2, long long row, int c) {
    if (c < C1) return *reinterpret_cast<const uint4*>(x1 + row * C1 + c);
    return *reinterpret_cast<const uint4*>(x2 + row * C2 + (c - C1));
}

__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
    f[0] = bflo(u.x); f[1] = bfhi(u.x); f[2] = bflo(u.y); f[3] = bfhi(u.y);
    f[4] = bflo(u.z); f[5] = bfhi(u.z); f[6] = bflo(u.w); f[7] = bfhi(u.w);
}

// One block per statistics group: the slots' partial sums are added up in a FIXED order in fp64 (NT / 16 slot lanes per (group, component), then
// the lanes in order), mean / variance / rstd in fp64 (E[x^2] - E[x]^2 cancels in fp64, not fp32: inputs with |mean| >> std keep their
// variance), and the per-(statistics group, channel) affine of the normalisation is written as a table
//   table[stat][c] = (scale, shift) = (gamma[c] rstd[g],  beta[c] - mean[g] gamma[c] rstd[g])          y = x * scale + shift
// which is all a consumer needs: v3d_groupnorm_apply, and the convolutions that normalise their input tile on its way into LDS
// (v3d_gemm gn_in_table).  `sums` [n_stat][groups][2] fp64 is the frame-sharded runtime's hand-off: written when the slots are given,
// read (after the all-reduce over ranks) when they are not.
// The body runs on NT threads of ONE block with `smem` >= gn_fin_smem<NT>() bytes (8-byte aligned): as its own kernel (NT = 1024), and at
// the end of gn_stats_kernel<.., FUSE> on the 256 threads of the LAST block of a statistics group to finish (round 5: v3d_groupnorm_stats_table).
template <int NT>
__host__ __device__ constexpr int gn_fin_smem() { return (NT / 16) * 64 * 8 + 64 * 8 + 64 * 4; }

// four 16-byte loads that bypass the non-coherent cache levels (sc0 sc1: what another XCD's L2 may still hold is not what these return) - the
// fused statistics kernel's last block reads slots other blocks wrote DURING THIS LAUNCH (gemm_common.h sk_gather reads partials the same way)
__device__ __forceinline__ void gn_ld4_coherent(const float* p0, const float* p1, const float* p2, const float* p3, float4& v0, float4& v1, float4& v2, float4& v3) {
    f32x4 a, b, c, d;
    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
                 "global_load_dwordx4 %1, %5, off sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc0 sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                 : "memory");
    v0 = make_float4(a[0], a[1], a[2], a[3]); v1 = make_float4(b[0], b[1], b[2], b[3]);
    v2 = make_float4(c[0], c[1], c[2], c[3]); v3 = make_float4(d[0], d[1], d[2], d[3]);
}

template <int NT, bool COHERENT = false>
__device__ __forceinline__ void gn_finalize_block(unsigned char* smem, const float* __restrict__ stats, long long nslots, long long nsum, double* __restrict__ sums,
                                                  int groups, const float* __restrict__ gamma, const float* __restrict__ beta, long long C,
                                                  double inv_count, float eps, float* __restrict__ table, long long st) {
    // NT threads: NT / 16 slot lanes x 16 float4 columns of a slot row (groups * 2 <= 64 floats); every lane adds its slots (stride NT / 16) in
    // fp64 with 4 independent loads in flight, the lanes meet in LDS and are added in lane order - a fixed order whatever the scheduling.
    // (The first version walked the slots with 4 lanes of 64 threads: 45 us per launch for the 1154 slots of a 64x64-level 3-D norm.)
    constexpr int LANES = NT / 16, QL = LANES / 4;
    double (*part)[64] = reinterpret_cast<double (*)[64]>(smem);
    double* tot = reinterpret_cast<double*>(smem + LANES * 64 * 8);
    float* ms = reinterpret_cast<float*>(smem + LANES * 64 * 8 + 64 * 8);           // [group][2]: mean, rstd
    const int tid = threadIdx.x;
    const int npair = groups * 2;      // <= 64, % 4 == 0 (host-checked)
    const int vec = tid & 15, sl = tid >> 4;
    if (stats) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        if (vec * 4 < npair) {
            const float* sp = stats + (st * nslots) * npair + vec * 4;
            long long k = sl;
            for (; k + 3 * LANES < nsum; k += 4 * LANES) {          // (nsum <= nslots: the slots to add; nslots = the slot stride of a statistics group)
                float4 v0, v1, v2, v3;
                if constexpr (COHERENT) {
                    gn_ld4_coherent(sp + k * npair, sp + (k + LANES) * npair, sp + (k + 2 * LANES) * npair, sp + (k + 3 * LANES) * npair, v0, v1, v2, v3);
                } else {
                    v0 = *reinterpret_cast<const float4*>(sp + k * npair);
                    v1 = *reinterpret_cast<const float4*>(sp + (k + LANES) * npair);
                    v2 = *reinterpret_cast<const float4*>(sp + (k + 2 * LANES) * npair);
                    v3 = *reinterpret_cast<const float4*>(sp + (k + 3 * LANES) * npair);
                }
                a0 += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
                a1 += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
                a2 += ((double)v0.z + (double)v1.z) + ((double)v2.z + (double)v3.z);
                a3 += ((double)v0.w + (double)v1.w) + ((double)v2.w + (double)v3.w);
            }
            for (; k < nsum; k += LANES) {
                float4 v0;
                if constexpr (COHERENT) {
                    float4 u1, u2, u3;
                    gn_ld4_coherent(sp + k * npair, sp + k * npair, sp + k * npair, sp + k * npair, v0, u1, u2, u3);
                } else {
                    v0 = *reinterpret_cast<const float4*>(sp + k * npair);
                }
                a0 += (double)v0.x; a1 += (double)v0.y; a2 += (double)v0.z; a3 += (double)v0.w;
            }
        }
        part[sl][vec * 4 + 0] = a0;
        part[sl][vec * 4 + 1] = a1;
        part[sl][vec * 4 + 2] = a2;
        part[sl][vec * 4 + 3] = a3;
        __syncthreads();
        // the lanes of a pair: a fixed binary tree (4 threads per pair add LANES / 4 lanes each in order, then ((0 + 1) + (2 + 3)))
        if (tid < npair * 4) {
            const int pr = tid >> 2, q = tid & 3;
            double a = 0.0;
#pragma unroll
            for (int l = 0; l < QL; ++l) a += part[q * QL + l][pr];
            part[q * QL][pr] = a;          // (only this thread reads rows q * QL .. q * QL + QL - 1 of column pr)
        }
        __syncthreads();
        if (tid < npair) {
            const double a = (part[0][tid] + part[QL][tid]) + (part[2 * QL][tid] + part[3 * QL][tid]);
            tot[tid] = a;
            if (sums) sums[st * npair + tid] = a;
        }
    } else if (tid < npair) {
        tot[tid] = sums[st * npair + tid];
    }
    __syncthreads();
    if (!table) return;
    if (tid < groups) {
        const double mean = tot[tid * 2] * inv_count;
        double var = tot[tid * 2 + 1] * inv_count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        ms[tid * 2] = (float)mean;
        ms[tid * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cpg = (int)(C / groups);
    for (long long c = tid; c < C; c += NT) {
        const int gidx = (int)(c / cpg);
        const float sc = gamma[c] * ms[gidx * 2 + 1];
        *reinterpret_cast<float2*>(table + (st * C + c) * 2) = make_float2(sc, beta[c] - ms[gidx * 2] * sc);
    }
}

__global__ __launch_bounds__(1024) void gn_finalize_kernel(const float* __restrict__ stats, long long nslots, double* __restrict__ sums, int groups,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, long long C,
                                                           double inv_count, float eps, float* __restrict__ table) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[gn_fin_smem<1024>()];
    gn_finalize_block<1024>(smem, stats, nslots, nslots, sums, groups, gamma, beta, C, inv_count, eps, table, (long long)blockIdx.x);
}

// grid (chunks, n_img); block 256.  Each block reduces rows [chunk*rpb, (chunk+1)*rpb) of one image and stores its (sum, sumsq) per
// group into ITS OWN slot of the statistics buffer: slot = (img % imgs_per_stat) * chunks + chunk - plain stores, no atomics anywhere,
// so the statistics (and with them every GroupNorm output) are bit-reproducible run to run.  (Rounds 1-2 met in fp32 atomics: LDS atomics
// across the rows of a block, global atomics across blocks; two identical evaluations differed by 1-2e-2 after 50 layers.)
// NV (vector columns per thread) is a template parameter and the per-channel LDS partials are sized by C (dynamic shared
// memory): the first version carried the dead second column through every load / fma and declared 32 KiB of LDS, which
// capped a CU at 5 blocks - 56 us for the 94 MB 64x64 level where the read+write gn_apply takes 36 us.
// FUSE (round 5, v3d_groupnorm_stats_table): the block takes a ticket per statistics group once its slot is written and visible; the LAST block of
// a group to arrive folds the group's slots into the (scale, shift) table right here (gn_finalize_block on its 256 threads) - the separate
// finalize launch (7-11 us of launch-bound time, 41 per U-Net evaluation behind stand-alone statistics passes) disappears.  Which block is
// last varies run to run; what it computes does not (fixed slot order, fp64): the table stays bit-reproducible.
struct GNFuse {
    unsigned* tickets;      // [n_stat], zero on entry, reset to zero by the last block
    const float* gamma;
    const float* beta;
    float* table;
    double inv_count;
    float eps;
};

template <int NV, int UR, bool FUSE = false>
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x1, long long C1,
                                                       const bf16_t* __restrict__ x2, long long C2,
                                                       float* __restrict__ stats, long long nslots, long long S, int groups,
                                                       long long imgs_per_stat, long long rpb, GNFuse fz) {
    extern __shared__ __attribute__((aligned(16))) float gn_sh[];   // [2][RPP][C]: per (row lane, channel) sum, sum of squares (16-byte aligned: the fused fold reuses it for fp64)
    const long long C = C1 + C2;
    const GNGeom g = gn_geom(C);
    float* sh_s = gn_sh;
    float* sh_q = gn_sh + (long long)g.RPP * C;
    const int tid = threadIdx.x;
    const long long img = blockIdx.y;
    const long long r_begin = (long long)blockIdx.x * rpb;
    long long r_end = r_begin + rpb;
    if (r_end > S) r_end = S;

    const int trow = tid / g.TPR;
    const int tcol = tid - trow * g.TPR;
    float s[NV][8], q[NV][8];
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[j][e] = q[j][e] = 0.f;

    if (trow < g.RPP) {
        // UR rows per trip: the UR (x NV) 16-byte loads are issued back to back so several KB per wave are in flight
        for (long long r = r_begin + trow; r < r_end; r += (long long)UR * g.RPP) {
            uint4 v[UR][NV];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const long long rr = r + (long long)u * g.RPP;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int vc = tcol + j * g.TPR;
                    v[u][j] = (rr < r_end && vc < g.VC) ? load_vec2(x1, C1, x2, C2, img * S + rr, vc * 8) : make_uint4(0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    float f[8];
                    unpack8(v[u][j], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s[j][e] += f[e];
                        q[j][e] += f[e] * f[e];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int v = tcol + j * g.TPR;
            if (v < g.VC) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sh_s[(long long)trow * C + v * 8 + e] = s[j][e];
                    sh_q[(long long)trow * C + v * 8 + e] = q[j][e];
                }
            }
        }
    }
    __syncthreads();
    const int cpg = (int)(C / groups);
    if (tid < groups) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < g.RPP; ++r)
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {      // fixed order: the result does not depend on scheduling
                a += sh_s[(long long)r * C + c];
                b += sh_q[(long long)r * C + c];
            }
        const long long slot = (img % imgs_per_stat) * gridDim.x + blockIdx.x;
        float* dst = stats + (((img / imgs_per_stat) * nslots + slot) * groups + tid) * 2;
        if constexpr (FUSE) {
            // write-through (sc0 sc1) and waited for: the slot is at the coherence point before this block takes its ticket - no fence: an
            // agent-scope release fence writes back the whole L2 (the producing GEMM's output is still dirty in it), 25 us per launch measured
            const f32x2_t v = {a, b};
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(dst), "v"(v) : "memory");
        } else {
            dst[0] = a;
            dst[1] = b;
        }
    }
    if constexpr (FUSE) {
        __shared__ __attribute__((aligned(16))) unsigned ticket[4];       // (a 16-byte object: the dynamic LDS behind it keeps its alignment for the fold's ds_read_b64 / ds_write_b64)
        const long long st = img / imgs_per_stat;
        const unsigned writers = (unsigned)(gridDim.x * imgs_per_stat);        // blocks of this statistics group = slots it fills (slots 0 .. writers - 1)
        __syncthreads();                     // every slot store of the block has been waited for
        if (tid == 0) ticket[0] = __hip_atomic_fetch_add(fz.tickets + st, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (ticket[0] != writers - 1) return;   // (block-uniform)
        // the last block: every other block's slots were complete before its ticket; read them past the non-coherent cache levels.
        // The per-channel partials in gn_sh are dead (read above, barrier passed): the fold uses the same LDS
        gn_finalize_block<256, true>(reinterpret_cast<unsigned char*>(gn_sh), stats, nslots, (long long)writers, nullptr, groups, fz.gamma, fz.beta, C, fz.inv_count,
                                     fz.eps, fz.table, st);
        if (tid == 0) __hip_atomic_store(fz.tickets + st, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x1, long long C1,
                                                       const bf16_t* __restrict__ x2, long long C2,
                                                       const float* __restrict__ table,
                                                       bf16_t* __restrict__ out, long long S,
                                                       long long imgs_per_stat, int silu, long long rpb) {
    const long long C = C1 + C2;
    const GNGeom g = gn_geom(C);
    const int tid = threadIdx.x;
    const long long img = blockIdx.y;
    const long long r_begin = (long long)blockIdx.x * rpb;
    long long r_end = r_begin + rpb;
    if (r_end > S) r_end = S;
    const int trow = tid / g.TPR;
    const int tcol = tid - trow * g.TPR;
    if (trow >= g.RPP) return;
    const float* tb = table + (img / imgs_per_stat) * C * 2;

    float sc[NVMAX][8], sf[NVMAX][8];
#pragma unroll
    for (int j = 0; j < NVMAX; ++j) {
        const int v = tcol + j * g.TPR;
        if (j < g.NV && v < g.VC) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float4 t = *reinterpret_cast<const float4*>(tb + (v * 8 + e) * 2);
                sc[j][e] = t.x; sf[j][e] = t.y; sc[j][e + 1] = t.z; sf[j][e + 1] = t.w;
            }
        }
    }
    for (long long r = r_begin + trow; r < r_end; r += g.RPP) {
        const long long row = img * S + r;
#pragma unroll
        for (int j = 0; j < NVMAX; ++j) {
            const int v = tcol + j * g.TPR;
            if (j < g.NV && v < g.VC) {
                float f[8];
                unpack8(load_vec2(x1, C1, x2, C2, row, v * 8), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = f[e] * sc[j][e] + sf[j][e];
                    f[e] = silu ? silu_f(y) : y;
                }
                uint4 o = make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
                *reinterpret_cast<uint4*>(out + row * C + v * 8) = o;
            }
        }
    }
}

// A wave normalises RPW rows per pass, every row kept in registers (NVW vectors of 8 channels per lane: C <= 512 * NVW).
// The loads of all RPW rows are issued before the first reduction, so a wave has RPW x 16 B per lane in flight (the first
// version handled one row per wave and 4 rows per block: 57 us for the 188 MB of the 64x64 level, gn_apply moves the
// same bytes in 36 us).
constexpr int LNV = 4;
template <int NVW, int RPW>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, const float* __restrict__ add,
                                                        long long add_rpg, long long add_ld, bf16_t* __restrict__ xsum,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        bf16_t* __restrict__ out, long long M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= M) return;
    const int VC = C / 8;
    const float invC = 1.0f / (float)C;
    float f[RPW][NVW][8];
    float sum[RPW];
    uint4 raw[RPW][NVW];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int j = 0; j < NVW; ++j) {
            const int v = lane + 64 * j;
            raw[r][j] = (v < VC && row0 + r < M) ? *reinterpret_cast<const uint4*>(x + (row0 + r) * C + v * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long long row = row0 + r;
        const bool rok = row < M;
        const float* av = (add && rok) ? add + (row / add_rpg) * add_ld : nullptr;
        sum[r] = 0.f;
#pragma unroll
        for (int j = 0; j < NVW; ++j) {
            const int v = lane + 64 * j;
            unpack8(raw[r][j], f[r][j]);
            if (v < VC && av) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[r][j][e] += av[v * 8 + e];
                if (xsum) {
                    uint4 o = make_uint4(pack2bf(f[r][j][0], f[r][j][1]), pack2bf(f[r][j][2], f[r][j][3]), pack2bf(f[r][j][4], f[r][j][5]),
                                         pack2bf(f[r][j][6], f[r][j][7]));
                    *reinterpret_cast<uint4*>(xsum + row * C + v * 8) = o;
                    unpack8(o, f[r][j]);  // normalise exactly what was stored
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) sum[r] += f[r][j][e];   // lanes past the row hold zeros
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) sum[r] = wave_sum(sum[r]);
    float var[RPW], mean[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        mean[r] = sum[r] * invC;
        var[r] = 0.f;
#pragma unroll
        for (int j = 0; j < NVW; ++j) {
            if (lane + 64 * j < VC) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = f[r][j][e] - mean[r];
                    var[r] += d * d;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) var[r] = wave_sum(var[r]);
#pragma unroll
    for (int j = 0; j < NVW; ++j) {
        const int v = lane + 64 * j;
        if (v < VC) {
            float gm[8], bt[8];
            *reinterpret_cast<float4*>(gm) = *reinterpret_cast<const float4*>(gamma + v * 8);
            *reinterpret_cast<float4*>(gm + 4) = *reinterpret_cast<const float4*>(gamma + v * 8 + 4);
            *reinterpret_cast<float4*>(bt) = *reinterpret_cast<const float4*>(beta + v * 8);
            *reinterpret_cast<float4*>(bt + 4) = *reinterpret_cast<const float4*>(beta + v * 8 + 4);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                if (row0 + r < M) {
                    const float rstd = rsqrtf(var[r] * invC + eps);
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = (f[r][j][e] - mean[r]) * rstd * gm[e] + bt[e];
                    uint4 o = make_uint4(pack2bf(y[0], y[1]), pack2bf(y[2], y[3]), pack2bf(y[4], y[5]), pack2bf(y[6], y[7]));
                    *reinterpret_cast<uint4*>(out + (row0 + r) * C + v * 8) = o;
                }
            }
        }
    }
}

// one block per row; L % 4 == 0
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long long L) {
    __shared__ float red[4];
    const long long row = blockIdx.x;
    const float* p = in + row * L;
    const int tid = threadIdx.x;
    float mx = -INFINITY;
    for (long long i = tid * 4; i < L; i += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(p + i);
        mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (long long i = tid * 4; i < L; i += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(p + i);
        sum += __expf(v.x - mx) + __expf(v.y - mx) + __expf(v.z - mx) + __expf(v.w - mx);
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    sum = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.0f / sum;
    bf16_t* o = out + row * L;
    for (long long i = tid * 4; i < L; i += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(p + i);
        const uint2 w = make_uint2(pack2bf(__expf(v.x - mx) * inv, __expf(v.y - mx) * inv),
                                   pack2bf(__expf(v.z - mx) * inv, __expf(v.w - mx) * inv));
        *reinterpret_cast<uint2*>(o + i) = w;
    }
}

inline void gn_grid(long long n_img, long long S, const GNGeom& g, long long& chunks, long long& rpb, long long target_blocks = 4096) {
    long long want = target_blocks / (n_img > 0 ? n_img : 1);
    if (want < 1) want = 1;
    long long maxchunks = (S + g.RPP - 1) / g.RPP;
    chunks = want < maxchunks ? want : maxchunks;
    if (chunks > 65535) chunks = 65535;
    rpb = (S + chunks - 1) / chunks;
    chunks = (S + rpb - 1) / rpb;
}

int gn_check(const char* who, const void* x1, long long C1, const void* x2, long long C2, long long n_img, long long S, int groups) {
    V3D_REQUIRE(x1 != nullptr && C1 > 0, "%s: null x1", who);
    V3D_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0, "%s: channel counts must be multiples of 8 (%lld,%lld)", who, C1, C2);
    V3D_REQUIRE((C2 == 0) == (x2 == nullptr), "%s: x2/C2 mismatch", who);
    V3D_REQUIRE(groups > 0 && groups <= 256 && (C1 + C2) % groups == 0, "%s: bad groups", who);
    V3D_REQUIRE(C1 + C2 <= CMAX, "%s: C > %d unsupported", who, CMAX);
    V3D_REQUIRE(n_img > 0 && n_img <= 65535 && S > 0, "%s: bad n_img/S", who);
    return V3D_OK;
}

}  // namespace

namespace {
int gn_stats_launch(const char* who, const void* x1, int64_t C1, const void* x2, int64_t C2, float* stats, int64_t nslots, int64_t n_img, int64_t S,
                    int32_t groups, int64_t imgs_per_stat, const GNFuse* fz, v3d_stream_t stream) {
    int rc = gn_check(who, x1, C1, x2, C2, n_img, S, groups);
    if (rc) return rc;
    V3D_REQUIRE(stats != nullptr && imgs_per_stat > 0 && n_img % imgs_per_stat == 0, "%s: bad stats/imgs_per_stat", who);
    V3D_REQUIRE(nslots >= imgs_per_stat, "%s: nslots (%lld) must be >= imgs_per_stat (%lld): one slot per block", who,
                (long long)nslots, (long long)imgs_per_stat);
    const GNGeom g = gn_geom(C1 + C2);
    long long chunks, rpb;
    static long long target = -1;
    if (target < 0) {
        const char* e = getenv("V3D_GN_BLOCKS");   // tuning knob (tools/gn_bench.py)
        target = e ? atoll(e) : 0;
    }
    // few, long blocks (rounds 1-2: every block ended in 64 global atomics; now one 8-byte store per group, but ~1 block per CU still
    // reads at the HBM rate); very large inputs (VAE, > 256 MB) want ~3 blocks per CU to keep HBM busy
    const long long bytes = n_img * S * (C1 + C2) * 2;
    gn_grid(n_img, S, g, chunks, rpb, target > 0 ? target : (bytes > (256ll << 20) ? 768 : 256));
    if (chunks * imgs_per_stat > nslots) {       // every block of a statistics group needs its own slot
        chunks = nslots / imgs_per_stat;
        rpb = (S + chunks - 1) / chunks;
        chunks = (S + rpb - 1) / rpb;
    }
    size_t shmem = (size_t)(C1 + C2) * 2 * g.RPP * sizeof(float);
    V3D_REQUIRE(shmem <= 64 * 1024, "%s: C=%lld needs %zu B of LDS", who, (long long)(C1 + C2), shmem);
    // the fused fold reuses the dynamic LDS: widths whose partial-sum table is smaller than the fold's scratch (1024 < C < 1120) get the larger of the two
    if (fz && shmem < (size_t)gn_fin_smem<256>()) shmem = (size_t)gn_fin_smem<256>();
    const dim3 grid((unsigned)chunks, (unsigned)n_img);
    const GNFuse none = {nullptr, nullptr, nullptr, nullptr, 0.0, 0.f};
    if (g.NV == 1) {
        if (fz)
            hipLaunchKernelGGL((gn_stats_kernel<1, 8, true>), grid, dim3(256), shmem, (hipStream_t)stream, (const bf16_t*)x1, (long long)C1,
                               (const bf16_t*)x2, (long long)C2, stats, (long long)nslots, (long long)S, groups, (long long)imgs_per_stat, rpb, *fz);
        else
            hipLaunchKernelGGL((gn_stats_kernel<1, 8>), grid, dim3(256), shmem, (hipStream_t)stream, (const bf16_t*)x1, (long long)C1,
                               (const bf16_t*)x2, (long long)C2, stats, (long long)nslots, (long long)S, groups, (long long)imgs_per_stat, rpb, none);
    } else {
        if (fz)
            hipLaunchKernelGGL((gn_stats_kernel<2, 4, true>), grid, dim3(256), shmem, (hipStream_t)stream, (const bf16_t*)x1, (long long)C1,
                               (const bf16_t*)x2, (long long)C2, stats, (long long)nslots, (long long)S, groups, (long long)imgs_per_stat, rpb, *fz);
        else
            hipLaunchKernelGGL((gn_stats_kernel<2, 4>), grid, dim3(256), shmem, (hipStream_t)stream, (const bf16_t*)x1, (long long)C1,
                               (const bf16_t*)x2, (long long)C2, stats, (long long)nslots, (long long)S, groups, (long long)imgs_per_stat, rpb, none);
    }
    return v3d_check_launch(who);
}
}  // namespace

extern "C" int v3d_groupnorm_stats(const void* x1, int64_t C1, const void* x2, int64_t C2, float* stats, int64_t nslots,
                                   int64_t n_img, int64_t S, int32_t groups, int64_t imgs_per_stat, v3d_stream_t stream) {
    return gn_stats_launch("v3d_groupnorm_stats", x1, C1, x2, C2, stats, nslots, n_img, S, groups, imgs_per_stat, nullptr, stream);
}

extern "C" int v3d_groupnorm_stats_table(const void* x1, int64_t C1, const void* x2, int64_t C2, float* stats, int64_t nslots, uint32_t* tickets,
                                         int64_t n_img, int64_t S, int32_t groups, int64_t imgs_per_stat, const float* gamma, const float* beta,
                                         double count, float eps, float* table, v3d_stream_t stream) {
    V3D_REQUIRE(tickets && gamma && beta && table && count > 0, "v3d_groupnorm_stats_table: null tickets / gamma / beta / table or count <= 0");
    V3D_REQUIRE(groups > 0 && groups <= 32 && groups % 2 == 0, "v3d_groupnorm_stats_table: groups must be even and <= 32");
    V3D_REQUIRE(((uintptr_t)stats & 15) == 0 && ((uintptr_t)table & 7) == 0 && ((uintptr_t)tickets & 3) == 0, "v3d_groupnorm_stats_table: stats 16-byte, table 8-byte aligned");
    const GNFuse fz = {tickets, gamma, beta, table, 1.0 / count, eps};
    return gn_stats_launch("v3d_groupnorm_stats_table", x1, C1, x2, C2, stats, nslots, n_img, S, groups, imgs_per_stat, &fz, stream);
}

extern "C" int v3d_groupnorm_finalize(const float* stats, int64_t nslots, double* sums, int64_t n_stat, int32_t groups,
                                      const float* gamma, const float* beta, int64_t C, double count, float eps, float* table,
                                      v3d_stream_t stream) {
    V3D_REQUIRE(stats || sums, "v3d_groupnorm_finalize: neither slot statistics nor reduced sums given");
    V3D_REQUIRE(!stats || nslots > 0, "v3d_groupnorm_finalize: nslots must be > 0");
    V3D_REQUIRE(n_stat > 0 && n_stat < (1ll << 31) && groups > 0 && groups <= 32 && groups % 2 == 0, "v3d_groupnorm_finalize: bad n_stat / groups (even, <= 32)");
    V3D_REQUIRE(!stats || ((uintptr_t)stats & 15) == 0, "v3d_groupnorm_finalize: stats must be 16-byte aligned");
    V3D_REQUIRE(table || (stats && sums), "v3d_groupnorm_finalize: nothing to write");
    if (table) {
        V3D_REQUIRE(gamma && beta && C > 0 && C % groups == 0 && count > 0, "v3d_groupnorm_finalize: table needs gamma, beta, C %% groups == 0, count > 0");
        V3D_REQUIRE(((uintptr_t)table & 7) == 0, "v3d_groupnorm_finalize: table must be 8-byte aligned");
    }
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)n_stat), dim3(1024), 0, (hipStream_t)stream, stats, (long long)nslots, sums, groups, gamma,
                       beta, (long long)C, count > 0 ? 1.0 / count : 0.0, eps, table);
    return v3d_check_launch("v3d_groupnorm_finalize");
}

extern "C" int v3d_groupnorm_apply(const void* x1, int64_t C1, const void* x2, int64_t C2, const float* table, void* out,
                                   int64_t n_img, int64_t S, int64_t imgs_per_stat, int32_t silu, v3d_stream_t stream) {
    int rc = gn_check("v3d_groupnorm_apply", x1, C1, x2, C2, n_img, S, 1);
    if (rc) return rc;
    V3D_REQUIRE(table && out, "v3d_groupnorm_apply: null pointer");
    V3D_REQUIRE(((uintptr_t)table & 15) == 0, "v3d_groupnorm_apply: table must be 16-byte aligned");
    V3D_REQUIRE(imgs_per_stat > 0 && n_img % imgs_per_stat == 0, "v3d_groupnorm_apply: bad imgs_per_stat");
    const GNGeom g = gn_geom(C1 + C2);
    long long chunks, rpb;
    gn_grid(n_img, S, g, chunks, rpb);
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)chunks, (unsigned)n_img), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x1, (long long)C1, (const bf16_t*)x2, (long long)C2, table,
                       (bf16_t*)out, (long long)S, (long long)imgs_per_stat, silu, rpb);
    return v3d_check_launch("v3d_groupnorm_apply");
}

// ---- GroupNorm of a SMALL statistics group in one launch ---------------------------------------------------------------------------------
// The 8 x 8 level of the U-Net (and the 16 x 16 level's transformer norms) ran three launches per GroupNorm - statistics 8-12 us, finalize
// 7 us, apply 7-13 us, two kernel boundaries - on tensors of 6-24 MB: launch-bound.  Here a block owns (statistics group, GB channel
// groups): it loads its slice once into registers (<= NVT 16-byte vectors per thread), reduces (sum, sumsq) per channel group in a fixed
// order (vector partials -> LDS -> one wave per group, lane partials added in lane order by a shuffle tree), and normalises what it holds.
// Deterministic like the three-step form; same arithmetic for the output (x * scale + shift, SiLU, bf16) with scale / shift from fp64 sums.
template <int NVT>
__global__ __launch_bounds__(256) void gn_small_kernel(const bf16_t* __restrict__ x1, long long C1, const bf16_t* __restrict__ x2, long long C2,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ out,
                                                       long long rows, int cpg, int GB, float eps, int silu) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gn_small_lds[];
    float2* part = reinterpret_cast<float2*>(gn_small_lds);                       // [rows * VB] per-vector (sum, sumsq)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long C = C1 + C2;
    const int VB = GB * cpg / 8;                                                  // vectors per row of the block's channel slice
    const int vpg = cpg / 8;                                                      // vectors per channel group and row
    const long long row0 = (long long)blockIdx.y * rows;
    const int c0 = blockIdx.x * GB * cpg;
    const int NV = (int)rows * VB;
    float2* scsh = part + NV;                                                     // [GB * cpg] (scale, shift) of the block's channels
    float2* gstat = scsh + GB * cpg;                                              // [GB] (mean, rstd)
    uint4 v[NVT];
    // vector i of the block = (row i / VB, column i % VB); a thread's vectors are 256 apart: (r, vc) advance by constant steps - no division
    // per vector (the runtime divisors cost ~30 instructions each, more than the arithmetic they index)
    const int dr = 256 / VB, dc = 256 - dr * VB;
    int r = tid / VB, vc = tid - r * VB;
    const int r_first = r, vc_first = vc;
    // all loads first (vectors past the slice re-read its first vector: no branch between the loads, NVT of them in flight per thread)
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
        const bool ok = tid + 256 * k < NV;
        v[k] = load_vec2(x1, C1, x2, C2, row0 + (ok ? r : 0), c0 + (ok ? vc : 0) * 8);
        r += dr; vc += dc;
        if (vc >= VB) { vc -= VB; ++r; }
    }
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
        const int i = tid + 256 * k;
        if (i < NV) {
            float f[8];
            unpack8(v[k], f);
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { sm += f[e]; sq += f[e] * f[e]; }
            part[i] = make_float2(sm, sq);
        }
    }
    __syncthreads();
    // group g of the block: vectors (row, g * vpg + j), j < vpg; wave w reduces groups w, w + 4, ...: a lane walks rows lane, lane + 64, ...
    // (fixed order), the 64 lane partials meet in a shuffle tree
    for (int g = wave; g < GB; g += 4) {
        double sm = 0.0, sq = 0.0;
        for (int rr = lane; rr < (int)rows; rr += 64) {
            const float2* pr = part + rr * VB + g * vpg;
            for (int j = 0; j < vpg; ++j) {
                sm += (double)pr[j].x;
                sq += (double)pr[j].y;
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            sm += __shfl_xor(sm, o, 64);
            sq += __shfl_xor(sq, o, 64);
        }
        if (lane == 0) {
            const double cnt = (double)rows * cpg;
            const double mean = sm / cnt;
            double var = sq / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            gstat[g] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
        }
    }
    __syncthreads();
    // the table form of v3d_groupnorm_finalize, once per channel of the block: scale = gamma rstd, shift = beta - mean scale
    for (int c = tid; c < GB * cpg; c += 256) {
        const float2 ms = gstat[c / cpg];
        const float sc = gamma[c0 + c] * ms.y;
        scsh[c] = make_float2(sc, beta[c0 + c] - ms.x * sc);
    }
    __syncthreads();
    r = r_first; vc = vc_first;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
        const int i = tid + 256 * k;
        if (i < NV) {
            float f[8];
            unpack8(v[k], f);
            const float4* t = reinterpret_cast<const float4*>(scsh + vc * 8);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float4 q = t[e / 2];                         // (scale, shift) of channels e, e + 1
                const float y0 = f[e] * q.x + q.y, y1 = f[e + 1] * q.z + q.w;
                f[e] = silu ? silu_f(y0) : y0;
                f[e + 1] = silu ? silu_f(y1) : y1;
            }
            *reinterpret_cast<uint4*>(out + (row0 + r) * C + c0 + vc * 8) = make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
        }
        r += dr; vc += dc;
        if (vc >= VB) { vc -= VB; ++r; }
    }
}

// channel groups per block for the one-launch form, 0 = the tensor does not fit it (the caller keeps statistics -> finalize -> apply)
static int gn_small_groups(long long rows, long long C1, long long C2, int groups) {
    const long long C = C1 + C2;
    if (groups != 32 || C % 32 || C1 % 8 || rows <= 0) return 0;
    const long long cpg = C / 32;
    if (cpg % 8) return 0;                                        // a 16-byte vector must lie inside one channel group
    static int gb_max = -1;
    if (gb_max < 0) {
        const char* e = getenv("V3D_GN_SMALL_GB");                // tuning knob (tools/gn_small_bench.py): largest channel-group count per block
        gb_max = e ? atoi(e) : 1;      // round 5 (profiles/r05_gn_small_gb.txt): the smallest slice with >= 64-byte row segments wins or ties everywhere
                                       // (16 x 16 transformer norm 29.6 -> 20.0 us, two-source 8 x 8 norm 17.8 -> 12.5 us): more, shorter blocks
        if (gb_max < 1) gb_max = 1;
    }
    for (int gb = 8; gb >= 1; gb >>= 1) {
        if (gb > gb_max && gb > 1 && (gb / 2) * cpg * 2 >= 64) continue;      // (a smaller slice still has row segments >= 64 bytes)
        if (C1 % (gb * cpg) && C2) continue;                      // a block's slice must lie inside one source
        if (rows * (gb * cpg / 8) <= 256 * 24 && gb * cpg * 2 >= 64) return gb;
    }
    return 0;
}

extern "C" int v3d_groupnorm_small_supported(int64_t C1, int64_t C2, int64_t S, int32_t groups, int64_t imgs_per_stat) {
    return gn_small_groups(imgs_per_stat * S, C1, C2, groups) > 0;
}

extern "C" int v3d_groupnorm_small(const void* x1, int64_t C1, const void* x2, int64_t C2, const float* gamma, const float* beta, void* out,
                                   int64_t n_img, int64_t S, int32_t groups, int64_t imgs_per_stat, float eps, int32_t silu, v3d_stream_t stream) {
    int rc = gn_check("v3d_groupnorm_small", x1, C1, x2, C2, n_img, S, groups);
    if (rc) return rc;
    V3D_REQUIRE(gamma && beta && out, "v3d_groupnorm_small: null pointer");
    V3D_REQUIRE((((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out) & 15) == 0, "v3d_groupnorm_small: gamma / beta / out must be 16-byte aligned");
    V3D_REQUIRE(imgs_per_stat > 0 && n_img % imgs_per_stat == 0, "v3d_groupnorm_small: bad imgs_per_stat");
    const long long rows = imgs_per_stat * S;
    const int gb = gn_small_groups(rows, C1, C2, groups);
    V3D_REQUIRE(gb > 0, "v3d_groupnorm_small: this tensor does not fit the one-launch form (v3d_groupnorm_small_supported)");
    const int cpg = (int)((C1 + C2) / 32);
    const long long nv = rows * (gb * cpg / 8);
    const size_t lds = (size_t)nv * 8 + (size_t)gb * cpg * 8 + 64;
    dim3 grid((unsigned)(32 / gb), (unsigned)(n_img / imgs_per_stat));
#define V3D_GNS_LAUNCH(NVT_)                                                                                                                    \
    hipLaunchKernelGGL((gn_small_kernel<NVT_>), grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)x1, (long long)C1, (const bf16_t*)x2,     \
                       (long long)C2, gamma, beta, (bf16_t*)out, rows, cpg, gb, eps, silu)
    if (nv <= 256 * 8) V3D_GNS_LAUNCH(8);
    else if (nv <= 256 * 16) V3D_GNS_LAUNCH(16);
    else V3D_GNS_LAUNCH(24);
#undef V3D_GNS_LAUNCH
    return v3d_check_launch("v3d_groupnorm_small");
}

extern "C" int v3d_layernorm(const void* x, const float* add, int64_t add_rpg, int64_t add_ld, void* xsum_out,
                             const float* gamma, const float* beta, void* out, int64_t M, int64_t C, float eps,
                             v3d_stream_t stream) {
    V3D_REQUIRE(x && gamma && beta && out, "v3d_layernorm: null pointer");
    V3D_REQUIRE(C % 8 == 0 && C > 0 && C <= 64 * 8 * LNV, "v3d_layernorm: C=%lld unsupported (multiple of 8, <= %d)", (long long)C, 64 * 8 * LNV);
    V3D_REQUIRE(M > 0, "v3d_layernorm: bad M");
    V3D_REQUIRE(!add || add_rpg > 0, "v3d_layernorm: add_rpg must be > 0");
    V3D_REQUIRE(!xsum_out || add, "v3d_layernorm: xsum_out requires add");
    V3D_REQUIRE((((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "v3d_layernorm: gamma/beta must be 16-byte aligned");
    const int nvw = C <= 512 ? 1 : (C <= 1024 ? 2 : 4);
    const int rpw = nvw == 1 ? 4 : (nvw == 2 ? 2 : 1);
    const long long blocks = (M + 4 * rpw - 1) / (4 * rpw);
    V3D_REQUIRE(blocks < (1ll << 31), "v3d_layernorm: M too large");
#define V3D_LN_LAUNCH(NVW_, RPW_)                                                                                                   \
    hipLaunchKernelGGL((layernorm_kernel<NVW_, RPW_>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, add, \
                       (long long)add_rpg, (long long)add_ld, (bf16_t*)xsum_out, gamma, beta, (bf16_t*)out, (long long)M, (int)C, eps)
    if (nvw == 1) V3D_LN_LAUNCH(1, 4);
    else if (nvw == 2) V3D_LN_LAUNCH(2, 2);
    else V3D_LN_LAUNCH(4, 1);
#undef V3D_LN_LAUNCH
    return v3d_check_launch("v3d_layernorm");
}

extern "C" int v3d_softmax_rows(const float* in, void* out, int64_t rows, int64_t L, v3d_stream_t stream) {
    V3D_REQUIRE(in && out && rows > 0 && L > 0 && L % 4 == 0, "v3d_softmax_rows: bad args");
    V3D_REQUIRE(rows < (1ll << 31), "v3d_softmax_rows: too many rows");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, (long long)L);
    return v3d_check_launch("v3d_softmax_rows");
}
