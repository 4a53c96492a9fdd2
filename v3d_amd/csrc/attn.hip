// Attention kernels for gfx950: spatial (streamed-softmax MFMA, d_head 64) and temporal (T <= 32 frames).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------------
// Spatial self-attention.  Block = 4 waves = 128 queries of one (image, head); each wave owns 32 queries.
// Per 32-key sub-tile a wave computes S^T = K . Q^T with v_mfma_f32_32x32x16_bf16 (A = K rows from LDS,
// B = Q held in registers), so each lane owns one query column and 16 of its 32 scores: the row max / sum need
// one cross-lane exchange (lane <-> lane+32) and P^T is already in B-operand position for O^T += V^T . P^T.
// The contraction index of that second MFMA is a permutation of the keys that is applied identically to the
// A operand (V^T read from LDS as two 8-byte pieces), so no transpose or lane shuffle of P is needed.
// V arrives pre-transposed ([C][S], produced directly by the value projection GEMM).
// ------------------------------------------------------------------------------------------------------
constexpr int AKT = 64;          // keys per LDS tile
constexpr int AROW = 64 + 8;     // padded LDS row (bf16)

__global__ __launch_bounds__(256, 2) void attn_spatial_kernel(const bf16_t* __restrict__ q, long long ldq,
                                                              const bf16_t* __restrict__ k, long long ldk,
                                                              const bf16_t* __restrict__ vT, bf16_t* __restrict__ out,
                                                              long long ldo, long long S, int heads, float scale2) {
    __shared__ __attribute__((aligned(16))) bf16_t sK[2][AKT * AROW];
    __shared__ __attribute__((aligned(16))) bf16_t sV[2][64 * AROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = lane & 31, hi = lane >> 5;
    const long long n = blockIdx.z;
    const int h = blockIdx.y;
    const long long C = (long long)heads * 64;
    const long long qrow = (long long)blockIdx.x * 128 + wave * 32 + qi;
    const bool q_ok = qrow < S;

    // per-(image, head) buffer descriptors: rows / keys past S read as zeros in hardware
    const bufrsrc_t rsQ = make_rsrc(q + n * S * ldq + h * 64, (unsigned)(((S - 1) * ldq + 64) * 2));
    const bufrsrc_t rsK = make_rsrc(k + n * S * ldk + h * 64, (unsigned)(((S - 1) * ldk + 64) * 2));
    const bufrsrc_t rsV = make_rsrc(vT + (n * C + h * 64) * S, (unsigned)(64 * S * 2));

    bf16x8 qf[4];
    {
        const unsigned qoff = q_ok ? (unsigned)((qrow * ldq + hi * 8) * 2) : kInvalid;
#pragma unroll
        for (int st = 0; st < 4; ++st)
            qf[st] = __builtin_bit_cast(bf16x8, buf_load16(rsQ, q_ok ? qoff + st * 32 : kInvalid));
    }

    // staging: thread owns chunks c = tid + 256*i, row = c>>3, kc = c&7
    const int srow0 = tid >> 3, skc = tid & 7;
    u32x4 rk[2], rv[2];
    const int ntiles = (int)((S + AKT - 1) / AKT);
    auto gload = [&](int t) {
        const long long k0 = (long long)t * AKT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = srow0 + 32 * i;
            rk[i] = buf_load16(rsK, (k0 + row < S) ? (unsigned)(((k0 + row) * ldk + skc * 8) * 2) : kInvalid);
            rv[i] = buf_load16(rsV, (k0 + skc * 8 < S) ? (unsigned)(((long long)row * S + k0 + skc * 8) * 2) : kInvalid);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = srow0 + 32 * i;
            *reinterpret_cast<u32x4*>(&sK[buf][row * AROW + skc * 8]) = rk[i];
            *reinterpret_cast<u32x4*>(&sV[buf][row * AROW + skc * 8]) = rv[i];
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    gload(0);
    lstore(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        const long long k0 = (long long)t * AKT;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x16 sT;
#pragma unroll
            for (int r = 0; r < 16; ++r) sT[r] = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&sK[buf][(sub * 32 + qi) * AROW + st * 16 + hi * 8]);
                sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], sT, 0, 0, 0);
            }
            // sT[r] = score(key = sub*32 + (r&3) + 8*(r>>2) + 4*hi, query = qi)
            float mx = -INFINITY;
            const bool tail = (k0 + AKT > S);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s = sT[r] * scale2;
                if (tail) {
                    const long long key = k0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= S) s = -INFINITY;
                }
                sT[r] = s;
                mx = fmaxf(mx, s);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            // m_new is finite from sub-tile 0 of tile 0 on (it always holds a valid key); bare v_exp_f32, see the v2 kernel
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float psum = 0.f;
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(sT[r] - m_new);
                psum += p[r];
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[0][r] *= alpha;
                o[1][r] *= alpha;
            }
            bf16x8 pf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 u = make_uint4(pack2bf(p[ks * 8 + 0], p[ks * 8 + 1]), pack2bf(p[ks * 8 + 2], p[ks * 8 + 3]),
                                     pack2bf(p[ks * 8 + 4], p[ks * 8 + 5]), pack2bf(p[ks * 8 + 6], p[ks * 8 + 7]));
                pf[ks] = __builtin_bit_cast(bf16x8, u);
            }
#pragma unroll
            for (int db = 0; db < 2; ++db) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16_t* vp = &sV[buf][(db * 32 + qi) * AROW + sub * 32 + ks * 16 + hi * 4];
                    const uint2 lo = *reinterpret_cast<const uint2*>(vp);
                    const uint2 hi2 = *reinterpret_cast<const uint2*>(vp + 8);
                    const bf16x8 vf = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi2.x, hi2.y));
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[ks], o[db], 0, 0, 0);
                }
            }
        }
        if (t + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_ok) {
        bf16_t* op = out + (n * S + qrow) * ldo + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + 8 * g + 4 * hi;
                const uint2 w = make_uint2(pack2bf(o[db][g * 4 + 0] * inv, o[db][g * 4 + 1] * inv),
                                           pack2bf(o[db][g * 4 + 2] * inv, o[db][g * 4 + 3] * inv));
                *reinterpret_cast<uint2*>(op + d0) = w;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Spatial self-attention v2 (default).  Same mathematics and the same operand trick as v1 (S^T = K.Q^T so that a lane
// owns a query column; the P.V contraction index is a key permutation applied identically to V^T), plus:
//   * K / V^T tiles arrive by LDS-DMA (global_load_lds_dwordx4) into a 3-deep ring, counted vmcnt, one barrier per tile;
//   * 128-byte LDS rows with the 16-byte chunk position XOR-swizzled by (row >> 1) & 7 (on the DMA source address):
//     every ds_read_b128 group of the K and the V^T fragment reads hits 16 distinct slots;
//   * the K rows are assigned to MFMA rows through pi(8A + 4h + c) = 16(A>>1) + 8h + 4(A&1) + c, which makes the 8 P
//     values a lane feeds into one P.V k-step 8 CONTIGUOUS keys -> V^T fragments are single ds_read_b128;
//   * QG query groups of 32 per wave share every K / V^T fragment read (QG = 2 for long sequences);
//   * one softmax update per 64-key tile (not per 32), O rescale skipped when no running max moved in the wave.
// ------------------------------------------------------------------------------------------------------

// round 5 (profiles/r05_attn_ab.txt, same-process A/B): deferred maximum + no SLP packing of the softmax arithmetic (v3d_amd/build.py FILE_FLAGS:
// v_pk_mul_f32 / v_pk_add_f32 beside MFMAs cost more than the scalar pairs they replace) 917 -> 874 us at S = 4096, 133.6 -> 128.1 us at S = 1024
#ifndef ATTN_DEFER_MAX
#define ATTN_DEFER_MAX 8
#endif
// round 6: the softmax loop is VALU-bound (per score and lane: v_fma 4 + v_exp_f32 16 + v_add 4 + v_max 4 + half a v_cvt_pk 2 = 30 cycles of the SIMD's
// vector ALU against 16 cycles of its matrix pipe), so two of those go away:
//   * Q is scaled by scale * log2(e) ONCE, when its fragments are loaded (re-rounded to bf16: 2^-9 relative per element, below the bf16 rounding of P),
//     and the score accumulators start at -m_run instead of 0: the MFMAs deliver s' - m_run, and while the running maximum stands (the deferred
//     maximum: nearly every tile after the first) the weight is a bare v_exp_f32 of the accumulator - no v_fma per score;
//   * the tile maximum is gathered with v_max3_f32 (two scores per instruction).
// A tile that raises the maximum by more than 2^ATTN_DEFER_MAX takes the wave-uniform slow path (one v_sub per score), as does tile 0.
#ifndef ATTN_FUSE_MAX
#define ATTN_FUSE_MAX 1
#endif
// round 6: inside a tile the P.V MFMAs of the first 32-key half run beside the exponentials of the second half (one scheduling region, the interleave fixed with
// sched_group_barrier: a wave's own MFMAs hide a few VALU / transcendental issues each) instead of "all exponentials, then all MFMAs".  Same arithmetic, same order
// of every accumulation: bit-identical results.  0 = the sequential form (A/B).
#ifndef ATTN_PIPE
#define ATTN_PIPE 1
#endif

template <int QG>
__global__ __launch_bounds__(256, 2) void attn_spatial_v2_kernel(const bf16_t* __restrict__ q, long long ldq,
                                                                 const bf16_t* __restrict__ k, long long ldk,
                                                                 const bf16_t* __restrict__ vT, bf16_t* __restrict__ out,
                                                                 long long ldo, long long S, int heads, float scale2) {
    constexpr int NS = 3;
    constexpr int TILE_BYTES = 2 * 64 * 128;   // K tile (64 keys x 128 B) + V^T tile (64 d-rows x 128 B)
    constexpr int PIECES = 4;                  // per wave per tile: 2 K pieces + 2 V pieces (8 rows x 128 B each)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31, hi = lane >> 5;
    const long long n = blockIdx.z;
    const int h = blockIdx.y;
    const long long C = (long long)heads * 64;
    const long long qbase = (long long)blockIdx.x * (128 * QG) + wave * (32 * QG);

    const bufrsrc_t rsQ = make_rsrc(q + n * S * ldq + h * 64, (unsigned)(((S - 1) * ldq + 64) * 2));
    bf16x8 qf[QG][4];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const long long qrow = qbase + g * 32 + qi;
#pragma unroll
        for (int st = 0; st < 4; ++st)
            qf[g][st] = __builtin_bit_cast(bf16x8, buf_load16(rsQ, qrow < S ? (unsigned)((qrow * ldq + hi * 8 + st * 16) * 2) : kInvalid));
    }
    if (ATTN_FUSE_MAX) {      // scores come out of the MFMAs in the exp2 domain
#pragma unroll
        for (int g = 0; g < QG; ++g)
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const u32x4 u = __builtin_bit_cast(u32x4, qf[g][st]);
                u32x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = pack2bf(bflo(u[e]) * scale2, bfhi(u[e]) * scale2);
                qf[g][st] = __builtin_bit_cast(bf16x8, w);
            }
    }

    // ---- LDS-DMA sources: piece = 8 rows x 128 B; lane l -> row (l >> 3), chunk position (l & 7), logical chunk pos ^ swz.
    //      Raw buffer loads to LDS with the whole byte offset in a VGPR that advances by a constant per tile (one v_add per piece):
    //      keys past the end of the sequence fall outside the K descriptor and arrive as zeros; the V^T tail columns of the last
    //      tile read the start of the next d-row (finite values, or zeros past the slab) and meet P = 0 there.  (The first version
    //      recomputed 64-bit pointers with bounds selects per tile: ~60 VALU / SALU instructions, 15 % of the loop.)
    const int prow = lane >> 3;
    const bufrsrc_t rsK = make_rsrc(k + n * S * ldk + h * 64, (unsigned)(((S - 1) * ldk + 64) * 2));
    const bufrsrc_t rsV = make_rsrc(vT + (n * C + h * 64) * S, (unsigned)(64 * S * 2));
    unsigned koffs[2], voffs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + prow;          // key row within the tile / d row of V^T
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        koffs[i] = (unsigned)((row * ldk + ch * 8) * 2);
        voffs[i] = (unsigned)(((long long)row * S + ch * 8) * 2);
    }
    const unsigned kstep = (unsigned)(64 * ldk * 2);
    const int ntiles = (int)((S + 63) / 64);
    auto issue = [&](int stage) {
        unsigned char* sb = lds + stage * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (__attribute__((address_space(3))) void*)(sb + (wave * 2 + i) * 1024), 16, (int)koffs[i], 0, 0, 0);
            koffs[i] += kstep;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (__attribute__((address_space(3))) void*)(sb + 8192 + (wave * 2 + i) * 1024), 16, (int)voffs[i], 0, 0, 0);
            voffs[i] += 128u;
        }
    };

    // fragment read offsets.  K: MFMA row i = qi holds key pi(i); logical chunk = 2*st + hi.
    const int A_ = qi >> 3, hh = (qi >> 2) & 1, cc = qi & 3;
    const int kkey = 16 * (A_ >> 1) + 8 * hh + 4 * (A_ & 1) + cc;                 // pi(qi)
    int koff[2][4];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int row = sub * 32 + kkey;
            koff[sub][st] = row * 128 + (((2 * st + hi) ^ ((row >> 1) & 7)) * 16);
        }
    // V^T: row d = db*32 + qi, logical chunk = sub*4 + ks*2 + hi (8 contiguous keys)
    int voff[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) voff[db] = 8192 + (db * 32 + qi) * 128;
    const int vsw = (qi >> 1) & 7;     // (row >> 1) & 7 with row = db*32 + qi  (db*32 >> 1 is a multiple of 8)

    f32x16 o[QG][2];
    float m_run[QG], l_run[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[g][0][r] = o[g][1][r] = 0.f;
        m_run[g] = -INFINITY;
        l_run[g] = 0.f;
    }

#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s);

    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (NS - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue((t + NS - 1) % NS);
        const unsigned char* sb = lds + (t % NS) * TILE_BYTES;
        const long long k0 = (long long)t * 64;
        const bool tail = (k0 + 64 > S);

        // ---- S^T = K . Q^T for both 32-key halves ----
        f32x16 sT[QG][2];
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            // (fused form: the accumulators start at -m_run, finite from tile 1 on)
            const float init = (ATTN_FUSE_MAX && t > 0) ? -m_run[g] : 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) sT[g][sub][r] = init;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sb + koff[sub][st]);
#pragma unroll
                for (int g = 0; g < QG; ++g) sT[g][sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[g][st], sT[g][sub], 0, 0, 0);
            }
        // sT[g][sub][r] = score(key = k0 + sub*32 + 16*(r>>3) + 8*hi + (r&7), query qbase + g*32 + qi)
        bf16x8 pf[QG][2][2];
        bool any_rescale = false;
        float alpha[QG];
        if (tail) {   // wave-uniform, last tile only: keys past the end of the sequence must not win the max or add to the sum
#pragma unroll
            for (int g = 0; g < QG; ++g)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((k0 + sub * 32 + 16 * (r >> 3) + 8 * hi + (r & 7)) >= S) sT[g][sub][r] = -INFINITY;
        }
        if (ATTN_FUSE_MAX) {
            // sT = s' - base in the exp2 domain (base = m_run, 0 on tile 0).  d = what the running maximum grows by (0 while it stands)
            float d[QG];
            bool need_any = false;
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                float mx = fmaxf(sT[g][0][0], sT[g][1][0]);
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = __builtin_fmaxf(__builtin_fmaxf(mx, sT[g][0][r]), sT[g][1][r]);        // (v_max3_f32)
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const bool need = (t == 0) || mx > (float)ATTN_DEFER_MAX;       // (every tile holds >= 1 valid key: mx is finite)
                d[g] = need ? mx : 0.f;
                alpha[g] = t == 0 ? 0.f : __builtin_amdgcn_exp2f(-d[g]);        // (tile 0: o = l = 0)
                any_rescale |= need;
                need_any |= need;
                m_run[g] = (t == 0 ? 0.f : m_run[g]) + d[g];
            }
            const bool slow = __any(need_any);
            if (ATTN_PIPE && !slow) {
                // fast path (the running maxima stand: alpha = 1, no rescale): weights of half 0, then { P.V of half 0 | weights of half 1 }, then P.V of half 1
                float psum[QG];
                auto weights = [&](auto sub_) __attribute__((always_inline)) {
                    constexpr int sub = decltype(sub_)::value;
#pragma unroll
                    for (int g = 0; g < QG; ++g) {
                        float pv[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) pv[r] = __builtin_amdgcn_exp2f(sT[g][sub][r]);
#pragma unroll
                        for (int r = 0; r < 16; ++r) psum[g] += pv[r];
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            const u32x4 u = {pack2bf(pv[ks * 8 + 0], pv[ks * 8 + 1]), pack2bf(pv[ks * 8 + 2], pv[ks * 8 + 3]),
                                             pack2bf(pv[ks * 8 + 4], pv[ks * 8 + 5]), pack2bf(pv[ks * 8 + 6], pv[ks * 8 + 7])};
                            pf[g][sub][ks] = __builtin_bit_cast(bf16x8, u);
                        }
                    }
                };
                auto pv_half = [&](auto sub_) __attribute__((always_inline)) {
                    constexpr int sub = decltype(sub_)::value;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int ch = ((sub * 4 + ks * 2 + hi) ^ vsw) * 16;
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sb + voff[db] + ch);
#pragma unroll
                            for (int g = 0; g < QG; ++g) o[g][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[g][sub][ks], o[g][db], 0, 0, 0);
                        }
                    }
                };
#pragma unroll
                for (int g = 0; g < QG; ++g) psum[g] = 0.f;
                weights(std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);
                pv_half(std::integral_constant<int, 0>{});
                weights(std::integral_constant<int, 1>{});
                // 4 QG MFMAs beside 16 QG exponentials + 16 QG adds + 8 QG conversions: one MFMA, (every other one) a V^T fragment read, then its share of the vector work
#pragma unroll
                for (int i = 0; i < 4 * QG; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i % QG == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x402, 10, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                pv_half(std::integral_constant<int, 1>{});
#pragma unroll
                for (int g = 0; g < QG; ++g) l_run[g] += psum[g];
                continue;
            }
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                float psum = 0.f;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    float pv[16];
                    if (slow) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) pv[r] = __builtin_amdgcn_exp2f(sT[g][sub][r] - d[g]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) pv[r] = __builtin_amdgcn_exp2f(sT[g][sub][r]);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) psum += pv[r];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const u32x4 u = {pack2bf(pv[ks * 8 + 0], pv[ks * 8 + 1]), pack2bf(pv[ks * 8 + 2], pv[ks * 8 + 3]),
                                         pack2bf(pv[ks * 8 + 4], pv[ks * 8 + 5]), pack2bf(pv[ks * 8 + 6], pv[ks * 8 + 7])};
                        pf[g][sub][ks] = __builtin_bit_cast(bf16x8, u);
                    }
                }
                l_run[g] = l_run[g] * alpha[g] + psum;
            }
        } else {
        // softmax in the exp2 domain on RAW scores: p = exp2(s * scale2 - m * scale2) is one v_fma + one bare v_exp_f32 per
        // score (libm's exp2f wraps every v_exp_f32 in a denormal-range compare / select / ldexp: ~6 extra VALU per score, and
        // the per-score scale multiply and -inf selects were another 3 - the loop was VALU-bound on them)
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            float mx = -INFINITY;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sT[g][sub][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float m_new = fmaxf(m_run[g], mx);                // raw-score domain; every tile holds >= 1 valid key, so finite
            // deferred maximum (ATTN_DEFER_MAX > 0): a running maximum that grew by less than 2^ATTN_DEFER_MAX keeps its old value - the
            // tile's weights are then at most 2^ATTN_DEFER_MAX (exact in fp32 / bf16 alike: the format is scale-free) and alpha = 1, so the
            // 128-multiply rescale of O^T below, which some lane of a 128-query wave triggers on almost every one of the 64 tiles at
            // S = 4096, runs only while the first tiles settle.  O / l is the same quotient either way.  (-inf start: the difference is +inf)
            if (ATTN_DEFER_MAX > 0 && (m_new - m_run[g]) * scale2 <= (float)ATTN_DEFER_MAX) m_new = m_run[g];
            alpha[g] = __builtin_amdgcn_exp2f((m_run[g] - m_new) * scale2);   // first tile: exp2(-inf) = 0 on o = l = 0
            any_rescale |= (alpha[g] != 1.f);
            const float mb = -m_new * scale2;
            float psum = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                float pv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pv[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sT[g][sub][r], scale2, mb));
                    psum += pv[r];
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const u32x4 u = {pack2bf(pv[ks * 8 + 0], pv[ks * 8 + 1]), pack2bf(pv[ks * 8 + 2], pv[ks * 8 + 3]),
                                     pack2bf(pv[ks * 8 + 4], pv[ks * 8 + 5]), pack2bf(pv[ks * 8 + 6], pv[ks * 8 + 7])};
                    pf[g][sub][ks] = __builtin_bit_cast(bf16x8, u);
                }
            }
            l_run[g] = l_run[g] * alpha[g] + psum;
            m_run[g] = m_new;
        }
        }
        if (__any(any_rescale)) {
#pragma unroll
            for (int g = 0; g < QG; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o[g][0][r] *= alpha[g];
                    o[g][1][r] *= alpha[g];
                }
        }
        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ch = ((sub * 4 + ks * 2 + hi) ^ vsw) * 16;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sb + voff[db] + ch);
#pragma unroll
                    for (int g = 0; g < QG; ++g) o[g][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[g][sub][ks], o[g][db], 0, 0, 0);
                }
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const long long qrow = qbase + g * 32 + qi;
        const float l_tot = l_run[g] + __shfl_xor(l_run[g], 32, 64);
        const float inv = 1.0f / l_tot;
        if (qrow < S) {
            bf16_t* op = out + (n * S + qrow) * ldo + h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const int d0 = db * 32 + 8 * gg + 4 * hi;
                    *reinterpret_cast<uint2*>(op + d0) = make_uint2(pack2bf(o[g][db][gg * 4 + 0] * inv, o[g][db][gg * 4 + 1] * inv),
                                                                    pack2bf(o[g][db][gg * 4 + 2] * inv, o[g][db][gg * 4 + 3] * inv));
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Temporal self-attention (frame axis).  A "problem" is one (sample b, position s, head h): Tq x Tk scores,
// d_head 64.  G = 64 / Tq problems share a wave; lane (slot, i) owns query frame i of its slot's problem, keeps
// q and the output row in registers and walks the Tk keys/values staged in LDS (all lanes of a slot read the same
// LDS address -> broadcast).  Loads follow the frame stride of the channels-last layout directly, so the
// "(b t) s c -> (b s) t c" transposes of the reference never happen.
// ------------------------------------------------------------------------------------------------------
constexpr int TMAX = 32;

struct TP {
    const bf16_t* q; long long q_sb, q_st, q_ss;
    const bf16_t* k; const bf16_t* v; long long kv_sb, kv_st, kv_ss;
    bf16_t* out; long long o_sb, o_st, o_ss;
    long long P, S; int heads, Tq, Tk, G; float scale;
};

__global__ __launch_bounds__(256) void attn_temporal_kernel(TP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_wave = p.G * p.Tk * 64;  // bf16 elements of K (and of V) per wave
    bf16_t* sK = reinterpret_cast<bf16_t*>(smem_raw) + (size_t)wave * 2 * per_wave;
    bf16_t* sV = sK + per_wave;

    const long long wave_id = (long long)blockIdx.x * 4 + wave;
    const long long p0 = wave_id * p.G;

    // stage K and V rows: chunk c -> (row = c>>3, kc = c&7), row -> (slot, j).  The (b, s, h) decomposition of a slot's problem
    // is done once by lane `slot` and fetched with two bpermutes per chunk, and the chunks go in batches of UB: all 2*UB
    // 16-byte loads of a batch are in flight before the first LDS store (the first version divided 64-bit indices and waited
    // for its two loads in every one of the ~7 trips: a serial load -> store chain, 194 us against 75 us of HBM time at 64x64)
    const int nchunks = p.G * p.Tk * 8;
    long long slot_base = -1;          // element offset of (b, t = 0, s, h) for slot `lane`
    if (lane < p.G && p0 + lane < p.P) {
        const long long pp = p0 + lane;
        const int h = (int)(pp % p.heads);
        const long long bs = pp / p.heads;
        const long long s_ = bs % p.S, b_ = bs / p.S;
        slot_base = b_ * p.kv_sb + s_ * p.kv_ss + h * 64;
    }
    constexpr int UB = 4;
    for (int c0 = 0; c0 < nchunks; c0 += 64 * UB) {
        uint4 uk[UB], uv[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int c = c0 + u * 64 + lane;
            const int row = c >> 3, kc = c & 7;
            const int slot_ = (int)((unsigned)row / (unsigned)p.Tk), j = row - slot_ * p.Tk;
            const int src_lane = slot_ < 64 ? slot_ : 0;
            const unsigned lo = (unsigned)__shfl((int)(unsigned)(slot_base & 0xffffffffll), src_lane, 64);
            const int hi = __shfl((int)(slot_base >> 32), src_lane, 64);
            const long long base = ((long long)hi << 32) | lo;
            uk[u] = make_uint4(0, 0, 0, 0);
            uv[u] = uk[u];
            if (c < nchunks && base >= 0) {
                const long long off = base + (long long)j * p.kv_st + kc * 8;
                uk[u] = *reinterpret_cast<const uint4*>(p.k + off);
                uv[u] = *reinterpret_cast<const uint4*>(p.v + off);
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int c = c0 + u * 64 + lane;
            if (c < nchunks) {
                *reinterpret_cast<uint4*>(sK + (c >> 3) * 64 + (c & 7) * 8) = uk[u];
                *reinterpret_cast<uint4*>(sV + (c >> 3) * 64 + (c & 7) * 8) = uv[u];
            }
        }
    }
    __syncthreads();

    const int slot = lane / p.Tq, i = lane - slot * p.Tq;
    const long long pp = p0 + slot;
    if (slot >= p.G || pp >= p.P) return;
    const int h = (int)(pp % p.heads);
    const long long bs = pp / p.heads;
    const long long s = bs % p.S, b = bs / p.S;

    // q stays packed (32 bf16 pairs); scores by v_dot2c_f32_bf16 (2 MACs per instruction, no unpacking), P rounded to bf16
    // pairs and P.V by dot2 over key pairs with the two V rows interleaved by v_perm_b32: ~1800 VALU per lane instead of
    // ~4700 (shift/and unpack + fma per element) - the kernel was VALU-bound at 2.5x its HBM time
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    uint32_t qp[32];
    {
        const bf16_t* qptr = p.q + b * p.q_sb + (long long)i * p.q_st + s * p.q_ss + h * 64;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint4 u = *reinterpret_cast<const uint4*>(qptr + c * 8);
            qp[c * 4 + 0] = u.x; qp[c * 4 + 1] = u.y; qp[c * 4 + 2] = u.z; qp[c * 4 + 3] = u.w;
        }
    }
    const bf16_t* kk = sK + slot * p.Tk * 64;
    const bf16_t* vv = sV + slot * p.Tk * 64;
    auto dot2 = [](uint32_t a_, uint32_t b_, float c_) __attribute__((always_inline)) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, a_), __builtin_bit_cast(bf16x2v, b_), c_, false);
    };
    float sc[TMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
        sc[j] = -INFINITY;
        if (j < p.Tk) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint4 u = *reinterpret_cast<const uint4*>(kk + j * 64 + c * 8);
                a0 = dot2(qp[c * 4 + 0], u.x, a0);
                a1 = dot2(qp[c * 4 + 1], u.y, a1);
                a0 = dot2(qp[c * 4 + 2], u.z, a0);
                a1 = dot2(qp[c * 4 + 3], u.w, a1);
            }
            sc[j] = (a0 + a1) * p.scale;
            mx = fmaxf(mx, sc[j]);
        }
    }
    float ov[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) ov[d] = 0.f;
    float l = 0.f;
    const float mxl = mx * 1.44269504088896340736f;
#pragma unroll
    for (int jp = 0; jp < TMAX / 2; ++jp) {
        if (2 * jp < p.Tk) {
            const bool two = 2 * jp + 1 < p.Tk;            // wave-uniform
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[2 * jp], 1.44269504088896340736f, -mxl));
            const float p1 = two ? __builtin_amdgcn_exp2f(__builtin_fmaf(sc[2 * jp + 1], 1.44269504088896340736f, -mxl)) : 0.f;
            const uint32_t pp = pack2bf(p0, p1);
            l += bflo(pp) + bfhi(pp);                      // normalise by exactly the (rounded) weights that are applied
            const bf16_t* v0 = vv + (2 * jp) * 64;
            const bf16_t* v1 = two ? v0 + 64 : v0;         // (weight 0 on the duplicate row)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint4 ua = *reinterpret_cast<const uint4*>(v0 + c * 8);
                const uint4 ub = *reinterpret_cast<const uint4*>(v1 + c * 8);
                const uint32_t wa[4] = {ua.x, ua.y, ua.z, ua.w}, wb[4] = {ub.x, ub.y, ub.z, ub.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t lo = __builtin_amdgcn_perm(wb[e], wa[e], 0x05040100u);   // (v[j][d], v[j+1][d]),  d = c*8 + 2e
                    const uint32_t hi = __builtin_amdgcn_perm(wb[e], wa[e], 0x07060302u);   // d + 1
                    ov[c * 8 + 2 * e] = dot2(pp, lo, ov[c * 8 + 2 * e]);
                    ov[c * 8 + 2 * e + 1] = dot2(pp, hi, ov[c * 8 + 2 * e + 1]);
                }
            }
        }
    }
    const float inv = 1.0f / l;
    bf16_t* op = p.out + b * p.o_sb + (long long)i * p.o_st + s * p.o_ss + h * 64;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint4 w = make_uint4(pack2bf(ov[c * 8 + 0] * inv, ov[c * 8 + 1] * inv), pack2bf(ov[c * 8 + 2] * inv, ov[c * 8 + 3] * inv),
                                   pack2bf(ov[c * 8 + 4] * inv, ov[c * 8 + 5] * inv), pack2bf(ov[c * 8 + 6] * inv, ov[c * 8 + 7] * inv));
        *reinterpret_cast<uint4*>(op + c * 8) = w;
    }
}

// ------------------------------------------------------------------------------------------------------
// Temporal self-attention on the matrix cores (round 5).  One wave per problem (sample b, position s, head h): T <= 32 frames, d_head 64.
// The VALU kernel above spends ~2400 issue cycles per problem on dot2 arithmetic (46 us of vector ALU beside 75 us of HBM time at the 64 x 64
// level, 2.8 TB/s measured); here the two contractions are 16 x 16 x 32 MFMAs on operands that need no repacking:
//   S^T = K Q^T    A = K rows, B = Q rows: a lane (row = lane & 15, k-group g = lane >> 4) loads 16 bytes of frame `row` at channel
//                  32 ks + 8 g - 64 contiguous bytes per frame per instruction, whole 128-byte lines over the two k steps, straight from HBM
//                  into the operand registers (the frame stride of the channels-last layout is the row stride: no transpose, no staging).
//                  Frames 16 .. 31 are the second row tile (T = 18: two live rows).  C: lane (query i = lane & 15, g) holds keys 4 g + r of
//                  key tile jt - the softmax of a query is 8 register values x 4 lanes (two shuffle steps for the max and for the sum).
//   O^T = V^T P^T  contraction slot 8 g + e  <->  key 16 (e >> 2) + 4 g + (e & 3): with THIS slot order the B operand of a lane is exactly the
//                  eight probabilities it already holds (rounded to bf16 pairs), no lane exchange; the A operand V^T[d][slot] is gathered from
//                  the wave's LDS copy of V (rows padded to 144 bytes: the four k-groups read four different 32-byte bank groups).
//                  C: lane (query i, g) holds channels 16 dt + 4 g + r: an 8-byte store per (query tile, channel tile).
// P is rounded to bf16 and the row is normalised by the sum of the ROUNDED weights, exactly as the VALU kernel (and the emulator) do.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_temporal_mfma_kernel(TP p) {
    constexpr int VP = 72;                                  // LDS row pitch of V in bf16 (144 bytes)
    __shared__ __attribute__((aligned(16))) bf16_t sVall[4][32 * VP];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long pp = (long long)blockIdx.x * 4 + wave;
    if (pp >= p.P) return;                                  // (wave-uniform; the LDS region is wave-private: no block barrier below)
    bf16_t* sv = sVall[wave];
    const int h = (int)(pp % p.heads);
    const long long bs = pp / p.heads;
    const long long s = bs % p.S, b = bs / p.S;
    const int m = lane & 15, g = lane >> 4;
    const int Tq = p.Tq, Tk = p.Tk;
    const bf16_t* qb = p.q + b * p.q_sb + s * p.q_ss + h * 64;
    const bf16_t* kb = p.k + b * p.kv_sb + s * p.kv_ss + h * 64;
    const bf16_t* vb = p.v + b * p.kv_sb + s * p.kv_ss + h * 64;
    const uint4 z4 = make_uint4(0, 0, 0, 0);

    // ---- all global loads of the problem up front: V chunks (-> LDS), K and Q operand fragments
    uint4 vch[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = u * 64 + lane;                        // chunk c: frame c >> 3, channels 8 (c & 7) ..
        const int cc = c < Tk * 8 ? c : Tk * 8 - 1;         // (clamped, not predicated: a select between a load and zero became a select of POINTERS - flat loads from a zeroed scratch slot)
        vch[u] = *reinterpret_cast<const uint4*>(vb + (long long)(cc >> 3) * p.kv_st + (cc & 7) * 8);
    }
    uint4 kf[2][2], qf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = 16 * t + m;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // rows past T are loaded from the last frame and zeroed in registers (keys past Tk are masked to -inf below; queries past Tq are never stored)
            const uint4 ku = *reinterpret_cast<const uint4*>(kb + (long long)(row < Tk ? row : Tk - 1) * p.kv_st + ks * 32 + g * 8);
            const uint4 qu = *reinterpret_cast<const uint4*>(qb + (long long)(row < Tq ? row : Tq - 1) * p.q_st + ks * 32 + g * 8);
            kf[t][ks] = row < Tk ? ku : z4;
            qf[t][ks] = row < Tq ? qu : z4;
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = u * 64 + lane;
        if (c < Tk * 8) *reinterpret_cast<uint4*>(sv + (c >> 3) * VP + (c & 7) * 8) = vch[u];
    }

    // ---- S^T = K Q^T
    f32x4 sacc[2][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            sacc[jt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (jt * 16 < Tk && it * 16 < Tq) {             // (wave-uniform)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    sacc[jt][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[jt][ks]), __builtin_bit_cast(bf16x8, qf[it][ks]),
                                                                           sacc[jt][it], 0, 0, 0);
            }
        }

    // ---- softmax over the keys of each query (lane & 15, query tile it): 8 values here, the other keys in the lanes +-16, +-32
    const float sl2 = p.scale * 1.44269504088896340736f;
    uint4 pf[2];
    float inv[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        float v[8];
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int key = 16 * (e >> 2) + 4 * g + (e & 3);
            v[e] = key < Tk ? sacc[e >> 2][it][e & 3] * sl2 : -INFINITY;
            mx = fmaxf(mx, v[e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        uint32_t w[4];
        float l = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float p0 = __builtin_amdgcn_exp2f(v[e] - mx), p1 = __builtin_amdgcn_exp2f(v[e + 1] - mx);
            w[e >> 1] = pack2bf(p0, p1);
            l += bflo(w[e >> 1]) + bfhi(w[e >> 1]);         // normalise by exactly the (rounded) weights that are applied
        }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        inv[it] = 1.0f / l;
        pf[it] = make_uint4(w[0], w[1], w[2], w[3]);
    }

    // ---- O^T = V^T P^T: the wave's V rows are in LDS (written by other lanes of this wave: order the reads behind the writes)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    f32x4 oacc[2][4];
    const unsigned short* svu = reinterpret_cast<const unsigned short*>(sv);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        uint32_t a[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const int k0 = 16 * (e >> 2) + 4 * g + (e & 3);
            const uint32_t lo = k0 < Tk ? (uint32_t)svu[k0 * VP + 16 * dt + m] : 0u;
            const uint32_t hi = k0 + 1 < Tk ? (uint32_t)svu[(k0 + 1) * VP + 16 * dt + m] : 0u;
            a[e >> 1] = lo | (hi << 16);
        }
        const bf16x8 vf = __builtin_bit_cast(bf16x8, make_uint4(a[0], a[1], a[2], a[3]));
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            oacc[it][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (it * 16 < Tq) oacc[it][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, __builtin_bit_cast(bf16x8, pf[it]), oacc[it][dt], 0, 0, 0);
        }
    }

    // ---- out[query i][channels 16 dt + 4 g ..]: 8 bytes per (it, dt)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = 16 * it + m;
        if (i < Tq) {
            bf16_t* op = p.out + b * p.o_sb + (long long)i * p.o_st + s * p.o_ss + h * 64 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<uint2*>(op + 16 * dt) = make_uint2(pack2bf(oacc[it][dt][0] * inv[it], oacc[it][dt][1] * inv[it]),
                                                                     pack2bf(oacc[it][dt][2] * inv[it], oacc[it][dt][3] * inv[it]));
        }
    }
}

}  // namespace

extern "C" int v3d_attn_spatial(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vT, void* out,
                                int64_t ldo, int64_t n_img, int64_t S, int32_t heads, float scale, v3d_stream_t stream) {
    V3D_REQUIRE(q && k && vT && out, "v3d_attn_spatial: null pointer");
    V3D_REQUIRE(n_img > 0 && n_img <= 65535 && heads > 0 && heads <= 65535 && S > 0, "v3d_attn_spatial: bad sizes");
    V3D_REQUIRE(S % 8 == 0, "v3d_attn_spatial: S must be a multiple of 8 (got %lld)", (long long)S);
    V3D_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 4 == 0, "v3d_attn_spatial: ldq/ldk must be multiples of 8, ldo of 4");
    V3D_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vT) & 15) == 0 && ((uintptr_t)out & 7) == 0, "v3d_attn_spatial: misaligned pointer");
    V3D_REQUIRE((unsigned long long)(S + 192) * (ldq > ldk ? ldq : ldk) * 2ull <= kMaxBufBytes && (unsigned long long)(64 * S + 256) * 2ull <= kMaxBufBytes,
                "v3d_attn_spatial: per-image q / k / v slab exceeds 4 GiB");
    static int impl = -1;
    if (impl < 0) {
        const char* e = getenv("V3D_ATTN_IMPL");
        impl = e ? atoi(e) : 2;
    }
    const float sc2 = scale * 1.44269504088896340736f;
    if (impl == 1) {
        dim3 grid((unsigned)((S + 127) / 128), (unsigned)heads, (unsigned)n_img);
        hipLaunchKernelGGL(attn_spatial_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, (long long)ldq,
                           (const bf16_t*)k, (long long)ldk, (const bf16_t*)vT, (bf16_t*)out, (long long)ldo, (long long)S, heads, sc2);
    } else if (S >= 1024 && impl != 3) {
        dim3 grid((unsigned)((S + 255) / 256), (unsigned)heads, (unsigned)n_img);
        hipLaunchKernelGGL(attn_spatial_v2_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, (long long)ldq,
                           (const bf16_t*)k, (long long)ldk, (const bf16_t*)vT, (bf16_t*)out, (long long)ldo, (long long)S, heads, sc2);
    } else {
        dim3 grid((unsigned)((S + 127) / 128), (unsigned)heads, (unsigned)n_img);
        hipLaunchKernelGGL(attn_spatial_v2_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, (long long)ldq,
                           (const bf16_t*)k, (long long)ldk, (const bf16_t*)vT, (bf16_t*)out, (long long)ldo, (long long)S, heads, sc2);
    }
    return v3d_check_launch("v3d_attn_spatial");
}

extern "C" int v3d_attn_temporal(const void* q, int64_t q_sb, int64_t q_st, int64_t q_ss,
                                 const void* k, const void* v, int64_t kv_sb, int64_t kv_st, int64_t kv_ss,
                                 void* out, int64_t o_sb, int64_t o_st, int64_t o_ss,
                                 int64_t B, int32_t Tq, int32_t Tk, int64_t S, int32_t heads, float scale,
                                 v3d_stream_t stream) {
    V3D_REQUIRE(q && k && v && out, "v3d_attn_temporal: null pointer");
    V3D_REQUIRE(Tq >= 1 && Tq <= TMAX && Tk >= 1 && Tk <= TMAX, "v3d_attn_temporal: Tq/Tk must be in [1,%d] (got %d,%d)", TMAX, Tq, Tk);
    V3D_REQUIRE(B > 0 && S > 0 && heads > 0, "v3d_attn_temporal: bad sizes");
    V3D_REQUIRE((q_sb | q_st | q_ss | kv_sb | kv_st | kv_ss | o_sb | o_st | o_ss) % 8 == 0, "v3d_attn_temporal: strides must be multiples of 8 elements");
    V3D_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0, "v3d_attn_temporal: misaligned pointer");
    TP p;
    p.q = (const bf16_t*)q; p.q_sb = q_sb; p.q_st = q_st; p.q_ss = q_ss;
    p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.kv_sb = kv_sb; p.kv_st = kv_st; p.kv_ss = kv_ss;
    p.out = (bf16_t*)out; p.o_sb = o_sb; p.o_st = o_st; p.o_ss = o_ss;
    p.P = (long long)B * S * heads; p.S = S; p.heads = heads; p.Tq = Tq; p.Tk = Tk; p.scale = scale;
    static int timpl = -1;
    if (timpl < 0) {
        const char* e = getenv("V3D_ATTN_TEMPORAL_IMPL");   // A/B knob: 1 = the VALU (dot2) kernel of rounds 1-4, 2 = the MFMA kernel (default)
        timpl = e ? atoi(e) : 2;
    }
    if (timpl != 1) {
        p.G = 1;
        const long long blocks = (p.P + 3) / 4;             // one wave per problem
        V3D_REQUIRE(blocks < (1ll << 31), "v3d_attn_temporal: grid too large");
        hipLaunchKernelGGL(attn_temporal_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
        return v3d_check_launch("v3d_attn_temporal");
    }
    {   // problems per wave: as many Tq-lane slots as fit in a wave, capped so K+V staging stays <= 16 KiB per wave
        int g = 64 / Tq;
        const int cap = 16384 / (Tk * 256);
        if (g > cap) g = cap;
        if (g < 1) g = 1;
        p.G = g;
    }
    const long long nwaves = (p.P + p.G - 1) / p.G;
    const long long blocks = (nwaves + 3) / 4;
    V3D_REQUIRE(blocks < (1ll << 31), "v3d_attn_temporal: grid too large");
    const size_t shmem = (size_t)4 * 2 * p.G * Tk * 64 * sizeof(bf16_t);
    V3D_REQUIRE(shmem <= 64 * 1024, "v3d_attn_temporal: LDS request %zu too large", shmem);
    hipLaunchKernelGGL(attn_temporal_kernel, dim3((unsigned)blocks), dim3(256), shmem, (hipStream_t)stream, p);
    return v3d_check_launch("v3d_attn_temporal");
}
