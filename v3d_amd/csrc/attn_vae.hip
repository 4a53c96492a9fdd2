// v3d_attn_vae_d512: single-head self-attention of the VAE AttnBlock (head dim = C = 512, 4096 .. 9216 tokens) as a streamed-softmax
// MFMA kernel for gfx950 - the [S, S] score matrix never exists in memory (reference: sgm/modules/diffusionmodules/model.py:180-201,
// F.scaled_dot_product_attention over one 512-wide head).
//
// Block = 4 waves, ONE PER SIMD (the kernel owns the whole 512-register file of each SIMD, like ff_fused_kernel): a wave keeps
// 32 queries x C channels of Q (C/4 VGPRs, MFMA B operand) and the fp32 output tile O^T [C x 32] (C/2 VGPRs) in registers and
// walks the keys in tiles of 32: S^T = K . Q^T  (C/16 x v_mfma_f32_32x32x16_bf16, a lane owns one query column -> the softmax row
// reductions are 15 in-lane ops + one shfl_xor 32), P^T straight from registers as the B operand of  O^T += V^T . P^T
// (C/16 MFMAs) - the same operand arrangement as attn_spatial_v2_kernel (attn.hip): the K rows are assigned to MFMA rows through
// pi(8A + 4h + c) = 16(A>>1) + 8h + 4(A&1) + c so that the 8 P values a lane feeds into one P.V k-step are 8 CONTIGUOUS keys and
// the V^T fragments (V arrives pre-transposed [C][S] from a swapped projection GEMM) are single ds_read_b128.
// K / V^T tiles (C x 64 B each) arrive by buffer-load LDS-DMA into a 2-deep ring (2 x 64 KiB at C = 512), one barrier per tile;
// the 16-byte chunk position is XOR-swizzled on the DMA source address: K rows (C * 2 bytes, a multiple of the 256-B bank row)
// by (row & 15), V^T rows (64 B) by {0,2,3,1}[(row >> 2) & 3] - every ds_read_b128 lane group hits 16 distinct slots.
// TWO passes over the keys instead of a running max: the O^T accumulator fills the accumulator half of the register file and Q
// half of the other, so the "O^T *= alpha" of an online softmax (VALU cannot touch accumulator registers: copy out, multiply, copy
// back - the compiler materialises a second copy of O^T and spills 200+ registers) has no room.  Pass 1 streams K only and takes
// the exact row maxima (C/16 MFMAs per tile), pass 2 recomputes the scores, P = exp2((s - max) scale) <= 1 and accumulates: 1.5x the
// MFMA work of a single pass, no data-dependent branch, no rescale error - for 0.6 TFLOP per 18-frame decode (~1 ms) that is the
// better trade.  Softmax runs in the exp2 domain on raw scores (one v_fma + one bare v_exp_f32 per score).
// A block covers 128 queries; K / V^T are streamed once per block: 4 S^2 C flop against (S / 128) * 4 S C bytes of L2 -> LDS fill.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int DT>   // DT = C / 32 output channel tiles: 16 (C = 512), 8, 4
__global__ __launch_bounds__(256, 1) void attn_vae_kernel(const bf16_t* __restrict__ q, long long ldq, const bf16_t* __restrict__ k, long long ldk,
                                                          const bf16_t* __restrict__ vT, const float* __restrict__ bias,
                                                          bf16_t* __restrict__ out, long long ldo, long long S, float scale2) {
    constexpr int C = DT * 32;
    constexpr int KSTEPS = C / 16;               // MFMA k-steps of the S^T contraction
    constexpr int KROWB = C * 2;                 // bytes of one K row
    constexpr int CPR = C / 8;                   // 16-byte chunks per K row (16, 32 or 64)
    constexpr int KTILE = 32 * KROWB;            // 32 keys
    constexpr int VTILE = C * 64;                // C rows of 32 keys
    constexpr int STAGE = KTILE + VTILE;
    constexpr int KPIECES = KTILE / 1024, VPIECES = VTILE / 1024;   // C / 16 each
    constexpr int KPW = KPIECES / 4, VPW = VPIECES / 4;             // pieces per wave per tile
    static_assert(KPW >= 1 && VPW >= 1, "C >= 128");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31, hi = lane >> 5;
    const long long n = blockIdx.y;
    const long long qrow = (long long)blockIdx.x * 128 + wave * 32 + qi;

    // ---- Q fragments (B operand of S^T = K . Q^T): lane (qi, hi) holds channels 16 st + 8 hi .. + 7 of query qi
    const bufrsrc_t rsQ = make_rsrc(q + n * S * ldq, (unsigned)(((S - 1) * ldq + C) * 2));
    bf16x8 qf[KSTEPS];
#pragma unroll
    for (int st = 0; st < KSTEPS; ++st)
        qf[st] = __builtin_bit_cast(bf16x8, buf_load16(rsQ, qrow < S ? (unsigned)((qrow * ldq + hi * 8 + st * 16) * 2) : kInvalid));

    // ---- LDS-DMA sources.  K piece = 1 KiB = 1024 / KROWB key rows: lane l -> row l / CPR of the piece, chunk position l % CPR, which
    //      holds logical chunk pos ^ (row & 15).  V^T piece = 16 channel rows x 64 B: lane l -> row l >> 2, position l & 3, logical chunk
    //      pos ^ swz(row).  Keys past S fall outside the K descriptor and arrive as zeros (masked to -inf below); V^T columns past S read
    //      the next row's start (finite) or zeros and meet P = 0.
    const bufrsrc_t rsK = make_rsrc(k + n * S * ldk, (unsigned)(((S - 1) * ldk + C) * 2));
    const bufrsrc_t rsV = make_rsrc(vT + n * (long long)C * S, (unsigned)((long long)C * S * 2));
    unsigned koffs[KPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int piece = wave * KPW + i;
        const int row = piece * (1024 / KROWB) + lane / CPR;      // key row within the tile
        const int pos = lane % CPR;
        koffs[i] = (unsigned)((row * ldk + ((pos ^ (row & 15)) * 8)) * 2);
    }
    const int vrow_in_piece = lane >> 2;
    const int vsw_src = (0x78 >> (((vrow_in_piece >> 2) & 3) * 2)) & 3;
    unsigned voffs = (unsigned)((((long long)(wave * VPW * 16 + vrow_in_piece)) * S + (((lane & 3) ^ vsw_src) * 8)) * 2);
    const unsigned kstep = (unsigned)(32 * ldk * 2);
    const int ntiles = (int)((S + 31) / 32);
    auto issue = [&](int stage, bool with_v) __attribute__((always_inline)) {
        unsigned char* sb = lds + stage * STAGE;
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (__attribute__((address_space(3))) void*)(sb + (wave * KPW + i) * 1024), 16, (int)koffs[i], 0, 0, 0);
            koffs[i] += kstep;
        }
        if (with_v) {
#pragma unroll
            for (int i = 0; i < VPW; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (__attribute__((address_space(3))) void*)(sb + KTILE + (wave * VPW + i) * 1024), 16, (int)voffs,
                                                         (int)((long long)i * 16 * S * 2), 0, 0);
            voffs += 64u;
        }
    };

    // ---- fragment read offsets.  K: MFMA row i = qi holds key pi(qi); logical chunk 2 st + hi = 16 (st >> 3) + (2 (st & 7) + hi): the
    //      swizzle touches the low 4 bits only, so 8 per-lane bases + an immediate 256 (st >> 3) cover every k-step.
    const int A_ = qi >> 3, hh = (qi >> 2) & 1, cc = qi & 3;
    const int kkey = 16 * (A_ >> 1) + 8 * hh + 4 * (A_ & 1) + cc;                 // pi(qi)
    int koff8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) koff8[j] = kkey * KROWB + (((2 * j + hi) ^ (kkey & 15)) * 16);
    // V^T: row d = dt * 32 + qi (64-B rows), logical chunk 2 ks + hi = the 8 contiguous keys 16 ks + 8 hi ..
    const int vsw = (0x78 >> (((qi >> 2) & 3) * 2)) & 3;
    int voff2[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) voff2[ks] = KTILE + qi * 64 + (((2 * ks + hi) ^ vsw) * 16);

    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_row = -INFINITY, l_run = 0.f;

    // S^T = K . Q^T for the 32 keys of the tile in ring slot `sb`:  sT[r] = score(key k0 + 16 (r >> 3) + 8 hi + (r & 7), query qi)
    auto scores = [&](const unsigned char* sb, long long k0, f32x16& sT) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sT[r] = 0.f;
        // (groups of 4 k-steps fenced off from each other: left alone the scheduler hoists all C/16 fragment reads of a tile to the top and
        //  spills - Q and O^T already take 3/4 of the register file)
        static_for<0, KSTEPS / 4>([&](auto g_) {
            constexpr int g4 = decltype(g_)::value;
            bf16x8 kf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) kf[j] = *reinterpret_cast<const bf16x8*>(sb + koff8[(g4 * 4 + j) & 7] + 256 * ((g4 * 4 + j) >> 3));
            static_for<0, 4>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[j], qf[g4 * 4 + j], sT, 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        if (k0 + 32 > S) {   // wave-uniform, last tile only: keys past the end of the sequence
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((k0 + 16 * (r >> 3) + 8 * hi + (r & 7)) >= S) sT[r] = -INFINITY;
        }
    };

    // ================= pass 1: the row maxima (K stream only) =================
    issue(0, false);
    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of tile t landed
        __builtin_amdgcn_s_barrier();                      // everyone's pieces landed; everyone finished reading tile t - 1
        asm volatile("" ::: "memory");
        if (t + 1 < ntiles) issue((t + 1) & 1, false);
        f32x16 sT;
        scores(lds + (t & 1) * STAGE, (long long)t * 32, sT);
#pragma unroll
        for (int r = 0; r < 16; ++r) m_row = fmaxf(m_row, sT[r]);
    }
    m_row = fmaxf(m_row, __shfl_xor(m_row, 32, 64));
    const float mb = -m_row * scale2;                      // every tile holds at least one valid key: finite
    __builtin_amdgcn_s_barrier();                          // the last tile's slot is free again
#pragma unroll
    for (int i = 0; i < KPW; ++i) koffs[i] -= (unsigned)ntiles * kstep;

    // ================= pass 2: P = exp2((s - max) scale) <= 1 and O^T += V^T . P^T, no rescaling =================
    issue(0, true);
    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 1 < ntiles) issue((t + 1) & 1, true);
        const unsigned char* sb = lds + (t & 1) * STAGE;
        f32x16 sT;
        scores(sb, (long long)t * 32, sT);
        float psum = 0.f;
        bf16x8 pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pv[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(sT[ks * 8 + j], scale2, mb));
                psum += pv[j];
            }
            const u32x4 u = {pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3]), pack2bf(pv[4], pv[5]), pack2bf(pv[6], pv[7])};
            pf[ks] = __builtin_bit_cast(bf16x8, u);
        }
        l_run += psum;
        // ---- O^T += V^T . P^T ----
        static_for<0, DT / 2>([&](auto g_) {
            constexpr int g2 = decltype(g_)::value;
            bf16x8 vf[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) vf[j][ks] = *reinterpret_cast<const bf16x8*>(sb + voff2[ks] + (g2 * 2 + j) * 2048);
            static_for<0, 2>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                o[g2 * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[j][0], pf[0], o[g2 * 2 + j], 0, 0, 0);
                o[g2 * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[j][1], pf[1], o[g2 * 2 + j], 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qrow < S) {
        bf16_t* op = out + (n * S + qrow) * ldo;
        auto store = [&](auto has_bias_) __attribute__((always_inline)) {
            constexpr bool HAS_BIAS = decltype(has_bias_)::value;
            static_for<0, DT>([&](auto dt_) {
                constexpr int dt = decltype(dt_)::value;
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const int d0 = dt * 32 + 8 * gg + 4 * hi;
                    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (HAS_BIAS) bb = *reinterpret_cast<const float4*>(bias + d0);
                    *reinterpret_cast<uint2*>(op + d0) = make_uint2(pack2bf(o[dt][gg * 4 + 0] * inv + bb.x, o[dt][gg * 4 + 1] * inv + bb.y),
                                                                    pack2bf(o[dt][gg * 4 + 2] * inv + bb.z, o[dt][gg * 4 + 3] * inv + bb.w));
                }
            });
        };
        if (bias) store(std::true_type{});
        else store(std::false_type{});
    }
}

}  // namespace

extern "C" int v3d_attn_vae_d512(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vT, const float* bias, void* out,
                                 int64_t ldo, int64_t n_img, int64_t S, int32_t C, float scale, v3d_stream_t stream) {
    V3D_REQUIRE(q && k && vT && out, "v3d_attn_vae_d512: null pointer");
    V3D_REQUIRE(C == 512 || C == 256 || C == 128, "v3d_attn_vae_d512: C must be 512 (the V3D / SVD first stage), 256 or 128 (got %d)", C);
    V3D_REQUIRE(n_img > 0 && n_img <= 65535 && S > 0 && S % 8 == 0, "v3d_attn_vae_d512: bad n_img / S (S must be a multiple of 8)");
    V3D_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 4 == 0 && ldq >= C && ldk >= C && ldo >= C, "v3d_attn_vae_d512: bad row strides");
    V3D_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vT) & 15) == 0 && ((uintptr_t)out & 7) == 0 && ((uintptr_t)bias & 15) == 0,
                "v3d_attn_vae_d512: misaligned pointer");
    V3D_REQUIRE((unsigned long long)(S + 64) * (ldq > ldk ? ldq : ldk) * 2ull <= kMaxBufBytes && (unsigned long long)C * (S + 64) * 2ull <= kMaxBufBytes,
                "v3d_attn_vae_d512: per-image q / k / v slab exceeds 4 GiB");
    const float sc2 = scale * 1.44269504088896340736f;
    const dim3 grid((unsigned)((S + 127) / 128), (unsigned)n_img);
#define V3D_AV_LAUNCH(DT_)                                                                                                                   \
    hipLaunchKernelGGL((attn_vae_kernel<DT_>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, (long long)ldq, (const bf16_t*)k, \
                       (long long)ldk, (const bf16_t*)vT, bias, (bf16_t*)out, (long long)ldo, (long long)S, sc2)
    if (C == 512) V3D_AV_LAUNCH(16);
    else if (C == 256) V3D_AV_LAUNCH(8);
    else V3D_AV_LAUNCH(4);
#undef V3D_AV_LAUNCH
    return v3d_check_launch("v3d_attn_vae_d512");
}
