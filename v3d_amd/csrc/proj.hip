// v3d_ln_proj: LayerNorm + the q | k | v projection of a transformer block in one kernel (C = 320, the 64x64 level), gfx950.
//
// Reference: BasicTransformerBlock / VideoTransformerBlock  x -> norm1(x) -> attn1.to_q / to_k / to_v  (sgm/modules/attention.py:556-563,
// 286-290; video_attention.py:122-125).  Unfused that is a LayerNorm kernel (read + write of the 94 MB token tensor), one GEMM for q | k
// and a swapped batched GEMM for V^T, every one of them HBM-bound at this level (K = N = 320: 107 flop / byte): 38 + 114 + 78 us.
// Here a block of 128 token rows keeps its rows in registers as MFMA operand fragments (the ff_fused_kernel arrangement: 4 waves, one
// per SIMD, 32 rows each, 80 VGPRs of bf16 fragments), normalises them there (fp32 statistics, two passes like layernorm_kernel; the
// normalised rows are rounded to bf16 exactly where the unfused path stores them), and streams the concatenated weight matrix through a
// 3-deep LDS ring in slabs of 64 output channels (40 KiB, LDS-DMA pieces of contiguous 1-KiB: the weights are stored in DMA-piece
// order with the LDS swizzle pre-applied, packing.py ff_dma_tile_index).  Per slab and wave: 40 x v_mfma_f32_32x32x16_bf16.
//   * slabs [0, n_rm / 64): out[m][n] row-major - the weight fragment is the A operand, a lane ends up with 4 consecutive channels
//     of one token;
//   * the remaining slabs: outT[image][n - n_rm][token] (keys contiguous: the V^T layout the attention kernel reads) - the SAME
//     fragments with the MFMA operands swapped, so a lane ends up with 4 consecutive tokens of one channel: no transpose anywhere.
// The token tensor is read once and no normalised copy is ever written.  The weight stream is block-independent and runs across
// row-block boundaries; the next block's rows are fetched while the current block computes.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

struct PP {
    const bf16_t* x;
    const float* gamma;
    const float* beta;
    const bf16_t* W;
    bf16_t* out;
    bf16_t* outT;
    long long M, ldx, ldo, S;
    int N, n_rm, Ct;
    unsigned w_bytes;
    float eps;
};

template <int I, int N, typename F>
__device__ __forceinline__ void pj_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pj_static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ int pj_swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // 64-byte LDS rows, see gemm.hip

template <int C>
__global__ __launch_bounds__(256, 1) void ln_proj_kernel(PP p) {
    constexpr int NK = C / 16;                  // k16 steps = resident row fragments
    constexpr int NT = C / 32;                  // 32-k LDS stages of a slab
    constexpr int SLAB = 64 * C * 2;            // 64 weight rows
    constexpr int PPW = NT;                     // 1-KiB pieces per wave per slab (NT * 4 pieces / 4 waves)
    constexpr int NSLOT = 3;
    constexpr int GB_OFF = NSLOT * SLAB;        // gamma | beta copies
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSLOT * SLAB + 2 * C * 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const long long nblocks = p.M / 128;
    const int nslab = p.N / 64;
    const int nrm_slabs = p.n_rm / 64;
    float* gsm = reinterpret_cast<float*>(lds + GB_OFF);

    for (int c = tid; c < C; c += 256) {
        gsm[c] = p.gamma[c];
        gsm[C + c] = p.beta[c];
    }

    // ---- weight stream: flat over (row block, slab); stream index j -> slab j % nslab, ring slot j % 3
    const bufrsrc_t rsW = make_rsrc(p.W, p.w_bytes);
    const unsigned voff = (unsigned)(wave * 1024 + lane * 16);
    const long long my_blocks = (nblocks - (long long)blockIdx.x + gridDim.x - 1) / gridDim.x;
    const long long total = my_blocks * nslab;
    int ld_slab = 0, ld_slot = 0;
    auto issue_slab = [&]() __attribute__((always_inline)) {
        unsigned char* dst = lds + ld_slot * SLAB;
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + (wave + 4 * i) * 1024), 16, (int)voff,
                                                     ld_slab * SLAB + i * 4096, 0, 0);
        ld_slab = (ld_slab + 1 == nslab) ? 0 : ld_slab + 1;
        ld_slot = (ld_slot + 1 == NSLOT) ? 0 : ld_slot + 1;
    };

    // fragment of 32 rows x 16 k out of 64-byte rows: row l31, logical 16-byte chunk 2 (ks & 1) + hi
    const int foff0 = l31 * 64 + (((0 + hi) ^ pj_swz(l31)) * 16);
    const int foff1 = l31 * 64 + (((2 + hi) ^ pj_swz(l31)) * 16);

    bf16x8 xr[NK], xn[NK];
    auto load_rows = [&](long long blk, bf16x8 (&dst)[NK]) __attribute__((always_inline)) {
        const bf16_t* xz = p.x + (blk * 128 + wave * 32 + l31) * p.ldx + hi * 8;
#pragma unroll
        for (int k = 0; k < NK; ++k) dst[k] = *reinterpret_cast<const bf16x8*>(xz + k * 16);
    };
    // LayerNorm of the wave's 32 rows in place: lane (l31, hi) holds channels 16 k + 8 hi .. + 7 of row l31, its partner lane the rest
    auto normalise = [&](bf16x8 (&v)[NK]) __attribute__((always_inline)) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const u32x4 u = __builtin_bit_cast(u32x4, v[k]);
#pragma unroll
            for (int e = 0; e < 4; ++e) s += bflo(u[e]) + bfhi(u[e]);
        }
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const u32x4 u = __builtin_bit_cast(u32x4, v[k]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = bflo(u[e]) - mean, b = bfhi(u[e]) - mean;
                q += a * a + b * b;
            }
        }
        q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q * (1.0f / C) + p.eps);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const u32x4 u = __builtin_bit_cast(u32x4, v[k]);
            const float4 g0 = *reinterpret_cast<const float4*>(gsm + k * 16 + hi * 8), g1 = *reinterpret_cast<const float4*>(gsm + k * 16 + hi * 8 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(gsm + C + k * 16 + hi * 8), b1 = *reinterpret_cast<const float4*>(gsm + C + k * 16 + hi * 8 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = pack2bf((bflo(u[e]) - mean) * rstd * gg[2 * e] + bb[2 * e], (bfhi(u[e]) - mean) * rstd * gg[2 * e + 1] + bb[2 * e + 1]);
            v[k] = __builtin_bit_cast(bf16x8, o);
        }
    };

    issue_slab();
    issue_slab();
    load_rows(blockIdx.x, xn);
    __syncthreads();                       // gamma / beta visible (drains the two slabs in flight once, at kernel start)

    long long j = 0;                       // weight-stream index of the slab about to be consumed
    int rd_slot = 0;
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
#pragma unroll
        for (int k = 0; k < NK; ++k) xr[k] = xn[k];
        normalise(xr);
        const long long row0 = blk * 128 + wave * 32;
        const long long img = row0 / p.S, pix0 = row0 - img * p.S;
        for (int sl = 0; sl < nslab; ++sl, ++j) {
            // this wave's pieces of slab j have landed when only the ops issued after them are outstanding: the next slab's 10 pieces,
            // the stores of the last two slabs (8 each) and - on the first two slabs of a block - the 20 row loads of the next block
            if (j < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (sl == 1 || sl == 2) asm volatile("s_waitcnt vmcnt(46)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // everyone's pieces landed; everyone finished reading the slot refilled below
            asm volatile("" ::: "memory");
            if (j + 2 < total) issue_slab();
            else {                                   // stream tail: keep the per-slab op count constant for the counted waits
#pragma unroll
                for (int i = 0; i < PPW; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(lds + ld_slot * SLAB + (wave + 4 * i) * 1024), 16,
                                                             (int)kInvalid, 0, 0, 0);
                ld_slot = (ld_slot + 1 == NSLOT) ? 0 : ld_slot + 1;
            }
            if (sl == 0) {
                const long long nxt = blk + gridDim.x;
                load_rows(nxt < nblocks ? nxt : blk, xn);
            }
            const unsigned char* sb = lds + rd_slot * SLAB;
            rd_slot = (rd_slot + 1 == NSLOT) ? 0 : rd_slot + 1;
            f32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            const bool transposed = sl >= nrm_slabs;   // wave-uniform
            if (!transposed) {
                // (groups of 4 k-steps fenced off from each other: left alone the scheduler hoists all 40 fragment reads of a slab to the top,
                //  spills, and the scratch reloads' vmcnt(0) waits drain the LDS-DMA stream)
                pj_static_for<0, NK / 4>([&](auto g_) {
                    constexpr int g4 = decltype(g_)::value;
                    bf16x8 wf[4][2];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            wf[q][t] = *reinterpret_cast<const bf16x8*>(sb + ((g4 * 4 + q) >> 1) * 4096 + t * 2048 + (((g4 * 4 + q) & 1) ? foff1 : foff0));
                    pj_static_for<0, 4>([&](auto q_) {
                        constexpr int q = decltype(q_)::value;
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[q][0], xr[g4 * 4 + q], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[q][1], xr[g4 * 4 + q], acc[1], 0, 0, 0);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
                // acc[t][4 g + c] = out[token row0 + l31][channel 64 sl + 32 t + 8 g + 4 hi + c]
                bf16_t* op = p.out + (row0 + l31) * p.ldo + sl * 64 + 4 * hi;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<uint2*>(op + t * 32 + 8 * g) = make_uint2(pack2bf(acc[t][4 * g + 0], acc[t][4 * g + 1]), pack2bf(acc[t][4 * g + 2], acc[t][4 * g + 3]));
            } else {
                pj_static_for<0, NK / 4>([&](auto g_) {
                    constexpr int g4 = decltype(g_)::value;
                    bf16x8 wf[4][2];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            wf[q][t] = *reinterpret_cast<const bf16x8*>(sb + ((g4 * 4 + q) >> 1) * 4096 + t * 2048 + (((g4 * 4 + q) & 1) ? foff1 : foff0));
                    pj_static_for<0, 4>([&](auto q_) {
                        constexpr int q = decltype(q_)::value;
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xr[g4 * 4 + q], wf[q][0], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xr[g4 * 4 + q], wf[q][1], acc[1], 0, 0, 0);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
                // acc[t][4 g + c] = outT[image][channel 64 (sl - nrm_slabs) + 32 t + l31][token pix0 + 8 g + 4 hi + c]
                bf16_t* op = p.outT + (img * p.Ct + (sl - nrm_slabs) * 64 + l31) * p.S + pix0 + 4 * hi;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<uint2*>(op + (long long)t * 32 * p.S + 8 * g) = make_uint2(pack2bf(acc[t][4 * g + 0], acc[t][4 * g + 1]), pack2bf(acc[t][4 * g + 2], acc[t][4 * g + 3]));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

extern "C" int v3d_ln_proj(const void* x, int64_t ldx, const float* gamma, const float* beta, float eps, const void* Wp, void* out, int64_t ldo,
                           void* outT, int64_t M, int32_t C, int32_t N, int32_t n_rm, int64_t S, v3d_stream_t stream) {
    V3D_REQUIRE(x && gamma && beta && Wp, "v3d_ln_proj: null pointer");
    V3D_REQUIRE(C == 320, "v3d_ln_proj: C must be 320 (the 64x64 level; other widths use v3d_layernorm + v3d_gemm), got %d", C);
    V3D_REQUIRE(M > 0 && M % 128 == 0, "v3d_ln_proj: M must be a positive multiple of 128 (got %lld)", (long long)M);
    V3D_REQUIRE(N >= 128 && N % 64 == 0 && n_rm >= 0 && n_rm <= N && n_rm % 64 == 0, "v3d_ln_proj: N / n_rm must be multiples of 64 with n_rm <= N, N >= 128");
    V3D_REQUIRE(n_rm == 0 || (out && ldo % 4 == 0 && ldo >= n_rm), "v3d_ln_proj: bad out / ldo");
    V3D_REQUIRE(n_rm == N || (outT && S > 0 && S % 128 == 0 && M % S == 0), "v3d_ln_proj: the transposed part needs outT and S %% 128 == 0, M %% S == 0");
    V3D_REQUIRE(ldx % 8 == 0 && ldx >= C, "v3d_ln_proj: bad ldx");
    V3D_REQUIRE((((uintptr_t)x | (uintptr_t)Wp) & 15) == 0 && (((uintptr_t)out | (uintptr_t)outT) & 7) == 0 && (((uintptr_t)gamma | (uintptr_t)beta) & 3) == 0,
                "v3d_ln_proj: misaligned pointer");
    PP p;
    p.x = (const bf16_t*)x; p.gamma = gamma; p.beta = beta; p.W = (const bf16_t*)Wp; p.out = (bf16_t*)out; p.outT = (bf16_t*)outT;
    p.M = M; p.ldx = ldx; p.ldo = ldo; p.S = S > 0 ? S : M;
    p.N = N; p.n_rm = n_rm; p.Ct = N - n_rm;
    p.w_bytes = (unsigned)((size_t)N * C * 2);
    p.eps = eps;
    const long long nblocks = M / 128;
    const int cus = v3d_num_cus();
    const int grid = nblocks < cus ? (int)nblocks : cus;
    hipLaunchKernelGGL((ln_proj_kernel<320>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    return v3d_check_launch("v3d_ln_proj");
}
