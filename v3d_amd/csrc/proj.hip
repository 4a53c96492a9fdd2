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
    const float* bias;    // [N] fp32: W beta (LayerNorm shift folded through the projection), or NULL
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

__device__ unsigned long long g_pj_dbg[8 * 32 * 8];   // timeline build (V3D_LNPROJ_TIMELINE=1): [wave][stream slab 16..47][stamp]

// NW waves of 32 rows share every weight slab.  NW = 8 (two waves per SIMD, 256 rows per block) is the default: with one wave per SIMD every
// VMEM issue stall of a slab (LDS-DMA pieces ~70 cycles each, 16-byte stores ~375 each with the queue full of DMA traffic) starved the matrix
// pipe - 5500 cycles per slab against 1280 of MFMA work (tools/lnproj_timeline.py) - the second wave runs its MFMAs in those gaps, and the weight
// stream per row halves.
template <int C, int NW, bool DBG = false>
__global__ __launch_bounds__(64 * NW, NW / 4) void ln_proj_kernel(PP p) {
    constexpr int NK = C / 16;                  // k16 steps = resident row fragments
    constexpr int NT = C / 32;                  // 32-k LDS stages of a slab
    constexpr int SLAB = 64 * C * 2;            // 64 weight rows
    constexpr int PPW = NT * 4 / NW;            // 1-KiB pieces per wave per slab
    constexpr int BR = 32 * NW;                 // rows per block
    static_assert(PPW * NW == NT * 4, "pieces per wave");
    constexpr int NSLOT = 3;
    constexpr int GB_OFF = NSLOT * SLAB;        // output bias (W beta) copy
    constexpr int ST_OFF = GB_OFF + 960 * 4;    // (output bias copy: N <= 960 floats)  // per-wave output staging: 32 rows x (128 + 16) B, or 64 rows x (64 + 16) B for the transposed slabs
    constexpr int ST_BYTES = 32 * 144;          // 4608 >= 64 * 64 (the transposed tile is stored unpadded: LDS is full at 8 waves)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[ST_OFF + NW * ST_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const long long nblocks = p.M / BR;
    const int nslab = p.N / 64;
    const int nrm_slabs = p.n_rm / 64;
    float* gsm = reinterpret_cast<float*>(lds + GB_OFF);
    unsigned char* stg = lds + ST_OFF + wave * ST_BYTES;

    for (int c = tid; c < p.N; c += 64 * NW) gsm[c] = p.bias ? p.bias[c] : 0.f;

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
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + (wave + NW * i) * 1024), 16, (int)voff,
                                                     ld_slab * SLAB + i * NW * 1024, 0, 0);
        ld_slab = (ld_slab + 1 == nslab) ? 0 : ld_slab + 1;
        ld_slot = (ld_slot + 1 == NSLOT) ? 0 : ld_slot + 1;
    };

    // fragment of 32 rows x 16 k out of 64-byte rows: row l31, logical 16-byte chunk 2 (ks & 1) + hi
    const int foff0 = l31 * 64 + (((0 + hi) ^ pj_swz(l31)) * 16);
    const int foff1 = l31 * 64 + (((2 + hi) ^ pj_swz(l31)) * 16);

    bf16x8 xr[NK];
    auto load_rows = [&](long long blk) __attribute__((always_inline)) {
        const bf16_t* xz = p.x + (blk * BR + wave * 32 + l31) * p.ldx + hi * 8;
#pragma unroll
        for (int k = 0; k < NK; ++k) xr[k] = *reinterpret_cast<const bf16x8*>(xz + k * 16);
    };
    // LayerNorm of the wave's 32 rows in place: lane (l31, hi) holds channels 16 k + 8 hi .. + 7 of row l31, its partner lane the rest.
    // gamma and beta are folded into the weights / the output bias at pack time (W diag(gamma), W beta), so this is (x - mean) * rstd only.
    // A wave alone on its SIMD issues one VALU instruction per ~4-5 cycles, so the instruction count is what matters here: sums and sums of
    // squares by v_dot2_f32_bf16 on the packed pairs (1 instruction per 2 elements, exact bf16 products, fp32 accumulation) instead of unpack +
    // add / fma; variance as E[x^2] - mean^2 in fp32 (the first version: 3 unpacking passes + gamma / beta from LDS = 37k cycles per block).
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    auto normalise = [&]() __attribute__((always_inline)) {
        const bf16x2v ones = __builtin_bit_cast(bf16x2v, 0x3f803f80u);
        float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            // (element pairs picked with shufflevector: the u32x4 bit_cast + runtime-indexed subscript form of this loop was miscompiled by
            //  ROCm 7.2's clang - every dot2 read dword 0 of the fragment)
            const bf16x2v a = __builtin_shufflevector(xr[k], xr[k], 0, 1), b = __builtin_shufflevector(xr[k], xr[k], 2, 3);
            const bf16x2v c = __builtin_shufflevector(xr[k], xr[k], 4, 5), d = __builtin_shufflevector(xr[k], xr[k], 6, 7);
            s0 = __builtin_amdgcn_fdot2_f32_bf16(a, ones, s0, false);
            s1 = __builtin_amdgcn_fdot2_f32_bf16(b, ones, s1, false);
            q0 = __builtin_amdgcn_fdot2_f32_bf16(a, a, q0, false);
            q1 = __builtin_amdgcn_fdot2_f32_bf16(b, b, q1, false);
            s0 = __builtin_amdgcn_fdot2_f32_bf16(c, ones, s0, false);
            s1 = __builtin_amdgcn_fdot2_f32_bf16(d, ones, s1, false);
            q0 = __builtin_amdgcn_fdot2_f32_bf16(c, c, q0, false);
            q1 = __builtin_amdgcn_fdot2_f32_bf16(d, d, q1, false);
        }
        float s = s0 + s1, q = q0 + q1;
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        const float mean = s * (1.0f / C);
        const float var = fmaxf(q * (1.0f / C) - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps);
        const float sh = -mean * rstd;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const u32x4 u = __builtin_bit_cast(u32x4, xr[k]);
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack2bf(__builtin_fmaf(bflo(u[e]), rstd, sh), __builtin_fmaf(bfhi(u[e]), rstd, sh));
            xr[k] = __builtin_bit_cast(bf16x8, o);
        }
    };

    issue_slab();
    issue_slab();
    load_rows(blockIdx.x);
    __syncthreads();                       // the output bias copy is visible (drains the two slabs in flight once, at kernel start)

    long long j = 0;                       // weight-stream index of the slab about to be consumed
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if (DBG && blockIdx.x == 0 && j >= 16 && j < 48 && lane == 0) g_pj_dbg[(wave * 32 + (int)(j - 16)) * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    int rd_slot = 0;
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        normalise();
        const long long row0 = blk * BR + wave * 32;
        const long long img = row0 / p.S, pix0 = row0 - img * p.S;
        for (int sl = 0; sl < nslab; ++sl, ++j) {
            // this wave's pieces of slab j have landed when only the ops issued after them are outstanding: the next slab's PPW pieces,
            // the stores of the last two slabs (4 each) and - on the first two slabs of a block - the 20 row loads of the next block
            stamp(0);
            if (j < 2 || sl < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (after a block boundary the row loads sit in the queue: drain once)
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 8) : "memory");
            stamp(1);
            __builtin_amdgcn_s_barrier();          // everyone's pieces landed; everyone finished reading the slot refilled below
            asm volatile("" ::: "memory");
            stamp(2);
            if (j + 2 < total) issue_slab();
            else {                                   // stream tail: keep the per-slab op count constant for the counted waits
#pragma unroll
                for (int i = 0; i < PPW; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(lds + ld_slot * SLAB + (wave + NW * i) * 1024), 16,
                                                             (int)kInvalid, 0, 0, 0);
                ld_slot = (ld_slot + 1 == NSLOT) ? 0 : ld_slot + 1;
            }
            stamp(3);
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
                stamp(4);
                // through the wave's LDS staging tile, then whole 128-byte row segments as 16-byte-per-lane stores: 8-byte stores at a row
                // stride (32 rows per instruction) are store-ISSUE bound - the first version of this kernel spent 2/3 of its time in them
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                    {
                        const float4 bb = *reinterpret_cast<const float4*>(gsm + sl * 64 + t * 32 + 8 * g + 4 * hi);
                        *reinterpret_cast<uint2*>(stg + l31 * 144 + (t * 32 + 8 * g + 4 * hi) * 2) =
                            make_uint2(pack2bf(acc[t][4 * g + 0] + bb.x, acc[t][4 * g + 1] + bb.y), pack2bf(acc[t][4 * g + 2] + bb.z, acc[t][4 * g + 3] + bb.w));
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bf16_t* op = p.out + row0 * p.ldo + sl * 64;
                stamp(5);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + (lane >> 3), ch = lane & 7;
                    *reinterpret_cast<u32x4*>(op + (long long)row * p.ldo + ch * 8) = *reinterpret_cast<const u32x4*>(stg + row * 144 + ch * 16);
                }
                stamp(6);
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
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                    {
                        const float bb = gsm[sl * 64 + t * 32 + l31];
                        *reinterpret_cast<uint2*>(stg + (t * 32 + l31) * 64 + (8 * g + 4 * hi) * 2) =
                            make_uint2(pack2bf(acc[t][4 * g + 0] + bb, acc[t][4 * g + 1] + bb), pack2bf(acc[t][4 * g + 2] + bb, acc[t][4 * g + 3] + bb));
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bf16_t* op = p.outT + (img * p.Ct + (sl - nrm_slabs) * 64) * p.S + pix0;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 16 + (lane >> 2), ch = lane & 3;      // channel row of the slab, 16-byte piece of its 32 tokens
                    *reinterpret_cast<u32x4*>(op + (long long)row * p.S + ch * 8) = *reinterpret_cast<const u32x4*>(stg + row * 64 + ch * 16);
                }
            }
        }
        // the next block's rows come straight into the (now dead) fragment registers: the load latency is exposed once per block, but a
        // second register set held across the block spilled, and the scratch reloads' vmcnt(0) waits drained the LDS-DMA stream
        const long long nxt = blk + gridDim.x;
        if (nxt < nblocks) load_rows(nxt);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

// experiments only (not part of the ABI header): copy the slab timeline of the instrumented build out
extern "C" int v3d_debug_lnproj_timeline(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pj_dbg), sizeof(g_pj_dbg)) == hipSuccess ? 0 : -1;
}

extern "C" int v3d_ln_proj(const void* x, int64_t ldx, float eps, const void* Wp, const float* bias, void* out, int64_t ldo, void* outT,
                           int64_t M, int32_t C, int32_t N, int32_t n_rm, int64_t S, v3d_stream_t stream) {
    V3D_REQUIRE(x && Wp, "v3d_ln_proj: null pointer");
    V3D_REQUIRE(C == 320, "v3d_ln_proj: C must be 320 (the 64x64 level; other widths use v3d_layernorm + v3d_gemm), got %d", C);
    V3D_REQUIRE(M > 0 && M % 128 == 0, "v3d_ln_proj: M must be a positive multiple of 128 (got %lld)", (long long)M);
    V3D_REQUIRE(N >= 128 && N <= 960 && N % 64 == 0 && n_rm >= 0 && n_rm <= N && n_rm % 64 == 0, "v3d_ln_proj: N / n_rm must be multiples of 64 with n_rm <= N, 128 <= N <= 960");
    V3D_REQUIRE(n_rm == 0 || (out && ldo % 8 == 0 && ldo >= n_rm), "v3d_ln_proj: bad out / ldo (ldo %% 8 == 0)");
    V3D_REQUIRE(n_rm == N || (outT && S > 0 && S % 128 == 0 && M % S == 0), "v3d_ln_proj: the transposed part needs outT and S %% 128 == 0, M %% S == 0");
    V3D_REQUIRE(ldx % 8 == 0 && ldx >= C, "v3d_ln_proj: bad ldx");
    V3D_REQUIRE((((uintptr_t)x | (uintptr_t)Wp) & 15) == 0 && (((uintptr_t)out | (uintptr_t)outT) & 15) == 0 && ((uintptr_t)bias & 3) == 0,
                "v3d_ln_proj: misaligned pointer");
    PP p;
    p.x = (const bf16_t*)x; p.bias = bias; p.W = (const bf16_t*)Wp; p.out = (bf16_t*)out; p.outT = (bf16_t*)outT;
    p.M = M; p.ldx = ldx; p.ldo = ldo; p.S = S > 0 ? S : M;
    p.N = N; p.n_rm = n_rm; p.Ct = N - n_rm;
    p.w_bytes = (unsigned)((size_t)N * C * 2);
    p.eps = eps;
    static int tl = -1, w4 = -1;
    if (tl < 0) { const char* e = getenv("V3D_LNPROJ_TIMELINE"); tl = e ? atoi(e) : 0; }
    if (w4 < 0) { const char* e = getenv("V3D_LNPROJ_WAVES"); w4 = e ? atoi(e) : 8; }     // A/B knob: 4 = one wave per SIMD, 128-row blocks
    const bool eight = w4 != 4 && M % 256 == 0 && (n_rm == N || S % 256 == 0);
    const long long nblocks = M / (eight ? 256 : 128);
    const int cus = v3d_num_cus();
    const int grid = nblocks < cus ? (int)nblocks : cus;
    if (eight) {
        if (tl) hipLaunchKernelGGL((ln_proj_kernel<320, 8, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((ln_proj_kernel<320, 8>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
    } else {
        if (tl) hipLaunchKernelGGL((ln_proj_kernel<320, 4, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((ln_proj_kernel<320, 4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    }
    return v3d_check_launch("v3d_ln_proj");
}
