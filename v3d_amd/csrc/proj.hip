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
    // work list (host: v3d_ln_proj): every block runs `full` whole row blocks (b, b + G, ...); the R = nblocks - full * G row blocks that are left are cut
    // into `parts` slab ranges each (the output channels of a row block are independent: no reduction, the rows are simply loaded by `parts` blocks) and
    // block b < R * parts takes range b % parts of row block full * G + b / parts.  parts = 1: the classic assignment (R <= G blocks take one more).
    int full, tail_blocks, parts;
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

// NW waves of RF x 32 rows share every weight slab (block = NW * RF * 32 rows):
//   <8, 1>  two waves per SIMD, 32 rows each;   <4, 2>  one wave per SIMD, 64 rows each (every weight fragment read from LDS feeds two MFMAs:
//   at 32 rows per wave the fragment reads alone need the whole LDS port, 1 KiB per 32-cycle MFMA per SIMD);   <4, 1>  128-row blocks for
//   shapes the other two do not divide.
// One iteration of the slab loop = the 40 * RF MFMAs of slab j in 10 fenced steps of 2 k-steps, with the PREVIOUS slab's output path slotted
// between them: its tile was rounded to bf16 (bias added) into 16 * RF registers at the top of the iteration, ahead of the barrier; steps 0-4
// carry the LDS-DMA pieces of slab j + 2 and the staging writes, steps 5-7 the staging reads and the 16-byte-per-lane global stores.  (The
// first version ran "multiply, then stage, then store" per slab with all waves in the same phase: 5900 cycles per slab for 2560 of MFMA.)
template <int C, int NW, int RF, bool DBG = false>
__global__ __launch_bounds__(64 * NW, NW / 4) void ln_proj_kernel(PP p) {
    constexpr int NK = C / 16;                  // k16 steps = resident row fragments per 32 rows
    constexpr int NT = C / 32;                  // 32-k LDS stages of a slab
    constexpr int SLAB = 64 * C * 2;            // 64 weight rows
    constexpr int PPW = NT * 4 / NW;            // 1-KiB pieces per wave per slab
    constexpr int RW = 32 * RF;                 // rows per wave
    constexpr int BR = RW * NW;                 // rows per block
    constexpr int NSTEP = NK / 4 * 2;           // fenced steps of 2 k-steps
    constexpr int NST = RW / 8;                 // 1-KiB store instructions per wave per slab
    constexpr int NP = RF * 8;                  // (row fragment, channel half, g) register pairs of a tile
    static_assert(PPW * NW == NT * 4 && PPW % 5 == 0 && NK % 4 == 0 && NSTEP == 10, "schedule below is written for C = 320");
    constexpr int NSLOT = 3;
    constexpr int ST_BYTES = RW * 144;          // per-wave output staging: RW rows x (128 + 16) B row-major; 64 channel rows x RW tokens (swizzled) transposed
    constexpr int RB = RW * 2, NCH = RB / 16;   // transposed staging tile: bytes and 16-byte chunks per channel row
    // ONE LDS object, and the staging writes as inline asm: with LDS-DMA in flight the compiler's wait-count pass puts s_waitcnt vmcnt(0) in
    // front of every ds_write it can see (the first version's 1200-cycle "staging" phase was that wait); with several LDS objects it has alias
    // scopes and spares the staging writes, but then makes every ring READ wait for the DMA piece issued just before it.  A single object
    // gets no such waits on loads, and it cannot see these stores.
    constexpr int GB_OFF = NSLOT * SLAB;        // output bias (W beta) copy: N <= 960 floats
    constexpr int ST_OFF = GB_OFF + 960 * 4;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[ST_OFF + NW * ST_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const long long nblocks = p.M / BR;
    const int nslab = p.N / 64;
    const int nrm_slabs = p.n_rm / 64;
    float* gsm = reinterpret_cast<float*>(lds + GB_OFF);
    unsigned char* stg = lds + ST_OFF + wave * ST_BYTES;
    const unsigned stg_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)stg;      // LDS byte address

    for (int c = tid; c < p.N; c += 64 * NW) gsm[c] = p.bias ? p.bias[c] : 0.f;

    // ---- weight stream: flat over (row block, slab); stream index j -> slab j % nslab, ring slot j % 3
    const bufrsrc_t rsW = make_rsrc(p.W, p.w_bytes);
    const unsigned voff = (unsigned)(wave * 1024 + lane * 16);
    const int bid = (int)blockIdx.x;
    const bool has_tail = bid < p.tail_blocks * p.parts;
    const int tpart = has_tail ? bid % p.parts : 0;
    const int ts0 = has_tail ? tpart * nslab / p.parts : 0, ts1 = has_tail ? (tpart + 1) * nslab / p.parts : 0;      // the tail item's slabs [ts0, ts1)
    const long long tail_blk = (long long)p.full * gridDim.x + (has_tail ? bid / p.parts : 0);
    const long long total = (long long)p.full * nslab + (ts1 - ts0);
    (void)nblocks;
    int ld_slab = p.full > 0 ? 0 : ts0, ld_slot = 0, ld_full_left = p.full;
    bool ld_live = true;
    auto issue_piece = [&](int i) __attribute__((always_inline)) {      // piece i of this wave's PPW pieces of the slab being loaded
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(lds + ld_slot * SLAB + (wave + NW * i) * 1024), 16,
                                                 (int)(ld_live ? voff : kInvalid), ld_live ? ld_slab * SLAB + i * NW * 1024 : 0, 0, 0);
    };
    auto advance_load = [&]() __attribute__((always_inline)) {
        // the loader walks the consumer's slab sequence: 0 .. nslab - 1 per whole row block, then the tail item's range (past the end: harmless wrap-around)
        if (++ld_slab == nslab && ld_full_left > 0) {
            ld_slab = 0;
            if (--ld_full_left == 0) ld_slab = ts0;
        } else if (ld_slab >= nslab) {
            ld_slab = 0;
        }
        ld_slot = (ld_slot + 1 == NSLOT) ? 0 : ld_slot + 1;
    };

    // fragment of 32 rows x 16 k out of 64-byte rows: row l31, logical 16-byte chunk 2 (ks & 1) + hi
    const int foff0 = l31 * 64 + (((0 + hi) ^ pj_swz(l31)) * 16);
    const int foff1 = l31 * 64 + (((2 + hi) ^ pj_swz(l31)) * 16);

    bf16x8 xr[RF][NK];
    auto load_rows = [&](long long blk) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < RF; ++r) {
            const bf16_t* xz = p.x + (blk * BR + wave * RW + r * 32 + l31) * p.ldx + hi * 8;
#pragma unroll
            for (int k = 0; k < NK; ++k) xr[r][k] = *reinterpret_cast<const bf16x8*>(xz + k * 16);
        }
    };
    // LayerNorm of the wave's rows in place (common.h ln_rows_inplace: dot2 statistics, gamma / beta folded into the weights / the output bias)
    auto normalise = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < RF; ++r) ln_rows_inplace<NK>(xr[r], p.eps);
    };

    // ---- output path of one slab tile, cut into the pieces the slab loop slots between its MFMAs
    //   row-major (TRE = 0): acc[r][t][4 g + c] = out[token row0 + 32 r + l31][channel 64 sl + 32 t + 8 g + 4 hi + c]
    //   transposed (TRE = 1): acc[r][t][4 g + c] = outT[image][channel 64 (sl - nrm_slabs) + 32 t + l31][token pix0 + 32 r + 8 g + 4 hi + c]
    f32x16 acc[RF][2];
    u32x2 pk[RF][2][4];
    u32x4 sr[NST];
    long long p_row0 = 0, p_img = 0, p_pix0 = 0;    // the tile in pk / staging: its rows and slab
    int p_sl = 0;
    auto tswz = [&](int row) __attribute__((always_inline)) { return (NCH == 4 ? (row >> 1) : row) & (NCH - 1); };
    auto pack_tile = [&](auto tre_) __attribute__((always_inline)) {
        constexpr bool TRE = decltype(tre_)::value;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if constexpr (!TRE) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(gsm + p_sl * 64 + t * 32 + 8 * g + 4 * hi);
#pragma unroll
                    for (int r = 0; r < RF; ++r)
                        pk[r][t][g] = u32x2{pack2bf(acc[r][t][4 * g + 0] + bb[0], acc[r][t][4 * g + 1] + bb[1]),
                                            pack2bf(acc[r][t][4 * g + 2] + bb[2], acc[r][t][4 * g + 3] + bb[3])};
                }
            } else {
                const float bb = gsm[p_sl * 64 + t * 32 + l31];
#pragma unroll
                for (int r = 0; r < RF; ++r)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        pk[r][t][g] = u32x2{pack2bf(acc[r][t][4 * g + 0] + bb, acc[r][t][4 * g + 1] + bb),
                                            pack2bf(acc[r][t][4 * g + 2] + bb, acc[r][t][4 * g + 3] + bb)};
            }
        }
    };
    auto stage_write = [&](auto tre_, auto i_) __attribute__((always_inline)) {       // pair i = (r, t, g)
        constexpr bool TRE = decltype(tre_)::value;
        constexpr int i = decltype(i_)::value, r = i / 8, t = (i / 4) & 1, g = i & 3;
        if constexpr (!TRE) {
            const unsigned a = stg_a + l31 * 144 + hi * 8;
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(pk[r][t][g]), "n"(r * 32 * 144 + (t * 32 + 8 * g) * 2) : "memory");
        } else {
            const int row = t * 32 + l31;
            const unsigned a = stg_a + row * RB + (((r * 4 + g) ^ tswz(row)) * 16) + hi * 8;
            asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(pk[r][t][g]) : "memory");
        }
    };
    auto stage_read = [&](auto tre_, auto i_) __attribute__((always_inline)) {
        constexpr bool TRE = decltype(tre_)::value;
        constexpr int it = decltype(i_)::value;
        if constexpr (!TRE) {
            const int row = it * 8 + (lane >> 3), ch = lane & 7;
            sr[it] = *reinterpret_cast<const u32x4*>(stg + row * 144 + ch * 16);
        } else {
            const int row = it * (64 / NCH) + lane / NCH, ch = lane % NCH;      // channel row of the slab, 16-byte piece of its RW tokens
            sr[it] = *reinterpret_cast<const u32x4*>(stg + row * RB + ((ch ^ tswz(row)) * 16));
        }
    };
    auto store_out = [&](auto tre_, auto i_) __attribute__((always_inline)) {
        constexpr bool TRE = decltype(tre_)::value;
        constexpr int it = decltype(i_)::value;
        if constexpr (!TRE) {
            const int row = it * 8 + (lane >> 3), ch = lane & 7;
            *reinterpret_cast<u32x4*>(p.out + (p_row0 + row) * p.ldo + p_sl * 64 + ch * 8) = sr[it];
        } else {
            const int row = it * (64 / NCH) + lane / NCH, ch = lane % NCH;
            *reinterpret_cast<u32x4*>(p.outT + (p_img * p.Ct + (p_sl - nrm_slabs) * 64 + row) * p.S + p_pix0 + ch * 8) = sr[it];
        }
    };

    auto issue_all = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(i);
        advance_load();
    };
    issue_all();
    issue_all();
#pragma unroll
    for (int r = 0; r < RF; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][t][e] = 0.f;

    long long j = 0;                       // weight-stream index of the slab about to be consumed
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if (DBG && blockIdx.x == 0 && j >= 16 && j < 48 && lane == 0) g_pj_dbg[(wave * 32 + (int)(j - 16)) * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    int rd_slot = 0, sl = p.full > 0 ? 0 : ts0, sl0 = sl, sl_end = p.full > 0 ? nslab : ts1, full_left = p.full;
    long long blk = p.full > 0 ? (long long)blockIdx.x : tail_blk, row0 = 0, img = 0, pix0 = 0;

    // one slab: multiply slab `sl` of the current block (TRM: transposed output, MFMA operands swapped), push the previous tile (TRE) out
    auto iter = [&](auto trm_, auto tre_) __attribute__((always_inline)) {
        constexpr bool TRM = decltype(trm_)::value;
        pack_tile(tre_);
#pragma unroll
        for (int r = 0; r < RF; ++r)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[r][t][e] = 0.f;
        // this wave's pieces of slab j have landed when only the ops issued after them are outstanding: the stores of tile j - 3 (issued
        // after the pieces in iteration j - 2), and iteration j - 1's pieces and stores.  Around a block boundary (row loads in the queue,
        // no stores in the very first iteration) drain instead.
        stamp(0);
        if (j < 3 || sl - sl0 < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 2 * NST) : "memory");
        stamp(1);
        __builtin_amdgcn_s_barrier();          // everyone's pieces landed; everyone finished reading the slot refilled below
        asm volatile("" ::: "memory");
        stamp(2);
        ld_live = j + 2 < total;               // stream tail: dummy pieces keep the per-iteration op count constant for the counted waits
        const unsigned char* sb = lds + rd_slot * SLAB;
        rd_slot = (rd_slot + 1 == NSLOT) ? 0 : rd_slot + 1;
        const bool have_prev = j > 0;
        bf16x8 wf[2][2][2];                    // [buffer][k-step of the pair][channel half]
        auto load_frags = [&](auto s_) __attribute__((always_inline)) {
            constexpr int s = decltype(s_)::value;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    wf[s & 1][q][t] = *reinterpret_cast<const bf16x8*>(sb + s * 4096 + t * 2048 + (q ? foff1 : foff0));
        };
        load_frags(std::integral_constant<int, 0>{});
        pj_static_for<0, NSTEP>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            if constexpr (s + 1 < NSTEP) load_frags(std::integral_constant<int, s + 1>{});
            pj_static_for<0, 2>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                pj_static_for<0, RF>([&](auto r_) {
                    constexpr int r = decltype(r_)::value;
                    if constexpr (!TRM) {
                        acc[r][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s & 1][q][0], xr[r][2 * s + q], acc[r][0], 0, 0, 0);
                        acc[r][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s & 1][q][1], xr[r][2 * s + q], acc[r][1], 0, 0, 0);
                    } else {
                        acc[r][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xr[r][2 * s + q], wf[s & 1][q][0], acc[r][0], 0, 0, 0);
                        acc[r][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xr[r][2 * s + q], wf[s & 1][q][1], acc[r][1], 0, 0, 0);
                    }
                });
            });
            if constexpr (s < 5) {
                pj_static_for<s * (PPW / 5), (s + 1) * (PPW / 5)>([&](auto i_) { issue_piece(decltype(i_)::value); });
                pj_static_for<s * NP / 5, (s + 1) * NP / 5>([&](auto i_) { stage_write(tre_, i_); });
            } else if constexpr (s == 5) {
                pj_static_for<0, NST / 2>([&](auto i_) { stage_read(tre_, i_); });
            } else if constexpr (s == 6) {
                if (have_prev) pj_static_for<0, NST / 2>([&](auto i_) { store_out(tre_, i_); });
                pj_static_for<NST / 2, NST>([&](auto i_) { stage_read(tre_, i_); });
            } else if constexpr (s == 7) {
                if (have_prev) pj_static_for<NST / 2, NST>([&](auto i_) { store_out(tre_, i_); });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        advance_load();
        stamp(3);
    };

    // (rows are loaded AND normalised in one place, so that no row load is pending on the loop back edge: with the normalisation at the top
    //  of the next iteration the compiler kept vmcnt(10..20) waits in front of the MFMAs of every step, which drain the weight stream)
    auto begin_block = [&]() __attribute__((always_inline)) {
        load_rows(blk);
        normalise();
        row0 = blk * BR + wave * RW;
        img = row0 / p.S;
        pix0 = row0 - img * p.S;
    };
    if (total > 0) begin_block();
    __syncthreads();                       // the output bias copy is visible
    for (; j < total; ++j) {
        const bool trm = sl >= nrm_slabs, tre = p_sl >= nrm_slabs;      // wave-uniform
        if (!trm) {
            if (!tre) iter(std::false_type{}, std::false_type{});
            else iter(std::false_type{}, std::true_type{});
        } else {
            if (!tre) iter(std::true_type{}, std::false_type{});
            else iter(std::true_type{}, std::true_type{});
        }
        p_row0 = row0; p_img = img; p_pix0 = pix0; p_sl = sl;
        if (++sl == sl_end) {
            // the next item's rows come straight into the (now dead) fragment registers: the load latency is exposed once per item
            if (--full_left > 0) {
                sl = sl0 = 0;
                blk += gridDim.x;
                begin_block();
            } else if (full_left == 0 && has_tail) {
                sl = sl0 = ts0;
                sl_end = ts1;
                blk = tail_blk;
                begin_block();
            }
        }
    }
    // the last tile
    auto drain = [&](auto tre_) __attribute__((always_inline)) {
        pack_tile(tre_);
        pj_static_for<0, NP>([&](auto i_) { stage_write(tre_, i_); });
        pj_static_for<0, NST>([&](auto i_) { stage_read(tre_, i_); });
        pj_static_for<0, NST>([&](auto i_) { store_out(tre_, i_); });
    };
    if (total > 0) {
        if (p_sl >= nrm_slabs) drain(std::true_type{});
        else drain(std::false_type{});
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
    static int tl = -1, cfg = -1;
    if (tl < 0) { const char* e = getenv("V3D_LNPROJ_TIMELINE"); tl = e ? atoi(e) : 0; }
    if (cfg < 0) { const char* e = getenv("V3D_LNPROJ_CFG"); cfg = e ? atoi(e) : 1; }     // A/B knob: 0 = <4 waves, 64 rows> (161 us at level 0), 1 = <8, 32> (148 us, default), 2 = <4, 32> (164 us)
    const bool big = cfg != 2 && M % 256 == 0 && (n_rm == N || S % 256 == 0);
    const long long nblocks = M / (big ? 256 : 128);
    const int cus = v3d_num_cus();
    // work list: whole rounds of row blocks, then the remaining R row blocks cut into `parts` slab ranges each so that the last round is (nearly) as
    // wide as the chip: M = 147456 at 256 rows = 576 row blocks = 2 rounds + 64 -> 4 parts of 3-4 of the 15 slabs instead of a third round a quarter
    // full.  V3D_LNPROJ_SPLIT=0: the classic assignment (A/B knob).  At least 3 slabs per part (the rows are loaded and normalised once per part).
    static int split = -1;
    if (split < 0) { const char* e = getenv("V3D_LNPROJ_SPLIT"); split = e ? atoi(e) : 1; }
    int grid = nblocks < cus ? (int)nblocks : cus;
    p.full = (int)(nblocks / grid);
    p.tail_blocks = (int)(nblocks - (long long)p.full * grid);
    p.parts = 1;
    if (p.tail_blocks == 0 && p.full > 0) { p.full -= 1; p.tail_blocks = grid; }      // (uniform form: the last round is the "tail" of one whole part per block)
    if (split && p.tail_blocks > 0) {
        int parts = cus / p.tail_blocks, maxp = (N / 64) / 3;
        if (parts > maxp) parts = maxp;
        if (parts < 1) parts = 1;
        p.parts = parts;
        if (p.full == 0) grid = p.tail_blocks * parts;
    }
    if (big && cfg == 0) {
        if (tl) hipLaunchKernelGGL((ln_proj_kernel<320, 4, 2, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((ln_proj_kernel<320, 4, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    } else if (big) {
        if (tl) hipLaunchKernelGGL((ln_proj_kernel<320, 8, 1, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((ln_proj_kernel<320, 8, 1>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
    } else {
        hipLaunchKernelGGL((ln_proj_kernel<320, 4, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    }
    return v3d_check_launch("v3d_ln_proj");
}
