// LAB KERNEL (round 6; experiments library only, V3D_GEMM_V5=1): the one-wave-per-SIMD persistent GEMM of gemm4.hip on v_mfma_f32_32x32x16_bf16.
//
// Why: every GEMM / convolution main loop of the product is built on v_mfma_f32_16x16x32_bf16.  Round 4 found that inside ONE wave
// nothing overlaps such an MFMA: it holds the wave for its 16 issue cycles (4 issue slots, ~3 of them the MFMA's own), so every fragment read / LDS-DMA
// piece of the same wave adds its own issue time - which is why v4 (this structure on 16 x 16 x 32) lost 5 % to the two-group v3 loop.
// A 32 x 32 x 16 MFMA does twice the work per instruction (32 cycles = 8 issue slots; MI355X_MICROARCH.md measures <= 5 single-issue fillers HIDDEN
// per gap, and a 2382 vs 2075 TF/s micro-benchmark ceiling): the same 16 fragment reads + 8 DMA pieces per 32-k step now face 32 MFMAs with ~160
// free slots instead of 64 MFMAs with ~64.
//
// Structure (as v4): 4 waves, each alone on its SIMD with the whole 512-register file; 256 x 256 block tile, wave tile 128 x 128 = 4 x 4 accumulator
// tiles of 32 x 32 (16 registers each) in literal AGPRs; 4-stage LDS-DMA ring of 32-k stages (64-byte rows, chunk position XOR-swizzled on the source
// address - the swizzle of gemm.hip is conflict-free for the 32-row fragments too: the four ds_read_b128 lane groups {0-3,12-15,20-27}, ... each cover
// four 4-row blocks with distinct (row >> 2) & 3); two fragment register sets, the reads of step s + 1 and the DMA pieces of step s + 4 issued from
// inside the MFMA sequence of step s; ONE {lgkmcnt, counted vmcnt, barrier} point per step.
// Weight fragment = A operand (32 channels x 16 k), activation fragment = B operand (16 k x 32 pixels): lane (p = lane & 31, h = lane >> 5) ends up with
// channels 8 g + 4 h + (0..3), g = 0..3, of pixel p per accumulator tile - four 8-byte bf16 pieces of one output row.
#include <stdlib.h>

#include "gemm_common.h"
#include "agpr.h"

// timing experiments (results are garbage by design): 1 no vmcnt wait, 2 no barrier, 4 no lgkmcnt wait, 8 no fragment reads, 16 no LDS-DMA, 32 no epilogue
#ifndef V5_ABL
#define V5_ABL 0
#endif
#define V5A(bit) ((V5_ABL & (bit)) != 0)
// slot plan of a step: 0 = two reads + one DMA piece per three MFMAs (everything issued in the first 24 gaps), 1 = a read in every even gap, a DMA piece
// in every fourth odd gap, 2 = all reads back to back in the first gaps (one per MFMA), DMA pieces behind them, 3 = WAVE-STAGGERED: the four waves of a
// block run in lockstep behind the step barrier, so identical streams send their four DMA pieces to the one TA (16 cycles per 1-KiB piece) in the same
// cycle and three of them queue; here wave w issues its d-th piece in gap 4 d + w and its reads in the other gaps - every gap sees ONE piece CU-wide
#ifndef V5_PLAN
#define V5_PLAN 0
#endif

namespace {

template <int N>
__device__ __forceinline__ void acc32_mfma(bf16x8 w, bf16x8 x) {      // acc tile N += W fragment (A operand) x activation fragment (B operand)
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(x), "i"(16 * N), "i"(16 * N + 15));
}
template <int R>
__device__ __forceinline__ void agpr_zero4() {
    asm volatile("v_accvgpr_write_b32 a[%c0], 0\n\tv_accvgpr_write_b32 a[%c1], 0\n\tv_accvgpr_write_b32 a[%c2], 0\n\tv_accvgpr_write_b32 a[%c3], 0" ::"i"(R), "i"(R + 1), "i"(R + 2),
                 "i"(R + 3));
}
template <int R>
__device__ __forceinline__ f32x4 agpr_read4() {
    f32x4 r;
    asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                 : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3])
                 : "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
    return r;
}

__device__ __forceinline__ int wswz4(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }     // as gemm.hip swz_row<1>

// ---- epilogue: one UNIT = 32 rows x 64 channels (two accumulator tiles side by side) --------------------------------------------------------------
// Same arithmetic and the same discipline as gemm_common.h e4_* (asm loads one unit ahead, asm staging writes, counted waits, no wait for a store).
// Piece q = jj * 4 + g of a lane: channels jj * 32 + 8 g + 4 h .. + 3 of row p -> bytes [q * 16 + h * 8, + 8) of the staged 128-byte row.
constexpr int E5_SROW = 128 + 16;
struct E5Res { u32x4 a0, a1, a2, a3; };            // residual rows of a unit: chunk c = k * 64 + lane -> row c >> 3, 16-byte column chunk c & 7
struct E5Tile {
    f32x4 ba[8];                                   // bias + per-row-group vector of the lane's 8 pieces
    float ca, c1, c2;
    unsigned add_grp, coef_grp, add_rem, coef_rem;
};
__device__ __forceinline__ void e5_load_res(const GP& p, long long m0, long long nwu, int lane, E5Res& r) {
    const bf16_t* b = p.res1 + (m0 + (lane >> 3)) * p.ldr1 + nwu + (lane & 7) * 8;
    r.a0 = e4_load16(b);
    r.a1 = e4_load16(b + 8 * p.ldr1);
    r.a2 = e4_load16(b + 16 * p.ldr1);
    r.a3 = e4_load16(b + 24 * p.ldr1);
}
template <int N>
__device__ __forceinline__ void e5_wait_cnt(E5Res& r) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r.a0), "+v"(r.a1), "+v"(r.a2), "+v"(r.a3) : "n"(N) : "memory");
}
__device__ __forceinline__ void e5_consts(const GP& p, long long m0, long long nwu, int lane, E5Tile& t) {
    const int nb = (int)nwu + (lane >> 5) * 4;
    f32x4 bv[8], av[8];
    float cf0 = p.c_acc, cf1 = p.c_res1, cf2 = p.c_res2;
#pragma unroll
    for (int q = 0; q < 8; ++q) bv[q] = av[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int q = 0; q < 8; ++q) bv[q] = __builtin_bit_cast(f32x4, e4_load16(p.bias + nb + (q >> 2) * 32 + (q & 3) * 8));
    }
    t.add_grp = p.add ? e4_udiv((unsigned)m0, (unsigned)p.add_rpg) : 0u;
    t.add_rem = p.add ? (unsigned)m0 - t.add_grp * (unsigned)p.add_rpg : 0u;
    if (p.add) {
        const float* av0 = p.add + (long long)t.add_grp * p.add_ld + nb;
#pragma unroll
        for (int q = 0; q < 8; ++q) av[q] = __builtin_bit_cast(f32x4, e4_load16(av0 + (q >> 2) * 32 + (q & 3) * 8));
    }
    t.coef_grp = p.coef ? e4_udiv((unsigned)m0, (unsigned)p.coef_rpg) : 0u;
    t.coef_rem = p.coef ? (unsigned)m0 - t.coef_grp * (unsigned)p.coef_rpg : 0u;
    if (p.coef) {
        const float* cf = p.coef + (long long)t.coef_grp * 3;
        cf0 = e4_load4(cf);
        cf1 = e4_load4(cf + 1);
        cf2 = e4_load4(cf + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]), "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]),
                   "+v"(av[4]), "+v"(av[5]), "+v"(av[6]), "+v"(av[7]), "+v"(cf0), "+v"(cf1), "+v"(cf2)::"memory");
#pragma unroll
    for (int q = 0; q < 8; ++q) t.ba[q] = bv[q] + av[q];
    t.ca = cf0;
    t.c1 = cf1;
    t.c2 = cf2;
}
__device__ __forceinline__ void e5_unit(const GP& p, const f32x4 (&acc)[8], long long m0, long long nwu, int lane, unsigned char* stage, const E5Res& cur, const E5Tile& t) {
    const unsigned sbase = lds_addr(stage);
    const bool has1 = p.res1 != nullptr;
    const int prow = lane >> 3, pch = lane & 7;
    if (has1) {
        const unsigned a = sbase + (unsigned)(prow * E5_SROW + pch * 16);
        e4_lds_write16(a, cur.a0);
        e4_lds_write16(a + 8 * E5_SROW, cur.a1);
        e4_lds_write16(a + 16 * E5_SROW, cur.a2);
        e4_lds_write16(a + 24 * E5_SROW, cur.a3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const int mine_off = (lane & 31) * E5_SROW + (lane >> 5) * 8;
    const unsigned mine = sbase + (unsigned)mine_off;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (acc[q][r] + t.ba[q][r]) * t.ca;
        if (has1) {
            const uint2 rr = *reinterpret_cast<const uint2*>(stage + mine_off + q * 16);
            o[0] += t.c1 * bflo(rr.x); o[1] += t.c1 * bfhi(rr.x); o[2] += t.c1 * bflo(rr.y); o[3] += t.c1 * bfhi(rr.y);
        }
        e4_lds_write8(mine + q * 16, pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bf16_t* outz = reinterpret_cast<bf16_t*>(p.out) + (m0 + prow) * p.ldo + nwu + pch * 8;
    const unsigned char* srow = stage + prow * E5_SROW + pch * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<uint4*>(outz + (long long)(k * 8) * p.ldo) = *reinterpret_cast<const uint4*>(srow + k * 8 * E5_SROW);
}
// retire row block F .. MB - 1 of the 64-channel column H of the wave tile (accumulator tiles (F, 2 H), (F, 2 H + 1))
template <int F, int MB, int NB, int H>
__device__ __forceinline__ void e5_retire(const GP& p, long long mw0, long long nwu, int lane, unsigned char* stage, E5Res cur, E5Res nxt, E5Tile t) {
    if constexpr (F < MB) {
        const long long m0 = mw0 + F * 32;
        const bool has1 = p.res1 != nullptr;
        if constexpr (F + 1 < MB) {
            if (has1) e5_load_res(p, m0 + 32, nwu, lane, nxt);
        }
        if constexpr (F >= 1) {
            // pieces of unit F: issued at the top of unit F - 1; behind them that unit's 4 stores and the 4 loads of unit F + 1 (if it exists)
            if (has1) e5_wait_cnt<4 + (F + 1 < MB ? 4 : 0)>(cur);
        }
        if constexpr (F > 0) {
            t.add_rem += 32;
            t.coef_rem += 32;
            if ((p.add && t.add_rem >= (unsigned)p.add_rpg) || (p.coef && t.coef_rem >= (unsigned)p.coef_rpg)) {
                e5_consts(p, m0, nwu, lane, t);       // (rare: the wave tile straddles two row groups; its vmcnt(0) lands the look-ahead pieces too)
                asm volatile("" : "+v"(nxt.a0), "+v"(nxt.a1), "+v"(nxt.a2), "+v"(nxt.a3));
            }
        }
        {
            f32x4 a[8];
            static_for<0, 8>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                a[q] = agpr_read4<(F * NB + 2 * H + (q >> 2)) * 16 + (q & 3) * 4>();
            });
            e5_unit(p, a, m0, nwu, lane, stage, cur, t);
        }
        e5_retire<F + 1, MB, NB, H>(p, mw0, nwu, lane, stage, nxt, nxt, t);
    }
}
template <int MB, int NB, int H>
__device__ __forceinline__ void e5_retire_column(const GP& p, long long mw0, long long nwu, int lane, unsigned char* stage) {
    E5Res r0;
    r0.a0 = r0.a1 = r0.a2 = r0.a3 = u32x4{0u, 0u, 0u, 0u};
    if (p.res1) e5_load_res(p, mw0, nwu, lane, r0);
    E5Tile t;
    e5_consts(p, mw0, nwu, lane, t);                  // (its vmcnt(0) also lands the residual pieces issued above)
    asm volatile("" : "+v"(r0.a0), "+v"(r0.a1), "+v"(r0.a2), "+v"(r0.a3));
    e5_retire<0, MB, NB, H>(p, mw0, nwu, lane, stage, r0, r0, t);
}

template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256, 1) void gemm_kernel_v5(GP p, int ntiles) {
    constexpr int ROWB = 64, NW = 4, NS = 4;
    constexpr int WM = BM / 2, WN = BN / 2;               // waves 2 (M) x 2 (N)
    constexpr int MB = WM / 32, NB = WN / 32;
    static_assert(MB * NB * 16 <= 256 && NB % 2 == 0, "accumulator tiles / 64-channel epilogue columns");
    constexpr int NPIECE = (BM + BN) / 16, PPW = NPIECE / NW, APIECES = BM / 16;
    static_assert(PPW * NW == NPIECE, "tile / wave-count mismatch");
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int EPI_REGION = 32 * E5_SROW;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * STAGE_BYTES + NW * EPI_REGION];        // the ONLY __shared__ object
    static_assert(sizeof(lds) <= 160 * 1024, "LDS budget");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    unsigned char* estage = lds + NS * STAGE_BYTES + wave * EPI_REGION;

    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    auto tile_origin = [&](int it, long long& m0, long long& n0) __attribute__((always_inline)) {
        const int id = xcd_remap((int)blockIdx.x + it * G, ntiles);
        int tm, tn;
        tile_coords(p, id, tm, tn);
        n0 = (long long)tn * BN;
        m0 = (long long)tm * BM;
    };

    // ---- loader (as v3 / v4)
    const bufrsrc_t rsA = make_rsrc(p.A, p.a_bytes);
    const bufrsrc_t rsW = make_rsrc(p.W, p.w_bytes);
    const int prow = lane >> 2;
    const unsigned kchunk_b = (unsigned)(((lane & 3) ^ wswz4(prow)) * 16);
    RowInfo<MODE> ri[PPW];
    unsigned voff[PPW];
    long long ld_n0 = 0;
    int ld_it = 0, ld_tap = 0, ld_k0 = 0;
    auto set_tap = [&](int tap) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave + NW * i;
            if (q < APIECES) {
                long long s_;
                const bool ok = ri[i].tap(p, tap, s_);
                voff[i] = ok ? (unsigned)((s_ + p.a_row0) * p.lda * 2) + kchunk_b : kInvalid;
            } else {
                const long long n = ld_n0 + (q - APIECES) * 16 + prow;
                voff[i] = (n < p.N) ? (unsigned)(((long long)tap * p.N + n) * p.ldw * 2) + kchunk_b : kInvalid;
            }
        }
    };
    auto set_tile = [&](int it) __attribute__((always_inline)) {
        long long m0;
        tile_origin(it, m0, ld_n0);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave + NW * i;
            if (q < APIECES) ri[i].init(p, m0 + q * 16 + prow);
        }
        set_tap(0);
    };
    set_tile(0);
    u32x4 dummy_ld = {0u, 0u, 0u, 0u}, dummy_w = {1u, 2u, 3u, (unsigned)lane};
    asm volatile("" : "+v"(dummy_w));
    auto issue_piece = [&](int stage, int i, int so) __attribute__((always_inline)) {
        const int q = wave + NW * i;
        if constexpr (V5A(64)) {
            // timing experiment: the piece as a plain 16-byte buffer load into registers (what register staging would issue), never waited for or used;
            // bit 128 adds the ds_write_b128 a register-staged pipeline would need per piece (writes whatever the register holds)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dummy_ld) : "v"(voff[i]), "s"(q < APIECES ? rsA : rsW), "s"(so) : "memory");
            if constexpr (V5A(128)) asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(stage * STAGE_BYTES + q * 1024 + lane * 16)), "v"(dummy_w) : "memory");
            return;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(q < APIECES ? rsA : rsW, (__attribute__((address_space(3))) void*)(lds + stage * STAGE_BYTES + q * 1024), 16, (int)voff[i], so, 0, 0);
    };
    auto issue_advance = [&]() __attribute__((always_inline)) {
        ld_k0 += 32;
        if (ld_k0 >= (int)p.K) {
            ld_k0 = 0;
            if (++ld_tap < ntaps<MODE>()) {
                set_tap(ld_tap);
            } else {
                ld_tap = 0;
                if (++ld_it < my_tiles) set_tile(ld_it);   // past the last tile: harmless re-reads keep the DMA count constant
                else if (ntaps<MODE>() > 1) set_tap(0);
            }
        }
    };

    // accumulator tile n = i * NB + j (row block i, channel block j of the wave tile) lives in a[16 n .. 16 n + 15]
    asm volatile("" ::: V3D_ALL_AGPRS);
    static_for<0, MB * NB * 4>([&](auto n_) { agpr_zero4<decltype(n_)::value * 4>(); });
    // fragment of 32 rows x 16 k: lane (p = lane & 31, h = lane >> 5) reads the 16-byte k-chunk 2 ks + h of row p (stored at chunk ^ swizzle(row))
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned fpos0 = (unsigned)(l31 * ROWB + ((lh ^ wswz4(l31)) * 16)), fpos1 = (unsigned)(l31 * ROWB + (((2 + lh) ^ wswz4(l31)) * 16));
    const unsigned frag_a[2] = {fpos0 + wm * WM * ROWB, fpos1 + wm * WM * ROWB};
    const unsigned frag_b[2] = {fpos0 + BM * ROWB + wn * WN * ROWB, fpos1 + BM * ROWB + wn * WN * ROWB};
    bf16x8 xa[2][MB], wa[2][NB], xb[2][MB], wb[2][NB];               // fragment sets of even / odd steps

    const int nsteps = (int)(p.K / 32) * ntaps<MODE>();  // (host: even)

    // ---- prologue: stages 0 .. 3 fill the ring; stage 0 -> set a
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const int so = __builtin_amdgcn_readfirstlane(ld_k0 * 2);
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(st, i, so);
        issue_advance();
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NS - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < MB; ++i) xa[ks][i] = *reinterpret_cast<const bf16x8*>(lds + frag_a[ks] + i * 32 * ROWB);
#pragma unroll
        for (int j = 0; j < NB; ++j) wa[ks][j] = *reinterpret_cast<const bf16x8*>(lds + frag_b[ks] + j * 32 * ROWB);
    }

    int rd = 1;       // ring slot of the stage this step READS (step s reads stage s + 1); the slot before it (stage s) is refilled
    auto step = [&](auto w_, bf16x8 (&xc)[2][MB], bf16x8 (&wc)[2][NB], bf16x8 (&xn)[2][MB], bf16x8 (&wn_)[2][NB]) __attribute__((always_inline)) {
        constexpr int W = decltype(w_)::value;
        if (!V5A(4)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!V5A(1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NS - 2)) : "memory");
        if (!V5A(2)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sb = lds + rd * STAGE_BYTES;
        const int wst = rd == 0 ? NS - 1 : rd - 1;
        rd = (rd + 1 == NS) ? 0 : rd + 1;
        const int so = __builtin_amdgcn_readfirstlane(ld_k0 * 2);
        constexpr int TILES = MB * NB, NM = 2 * TILES, NR = 2 * (MB + NB);
        static_for<0, NM>([&](auto n_) {
            constexpr int n = decltype(n_)::value, ks = n / TILES, r = n % TILES, i = r / NB, j = r % NB;
            acc32_mfma<i * NB + j>(wc[ks][j], xc[ks][i]);
            // which read / DMA piece (if any) rides in the gap behind MFMA n
            constexpr int rdx = V5_PLAN == 0 ? ((n % 3 != 2) ? n - n / 3 : -1) : (V5_PLAN == 1 ? ((n % 2 == 0) ? n / 2 : -1) : (V5_PLAN == 2 ? n : ((n % 4 != W) ? n - (n + 3 - W) / 4 : -1)));
            constexpr int ddx = V5_PLAN == 0 ? ((n % 3 == 2) ? n / 3 : -1) : (V5_PLAN == 1 ? ((n % 4 == 1) ? n / 4 : -1) : (V5_PLAN == 2 ? n - NR : ((n % 4 == W) ? n / 4 : -1)));
            if constexpr (rdx >= 0 && rdx < NR && !V5A(8)) {
                // order: the ks = 0 fragments (weights, then activations), then ks = 1
                constexpr int rks = rdx / (MB + NB), rr = rdx % (MB + NB);
                if constexpr (rr < NB) wn_[rks][rr] = *reinterpret_cast<const bf16x8*>(sb + frag_b[rks] + rr * 32 * ROWB);
                else xn[rks][rr - NB] = *reinterpret_cast<const bf16x8*>(sb + frag_a[rks] + (rr - NB) * 32 * ROWB);
            }
            if constexpr (ddx >= 0 && ddx < PPW && !V5A(16)) issue_piece(wst, ddx, so);
            __builtin_amdgcn_sched_barrier(0);
        });
        issue_advance();
    };

    for (int it = 0; it < my_tiles; ++it) {
        auto run_steps = [&](auto w_) __attribute__((always_inline)) {
            for (int kt = 0; kt < nsteps; kt += 2) {
                step(w_, xa, wa, xb, wb);
                step(w_, xb, wb, xa, wa);
            }
        };
        if constexpr (V5_PLAN == 3) {
            if (wave == 0) run_steps(std::integral_constant<int, 0>{});
            else if (wave == 1) run_steps(std::integral_constant<int, 1>{});
            else if (wave == 2) run_steps(std::integral_constant<int, 2>{});
            else run_steps(std::integral_constant<int, 3>{});
        } else {
            run_steps(std::integral_constant<int, 0>{});
        }
        long long e_m0, e_n0;
        tile_origin(it, e_m0, e_n0);
        const long long mw0 = e_m0 + wm * WM, nw0 = e_n0 + wn * WN;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));          // (keeps what the epilogue derives from the lane id out of the loop-invariant set)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // (the last MFMAs' results are in the accumulator file before the first read)
        if (mw0 < p.M && !V5A(32)) {
            static_for<0, NB / 2>([&](auto h_) {
                constexpr int H = decltype(h_)::value;
                if (nw0 + H * 64 < p.N) e5_retire_column<MB, NB, H>(p, mw0, nw0 + H * 64, lane_e, estage);
            });
        }
        static_for<0, MB * NB * 4>([&](auto n_) { agpr_zero4<decltype(n_)::value * 4>(); });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int MODE>
int launch_v5_mode(const GP& p0, hipStream_t st) {
    GP p = p0;
    p.mt = (int)((p.M + 255) / 256);
    p.nt = (int)((p.N + 255) / 256);
    const int ntiles = p.mt * p.nt;
    const int grid = ntiles < v3d_num_cus() ? ntiles : v3d_num_cus();
    hipLaunchKernelGGL((gemm_kernel_v5<256, 256, MODE>), dim3(grid), dim3(256), 0, st, p, ntiles);
    v3d_note_launch(6, 256, 256, ntiles, 1, 0);
    return v3d_check_launch("v3d_gemm(v5)");
}

}  // namespace

// 0 = not a launch of this kernel (the caller keeps its v3 / v2 path); 2 = 256 x 256 tiles.  (Lab: no residual #2, no GroupNorm-statistics epilogue.)
int v3d_gemm_v5_variant(const V3dGemmParams& p, int mode, int v3_variant) {
    if (v3_variant != 0 || p.out_fp32 || p.split_n > 1 || p.K % 64 || p.K * 2 > 65536 || p.res2 || p.gn_stats) return 0;
    const int taps = mode == V3D_GEMM_LINEAR ? 1 : (mode == V3D_GEMM_CONV3X3 ? 9 : 3);
    if (((p.K / 32) * taps) % 2) return 0;
    if (!e4_ok(p, 128, 64)) return 0;
    if ((p.add && p.add_rpg % 32) || (p.coef && p.coef_rpg % 32)) return 0;
    return 2;
}

int v3d_gemm_v5_launch(const V3dGemmParams& p, int mode, int variant, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    (void)variant;
    switch (mode) {
        case V3D_GEMM_LINEAR: return launch_v5_mode<V3D_GEMM_LINEAR>(p, st);
        case V3D_GEMM_CONV3X3: return launch_v5_mode<V3D_GEMM_CONV3X3>(p, st);
        case V3D_GEMM_CONVT3: return launch_v5_mode<V3D_GEMM_CONVT3>(p, st);
    }
    v3d_set_error("v3d_gemm(v5): unknown mode %d", mode);
    return V3D_ERR_ARG;
}
