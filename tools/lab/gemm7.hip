// LAB KERNEL (round 6; experiments library only, V3D_GEMM_V7=1): the K <= 640, N = 320 linears (to_out / proj_in / proj_out / skip of the 64 x 64 level) with a
// DEFERRED epilogue.
//
// Why (profiles/r06_timeline_k320_*.txt): on M = 147456, N = K = 320 the persistent 192 x 320 kernel spends 25 k cycles in the ten steps of a tile and then 14 k
// (bias only) to 33 k (bias + vector + residual) cycles in the tile's epilogue - six fragments, each a memory round trip - during which no MFMA runs and no load of
// the next tile is issued.  Here nothing of the epilogue waits for memory:
//   * the residual rows of tile t are fetched (inline-asm loads into registers) during steps 3 .. 5 of tile t, four steps before they are used;
//   * at the end of tile t the wave FINALISES its 48 x 80 outputs in registers (bias / vector, residual through the staging rows, bf16 pack) - VALU and LDS only;
//   * the packed rows leave during steps 0 .. 2 of tile t + 1 (staging rows -> 16-byte row stores), in the shadow of that tile's main loop.
// The memory operations that ride in the steps are counted EXACTLY in the steps' counted vmcnt waits (an uncounted one would make the wait either race or drain):
// the first ten steps of a tile are therefore straight-line code with a static schedule; K = 640 runs ten more plain steps.
// Tile 96 x 320 (one tile column: N = 320), 8 waves = two groups of four, wave tile 48 x 80 (3 x 5 fragments: 60 accumulator registers), the two-group loop and the
// 4-stage LDS-DMA ring of gemm_kernel_v3 (26 pieces per stage + 6 dummies so that every wave issues 4).
#include <stdlib.h>

#include "gemm_common.h"

namespace {

template <int OFF>
__device__ __forceinline__ void frag_read(bf16x8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory"); }

__device__ __forceinline__ int swz7(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }     // as gemm.hip swz_row<1>

struct E7Res { u32x4 a0, a1, a2; };

template <bool RES>
__global__ __launch_bounds__(512, 2) void gemm_kernel_v7(GP p, int ntiles) {
    constexpr int BM = 96, BN = 320, ROWB = 64, NW = 8, NS = 4, WGN = 4;
    constexpr int WM = 48, WN = 80, MF = 3, NF = 5;
    constexpr int NPIECE = (BM + BN) / 16, APIECES = BM / 16, PPW = 4;                 // 26 pieces + 6 dummies
    constexpr int STAGE_BYTES = NPIECE * 1024;
    constexpr int DUMMY_OFF = NS * STAGE_BYTES, EPI_OFF = DUMMY_OFF + 1024, SROW = NF * 32 + 16, EPI_REGION = 16 * SROW;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[EPI_OFF + NW * EPI_REGION];      // the ONLY __shared__ object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave / WGN, wn = wave % WGN;
    unsigned char* estage = lds + EPI_OFF + wave * EPI_REGION;
    const unsigned sbase = lds_addr(estage);
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    auto tile_m0 = [&](int it) __attribute__((always_inline)) -> long long { return ((long long)blockIdx.x + (long long)it * G) * BM; };

    // ---- loader: piece q = wave + 8 i; q < 6 activation rows, q < 26 weight rows (N = 320: every tile needs all of them), else a dummy
    const bufrsrc_t rsA = make_rsrc(p.A, p.a_bytes);
    const bufrsrc_t rsW = make_rsrc(p.W, p.w_bytes);
    const int prow = lane >> 2;
    const unsigned kchunk_b = (unsigned)(((lane & 3) ^ swz7(prow)) * 16);
    unsigned voff[PPW];
    int ld_it = 0, ld_k0 = 0;
    auto set_tile = [&](int it) __attribute__((always_inline)) {
        const long long m0 = tile_m0(it);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave + NW * i;
            if (q < APIECES) {
                const long long m = m0 + q * 16 + prow;
                voff[i] = m < p.M ? (unsigned)((m + p.a_row0) * p.lda * 2) + kchunk_b : kInvalid;
            } else if (q < NPIECE) {
                voff[i] = (unsigned)(((long long)(q - APIECES) * 16 + prow) * p.ldw * 2) + kchunk_b;
            } else {
                voff[i] = kInvalid;
            }
        }
    };
    set_tile(0);
    auto issue = [&](int stage) __attribute__((always_inline)) {
        const int so = __builtin_amdgcn_readfirstlane(ld_k0 * 2);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave + NW * i;
            const int dst = q < NPIECE ? stage * STAGE_BYTES + q * 1024 : DUMMY_OFF;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(q < APIECES ? rsA : rsW, (__attribute__((address_space(3))) void*)(lds + dst), 16, (int)voff[i], so, 0, 0);
        }
        ld_k0 += 32;
        if (ld_k0 >= (int)p.K) {
            ld_k0 = 0;
            if (++ld_it < my_tiles) set_tile(ld_it);      // past the last tile: harmless re-reads keep the DMA count constant
        }
    };

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frag_off = (lane & 15) * ROWB + (((lane >> 4) ^ swz7(lane & 15)) * 16);
    const int a_base = wm * WM * ROWB;
    const int b_base = BM * ROWB + wn * WN * ROWB;
    bf16x8 xf[MF], wf[NF];
    static_assert(MF == 3 && NF == 5, "the fragment reads of step() are written out");
    const unsigned a_addr = lds_addr(lds) + (unsigned)(a_base + frag_off), b_addr = lds_addr(lds) + (unsigned)(b_base + frag_off);
    const int nsteps = (int)(p.K / 32);                // (host: 10 or 20)

    // ---- per-wave epilogue state
    const int fr = lane & 15, fq = (lane >> 4) * 4;
    const long long nw0 = wn * WN;
    f32x4 ba[NF];                                       // bias + per-row-group vector of the lane's 4 channels per fragment column (group add_grp)
    unsigned add_grp = 0xffffffffu;
    auto load_consts = [&](unsigned grp_) __attribute__((always_inline)) {       // synchronous: kernel start and when a fragment enters another row group (rare)
        const int nb = (int)nw0 + fq;
        f32x4 bv[NF], av[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) bv[j] = av[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < NF; ++j) bv[j] = __builtin_bit_cast(f32x4, e4_load16(p.bias + nb + j * 16));
        }
        if (p.add) {
            const float* av0 = p.add + (long long)grp_ * p.add_ld + nb;
#pragma unroll
            for (int j = 0; j < NF; ++j) av[j] = __builtin_bit_cast(f32x4, e4_load16(av0 + j * 16));
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4])::"memory");
#pragma unroll
        for (int j = 0; j < NF; ++j) ba[j] = bv[j] + av[j];
        add_grp = grp_;
    };
    E7Res rp[MF];                                       // residual rows of the CURRENT tile: 16-byte row pieces per fragment (chunk c = k * 64 + lane of the 16 x 10 grid)
#pragma unroll
    for (int i = 0; i < MF; ++i) rp[i].a0 = rp[i].a1 = rp[i].a2 = u32x4{0u, 0u, 0u, 0u};
    u32x2 pend[MF][NF];                                 // packed outputs of the PREVIOUS tile (lane: 4 channels of row fr per fragment column)
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) pend[i][j] = u32x2{0u, 0u};
    long long pend_mw0 = 0;
    constexpr int CPRO = NF * 2, NP = 16 * CPRO;        // 16-byte chunks per row / per fragment (160)
    auto piece_addr = [&](int k) __attribute__((always_inline)) -> unsigned {
        const int c = k * 64 + lane;
        return sbase + (unsigned)((c / CPRO) * SROW + (c % CPRO) * 16);
    };
    auto load_res_frag = [&](long long m0f, E7Res& r) __attribute__((always_inline)) {
        int ln = lane;
        asm volatile("" : "+v"(ln));          // (opaque: as loop invariants the piece addresses were hoisted in front of the tile loop and spilled - scratch reloads inside the steps)
        auto piece = [&](int k) __attribute__((always_inline)) -> u32x4 {
            int c = k * 64 + ln;
            if (c >= NP) c = ln;
            return e4_load16(p.res1 + (m0f + c / CPRO) * p.ldr1 + nw0 + (c % CPRO) * 8);
        };
        r.a0 = piece(0);
        r.a1 = piece(1);
        r.a2 = piece(2);
    };
    // pending fragment F: staging rows -> 16-byte-per-lane row stores (3 VMEM operations)
    auto store_pending = [&](auto f_) __attribute__((always_inline)) {
        constexpr int F = decltype(f_)::value;
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        const int fr = lane & 15, fq = (lane >> 4) * 4;
        const unsigned mine = sbase + (unsigned)(fr * SROW + fq * 2);
#pragma unroll
        for (int j = 0; j < NF; ++j) e4_lds_write8(mine + j * 32, pend[F][j][0], pend[F][j][1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bf16_t* outz = reinterpret_cast<bf16_t*>(p.out) + (pend_mw0 + F * 16) * p.ldo + nw0;
        u32x4 sr[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int c = k * 64 + lane;
            if (c >= NP) c = lane;
            e4_lds_read16(sr[k], sbase + (unsigned)((c / CPRO) * SROW + (c % CPRO) * 16));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sr[0]), "+v"(sr[1]), "+v"(sr[2])::"memory");
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = k * 64 + lane;
            const int row = c / CPRO, ch = c % CPRO;
            if (c < NP) *reinterpret_cast<u32x4*>(outz + (long long)row * p.ldo + ch * 8) = sr[k];
        }
    };
    // end of a tile: acc (+ constants, residual) -> bf16 -> pend.  VALU and LDS only (the residual pieces landed steps ago: the statement below only NAMES them)
    auto finalize = [&](long long mw0) __attribute__((always_inline)) {
        if (RES) asm volatile("s_waitcnt vmcnt(63)" : "+v"(rp[0].a0), "+v"(rp[0].a1), "+v"(rp[0].a2), "+v"(rp[1].a0), "+v"(rp[1].a1), "+v"(rp[1].a2), "+v"(rp[2].a0), "+v"(rp[2].a1),
                                                                "+v"(rp[2].a2)::"memory");
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        const int fr = lane & 15, fq = (lane >> 4) * 4;
        const int mine_off = fr * SROW + fq * 2;
        auto piece_addr = [&](int k) __attribute__((always_inline)) -> unsigned {
            const int c = k * 64 + lane;
            return sbase + (unsigned)((c / CPRO) * SROW + (c % CPRO) * 16);
        };
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const long long m0f = mw0 + i * 16;
            if (p.add) {
                const unsigned g_ = e4_udiv((unsigned)m0f, (unsigned)p.add_rpg);
                if (g_ != add_grp) load_consts(g_);
            }
            if (RES) {
                e4_lds_write16(piece_addr(0), rp[i].a0);
                e4_lds_write16(piece_addr(1), rp[i].a1);
                if (lane < NP - 128) e4_lds_write16(piece_addr(2), rp[i].a2);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            u32x2 rr[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) rr[j] = u32x2{0u, 0u};
            if (RES) {
#pragma unroll
                for (int j = 0; j < NF; ++j) e4_lds_read8(rr[j], sbase + (unsigned)(mine_off + j * 32));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rr[0]), "+v"(rr[1]), "+v"(rr[2]), "+v"(rr[3]), "+v"(rr[4])::"memory");
            }
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (acc[i][j][r] + ba[j][r]) * p.c_acc;
                if (RES) {
                    o[0] += p.c_res1 * bflo(rr[j][0]); o[1] += p.c_res1 * bfhi(rr[j][0]); o[2] += p.c_res1 * bflo(rr[j][1]); o[3] += p.c_res1 * bfhi(rr[j][1]);
                }
                pend[i][j] = u32x2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        pend_mw0 = mw0;
    };

    load_consts(0u);
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) issue(st);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NS - 2)) : "memory");
    __builtin_amdgcn_s_barrier();   // B_0: stage 0 landed
    __builtin_amdgcn_sched_barrier(0);

    int rd = 0;
    // one step of the two-group loop (gemm.hip gemm_kernel_v3) with the memory operations TRICKLE issues behind its MFMAs; XM2 / XM1 / X0 = how many VMEM
    // operations the trickles of steps s - 2 / s - 1 / s issue (exact: they are younger than the pieces of stage s + 1 the step-end wait is for)
    auto step = [&](auto xm2_, auto xm1_, auto x0_, auto trickle) __attribute__((always_inline)) {
        constexpr int XM2 = decltype(xm2_)::value, XM1 = decltype(xm1_)::value, X0 = decltype(x0_)::value;
        {
            // (asm reads: with the trickled stores in flight the compiler would put vmcnt(0) in front of every LDS read it can see)
            const unsigned sa = a_addr + (unsigned)(rd * STAGE_BYTES), sw = b_addr + (unsigned)(rd * STAGE_BYTES);
            frag_read<0>(xf[0], sa); frag_read<16 * ROWB>(xf[1], sa); frag_read<32 * ROWB>(xf[2], sa);
            frag_read<0>(wf[0], sw); frag_read<16 * ROWB>(wf[1], sw); frag_read<32 * ROWB>(wf[2], sw); frag_read<48 * ROWB>(wf[3], sw); frag_read<64 * ROWB>(wf[4], sw);
        }
        issue(rd == 0 ? NS - 1 : rd - 1);
        rd = (rd + 1 == NS) ? 0 : rd + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(wf[4])::"memory");
        if (grp == 1) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NS - 2) + XM2 + XM1) : "memory");   // own pieces of stage s+1 landed (this step's trickle comes after)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        trickle();
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NS - 2) + XM2 + XM1 + X0) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using Z = std::integral_constant<int, 0>;
    using T3 = std::integral_constant<int, 3>;
    auto none = [&]() __attribute__((always_inline)) {};

    // the first ten steps of a tile: the pending rows of the PREVIOUS tile in steps 0 .. 2, the residual fetch of THIS tile in steps 3 .. 5
    auto head = [&](auto pend_, long long mw0) __attribute__((always_inline)) {
        constexpr bool PEND = decltype(pend_)::value;
        using R = std::integral_constant<int, RES ? 3 : 0>;
        using S = std::integral_constant<int, PEND ? 3 : 0>;
        auto res0 = [&]() __attribute__((always_inline)) { if (RES) load_res_frag(mw0, rp[0]); };
        auto res1 = [&]() __attribute__((always_inline)) { if (RES) load_res_frag(mw0 + 16, rp[1]); };
        auto res2 = [&]() __attribute__((always_inline)) { if (RES) load_res_frag(mw0 + 32, rp[2]); };
        auto st0 = [&]() __attribute__((always_inline)) { if (PEND) store_pending(std::integral_constant<int, 0>{}); };
        auto st1 = [&]() __attribute__((always_inline)) { if (PEND) store_pending(std::integral_constant<int, 1>{}); };
        auto st2 = [&]() __attribute__((always_inline)) { if (PEND) store_pending(std::integral_constant<int, 2>{}); };
        // (the pending rows leave FIRST and the residual pieces are fetched behind them: `pend` and `rp` are never live together - 30 registers less)
        step(Z{}, Z{}, S{}, st0);       // 0  (the previous tile's steps 8, 9 carried nothing)
        step(Z{}, S{}, S{}, st1);       // 1
        step(S{}, S{}, S{}, st2);       // 2
        step(S{}, S{}, R{}, res0);      // 3
        step(S{}, R{}, R{}, res1);      // 4
        step(R{}, R{}, R{}, res2);      // 5  (four steps - and two counted waits - ahead of the finalize that reads them)
        step(R{}, R{}, Z{}, none);      // 6
        step(R{}, Z{}, Z{}, none);      // 7
        step(Z{}, Z{}, Z{}, none);      // 8
        step(Z{}, Z{}, Z{}, none);      // 9
    };
    for (int it = 0; it < my_tiles; ++it) {
        const long long mw0 = tile_m0(it) + wm * WM;
        if (it == 0) head(std::false_type{}, mw0);
        else head(std::true_type{}, mw0);
        for (int kt = 10; kt < nsteps; ++kt) step(Z{}, Z{}, Z{}, none);
        finalize(mw0);
    }
    // the last tile's rows
    if (my_tiles > 0) {
        store_pending(std::integral_constant<int, 0>{});
        store_pending(std::integral_constant<int, 1>{});
        store_pending(std::integral_constant<int, 2>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

// 0 = not a launch of this kernel.  (Lab: LINEAR, N = 320, K = 320 / 640, bias / per-row-group vector / residual #1 with scalar coefficients.)
int v3d_gemm_v7_variant(const V3dGemmParams& p, int mode) {
    auto al = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) % a) == 0; };
    if (mode != V3D_GEMM_LINEAR || p.N != 320 || (p.K != 320 && p.K != 640) || p.M % 96 || p.out_fp32 || p.split_n > 1 || p.res2 || p.coef || p.gn_stats) return 0;
    if (p.ldo % 8 || !al(p.out, 16) || p.M >= (1ll << 31)) return 0;
    if (p.bias && !al(p.bias, 16)) return 0;
    if (p.add && (!al(p.add, 16) || p.add_ld % 4 || p.add_rpg % 16)) return 0;
    if (p.res1 && (!al(p.res1, 16) || p.ldr1 % 8)) return 0;
    return 1;
}

int v3d_gemm_v7_launch(const V3dGemmParams& p0, void* stream) {
    V3dGemmParams p = p0;
    const int ntiles = (int)(p.M / 96);
    const int grid = ntiles < v3d_num_cus() ? ntiles : v3d_num_cus();
    if (p.res1) hipLaunchKernelGGL((gemm_kernel_v7<true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, ntiles);
    else hipLaunchKernelGGL((gemm_kernel_v7<false>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, ntiles);
    v3d_note_launch(7, 96, 320, ntiles, 1, 0);
    return v3d_check_launch("v3d_gemm(v7)");
}
