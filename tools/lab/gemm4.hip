// LAB RECORD, NOT PRODUCT CODE (round 5: moved out of v3d_amd/csrc; linked only into the experiments library, `python -m v3d_amd.build --experiments`,
// and selected there with V3D_GEMM_V4=1).  Bit-equal to v3 and 5 % slower (NOTES 11.2).  Its literal-AGPR accumulators are protected by nothing but
// register pressure: the AGPR audit script its comments cite (tools/check_agpr.py) was never committed - do not promote this file into
// the product library without one.
//
// v4: persistent 4-wave kernels of the v3d_gemm family - ONE wave per SIMD, software-pipelined inside the wave (gfx950).
//
// Why (round 4; profiles/r04_mainloop_ab.txt, r04_sq_counters.txt): the v3 kernels run two wave groups per SIMD half a step apart - a wave is
// either in its "read" phase (11-12 ds_read_b128, its LDS-DMA pieces, waits, the step barrier) or in its MFMA phase, and the partner covers
// the other half.  SQ counters of the 64 x 64 convolution: a wave spends 32 % of its cycles parked on s_waitcnt / s_barrier and 32 % stalled
// at issue; the matrix pipes are busy 41-50 %.  The read phase is not hidden by anything inside the wave itself, the per-step barrier ties
// eight waves together, and every piece of work that is not an MFMA (LDS-DMA issue: ~66 cycles per piece with four waves queueing on the TA)
// lengthens a phase the partner has to match.
// Here a block has four waves, each alone on its SIMD with the whole 512-register file:
//   * wave tile 96 x 160 (192 x 320 block tile, the N = 320 k family) or 128 x 128 (256 x 256): 16 fragment reads feed 60 / 64 MFMAs
//     (v3: 11 / 12 reads per 30 / 32) - a quarter less LDS traffic per flop;
//   * TWO fragment register sets: the reads of step s + 1 and the LDS-DMA pieces of step s + 4 are issued from INSIDE the MFMA sequence of
//     step s (one per issue slot, fenced the ff.hip way), so the matrix pipe of a SIMD is fed by one uninterrupted instruction stream;
//   * one synchronisation point per step: {own fragments landed, own DMA pieces of the next stage landed, s_barrier} - the ring runs four
//     stages ahead of the MFMAs (three ahead of the reads), across tile boundaries;
//   * the epilogue is the hand-managed one of the v3 kernels (gemm_common.h e4_*), run per 64- / 80-channel half of the wave tile.
// Modes: LINEAR and the plain (no operand-path GroupNorm) CONV3X3 / CONVT3 implicit GEMMs through the same RowInfo addressing as v3.
#include <stdlib.h>

#include "gemm_common.h"

// timing experiments (tools/build_variant.sh <tag> "-DV4_ABL=<bits>" gemm4.hip; results are garbage by design): 1 no vmcnt wait, 2 no barrier,
// 4 no lgkmcnt wait, 8 no fragment reads, 16 no LDS-DMA, 32 no epilogue
#ifndef V4_ABL
#define V4_ABL 0
#endif
#define V4A(bit) ((V4_ABL & (bit)) != 0)

namespace {

// ---- accumulators in the accumulator file, by name -----------------------------------------------------------------------------------------
// 240 / 256 accumulator registers + two fragment sets (128) + loader state exceed the 256 architectural VGPRs of a wave; left to the register
// allocator the accumulators wander between the two files (the first build of this kernel carried 4 v_accvgpr_write / v_accvgpr_mov per MFMA).
// So the accumulators are literal AGPRs a[4 n .. 4 n + 3] that only the statements below touch: the MFMAs (C = D = the same AGPRs), the zero
// fill, and the epilogue's reads.  One statement lists all of them as clobbers, which makes the kernel descriptor allocate them; everything
// else in the kernel stays below 256 VGPRs, so the compiler has no reason to use an AGPR (tools/check_agpr.py audits the assembly for
// compiler-made v_accvgpr_* - cdna_hip_programming.md section 5.7 item 4).
#include "agpr.h"
template <int N>
__device__ __forceinline__ void acc_mfma(bf16x8 w, bf16x8 x) {      // acc[N] += W fragment (A operand) x activation fragment (B operand)
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(x), "i"(4 * N), "i"(4 * N + 3));
}
template <int N>
__device__ __forceinline__ void acc_zero() {
    asm volatile("v_accvgpr_write_b32 a[%c0], 0\n\tv_accvgpr_write_b32 a[%c1], 0\n\tv_accvgpr_write_b32 a[%c2], 0\n\tv_accvgpr_write_b32 a[%c3], 0" ::"i"(4 * N), "i"(4 * N + 1),
                 "i"(4 * N + 2), "i"(4 * N + 3));
}
template <int N>
__device__ __forceinline__ f32x4 acc_read() {
    f32x4 r;
    asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                 : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3])
                 : "i"(4 * N), "i"(4 * N + 1), "i"(4 * N + 2), "i"(4 * N + 3));
    return r;
}
// accumulator source of the hand-managed epilogue (gemm_common.h E4AccArray is the register-array form): fragment row F of half H
template <int NF, int NFH, int H>
struct AgprAcc {
    template <int F>
    __device__ __forceinline__ void get(f32x4 (&out)[NFH]) const {
        static_for<0, NFH>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            out[j] = acc_read<F * NF + H * NFH + j>();
        });
    }
};

__device__ __forceinline__ int wswz4(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }     // as gemm.hip swz_row<1>

template <int BM, int BN, int MODE, bool GN>
__global__ __launch_bounds__(256, 1) void gemm_kernel_v4(GP p, int ntiles) {
    constexpr int ROWB = 64, NW = 4, NS = 4;
    constexpr int WM = BM / 2, WN = BN / 2;               // waves 2 (M) x 2 (N)
    constexpr int MF = WM / 16, NF = WN / 16, NFH = NF / 2;
    static_assert(NF % 2 == 0 && (NFH == 4 || NFH == 5), "the epilogue retires 64- or 80-channel halves");
    constexpr int NPIECE = (BM + BN) / 16, PPW = NPIECE / NW, APIECES = BM / 16;
    static_assert(PPW * NW == NPIECE, "tile / wave-count mismatch");
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int EPI_REGION = 16 * (NFH * 32 + 16);
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

    // ---- loader (as v3: raw buffer loads straight to LDS, per-lane (row, k-chunk) byte offset in a VGPR that changes per tile / tap, the k
    //      position in an SGPR; padding rows / tails are out-of-range offsets that DMA zeros)
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
    auto issue_piece = [&](int stage, int i, int so) __attribute__((always_inline)) {
        const int q = wave + NW * i;
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

    // accumulator n = i * NF + j (row fragment i, channel fragment j of the wave tile) lives in a[4 n .. 4 n + 3]
    asm volatile("" ::: V3D_ALL_AGPRS);
    static_for<0, MF * NF>([&](auto n_) { acc_zero<decltype(n_)::value>(); });
    const unsigned frag_a = (unsigned)((lane & 15) * ROWB + (((lane >> 4) ^ wswz4(lane & 15)) * 16) + wm * WM * ROWB);
    const unsigned frag_b = (unsigned)((lane & 15) * ROWB + (((lane >> 4) ^ wswz4(lane & 15)) * 16) + BM * ROWB + wn * WN * ROWB);
    bf16x8 xa[MF], wa[NF], xb[MF], wb[NF];               // fragment sets of even / odd steps

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
    for (int i = 0; i < MF; ++i) xa[i] = *reinterpret_cast<const bf16x8*>(lds + frag_a + i * 16 * ROWB);
#pragma unroll
    for (int j = 0; j < NF; ++j) wa[j] = *reinterpret_cast<const bf16x8*>(lds + frag_b + j * 16 * ROWB);

    int rd = 1;       // ring slot of the stage this step READS (step s reads stage s + 1); the slot before it (stage s) is refilled
    // one step: MFMAs of the fragments in (xc, wc); meanwhile stage s + 1 -> (xn, wn_) and the DMA pieces of stage s + 4
    auto step = [&](bf16x8 (&xc)[MF], bf16x8 (&wc)[NF], bf16x8 (&xn)[MF], bf16x8 (&wn_)[NF]) __attribute__((always_inline)) {
        // the one synchronisation point of the step: own fragments of this step in registers (read during the previous step), own pieces of
        // the next stage landed (two younger stages in flight), everybody there: stage s + 1 is complete and visible, and nobody reads stage
        // s any more - its slot is the refill target
        if (!V4A(4)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!V4A(1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NS - 2)) : "memory");
        if (!V4A(2)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sb = lds + rd * STAGE_BYTES;
        const int wst = rd == 0 ? NS - 1 : rd - 1;
        rd = (rd + 1 == NS) ? 0 : rd + 1;
        const int so = __builtin_amdgcn_readfirstlane(ld_k0 * 2);
        constexpr int NM = MF * NF, NR = MF + NF;
        constexpr int RS = (NM - 8) / NR;                 // a fragment read every RS slots, the last one 8+ slots before the step ends
        constexpr int DS = NM / PPW;                      // a DMA piece every DS slots
        static_assert(RS >= 1 && DS >= 1, "slot plan");
        static_for<0, NM>([&](auto n_) {
            constexpr int n = decltype(n_)::value, i = n / NF, j = n % NF;
            acc_mfma<n>(wc[j], xc[i]);
            if constexpr (n % RS == 0 && n / RS < NR && !V4A(8)) {
                constexpr int r = n / RS;
                // weight fragments first: the next step's first MFMAs need all of them and one activation fragment
                if constexpr (r < NF) wn_[r] = *reinterpret_cast<const bf16x8*>(sb + frag_b + r * 16 * ROWB);
                else xn[r - NF] = *reinterpret_cast<const bf16x8*>(sb + frag_a + (r - NF) * 16 * ROWB);
            }
            if constexpr (n % DS == DS / 2 && n / DS < PPW && !V4A(16)) issue_piece(wst, n / DS, so);
            __builtin_amdgcn_sched_barrier(0);
        });
        issue_advance();
    };

    for (int it = 0; it < my_tiles; ++it) {
        for (int kt = 0; kt < nsteps; kt += 2) {
            step(xa, wa, xb, wb);
            step(xb, wb, xa, wa);
        }
        // ---------------- tile finished: retire it (hand-managed epilogue, one 64- / 80-channel half of the wave tile at a time)
        long long e_m0, e_n0;
        tile_origin(it, e_m0, e_n0);
        const long long mw0 = e_m0 + wm * WM;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));          // (keeps what the epilogue derives from the lane id out of the loop-invariant set)
        auto rowfn = [&](int f) __attribute__((always_inline)) -> long long { return mw0 + f * 16; };
        E4GnRun<WM, MF> run0, run1;
        if constexpr (GN) {
            run0.init(mw0, p.gn_rps);
            run1 = run0;
        }
        auto flush0 = [&](int f, long long, unsigned& slot, unsigned& sid) __attribute__((always_inline)) -> bool { return run0.step(f, slot, sid); };
        auto flush1 = [&](int f, long long, unsigned& slot, unsigned& sid) __attribute__((always_inline)) -> bool { return run1.step(f, slot, sid); };
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // (the last MFMAs' results are in the accumulator file before the first read)
        if (mw0 < p.M && !V4A(32)) {
            const long long nwa = e_n0 + wn * WN;
            if (nwa < p.N) e4_retire_tile_src<MF, NFH, GN>(p, AgprAcc<NF, NFH, 0>{}, nwa, lane_e, estage, rowfn, flush0);
            if (nwa + NFH * 16 < p.N) e4_retire_tile_src<MF, NFH, GN>(p, AgprAcc<NF, NFH, 1>{}, nwa + NFH * 16, lane_e, estage, rowfn, flush1);
        }
        static_for<0, MF * NF>([&](auto n_) { acc_zero<decltype(n_)::value>(); });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int MODE>
int launch_v4_mode(const GP& p0, int variant, hipStream_t st) {
    GP p = p0;
    const int bm = variant == 1 ? 192 : 256, bn = variant == 1 ? 320 : 256;
    p.mt = (int)((p.M + bm - 1) / bm);
    p.nt = (int)((p.N + bn - 1) / bn);
    const int ntiles = p.mt * p.nt;
    const int grid = ntiles < v3d_num_cus() ? ntiles : v3d_num_cus();
    const bool gn = p.gn_stats != nullptr;
    if (variant == 1) {
        if (gn) hipLaunchKernelGGL((gemm_kernel_v4<192, 320, MODE, true>), dim3(grid), dim3(256), 0, st, p, ntiles);
        else hipLaunchKernelGGL((gemm_kernel_v4<192, 320, MODE, false>), dim3(grid), dim3(256), 0, st, p, ntiles);
    } else {
        if (gn) hipLaunchKernelGGL((gemm_kernel_v4<256, 256, MODE, true>), dim3(grid), dim3(256), 0, st, p, ntiles);
        else hipLaunchKernelGGL((gemm_kernel_v4<256, 256, MODE, false>), dim3(grid), dim3(256), 0, st, p, ntiles);
    }
    return v3d_check_launch("v3d_gemm(v4)");
}

}  // namespace

// 0 = not a launch of these kernels (the caller keeps its v3 / v2 path); 1 = 192 x 320 tiles, 2 = 256 x 256 tiles.
// (the caller has already decided that persistent big tiles fill the chip for this shape)
int v3d_gemm_v4_variant(const V3dGemmParams& p, int mode, int v3_variant) {
    if (p.out_fp32 || p.split_n > 1 || p.K % 64 || p.K * 2 > 65536) return 0;
    const int taps = mode == V3D_GEMM_LINEAR ? 1 : (mode == V3D_GEMM_CONV3X3 ? 9 : 3);
    if (((p.K / 32) * taps) % 2) return 0;
    if (v3_variant == 1) {
        if (!e4_ok(p, 96, 80)) return 0;
        if (p.gn_stats && (80 % p.gn_cpg || p.gn_nslots < p.gn_rps / 96 + 2)) return 0;
        return 1;
    }
    if (!e4_ok(p, 128, 64)) return 0;
    if (p.gn_stats && (64 % p.gn_cpg || p.gn_nslots < p.gn_rps / 128 + 2)) return 0;
    return 2;
}

int v3d_gemm_v4_launch(const V3dGemmParams& p, int mode, int variant, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
        case V3D_GEMM_LINEAR: return launch_v4_mode<V3D_GEMM_LINEAR>(p, variant, st);
        case V3D_GEMM_CONV3X3: return launch_v4_mode<V3D_GEMM_CONV3X3>(p, variant, st);
        case V3D_GEMM_CONVT3: return launch_v4_mode<V3D_GEMM_CONVT3>(p, variant, st);
    }
    v3d_set_error("v3d_gemm(v4): unknown mode %d", mode);
    return V3D_ERR_ARG;
}
