// v3d_gemm: multi-tap bf16 MFMA contraction for gfx950 (see include/v3d_hip.h for the op contract).
//
// One kernel family covers nn.Linear / 1x1 conv (1 tap), Conv2d 3x3 incl. stride-2 and fused nearest-2x upsample
// (9 taps, implicit GEMM over channels-last pixels) and the (3,1,1) temporal conv (3 taps along the frame stride).
// v_mfma_f32_16x16x32_bf16 with the WEIGHT fragment as the A operand, so each lane ends up with 4 consecutive output
// channels of one pixel (8-byte bf16 stores, float4 bias loads).
//
// Three main loops share the tap / row addressing (RowInfo) and the fused epilogue below:
//   v3 (default wherever its tiles fill the CUs): persistent 8-wave blocks on 256 x 256 / 192 x 320 / 256 x 128 tiles, buffer-load
//      LDS-DMA ring across tile boundaries, two wave groups per SIMD offset by half a step around ONE barrier per 32 k - see the
//      block comment at gemm_kernel_v3.
//   v2 (small-M levels, ragged edges, batched launches, split-K): one 128 x 128 / 256 x 128 / 256 x 64 tile per block, 2-3 blocks
//      per CU; HBM/L2 -> LDS by global_load_lds_dwordx4 into an NS-deep ring of BK = 32 / 64 stages, counted s_waitcnt vmcnt(N) +
//      one raw s_barrier per stage (never vmcnt(0) inside the loop); conv zero padding / M, N tails read a 64 KiB zero page.
//   v1 (K % 32 != 0, V3D_GEMM_IMPL=1): register-staged double buffer with bounds-checked buffer loads.
// LDS rows are 64 B (128 B for BK = 64); the 16-byte chunk position of a row is XOR-swizzled on the per-lane SOURCE address (the
// LDS-DMA destination is lane-linear) so every ds_read_b128 lane group hits 16 distinct 16-byte slots.
#include <stdlib.h>

#include <map>
#include <mutex>
#include <vector>
#include <type_traits>

#include "gemm_common.h"

// Tile walk / tap order knobs (same-box A/B: tools/gemm_ab.sh, profiles/r02b_gemm_ab.txt):
//   V3D_GEMM_GROUPM   -1 heuristic (default), 0 = row-major walk, n = groups of n tile rows.  Measured (TF/s, row-major -> 4 -> 8): GEGLU projection
//                     N = 10240: 824 -> 877 -> 861, N = 5120: 714 -> 703 -> 737; temporal qkv N = 3840: 674 -> 738 -> 726; the feed-forward
//                     out-projections +3 %; convolutions (N <= 5 tile columns) +-2 %.  Heuristic: 8 for >= 96 tile rows, else 4.
//   V3D_GEMM_TAPINNER 1 = (k outer, tap inner) stage order.  It removes the 7-19x L2-miss re-reads of the K >= 640 convolutions but recomputes
//                     the per-lane row offsets every 32-k step: 25-30 % SLOWER on every convolution (1044 -> 769 TF/s at 1920 -> 640) - the
//                     loaders have no VALU to spare, the re-reads come from the Infinity Cache and are not what bounds these launches.  Off.
#ifndef V3D_GEMM_V6_DEFAULT
#define V3D_GEMM_V6_DEFAULT 1
#endif
#ifndef V3D_GEMM_V6_MIN_TILES
#define V3D_GEMM_V6_MIN_TILES 512
#endif
#ifndef V3D_GEMM_TAPINNER_DEFAULT
#define V3D_GEMM_TAPINNER_DEFAULT 0
#endif

namespace {

// 64 KiB of zeros: an invalid (padding / tail) lane of the LDS-DMA points here and can still be advanced by k0 like a
// real row pointer (K * 2 bytes <= 64 KiB is checked on the host), so the main loop has no per-step selects.
__device__ __attribute__((aligned(256))) unsigned int g_zero_page[16384] = {0};

// =====================================================================================================================
// v2: LDS-DMA ring pipeline
// =====================================================================================================================
// A stage holds KS k-slices of 32: LDS rows of 64 B (KS=1) or 128 B (KS=2).
// Chunk-position swizzle per 4-row block b = (row >> 2) & 3: {0, 2, 3, 1}.  With 64-byte rows the 256-byte LDS bank row
// holds 4 rows x 4 chunks; the 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) read rows
// {0-3,12-15} at chunk c and rows {4-11} at chunk c^1: positions {f0, f3, 1^f1, 1^f2} and {f1, f2, 1^f0, 1^f3} (and the
// same XOR 2 for the upper half-wave) are all distinct for f = {0,2,3,1} -> 16 distinct slots per group, no conflicts.
// With 128-byte rows (2 rows per bank row, 8 chunk positions) the same argument gives f(row) = (row >> 1) & 7.
template <int KS>
__device__ __forceinline__ int swz_row(int row) {
    if (KS == 1) return (0x78 >> (((row >> 2) & 3) * 2)) & 3;
    return (row >> 1) & 7;
}

template <int BM, int BN, int WGM, int WGN, int NS, int KS, int MODE, bool GEGLU>
__global__ __launch_bounds__(64 * WGM * WGN, (64 * WGM * WGN) == 256 ? 2 : 1) void gemm_kernel_v2(GP p) {
    constexpr int BK2 = 32 * KS;             // k extent of one stage
    constexpr int ROWB = BK2 * 2;            // bytes per LDS row
    constexpr int RPP = 1024 / ROWB;         // rows per 1-KiB piece (one wave-wide LDS-DMA): 16 or 8
    constexpr int CPR = ROWB / 16;           // 16-byte chunks per row: 4 or 8
    constexpr int NW = WGM * WGN;            // waves per block
    constexpr int WM = BM / WGM;             // wave tile rows (pixels)
    constexpr int WN = BN / WGN;             // wave tile cols (out channels)
    constexpr int MF = WM / 16, NF = WN / 16;
    constexpr int APW = BM / RPP / NW;       // A pieces per wave per stage
    constexpr int BPW = BN / RPP / NW;
    static_assert(APW >= 1 && BPW >= 1 && APW * NW * RPP == BM && BPW * NW * RPP == BN, "tile / wave-count mismatch");
    constexpr int PIECES = APW + BPW;
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    static_assert(PIECES * (NS - 2) < 64, "vmcnt range");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * STAGE_BYTES];   // the ONLY __shared__ object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    const int bid = xcd_remap(blockIdx.x, p.mt * p.nt);
    int tile_m, tile_n;
    tile_coords(p, bid, tile_m, tile_n);
    const long long m0 = (long long)tile_m * BM;
    const long long n0 = (long long)tile_n * BN;
    const long long z = blockIdx.y;                       // batch index, or split index of a split-K launch (sA = sW = 0 then)
    const bf16_t* __restrict__ A = p.A + z * p.sA;
    const bf16_t* __restrict__ W = p.W + z * p.sW;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);

    // ---- LDS-DMA source addressing: lane l of a piece covers row (l / CPR), LDS chunk position (l % CPR) of that row;
    //      the position holds logical k-chunk pos ^ swz(row)  ->  this lane loads that logical chunk.  The swizzle of a
    //      row only depends on its index modulo 16, and every piece starts at a multiple of RPP rows.
    const int prow = lane / CPR;
    int kchunk_a[APW], kchunk_b[BPW];
#pragma unroll
    for (int i = 0; i < APW; ++i) kchunk_a[i] = (lane % CPR) ^ swz_row<KS>((wave + NW * i) * RPP + prow);
#pragma unroll
    for (int i = 0; i < BPW; ++i) kchunk_b[i] = (lane % CPR) ^ swz_row<KS>((wave + NW * i) * RPP + prow);
    RowInfo<MODE> ri[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i) ri[i].init(p, m0 + (wave + NW * i) * RPP + prow);
    const bf16_t* arow[APW];   // source row pointer (+ k-chunk) for the current tap; invalid rows point into the zero page
    const bf16_t* brow[BPW];
    auto set_tap = [&](int tap) {
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            long long s;
            const bool ok = ri[i].tap(p, tap, s);
            arow[i] = (ok ? A + (s + p.a_row0) * p.lda : zero) + kchunk_a[i] * 8;
        }
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const long long n = n0 + (wave + NW * i) * RPP + prow;
            brow[i] = ((n < p.N) ? W + ((long long)tap * p.N + n) * p.ldw : zero) + kchunk_b[i] * 8;
        }
    };
    const int ksteps = (int)(p.K / BK2);                 // host guarantees K % BK2 == 0 for this kernel
    // split-K: this block contracts the flat (tap, k) step range [s0, s1) only
    const int total_steps = ksteps * ntaps<MODE>();
    const int s0 = p.split_n > 1 ? (int)((long long)total_steps * blockIdx.y / p.split_n) : 0;
    const int s1 = p.split_n > 1 ? (int)((long long)total_steps * (blockIdx.y + 1) / p.split_n) : total_steps;
    const int nsteps = s1 - s0;
    const bool tap_inner = ntaps<MODE>() > 1 && p.tap_inner;
    int ld_tap = tap_inner ? s0 % ntaps<MODE>() : s0 / ksteps;
    int ld_k0 = tap_inner ? (s0 / ntaps<MODE>()) * BK2 : (s0 - ld_tap * ksteps) * BK2;
    set_tap(ld_tap);
    // steps issued past the end of the contraction (ring tail) re-read valid rows of the last tap: harmless dummies that
    // keep the per-wave DMA count per stage constant for the counted vmcnt waits
    auto issue = [&](int stage) {
        unsigned char* sbase = lds + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < APW; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(arow[i] + ld_k0), (__attribute__((address_space(3))) void*)(sbase + (wave + NW * i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BPW; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(brow[i] + ld_k0), (__attribute__((address_space(3))) void*)(sbase + BM * ROWB + (wave + NW * i) * 1024), 16, 0, 0);
        if (tap_inner) {   // (k outer, tap inner): the nine taps of a k-slice follow each other while their rows are still in L2
            if (++ld_tap >= ntaps<MODE>()) {
                ld_tap = 0;
                ld_k0 += BK2;
                if (ld_k0 >= p.K) ld_k0 = 0;   // (ring-tail dummies)
            }
            set_tap(ld_tap);
            return;
        }
        ld_k0 += BK2;
        if (ld_k0 >= p.K) {
            ld_k0 = 0;
            if (ntaps<MODE>() > 1 && ++ld_tap < ntaps<MODE>()) set_tap(ld_tap);
        }
    };

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row (lane & 15) of a 16-row fragment, logical chunk kk*4 + (lane >> 4), stored at chunk ^ swz
    int frag_off[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) frag_off[kk] = (lane & 15) * ROWB + (((kk * 4 + (lane >> 4)) ^ swz_row<KS>(lane & 15)) * 16);
    const int a_base = wm * WM * ROWB;
    const int b_base = BM * ROWB + wn * WN * ROWB;

#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s);

    // main-loop waves outrank co-resident blocks that are in their (VALU-dense) epilogue: without it the two do not overlap -
    // time(K) = time(epilogue only) + time(main loop only) on the GEGLU projections (tools/gemm_floor.py)
    if (!V3D_ABL(p, 64)) __builtin_amdgcn_s_setprio(2);
    for (int t = 0; t < nsteps; ++t) {
        // this wave's pieces of stage t have landed when at most PIECES*(NS-2) newer DMA ops are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (NS - 2)) : "memory");
        __builtin_amdgcn_s_barrier();   // everyone's pieces landed; everyone finished reading stage (t-1) % NS
        asm volatile("" ::: "memory");
        issue((t + NS - 1) % NS);       // refill the stage consumed in the previous iteration
        const unsigned char* sb = lds + (t % NS) * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            bf16x8 xf[MF], wf[NF];
#pragma unroll
            for (int i = 0; i < MF; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(sb + a_base + frag_off[kk] + i * 16 * ROWB);
#pragma unroll
            for (int j = 0; j < NF; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + b_base + frag_off[kk] + j * 16 * ROWB);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the (dummy) tail DMAs before the ring is reused / released
    __builtin_amdgcn_s_barrier();                      // every wave has finished reading the last stage
    asm volatile("" ::: "memory");
    constexpr int NFO_ = GEGLU ? NF / 2 : NF;
    constexpr int STAGE_REGION = MF * 16 * (NFO_ * 32 + 16);
    constexpr bool CAN_STAGE = NW * STAGE_REGION <= NS * STAGE_BYTES;   // else: direct MFMA-layout stores
    epilogue<MF, NF, GEGLU, CAN_STAGE>(p, acc, m0 + wm * WM, n0 + wn * WN, z, lane, lds + (CAN_STAGE ? wave * STAGE_REGION : 0));
}

// =====================================================================================================================
// v1: register-staged double buffer (buffer loads with hardware bounds checking)
// =====================================================================================================================
constexpr int BK = 64;
constexpr int LROW = BK + 8;  // padded LDS row (bf16 elements) = 144 B, keeps 16-B alignment

template <int BM, int BN, int MODE, bool GEGLU>
__global__ __launch_bounds__(256, 2) void gemm_kernel_v1(GP p) {
    constexpr int AI = BM / 32;
    constexpr int BI = BN / 32;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int MF = WM / 16, NF = WN / 16;
    __shared__ __attribute__((aligned(16))) bf16_t sA[2][BM * LROW];
    __shared__ __attribute__((aligned(16))) bf16_t sB[2][BN * LROW];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = xcd_remap(blockIdx.x, p.mt * p.nt);
    int tile_m, tile_n;
    tile_coords(p, bid, tile_m, tile_n);
    const long long m0 = (long long)tile_m * BM;
    const long long n0 = (long long)tile_n * BN;
    const long long z = blockIdx.y;
    const bufrsrc_t rsA = make_rsrc(p.A + z * p.sA, p.a_bytes);
    const bufrsrc_t rsW = make_rsrc(p.W + z * p.sW, p.w_bytes);
    const int kc = tid & 7;
    const int r0 = tid >> 3;
    RowInfo<MODE> ri[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) ri[i].init(p, m0 + r0 + 32 * i);
    u32x4 ra[AI], rb[BI];
    unsigned aoff[AI], boff[BI];
    const int ksteps = (int)((p.K + BK - 1) / BK);
    const int nsteps = ksteps * ntaps<MODE>();
    int ld_tap = 0, ld_k0 = 0;
    auto set_tap = [&](int tap) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            long long srow;
            const bool ok = ri[i].tap(p, tap, srow);
            aoff[i] = ok ? (unsigned)(((srow + p.a_row0) * p.lda + kc * 8) * 2) : kInvalid;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const long long n = n0 + r0 + 32 * i;
            boff[i] = (n < p.N) ? (unsigned)((((long long)tap * p.N + n) * p.ldw + kc * 8) * 2) : kInvalid;
        }
    };
    set_tap(0);
    auto gload = [&]() {
        const bool kok = (ld_k0 + kc * 8) < p.K;
        const unsigned kb = (unsigned)ld_k0 * 2u;
#pragma unroll
        for (int i = 0; i < AI; ++i) ra[i] = buf_load16(rsA, (kok && aoff[i] != kInvalid) ? aoff[i] + kb : kInvalid);
#pragma unroll
        for (int i = 0; i < BI; ++i) rb[i] = buf_load16(rsW, (kok && boff[i] != kInvalid) ? boff[i] + kb : kInvalid);
        ld_k0 += BK;
        if (ld_k0 >= p.K) {
            ld_k0 = 0;
            ++ld_tap;
            if (ntaps<MODE>() > 1 && ld_tap < ntaps<MODE>()) set_tap(ld_tap);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<u32x4*>(&sA[buf][(r0 + 32 * i) * LROW + kc * 8]) = ra[i];
#pragma unroll
        for (int i = 0; i < BI; ++i) *reinterpret_cast<u32x4*>(&sB[buf][(r0 + 32 * i) * LROW + kc * 8]) = rb[i];
    };
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    gload();
    lstore(0);
    __syncthreads();
    const int frow = lane & 15;
    const int fk = (lane >> 4) * 8;
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        if (step + 1 < nsteps) gload();
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 xf[MF], wf[NF];
#pragma unroll
            for (int i = 0; i < MF; ++i)
                xf[i] = *reinterpret_cast<const bf16x8*>(&sA[buf][(wm * WM + i * 16 + frow) * LROW + kk * 32 + fk]);
#pragma unroll
            for (int j = 0; j < NF; ++j)
                wf[j] = *reinterpret_cast<const bf16x8*>(&sB[buf][(wn * WN + j * 16 + frow) * LROW + kk * 32 + fk]);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
        if (step + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }
    epilogue<MF, NF, GEGLU, false>(p, acc, m0 + wm * WM, n0 + wn * WN, z, lane, nullptr);
}

// =====================================================================================================================
// v3: persistent 8-wave ping-pong pipeline on big tiles (256 x 256, or 192 x 320 for the N = 320 family)
// =====================================================================================================================
// Why: a 128 x 128 tile needs (128+128)*2 B of LDS fill per k for 2*128*128 flop = 64 B/clk/CU at the MFMA peak, which is
// the whole L1/TA fill rate of a CU -- the v2 kernels sit at ~43 % MFMA busy with both resident waves of a SIMD parked on
// vmcnt at the same time (profiles/r01_gemm_ablation.txt + PMC).  A 256 x 256 tile halves the fill per flop; it leaves one
// workgroup per CU, so the overlap that v2 got from independent co-resident blocks has to be built into the block:
//   * 8 waves = 2 groups of 4 (waves w and w+4 share a SIMD).  Every step of 32 k has two slots separated by s_barrier;
//     in a slot one group runs its 32 MFMAs from registers while the other group issues LDS-DMA for 3 stages ahead and
//     ds_reads its fragments of the next stage, then they swap (group 1 runs one slot behind group 0).
//   * persistent: a block walks tiles b, b+G, b+2G ... (XCD-aware order); the (tile, tap, k) stage sequence is flat, so
//     the ring keeps prefetching across tile boundaries and a group's epilogue overlaps the other group's MFMA slot.
//   * counted s_waitcnt vmcnt(8): two younger stages (4 DMA ops each per wave) may stay in flight; epilogue stores in
//     flight only make the count conservative (loads return in order among themselves).

// slot-level timeline of the v3 loop (V3D_GEMM_ABLATE bit 8, LINEAR only): [group][step 32..63][stamp] s_memtime ticks
__device__ unsigned long long g_v3_dbg[2 * 32 * 8];

template <int BM, int BN, int WGM, int WGN, int MODE, bool GEGLU, int EMF, bool DBG = false, int NS = 4, int SPAD = 16, bool GN = false>
__global__ __launch_bounds__(512, NS == 3 ? 4 : 2) void gemm_kernel_v3(GP p, int ntiles) {   // (HIP: 2nd arg = min waves per SIMD)
    // NS = 4: one block per CU (128 KiB ring).  NS = 3 with a 256 x 128 tile: 72 KiB ring + 8 KiB staging = 80 KiB -> TWO blocks per
    // CU (128 VGPRs per wave), so one block's VALU-bound epilogue (GEGLU) overlaps the other block's main loop.
    constexpr int ROWB = 64, NW = 8;
    static_assert(WGM * WGN == NW, "8 waves");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MF = WM / 16, NF = WN / 16;
    static_assert(MF % EMF == 0, "epilogue chunking");
    constexpr int NPIECE = (BM + BN) / 16;          // 1-KiB pieces (16 rows x 64 B) per stage
    constexpr int PPW = NPIECE / NW;                // pieces per wave per stage
    static_assert(PPW * NW == NPIECE && PPW * 2 < 64, "tile / wave-count mismatch");
    constexpr int APIECES = BM / 16;
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int NFO = GEGLU ? NF / 2 : NF;
    constexpr int EPI_REGION = EMF * 16 * (NFO * 32 + SPAD);
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * STAGE_BYTES + NW * EPI_REGION + (DBG ? 4096 : 0)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(lds + NS * STAGE_BYTES + NW * EPI_REGION);
    const bool dbg_on = DBG && blockIdx.x == 0 && (wave & 3) == 0;
    auto stamp = [&](int s_, int k) __attribute__((always_inline)) {
        if (DBG && dbg_on && s_ >= 32 && s_ < 64 && lane == 0) dbg[(grp * 32 + (s_ - 32)) * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    const int wm = wave / WGN, wn = wave % WGN;
    const bf16_t* __restrict__ A = p.A;
    const bf16_t* __restrict__ W = p.W;
    unsigned char* estage = lds + NS * STAGE_BYTES + wave * EPI_REGION;

    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    auto tile_origin = [&](int it, long long& m0, long long& n0) __attribute__((always_inline)) {
        const int id = xcd_remap((int)blockIdx.x + it * G, ntiles);
        int tm, tn;
        tile_coords(p, id, tm, tn);
        n0 = (long long)tn * BN;
        m0 = p.m_off + (long long)tm * BM;
    };

    // ---- loader state (runs up to 3 stages ahead of the consumer, across tile boundaries).  Raw buffer loads straight to
    //      LDS: per-lane byte offset (row, k-chunk) in a VGPR that only changes per tile / tap, the k position in an SGPR
    //      soffset -> no vector ALU work per step; padding rows / tails use an out-of-range offset (the load writes zeros).
    const bufrsrc_t rsA = make_rsrc(A, p.a_bytes);
    const bufrsrc_t rsW = make_rsrc(W, p.w_bytes);
    const int prow = lane >> 2;
    const unsigned kchunk_b = (unsigned)(((lane & 3) ^ swz_row<1>(prow)) * 16);   // every piece starts at a multiple of 16 rows
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
        if V3D_ABL(p, 4) return;
        const int q = wave + NW * i;
        // experiment (bit 2048): activation pieces issued out of range - same DMA op count, zeros instead of an L2 fetch: what the L2->LDS bytes
        // of the activation operand cost (a lower bound of what an LDS-resident halo tile would save a 3x3 convolution)
        const unsigned vo = (V3D_ABL(p, 2048) && q < APIECES) ? kInvalid : voff[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(q < APIECES ? rsA : rsW, (__attribute__((address_space(3))) void*)(lds + stage * STAGE_BYTES + q * 1024), 16, (int)vo, so, 0, 0);
    };
    auto issue_advance = [&]() __attribute__((always_inline)) {
        if (ntaps<MODE>() > 1 && p.tap_inner) {   // (k outer, tap inner), see the v2 loader
            if (++ld_tap < ntaps<MODE>()) {
                set_tap(ld_tap);
                return;
            }
            ld_tap = 0;
            ld_k0 += 32;
            if (ld_k0 >= (int)p.K) {
                ld_k0 = 0;
                if (++ld_it < my_tiles) {
                    set_tile(ld_it);
                    return;
                }
            }
            set_tap(0);
            return;
        }
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
    auto issue = [&](int stage) __attribute__((always_inline)) {
        // the k offset is wave-uniform, but in the multi-tap modes the compiler takes it for lane-varying, keeps it in a vector register and
        // wraps EVERY LDS-DMA issue of the main loop into a readfirstlane "waterfall" loop (rounds 1-2 shipped the 3x3 and temporal
        // convolutions that way; tools/check_loop_scratch.py flags it now).  One explicit readfirstlane per step: conv3x3 launches -5...-9 %.
        const int so = __builtin_amdgcn_readfirstlane(ld_k0 * 2);
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(stage, i, so);
        issue_advance();
    };

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frag_off = (lane & 15) * ROWB + (((lane >> 4) ^ swz_row<1>(lane & 15)) * 16);
    const int a_base = wm * WM * ROWB;
    const int b_base = BM * ROWB + wn * WN * ROWB;
    bf16x8 xf[MF], wf[NF];

    const int nsteps = (int)(p.K / 32) * ntaps<MODE>();

#pragma unroll
    for (int st = 0; st < NS - 1; ++st) issue(st);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NS - 2)) : "memory");
    __builtin_amdgcn_s_barrier();   // B_0: stage 0 landed
    __builtin_amdgcn_sched_barrier(0);
    // One barrier B_{s+1} per step ("stage s+1 landed, stage s-... buffers reusable").  Group 0 passes it AFTER its MFMAs of
    // step s, group 1 BEFORE them: between two barriers group 0 runs {read s, MFMA s} while group 1 runs {MFMA s-1, read s},
    // so one group's LDS reads / DMA issue always face the other group's MFMAs on the same SIMD.
    int s = 0;        // flat step counter
    int rd = 0;       // ring slot of step s; the refill target (step s + NS - 1) is the slot before it
    for (int it = 0; it < my_tiles; ++it) {
        for (int kt = 0; kt < nsteps; ++kt, ++s) {
            stamp(s, 0);
            {
                const unsigned char* sb = lds + rd * STAGE_BYTES;
#pragma unroll
                for (int i = 0; i < MF; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(sb + a_base + frag_off + i * 16 * ROWB);
#pragma unroll
                for (int j = 0; j < NF; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + b_base + frag_off + j * 16 * ROWB);
            }
            // refill the ring 3 stages ahead (the buffer of step s-1: its last reader, group 1, finished before B_s) while the
            // fragment reads are in flight
            issue(rd == 0 ? NS - 1 : rd - 1);
            rd = (rd + 1 == NS) ? 0 : rd + 1;
            stamp(s, 1);
            // group 1 must have its fragments in registers before it passes the barrier (the slot is refilled after it);
            // group 0 goes straight into its MFMAs and lets the compiler's counted lgkmcnt waits release them fragment by fragment
            if (grp == 1 || V3D_ABL(p, 512)) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            stamp(s, 2);
            if (grp == 1) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NS - 2)) : "memory");   // own pieces of stage s+1 landed
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            stamp(s, 3);
            if (!V3D_ABL(p, 256)) __builtin_amdgcn_s_setprio(1);
            if (!V3D_ABL(p, 2)) {
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
            }
            if (!V3D_ABL(p, 256)) __builtin_amdgcn_s_setprio(0);
            stamp(s, 4);
            __builtin_amdgcn_sched_barrier(0);
            if (grp == 0) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NS - 2)) : "memory");   // own pieces of stage s+1 landed
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            stamp(s, 5);
        }
        // ---------------- tile finished for this group: retire it while the other group keeps the MFMA pipe busy
        long long e_m0, e_n0;
        tile_origin(it, e_m0, e_n0);
        if constexpr (!GEGLU && EMF == 1 && SPAD == 16) {
            // hand-managed epilogue (gemm_common.h e4_*): asm loads one fragment ahead of their use, one wait per 16-row fragment.  The host only
            // launches these kernels when e4_ok holds (whole wave tiles inside or outside the output, aligned operands, bf16 output).
            const long long mw0 = e_m0 + wm * WM, nw0 = e_n0 + wn * WN;
            if (mw0 < p.M && nw0 < p.N) {
                int lane_e = lane;
                asm volatile("" : "+v"(lane_e));          // (keeps what the epilogue derives from the lane id out of the loop-invariant set: no spills)
                auto rowfn = [&](int f) __attribute__((always_inline)) -> long long { return mw0 + f * 16; };
                E4GnRun<WM, MF> run;
                if constexpr (GN) run.init(mw0, p.gn_rps);
                auto flushfn = [&](int f, long long, unsigned& slot, unsigned& sid) __attribute__((always_inline)) -> bool { return run.step(f, slot, sid); };
                e4_retire_tile<MF, NF, GN, (MODE == V3D_GEMM_LINEAR ? E4_DEPTH_LINEAR : E4_DEPTH)>(p, acc, nw0, lane_e, estage, rowfn, flushfn);
            }
        } else
        {
            const long long mw0 = e_m0 + wm * WM, nw0 = e_n0 + wn * WN;
            const bool inside = mw0 < p.M && nw0 < p.N;
            const bool res_pre = p.res1 && !p.out_fp32 && (p.ldr1 % 8 == 0) && (reinterpret_cast<uintptr_t>(p.res1) % 16 == 0) && inside;
            u32x4 r0 = {0u, 0u, 0u, 0u}, r1 = r0, r2 = r0;
            if (res_pre) {
                r0 = load_res_piece<0, EMF, NF, GEGLU>(p, mw0, nw0, lane);
                r1 = load_res_piece<1, EMF, NF, GEGLU>(p, mw0, nw0, lane);
                r2 = load_res_piece<2, EMF, NF, GEGLU>(p, mw0, nw0, lane);
            }
            float4 bv[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j)
                bv[j] = (p.bias && inside) ? *reinterpret_cast<const float4*>(p.bias + (int)nw0 + (lane >> 4) * 4 + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            GnAcc<GN ? NF : 1> gn;
            long long sid0 = 0;
            unsigned rem0 = 0;
            if constexpr (GN) {
                gn_zero(gn);
                sid0 = mw0 / p.gn_rps;
                rem0 = (unsigned)(mw0 - sid0 * p.gn_rps);
            }
            if (!GN || inside) v3_retire_chunks<0, MF / EMF, EMF, NF, GEGLU, SPAD, GN>(p, acc, mw0, nw0, lane, estage, r0, r1, r2, res_pre, bv, gn, sid0, rem0);
        }
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (DBG && dbg_on && lane < 32)
        for (int k = 0; k < 8; ++k) g_v3_dbg[(grp * 32 + lane) * 8 + k] = dbg[(grp * 32 + lane) * 8 + k];
}

// =====================================================================================================================
// v6: persistent 4-wave blocks, TWO per CU, for the HBM-bound linears (K <= 640: to_out / proj_in / proj_out / skip of the 64 x 64 and 32 x 32 levels)
// =====================================================================================================================
// Why (round 6, tools/v3_timeline_k320.py -> profiles/r06_timeline_k320_*.txt): on M = 147456, N = K = 320 the v3 block spends 25 k cycles in the ten
// steps of a tile (8 k of them waiting for rows from HBM) and then 14 k (bias only) to 33 k (bias + vector + residual) cycles in the tile's epilogue,
// with NO load of the next tile issued meanwhile beyond the three ring stages - every CU of the chip alternates between a read phase and a
// (read +) write phase, each bound by the bytes one block keeps in flight, and the launch moves 3.4 TB/s.  The work of a CU has to be two
// independent streams whose phases overlap.  The wave groups of a v3 block cannot be that (one s_barrier per workgroup), so here the groups
// are BLOCKS: 256 threads, 192 x 160 tiles (wave tile 96 x 80 as v3's 192 x 320: same fragments, same hand-managed epilogue), a 3-stage
// ring of 22-KiB stages -> 79 KiB of LDS and 256 registers per wave: two blocks per CU, each with its own ring, barrier and tile sequence.
// While one block retires a tile the other runs its main loop: its LDS-DMA stream and the first block's residual loads / row stores share
// the memory system instead of taking turns.  The two N-halves of a row tile are consecutive tile ids (row-major walk) = blocks 8 apart on one
// XCD: the second reader of the 192 activation rows finds them in that XCD's L2.
// One wave per SIMD and block, so a wave runs {fragment reads, DMA issue, MFMAs, counted wait, barrier} in sequence and the co-resident block's
// wave fills the matrix pipe meanwhile - unsynchronised, which is fine for launches that are bound by HBM, not by the matrix pipe.
template <int BM, int BN, int NS, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_kernel_v6(GP p, int ntiles) {
    constexpr int ROWB = 64, NW = 4, WGN = 2;
    constexpr int WM = BM / 2, WN = BN / WGN, MF = WM / 16, NF = WN / 16;
    static_assert(NF == 5 || NF == 4, "the hand-managed epilogue retires 80- or 64-channel wave tiles");
    constexpr int NPIECE = (BM + BN) / 16, PPW = (NPIECE + NW - 1) / NW, APIECES = BM / 16, ARI = (APIECES + NW - 1) / NW;
    static_assert(PPW * (NS - 2) < 64, "vmcnt range");
    constexpr int STAGE_BYTES = NPIECE * 1024;
    constexpr int DUMMY_OFF = NS * STAGE_BYTES, NDUMMY = PPW * NW - NPIECE;        // pieces past the stage keep every wave's DMA count per stage equal
    constexpr int EPI_OFF = DUMMY_OFF + NDUMMY * 1024, EPI_REGION = 16 * (NF * 32 + 16);
    __shared__ __attribute__((aligned(1024))) unsigned char lds[EPI_OFF + NW * EPI_REGION];      // the ONLY __shared__ object
    static_assert(sizeof(lds) <= 80 * 1024, "two blocks per CU");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    unsigned char* estage = lds + EPI_OFF + wave * EPI_REGION;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    auto tile_origin = [&](int it, long long& m0, long long& n0) __attribute__((always_inline)) {
        const int id = xcd_remap((int)blockIdx.x + it * G, ntiles);
        int tm, tn;
        tile_coords(p, id, tm, tn);
        n0 = (long long)tn * BN;
        m0 = p.m_off + (long long)tm * BM;
    };
    // ---- loader (as v3): piece q = wave + 4 i: q < APIECES activation rows, q < NPIECE weight rows, else a dummy into its own KiB
    const bufrsrc_t rsA = make_rsrc(p.A, p.a_bytes);
    const bufrsrc_t rsW = make_rsrc(p.W, p.w_bytes);
    const int prow = lane >> 2;
    const unsigned kchunk_b = (unsigned)(((lane & 3) ^ swz_row<1>(prow)) * 16);
    RowInfo<MODE> ri[ARI];
    unsigned voff[PPW];
    long long ld_n0 = 0;
    int ld_it = 0, ld_tap = 0, ld_k0 = 0;
    auto set_tap = [&](int tap) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave + NW * i;
            if (i < ARI && q < APIECES) {
                long long s_;
                const bool ok = ri[i < ARI ? i : 0].tap(p, tap, s_);
                voff[i] = ok ? (unsigned)((s_ + p.a_row0) * p.lda * 2) + kchunk_b : kInvalid;
            } else {
                const long long n = ld_n0 + (q - APIECES) * 16 + prow;
                voff[i] = (q < NPIECE && n < p.N) ? (unsigned)(((long long)tap * p.N + n) * p.ldw * 2) + kchunk_b : kInvalid;
            }
        }
    };
    auto set_tile = [&](int it) __attribute__((always_inline)) {
        long long m0;
        tile_origin(it, m0, ld_n0);
#pragma unroll
        for (int i = 0; i < ARI; ++i) {
            const int q = wave + NW * i;
            if (q < APIECES) ri[i].init(p, m0 + q * 16 + prow);
        }
        set_tap(0);
    };
    set_tile(0);
    auto issue = [&](int stage) __attribute__((always_inline)) {
        const int so = __builtin_amdgcn_readfirstlane(ld_k0 * 2);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave + NW * i;
            const int dst = q < NPIECE ? stage * STAGE_BYTES + q * 1024 : DUMMY_OFF + (q - NPIECE) * 1024;
            if V3D_ABL(p, 4) continue;
            const unsigned vo = (V3D_ABL(p, 2048) && q < APIECES) ? kInvalid : voff[i];      // (experiments: no fetch of the activation rows)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(q < APIECES ? rsA : rsW, (__attribute__((address_space(3))) void*)(lds + dst), 16, (int)vo, so, 0, 0);
        }
        ld_k0 += 32;
        if (ld_k0 >= (int)p.K) {
            ld_k0 = 0;
            if (++ld_tap < ntaps<MODE>()) {
                set_tap(ld_tap);
            } else {
                ld_tap = 0;
                if (++ld_it < my_tiles) set_tile(ld_it);      // past the last tile: harmless re-reads keep the DMA count constant
                else if (ntaps<MODE>() > 1) set_tap(0);
            }
        }
    };
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frag_off = (lane & 15) * ROWB + (((lane >> 4) ^ swz_row<1>(lane & 15)) * 16);
    const int a_base = wm * WM * ROWB;
    const int b_base = BM * ROWB + wn * WN * ROWB;
    bf16x8 xf[MF], wf[NF];
    const int nsteps = (int)(p.K / 32) * ntaps<MODE>();
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) issue(st);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NS - 2)) : "memory");
    __builtin_amdgcn_s_barrier();   // stage 0 landed
    __builtin_amdgcn_sched_barrier(0);
    int rd = 0;       // ring slot of the current step; the refill target (step s + NS - 1) is the slot before it, whose readers passed the last barrier
    for (int it = 0; it < my_tiles; ++it) {
        for (int kt = 0; kt < nsteps; ++kt) {
            {
                const unsigned char* sb = lds + rd * STAGE_BYTES;
#pragma unroll
                for (int i = 0; i < MF; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(sb + a_base + frag_off + i * 16 * ROWB);
#pragma unroll
                for (int j = 0; j < NF; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + b_base + frag_off + j * 16 * ROWB);
            }
            issue(rd == 0 ? NS - 1 : rd - 1);
            rd = (rd + 1 == NS) ? 0 : rd + 1;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    if (!V3D_ABL(p, 2)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NS - 2)) : "memory");   // own pieces of the next stage landed
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        long long e_m0, e_n0;
        tile_origin(it, e_m0, e_n0);
        const long long mw0 = e_m0 + wm * WM, nw0 = e_n0 + wn * WN;
        if (mw0 < p.M && nw0 < p.N) {
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));          // (keeps what the epilogue derives from the lane id out of the loop-invariant set: no spills)
            auto rowfn = [&](int f) __attribute__((always_inline)) -> long long { return mw0 + f * 16; };
            auto flushfn = [&](int, long long, unsigned&, unsigned&) __attribute__((always_inline)) -> bool { return false; };
            e4_retire_tile<MF, NF, false, (MODE == V3D_GEMM_LINEAR ? E4_DEPTH_V6 : 1)>(p, acc, nw0, lane_e, estage, rowfn, flushfn);
        }
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- split-K finalize: out = epilogue(sum over splits of ws) element-wise (4 consecutive channels per thread) ---------------------
__global__ __launch_bounds__(256) void splitk_finalize_kernel(GP p) {
    const long long n4 = p.N / 4;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.M * n4) return;
    const long long m = idx / n4;
    const int n = (int)(idx - m * n4) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sp = 0; sp < p.split_n; ++sp) {
        const float4 w = *reinterpret_cast<const float4*>(p.ws + ((long long)sp * p.M + m) * p.N + n);
        a.x += w.x; a.y += w.y; a.z += w.z; a.w += w.w;
    }
    if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (p.add) {
        const float4 b = *reinterpret_cast<const float4*>(p.add + (m / p.add_rpg) * p.add_ld + n);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float ca = p.c_acc, c1 = p.c_res1, c2 = p.c_res2;
    if (p.coef) {
        const float* cf = p.coef + (m / p.coef_rpg) * 3;
        ca = cf[0]; c1 = cf[1]; c2 = cf[2];
    }
    float o[4] = {ca * a.x, ca * a.y, ca * a.z, ca * a.w};
    if (p.res1) {
        const uint2 rr = *reinterpret_cast<const uint2*>(p.res1 + m * p.ldr1 + n);
        o[0] += c1 * bflo(rr.x); o[1] += c1 * bfhi(rr.x); o[2] += c1 * bflo(rr.y); o[3] += c1 * bfhi(rr.y);
    }
    if (p.res2) {
        const uint2 rr = *reinterpret_cast<const uint2*>(p.res2 + m * p.ldr2 + n);
        o[0] += c2 * bflo(rr.x); o[1] += c2 * bfhi(rr.x); o[2] += c2 * bflo(rr.y); o[3] += c2 * bfhi(rr.y);
    }
    if (p.out_fp32)
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + m * p.ldo + n) = make_float4(o[0], o[1], o[2], o[3]);
    else
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + m * p.ldo + n) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
}

// library-owned fp32 workspaces of the split-K path: ONE PER STREAM (two streams splitting at the same time must not share partial
// sums), grow-only, and a retired (outgrown) slab is never freed: a HIP graph captured earlier may still replay launches that
// point into it.  Growth is geometric, so the retired slabs add up to less than twice the live one.  A request that would have to
// allocate while its stream is being captured returns nullptr (hipMalloc is illegal in a capture) and the caller falls back to
// the unsplit launch.  Partials are written with plain stores, one slab per split: fp32 atomics into one slab were tried first and
// made the launch 1.4-2.5x slower than not splitting at all (8.8 M atomics per 8x8 conv).
float* splitk_workspace(size_t floats, hipStream_t st) {
    struct Slab { float* ptr; size_t cap; };
    static std::mutex mu;
    static std::map<hipStream_t, Slab> slabs;
    std::lock_guard<std::mutex> lock(mu);
    Slab& s = slabs[st];
    if (floats > s.cap) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
        const size_t want = floats + floats / 2;
        float* fresh = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&fresh), want * sizeof(float)) != hipSuccess) return nullptr;
        s.ptr = fresh;      // the old slab stays allocated (see above)
        s.cap = want;
    }
    return s.ptr;
}

// ---- launch record (tests / tools / bench.py --shard-sim): what the last v3d_gemm of this thread actually launched --------------------------------
// family (1 = v1, 2 = v2, 3 = v3 persistent, 5 = LDS-haloed (conv.hip)), tile, tile count, co-resident blocks per CU (the runtime's occupancy
// answer for that kernel), split-K ways, stream-K tail.  Read back through v3d_debug_last_gemm_launch.
thread_local V3dLaunchInfo g_last_launch = {0, 0, 0, 0, 0, 0, 0};
template <typename K>
int v3d_occupancy(K kernel, int threads) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, 0) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = 1;
    }
    return n;
}
void v3d_note_launch_splitk(int ways) { g_last_launch.splitk = ways; }
#define V3D_LAUNCH(FAM, BM_, BN_, TILES, KERNEL, GRID, THREADS, ST, ...)                               \
    do {                                                                                             \
        static const int occ_ = v3d_occupancy(KERNEL, THREADS);                                      \
        v3d_note_launch(FAM, BM_, BN_, (long long)(TILES), occ_, 0);                                 \
        hipLaunchKernelGGL(KERNEL, GRID, dim3(THREADS), 0, ST, __VA_ARGS__);                         \
    } while (0)

int impl_choice() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("V3D_GEMM_IMPL");
        v = e ? atoi(e) : 0;   // 0 = heuristic, 1 = v1 only, 2 = v1/v2 only, 3 = v3 wherever it is legal
    }
    return v;
}
// tuning knob for A/B sweeps (tools/gemm_sweep.py): forces one v2 tile / pipeline configuration (-1 = heuristic)
int cfg_choice() {
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("V3D_GEMM_CFG");
        v = e ? atoi(e) : -1;
    }
    return v;
}

int splitk_choice() {
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("V3D_GEMM_SPLITK");   // 0 = never, N = force N-way where legal, unset = heuristic
        v = e ? atoi(e) : -1;
    }
    return v;
}

template <int BM, int BN, int MODE, bool GEGLU>
int launch(const GP& p0, int batch, hipStream_t st) {
    GP p = p0;
    p.mt = (int)((p.M + BM - 1) / BM);
    p.nt = (int)((p.N + BN - 1) / BN);
    dim3 grid((unsigned)(p.mt * p.nt), (unsigned)batch, 1);
    // split-K: the 8x8 level has 180 tiles of 128 x 128 for 256 CUs and contractions of 160-720 steps, so a launch lasts as long as
    // ONE tile (conv 8x8 1280->1280: 153 us, 444 TF/s).  Splitting the flat (tap, k) range 2-4 ways fills the CUs; the partial sums
    // meet in an fp32 workspace (atomics) and a small finalize kernel applies the epilogue.
    if constexpr (!GEGLU && BM == 128) {
        const long long tiles = (long long)p.mt * p.nt, cus = v3d_num_cus();
        const long long steps = (long long)ntaps<MODE>() * (p.K / 32);
        int want = splitk_choice();
        if (want < 0) want = (tiles * 4 <= cus * 3 && steps >= 96) ? (int)((3 * cus + tiles - 1) / tiles) : 1;   // 8x8 conv: 546 / 693 / 617 / 699 TF/s at 1 / 2 / 3 / 4
        if (want > 4) want = 4;
        if (want > steps / 32) want = (int)(steps / 32);
        if (want > 1 && batch == 1 && impl_choice() != 1 && p.K % 64 == 0 && p.K * 2 <= 65536 && p.N % 4 == 0 && p.ldo % 4 == 0 &&
            (!p.res1 || (p.ldr1 % 4 == 0 && reinterpret_cast<uintptr_t>(p.res1) % 8 == 0)) &&
            (!p.res2 || (p.ldr2 % 4 == 0 && reinterpret_cast<uintptr_t>(p.res2) % 8 == 0)) &&
            (!p.add || (p.add_ld % 4 == 0 && reinterpret_cast<uintptr_t>(p.add) % 16 == 0)) &&
            reinterpret_cast<uintptr_t>(p.out) % 16 == 0) {
            float* ws = splitk_workspace((size_t)want * (size_t)p.M * (size_t)p.N, st);
            if (ws) {
                GP q = p;                 // the split launch: plain fp32 partial sums, slab `split`
                q.split_n = want;
                q.out = ws; q.out_fp32 = 1; q.ldo = p.N; q.sO = p.M * p.N; q.sA = 0; q.sW = 0;
                q.bias = nullptr; q.add = nullptr; q.res1 = nullptr; q.res2 = nullptr; q.coef = nullptr;
                q.c_acc = 1.f; q.c_res1 = 0.f; q.c_res2 = 0.f;
                dim3 g2((unsigned)(p.mt * p.nt), (unsigned)want, 1);
                V3D_LAUNCH(2, BM, BN, (long long)g2.x * g2.y, (gemm_kernel_v2<BM, BN, 2, 2, 2, 2, MODE, GEGLU>), g2, 256, st, q);
                p.split_n = want;
                p.ws = ws;
                const long long n4 = p.M * (p.N / 4);
                v3d_note_launch_splitk(want);
                hipLaunchKernelGGL(splitk_finalize_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, p);
                return v3d_check_launch("v3d_gemm(split-K)");
            }
        }
    }
    if (impl_choice() == 1 || p.K % 32 != 0 || p.K * 2 > 65536) {   // (impl 0 / 2 / 3 all land here for v2-class shapes)
        // v1 also serves ragged contractions (K % 32 != 0: the 8-channel input conv, odd test shapes)
        V3D_LAUNCH(1, BM, BN, (long long)grid.x * grid.y, (gemm_kernel_v1<BM, BN, MODE, GEGLU>), grid, 256, st, p);
        return v3d_check_launch("v3d_gemm");
    }
    int cfg = cfg_choice();
    if (cfg < 0) {
        // measured on MI355X (profiles/r01_gemm_sweep.txt): short contractions (<= 40 stages of 32: every K <= 1280 linear,
        // the 128-channel VAE convs) are bound by per-tile fill/drain bubbles -> more co-resident blocks (3-deep ring of
        // BK 32, 48 KiB LDS, 3 blocks/CU) wins; long ones prefer fewer barriers per MFMA (2 stages of BK 64).
        const long long stages32 = (long long)ntaps<MODE>() * ((p.K + 31) / 32);
        cfg = stages32 <= 40 ? 2 : 1;
        // 64-wide N tiles (N = 320 and friends): a 256 x 64 tile with the 4 waves stacked along M doubles the MFMAs per
        // barrier and halves the weight re-reads (lin_L0_320x320 433 -> 487, convt_L0_320 568 -> 734 TF/s)
        if (BN == 64) cfg = (p.K % 64 == 0) ? 9 : 8;
        // wide GEGLU projections at K >= 640 and the 128-channel VAE convs: 256 x 128 tile on 4 waves (wave tile 128 x 64,
        // 32 MFMAs per barrier, 25 % fewer LDS-fill bytes per flop): lin_L1_ff1_geglu 662 -> 779, lin_L2_ff1_geglu 708 -> 865
        if (BN == 128 && ((GEGLU && p.K >= 640) || (MODE == V3D_GEMM_CONV3X3 && p.K <= 128)) && p.M >= 4096) cfg = 11;
    }
    if (p.K % 64 != 0 && (cfg == 1 || cfg == 4 || cfg == 7 || cfg == 9)) cfg = (cfg == 4) ? 3 : (cfg == 9 ? 8 : 0);   // BK 64 stages need K % 64 == 0
    switch (cfg) {
        case 1:   // BK 64 stages, 2 deep (one in flight)
            V3D_LAUNCH(2, BM, BN, (long long)grid.x * grid.y, (gemm_kernel_v2<BM, BN, 2, 2, 2, 2, MODE, GEGLU>), grid, 256, st, p);
            break;
        case 2:   // BK 32 stages, 3 deep: 3 blocks / CU
            V3D_LAUNCH(2, BM, BN, (long long)grid.x * grid.y, (gemm_kernel_v2<BM, BN, 2, 2, 3, 1, MODE, GEGLU>), grid, 256, st, p);
            break;
        case 3:   // 256 x 128 tile, 8 waves (4 x 2), BK 32 x 4 stages (N tiles of 64 keep the 4-wave kernel)
        case 4:   // 256 x 128 tile, 8 waves, BK 64 x 3 stages
            if constexpr (BN == 128) {
                p.mt = (int)((p.M + 255) / 256);
                dim3 g2((unsigned)(p.mt * p.nt), (unsigned)batch, 1);
                if (cfg == 3)
                    V3D_LAUNCH(2, 256, BN, (long long)g2.x * g2.y, (gemm_kernel_v2<256, BN, 4, 2, 4, 1, MODE, GEGLU>), g2, 512, st, p);
                else
                    V3D_LAUNCH(2, 256, BN, (long long)g2.x * g2.y, (gemm_kernel_v2<256, BN, 4, 2, 3, 2, MODE, GEGLU>), g2, 512, st, p);
                break;
            }
            [[fallthrough]];
        case 11:  // 256 x 128 tile on 4 waves (wave tile 128 x 64, 32 MFMAs per barrier), BK 32 x 3: 72 KiB -> 2 blocks / CU
            if constexpr (BN == 128) {
                p.mt = (int)((p.M + 255) / 256);
                dim3 g2((unsigned)(p.mt * p.nt), (unsigned)batch, 1);
                V3D_LAUNCH(2, 256, 128, (long long)g2.x * g2.y, (gemm_kernel_v2<256, 128, 2, 2, 3, 1, MODE, GEGLU>), g2, 256, st, p);
                break;
            }
            [[fallthrough]];
        case 10:  // BK 32 stages, 2 deep: 32 KiB LDS -> 4 blocks / CU (VGPR-limited)
            V3D_LAUNCH(2, BM, BN, (long long)grid.x * grid.y, (gemm_kernel_v2<BM, BN, 2, 2, 2, 1, MODE, GEGLU>), grid, 256, st, p);
            break;
        case 7:   // BK 64 stages, 3 deep
            V3D_LAUNCH(2, BM, BN, (long long)grid.x * grid.y, (gemm_kernel_v2<BM, BN, 2, 2, 3, 2, MODE, GEGLU>), grid, 256, st, p);
            break;
        case 8:   // 256 x 64 tile, 4 waves stacked along M (wave tile 64 x 64), BK 32 x 3
        case 9:   // 256 x 64 tile, 4 waves, BK 64 x 2
            if constexpr (BN == 64) {
                p.mt = (int)((p.M + 255) / 256);
                dim3 g2((unsigned)(p.mt * p.nt), (unsigned)batch, 1);
                if (cfg == 8)
                    V3D_LAUNCH(2, 256, 64, (long long)g2.x * g2.y, (gemm_kernel_v2<256, 64, 4, 1, 3, 1, MODE, GEGLU>), g2, 256, st, p);
                else
                    V3D_LAUNCH(2, 256, 64, (long long)g2.x * g2.y, (gemm_kernel_v2<256, 64, 4, 1, 2, 2, MODE, GEGLU>), g2, 256, st, p);
                break;
            }
            [[fallthrough]];
        default:  // cfg 0: BK 32 stages, 4 deep
            V3D_LAUNCH(2, BM, BN, (long long)grid.x * grid.y, (gemm_kernel_v2<BM, BN, 2, 2, 4, 1, MODE, GEGLU>), grid, 256, st, p);
    }
    return v3d_check_launch("v3d_gemm");
}

template <int MODE, bool GEGLU>
int launch256(const GP& p0, int batch, hipStream_t st, int cfg) {
    GP p = p0;
    p.mt = (int)((p.M + 255) / 256);
    p.nt = (int)((p.N + 255) / 256);
    dim3 grid((unsigned)(p.mt * p.nt), (unsigned)batch, 1);
    if (cfg == 5)
        V3D_LAUNCH(2, 256, 256, (long long)grid.x * grid.y, (gemm_kernel_v2<256, 256, 4, 2, 3, 1, MODE, GEGLU>), grid, 512, st, p);
    else
        V3D_LAUNCH(2, 256, 256, (long long)grid.x * grid.y, (gemm_kernel_v2<256, 256, 4, 2, 2, 2, MODE, GEGLU>), grid, 512, st, p);
    return v3d_check_launch("v3d_gemm");
}

// the v3 kernels only carry the branch-free epilogue: every wave tile (wm x wn) must be fully inside or fully outside the
// output and all vector-access alignment conditions of the fast path must hold
bool v3_ok(const GP& p, int wm, int wn) {
    auto al = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) % a) == 0; };
    if (p.M % wm || p.N % wn) return false;
    if (p.out_fp32 ? (p.ldo % 4 != 0 || !al(p.out, 16)) : (p.ldo % 8 != 0 || !al(p.out, 16))) return false;
    if (p.add && (!al(p.add, 16) || p.add_ld % 4)) return false;
    if (p.res1 && (!al(p.res1, 8) || p.ldr1 % 4)) return false;
    if (p.res2 && (!al(p.res2, 8) || p.ldr2 % 4)) return false;
    if (p.bias && !al(p.bias, 16)) return false;
    return true;
}

int v3s_choice() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("V3D_GEMM_V3S");   // 0 disables the two-blocks-per-CU GEGLU variant.  In isolation it ties with the
        v = e ? atoi(e) : 2;                      // 256 x 256 kernel (402 vs 396 us); inside the sampler it is +0.9 % end to end (same-box A/B);
                                                  // 1 = K < 640 only, 2 = every GEGLU projection (+0.35 % more)
    }
    return v;
}

// set by launch_v3 when the launch it made gathers GroupNorm statistics in its epilogue (else v3d_gemm runs the stand-alone kernel)
thread_local bool g_gn_in_epilogue = false;
long long g_gn_epilogue_launches = 0;      // (tests: how many launches of this process gathered the statistics in their epilogue)

template <int MODE, bool GEGLU>
int launch_v3(const GP& p0, hipStream_t st, int variant, long long m_off = 0, long long m_rows = -1) {
    GP p = p0;
    const int bm = variant == 1 ? 192 : 256, bn = variant == 1 ? 320 : (variant == 2 ? 128 : 256);
    p.m_off = m_off;                                        // (split launches: rows [m_off, m_off + m_rows) of the operation)
    p.mt = (int)(((m_rows > 0 ? m_rows : p.M) + bm - 1) / bm);
    p.nt = (int)((p.N + bn - 1) / bn);
    const int ntiles = p.mt * p.nt;
    const int grid = ntiles < v3d_num_cus() ? ntiles : v3d_num_cus();
    if constexpr (MODE == V3D_GEMM_LINEAR && !GEGLU) {
        if V3D_ABL(p, 8) {
            if (variant == 1) V3D_LAUNCH(3, 192, 320, ntiles, (gemm_kernel_v3<192, 320, 2, 4, MODE, GEGLU, 1, true>), dim3(grid), 512, st, p, ntiles);
            else V3D_LAUNCH(3, 256, 256, ntiles, (gemm_kernel_v3<256, 256, 2, 4, MODE, GEGLU, 1, true>), dim3(grid), 512, st, p, ntiles);
            return v3d_check_launch("v3d_gemm");
        }
    }
    if constexpr (GEGLU && MODE == V3D_GEMM_LINEAR) {
        if (variant == 2) {   // epilogue-bound GEGLU projections (K = 320): 256 x 128 tile, two blocks per CU
            const int g2 = ntiles < 2 * v3d_num_cus() ? ntiles : 2 * v3d_num_cus();
            V3D_LAUNCH(3, 256, 128, ntiles, (gemm_kernel_v3<256, 128, 4, 2, MODE, GEGLU, 1, false, 3, 0>), dim3(g2), 512, st, p, ntiles);
            return v3d_check_launch("v3d_gemm");
        }
    }
    if constexpr (!GEGLU) {
        if (variant == 1) {   // the N = 320 family: 192 x 320 tile, wave tile 96 x 80 (147456 rows = 768 tiles = 3 per CU)
            // statistics epilogue: every writer (wave tile x statistics group) stores into its own slot - needs whole groups per wave column
            // and gn_rps / (wave tile rows) + 2 slots
            if (p.gn_stats && 80 % p.gn_cpg == 0 && p.gn_nslots >= p.gn_rps / 96 + 2) {
                g_gn_in_epilogue = true;
                ++g_gn_epilogue_launches;
                V3D_LAUNCH(3, 192, 320, ntiles, (gemm_kernel_v3<192, 320, 2, 4, MODE, GEGLU, 1, false, 4, 16, true>), dim3(grid), 512, st, p, ntiles);
            } else {
                V3D_LAUNCH(3, 192, 320, ntiles, (gemm_kernel_v3<192, 320, 2, 4, MODE, GEGLU, 1>), dim3(grid), 512, st, p, ntiles);
            }
            return v3d_check_launch("v3d_gemm");
        }
        if (p.gn_stats && variant == 0 && 64 % p.gn_cpg == 0 && p.gn_nslots >= p.gn_rps / 128 + 2) {
            g_gn_in_epilogue = true;
            ++g_gn_epilogue_launches;
            V3D_LAUNCH(3, 256, 256, ntiles, (gemm_kernel_v3<256, 256, 2, 4, MODE, GEGLU, 1, false, 4, 16, true>), dim3(grid), 512, st, p, ntiles);
            return v3d_check_launch("v3d_gemm");
        }
    }
    if constexpr (GEGLU) {
        // GEGLU halves the output width: with waves laid out 4 (M) x 2 (N) a wave owns 128 weight rows = 64 output channels = whole
        // 128-byte lines of every output row (the 2 x 4 layout wrote 64-byte half lines from two different waves)
        // measured (tools/gemm_floor.py, M = 147456, N = 2560): K = 320: 396 us vs 416 us (2 x 4) vs 402 us (two 256 x 128 blocks
        // per CU) vs 421 us (v2); at K >= 640 the 2 x 4 layout with 32-row chunks is ahead again (tools/gemm_sweep.py)
        if (p.K < 640 && !V3D_ABL(p, 128)) {
            V3D_LAUNCH(3, 256, 256, ntiles, (gemm_kernel_v3<256, 256, 4, 2, MODE, GEGLU, 1>), dim3(grid), 512, st, p, ntiles);
            return v3d_check_launch("v3d_gemm");
        }
    }
    // epilogue chunk: 2 row fragments for GEGLU (its staged rows are half as wide), 1 otherwise (LDS budget next to the ring)
    V3D_LAUNCH(3, 256, 256, ntiles, (gemm_kernel_v3<256, 256, 2, 4, MODE, GEGLU, GEGLU ? 2 : 1>), dim3(grid), 512, st, p, ntiles);
    return v3d_check_launch("v3d_gemm");
}

// v6 (two persistent 4-wave blocks per CU).  V3D_GEMM_V6: 0 = never, 1 = where v3's tiles quantise badly (the rule in dispatch), 2 = every legal launch (A/B knob)
int v6_choice() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("V3D_GEMM_V6");
        v = e ? atoi(e) : V3D_GEMM_V6_DEFAULT;
    }
    return v;
}
template <int MODE>
int launch_v6(const GP& p0, hipStream_t st) {
    GP p = p0;
    p.mt = (int)(p.M / 192);
    p.nt = (int)(p.N / 160);
    p.group_m = 0;                       // row-major walk: the N-tiles of a row tile are consecutive ids -> blocks 8 apart on one XCD share its activation rows in L2
    const int ntiles = p.mt * p.nt;
    const int grid = ntiles < 2 * v3d_num_cus() ? ntiles : 2 * v3d_num_cus();
    V3D_LAUNCH(6, 192, 160, ntiles, (gemm_kernel_v6<192, 160, 3, MODE>), dim3(grid), 256, st, p, ntiles);
    return v3d_check_launch("v3d_gemm(v6)");
}

// Tile quantisation of the persistent kernels (round 6).  A launch of t tiles on G block slots idles (ceil(t / G) G - t) / G of its last round; the
// 32 x 32 and 16 x 16 levels of the U-Net sit at 1.4 - 2.25 rounds of v3 tiles.  gemm_kernel_v6's 192 x 160 tiles on 512 slots halve the granule:
// measured (profiles/r06_v6_ab.txt) -9 ... -14 % on the M = 9216 projections (N = 2560 / 3840) and the N = K = 640 linears, -5 % on the 32 x 32 temporal
// convolution, a tie where v3's tiles fill >= 0.9 of their rounds or K = 2560 (v2 is as good there), +1 ... 6 % where v3 runs whole rounds (N = 320 at
// 64 x 64, M = 36864 x N = 1280) - those stay where they were.  Also measured and dropped (same record): a SPLIT launch (v3 on the rows that make whole rounds,
// the tail rows as a second launch on 192 x 160 / 96 x 160 tiles: +5 ... 8 % slower than either single launch - the second launch's ramp costs more than the
// idle half round), 192 x 128 tiles for N = 1280 at M = 9216 (480 tiles on 512 slots: +4 ... 18 % slower than v3's 192 tiles at 0.75 of a round), and 96 x 160 tiles for
// launches whose v3 tiles cover half the CUs or fewer (the 8 x 8 level, the 4 - 6-image ranks of a frame shard: +-3 %, and 1.6x slower than split-K at K = 5120).
// returns -1 when the rule does not apply (the caller keeps its v3 / v2 path).  V3D_GEMM_V6: 0 = never, 1 = the rule (default), 2 = every legal launch (A/B knob)
template <int MODE>
int try_v6(const GP& p, hipStream_t st) {
    const int v6 = v6_choice();
    if (!v6 || impl_choice() != 0 || cfg_choice() >= 0 || p.K % 32 || p.K * 2 > 65536 || p.M % 192 || p.N % 160 || p.gn_stats || !e4_ok(p, 96, 80)) return -1;
    const long long cus = v3d_num_cus(), slots = 2 * cus;
    const bool v1 = p.N % 320 == 0;
    const long long bm = v1 ? 192 : 256, bn = v1 ? 320 : 256;
    const long long t3 = ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn), t6 = (p.M / 192) * (p.N / 160);
    const double fill3 = (double)t3 / (double)(((t3 + cus - 1) / cus) * cus) * ((double)p.N / (double)(((p.N + bn - 1) / bn) * bn));
    const double fill6 = (double)t6 / (double)(((t6 + slots - 1) / slots) * slots);
    if (t6 >= slots && (v6 >= 2 || (p.K <= 1280 && fill3 < 0.9 && fill6 >= fill3))) return launch_v6<MODE>(p, st);
    return -1;
}

template <int MODE, bool GEGLU>
int dispatch(const GP& p, int batch, hipStream_t st) {
#ifdef V3D_EXPERIMENTS
    if constexpr (MODE == V3D_GEMM_LINEAR && !GEGLU) {
        static int v7 = -1;
        if (v7 < 0) {
            const char* e = getenv("V3D_GEMM_V7");
            v7 = e ? atoi(e) : 0;
        }
        if (v7 && batch == 1 && v3d_gemm_v7_variant(p, MODE)) return v3d_gemm_v7_launch(p, (void*)st);
    }
#endif
    if constexpr ((MODE == V3D_GEMM_LINEAR || MODE == V3D_GEMM_CONVT3) && !GEGLU) {
        if (batch == 1) {
            const int rc = try_v6<MODE>(p, st);
            if (rc >= 0) return rc;
        }
    }
    // v3 (persistent big tiles) unless forced off (V3D_GEMM_IMPL=1/2), forced on (=3), or the tile count fills the CUs badly
    if (impl_choice() != 1 && impl_choice() != 2 && cfg_choice() < 0 && batch == 1 && p.K % 32 == 0 && p.K * 2 <= 65536 && p.N >= 256) {
        const int variant = (!GEGLU && p.N % 320 == 0) ? 1 : 0;
        const long long bm = variant ? 192 : 256, bn = variant ? 320 : 256;
        const long long nt3 = ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn), cus = v3d_num_cus();
        const double fill3 = (double)nt3 / (double)(((nt3 + cus - 1) / cus) * cus) * ((double)p.N / (double)(((p.N + bn - 1) / bn) * bn));
        // v2 reference point: 128 x 128 (or x 64) tiles, 2 blocks per CU
        const long long w128 = ((p.N + 127) / 128) * 128, w64 = ((p.N + 63) / 64) * 64, wv2 = w64 < w128 ? w64 : w128;
        const long long nt2 = ((p.M + 127) / 128) * (wv2 / (w64 < w128 ? 64 : 128));
        const double fill2 = (double)nt2 / (double)(((nt2 + 2 * cus - 1) / (2 * cus)) * 2 * cus) * ((double)p.N / (double)wv2);
        // measured (profiles/r01f_op_times_v3.txt): v3 wins wherever its tiles fill the CUs about as well as v2's do
        if (GEGLU && (p.K < 640 || v3s_choice() >= 2) && v3s_choice() && v3_ok(p, 64, 64) && ((p.M + 255) / 256) * ((p.N + 127) / 128) >= 2 * cus)
            return launch_v3<MODE, GEGLU>(p, st, 2);
        static double fill_k = -1.0;
        if (fill_k < 0) {
            const char* e = getenv("V3D_GEMM_FILL");   // tuning knob
            fill_k = e ? atof(e) : 0.9;
        }
        const bool want = impl_choice() == 3 || fill3 >= fill_k * fill2;
        // (non-GEGLU v3 kernels only carry the hand-managed epilogue: its operand contract on top of the tile-shape one)
        const bool eok = GEGLU || (variant ? e4_ok(p, 96, 80) : e4_ok(p, 128, 64));
        if (want && eok && (variant ? v3_ok(p, 96, 80) : v3_ok(p, 128, 128))) {
#ifdef V3D_EXPERIMENTS
            if constexpr (!GEGLU) {
                // lab only (tools/lab/gemm4.hip, linked into the experiments library): the one-wave-per-SIMD kernels of round 4, V3D_GEMM_V4=1
                static int v4 = -1;
                if (v4 < 0) {
                    const char* e = getenv("V3D_GEMM_V4");
                    v4 = e ? atoi(e) : 0;
                }
                static int v5 = -1;
                if (v5 < 0) {
                    const char* e = getenv("V3D_GEMM_V5");
                    v5 = e ? atoi(e) : 0;
                }
                const int v5v = v5 ? v3d_gemm_v5_variant(p, MODE, variant) : 0;
                if (v5v) return v3d_gemm_v5_launch(p, MODE, v5v, (void*)st);
                const int v4v = v4 ? v3d_gemm_v4_variant(p, MODE, variant) : 0;
                if (v4v) {
                    if (p.gn_stats) {
                        g_gn_in_epilogue = true;
                        ++g_gn_epilogue_launches;
                    }
                    return v3d_gemm_v4_launch(p, MODE, v4v, (void*)st);
                }
            }
#endif
            return launch_v3<MODE, GEGLU>(p, st, variant);
        }
    }
    if ((cfg_choice() == 5 || cfg_choice() == 6) && impl_choice() != 1 && p.N % 256 == 0 && p.K % 64 == 0 && p.K * 2 <= 65536) return launch256<MODE, GEGLU>(p, batch, st, cfg_choice());
    // N tile: 128 unless a 64-wide tile wastes less (e.g. N = 320: 5 x 64 exact vs 3 x 128 = 17 % padding)
    const long long w128 = ((p.N + 127) / 128) * 128, w64 = ((p.N + 63) / 64) * 64;
    // (64-row tiles for the small-M 8x8 level were measured slower than under-filled 128-row tiles: conv_L3 461 vs 586 TF/s)
    if (w64 < w128) return launch<128, 64, MODE, GEGLU>(p, batch, st);
    return launch<128, 128, MODE, GEGLU>(p, batch, st);
}

}  // namespace

void v3d_note_launch(int family, int bm, int bn, long long tiles, int blocks_per_cu, int streamk) {
    g_last_launch = V3dLaunchInfo{family, bm, bn, tiles, blocks_per_cu, 0, streamk};
}

// tests / tools only (not part of the ABI header): out[8] = family, bm, bn, tiles, blocks per CU, split-K ways, stream-K tail, CUs
extern "C" int v3d_debug_last_gemm_launch(long long* out) {
    out[0] = g_last_launch.family; out[1] = g_last_launch.bm; out[2] = g_last_launch.bn; out[3] = g_last_launch.tiles;
    out[4] = g_last_launch.blocks_per_cu; out[5] = g_last_launch.splitk; out[6] = g_last_launch.streamk; out[7] = v3d_num_cus();
    return 0;
}

// tests only (not part of the ABI header)
extern "C" long long v3d_debug_gn_epilogue_launches(void) { return g_gn_epilogue_launches; }

// ---- stream-K plan (device side: gemm_common.h sk_*) ---------------------------------------------------------------------------------------
namespace {
struct SkWorkspace {
    hipStream_t st;
    float* ws;
    unsigned* flags;
    size_t slot_bytes;
    int G;
};
std::vector<SkWorkspace> g_sk;       // (one per stream that ever planned a stream-K launch; never freed: captured graphs point into them)
std::mutex g_sk_mu;
long long g_sk_launches = 0;
}  // namespace

// would a launch of `ntiles` tiles of `units` granules get a stream-K tail?  Worth it when the partial round leaves >= 1/8 of the chip idle,
// every block still gets a piece of >= min_units granules, and the idle time it removes ((G - R) / G of a tile, in granules) is >= min_saved:
// the hand-off costs 10-15 us per owner (publish 245 KB write-through, flag, read back).
bool v3d_sk_wanted(int ntiles, int units, int min_units, int min_saved) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("V3D_STREAMK"); on = e ? atoi(e) : 1; }          // A/B knob: 0 = classic tile assignment everywhere
    const int G = v3d_num_cus();
    const int R = ntiles % G;
    return on && R != 0 && (G - R) * 8 >= G && (long long)R * units / G >= min_units && (long long)(G - R) * units / G >= min_saved;
}

int v3d_sk_plan(V3dGemmParams& p, int ntiles, int units, int min_units, int min_saved, size_t slot_bytes, void* stream) {
    p.sk_tail = p.sk_full = 0;
    p.sk_units = units;
    p.sk_ws = nullptr;
    p.sk_flags = nullptr;
    const int G = v3d_num_cus();
    const int classic = ntiles < G ? ntiles : G;
    const int full = ntiles / G, R = ntiles % G;
    if (!v3d_sk_wanted(ntiles, units, min_units, min_saved)) return classic;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(g_sk_mu);
    const SkWorkspace* w = nullptr;
    for (const SkWorkspace& c : g_sk)
        if (c.st == st && c.slot_bytes >= slot_bytes && c.G == G) w = &c;
    if (!w) {
        // one workspace per stream (launches on one stream run back to back; two streams must not share slots); never allocated inside a capture
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return classic; }
        SkWorkspace n = {st, nullptr, nullptr, slot_bytes, G};
        if (hipMalloc(reinterpret_cast<void**>(&n.ws), (size_t)G * slot_bytes) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&n.flags), ((size_t)G * 8 + 16) * 4) != hipSuccess ||
            hipMemset(n.flags, 0, ((size_t)G * 8 + 16) * 4) != hipSuccess) {
            (void)hipGetLastError();
            return classic;
        }
        g_sk.push_back(n);
        w = &g_sk.back();
    }
    p.sk_tail = R;
    p.sk_full = full;
    p.sk_ws = w->ws;
    p.sk_flags = w->flags;
    ++g_sk_launches;
    return G;
}

// tests only (CPU): the work list block b of a G-block launch would build for `ntiles` tiles of `units` granules - the device code's own table
// (gemm_common.h sk_build_table), so the decomposition can be checked for every shape without a GPU.  out[12]: items, donor flag, then two
// pieces of (tile, u0, u1, role, d0, d1) - the table's layout without its padding.  Returns 0 when no tail is planned for this launch.
extern "C" int v3d_debug_sk_table(int ntiles, int units, int G, int b, int min_units, int min_saved, int* out) {
    const int R = ntiles % G;
    if (R == 0 || (G - R) * 8 < G || (long long)R * units / G < min_units || (long long)(G - R) * units / G < min_saved) return 0;
    V3dGemmParams p = {};
    p.sk_tail = R;
    p.sk_full = ntiles / G;
    p.sk_units = units;
    int tab[SK_TAB_BYTES / 4] = {0};
    sk_build_table(p, b, G, ntiles, units, tab);
    out[0] = tab[0];
    out[1] = tab[1];
    for (int k = 0; k < 2; ++k)
        for (int j = 0; j < 6; ++j) out[2 + 6 * k + j] = tab[4 + 8 * k + j];
    return 1;
}

// tests only: launches planned with a stream-K tail; hand-offs that gave up waiting (must stay 0)
extern "C" long long v3d_debug_sk_launches(void) { return g_sk_launches; }
extern "C" long long v3d_debug_sk_timeouts(void) {
    long long t = 0;
    std::lock_guard<std::mutex> lock(g_sk_mu);
    for (const SkWorkspace& c : g_sk) {
        unsigned v = 0;
        if (hipMemcpy(&v, c.flags + (size_t)c.G * 8, 4, hipMemcpyDeviceToHost) == hipSuccess) t += v;
    }
    return t;
}

// experiments only (not part of the ABI header): copy the v3 slot timeline out
extern "C" int v3d_debug_v3_timeline(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_v3_dbg), sizeof(g_v3_dbg)) == hipSuccess ? 0 : -1;
}

namespace {
int run_mode(const v3d_gemm_args* a, GP& p, hipStream_t st);
}

namespace {
// argument checks + the kernels' parameter block; *halo = the LDS-haloed kernel variant that takes the launch (conv.hip), 0 = none
int fill_params(const v3d_gemm_args* a, GP& p, int* halo) {
    V3D_REQUIRE(a != nullptr, "v3d_gemm: null args");
    V3D_REQUIRE(a->A && a->W && a->out, "v3d_gemm: null A/W/out");
    V3D_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "v3d_gemm: bad M/N/K (%lld,%lld,%lld)", (long long)a->M, (long long)a->N, (long long)a->K);
    V3D_REQUIRE(a->K % 8 == 0 && a->lda % 8 == 0, "v3d_gemm: K and lda must be multiples of 8 (K=%lld lda=%lld)", (long long)a->K, (long long)a->lda);
    V3D_REQUIRE(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->W & 15) == 0, "v3d_gemm: A/W must be 16-byte aligned");
    V3D_REQUIRE(a->batch >= 1 && a->batch <= 65535, "v3d_gemm: bad batch %d", a->batch);
    V3D_REQUIRE(!a->geglu || (a->N % 32 == 0 && a->mode == V3D_GEMM_LINEAR), "v3d_gemm: geglu needs LINEAR mode and N %% 32 == 0");
    V3D_REQUIRE(!a->add || a->add_rpg > 0, "v3d_gemm: add_rpg must be > 0");
    V3D_REQUIRE(!a->coef || a->coef_rpg > 0, "v3d_gemm: coef_rpg must be > 0");
    V3D_REQUIRE(a->sA % 8 == 0 && a->sW % 8 == 0, "v3d_gemm: batch strides must keep 16-byte alignment");
    V3D_REQUIRE(a->a_rows > 0 && a->a_row0 >= 0, "v3d_gemm: a_rows (rows addressable behind A) must be given");
    V3D_REQUIRE(!a->bias || ((uintptr_t)a->bias & 15) == 0, "v3d_gemm: bias must be 16-byte aligned");
    const long long mtiles = (a->M + 127) / 128, ntiles = (a->N + 63) / 64;
    V3D_REQUIRE(mtiles * ntiles < (1ll << 31), "v3d_gemm: grid too large");
    const int taps = a->mode == V3D_GEMM_LINEAR ? 1 : (a->mode == V3D_GEMM_CONV3X3 ? 9 : 3);
    const long long ldw = a->ldw ? a->ldw : a->K;
    V3D_REQUIRE(ldw >= a->K && ldw % 8 == 0, "v3d_gemm: ldw must be >= K and a multiple of 8");
    const unsigned long long a_bytes = (unsigned long long)a->a_rows * a->lda * 2ull;
    const unsigned long long w_bytes = ((unsigned long long)(taps * a->N - 1) * ldw + a->K) * 2ull;
    V3D_REQUIRE(a_bytes <= kMaxBufBytes && w_bytes <= kMaxBufBytes, "v3d_gemm: operand larger than 4 GiB - 256 B (A %llu B, W %llu B)", a_bytes, w_bytes);
    p.A = (const bf16_t*)a->A; p.W = (const bf16_t*)a->W; p.out = a->out;
    p.bias = a->bias; p.add = a->add; p.res1 = (const bf16_t*)a->res1; p.res2 = (const bf16_t*)a->res2; p.coef = a->coef;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.lda = a->lda; p.ldw = ldw; p.ldo = a->ldo; p.ldr1 = a->ldr1; p.ldr2 = a->ldr2;
    p.add_rpg = a->add_rpg; p.add_ld = a->add_ld; p.coef_rpg = a->coef_rpg;
    p.c_acc = a->c_acc; p.c_res1 = a->c_res1; p.c_res2 = a->c_res2;
    p.out_fp32 = a->out_fp32;
    p.a_row0 = a->a_row0; p.a_bytes = (unsigned)a_bytes; p.w_bytes = (unsigned)w_bytes;
    p.Hin = a->Hin; p.Win = a->Win; p.Hout = a->Hout; p.Wout = a->Wout; p.stride = a->stride;
    p.upshift = 0;
    p.pad_lo = 1;
    p.T = a->T; p.tmin = a->tmin; p.tmax = a->tmax; p.S = a->S;
    p.halo_rows = a->mode == V3D_GEMM_CONVT3 ? a->halo_rows : 0;
    p.sA = a->sA; p.sW = a->sW; p.sO = a->sO;
    p.mt = p.nt = 0;
    p.m_off = 0;
    {
        static int gm = -2, ti = -2;
        if (gm == -2) { const char* e = getenv("V3D_GEMM_GROUPM"); gm = e ? atoi(e) : -1; }      // -1 = heuristic, 0 / 1 = row-major walk, n = groups of n tile rows
        if (ti == -2) { const char* e = getenv("V3D_GEMM_TAPINNER"); ti = e ? atoi(e) : -1; }
        p.group_m = gm;        // (-1: resolved per launch from the tile-row count, see launch_group_m)
        p.tap_inner = ti < 0 ? V3D_GEMM_TAPINNER_DEFAULT : ti;
    }
    p.split_n = 1;
    p.ws = nullptr;
    p.ablate = 0;
    p.sk_tail = p.sk_full = p.sk_units = 0; p.sk_ws = nullptr; p.sk_flags = nullptr;
#ifdef V3D_EXPERIMENTS
    { static int ab = -1; if (ab < 0) { const char* e = getenv("V3D_GEMM_ABLATE"); ab = e ? atoi(e) : 0; } p.ablate = ab; }
#endif
    p.gn_stats = nullptr; p.gn_rps = 0; p.gn_cpg = 0; p.gn_nslots = 0;
    if (a->gn_stats) {
        V3D_REQUIRE(!a->geglu && !a->out_fp32 && a->batch == 1, "v3d_gemm: gn_stats needs a bf16, non-GEGLU, unbatched output");
        V3D_REQUIRE(a->gn_cpg > 0 && a->gn_cpg % 2 == 0 && a->N == 32ll * a->gn_cpg, "v3d_gemm: gn_stats needs N = 32 * gn_cpg (got N=%lld cpg=%d)", (long long)a->N, a->gn_cpg);
        V3D_REQUIRE(a->gn_rps >= 16 && a->gn_rps % 16 == 0 && a->gn_rps < (1ll << 30) && a->M % a->gn_rps == 0, "v3d_gemm: gn_rps must be a multiple of 16 that divides M");
        V3D_REQUIRE(a->ldo == a->N && ((uintptr_t)a->gn_stats & 7) == 0, "v3d_gemm: gn_stats needs a dense output (ldo == N) and an 8-byte aligned buffer");
        V3D_REQUIRE(a->gn_nslots >= 1, "v3d_gemm: gn_nslots must be >= 1");
        static int ep = -1;
        if (ep < 0) { const char* e = getenv("V3D_GEMM_GN_EPILOGUE"); ep = e ? atoi(e) : 1; }      // A/B knob: 0 = always the stand-alone statistics kernel
        if (ep) { p.gn_stats = a->gn_stats; p.gn_rps = a->gn_rps; p.gn_cpg = a->gn_cpg; p.gn_nslots = a->gn_nslots; }
    }
    // GroupNorm (+SiLU) of the input in the operand path / two-source input: only the LDS-haloed kernels (conv.hip) take these
    p.A2 = (const bf16_t*)a->A2; p.K1 = a->A2 ? a->K1 : a->K; p.lda2 = a->lda2; p.a2_bytes = 0;
    p.gn_in = a->gn_in_table; p.gn_in_rps = a->gn_in_rps; p.gn_in_silu = a->gn_in_silu; p.gn_in_bytes = 0;
    if (a->A2) {
        V3D_REQUIRE(a->gn_in_table != nullptr, "v3d_gemm: a two-source input (A2) is only defined together with gn_in_table");
        const unsigned long long a2b = (unsigned long long)a->a_rows * a->lda2 * 2ull;
        V3D_REQUIRE(a->lda2 >= a->K - a->K1 && a2b <= kMaxBufBytes, "v3d_gemm: bad lda2 / A2 larger than 4 GiB - 256 B");
        p.a2_bytes = (unsigned)a2b;
    }
    if (a->gn_in_table) {
        V3D_REQUIRE(a->gn_in_rps > 0 && a->gn_in_rows > 0, "v3d_gemm: gn_in_table needs gn_in_rps and gn_in_rows");
        const unsigned long long tb = (unsigned long long)a->gn_in_rows * a->K * 8ull;
        V3D_REQUIRE(tb <= kMaxBufBytes && a->batch == 1 && !a->geglu, "v3d_gemm: gn_in_table too large / batched / GEGLU");
        p.gn_in_bytes = (unsigned)tb;
    }
    *halo = 0;
    if (a->mode == V3D_GEMM_CONV3X3) {
        p.upshift = a->up - 1;
        p.pad_lo = a->pad_mode ? 0 : 1;
    }
    static int hk = -1;
    if (hk < 0) { const char* e = getenv("V3D_CONV_HALO"); hk = e ? atoi(e) : 1; }     // A/B knob: 0 = never, 1 = launches with gn_in_table (default), 2 = every launch of a fitting shape
    if (a->batch == 1 && !a->geglu && (a->mode == V3D_GEMM_CONV3X3 || a->mode == V3D_GEMM_CONVT3) && ((hk == 1 && a->gn_in_table) || hk >= 2))
        *halo = v3d_conv_halo_variant(p, a->mode);
    return V3D_OK;
}
}  // namespace

extern "C" int v3d_gemm_gn_in_supported(const v3d_gemm_args* a) {
    GP p;
    int halo = 0;
    if (!a || !a->gn_in_table || fill_params(a, p, &halo) != V3D_OK) return 0;
    return halo != 0;
}

extern "C" int v3d_gemm(const v3d_gemm_args* a, v3d_stream_t stream) {
    GP p;
    int halo = 0;
    const int frc = fill_params(a, p, &halo);
    if (frc != V3D_OK) return frc;
    hipStream_t st = (hipStream_t)stream;
    if (halo) {
        const int rc = v3d_conv_halo_launch(p, halo, stream);     // (its epilogue gathers gn_stats itself ...
        // ... unless the A/B knob V3D_GEMM_GN_EPILOGUE=0 took the request out of the parameter block: then the stand-alone pass runs here too)
        if (rc != V3D_OK || !a->gn_stats || p.gn_stats) return rc;
        return v3d_groupnorm_stats(a->out, a->N, nullptr, 0, a->gn_stats, a->gn_nslots, a->M / a->gn_rps, a->gn_rps, 32, 1, stream);
    }
    V3D_REQUIRE(!a->gn_in_table, "v3d_gemm: gn_in_table is set but this shape is not one of the LDS-haloed kernels' (v3d_gemm_gn_in_supported): "
                                 "normalise the input with v3d_groupnorm_apply first");
    if (p.gn_stats || a->gn_stats) {
        g_gn_in_epilogue = false;
        const int rc = run_mode(a, p, st);
        if (rc != V3D_OK || g_gn_in_epilogue) return rc;
        // the kernel that ran has no statistics epilogue (v1 / v2 tiles, split-K, ragged shapes): same result from the stand-alone kernel
        return v3d_groupnorm_stats(a->out, a->N, nullptr, 0, a->gn_stats, a->gn_nslots, a->M / a->gn_rps, a->gn_rps, 32, 1, stream);
    }
    return run_mode(a, p, st);
}

namespace {
int run_mode(const v3d_gemm_args* a, GP& p, hipStream_t st) {
    switch (a->mode) {
        case V3D_GEMM_LINEAR:
            if (a->geglu) return dispatch<V3D_GEMM_LINEAR, true>(p, a->batch, st);
            return dispatch<V3D_GEMM_LINEAR, false>(p, a->batch, st);
        case V3D_GEMM_CONV3X3:
            V3D_REQUIRE(a->up == 1 || a->up == 2, "v3d_gemm: up must be 1 or 2");
            V3D_REQUIRE(a->stride == 1 || a->stride == 2, "v3d_gemm: stride must be 1 or 2");
            V3D_REQUIRE(a->Hin > 0 && a->Win > 0 && a->Hout > 0 && a->Wout > 0, "v3d_gemm: bad conv geometry");
            V3D_REQUIRE(a->M % ((long long)a->Hout * a->Wout) == 0, "v3d_gemm: M must be n_img*Hout*Wout");
            V3D_REQUIRE(a->pad_mode == 0 || a->pad_mode == 1, "v3d_gemm: pad_mode must be 0 or 1");
            p.upshift = a->up - 1;
            p.pad_lo = a->pad_mode ? 0 : 1;
            return dispatch<V3D_GEMM_CONV3X3, false>(p, a->batch, st);
        case V3D_GEMM_CONVT3:
            V3D_REQUIRE(a->T > 0 && a->S > 0, "v3d_gemm: convt3 needs T,S");
            V3D_REQUIRE(a->halo_rows >= 0, "v3d_gemm: halo_rows must be >= 0");
            if (a->halo_rows > 0) {
                V3D_REQUIRE(a->M % ((long long)a->T * a->S) == 0 && a->halo_rows == a->M / a->T, "v3d_gemm: split-halo layout needs M = B*T*S and halo_rows = B*S");
                V3D_REQUIRE(a->a_row0 >= a->halo_rows && a->a_rows >= a->a_row0 + a->M + a->halo_rows, "v3d_gemm: split-halo layout: A must hold halo_rows rows on either side of the M local rows");
                V3D_REQUIRE(a->tmin >= -1 && a->tmax <= a->T, "v3d_gemm: split-halo layout carries one halo frame per side");
            }
            return dispatch<V3D_GEMM_CONVT3, false>(p, a->batch, st);
        default:
            v3d_set_error("v3d_gemm: unknown mode %d", a->mode);
            return V3D_ERR_ARG;
    }
}
}  // namespace
