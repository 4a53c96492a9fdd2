// Shared device code of the v3d_gemm kernel family (gemm.hip) and the LDS-haloed convolution (conv.hip): launch parameters, tap / row
// addressing, XCD-aware tile walk, the fused epilogue (bias / per-image vector / GEGLU / residuals / alpha blend, LDS-staged 16-byte row
// stores) and the GroupNorm-statistics epilogue.  Everything but the parameter block lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include <type_traits>

#include "common.h"

// LDS reads of the hand-managed epilogue (e4_fragment, gn_flush) as inline asm: 0 compiler-visible reads, 1 batched asm reads, 2 one piece per LDS round trip
// (see e4_lds_read8 below for why; e4_fragment takes the form as its template parameter RD, this is its default)
#ifndef E4_ASM_READS
#define E4_ASM_READS 1
#endif

// Experiment switches (skip epilogue / MFMAs / DMA / stores, slot timelines) exist only in builds made with -DV3D_EXPERIMENTS
// (tools/gemm_floor.py, tools/v3_timeline.py say how); the shipped library has no environment variable that changes results.
#ifdef V3D_EXPERIMENTS
#define V3D_ABL(p, bit) ((p).ablate & (bit))
#elif defined(V3D_ABL_STATIC)      // A/B builds only (tools/build_variant.sh <tag> "-DV3D_ABL_STATIC=<bits>"): the same switches at compile time - no branch in the measured code
#define V3D_ABL(p, bit) ((V3D_ABL_STATIC) & (bit))
#else
#define V3D_ABL(p, bit) (0)
#endif

struct V3dGemmParams {
    const bf16_t* A;
    const bf16_t* W;
    void* out;
    const float* bias;
    const float* add;
    const bf16_t* res1;
    const bf16_t* res2;
    const float* coef;
    long long M, N, K;
    long long lda, ldw, ldo, ldr1, ldr2;
    long long add_rpg, add_ld, coef_rpg;
    float c_acc, c_res1, c_res2;
    int out_fp32;
    long long a_row0;
    unsigned a_bytes, w_bytes;
    int Hin, Win, Hout, Wout, stride, upshift;
    int pad_lo;          // zero pixels before the first row / column (1, or 0 for the asymmetric right/bottom padding)
    int T, tmin, tmax;
    long long S;
    long long halo_rows; // CONVT3 split-halo layout (0 = dense)
    long long m_off;     // first output row of this launch (a launch over the row range [m_off, m_off + mt * BM) of the operation: gemm.hip split launches); M stays the operation's
    long long sA, sW, sO;
    int mt, nt;  // tile counts
    int group_m;         // tile walk: > 1 = ids run down groups of `group_m` tile rows first (L2-friendly patches), else row-major over N
    int tap_inner;       // multi-tap modes: 1 = stage order (k outer, tap inner): the taps re-read an activation tile while it is still in L2
    int split_n;         // > 1: split-K launch: blockIdx.y = split index = output slab (out = fp32 workspace [split][M][N], plain stores)
    const float* ws;     // finalize kernel only: the workspace to reduce
    int ablate;  // experiments only (env V3D_GEMM_ABLATE): 1 = no output stores, 2 = no MFMAs, 4 = no LDS-DMA loads
    float* gn_stats;     // GroupNorm partial sums of the output [M / gn_rps][gn_nslots][32][2], stored by the <GN> epilogues (else NULL)
    long long gn_rps;    // rows per statistics group
    long long gn_nslots; // slots per statistics group; a writer (wave tile x statistics group) owns slot ceil(first row in group / wave-tile rows)
    int gn_cpg;          // channels per group (N / 32)
    // GroupNorm (+SiLU) of the INPUT applied in the operand path (conv.hip): A (and A2 behind K1 channels) hold the raw tensor
    const bf16_t* A2;    // second channel-concatenated source (K - K1 channels, row stride lda2) or NULL
    long long K1, lda2;
    unsigned a2_bytes;
    const float* gn_in;  // [stat][K][2] fp32 (scale, shift) table of v3d_groupnorm_finalize, stat = source row / gn_in_rps
    long long gn_in_rps;
    int gn_in_silu;
    unsigned gn_in_bytes;   // size of the table (buffer descriptor)
    // stream-K tail of the persistent kernels (sk_* helpers below): 0 = classic (tiles b, b + G, ... per block)
    int sk_tail;         // tiles of the last, partial round, shared out over ALL blocks in units of sk_units-th of a tile
    int sk_full;         // full rounds in front of it (tiles b + i * G, i < sk_full)
    int sk_units;        // split granules per tile (conv.hip: 32-channel chunks; gemm.hip v3: 32-k steps)
    float* sk_ws;        // partial accumulators: one slot per block [G][8 waves][MF * NF][64 lanes] f32x4
    unsigned* sk_flags;  // [G][8]: 1 = wave w of block g has published its partial (reset to 0 by the consumer)
};

// gemm.hip: record of the last launch v3d_gemm made on this thread (v3d_debug_last_gemm_launch)
struct V3dLaunchInfo {
    int family, bm, bn;
    long long tiles;
    int blocks_per_cu, splitk, streamk;
};
void v3d_note_launch(int family, int bm, int bn, long long tiles, int blocks_per_cu, int streamk);
// conv.hip: the LDS-haloed kernels (GroupNorm + SiLU in the operand path)
int v3d_conv_halo_variant(const V3dGemmParams& p, int mode);          // 0 = not one of their shapes
int v3d_conv_halo_launch(const V3dGemmParams& p, int variant, void* stream);
#ifdef V3D_EXPERIMENTS
// tools/lab/gemm4.hip (experiments library only): the one-wave-per-SIMD persistent kernels of round 4 (0 = not one of their launches, else the
// variant: 1 = 192 x 320 tiles, 2 = 256 x 256).  5 % slower than v3 (NOTES 11.2): kept as a lab record, not part of the product library.
int v3d_gemm_v4_variant(const V3dGemmParams& p, int mode, int v3_variant);
int v3d_gemm_v4_launch(const V3dGemmParams& p, int mode, int variant, void* stream);
// tools/lab/gemm5.hip: the same structure on v_mfma_f32_32x32x16_bf16 (round 6, V3D_GEMM_V5=1)
int v3d_gemm_v5_variant(const V3dGemmParams& p, int mode, int v3_variant);
int v3d_gemm_v5_launch(const V3dGemmParams& p, int mode, int variant, void* stream);
// tools/lab/gemm7.hip: N = 320, K = 320 / 640 linears with a deferred epilogue (round 6, V3D_GEMM_V7=1)
int v3d_gemm_v7_variant(const V3dGemmParams& p, int mode);
int v3d_gemm_v7_launch(const V3dGemmParams& p, void* stream);
#endif
// gemm.hip: stream-K plan of a persistent launch (fills p.sk_*, returns the grid): ntiles tiles of `units` split granules on the device's CUs;
// slot_bytes = one block's accumulators in fp32.  Leaves the classic assignment (sk_tail = 0) when the tail round is full enough or too thin.
int v3d_sk_plan(V3dGemmParams& p, int ntiles, int units, int min_units, int min_saved, size_t slot_bytes, void* stream);
bool v3d_sk_wanted(int ntiles, int units, int min_units, int min_saved);

namespace {

using GP = V3dGemmParams;

// compile-time loop (bodies that pick between named register arrays must not wait for the late loop unroller: SROA has
// already given up on the arrays by then and they land in scratch)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int MODE>
struct RowInfo {};

template <>
struct RowInfo<V3D_GEMM_LINEAR> {
    long long src;
    bool ok;
    __device__ void init(const GP& p, long long m) {
        ok = m < p.M;
        src = m;
    }
    __device__ bool tap(const GP&, int, long long& s) const {
        s = src;
        return ok;
    }
};

template <>
struct RowInfo<V3D_GEMM_CONV3X3> {
    long long base;
    int iy0, ix0;
    bool ok;
    __device__ void init(const GP& p, long long m) {
        ok = m < p.M;
        long long hw = (long long)p.Hout * p.Wout;
        long long img = m / hw;
        int rem = (int)(m - img * hw);
        int oy = rem / p.Wout;
        int ox = rem - oy * p.Wout;
        base = img * (long long)p.Hin * p.Win;
        iy0 = oy * p.stride - p.pad_lo;
        ix0 = ox * p.stride - p.pad_lo;
    }
    __device__ bool tap(const GP& p, int t, long long& s) const {
        int ky = t / 3, kx = t - ky * 3;
        int iy = iy0 + ky, ix = ix0 + kx;
        bool v = ok && iy >= 0 && ix >= 0 && iy < (p.Hin << p.upshift) && ix < (p.Win << p.upshift);
        s = base + (long long)(iy >> p.upshift) * p.Win + (ix >> p.upshift);
        return v;
    }
};

template <>
struct RowInfo<V3D_GEMM_CONVT3> {
    long long m_;
    long long hoff_;   // split-halo layout only: b * S + s, the row of this (sample, position) inside a halo slab
    int t_;
    bool ok;
    __device__ void init(const GP& p, long long m) {
        ok = m < p.M;
        m_ = m;
        long long frame = m / p.S;
        t_ = (int)(frame % p.T);
        hoff_ = (frame / p.T) * p.S + (m - frame * p.S);
    }
    __device__ bool tap(const GP& p, int t, long long& s) const {
        int tt = t_ + t - 1;
        s = m_ + (long long)(t - 1) * p.S;
        // frame sharding (halo_rows = B * S): frame -1 of every sample lives in the slab in FRONT of the local frames, frame T in
        // the slab BEHIND them, so the +-1 frames of a sample never alias the neighbouring sample's frames and all B samples of
        // a rank go through one launch
        if (p.halo_rows > 0) {
            if (tt < 0) s = hoff_ - p.halo_rows;
            else if (tt >= p.T) s = p.M + hoff_;
        }
        return ok && tt >= p.tmin && tt <= p.tmax;
    }
};

template <int MODE>
constexpr int ntaps() {
    return MODE == V3D_GEMM_LINEAR ? 1 : (MODE == V3D_GEMM_CONV3X3 ? 9 : 3);
}

// XCD-aware bijective remap of a 1-D grid: consecutive logical tiles share an XCD (and its L2)
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// tile id -> (tile row, tile column).  Consecutive ids run concurrently on one XCD (xcd_remap), i.e. share one 4 MiB L2: walking N fastest
// makes 32-64 concurrent tiles of ONE tile row re-stream the whole weight matrix per row of tiles (PMC, profiles/r02a_pmc_per_launch.txt:
// the N = 10240 GEGLU projection fetched 20x its algorithmic bytes).  group_m > 1 walks down `group_m` tile rows before moving to the next
// tile column, so the concurrent set is a group_m x (concurrency / group_m) patch that shares both operands; bijective for any mt, nt.
__device__ __forceinline__ void tile_coords(const GP& p, int id, int& tm, int& tn) {
    const int gm = p.group_m >= 0 ? p.group_m : (p.mt >= 96 ? 8 : 4);     // < 0: heuristic (see the knob comment at the top)
    if (gm <= 1 || p.nt == 1) {
        tn = id % p.nt;
        tm = id / p.nt;
        return;
    }
    const int width = gm * p.nt;
    const int gid = id / width;
    const int first = gid * gm;
    const int gsz = (p.mt - first) < gm ? (p.mt - first) : gm;
    const int r = id - gid * width;
    tn = r / gsz;
    tm = first + (r - tn * gsz);
}

// ---- epilogue shared by both main loops --------------------------------------------------------------------------
// The wave holds MF x NF fragments; per fragment a lane owns 4 consecutive n (= (lane>>4)*4 + r) of pixel m = lane&15.
// bf16 outputs are staged through a wave-private LDS region (`stage`, >= MF*16 rows of (WNout*2 + 16) bytes) and leave as
// 16-byte-per-lane stores covering whole row segments.  Two code paths: a branch-free FAST path for wave tiles that lie
// completely inside the output (every predicate is wave-uniform: float4 bias / per-image vector loads, 8-byte residual
// loads, no per-element bounds checks) and the generic path with per-element predicates for ragged M / N edges.
// (The first version had only the generic path: 1500 VALU + 380 exec-mask branches per wave against 160 MFMAs on the
// K = 320 GEGLU GEMM, see profiles/r01_gemm_ablation.txt.)
// residual #1 rows of one (MF*16) x (NFO*16) wave tile as coalesced 16-byte pieces (the layout the staging buffer uses):
// piece K of lane l is chunk c = K*64 + l -> row c / CPRO, 16-byte column chunk c % CPRO.  The v3 kernels prefetch these one
// epilogue chunk ahead and hand them over BY VALUE (a struct / array handed over by reference went through scratch).
template <int MF, int NF, bool GEGLU>
struct ResGeom {
    static constexpr int NFO = GEGLU ? NF / 2 : NF;
    static constexpr int CPRO = NFO * 2, ROWS = MF * 16, NV = (ROWS * CPRO + 63) / 64;
};
template <int K, int MF, int NF, bool GEGLU>
__device__ __forceinline__ u32x4 load_res_piece(const GP& p, long long mw0, long long nw0, int lane) {
    using G = ResGeom<MF, NF, GEGLU>;
    u32x4 r = {0u, 0u, 0u, 0u};
    if constexpr (K < G::NV) {
        const long long ncol0 = GEGLU ? ((nw0 >> 5) * 16) : nw0;
        const int c = K * 64 + lane;
        if ((G::ROWS * G::CPRO) % 64 == 0 || c < G::ROWS * G::CPRO)
            r = *reinterpret_cast<const u32x4*>(p.res1 + (mw0 + c / G::CPRO) * p.ldr1 + ncol0 + (c % G::CPRO) * 8);
    }
    return r;
}
template <int K, int MF, int NF, bool GEGLU>
__device__ __forceinline__ void res_piece_to_stage(u32x4 r, unsigned char* stage, int srow, int lane) {
    using G = ResGeom<MF, NF, GEGLU>;
    if constexpr (K < G::NV) {
        const int c = K * 64 + lane;
        if ((G::ROWS * G::CPRO) % 64 == 0 || c < G::ROWS * G::CPRO) *reinterpret_cast<u32x4*>(stage + (c / G::CPRO) * srow + (c % G::CPRO) * 16) = r;
    }
}

// ---- GroupNorm statistics of the output, gathered where it is produced (v3 <GN> kernels) ----------------------------------------
// Every ResBlock convolution is followed by a GroupNorm of its output (openaimodel.py:267-271,302-305 / video_model.py:42-55); the
// stand-alone statistics kernel re-reads that tensor from HBM.  Here a wave adds up (sum, sum of squares) of the bf16-ROUNDED values it
// stores - the numbers v3d_groupnorm_stats would read back - per channel over the rows of its tile that belong to one statistics group
// (rows / gn_rps: an image, or the T images of a sample for the 3-D norm): in registers over the row fragments, across the 16 pixel lanes
// with DPP row shifts, through the wave's staging region to one lane per GroupNorm group, then ONE 8-byte store per group into the writer's
// own slot of the [stat][slot][32][2] buffer the stand-alone kernel fills (no atomics: the sums are bit-reproducible).  A writer is a wave
// tile's run of rows inside one statistics group; wave tiles are WM-aligned row ranges, so slot = ceil(first row within the group / WM) is
// unique per writer (the host checks WN % gn_cpg == 0: no group is shared by two wave columns, and gn_nslots >= gn_rps / WM + 2).
template <int NF>
struct GnAcc {
    float s[NF][2], q[NF][2];     // per fragment column j: channel pairs (4 q + 0, 1) and (4 q + 2, 3) of the lane's 4 channels - groups hold an even
};                                // number of channels, so a pair never straddles two of them
template <int NF>
__device__ __forceinline__ void gn_zero(GnAcc<NF>& a) {
#pragma unroll
    for (int j = 0; j < NF; ++j) a.s[j][0] = a.s[j][1] = a.q[j][0] = a.q[j][1] = 0.f;
}
// one packed bf16 pair of the stored tile: sum and sum of squares by v_dot2_f32_bf16 (exact products, fp32 accumulation, no unpacking)
typedef __bf16 gn_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gn_add_pair(float& s, float& q, uint32_t w) {
    const gn_bf16x2 v = __builtin_bit_cast(gn_bf16x2, w), ones = __builtin_bit_cast(gn_bf16x2, 0x3f803f80u);
    s = __builtin_amdgcn_fdot2_f32_bf16(v, ones, s, false);
    q = __builtin_amdgcn_fdot2_f32_bf16(v, v, q, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_row_shr(float v) {    // value of the lane CTRL positions below in the 16-lane row, 0 past the row start
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 | CTRL, 0xf, 0xf, true));
}
template <int NF>
__device__ __forceinline__ void gn_flush(const GP& p, GnAcc<NF>& a, long long sid, long long nw0, int lane, unsigned char* stage, unsigned slot) {
    float* sf = reinterpret_cast<float*>(stage);
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        f32x4 t = {a.s[j][0], a.s[j][1], a.q[j][0], a.q[j][1]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = t[e];
            v += dpp_row_shr<1>(v);
            v += dpp_row_shr<2>(v);
            v += dpp_row_shr<4>(v);
            v += dpp_row_shr<8>(v);     // lane 15 of each row: the row's total
            t[e] = v;
        }
        // 4 lanes: channel pairs of channels j * 16 + (lane >> 4) * 4 .. + 3 of the wave tile, as {s01, s23, q01, q23}
        if ((lane & 15) == 15) lds_store16_nowait(sf + (j * 4 + (lane >> 4)) * 4, t);
    }
    // one lane per GroupNorm group that intersects the wave tile's channels [nw0, nw0 + NF * 16)
    const int cpg = p.gn_cpg;
    const int g_lo = (int)(nw0 / cpg), g_hi = (int)((nw0 + NF * 16 - 1) / cpg);
    const int g = g_lo + lane;
    if (g <= g_hi) {
        const int c0 = g * cpg > (int)nw0 ? g * cpg - (int)nw0 : 0;
        const int c1 = (g + 1) * cpg - (int)nw0 < NF * 16 ? (g + 1) * cpg - (int)nw0 : NF * 16;
        float sa = 0.f, sb = 0.f;
        for (int cp = c0 >> 1; cp < (c1 >> 1); ++cp) {            // channel pair cp = channels 2 cp, 2 cp + 1 of the tile
#if E4_ASM_READS
            // (asm reads + an asm wait that names them: a ds_read the compiler can see gets s_waitcnt vmcnt(0) in front of it while LDS-DMA is in flight)
            const unsigned ea = lds_addr(sf + (cp >> 1) * 4 + (cp & 1));
            float e0, e2;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(e0), "=&v"(e2) : "v"(ea) : "memory");
#else
            const float* e = sf + (cp >> 1) * 4 + (cp & 1);
            const float e0 = e[0], e2 = e[2];
#endif
            sa += e0;
            sb += e2;
        }
        float* dst = p.gn_stats + ((sid * p.gn_nslots + slot) * 32 + g) * 2;
        *reinterpret_cast<float2*>(dst) = make_float2(sa, sb);
    }
    gn_zero(a);
}

template <int MF, int NF, bool GEGLU, bool CAN_STAGE, bool FAST_ONLY = false, int SPAD = 16, bool GN = false>
__device__ __forceinline__ void epilogue(const GP& p, f32x4 (&acc)[MF][NF], long long mw0, long long nw0, long long z, int lane,
                                         unsigned char* stage, u32x4 pre0, u32x4 pre1, u32x4 pre2, bool res_pre,
                                         const float4 (&bias_pre)[NF], bool has_bias_pre, GnAcc<GN ? NF : 1>* gn = nullptr) {
    if V3D_ABL(p, 1) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (sum == 123.456f) reinterpret_cast<float*>(p.out)[0] = sum;   // keeps the accumulators live
        return;
    }
    const long long Nout = GEGLU ? p.N / 2 : p.N;
    constexpr int NFO = GEGLU ? NF / 2 : NF;          // output fragments per wave-tile row
    constexpr int SROW = NFO * 32 + SPAD;             // staging row stride in bytes (16 B pad unless LDS is too tight)
    const long long ncol0 = GEGLU ? ((nw0 >> 5) * 16) : nw0;   // first output column of this wave tile
    const bool staged = CAN_STAGE && !p.out_fp32 && (p.ldo % 8 == 0) && (ncol0 % 8 == 0) &&
                        ((reinterpret_cast<uintptr_t>(p.out) + (size_t)z * p.sO * 2) % 16 == 0);
    const bool vec_ok = (p.ldo % 4 == 0) && (!p.res1 || p.ldr1 % 4 == 0) && (!p.res2 || p.ldr2 % 4 == 0);
    const bool fast = (mw0 + MF * 16 <= p.M) && (nw0 + NF * 16 <= p.N) && vec_ok && (staged || p.out_fp32) &&
                      (!p.add || ((reinterpret_cast<uintptr_t>(p.add) % 16 == 0) && (p.add_ld % 4 == 0))) &&
                      (!p.res1 || reinterpret_cast<uintptr_t>(p.res1) % 8 == 0) && (!p.res2 || reinterpret_cast<uintptr_t>(p.res2) % 8 == 0);
    const int fr = lane & 15, fq = (lane >> 4) * 4;
    if (FAST_ONLY && (mw0 >= p.M || nw0 >= p.N)) return;   // wave tile completely outside a partial edge tile
    if (FAST_ONLY || fast) {   // FAST_ONLY: the host has checked the fast-path conditions for every wave tile (v3 kernels)
        const int nb = (int)nw0 + fq;                  // this lane's first packed weight row (tile-relative math in 32 bit)
        constexpr int CPRO = NFO * 2;                  // 16-byte chunks per staged row
        constexpr int ROWS = MF * 16;
        constexpr bool WHOLE = (ROWS * CPRO) % 64 == 0;   // else the last wave-wide copy of the staged tile is partial
        // residual #1 comes in through the staging buffer: coalesced 16-byte row loads -> LDS, then each lane picks its 8 bytes in
        // MFMA layout (the direct 8-byte-per-lane residual loads touched 16 rows per instruction: 129 us vs 70 us for the
        // attention out-projection at 64x64, profiles/r01d_op_times_unet_eval.txt)
        const bool res1_lds = CAN_STAGE && p.res1 && !p.out_fp32 && (p.ldr1 % 8 == 0) && (reinterpret_cast<uintptr_t>(p.res1) % 16 == 0);
        if (CAN_STAGE && res1_lds) {
            if (res_pre) {   // (only offered by callers whose wave-tile chunk has at most 3 pieces per lane)
                res_piece_to_stage<0, MF, NF, GEGLU>(pre0, stage, SROW, lane);
                res_piece_to_stage<1, MF, NF, GEGLU>(pre1, stage, SROW, lane);
                res_piece_to_stage<2, MF, NF, GEGLU>(pre2, stage, SROW, lane);
            } else {
                const bf16_t* rz = p.res1 + mw0 * p.ldr1 + ncol0;
                uint4 rv[(ROWS * CPRO + 63) / 64];
#pragma unroll
                for (int c0 = 0; c0 < ROWS * CPRO; c0 += 64) {
                    const int c = c0 + lane;
                    if (WHOLE || c < ROWS * CPRO) rv[c0 / 64] = *reinterpret_cast<const uint4*>(rz + (long long)(c / CPRO) * p.ldr1 + (c % CPRO) * 8);
                }
#pragma unroll
                for (int c0 = 0; c0 < ROWS * CPRO; c0 += 64) {
                    const int c = c0 + lane;
                    if (WHOLE || c < ROWS * CPRO) *reinterpret_cast<uint4*>(stage + (c / CPRO) * SROW + (c % CPRO) * 16) = rv[c0 / 64];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        const bool scale_acc = p.coef != nullptr || p.c_acc != 1.0f;   // wave-uniform
        float4 bv[NF];
        if (has_bias_pre) {   // the caller loaded this wave tile's bias columns once for all of its row chunks
#pragma unroll
            for (int j = 0; j < NF; ++j) bv[j] = bias_pre[j];
        } else if (p.bias) {
#pragma unroll
            for (int j = 0; j < NF; ++j) bv[j] = *reinterpret_cast<const float4*>(p.bias + nb + j * 16);
        } else {
#pragma unroll
            for (int j = 0; j < NF; ++j) bv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const long long m = mw0 + i * 16 + fr;
            float ca = p.c_acc, c1 = p.c_res1, c2 = p.c_res2;
            if (p.coef) {
                const float* cf = p.coef + (m / p.coef_rpg) * 3;
                ca = cf[0]; c1 = cf[1]; c2 = cf[2];
            }
            const float* addv = p.add ? p.add + (m / p.add_rpg) * p.add_ld + nb : nullptr;
            const bf16_t* r1 = (p.res1 && !res1_lds) ? p.res1 + m * p.ldr1 + (int)ncol0 + fq : nullptr;
            const bf16_t* r2 = p.res2 ? p.res2 + m * p.ldr2 + (int)ncol0 + fq : nullptr;
            float* of = p.out_fp32 ? reinterpret_cast<float*>(p.out) + z * p.sO + m * p.ldo + (int)ncol0 + fq : nullptr;
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                if (GEGLU && (j & 1)) continue;
                float v[4] = {acc[i][j][0] + bv[j].x, acc[i][j][1] + bv[j].y, acc[i][j][2] + bv[j].z, acc[i][j][3] + bv[j].w};
                if (addv) {
                    const float4 a = *reinterpret_cast<const float4*>(addv + j * 16);
                    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
                }
                if (GEGLU) {
                    constexpr int dummy = 0;
                    const int jg = (j + 1 < NF) ? j + 1 : j;
                    float g[4] = {acc[i][jg][0] + bv[jg].x, acc[i][jg][1] + bv[jg].y, acc[i][jg][2] + bv[jg].z, acc[i][jg][3] + bv[jg].w};
                    if (addv) {
                        const float4 a = *reinterpret_cast<const float4*>(addv + jg * 16);
                        g[0] += a.x; g[1] += a.y; g[2] += a.z; g[3] += a.w;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = V3D_ABL(p, 16) ? v[r] * g[r] : geglu_mul(v[r], g[r]);
                    (void)dummy;
                }
                const int jo = GEGLU ? (j >> 1) : j;
                float o[4] = {v[0], v[1], v[2], v[3]};
                if (scale_acc) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] *= ca;
                }
                if (CAN_STAGE && res1_lds) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(stage + (i * 16 + fr) * SROW + jo * 32 + fq * 2);
                    o[0] += c1 * bflo(rr.x); o[1] += c1 * bfhi(rr.x); o[2] += c1 * bflo(rr.y); o[3] += c1 * bfhi(rr.y);
                } else if (r1) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(r1 + jo * 16);
                    o[0] += c1 * bflo(rr.x); o[1] += c1 * bfhi(rr.x); o[2] += c1 * bflo(rr.y); o[3] += c1 * bfhi(rr.y);
                }
                if (r2) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(r2 + jo * 16);
                    o[0] += c2 * bflo(rr.x); o[1] += c2 * bfhi(rr.x); o[2] += c2 * bflo(rr.y); o[3] += c2 * bfhi(rr.y);
                }
                if (of) {
                    *reinterpret_cast<float4*>(of + jo * 16) = make_float4(o[0], o[1], o[2], o[3]);
                } else if (CAN_STAGE) {
                    const uint32_t w0 = pack2bf(o[0], o[1]), w1 = pack2bf(o[2], o[3]);
                    *reinterpret_cast<uint2*>(stage + (i * 16 + fr) * SROW + jo * 32 + fq * 2) = make_uint2(w0, w1);
                    if constexpr (GN) {
                        gn_add_pair(gn->s[j][0], gn->q[j][0], w0);
                        gn_add_pair(gn->s[j][1], gn->q[j][1], w1);
                    }
                }
            }
        }
        if (CAN_STAGE && !p.out_fp32) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bf16_t* outz = reinterpret_cast<bf16_t*>(p.out) + z * p.sO + mw0 * p.ldo + ncol0;
#pragma unroll
            for (int c0 = 0; c0 < ROWS * CPRO; c0 += 64) {
                const int c = c0 + lane;
                const int row = c / CPRO, ch = c % CPRO;
                if ((WHOLE || c < ROWS * CPRO) && !V3D_ABL(p, 32)) *reinterpret_cast<uint4*>(outz + (long long)row * p.ldo + ch * 8) = *reinterpret_cast<const uint4*>(stage + row * SROW + ch * 16);
            }
        }
        return;
    }
    if (FAST_ONLY) return;
    // ---------------- generic path (ragged edges) ----------------
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const long long m = mw0 + i * 16 + fr;
        if (m >= p.M) continue;
        float ca = p.c_acc, c1 = p.c_res1, c2 = p.c_res2;
        if (p.coef) {
            const float* cf = p.coef + (m / p.coef_rpg) * 3;
            ca = cf[0];
            c1 = cf[1];
            c2 = cf[2];
        }
        const float* addv = p.add ? p.add + (m / p.add_rpg) * p.add_ld : nullptr;
#pragma unroll
        for (int j = 0; j < NF; j += 1) {
            if (GEGLU && (j & 1)) continue;  // gate fragments are consumed with their value fragment
            const long long np = nw0 + j * 16 + fq;  // packed weight-row index
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r];
                if (np + r < p.N) {
                    if (p.bias) v[r] += p.bias[np + r];
                    if (addv) v[r] += addv[np + r];
                }
            }
            long long col = np;
            if (GEGLU) {
                const int jg = (j + 1 < NF) ? j + 1 : j;
                const long long ng = np + 16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float g = acc[i][jg][r];
                    if (ng + r < p.N) {
                        if (p.bias) g += p.bias[ng + r];
                        if (addv) g += addv[ng + r];
                    }
                    v[r] = v[r] * gelu_erf_f(g);
                }
                col = (np >> 5) * 16 + (np & 15);
            }
            if (col >= Nout) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (col + r < Nout) {
                    float o = ca * v[r];
                    if (p.res1) o += c1 * bf2f(p.res1[m * p.ldr1 + col + r]);
                    if (p.res2) o += c2 * bf2f(p.res2[m * p.ldr2 + col + r]);
                    if (p.out_fp32)
                        (reinterpret_cast<float*>(p.out) + z * p.sO + m * p.ldo + col)[r] = o;
                    else
                        (reinterpret_cast<bf16_t*>(p.out) + z * p.sO + m * p.ldo + col)[r] = f2bf(o);
                }
            }
        }
    }
}

template <int MF, int NF, bool GEGLU, bool CAN_STAGE, bool FAST_ONLY = false>
__device__ __forceinline__ void epilogue(const GP& p, f32x4 (&acc)[MF][NF], long long mw0, long long nw0, long long z, int lane,
                                         unsigned char* stage) {
    const u32x4 none = {0u, 0u, 0u, 0u};
    float4 nob[NF];
    epilogue<MF, NF, GEGLU, CAN_STAGE, FAST_ONLY>(p, acc, mw0, nw0, z, lane, stage, none, none, none, false, nob, false);
}


// retire the finished tile of a v3 wave in chunks of EMF row fragments; residual rows come in one chunk ahead of their use
// (a per-chunk load -> LDS -> use chain exposed the full load latency 8 times per tile: the [bar] out-projections ran
// 15-50 % slower than on v2)
template <int C, int NCH, int EMF, int NF, bool GEGLU, int SPAD, bool GN>
__device__ __forceinline__ void v3_retire_chunks(const GP& p, f32x4 (&acc)[NCH * EMF][NF], long long mw0, long long nw0, int lane, unsigned char* estage,
                                                 u32x4 c0, u32x4 c1, u32x4 c2, bool res_pre, const float4 (&bv)[NF], GnAcc<GN ? NF : 1>& gn,
                                                 long long sid0, unsigned rem0) {
    static_assert(ResGeom<EMF, NF, GEGLU>::NV <= 3, "v3 epilogue chunk: at most 3 residual pieces per lane");
    if constexpr (C < NCH) {
        u32x4 n0 = c0, n1 = c1, n2 = c2;
        if constexpr (C + 1 < NCH) {
            if (res_pre) {
                n0 = load_res_piece<0, EMF, NF, GEGLU>(p, mw0 + (C + 1) * EMF * 16, nw0, lane);
                n1 = load_res_piece<1, EMF, NF, GEGLU>(p, mw0 + (C + 1) * EMF * 16, nw0, lane);
                n2 = load_res_piece<2, EMF, NF, GEGLU>(p, mw0 + (C + 1) * EMF * 16, nw0, lane);
            }
        }
        epilogue<EMF, NF, GEGLU, true, true, SPAD, GN>(p, *reinterpret_cast<f32x4(*)[EMF][NF]>(&acc[C * EMF]), mw0 + C * EMF * 16, nw0, 0, lane, estage, c0, c1, c2, res_pre, bv, true, &gn);
        if constexpr (GN) {
            // rows of this chunk belong to statistics group sid0 + (rem0 + C * 16) / gn_rps; hand the sums over when the next chunk
            // starts another group (a tile may straddle images: 4096 rows per image, 96 per wave tile) or the tile ends
            static_assert(EMF == 1, "GN epilogue: one row fragment per chunk");
            const unsigned rps = (unsigned)p.gn_rps;
            const unsigned here = (rem0 + C * 16) / rps, next = (rem0 + (C + 1) * 16) / rps;
            // the run of rows that ends here started at the wave tile's first row (first group of the tile) or at the group's first row
            const unsigned off = here == rem0 / rps ? rem0 - here * rps : 0u;
            if (C + 1 == NCH || next != here) gn_flush<NF>(p, gn, sid0 + here, nw0, lane, estage, (off + NCH * 16 - 1) / (NCH * 16));
        }
        v3_retire_chunks<C + 1, NCH, EMF, NF, GEGLU, SPAD, GN>(p, acc, mw0, nw0, lane, estage, n0, n1, n2, res_pre, bv, gn, sid0, rem0);
    }
}


// =====================================================================================================================================
// Hand-managed epilogue of the persistent kernels (round 3): same arithmetic as epilogue<> above for the branch-free case, but every
// memory access the compiler would wrap in a wait is taken out of its hands.
// Why: while LDS-DMA (buffer_load ... lds) is in flight hipcc answers every ordinary VGPR-destination load with s_waitcnt vmcnt(0) at its
// first use and puts another in front of every ds_write to the LDS object the DMA targets.  In the epilogue that turned each of the ~10
// loads / staging writes per 16-row fragment (per-image vector, residual pieces, blend coefficients, spilled scalars) into its own
// full memory round trip - 60-90 serialised round trips per tile; rocprof / disassembly: three tile epilogues cost 90 of the 319 us of a
// 64x64 convolution (profiles/r03_conv_ablation.txt), and the K = 320 linears (10 main-loop steps per tile) were mostly epilogue.
// Here: loads are inline asm (the compiler neither counts nor waits for them), issued ONE FRAGMENT AHEAD of their use; staging writes are
// inline asm; per fragment there is exactly one s_waitcnt vmcnt(0), placed after the fragment's arithmetic and before its stores, when
// everything outstanding (the previous fragment's stores, the next fragment's loads) is at least a fragment's work old.  Per-row-group
// constants (bias + per-image vector, blend coefficients) are loaded once per wave tile and again only when a fragment enters another row group.
// Host contract (e4_ok): bf16 output, whole wave tiles, ldo % 8 == 0, out 16-B aligned; add 16-B aligned, add_ld % 4 == 0, add_rpg % 16 == 0;
// res1 / res2 16-B aligned with row strides % 8 == 0; coef_rpg % 16 == 0.
struct E4Res {
    u32x4 a0, a1, a2;      // residual #1: 16-byte row pieces of one fragment (piece k of lane l = chunk k * 64 + l of the 16 x (NF * 2) chunk grid)
};                         // (residual #2 - the alpha blend behind the 32x32 / 16x16 feed-forwards, 10 launches per evaluation - is fetched where it is used)
template <int NF>
struct E4Tile {
    f32x4 ba[NF];          // bias + per-row-group vector of the lane's 4 channels per fragment column
    float ca, c1, c2;
    // row group of the fragment the constants belong to, and the fragment's first row inside it.  Round 3 compared m0f / add_rpg (and coef_rpg,
    // gn_rps) against these per FRAGMENT: five or six 64-bit software divisions, ~250 of the ~500 instructions a fragment's epilogue
    // executed.  Now one 32-bit division per divisor and tile; fragments advance the remainders (rows of a launch fit 32 bits: host contract).
    unsigned add_grp, coef_grp, add_rem, coef_rem;
};
__device__ __forceinline__ unsigned e4_udiv(unsigned a, unsigned b) {       // wave-uniform operands: one v_rcp-based 32-bit division, result back in an SGPR
    return (unsigned)__builtin_amdgcn_readfirstlane((int)(a / b));
}
__device__ __forceinline__ u32x4 e4_load16(const void* ptr) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
    return r;
}
__device__ __forceinline__ float e4_load4(const float* ptr) {
    float r;
    asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
    return r;
}
template <int NF>
__device__ __forceinline__ void e4_load_res(const GP& p, long long m0f, long long nw0, int lane, E4Res& r) {
    constexpr int CPRO = NF * 2, NP = 16 * CPRO;       // 16-byte chunks per row / per fragment
    auto piece = [&](const bf16_t* base, long long ld, int k) __attribute__((always_inline)) -> u32x4 {
        int c = k * 64 + lane;
        if (NP % 64 != 0 && c >= NP) c = lane;                              // (lanes past the grid: any valid address, the value is never used)
        return e4_load16(base + (m0f + c / CPRO) * ld + nw0 + (c % CPRO) * 8);
    };
    r.a0 = piece(p.res1, p.ldr1, 0);
    r.a1 = piece(p.res1, p.ldr1, 1);
    if constexpr (NP > 128) r.a2 = piece(p.res1, p.ldr1, 2);
}
__device__ __forceinline__ void e4_wait_res(E4Res& r) {          // everything outstanding has landed; names the destinations (asm loads: form (ii))
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.a0), "+v"(r.a1), "+v"(r.a2)::"memory");
}
// the pieces of `r` have landed once at most N younger VMEM operations of this wave are outstanding (VMEM retires in issue order); the
// statement names the destinations like e4_wait_res.  N is a LOWER bound of the operations issued behind the loads (an operation the
// count does not know about - a GroupNorm-statistics store, a reloaded constant - only makes the wait stricter).
template <int N>
__device__ __forceinline__ void e4_wait_cnt(E4Res& r) {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(r.a0), "+v"(r.a1), "+v"(r.a2) : "n"(N) : "memory");
}
// per-row-group constants of the wave tile's fragment starting at row m0f (synchronous: once per tile, and when a fragment enters another group)
template <int NF>
__device__ __forceinline__ void e4_tile_consts(const GP& p, long long m0f, long long nw0, int lane, E4Tile<NF>& t) {
    const int nb = (int)nw0 + (lane >> 4) * 4;
    f32x4 bv[NF], av[NF];
    float cf0 = p.c_acc, cf1 = p.c_res1, cf2 = p.c_res2;
#pragma unroll
    for (int j = 0; j < NF; ++j) bv[j] = av[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < NF; ++j) bv[j] = __builtin_bit_cast(f32x4, e4_load16(p.bias + nb + j * 16));
    }
    t.add_grp = p.add ? e4_udiv((unsigned)m0f, (unsigned)p.add_rpg) : 0u;
    t.add_rem = p.add ? (unsigned)m0f - t.add_grp * (unsigned)p.add_rpg : 0u;
    if (p.add) {
        const float* av0 = p.add + (long long)t.add_grp * p.add_ld + nb;
#pragma unroll
        for (int j = 0; j < NF; ++j) av[j] = __builtin_bit_cast(f32x4, e4_load16(av0 + j * 16));
    }
    t.coef_grp = p.coef ? e4_udiv((unsigned)m0f, (unsigned)p.coef_rpg) : 0u;
    t.coef_rem = p.coef ? (unsigned)m0f - t.coef_grp * (unsigned)p.coef_rpg : 0u;
    if (p.coef) {
        const float* cf = p.coef + (long long)t.coef_grp * 3;
        cf0 = e4_load4(cf);
        cf1 = e4_load4(cf + 1);
        cf2 = e4_load4(cf + 2);
    }
    // one wait for all of them (the statement names every destination: nothing above may be read, copied or spilled before it)
    static_assert(NF == 4 || NF == 5, "e4: 64- or 80-channel wave tiles");
    if constexpr (NF == 5)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]),
                     "+v"(av[4]), "+v"(cf0), "+v"(cf1), "+v"(cf2)::"memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(cf0),
                     "+v"(cf1), "+v"(cf2)::"memory");
#pragma unroll
    for (int j = 0; j < NF; ++j) t.ba[j] = bv[j] + av[j];
    t.ca = cf0;
    t.c1 = cf1;
    t.c2 = cf2;
}
// LDS READS of the staging rows as inline asm too (round 6).  With LDS-DMA in flight hipcc answers every ds_read IT can see in this code with s_waitcnt vmcnt(0)
// (it cannot tell the wave-private staging rows from the ring the DMA writes): five per fragment in front of the residual reads and three in front of the row reads -
// each a full drain of the wave's stores, look-ahead residual loads and ring prefetches, i.e. the memory round trip per fragment the round-6 timeline shows
// (profiles/r06_timeline_k320_bar.txt: 5.5 k cycles per fragment).  Rounds 3-5 believed this path wait-free; the disassembly has 61 vmcnt(0) per kernel.
// The asm reads are followed by an asm lgkmcnt(0) that NAMES their destinations (the compiler neither counts them nor may touch the registers before it).
__device__ __forceinline__ void e4_lds_read8(u32x2& dst, unsigned addr) { asm volatile("ds_read_b64 %0, %1" : "=v"(dst) : "v"(addr) : "memory"); }
__device__ __forceinline__ void e4_lds_read16(u32x4& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory"); }
__device__ __forceinline__ void e4_lds_write16(unsigned addr, u32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void e4_lds_write8(unsigned addr, uint32_t w0, uint32_t w1) {
    const u32x2 v = {w0, w1};
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// one 16-row fragment: acc (+ constants, residuals) -> bf16 -> staged rows -> 16-byte-per-lane row stores.  `cur` = this fragment's residual
// pieces (landed: the caller has waited for them).  Residual #1 goes through the staging rows IN PLACE: its row pieces are written where the
// output rows will stand, every lane reads its own 8 bytes per fragment column, adds, and writes the bf16 result back to the same 8 bytes
// (4 live floats per column instead of the whole fragment).  NOTHING here waits for a store: the rows leave as plain 16-byte stores that
// drain while the next fragments are computed (round 3 waited vmcnt(0) once per fragment: the previous fragment's stores and the next
// fragment's residual loads - 6 serialised memory round trips per wave tile with every CU of the chip in the same phase).  That sentence became
// true in round 6 only, with RD != 0: until then the compiler put its own vmcnt(0) in front of every staging read (see e4_lds_read8).
template <int NF, bool GN, int RD = E4_ASM_READS>
__device__ __forceinline__ void e4_fragment(const GP& p, f32x4 (&acc)[NF], long long m0f, long long nw0, int lane, unsigned char* stage, const E4Res& cur,
                                            const E4Tile<NF>& t, GnAcc<GN ? NF : 1>& gn) {
    constexpr int CPRO = NF * 2, NP = 16 * CPRO, SROW = NF * 32 + 16;
    const int fr = lane & 15, fq = (lane >> 4) * 4;
    const unsigned sbase = lds_addr(stage);
    auto piece_addr = [&](int k) __attribute__((always_inline)) -> unsigned {
        const int c = k * 64 + lane;
        return sbase + (unsigned)((c / CPRO) * SROW + (c % CPRO) * 16);
    };
    const int mine_off = fr * SROW + fq * 2;                               // the lane's 8 bytes of fragment column j sit at mine + j * 32
    const unsigned mine = sbase + (unsigned)mine_off;
    const bool has1 = p.res1 != nullptr, has2 = p.res2 != nullptr;
    if (has1) {
        e4_lds_write16(piece_addr(0), cur.a0);
        e4_lds_write16(piece_addr(1), cur.a1);
        if constexpr (NP > 128) {
            if (NP % 64 == 0 || lane < NP - 128) e4_lds_write16(piece_addr(2), cur.a2);
        }
    }
    // residual #2 (rare): the lane's 8 bytes per fragment column straight from memory (MFMA layout), fetched and waited for here (the
    // vmcnt(0) also lands whatever else the wave has in flight: stricter than the counted waits of e4_retire need, never weaker)
    u32x2 q2[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) q2[j] = u32x2{0u, 0u};
    if (has2) {
        const bf16_t* r2 = p.res2 + (m0f + fr) * p.ldr2 + nw0 + fq;
#pragma unroll
        for (int j = 0; j < NF; ++j) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(q2[j]) : "v"(r2 + j * 16) : "memory");
        static_assert(NF == 4 || NF == 5, "e4: 64- or 80-channel wave tiles");
        if constexpr (NF == 5)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(q2[0]), "+v"(q2[1]), "+v"(q2[2]), "+v"(q2[3]), "+v"(q2[4])::"memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(q2[0]), "+v"(q2[1]), "+v"(q2[2]), "+v"(q2[3])::"memory");
    }
    if constexpr (RD != 0) {
    // E4_ASM_READS 1: the fragment's NF residual pieces in one LDS round trip (2 NF registers); 2: one piece per round trip (2 registers: the LDS-haloed kernels
    // of conv.hip sit at 256 registers and spill inside their main loops with the batched form)
    u32x2 rr[RD == 1 ? NF : 1];
#pragma unroll
    for (int j = 0; j < (RD == 1 ? NF : 1); ++j) rr[j] = u32x2{0u, 0u};
    if (has1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (RD == 1) {
#pragma unroll
            for (int j = 0; j < NF; ++j) e4_lds_read8(rr[j], mine + j * 32);
            static_assert(NF == 4 || NF == 5, "e4: 64- or 80-channel wave tiles");
            if constexpr (NF == 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rr[0]), "+v"(rr[1]), "+v"(rr[2]), "+v"(rr[3]), "+v"(rr[RD == 1 ? 4 : 0])::"memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rr[0]), "+v"(rr[1]), "+v"(rr[2]), "+v"(rr[3])::"memory");
        }
    }
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (acc[j][r] + t.ba[j][r]) * t.ca;
        if (has1) {
            const int jr = RD == 1 ? j : 0;
            if constexpr (RD != 1) {
                e4_lds_read8(rr[0], mine + j * 32);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rr[0])::"memory");
            }
            o[0] += t.c1 * bflo(rr[jr][0]); o[1] += t.c1 * bfhi(rr[jr][0]); o[2] += t.c1 * bflo(rr[jr][1]); o[3] += t.c1 * bfhi(rr[jr][1]);
        }
        if (has2) {
            o[0] += t.c2 * bflo(q2[j][0]); o[1] += t.c2 * bfhi(q2[j][0]); o[2] += t.c2 * bflo(q2[j][1]); o[3] += t.c2 * bfhi(q2[j][1]);
        }
        const uint32_t w0 = pack2bf(o[0], o[1]), w1 = pack2bf(o[2], o[3]);
        e4_lds_write8(mine + j * 32, w0, w1);
        if constexpr (GN) {
            gn_add_pair(gn.s[j][0], gn.q[j][0], w0);
            gn_add_pair(gn.s[j][1], gn.q[j][1], w1);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bf16_t* outz = reinterpret_cast<bf16_t*>(p.out) + m0f * p.ldo + nw0;
    constexpr int NK = (NP + 63) / 64;
    static_assert(NK == 2 || NK == 3, "row pieces per lane");
    // (one piece at a time: three in flight cost 8 more registers, and the LDS-haloed kernels have none to spare - their main loops spilled)
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int c = k * 64 + lane;
        int cr = c;
        if (NP % 64 != 0 && cr >= NP) cr = lane;                   // (lanes past the grid read a valid address; the value is not stored)
        u32x4 sr;
        e4_lds_read16(sr, sbase + (unsigned)((cr / CPRO) * SROW + (cr % CPRO) * 16));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sr)::"memory");
        const int row = c / CPRO, ch = c % CPRO;
        if ((NP % 64 == 0 || c < NP) && !V3D_ABL(p, 1)) *reinterpret_cast<u32x4*>(outz + (long long)row * p.ldo + ch * 8) = sr;
    }
    } else {
    // (RD = 0: compiler-visible reads, each behind the compiler's vmcnt(0) - the 3 x 3 LDS-haloed kernels of conv.hip: their tiles are 90-360 steps long and the asm forms measured +-1 %)
    if (has1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (acc[j][r] + t.ba[j][r]) * t.ca;
        if (has1) {
            const uint2 rr = *reinterpret_cast<const uint2*>(stage + mine_off + j * 32);
            o[0] += t.c1 * bflo(rr.x); o[1] += t.c1 * bfhi(rr.x); o[2] += t.c1 * bflo(rr.y); o[3] += t.c1 * bfhi(rr.y);
        }
        if (has2) {
            o[0] += t.c2 * bflo(q2[j][0]); o[1] += t.c2 * bfhi(q2[j][0]); o[2] += t.c2 * bflo(q2[j][1]); o[3] += t.c2 * bfhi(q2[j][1]);
        }
        const uint32_t w0 = pack2bf(o[0], o[1]), w1 = pack2bf(o[2], o[3]);
        e4_lds_write8(mine + j * 32, w0, w1);
        if constexpr (GN) {
            gn_add_pair(gn.s[j][0], gn.q[j][0], w0);
            gn_add_pair(gn.s[j][1], gn.q[j][1], w1);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bf16_t* outz = reinterpret_cast<bf16_t*>(p.out) + m0f * p.ldo + nw0;
#pragma unroll
    for (int k = 0; k < (NP + 63) / 64; ++k) {
        const int c = k * 64 + lane;
        const int row = c / CPRO, ch = c % CPRO;
        if ((NP % 64 == 0 || c < NP) && !V3D_ABL(p, 1)) *reinterpret_cast<uint4*>(outz + (long long)row * p.ldo + ch * 8) = *reinterpret_cast<const uint4*>(stage + row * SROW + ch * 16);
    }
    }
}

// Residual pieces run E4_DEPTH fragments ahead of their use (round 3: one fragment, waited for together with the previous fragment's stores).
// A fragment's arithmetic is ~0.2 us, a loaded memory round trip 1-2 us: with the pieces of three fragments in flight (36 registers - the
// main loop's operand fragments are dead here) the wave pays the latency once per tile, in the constants' wait, not once per fragment.
#ifndef E4_DEPTH
#define E4_DEPTH 1
#endif
// the LINEAR v3 kernels have the registers for a deeper look-ahead (248-254 VGPRs, no spills at depth 2 / 3; the multi-tap loaders spill: depth 1)
#ifndef E4_DEPTH_LINEAR
#define E4_DEPTH_LINEAR 1
#endif
// (v6 ran depth 2 while the compiler's drains were in place: -2 % then; with the asm staging reads depth 1 measures the same or 1-3 % better per launch and
// keeps one set of in-flight registers less in rotation - profiles/r06_e4_asm_reads_ab.txt)
#ifndef E4_DEPTH_V6
#define E4_DEPTH_V6 1
#endif
template <int NF>
struct E4Cnt {
    static constexpr int NP = 16 * NF * 2;
    static constexpr int LOADS = NP > 128 ? 3 : 2;       // asm loads per fragment (e4_load_res)
    static constexpr int STORES = (NP + 63) / 64;        // 16-byte row stores per fragment (e4_fragment)
};
// GroupNorm-statistics writer bookkeeping of a wave tile whose fragments are CONSECUTIVE 16-row runs (v3 / v4 kernels, the 3x3 haloed kernel): the
// statistics group of the current fragment and its first row inside it, advanced per fragment (called once per fragment, in order).  A writer
// = the wave tile's run of rows inside one statistics group; slot = ceil(first row of the run inside the group / wave-tile rows).
template <int WMR, int MF>
struct E4GnRun {
    unsigned sid, rem, sid0, rem0, rps;
    __device__ __forceinline__ void init(long long mw0, long long gn_rps) {
        rps = (unsigned)gn_rps;
        sid0 = sid = rps ? e4_udiv((unsigned)mw0, rps) : 0u;
        rem0 = rem = (unsigned)mw0 - sid * rps;
    }
    // -> flush after this fragment?  (its group ends with it, or the tile does)
    __device__ __forceinline__ bool step(int f, unsigned& slot, unsigned& sid_out) {
        const unsigned first = sid == sid0 ? rem0 : 0u;
        slot = (first + WMR - 1) / WMR;
        sid_out = sid;
        rem += 16;
        const bool ends = rem >= rps;
        if (ends) {
            rem -= rps;
            ++sid;
        }
        return f + 1 == MF || ends;
    }
};
// retire the MF fragments of a finished wave tile; RowFn(f) = first output row of fragment f, FlushFn(f, m0f, slot&, sid&) = flush? of the
// GroupNorm-statistics epilogue.  Residual pieces travel by value (a reference went through a stack array, see conv.hip halo_retire):
// cur / n1 / n2 = the pieces of fragments F, F + 1, F + 2 (as far as E4_DEPTH reaches; the rest are dead values).
// accumulator source: get<F>(out) hands over the NF fragments of row fragment F.  Register-array form (v3 kernels, conv.hip):
template <int MF, int NF>
struct E4AccArray {
    f32x4 (&a)[MF][NF];
    template <int F>
    __device__ __forceinline__ void get(f32x4 (&out)[NF]) const {
#pragma unroll
        for (int j = 0; j < NF; ++j) out[j] = a[F][j];
    }
};
template <int F, int MF, int NF, bool GN, int D, int RD, typename Acc, typename RowFn, typename FlushFn>
__device__ __forceinline__ void e4_retire(const GP& p, const Acc& acc, long long nw0, int lane, unsigned char* stage, E4Res cur, E4Res n1, E4Res n2,
                                          E4Tile<NF> t, GnAcc<GN ? NF : 1>& gn, RowFn rowfn, FlushFn flushfn) {
    if constexpr (F < MF) {
        static_assert(D >= 1 && D <= 3, "residual look-ahead: 1..3 fragments");
        // depth 3 needs the compiler-visible reads (RD = 0): without their drains the register allocator's copies of the rotating piece sets (cur <- n1 <- n2 <- n3) read
        // registers whose asm load is still in flight - wrong results in profiles/r06_e4_asm_reads_ab.txt, found by tools/check_inflight_regs.py; depths 1 and 2 are clean
        static_assert(D <= 2 || RD == 0, "e4: look-ahead depth 3 only with compiler-visible staging reads");
        const long long m0f = rowfn(F);
        const bool has1 = p.res1 != nullptr;
        E4Res n3 = n2;
        if constexpr (F + D < MF) {
            if (has1) e4_load_res<NF>(p, rowfn(F + D), nw0, lane, D == 1 ? n1 : (D == 2 ? n2 : n3));
        }
        if constexpr (F >= D) {
            // pieces of fragment F: issued at the top of fragment F - D; behind them D fragments' stores and the loads of the fragments
            // F + 1 .. F + D that exist (the prologue's pieces, F < D, landed in the constants' wait)
            constexpr int younger_loads = (F + D < MF ? D : (MF - 1 - F > 0 ? MF - 1 - F : 0));
            if (has1) e4_wait_cnt<D * E4Cnt<NF>::STORES + younger_loads * E4Cnt<NF>::LOADS>(cur);
        }
        bool regroup = false;
        if constexpr (F > 0) {
            // the fragment's first row inside the constants' row groups (fragments never straddle: the groups are multiples of 16 rows)
            const unsigned d = (unsigned)(m0f - rowfn(F - 1));
            t.add_rem += d;
            t.coef_rem += d;
            regroup = (p.add && t.add_rem >= (unsigned)p.add_rpg) || (p.coef && t.coef_rem >= (unsigned)p.coef_rpg);
        }
        if (regroup) {
            // (rare: the wave tile straddles two row groups - the constants' vmcnt(0) lands the look-ahead pieces too and must name them)
            e4_tile_consts<NF>(p, m0f, nw0, lane, t);
            asm volatile("" : "+v"(n1.a0), "+v"(n1.a1), "+v"(n1.a2), "+v"(n2.a0), "+v"(n2.a1), "+v"(n2.a2), "+v"(n3.a0), "+v"(n3.a1), "+v"(n3.a2));
        }
        {
            f32x4 accf[NF];
            acc.template get<F>(accf);
            e4_fragment<NF, GN, RD>(p, accf, m0f, nw0, lane, stage, cur, t, gn);
        }
        if constexpr (GN) {
            unsigned slot = 0, sid = 0;
            if (flushfn(F, m0f, slot, sid)) gn_flush<NF>(p, gn, (long long)sid, nw0, lane, stage, slot);
        }
        e4_retire<F + 1, MF, NF, GN, D, RD>(p, acc, nw0, lane, stage, n1, n2, n3, t, gn, rowfn, flushfn);
    }
}
template <int MF, int NF, bool GN, int D = E4_DEPTH, int RD = E4_ASM_READS, typename Acc, typename RowFn, typename FlushFn>
__device__ __forceinline__ void e4_retire_tile_src(const GP& p, const Acc& acc, long long nw0, int lane, unsigned char* stage, RowFn rowfn, FlushFn flushfn) {
    E4Res r0, r1, r2;
    r0.a0 = r0.a1 = r0.a2 = u32x4{0u, 0u, 0u, 0u};
    r1 = r0;
    r2 = r0;
    if (p.res1) {
        e4_load_res<NF>(p, rowfn(0), nw0, lane, r0);
        if constexpr (D >= 2 && MF > 1) e4_load_res<NF>(p, rowfn(1), nw0, lane, r1);
        if constexpr (D >= 3 && MF > 2) e4_load_res<NF>(p, rowfn(2), nw0, lane, r2);
    }
    E4Tile<NF> t;
    e4_tile_consts<NF>(p, rowfn(0), nw0, lane, t);          // (its vmcnt(0) also lands the residual pieces issued above)
    asm volatile("" : "+v"(r0.a0), "+v"(r0.a1), "+v"(r0.a2), "+v"(r1.a0), "+v"(r1.a1), "+v"(r1.a2), "+v"(r2.a0), "+v"(r2.a1), "+v"(r2.a2));
    GnAcc<GN ? NF : 1> gn;
    if constexpr (GN) gn_zero(gn);
    e4_retire<0, MF, NF, GN, D, RD>(p, acc, nw0, lane, stage, r0, r1, r2, t, gn, rowfn, flushfn);
}
template <int MF, int NF, bool GN, int D = E4_DEPTH, int RD = E4_ASM_READS, typename RowFn, typename FlushFn>
__device__ __forceinline__ void e4_retire_tile(const GP& p, f32x4 (&acc)[MF][NF], long long nw0, int lane, unsigned char* stage, RowFn rowfn, FlushFn flushfn) {
    e4_retire_tile_src<MF, NF, GN, D, RD>(p, E4AccArray<MF, NF>{acc}, nw0, lane, stage, rowfn, flushfn);
}

// ---- stream-K tail ------------------------------------------------------------------------------------------------------------------------
// A persistent launch of `ntiles` tiles on G blocks (one per CU) leaves CUs idle in its last round unless ntiles % G == 0: the 32 x 32 level
// of the U-Net runs 384 tiles (1.5 rounds: half the chip idles through the second), the 16 x 16 level 192 (a quarter idles throughout).  With a
// plan (v3d_sk_plan) the R = ntiles % G tiles of that last round are cut along K into R * U granules (U per tile) and block b takes the
// granules [b R U / G, (b + 1) R U / G): at most two pieces of two neighbouring tiles.  A piece that starts inside a tile is a DONOR piece:
// the block runs it FIRST, parks its fp32 accumulators in its workspace slot (16-byte write-through stores, MI355X_MICROARCH.md "publish-large")
// and raises one flag per wave.  A piece that starts a tile is the OWNER piece: the block runs it LAST, then adds the donors' partials in
// block order (a fixed order: the result does not depend on timing) and runs the normal epilogue.  Donors never wait, owners wait only for
// donors, every block is resident (G <= CUs, one block per CU): no cycle.  Hand-off per wave, not per block (wave w of the owner consumes what
// wave w of the donor stored - same lanes, same registers), so the two wave groups of a block keep their half-step offset through it.
struct SkItem {
    int tile;      // raw tile id (before xcd_remap); < 0: past the block's work (loaders keep their DMA counts with harmless re-reads)
    int u0, u1;    // granules [u0, u1) of the tile
    int role;      // 0 whole tile, 1 donor piece, 2 owner piece
    int d0, d1;    // owner: blocks d0 .. d1 hold the rest of the tile
};
// Order of a block's items: its donor piece FIRST (published long before the owner - which runs that tile's head as ITS last item - asks for
// it), then its whole tiles, then its owner piece.  (The donor piece also gets its own copy of the main loop in the kernels: one loop with
// all three ways of retiring an item - park / epilogue / gather + epilogue - pushed accumulators into scratch inside the loop.)
// The work list is computed once, by one thread, into a 96-byte LDS table (the divisions cost ~300 instructions and a dozen registers);
// U = granules per tile as the KERNEL counts them (the host planned with the same number).
// tab[0] = items of the block, tab[1] = 1 if item 0 is a donor piece, tab[4 + 8 k ..] = its k-th piece of the last round (k < 2).
constexpr int SK_TAB_BYTES = 96;
__host__ __device__ __forceinline__ void sk_build_table(const GP& p, int b, int G, int ntiles, int U, int* tab) {
    tab[1] = 0;
    if (p.sk_tail == 0) {
        tab[0] = (ntiles - b + G - 1) / G;
        return;
    }
    const long long RU = (long long)p.sk_tail * U;
    const long long lo = b * RU / G, hi = (b + 1) * RU / G;
    int n = 0;
    for (long long a = lo; a < hi; ++n) {
        const long long jt = a / U, t0 = jt * U, t1 = t0 + U;
        const long long e = hi < t1 ? hi : t1;
        int* q = tab + 4 + 8 * n;
        q[0] = p.sk_full * G + (int)jt;          // raw tile id
        q[1] = (int)(a - t0);                    // granules [u0, u1)
        q[2] = (int)(e - t0);
        q[3] = a > t0 ? 1 : (e < t1 ? 2 : 0);    // donor piece / owner piece / whole tile
        q[4] = b + 1;                            // owner: blocks d0 .. d1 hold the rest of the tile (d1 = last block whose range starts inside it)
        q[5] = (int)((t1 * G + RU - 1) / RU) - 1;
        if (n == 0 && q[3] == 1) tab[1] = 1;
        a = e;
    }
    tab[0] = p.sk_full + n;
}
__device__ __forceinline__ SkItem sk_item(const GP& p, int b, int G, int i, int nitems, int ndonor, int U, const int* tab) {
    SkItem it;
    it.u0 = 0; it.u1 = U; it.role = 0; it.d0 = it.d1 = 0;
    it.tile = i < nitems ? b + (i - ndonor) * G : -1;
    if (p.sk_tail != 0 && i < nitems && (i < ndonor || i >= ndonor + p.sk_full)) {
        const int* q = tab + 4 + 8 * (i < ndonor ? 0 : ndonor);
        it.tile = __builtin_amdgcn_readfirstlane(q[0]);
        it.u0 = __builtin_amdgcn_readfirstlane(q[1]);
        it.u1 = __builtin_amdgcn_readfirstlane(q[2]);
        it.role = __builtin_amdgcn_readfirstlane(q[3]);
        it.d0 = __builtin_amdgcn_readfirstlane(q[4]);
        it.d1 = __builtin_amdgcn_readfirstlane(q[5]);
    }
    return it;
}
// donor: park the wave's accumulators (slot = the block), publish.  Every VMEM op of the wave has drained before the flag leaves (the LDS-DMA
// prefetches in flight included: a bubble of one memory round trip, once per launch and block).
template <int MF, int NF>
__device__ __forceinline__ void sk_publish(const GP& p, const f32x4 (&acc)[MF][NF], int blk, int wave, int lane) {
    // (opaque lane id: as loop invariants the 30 store addresses were hoisted in front of the main loop and spilled there)
    asm volatile("" : "+v"(lane));
    const float* base = p.sk_ws + ((size_t)blk * 8 + wave) * (MF * NF) * 256;            // wave-uniform: SGPR pair
    unsigned voff = (unsigned)lane * 16u;
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            constexpr int dummy = 0;
            (void)dummy;
            const int k = i * NF + j;
            if (k % 4 == 0 && k > 0) voff += 4096u;
            // (s_nop 1: a VALU write to the data registers of a 16-byte store needs two wait states behind it; the compiler inserts them for its
            // own stores, not behind inline asm - its next instruction re-used the first data register for the next offset and the lanes the
            // memory pipeline reads last, 12..15 of every 16, stored that instead)
            switch (k % 4) {       // (immediate offsets reach 4095 bytes: four 1-KiB fragments per base step)
                case 0: asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(voff), "v"(acc[i][j]), "s"(base) : "memory"); break;
                case 1: asm volatile("global_store_dwordx4 %0, %1, %2 offset:1024 sc0 sc1\n\ts_nop 1" ::"v"(voff), "v"(acc[i][j]), "s"(base) : "memory"); break;
                case 2: asm volatile("global_store_dwordx4 %0, %1, %2 offset:2048 sc0 sc1\n\ts_nop 1" ::"v"(voff), "v"(acc[i][j]), "s"(base) : "memory"); break;
                default: asm volatile("global_store_dwordx4 %0, %1, %2 offset:3072 sc0 sc1\n\ts_nop 1" ::"v"(voff), "v"(acc[i][j]), "s"(base) : "memory"); break;
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(p.sk_flags + blk * 8 + wave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// owner: acc += the partials of blocks d0 .. d1, in that order.  Bounded spin (a donor that never shows up would otherwise hang the device).
template <int MF, int NF>
__device__ __forceinline__ void sk_gather(const GP& p, f32x4 (&acc)[MF][NF], int d0, int d1, int wave, int lane, int G) {
    asm volatile("" : "+v"(lane));
    for (int d = d0; d <= d1; ++d) {
        unsigned* flag = p.sk_flags + d * 8 + wave;
        bool ok = false;
        for (int spin = 0; spin < (1 << 21); ++spin) {
            // (every lane polls the same word; readfirstlane keeps the loop exit - and with it everything behind the loop - wave-uniform)
            if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) { ok = true; break; }
            __builtin_amdgcn_s_sleep(4);
        }
        if (!ok) {
            if (lane == 0) atomicAdd(p.sk_flags + G * 8, 1u);      // (word behind the flags: hand-offs that gave up - never expected, tests read it)
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float* base = p.sk_ws + ((size_t)d * 8 + wave) * (MF * NF) * 256;          // wave-uniform: SGPR pair
        // two fragment rows (2 NF vectors) per round trip.  The loads are inline asm: a destination register holds nothing until the wait
        // below, and nothing may touch it before - so every destination is an operand of the wait statement itself (a copy or a spill the
        // register allocator placed between a load and a later, anonymous wait would move the register's OLD content: seen as launch-to-
        // launch differences in the first version, which kept a second row in flight across the adds).
        static_assert(MF % 2 == 0, "rows are gathered in pairs");
        auto load = [&](f32x4& dst, int k) __attribute__((always_inline)) {
            const unsigned voff = (unsigned)lane * 16u + (unsigned)(k / 4) * 4096u;
            switch (k % 4) {
                case 0: asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(dst) : "v"(voff), "s"(base) : "memory"); break;
                case 1: asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024 sc0 sc1" : "=v"(dst) : "v"(voff), "s"(base) : "memory"); break;
                case 2: asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048 sc0 sc1" : "=v"(dst) : "v"(voff), "s"(base) : "memory"); break;
                default: asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072 sc0 sc1" : "=v"(dst) : "v"(voff), "s"(base) : "memory"); break;
            }
        };
#pragma unroll
        for (int i = 0; i < MF; i += 2) {
            f32x4 r[2 * NF];
#pragma unroll
            for (int j = 0; j < 2 * NF; ++j) load(r[j], i * NF + j);
            if constexpr (NF == 5)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9])::"memory");
            else if constexpr (NF == 4)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])::"memory");
            else
                static_assert(NF == 4 || NF == 5, "wait statement for this fragment count");
#pragma unroll
            for (int j = 0; j < 2 * NF; ++j) acc[i + j / NF][j % NF] += r[j];
        }
        if (lane == 0) __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    }
}

// host: may a launch use the hand-managed epilogue? (wm x wn = wave tile)
inline bool e4_ok(const V3dGemmParams& p, int wm, int wn) {
    auto al = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) % a) == 0; };
    if (p.out_fp32 || p.M % wm || p.N % wn || p.ldo % 8 || !al(p.out, 16)) return false;
    // the epilogue tracks rows and row-group remainders in 32 bits (e4_udiv on (unsigned)m0f, E4GnRun): larger launches take the generic epilogue
    if (p.M >= (1ll << 32) || (p.add && p.add_rpg >= (1ll << 32)) || (p.coef && p.coef_rpg >= (1ll << 32))) return false;
    if (p.bias && !al(p.bias, 16)) return false;
    if (p.add && (!al(p.add, 16) || p.add_ld % 4 || p.add_rpg % 16)) return false;
    if (p.res1 && (!al(p.res1, 16) || p.ldr1 % 8)) return false;
    if (p.res2 && (!al(p.res2, 8) || p.ldr2 % 4)) return false;
    if (p.coef && p.coef_rpg % 16) return false;
    return true;
}

}  // namespace
