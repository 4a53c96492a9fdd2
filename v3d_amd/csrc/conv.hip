// LDS-haloed convolution with GroupNorm (+SiLU) applied in the operand path (gfx950) - the "GN -> SiLU -> conv" of every ResBlock half.
//
// Reference: ResBlock in_layers / out_layers = GroupNorm32 -> SiLU -> conv (sgm/modules/diffusionmodules/openaimodel.py:267-271,302-314),
// the same with the (3,1,1) Conv3d in the time_stack of VideoResBlock (video_model.py:42-55, temporal_ae.py:32-44).  Rounds 1-2 ran it as
// v3d_groupnorm_apply (read + write of the activation tensor) followed by an implicit-GEMM convolution whose every tap re-fetched its
// activation rows L2 -> LDS (9 x per element for the 3x3, 5.8x the algorithmic bytes measured).  Here:
//   * a tile = 192 consecutive output pixels (whole image rows) x 320 output channels, persistent blocks of 8 waves, accumulators and
//     epilogue exactly as gemm_kernel_v3<192, 320> (gemm.hip) - bias / emb vector / residual / alpha blend / GroupNorm statistics of the
//     output all stay fused;
//   * per 32-channel chunk the tile's HALO - (rows + 2) x (W + 2) pixels for the 3x3, (frames + 2) x 32 positions for the temporal conv -
//     is fetched ONCE, raw, by LDS-DMA into a double-buffered LDS image (64 B per pixel, lines padded to a multiple of 8 pixels so a tap's
//     line offset never changes the bank swizzle);
//   * the wave that fetched a 1-KiB piece normalises it IN PLACE: ds_read 16 B -> x * scale + shift (the per-(image, channel) affine of
//     v3d_groupnorm_finalize, staged per chunk in a 512-byte wave-private LDS slot) -> SiLU -> bf16 -> ds_write, one instruction of that
//     chain per MFMA issue slot of the tap steps (fenced, the ff.hip way), so the VALU work hides behind the matrix pipe;
//   * the nine (three) taps read SHIFTED fragments of the same image - fragment address = per-lane base for the tap's dx + an immediate
//     for its dy - and only the weights stream per (chunk, tap) step through the 4-deep LDS-DMA ring of the v3 kernel;
//   * pixels outside the image are out-of-range DMA sources (zeros) normalised with a zero affine (-> 0, what the padding is); (row, dy)
//     pairs that would cross an image boundary inside the flat pixel order read their fragment from a zero page instead.
// L2 -> LDS fill per chunk and CU drops from 9 x 12 KiB (activations) + 180 KiB (weights) to 23 + 180 KiB, the GroupNorm apply pass and
// its 2 x 94 MB of HBM traffic per use disappear.
#include <stdlib.h>

#define E4_ASM_READS 0        // this file chooses the epilogue's LDS read form per kernel (CONV_TEMP_READS below); gn_flush keeps the compiler-visible reads
#include "gemm_common.h"

#ifndef CONV_3X3_READS
#define CONV_3X3_READS 0       // ... of the 3 x 3 haloed kernels: measured -1 ... +6 % per launch with the lean asm form (2), see the call
#endif
#ifndef CONV_TEMP_READS
#define CONV_TEMP_READS 2      // e4_fragment read form of the temporal haloed kernels (A/B: -DCONV_TEMP_READS=0)
#endif

// Timing experiments (tools/conv_ablate.sh, results in profiles/r03_conv_ablation.txt): -DCONV_ABL=<bits> builds of this file only -
// compile-time, so the measured loop carries no extra branches.  1 no epilogue, 2 no MFMA, 4 no weight DMA, 4096 no fragment reads,
// 8192 no halo DMA, 16384 no normalisation chain, 32768 no per-step waits / barrier, 65536 halo source rows folded into an L2-resident window.
// The product build has CONV_ABL = 0.
#ifndef CONV_ABL
#define CONV_ABL 0
#endif
#define CABL(bit) ((CONV_ABL & (bit)) != 0)
// (round 4 A/B'ed the placement of this loop's pieces - weight DMA inside the MFMA sequence, halo behind the weights, odd waves' DMA in front of
// their reads, no s_setprio: all neutral or slower, profiles/r04_mainloop_ab.txt; the variants are in profiles/r04_mainloop_variants.patch)

namespace {

enum { HM_CONV = 0, HM_TEMP = 1 };

template <int HM, int W_>
struct HaloGeom {
    static constexpr int BM = 192, BN = 320;
    static constexpr int NT = HM == HM_CONV ? 9 : 3;                                  // taps = steps per 32-channel chunk
    static constexpr int LINE = HM == HM_CONV ? ((W_ + 2 + 7) / 8) * 8 : 32;          // halo pixels per line (image row / frame)
    // W = 8 (round 6): a tile is three WHOLE 8 x 8 images, so the lines above and below it belong to other images and are never read (the
    // taps that would are masked to the zero page): the halo holds the tile's 24 lines only (TOP = 0) and up to three statistics groups
    static constexpr int TOP = HM == HM_CONV && W_ == 8 ? 0 : 1;                      // halo lines above the tile's first line
    static constexpr int NSTAT = HM == HM_CONV && W_ == 8 ? 3 : 2;                    // statistics groups (images) a halo can touch
    static constexpr int NLINES = HM == HM_CONV ? BM / W_ + 2 * TOP : 8;              // 3x3: tile rows + 2; temporal: 6 frames + 2
    static constexpr int HROWS = NLINES * LINE;
    static constexpr int HPW = ((HROWS + 15) / 16 + 7) / 8;                           // 1-KiB halo pieces per wave
    static constexpr int HBYTES = HPW * 8 * 1024;
    static constexpr int NBUF = HM == HM_CONV ? 2 : 3;                                // halo images in flight
    static constexpr int LA = NBUF - 1;                                               // chunks of look-ahead of the halo loader
    static constexpr int XF0 = 3;                                                     // first step (after the issue step) whose MFMA slots carry the transform
    static constexpr int XFN = LA * NT - 4;                                           // ... and how many steps do
    static constexpr int NTAB = XF0 >= NT ? 2 : 1;                                    // (scale, shift) slots per wave: 2 when a chunk's chain runs beside the next issue
    static_assert(XFN >= 1, "halo pipeline too shallow");
};

// halo pixel row -> XOR on the 16-byte chunk position.  Fragments start at ARBITRARY pixel rows (tap shifts), so the swizzle must keep
// ds_read_b128 conflict-free for every start row: its lane groups read rows {r..r+3, r+12..r+15} at chunk c and {r+4..r+11} at c ^ 1; per
// row residue mod 4 that is four consecutive 4-row blocks q0..q0+3 at chunk bits (0,1,1,0), and f(q) = 2 (q & 1) makes the four 16-byte
// slots {f(q0), 1^f(q0+1), 1^f(q0+2), f(q0+3)} distinct for either parity of q0 (the gemm.hip swizzle {0,2,3,1} only does for q0 = 0).
__device__ __forceinline__ int hswz(int hp) { return (hp >> 1) & 2; }
__device__ __forceinline__ int wswz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // weight rows: as gemm.hip swz_row<1>

// one instruction of the in-place normalisation chain per call; 15 calls handle one channel pair (see the kernel comment)
struct XfState {
    u32x4 raw;             // the 16-byte vector being normalised (re-loaded with the next one as soon as its last pair is unpacked)
    f32x4 tab;             // (scale, shift) x 2 of the current channel pair (re-loaded for the next pair once both fmas have issued)
    u32x4 out;
    float x0, x1, y0, y1, t0, t1;
};

// retire fragment F (16 rows x the wave's 80 channels) of a finished tile, residual rows of fragment F + 1 in flight meanwhile.  The pieces
// travel BY VALUE through the recursion (gemm.hip v3_retire_chunks: captured by reference in a lambda they went through a stack array -
// every residual load was followed by s_waitcnt vmcnt(0) and a scratch store: 18 serialised memory round trips per tile, the [res]
// convolutions ran 50 us slower than their v3 twins).  frow(f) = first output row of fragment f; slot = the wave tile's statistics slot.
struct RetireGeo {
    long long mw0;      // first row of the wave tile (3x3) / of its first frame (temporal)
    long long S;        // temporal: rows per frame
    unsigned tslot;     // temporal: the wave tile's statistics slot
};
template <int HM>
__device__ __forceinline__ long long frag_row0(const RetireGeo& q, int f) {
    return HM == 0 ? q.mw0 + f * 16 : q.mw0 + (long long)(f / 2) * q.S + (f % 2) * 16;
}
template <int F, int MF, int NF, int HM, bool GN>
__device__ __forceinline__ void halo_retire(const GP& p, f32x4 (&acc)[MF][NF], const RetireGeo& q, long long nw0, int lane, unsigned char* estage,
                                            u32x4 c0, u32x4 c1, u32x4 c2, bool res_pre, const float4 (&bv)[NF], GnAcc<GN ? NF : 1>& gn) {
    if constexpr (F < MF) {
        constexpr int WMR = MF * 16;
        const long long m0f = frag_row0<HM>(q, F);
        u32x4 n0 = c0, n1 = c1, n2 = c2;
        if constexpr (F + 1 < MF) {
            if (res_pre) {
                const long long m1 = frag_row0<HM>(q, F + 1);
                n0 = load_res_piece<0, 1, NF, false>(p, m1, nw0, lane);
                n1 = load_res_piece<1, 1, NF, false>(p, m1, nw0, lane);
                n2 = load_res_piece<2, 1, NF, false>(p, m1, nw0, lane);
            }
        }
        epilogue<1, NF, false, true, true, 16, GN>(p, *reinterpret_cast<f32x4(*)[1][NF]>(&acc[F]), m0f, nw0, 0, lane, estage, c0, c1, c2, res_pre, bv, true, &gn);
        if constexpr (GN) {
            // writer = this wave tile's run of rows inside one statistics group; its slot is unique inside the group (gemm_common.h gn_flush)
            const long long sid = m0f / p.gn_rps;
            bool flush = F + 1 == MF;
            unsigned slot = q.tslot;
            if (HM == 0) {
                flush = flush || (m0f + 16) / p.gn_rps != sid;
                const long long first = q.mw0 / p.gn_rps == sid ? q.mw0 - sid * p.gn_rps : 0;
                slot = (unsigned)((first + WMR - 1) / WMR);
            }
            if (flush) gn_flush<NF>(p, gn, sid, nw0, lane, estage, slot);
        }
        halo_retire<F + 1, MF, NF, HM, GN>(p, acc, q, nw0, lane, estage, n0, n1, n2, res_pre, bv, gn);
    }
}

template <int HM, int W_, bool XF, bool GN>
__global__ __launch_bounds__(512, 2) void conv_halo_kernel(GP p, int ntiles) {
    using G = HaloGeom<HM, W_>;
    constexpr int BM = G::BM, BN = G::BN, NT = G::NT, LINE = G::LINE, HPW = G::HPW, HBYTES = G::HBYTES, NBUF = G::NBUF;
    constexpr int NS = 4, NW = 8, ROWB = 64;
    constexpr int WM = 96, WN = 80, MF = 6, NF = 5;
    constexpr int WSTAGE = BN * ROWB;                     // 20 KiB of weights per (chunk, tap) step
    constexpr int EPI_REGION = 16 * (NF * 32 + 16);
    constexpr int TABB = G::NSTAT * 256;                 // (scale, shift) of a chunk's 32 channels for each statistics group of the halo
    constexpr int RING_OFF = 0, HALO_OFF = NS * WSTAGE, ZERO_OFF = HALO_OFF + NBUF * HBYTES, TAB_OFF = ZERO_OFF + 1024,
                  EPI_OFF = TAB_OFF + NW * TABB * G::NTAB, SK_OFF = EPI_OFF + NW * EPI_REGION;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[SK_OFF + SK_TAB_BYTES];        // the ONLY __shared__ object
    static_assert(sizeof(lds) <= 160 * 1024, "LDS budget");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave >> 2, wn = wave & 3;             // waves 2 (M) x 4 (N): wave tile 96 x 80
    unsigned char* estage = lds + EPI_OFF + wave * EPI_REGION;

    // work list of the block: whole tiles, then (with a stream-K plan, gemm_common.h sk_*) a donor piece and / or an owner piece of the last
    // round's tiles, cut at 32-channel chunks.  The three cursors below (weights, halo, consumer) walk the same (item, chunk) sequence.
    const int nchunks = (int)(p.K / 32);
    const int Gd = gridDim.x;
    const int* sktab = reinterpret_cast<const int*>(lds + SK_OFF);
    if (tid == 0) sk_build_table(p, (int)blockIdx.x, Gd, ntiles, nchunks, reinterpret_cast<int*>(lds + SK_OFF));
    __syncthreads();
    const int my_items = __builtin_amdgcn_readfirstlane(sktab[0]);
    const int my_donor = __builtin_amdgcn_readfirstlane(sktab[1]);      // 1: item 0 is a donor piece
    auto item_at = [&](int i) __attribute__((always_inline)) -> SkItem { return sk_item(p, (int)blockIdx.x, Gd, i, my_items, my_donor, nchunks, sktab); };
    auto tile_origin = [&](int raw, int& tm, long long& n0) __attribute__((always_inline)) {
        const int id = xcd_remap(raw, ntiles);
        int tn;
        tile_coords(p, id, tm, tn);
        n0 = (long long)tn * BN;
    };
    // first source pixel row of tile row-block tm (3x3: flat pixel order; temporal: (sample, frame block, position block))
    const int k1chunks = (int)(p.K1 / 32);
    const int H = p.Hout;
    const int sblocks = HM == HM_TEMP ? (int)(p.S / 32) : 1;      // temporal: position blocks per frame
    const int tblocks = HM == HM_TEMP ? p.T / 6 : 1;

    // ---------------------------------------------------------------- weight loader (3 steps ahead of the consumer)
    const bufrsrc_t rsW = make_rsrc(p.W, p.w_bytes);
    const int prow = lane >> 2;
    const unsigned kchunk_w = (unsigned)(((lane & 3) ^ wswz(prow)) * 16);
    const unsigned voffW = (unsigned)(prow * (int)p.ldw * 2) + kchunk_w;     // per-lane part of a weight piece's source offset (row inside the piece, k-chunk)
    int w_it = 0, w_c = 0, w_c1 = nchunks;                // weight cursor: item, chunk, end chunk of the item (its tap is static in the unrolled step bodies)
    int w_n0 = 0;                                         // first output channel of the cursor's tile
    const unsigned tap_bytes = (unsigned)(p.N * p.ldw * 2);
    const int row16_bytes = (int)(16 * p.ldw * 2);
    auto set_wtile = [&](int it) __attribute__((always_inline)) {
        int tm;
        long long n0 = 0;
        const SkItem q = item_at(it);
        if (q.tile >= 0) tile_origin(q.tile, tm, n0);    // (past the last item: harmless re-reads of tile column 0 keep the DMA count constant)
        w_n0 = (int)n0;
        w_c = q.u0;
        w_c1 = q.u1;
    };
    auto issue_w_piece = [&](int stage, int ltap, int i) __attribute__((always_inline)) {
        if (CABL(4)) return;
        // scalar offset: tap, chunk and the piece's 16-row block (N % 320 == 0: every row of the tile exists)
        const int so = (int)(ltap * tap_bytes) + w_c * 64 + (w_n0 / 16 + wave + NW * i) * row16_bytes;
        if (i < 2 || grp == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(lds + RING_OFF + stage * WSTAGE + (wave + NW * i) * 1024), 16,
                                                     (int)voffW, so, 0, 0);
    };
    auto issue_w = [&](int stage, int ltap) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) issue_w_piece(stage, ltap, i);
    };
    // ---------------------------------------------------------------- halo loader (LA chunks ahead)
    const bufrsrc_t rsA1 = make_rsrc(p.A, p.a_bytes);
    const bufrsrc_t rsA2 = make_rsrc(p.A2 ? p.A2 : p.A, p.A2 ? p.a2_bytes : p.a_bytes);
    const bufrsrc_t rsT = make_rsrc(p.gn_in ? (const void*)p.gn_in : (const void*)p.A, p.gn_in ? p.gn_in_bytes : 0u);
    int h_it = 0, h_c = 0, h_c1 = 0, h_buf = 0;          // halo cursor: item, chunk, end chunk of the item, LDS image it writes
    // per-tile state of the cursor, all wave-uniform (the per-lane source rows are recomputed where they are used: a dozen VALU per piece and
    // chunk against registers held through every step)
    struct HTile {
        int live;          // the cursor is inside this block's tiles
        int fr0;           // 3x3: first flat image row of the tile;  temporal: sample * T + first frame of the tile's frame block
        int frB;           // 3x3: first flat image row of the halo's SECOND statistics group (image);  temporal: position block * 32
        int frC;           // 3x3, W = 8: first flat image row of the halo's THIRD statistics group
        unsigned statA;    // (scale, shift) rows of the halo's first / last source row
        unsigned statB;
    };
    HTile ht = {0, 0, 0, 0, 0u, 0u}, xt = {0, 0, 0, 0, 0u, 0u};      // cursor's tile / the tile of the image issued last (picked up by its normalisation chain)
    int h_tab = 0, x_tab = 0;                             // table slot the next issue fills / the last issue filled
    unsigned latch_base = 0;
    // piece i of this wave = halo pixel rows (wave + 8 i) * 16 .. + 15; the lane holds 16 bytes of pixel row + (lane >> 2).  Pieces start at
    // multiples of 16 rows, so the swizzle (bit 2 of the pixel row) - and with it the logical 16-byte chunk (8 channels) the lane holds - is
    // the same for every piece
    const int hchunkpos = (lane & 3) ^ hswz(lane >> 2);
    const unsigned hvec0 = (unsigned)(wave * 1024 + lane * 16);      // LDS byte offset of the lane's vector of piece 0 inside a halo image
    auto set_htile = [&](int it) __attribute__((always_inline)) {
        int tm = 0;
        long long n0;
        const SkItem q = item_at(it);
        ht.live = q.tile >= 0;
        if (ht.live) tile_origin(q.tile, tm, n0);
        h_c = q.u0;
        h_c1 = q.u1;
        const long long rps = p.gn_in_rps > 0 ? p.gn_in_rps : 1;
        if (HM == HM_CONV) {
            const int fr0 = tm * (BM / W_);                                    // first flat image row of the tile
            const int nfr = (int)(p.M / W_);
            const int f_lo = fr0 - G::TOP < 0 ? 0 : fr0 - G::TOP, f_hi = fr0 + BM / W_ - 1 + G::TOP >= nfr ? nfr - 1 : fr0 + BM / W_ - 1 + G::TOP;
            ht.fr0 = fr0;
            ht.statA = (unsigned)((long long)f_lo * W_ / rps);
            ht.statB = (unsigned)((long long)f_hi * W_ / rps);
            ht.frB = (int)(((long long)ht.statA + 1) * rps / W_);
            ht.frC = (int)(((long long)ht.statA + 2) * rps / W_);
        } else {
            // tile row-block tm = (sample b, frame block tb, position block sb), positions fastest
            const int sb = tm % sblocks, tb = (tm / sblocks) % tblocks, b = tm / (sblocks * tblocks);
            ht.fr0 = b * p.T + tb * 6;
            ht.frB = sb * 32;
            ht.frC = 0;
            ht.statA = ht.statB = (unsigned)(((long long)b * p.T * p.S) / rps);          // one statistics group per sample (the 3-D GroupNorm)
        }
    };
    // source pixel row of this lane's halo pixel of piece i in tile t (kInvalid: padding / outside the tensor); second = it belongs to statB
    // (ln = an OPAQUE copy of the lane id, made where the rows are needed: as loop invariants the pieces' (line, x) pairs were hoisted in front of
    // the main loop, a dozen registers held through every step - or, once the loop was a register short, re-read from scratch inside it)
    auto halo_row = [&](const HTile& t, int i, int& second, int ln) __attribute__((always_inline)) -> unsigned {
        const int hp = (wave + NW * i) * 16 + (ln >> 2);
        second = 0;
        if (HM == HM_CONV) {
            const int line = hp / LINE, x = hp - line * LINE - 1;
            const int fr = t.fr0 - G::TOP + line;
            const int nfr = (int)(p.M / W_);
            const bool ok = t.live && x >= 0 && x < W_ && fr >= 0 && fr < nfr && line < G::NLINES;
            second = (fr >= t.frB ? 1 : 0) + (G::NSTAT > 2 && fr >= t.frC ? 1 : 0);       // statistics group of the line, relative to statA
            return ok ? (unsigned)(fr * W_ + x) : kInvalid;
        } else {
            const int line = hp >> 5, sp = hp & 31;
            const int b = t.fr0 / p.T, f = t.fr0 - b * p.T - 1 + line;          // frame of the sample
            const bool ok = t.live && f >= p.tmin && f <= p.tmax;
            long long row = ((long long)t.fr0 - 1 + line) * p.S + t.frB + sp;
            if (p.halo_rows > 0) {                                             // frame sharding: frame -1 / T live in the slabs around the local frames
                if (f < 0) row = (long long)b * p.S + t.frB + sp - p.halo_rows;
                else if (f >= p.T) row = p.M + (long long)b * p.S + t.frB + sp;
            }
            return ok ? (unsigned)(row + p.a_row0) : kInvalid;
        }
    };
    auto issue_halo = [&]() __attribute__((always_inline)) {
        if (CABL(8192)) return;
        if (h_c == h_c1) set_htile(h_it);                 // (cursor at the end of its item - or at the very start: 0 == 0 - : enter item h_it)
        const bool second = h_c >= k1chunks;
        const int cc = second ? h_c - k1chunks : h_c;
        const unsigned ld2 = (unsigned)((second ? p.lda2 : p.lda) * 2);
        if (XF && lane < 16 * G::NSTAT) {
            // (scale, shift) of this chunk's 32 channels for the (at most NSTAT) statistics groups of the halo: NSTAT x 256 B
            const unsigned sg = ht.statA + (unsigned)(lane >> 4);
            const unsigned st = G::NSTAT == 2 ? (lane < 16 ? ht.statA : ht.statB) : (sg < ht.statB ? sg : ht.statB);
            const unsigned vo = (unsigned)(((long long)st * p.K + (long long)h_c * 32) * 8) + (unsigned)(lane & 15) * 16u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsT, (__attribute__((address_space(3))) void*)(lds + TAB_OFF + (wave * G::NTAB + h_tab) * TABB), 16, (int)(ht.live ? vo : kInvalid), 0, 0, 0);
        }
        int ln = lane;
        asm volatile("" : "+v"(ln));
        latch_base = (unsigned)(HALO_OFF + h_buf * HBYTES);
        xt = ht;                                          // (the chain of this image starts after the issue: xf_begin picks these up)
        x_tab = h_tab;
        if (G::NTAB == 2) h_tab ^= 1;
#pragma unroll
        for (int i = 0; i < HPW; ++i) {
            int second_stat;
            const unsigned row = halo_row(ht, i, second_stat, ln);
            const unsigned vo = row == kInvalid ? kInvalid : (CABL(65536) ? (row & 1023u) : row) * ld2 + (unsigned)hchunkpos * 16u;     // (65536: L2-hot source rows)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? rsA2 : rsA1, (__attribute__((address_space(3))) void*)(lds + HALO_OFF + h_buf * HBYTES + (wave + NW * i) * 1024), 16,
                                                     (int)vo, cc * 64, 0, 0);
        }
        if (++h_c == h_c1) ++h_it;
        h_buf = h_buf + 1 == NBUF ? 0 : h_buf + 1;
    };
    // ---------------------------------------------------------------- the in-place normalisation of the image the loader filled last
    // image being normalised: the one issue_halo wrote at the last issue step
    unsigned xf_base = 0;                                 // LDS byte address of that image
    HTile ct = {0, 0, 0, 0, 0u, 0u};                         // its tile
    int c_tab = 0;                                        // its table slot
    XfState xs;
    auto xf_read_vec = [&](int v) __attribute__((always_inline)) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(lds + xf_base + hvec0 + v * 8192);
    };
    // (scale, shift) x 2 of channel pair e of vector v: the wave's table slot (+256: the halo's second statistics group), or the zero page
    // for padding pixels (x * 0 + 0 -> SiLU(0) = 0: what the convolution's zero padding is)
    unsigned ctab[HPW];                                   // (computed once per image in xf_begin: live through the chain's steps only)
    auto xf_read_tab = [&](int v, int e) __attribute__((always_inline)) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(lds + ctab[v] + e * 16);
    };
    // OP = 15 * (4 * vector + pair) + stage.  Every result is pinned to its slot by an empty volatile asm: the steps are several basic blocks (the
    // two wave groups pass the barrier at different points) and without the pins MachineSink moves the whole chain down to the block of its
    // only consumer, the ds_write - en bloc in front of the next step's MFMAs.
    auto xf_op = [&](auto op_) __attribute__((always_inline)) {
        constexpr int OP = decltype(op_)::value;
        constexpr int NOPS = HPW * 4 * 15;
        if constexpr (XF && OP >= 0 && OP < NOPS) {
            constexpr int pr = OP / 15, st = OP % 15, v = pr / 4, e = pr % 4;
            if constexpr (st == 0) {
                xs.x0 = bflo(xs.raw[e]); asm volatile("" : "+v"(xs.x0));
            } else if constexpr (st == 1) {
                xs.x1 = bfhi(xs.raw[e]); asm volatile("" : "+v"(xs.x1));
                if constexpr (e == 3 && v + 1 < HPW) xs.raw = xf_read_vec(v + 1);     // 14 slots ahead of its first use
            } else if constexpr (st == 2) {
                xs.y0 = __builtin_fmaf(xs.x0, xs.tab[0], xs.tab[1]); asm volatile("" : "+v"(xs.y0));
            } else if constexpr (st == 3) {
                xs.y1 = __builtin_fmaf(xs.x1, xs.tab[2], xs.tab[3]); asm volatile("" : "+v"(xs.y1));
                if constexpr (pr + 1 < HPW * 4) xs.tab = xf_read_tab((pr + 1) / 4, (pr + 1) % 4);   // 14 slots ahead of its first use
            } else if constexpr (st == 4) {
                xs.t0 = xs.y0 * -1.4426950408889634f; asm volatile("" : "+v"(xs.t0));
            } else if constexpr (st == 5) {
                xs.t1 = xs.y1 * -1.4426950408889634f; asm volatile("" : "+v"(xs.t1));
            } else if constexpr (st == 6) {
                xs.t0 = __builtin_amdgcn_exp2f(xs.t0); asm volatile("" : "+v"(xs.t0));
            } else if constexpr (st == 7) {
                xs.t1 = __builtin_amdgcn_exp2f(xs.t1); asm volatile("" : "+v"(xs.t1));
            } else if constexpr (st == 8) {
                xs.t0 = 1.0f + xs.t0; asm volatile("" : "+v"(xs.t0));
            } else if constexpr (st == 9) {
                xs.t1 = 1.0f + xs.t1; asm volatile("" : "+v"(xs.t1));
            } else if constexpr (st == 10) {
                xs.t0 = __builtin_amdgcn_rcpf(xs.t0); asm volatile("" : "+v"(xs.t0));
            } else if constexpr (st == 11) {
                xs.t1 = __builtin_amdgcn_rcpf(xs.t1); asm volatile("" : "+v"(xs.t1));
            } else if constexpr (st == 12) {
                xs.y0 = xs.y0 * xs.t0; asm volatile("" : "+v"(xs.y0));      // (SiLU always: the host refuses a bare norm)
            } else if constexpr (st == 13) {
                xs.y1 = xs.y1 * xs.t1; asm volatile("" : "+v"(xs.y1));
            } else {
                { unsigned pk = pack2bf(xs.y0, xs.y1); asm volatile("" : "+v"(pk)); xs.out[e] = pk; }
                if constexpr (e == 3) {
                    // (the piece's 8-KiB stride rides in the instruction's offset field: as separate per-lane sums the three addresses were hoisted out of the
                    // loop and - once the kernel was a register short - reloaded from scratch inside it, each reload behind a vmcnt(0))
                    const unsigned wa = lds_addr(lds) + xf_base + hvec0;
                    const u32x4 wv = xs.out;
                    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wa), "v"(wv), "n"(v * 8192) : "memory");
                }
            }
        }
    };
    auto xf_begin = [&]() __attribute__((always_inline)) {          // first reads of an image (its DMA pieces have landed: counted waits below)
        if (XF) {
            xf_base = latch_base;
            ct = xt;
            c_tab = x_tab;
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int v = 0; v < HPW; ++v) {
                int second;
                const unsigned row = halo_row(ct, v, second, ln);
                ctab[v] = (row == kInvalid ? (unsigned)ZERO_OFF : (unsigned)(TAB_OFF + (wave * G::NTAB + c_tab) * TABB) + (unsigned)second * 256u) + (unsigned)hchunkpos * 64u;
            }
            xs.raw = xf_read_vec(0);
            xs.tab = xf_read_tab(0, 0);
        }
    };

    // ---------------------------------------------------------------- consumer state
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[MF], wf[NF];
#if CONV_ABL
#pragma unroll
    for (int i = 0; i < MF; ++i) xf[i] = bf16x8{};      // (the "no fragment reads" timing experiment multiplies whatever is here)
#pragma unroll
    for (int j = 0; j < NF; ++j) wf[j] = bf16x8{};
#endif
    const int wfrag_off = (lane & 15) * ROWB + (((lane >> 4) ^ wswz(lane & 15)) * 16) + wn * WN * ROWB;
    // activation fragment i of the wave = tile rows wm * 96 + i * 16 .. + 15 = 16 consecutive halo pixels starting at pixel row hp0(i) (a
    // multiple of 8, wave-uniform) + tap column dx.  Per-lane byte offset inside a halo image = faddr[dx] (pixel (lane & 15) + dx, chunk
    // (lane >> 4) swizzled by bit 2 of that pixel row - hp0(i) does not touch bit 2) + foff(i) (wave-uniform: SGPR) + dy * LINE * 64 (immediate)
    constexpr int NDX = HM == HM_CONV ? 3 : 1;
    unsigned faddr[NDX];
#pragma unroll
    for (int dx = 0; dx < NDX; ++dx) {
        // (W = 8: a fragment's 16 rows are two image lines of 8 pixels - lanes 8 .. 15 of a 16-lane group sit one halo line further; bit 2 of the pixel row, which
        // the swizzle keys on, and the row's bank class mod 4 are what they would be for 16 consecutive pixels)
        const int hp = HM == HM_CONV && W_ == 8 ? (lane & 7) + dx + ((lane >> 3) & 1) * LINE : (lane & 15) + dx;
        faddr[dx] = (unsigned)(HALO_OFF + hp * 64 + (((lane >> 4) ^ hswz(hp)) * 16));
    }
    auto foff = [&](int i) __attribute__((always_inline)) -> unsigned {
        const int r = wm * WM + i * 16;
        return (unsigned)((HM == HM_CONV ? (r / W_ + G::TOP - 1) * LINE + (r % W_) : r) * 64);      // (TOP = 0: line - 1 + dy; the dy = 0 tap of tile line 0 is masked)
    };
    const bool upper_half = HM == HM_CONV && W_ == 8 && ((lane >> 3) & 1);      // W = 8: the lane's pixel is in the fragment's second image line
    const unsigned zfrag = (unsigned)(ZERO_OFF + (lane & 15) * 64 + (lane >> 4) * 16);
    int c_buf = 0;                                        // halo image of the chunk being consumed (its offset is folded into faddr)
    unsigned vmask = 0;                                   // 3x3: bit (2 i) = fragment i may use dy = 0, bit (2 i + 1) = dy = 2 (same image)

    // ---------------------------------------------------------------- prologue
    for (int i = tid; i < 256; i += 512) reinterpret_cast<unsigned*>(lds + ZERO_OFF)[i] = 0u;
    set_wtile(0);
    // halo image 0 .. LA-1 of the first tile(s), normalised synchronously
#pragma unroll 1
    for (int pre = 0; pre < G::LA; ++pre) {
        issue_halo();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // (zero page written; first pass only matters)
        // (with NT <= XF0 the chain of an image runs beside the NEXT issue: the last image of the prologue is left to the loop's first chain)
        if (XF && (G::XF0 < NT || pre + 1 < G::LA)) {
            xf_begin();
            static_for<0, HPW * 4 * 15>([&](auto o) { xf_op(o); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    {   // weight stages 0 .. NS-2: taps 0 .. NS-2 of chunk 0
        static_for<0, NS - 1>([&](auto s_) { issue_w(decltype(s_)::value, decltype(s_)::value); });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int rd = 0;                                           // ring slot of the current step
    // one item: its chunks, then its retirement.  DONOR = the item is the block's donor piece (a compile-time copy of the loop, see sk_item)
    auto run_item = [&](int it, auto donor_) __attribute__((always_inline)) {
        constexpr bool DONOR = decltype(donor_)::value;
        int tm;
        long long e_n0;
        const SkItem item = item_at(it);
        tile_origin(item.tile, tm, e_n0);
        if (HM == HM_CONV) {
            vmask = 0;
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int fr = (int)(((long long)tm * BM + wm * WM + i * 16) / W_);
                const int y = fr % H;
                // (W = 8: the fragment is the image lines y and y + 1, y even - only its first line can lack the line above, only its second the line below)
                vmask |= (y > 0 ? 1u : 0u) << (2 * i);
                vmask |= (y + (W_ == 8 ? 1 : 0) < H - 1 ? 1u : 0u) << (2 * i + 1);
            }
            vmask = (unsigned)__builtin_amdgcn_readfirstlane((int)vmask);
        }
        for (int c = item.u0; c < item.u1; ++c) {
            static_for<0, NT>([&](auto tap_) {
                constexpr int tap = decltype(tap_)::value;
                constexpr int dy = HM == HM_CONV ? tap / 3 : tap, dx = HM == HM_CONV ? tap % 3 : 0;
                constexpr int ltap = (tap + NS - 1) % NT;             // the weight loader's tap, NS-1 steps ahead
                // ---- fragment reads of this step
                if (!CABL(4096)) {
                    const unsigned char* sb = lds + RING_OFF + rd * WSTAGE + wfrag_off;
                    // (the per-fragment sums below are loop invariants the compiler would otherwise hoist out of the nine tap bodies and keep in
                    // 30 registers; an opaque copy of the base pins them to the step)
                    unsigned fb = faddr[dx];
                    asm volatile("" : "+v"(fb));
#pragma unroll
                    for (int i = 0; i < MF; ++i) {
                        // (A/B, profiles/r04_mainloop_ab.txt: switching the SCALAR part of the address instead - a zero page as large as a fragment's
                        // pixel range, one v_add per fragment - removes 2 v_readlane + 1 v_cndmask per fragment and step and measured 4-8 % SLOWER)
                        unsigned a = fb + foff(i);
                        if (HM == HM_CONV && dy != 1) {
                            bool ok = (vmask >> (2 * i + (dy == 2 ? 1 : 0))) & 1u;
                            if (W_ == 8) ok = ok || (upper_half != (dy == 2));      // (per lane: the other image line of the fragment has its neighbour inside the image)
                            a = ok ? a : zfrag - (unsigned)(dy * LINE * 64);
                        }
                        xf[i] = *reinterpret_cast<const bf16x8*>(lds + a + dy * LINE * 64);
                    }
#pragma unroll
                    for (int j = 0; j < NF; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + j * 16 * ROWB);
                }
                // ---- loaders: the halo image LA chunks ahead (first step of a chunk), then the weights 3 steps ahead
                if constexpr (tap == G::XF0 % NT) xf_begin();     // (ahead of this step's issue: with NT <= XF0 the chain belongs to the PREVIOUS issue)
                if constexpr (tap == 0) issue_halo();
                if constexpr (ltap == 0) {                            // the weight cursor enters the next chunk
                    if (++w_c == w_c1) set_wtile(++w_it);
                }
                const int wst = rd == 0 ? NS - 1 : rd - 1;
                issue_w(wst, ltap);
                rd = (rd + 1 == NS) ? 0 : rd + 1;
                // DMA ops younger than the pieces of the NEXT step's weight stage (issued two steps ago): two weight stages, plus the halo
                // pieces + table piece when one of the last two issue points was a chunk's first step (they are issued AHEAD of that step's weights)
                constexpr int HL = (tap == 0 || tap == 1) ? HPW + (XF ? 1 : 0) : 0;
                if (grp == 1 && !CABL(32768)) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 2 + HL) : "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_setprio(1);
                // ---- 30 MFMAs, one instruction of the normalisation chain per issue slot
                constexpr int XSTEP = ((tap - G::XF0) % NT + NT) % NT;    // (3x3: taps 3..7 carry the chain of the image issued at tap 0)
                constexpr int NOPS = HPW * 4 * 15, NSLOT = G::XFN * 30;
                static_for<0, MF * NF>([&](auto n_) {
                    constexpr int n = decltype(n_)::value, i = n / NF, j = n % NF;
                    if (!CABL(2)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
                    if constexpr (XF && XSTEP >= 0 && XSTEP < G::XFN && !CABL(16384)) {
                        constexpr int s0 = XSTEP * 30 + n;
                        constexpr int o0 = (s0 * NOPS) / NSLOT, o1 = ((s0 + 1) * NOPS) / NSLOT;
                        static_for<o0, o1>([&](auto o) { xf_op(o); });
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (grp == 0 && !CABL(32768)) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (its in-place ds_writes are inline asm)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 3 + HL) : "memory");
                    __builtin_amdgcn_s_barrier();
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            {   // next chunk reads the next halo image
                const unsigned d = c_buf + 1 == NBUF ? (unsigned)(-(NBUF - 1) * HBYTES) : (unsigned)HBYTES;
                c_buf = c_buf + 1 == NBUF ? 0 : c_buf + 1;
#pragma unroll
                for (int dx = 0; dx < NDX; ++dx) faddr[dx] += d;
            }
        }
        // ---------------- item finished for this group.  Donor piece: park the accumulators for the tile's owner.  Owner piece: add the donors'.
        // Then (whole tile / owner) retire it (v3 epilogue: 16-row chunks through the wave's staging region)
        if constexpr (DONOR) {
            sk_publish<MF, NF>(p, acc, (int)blockIdx.x, wave, lane);
        } else if (CABL(1)) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            if (sum == 123.456f) reinterpret_cast<float*>(p.out)[0] = sum;   // keeps the accumulators live
        } else {
            long long mw0;
            if (HM == HM_CONV) {
                mw0 = (long long)tm * BM + wm * WM;
            } else {
                const int sb = tm % sblocks, tb = (tm / sblocks) % tblocks, b = tm / (sblocks * tblocks);
                mw0 = ((long long)b * p.T + tb * 6 + wm * 3) * p.S + sb * 32;      // first row of the wave's 3 frames x 32 positions
            }
            const long long nw0 = e_n0 + wn * WN;
            // a fragment's 16 rows are consecutive in both modes (3x3: flat pixels; temporal: 16 of the 32 positions of one frame)
            RetireGeo q;
            q.mw0 = mw0;
            q.S = p.S;
            q.tslot = 0;
            if (HM == HM_TEMP) {
                // temporal tiles: the host only offers the statistics epilogue for the 3-D GroupNorm (gn_rps = T * S: the wave's 3 frames x 32
                // positions lie in one group); writer index inside the sample = (frame block, position block, wave row)
                const int sb = tm % sblocks, tb = (tm / sblocks) % tblocks;
                q.tslot = (unsigned)((tb * sblocks + sb) * 2 + wm);
            }
            // hand-managed epilogue (gemm_common.h e4_*): asm loads one fragment ahead, one wait per fragment
            auto rowfn = [&](int f) __attribute__((always_inline)) -> long long { return frag_row0<HM>(q, f); };
            // (3x3: consecutive 16-row fragments, E4GnRun; temporal: the wave's 3 frames x 32 positions lie in one statistics group - the sample)
            E4GnRun<WM, MF> run;
            unsigned tsid = 0;
            if constexpr (GN) {
                if (HM == HM_CONV) run.init(q.mw0, p.gn_rps);
                else tsid = e4_udiv((unsigned)q.mw0, (unsigned)p.gn_rps);
            }
            auto flushfn = [&](int f, long long, unsigned& slot, unsigned& sid) __attribute__((always_inline)) -> bool {
                if (HM == HM_CONV) return run.step(f, slot, sid);
                slot = q.tslot;
                sid = tsid;
                return f + 1 == MF;
            };
            // (an opaque copy of the lane id: everything the epilogue derives from it is computed here, per tile - as loop invariants those values
            // were hoisted in front of the main loop, spilled there, and re-read from scratch ~20 times per fragment)
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const SkItem done = item_at(it);          // (re-read: role / donors are not carried through the chunk loop)
            if (done.role == 2) sk_gather<MF, NF>(p, acc, done.d0, done.d1, wave, lane_e, Gd);
            // staging reads of the epilogue (gemm_common.h e4_fragment RD): the temporal kernels take the lean asm form (one piece per LDS round trip: -6 %).  The 3 x 3
            // kernels keep the compiler-visible reads and their vmcnt(0) drains: once the in-place ds_write of the normalisation chain had freed their registers the
            // lean form fits all but the <W = 32, GroupNorm operand, no statistics> variant (tools/check_loop_scratch.py), but with 90-360 steps per tile the
            // drains are not what their epilogue costs - -DCONV_3X3_READS=2 measured +-1 % on five launch classes and +6 % on one (profiles/r06_conv_w8_ab.txt)
            e4_retire_tile<MF, NF, GN, E4_DEPTH, (HM == HM_TEMP ? CONV_TEMP_READS : (W_ == 32 && XF && !GN ? 0 : CONV_3X3_READS))>(p, acc, nw0, lane_e, estage, rowfn, flushfn);
        }
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    if (my_donor) run_item(0, std::true_type{});
    for (int it = my_donor; it < my_items; ++it) run_item(it, std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------------------------
// 0 = this launch is not one of the haloed kernels' shapes (the caller keeps its v3d_gemm path), else the HaloGeom variant id
int v3d_conv_halo_variant(const V3dGemmParams& p, int mode) {
    auto al = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) % a) == 0; };
    if (p.N % 320 || p.K % 32 || p.K * 2 > 65536 || p.out_fp32 || p.split_n > 1) return 0;
    if (!e4_ok(p, 96, 80)) return 0;                                // the kernels only carry the hand-managed epilogue
    if (p.A2 && (p.K1 <= 0 || p.K1 >= p.K || p.K1 % 32 || p.lda2 % 8 || !al(p.A2, 16))) return 0;
    if (p.ldo % 8 || !al(p.out, 16)) return 0;
    if (p.add && (!al(p.add, 16) || p.add_ld % 4)) return 0;
    if (p.res1 && (!al(p.res1, 16) || p.ldr1 % 8)) return 0;
    if (p.res2 && (!al(p.res2, 8) || p.ldr2 % 4)) return 0;
    if (p.bias && !al(p.bias, 16)) return 0;
    if (p.gn_in && (p.gn_in_rps <= 0 || !al(p.gn_in, 16))) return 0;
    if (p.gn_in && !p.gn_in_silu) return 0;                         // the operand path applies GroupNorm + SiLU (every ResBlock half); a bare norm keeps the apply pass
    if (p.gn_stats && (80 % p.gn_cpg || p.gn_rps % 16)) return 0;
    if (mode == V3D_GEMM_CONV3X3) {
        if (p.stride != 1 || p.upshift != 0 || p.pad_lo != 1 || p.Hin != p.Hout || p.Win != p.Wout) return 0;
        if (p.M % 192) return 0;
        const int W = p.Wout, H = p.Hout;
        if (W == 8) {
            // three whole 8 x 8 images per tile: no line of another image is ever needed; one statistics group per image
            if (H != 8 || (p.gn_in && p.gn_in_rps != 64)) return 0;
            if (p.gn_stats && p.gn_nslots < p.gn_rps / 96 + 2) return 0;
            return 5;
        }
        if (W != 64 && W != 32 && W != 16) return 0;
        if (H < 192 / W + 1) return 0;                              // a halo (tile rows + 2 lines) touches at most two images
        if (p.gn_in && p.gn_in_rps != (long long)H * W) return 0;   // one (scale, shift) row per image: the 2-D GroupNorm of the ResBlocks
        if (p.gn_stats && p.gn_nslots < p.gn_rps / 96 + 2) return 0;
        return W == 64 ? 1 : (W == 32 ? 2 : 3);
    }
    if (mode == V3D_GEMM_CONVT3) {
        if (p.T % 6 || p.S % 32 || p.M % ((long long)p.T * p.S)) return 0;
        if (p.gn_in && p.gn_in_rps != (long long)p.T * p.S) return 0;      // the 3-D GroupNorm: one row per sample
        if (p.gn_stats && (p.gn_rps != (long long)p.T * p.S || p.gn_nslots < (p.T / 6) * (p.S / 32) * 2)) return 0;
        return 4;
    }
    return 0;
}

template <int HM, int W_>
static int conv_halo_launch_t(const V3dGemmParams& p0, hipStream_t st) {
    V3dGemmParams p = p0;
    p.mt = (int)(p.M / 192);
    p.nt = (int)(p.N / 320);
    const int ntiles = p.mt * p.nt;
    // (stream-K tail: the last round's tiles are cut at 32-channel chunks and shared out over all CUs; at least 4 chunks per piece and 2 chunks
    // (18 / 6 steps) of idling removed)
    const int grid = v3d_sk_plan(p, ntiles, (int)(p.K / 32), 4, 2, (size_t)192 * 320 * 4, (void*)st);
    const bool xf = p.gn_in != nullptr, gn = p.gn_stats != nullptr;
    v3d_note_launch(5, 192, 320, ntiles, 1, p.sk_tail);
    if (xf && gn) hipLaunchKernelGGL((conv_halo_kernel<HM, W_, true, true>), dim3(grid), dim3(512), 0, st, p, ntiles);
    else if (xf) hipLaunchKernelGGL((conv_halo_kernel<HM, W_, true, false>), dim3(grid), dim3(512), 0, st, p, ntiles);
    else if (gn) hipLaunchKernelGGL((conv_halo_kernel<HM, W_, false, true>), dim3(grid), dim3(512), 0, st, p, ntiles);
    else hipLaunchKernelGGL((conv_halo_kernel<HM, W_, false, false>), dim3(grid), dim3(512), 0, st, p, ntiles);
    return v3d_check_launch("v3d_gemm(haloed)");
}

int v3d_conv_halo_launch(const V3dGemmParams& p, int variant, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
        case 1: return conv_halo_launch_t<HM_CONV, 64>(p, st);
        case 2: return conv_halo_launch_t<HM_CONV, 32>(p, st);
        case 3: return conv_halo_launch_t<HM_CONV, 16>(p, st);
        case 4: return conv_halo_launch_t<HM_TEMP, 32>(p, st);
        case 5: return conv_halo_launch_t<HM_CONV, 8>(p, st);
    }
    v3d_set_error("v3d_gemm(haloed): unknown variant %d", variant);
    return V3D_ERR_ARG;
}

