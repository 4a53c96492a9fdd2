/*
 * v3d_hip.h — C ABI of libv3d_hip.so: the hand-written gfx950 (MI355X / CDNA4) kernels behind the
 * V3D dense-multi-view hot path (EulerEDM loop -> VideoUNet -> VideoDecoder).
 *
 * The reference (heheyas/V3D) has no native layer: every device op on this path is a PyTorch ATen /
 * xformers call made from Python.  Each entry point below therefore names the *reference call site*
 * whose device work it replaces (paths relative to the reference tree).  The Python plugin classes in
 * v3d_amd/sgm/ (same `target:` API as the reference sgm/ package) are the only callers.
 *
 * Conventions
 *   - every entry returns 0 on success, <0 on error; v3d_last_error() gives a thread-local message
 *   - all pointers are DEVICE pointers on the current device; the caller allocates every output
 *   - kernels are stateless and re-entrant per stream; `stream` is a hipStream_t passed as void*
 *   - activations are bf16, "channels last": act[n_img][H*W][C] row-major; n_img = (cfg*B*T) ordered (b t)
 *   - weights are bf16, K-contiguous: W[taps][N][K]; vectors (bias, norm affine, per-image adds) are fp32
 *   - all matrix contractions accumulate in fp32 on MFMA; norm statistics are fp32
 */
#ifndef V3D_HIP_H
#define V3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V3D_ABI_VERSION 5

typedef void* v3d_stream_t; /* hipStream_t */

enum {
    V3D_OK = 0,
    V3D_ERR_ARG = -1,    /* bad argument (shape / alignment / null pointer) */
    V3D_ERR_LAUNCH = -2, /* HIP launch failure */
};

int v3d_abi_version(void);
const char* v3d_last_error(void);
/* Device properties of the current device: out[0]=CU count, out[1]=LDS bytes/CU (as reported), out[2]=wave size, out[3]=gfx arch number (950) */
int v3d_device_info(int32_t* out4);

/* ------------------------------------------------------------------------------------------------
 * v3d_gemm — multi-tap MFMA contraction with fused epilogue.
 *
 *   acc[m][n] = sum_{tap} sum_{k<K} A[src(m, tap)][k] * W[tap][n][k]          (fp32 accumulate)
 *   v         = acc + bias[n] + add[(m / add_rpg) * add_ld + n]
 *   if geglu:   v = v_value * gelu_erf(v_gate)        (W rows packed in 16-row value/gate groups, N_out = N/2)
 *   out[m][n] = c_acc * v + c_res1 * res1[m][n] + c_res2 * res2[m][n]         (bf16 or fp32 store)
 *   (c_* come from coef[(m / coef_rpg)*3 + {0,1,2}] when coef != NULL, else from the scalar fields)
 *
 * mode V3D_GEMM_LINEAR : 1 tap, src(m) = m.
 *      replaces nn.Linear / 1x1 Conv2d:  sgm/modules/attention.py:95-118,277-283,692-704;
 *      sgm/modules/video_attention.py:50-71,219-224; sgm/modules/diffusionmodules/openaimodel.py:294-300,324;
 *      sgm/modules/diffusionmodules/model.py:161-172 (VAE q/k/v/proj 1x1); batched form = the two SDPA
 *      GEMMs of the VAE AttnBlock (model.py:180-201).
 * mode V3D_GEMM_CONV3X3: 9 taps (ky,kx) row-major; rows are output pixels (img, oy, ox) of an Hout x Wout map;
 *      input pixel = (oy*stride + ky - 1, ox*stride + kx - 1) in the logical input of size (Hin*up) x (Win*up),
 *      zero outside; up=2 reads the nearest-neighbour 2x upsample of the stored Hin x Win map on the fly.
 *      replaces Conv2d 3x3 pad 1 (openaimodel.py:270,307-313; video_model.py:189,439; model.py:111-119),
 *      Downsample stride 2 (openaimodel.py:202-209), Upsample nearest+conv (openaimodel.py:164-166; model.py:67-71).
 * mode V3D_GEMM_CONVT3 : 3 taps along the frame axis; rows are (frame, s) with S rows per frame; tap dt reads
 *      row m + (dt-1)*S when tmin <= (frame % T) + dt - 1 <= tmax, else zero.
 *      replaces Conv3d (3,1,1) pad (1,0,0) (video_model.py:42-55 via openaimodel.py:267-313; temporal_ae.py:32-44).
 *      With frame sharding the caller points A at a buffer carrying +-1 halo frames and widens [tmin,tmax]: either one sample
 *      with its halo frames in line (a_row0 = S, frames -1 .. T), or (ABI 3) the split-halo layout `halo_rows` = B*S > 0 for B
 *      samples in one launch: rows [a_row0 - B*S, a_row0) hold frame -1 of sample 0 .. B-1 (received from the previous rank),
 *      rows [a_row0, a_row0 + M) the B*T local frames, rows [a_row0 + M, a_row0 + M + B*S) frame T of every sample.
 * ---------------------------------------------------------------------------------------------- */
enum { V3D_GEMM_LINEAR = 0, V3D_GEMM_CONV3X3 = 1, V3D_GEMM_CONVT3 = 2 };

typedef struct v3d_gemm_args {
    const void* A;      /* bf16 [rows][lda] */
    const void* W;      /* bf16 [taps][N][K] */
    void* out;          /* bf16 or fp32 [M][ldo] */
    const float* bias;  /* [N] or NULL (packed order when geglu) */
    const float* add;   /* row-group vectors or NULL */
    const void* res1;   /* bf16 [M][ldr1] or NULL */
    const void* res2;   /* bf16 [M][ldr2] or NULL */
    const float* coef;  /* [groups][3] or NULL */
    int64_t M, N, K;    /* N = rows of W per tap (2*N_out when geglu); K = contraction per tap, K % 8 == 0 */
    int64_t lda, ldw, ldo, ldr1, ldr2; /* row strides (elements): A, W (0 -> K; lets a K-slice of a wider matrix be used), out, res1, res2 */
    int64_t a_rows;     /* rows of A addressable behind the pointer (hardware bounds check; < 4 GiB total) */
    int64_t a_row0;     /* row offset added to every source row (CONVT3 halo frames in front of frame 0), >= 0 */
    int64_t add_rpg, add_ld; /* rows per add group, stride (floats) between groups' vectors */
    int64_t coef_rpg;
    float c_acc, c_res1, c_res2;
    int32_t mode, geglu, out_fp32;
    int32_t Hin, Win, Hout, Wout, stride, up; /* CONV3X3 */
    int32_t T, tmin, tmax;                    /* CONVT3 (S = rows per frame) */
    int64_t S;
    int32_t batch;                            /* >= 1; grid.y */
    int32_t pad_mode;                         /* CONV3X3 (ABI 2): 0 = one zero pixel on every side (Conv2d padding=1); 1 = right/bottom only,
                                                 F.pad(x,(0,1,0,1)) + padding=0: the VAE encoder's Downsample (diffusionmodules/model.py:74-91) */
    int64_t sA, sW, sO;                       /* element strides between batches (A, W, out) */
    int64_t halo_rows;                        /* CONVT3 (ABI 3): 0 = dense frames; B*S = split-halo layout, see above */
    /* GroupNorm statistics of the OUTPUT, gathered by the epilogue so that the next GroupNorm needs no statistics pass (ABI 3; slots ABI 4):
     * gn_stats[(m / gn_rps)][slot][n / gn_cpg][2] = (sum, sumsq) over the bf16-rounded out[m][n] of ONE writer's rows (layout of
     * v3d_groupnorm_stats: gn_nslots slots per statistics group, caller zeroes, every slot is written at most once with plain stores - no
     * atomics, the sums are bit-reproducible).  NULL = off.  Needs a dense bf16 out (ldo == N), !geglu, batch 1, 32 groups of an even
     * number of channels (N == 32 * gn_cpg), gn_rps % 16 == 0, M % gn_rps == 0, gn_nslots >= gn_rps / 64 + 2.  The persistent big-tile
     * kernels store the sums from their epilogue; every other launch runs v3d_groupnorm_stats on the output before returning. */
    float* gn_stats;
    int64_t gn_rps;                           /* rows per statistics group (imgs_per_stat * S) */
    int32_t gn_cpg;                           /* channels per group */
    int32_t gn_in_silu;                       /* (ABI 4) see gn_in_table */
    int64_t gn_nslots;                        /* (ABI 4) slots per statistics group of gn_stats */
    /* (ABI 4) GroupNorm (+SiLU) of the INPUT applied in the operand path: the "GN -> SiLU -> conv" of every ResBlock half
     * (openaimodel.py:267-271,302-314; video_model.py:42-55) and the "GN -> proj_in" of the transformer (attention.py:130-133,692-704)
     * without the normalised tensor ever existing in memory.  With gn_in_table != NULL, A holds the RAW tensor and the contraction runs on
     *     a(row, k) = act(A[row][k] * table[row / gn_in_rps][k][0] + table[row / gn_in_rps][k][1]),  act = SiLU when gn_in_silu, rounded to
     * bf16 (the value v3d_groupnorm_apply would have stored); zero padding stays zero.  table = v3d_groupnorm_finalize's (scale, shift)
     * table [n_stat][K][2] fp32, gn_in_rows = its n_stat.  A2 != NULL: the input is the channel concatenation A[:, :K1] | A2[:, :K - K1]
     * (th.cat([h, hs.pop()], 1) of the U-Net's up path, video_model.py:483, never materialised), row stride lda2, same row count.
     * Only the LDS-haloed kernels implement this (v3d_gemm_gn_in_supported says whether a call is one of their shapes); any other call
     * with gn_in_table set is refused with V3D_ERR_ARG - normalise with v3d_groupnorm_apply first.  Since round 4 the operand path always
     * applies SiLU (every ResBlock half does): a call with gn_in_silu == 0 is not one of the kernels' shapes (v3d_gemm_gn_in_supported = 0). */
    const float* gn_in_table;
    int64_t gn_in_rps;                        /* source rows per table row (S for a 2-D norm, T*S for the 3-D norm) */
    int64_t gn_in_rows;                       /* rows (statistics groups) of the table */
    const void* A2;
    int64_t K1, lda2;
} v3d_gemm_args;

int v3d_gemm(const v3d_gemm_args* args, v3d_stream_t stream);
/* 1 when v3d_gemm would run `args` (with its gn_in_table / A2 fields) on a kernel that normalises the operand in flight, else 0 */
int v3d_gemm_gn_in_supported(const v3d_gemm_args* args);

/* FeedForward(GEGLU) of the transformer blocks, fused end to end (sgm/modules/attention.py:82-113: GEGLU = Linear(C, 2*hidden) ->
 * value * gelu(gate); FeedForward = GEGLU, Dropout(0), Linear(hidden, C)); the hidden tensor never leaves the CU.
 *   out[m][:] = ca * (W2 . geglu(W1 x[m] + b1) + b2) + c1 * res1[m] + c2 * res2[m]      (ca, c1, c2 = c_acc, c_res1, c_res2, or
 *                                                                                         coef[(m / coef_rpg)][3] when coef != NULL)
 * x [M][C] bf16 (row stride ldx); b2 [C] fp32; out / res bf16.  The weights are packed in the order the 32x32 MFMA tiles hand
 * their rows to the lanes (v3d_amd/engine/packing.py ff_fused_pack; a, h, t in {0,1}, g in 0..3, c in 0..3, s = 32-channel slab):
 *   W1p [2*hidden][C] bf16, b1 [2*hidden] fp32: row 64 s + 32 a + 8 g + 4 h + c = the (g odd ? gate : value) row of hidden
 *                                               channel 32 s + 16 a + 8 (g >> 1) + 4 h + c
 *   W2p [C][hidden] bf16:                       W2p[o][32 s + 16 a + 8 h + 4 t + c] = W2[o][32 s + 16 a + 8 t + 4 h + c]
 * and both matrices are then stored in the order the kernel's LDS-DMA stream reads them (ff_dma_tile_index): slabs (64 rows of W1p /
 * 32 columns of W2p), a slab as 1-KiB pieces of 16 rows x 32 columns ordered (column block, row block), and inside a piece the 16-byte
 * unit 4 r + p (r = row 0..15, p = 0..3) holds columns 8 (p ^ swz(r)) .. + 7 of row r, swz(r) = {0,2,3,1}[(r >> 2) & 3].
 * Requires C == 320 (the 64x64 level; wider levels use two v3d_gemm launches), M % 128 == 0, hidden % 64 == 0, hidden >= 128. */
int v3d_ff_fused(const void* x, int64_t ldx, const void* W1p, const float* b1, const void* W2p, const float* b2,
                 const void* res1, int64_t ldr1, const void* res2, int64_t ldr2, const float* coef, int64_t coef_rpg,
                 float c_acc, float c_res1, float c_res2, void* out, int64_t ldo, int64_t M, int32_t C, int32_t hidden,
                 v3d_stream_t stream);

/* LayerNorm + the same feed-forward in ONE launch - BasicTransformerBlock `x = ff(norm3(x)) + x` (sgm/modules/attention.py:575-577) and the
 * VideoTransformerBlock's `norm3 / ff` (video_attention.py:133-140): the rows are normalised in registers, out = ... W1 ((x[m] - mean) * rstd)
 * with the LayerNorm affine folded into the weights by the caller: W1 <- W1 diag(gamma), b1 <- b1 + W1 beta BEFORE the packing above
 * (v3d_amd/engine/packing.py ff_fused_pack(..., ln=(gamma, beta))).  res1 is normally x itself.  ln_eps > 0.  Same shape rules. */
int v3d_ln_ff_fused(const void* x, int64_t ldx, float ln_eps, const void* W1p, const float* b1, const void* W2p, const float* b2,
                 const void* res1, int64_t ldr1, const void* res2, int64_t ldr2, const float* coef, int64_t coef_rpg,
                 float c_acc, float c_res1, float c_res2, void* out, int64_t ldo, int64_t M, int32_t C, int32_t hidden,
                 v3d_stream_t stream);
/* LayerNorm + q | k | v projection of a transformer block in one kernel (ABI 3; C = 320, the 64x64 level).  Reference:
 * BasicTransformerBlock / VideoTransformerBlock x -> norm1(x) -> attn1.to_q / to_k / to_v (sgm/modules/attention.py:556-563,286-290;
 * sgm/modules/video_attention.py:122-125; the three Linears have no bias).  The LayerNorm affine is folded through the projection at pack time
 * (W' = W diag(gamma), bias = W beta: v3d_amd/engine/packing.py ln_proj_pack); xh = bf16((x[m] - mean) * rstd) stays in registers;
 *   out[m][n]                    = sum_k xh[m][k] W'[n][k] + bias[n]     for n <  n_rm   (row-major, row stride ldo)
 *   outT[m / S][n - n_rm][m % S] = sum_k xh[m][k] W'[n][k] + bias[n]     for n >= n_rm   (V^T layout of v3d_attn_spatial: keys contiguous)
 * Wp = the concatenated [N][C] weight (q ; k ; v, gamma folded in) stored in the kernel's LDS-DMA order (ff_dma_tile_index(N, C, 64, C): slabs of
 * 64 rows, 1-KiB pieces of 16 rows x 32 columns ordered (column block, row block), 16-byte unit 4 r + p of a piece = columns 8 (p ^ swz(r)) .. of
 * row r, swz(r) = {0,2,3,1}[(r >> 2) & 3]).  M % 128 == 0, N % 64 == 0 (128 <= N <= 960), n_rm % 64 == 0; S % 128 == 0 when n_rm < N. */
int v3d_ln_proj(const void* x, int64_t ldx, float eps, const void* Wp, const float* bias, void* out, int64_t ldo, void* outT,
                int64_t M, int32_t C, int32_t N, int32_t n_rm, int64_t S, v3d_stream_t stream);
/* sizeof(v3d_gemm_args) as compiled into the library: lets a foreign-language binding verify its struct mirror */
int v3d_sizeof_gemm_args(void);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm over channels-last activations, optionally over two channel-concatenated sources (the U-Net skip concat
 * th.cat([h, hs.pop()], 1) at video_model.py:483 is never materialised), in three steps (ABI 4):
 *   v3d_groupnorm_stats:    stats[g_img][slot][group][2] = (sum, sumsq) over the rows one block read: every block owns ONE slot
 *                           (slot = (img % imgs_per_stat) * blocks_per_image + block) and writes it with a plain store - no atomics, the
 *                           result does not depend on scheduling.  Caller zeroes `stats` [n_img / imgs_per_stat][nslots][groups][2];
 *                           nslots >= imgs_per_stat (more slots = more blocks; the library fits its grid to nslots).
 *                           imgs_per_stat = 1 for the 2-D GroupNorm, = frames per sample for the 3-D GroupNorm whose statistics span all
 *                           frames (openaimodel.py:267-271,302-305 with dims=3).  v3d_gemm's gn_stats epilogue fills the same layout.
 *   v3d_groupnorm_finalize: adds the slots up in a fixed order in fp64, mean / rstd in fp64, and writes the affine the normalisation
 *                           amounts to per (statistics group, channel):  table[stat][c] = (gamma[c] rstd, beta[c] - mean gamma[c] rstd),
 *                           y = x * table[..][0] + table[..][1].  `count` = elements per group (global under frame sharding).
 *                           `sums` [n_stat][groups][2] fp64 is the hand-off of the frame-sharded runtime: stats != NULL writes it (when
 *                           given), stats == NULL reads it (after the all-reduce over ranks); table == NULL skips the table.
 *   v3d_groupnorm_apply:    y = x * scale + shift, then SiLU when silu != 0; bf16 out [n_img*S][C1+C2].  Consumers that normalise their
 *                           operand in flight (v3d_gemm gn_in_table) take the table instead.
 * replaces GroupNorm32 / Normalize (+SiLU / swish): diffusionmodules/util.py:259-276; attention.py:130-133;
 *   model.py:52-55,43-45; openaimodel.py:267-271,302-305.
 * ---------------------------------------------------------------------------------------------- */
int v3d_groupnorm_stats(const void* x1, int64_t C1, const void* x2, int64_t C2, float* stats, int64_t nslots,
                        int64_t n_img, int64_t S, int32_t groups, int64_t imgs_per_stat, v3d_stream_t stream);
/* ABI 5: statistics AND table in one launch - v3d_groupnorm_stats whose LAST block of each statistics group to finish runs the fold of
 * v3d_groupnorm_finalize on the group's slots (same fp64 fixed-order sums, so the table is bit-reproducible whichever block is last):
 * the launch-bound finalize kernel behind every stand-alone statistics pass disappears (41 per U-Net evaluation).  tickets: [n_img /
 * imgs_per_stat] uint32, ZERO on entry, left zero.  groups even and <= 32.  Same reference call sites as the three-step form. */
int v3d_groupnorm_stats_table(const void* x1, int64_t C1, const void* x2, int64_t C2, float* stats, int64_t nslots, uint32_t* tickets,
                              int64_t n_img, int64_t S, int32_t groups, int64_t imgs_per_stat, const float* gamma, const float* beta,
                              double count, float eps, float* table, v3d_stream_t stream);
int v3d_groupnorm_finalize(const float* stats, int64_t nslots, double* sums, int64_t n_stat, int32_t groups,
                           const float* gamma, const float* beta, int64_t C, double count, float eps, float* table,
                           v3d_stream_t stream);
int v3d_groupnorm_apply(const void* x1, int64_t C1, const void* x2, int64_t C2, const float* table, void* out,
                        int64_t n_img, int64_t S, int64_t imgs_per_stat, int32_t silu, v3d_stream_t stream);

/* GroupNorm (+SiLU) of SMALL statistics groups in ONE launch (the 8 x 8 level of the U-Net, the 16 x 16 level's transformer norms: the
 * three-step form is launch-bound there): a block keeps (statistics group, 1..8 channel groups) in registers, reduces in a fixed order,
 * normalises what it holds.  Same result contract as stats -> finalize -> apply (fp64 sums, scale = gamma rstd, shift = beta - mean scale,
 * bf16 out); deterministic.  v3d_groupnorm_small_supported: 32 groups of a multiple of 8 channels and imgs_per_stat * S * (channels of the
 * block's groups) <= 49152 elements with row segments >= 64 bytes; anything else is refused (use the three-step form). */
int v3d_groupnorm_small_supported(int64_t C1, int64_t C2, int64_t S, int32_t groups, int64_t imgs_per_stat);
int v3d_groupnorm_small(const void* x1, int64_t C1, const void* x2, int64_t C2, const float* gamma, const float* beta, void* out,
                        int64_t n_img, int64_t S, int32_t groups, int64_t imgs_per_stat, float eps, int32_t silu, v3d_stream_t stream);

/* LayerNorm over the last dim of bf16 [M][C]; optional fp32 row-group vector added first:
 *   xs = x + add[(m / add_rpg) * add_ld + c];  if xsum_out: xsum_out[m] = bf16(xs);  out = LN(xs)*gamma + beta
 * replaces nn.LayerNorm (attention.py:525-527; video_attention.py:51,79,93-94) and the frame-position add
 * `x_mix = x + emb` (video_attention.py:286-287). */
int v3d_layernorm(const void* x, const float* add, int64_t add_rpg, int64_t add_ld, void* xsum_out,
                  const float* gamma, const float* beta, void* out, int64_t M, int64_t C, float eps,
                  v3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Spatial self-attention, head dim 64, no mask: out[n][s][h*64+d] = softmax(q k^T * scale) v.
 *   q[n][s][h*64+d] at q + (n*S+s)*ldq ; k likewise at k + (n*S+s)*ldk ; vT[n][h*64+d][s] (keys contiguous).
 * replaces F.scaled_dot_product_attention / xformers.ops.memory_efficient_attention in
 *   CrossAttention.forward (attention.py:337-341) / MemoryEfficientCrossAttention.forward (attention.py:432-444)
 *   for BasicTransformerBlock.attn1 (attention.py:559-569).
 * Self-attention only.  The cross-attention of the path (attn2, attention.py:570-574; VideoTransformerBlock.attn2, video_attention.py:126-132)
 * attends to ONE context token per image in every SVD / V3D config (the CLIP image embedding), where softmax == 1 and the layer reduces to
 * to_out(to_v(context)): the engine folds it into a per-image vector at pack time (v3d_amd/engine/unet.py asserts context.shape[1] == 1).  The
 * general N-token form (attention.py:286-349) has no entry point in this library.
 * ---------------------------------------------------------------------------------------------- */
int v3d_attn_spatial(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vT, void* out,
                     int64_t ldo, int64_t n_img, int64_t S, int32_t heads, float scale, v3d_stream_t stream);

/* fp8 (OCP e4m3fn) variant of the spatial self-attention (ABI 3) - BASELINE.json configs[4] ("scene" shape, 9216 tokens at level 0) names
 * fp8 MFMA attention; the reference has no fp8 path, parity is stated against v3d_attn_spatial (tests: cosine >= 0.995).  Never used by the
 * headline benchmark.  Three steps:
 *   v3d_quant_fp8_tiles: x [n_img*S][ldx] bf16, first `ncols` columns (the q | k projection, ncols = 2*heads*64) -> x8 [n_img*S][ld8] e4m3 bytes
 *     and scales[n_img][ceil(S/64)][ncols/64] fp32: one dequantisation factor (amax / 448) per 64-row x 64-column tile.
 *   v3d_quant_fp8_slab:  vT [n_img][heads*64][S] bf16 -> v8 (same shape, e4m3) and vscale[n_img][heads]: one factor per (image, head) slab
 *     (`amax_scratch`: n_img*heads uint32 of device scratch).
 *   v3d_attn_spatial_fp8: out = softmax(q k^T * scale) v on v_mfma_f32_32x32x64_f8f6f4 (QK^T and P.V both fp8 x fp8 -> fp32; P as e4m3(256 p)),
 *     q at qk8 + (n*S+s)*ld8 + h*64, k at the same row + heads*64; out bf16 [n_img*S][ldo].  S % 16 == 0. */
int v3d_quant_fp8_tiles(const void* x, int64_t ldx, void* x8, int64_t ld8, float* scales, int64_t n_img, int64_t S, int32_t ncols,
                        v3d_stream_t stream);
int v3d_quant_fp8_slab(const void* vT, void* v8, float* vscale, void* amax_scratch, int64_t n_img, int64_t S, int32_t heads, v3d_stream_t stream);
int v3d_attn_spatial_fp8(const void* qk8, int64_t ld8, const float* scales, const void* v8, const float* vscale, void* out, int64_t ldo,
                         int64_t n_img, int64_t S, int32_t heads, float scale, v3d_stream_t stream);

/* Temporal self-attention over the frame axis (Tq local queries x Tk keys, Tq,Tk <= 32), head dim 64.
 *   problem p = (b, s, h); element (b, t, s, h*64+d) of q at q + b*q_sb + t*q_st + s*q_ss + h*64 + d (same for k, v, out).
 * replaces VideoTransformerBlock.attn1 on "(b t) s c -> (b s) t c" (video_attention.py:114,122-125) without the
 * two full-tensor transposes. */
int v3d_attn_temporal(const void* q, int64_t q_sb, int64_t q_st, int64_t q_ss,
                      const void* k, const void* v, int64_t kv_sb, int64_t kv_st, int64_t kv_ss,
                      void* out, int64_t o_sb, int64_t o_st, int64_t o_ss,
                      int64_t B, int32_t Tq, int32_t Tk, int64_t S, int32_t heads, float scale,
                      v3d_stream_t stream);

/* Single-head self-attention of the VAE AttnBlock (ABI 3; reference sgm/modules/diffusionmodules/model.py:180-201: q, k, v 1x1 convs ->
 * F.scaled_dot_product_attention over ONE head of width C -> proj_out), streamed softmax: no [S][S] tensor is ever written.
 *   out[n][s][c] = sum_j softmax_j(q[n][s] . k[n][j] * scale) * v[n][j][c] + bias[c]
 *   q at q + (n*S+s)*ldq, k at k + (n*S+j)*ldk (bf16, C channels), vT[n][c][j] (keys contiguous: the value projection is computed as a
 *   swapped GEMM, its bias is added here - softmax rows sum to 1); out bf16 at out + (n*S+s)*ldo; bias fp32 [C] or NULL.
 * C = 512 is the V3D / SVD first stage (4096 tokens at 512x512, 9216 at 576x1024); C = 256 / 128 serve reduced-width test models. */
int v3d_attn_vae_d512(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vT, const float* bias, void* out,
                      int64_t ldo, int64_t n_img, int64_t S, int32_t C, float scale, v3d_stream_t stream);

/* Row softmax: out_bf16[r][j] = softmax_j(in_f32[r][:]) ; used by the VAE AttnBlock (model.py:190-192) in
 * its unfused (batched GEMM -> softmax -> batched GEMM) form. */
int v3d_softmax_rows(const float* in, void* out, int64_t rows, int64_t L, v3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Small elementwise kernels
 * ---------------------------------------------------------------------------------------------- */
/* sinusoidal embedding [cos | sin], freqs exp(-ln(max_period) * i / half)  (diffusionmodules/util.py:207-231) */
int v3d_timestep_embedding(const float* t, void* out_bf16, int64_t n, int32_t dim, float max_period, v3d_stream_t stream);
/* out_bf16 = silu(in_f32 [+ in2_f32]) : emb = time_embed(..) + label_emb(..) then nn.SiLU of emb_layers (openaimodel.py:294-300) */
int v3d_silu_add(const float* in, const float* in2, void* out_bf16, int64_t n, v3d_stream_t stream);
/* EDM v-prediction scalings (denoiser_scaling.py:51-59): c_skip, c_out, c_in, c_noise per image */
int v3d_edm_scalings(const float* sigma, float* c_skip, float* c_out, float* c_in, float* c_noise, int64_t n, v3d_stream_t stream);
/* U-Net input assembly: out_bf16[n][s][c] = c<C1 ? x[n][c][s]*scale[n] : cond[n][c-C1][s], zero-padded to Cpad
 * (denoiser.py:36-37 `input * c_in`, wrappers.py:27 torch.cat((x, c["concat"]), 1), NCHW fp32 -> channels-last bf16) */
int v3d_pack_input(const float* x, const float* scale, int64_t C1, const float* cond, int64_t C2, void* out_bf16,
                   int64_t n, int64_t S, int64_t Cpad, v3d_stream_t stream);
/* The same assembly as the im2col of the U-Net's first convolution (openaimodel.py input_blocks[0]: conv_nd(dims, in_channels, model_channels, 3,
 * padding=1) on 8 input channels): out_bf16[n][y][x][tap*8 + c] = packed(n, y+dy-1, x+dx-1, c), tap = dy*3+dx, zero outside the image and in the
 * columns behind 72 (row width Kpad, a multiple of 8) - the convolution then is ONE GEMM with K = Kpad against the weight re-packed as
 * W[o][tap*8 + c] (the implicit-GEMM kernels need K % 32 == 0 per tap: with K = 8 the launch ran on the generic kernel, 258 us for 28 MB). */
int v3d_pack_input_im2col3x3(const float* x, const float* scale, int64_t C1, const float* cond, int64_t C2, void* out_bf16,
                             int64_t n, int32_t H, int32_t W, int64_t Kpad, v3d_stream_t stream);
/* 3x3 convolution with very few output channels (the U-Net's `out` conv: model_channels -> 4, openaimodel.py self.out[2]) as GEMM + gather:
 * y fp32 [n*H*W][ldy] holds, per UNSHIFTED pixel, the nine taps' products y[m][tap*C + c] = x[m] . W[tap][c] (one GEMM with N = 9 C rows);
 * out[m][c] = bias[c] + sum over the taps whose source pixel lies inside the image of y[m + (dy-1) W + (dx-1)][tap*C + c], fixed tap order. */
int v3d_tapsum3x3(const float* y, int64_t ldy, const float* bias, float* out, int64_t n, int32_t H, int32_t W, int32_t C,
                  v3d_stream_t stream);
/* denoised[n][c][s] = net[n][s][c] * c_out[n] + x[n][c][s] * c_skip[n]   (denoiser.py:36-39; net is fp32 channels-last, ld = ldn) */
int v3d_denoise_combine(const float* net, int64_t ldn, const float* x, const float* c_out, const float* c_skip,
                        float* out, int64_t n, int64_t C, int64_t S, v3d_stream_t stream);
/* CFG with per-frame scale (guiders.py:78-86): out[i] = xu[i] + scale[i % T] * (xc[i] - xu[i]), xu = x[:n], xc = x[n:] */
int v3d_cfg_combine(const float* x, const float* scale, float* out, int64_t n, int64_t T, int64_t chw, v3d_stream_t stream);
/* Euler step (sampling.py:96-110, sampling_utils.py:34-35): d = (x - den)/sigma[n]; out = x + (next[n]-sigma[n])*d */
int v3d_euler_step(const float* x, const float* den, const float* sigma, const float* next_sigma, float* out,
                   int64_t n, int64_t chw, v3d_stream_t stream);
/* Heun correction (HeunEDMSampler.possible_correction_step, sampling.py:221-237; to_d: sampling_utils.py:34-35):
 * d = (x - den)/sigma[n]; d_new = (euler - den2)/next[n]; out = next[n] > 0 ? x + (next[n]-sigma[n]) * (d + d_new)/2 : euler */
int v3d_heun_step(const float* x, const float* den, const float* euler, const float* den2, const float* sigma,
                  const float* next_sigma, float* out, int64_t n, int64_t chw, v3d_stream_t stream);
/* CLIP image front-end: FrozenOpenCLIPImageEmbedder.preprocess (sgm/modules/encoders/modules.py:645-657 =
 * kornia.geometry.resize(x, (S,S), "bicubic", align_corners=True, antialias): Gaussian blur (sigma = max((in/out - 1)/2, .001), kernel
 * int(max(4 sigma, 3)) made odd, reflect border) when down-scaling, then bicubic A = -0.75; (x+1)/2; (x - mean)/std) fused with the patch
 * unfold of open_clip VisionTransformer.conv1 (kernel = stride = P): img [B][3][H][W] fp32 in [-1,1] -> patches [B*(S/P)^2][Kpad] bf16,
 * column c*P*P + ky*P + kx (= conv1.weight.reshape(width, 3*P*P) order), columns >= 3*P*P zero. mean3 / std3 are HOST pointers. */
int v3d_clip_preprocess(const float* img, int64_t B, int32_t H, int32_t W, int32_t S, int32_t P, int32_t antialias,
                        const float* mean3, const float* std3, void* patches_bf16, int32_t Kpad, v3d_stream_t stream);
/* out = gelu(in), exact-erf form (nn.GELU in open_clip's ViT MLP), bf16, n % 8 == 0 */
int v3d_gelu_bf16(const void* in, void* out, int64_t n, v3d_stream_t stream);
/* Output stage (scripts/pub/V3D_512.py:286-303): frames x [n][C][S] fp32 in [-1,1] ->
 * out [n][S][C] uint8 = (uint8)(clamp((x + 1) / 2, 0, 1) * 255), i.e. numpy's truncating astype; C <= 4.  Bit-exact with the reference. */
int v3d_frames_to_uint8(const float* x, void* out_u8, int64_t n, int32_t C, int64_t S, v3d_stream_t stream);
/* x[n][...] *= s  (sampling.py:50) ; generic y = a*x + b on fp32 */
int v3d_axpb_f32(const float* x, float a, float b, float* out, int64_t n, v3d_stream_t stream);
/* AlphaBlender coefficients (diffusionmodules/util.py:341-369): for mixer i with alpha_i = sigmoid(mix_factor_i)
 * and image g: a = ioi[g] ? 1 : alpha_i ; kind 0 (VideoResBlock): (1-a, 1, 0) ; kind 1 (SpatialVideoTransformer): (1-a, 1-a, a)
 * out[i][g][3] */
int v3d_blend_coefs(const float* alpha, const int32_t* kind, const float* ioi, float* out, int64_t n_mixers, int64_t n_img, v3d_stream_t stream);
/* layout / dtype moves: fp32 NCHW <-> bf16 channels-last (zero-pad channels up to Cpad) */
int v3d_nchw_to_nhwc_bf16(const float* x, float scale, void* out_bf16, int64_t n, int64_t C, int64_t S, int64_t Cpad, v3d_stream_t stream);
/* AE3DConv.time_mix_conv (temporal_ae.py:94-107): Conv3d (3,1,1) on Cc<=4 channels over frames of a fp32 channels-last
 * [B*T][S][ld] map, output fp32 NCHW [B*T][Cc][S]; w[co][ci][3], b[co]; tmin/tmax as in CONVT3; row0 = rows in front
 * of frame 0 (a leading halo frame under frame sharding) */
int v3d_tmix_small(const float* x, int64_t ld, const float* w, const float* b, float* out, int64_t B, int32_t T,
                   int64_t S, int32_t Cc, int32_t tmin, int32_t tmax, int64_t row0, v3d_stream_t stream);
/* dst_bf16[r][dst_off + c] = src_bf16[r][src_off + c], c < C  (strided 2-D bf16 copy; C % 8 == 0) */
int v3d_copy2d_bf16(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t C, v3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* V3D_HIP_H */
