// v3d_ff_fused: FeedForward(GEGLU) of the transformer blocks with the 4x-wide hidden tensor kept on the CU (gfx950).
//
// Reference: sgm/modules/attention.py:82-113 (GEGLU: Linear(C, 2*4C) -> value * gelu(gate); FeedForward: GEGLU, Dropout, Linear(4C, C)).
// The unfused pair (v3d_gemm with the GEGLU epilogue, then v3d_gemm) writes and re-reads a 377 MB hidden tensor per block at the
// 64x64 level; the second GEMM streams it at ~2.9 TB/s and the first one's epilogue (one GELU per output + the stores) costs as much
// as its main loop.  Here a block of 128 pixel rows walks the hidden dimension in slabs of 32 channels:
//   phase A   S[64 x 32px] = W1 slab (64 packed rows = 32 value + 32 gate, 16-interleaved) . x^T      (x fragments stay in registers)
//   GEGLU     h[32 ch] = (S_value + b) * gelu(S_gate + b)  - the lane now holds 8 hidden values of its pixel
//   phase B   out[C x 32px] += W2[:, slab] . h               (h is the MFMA operand straight from those registers: the K order of
//                                                             W2 inside a slab is permuted at pack time to the order the lanes hold)
// 4 waves, ONE per SIMD (512-VGPR budget: 160 output accumulators + 80 resident x fragments + 32 slab accumulators), both weight
// streams through a double-buffered LDS slab (buffer-load LDS-DMA, one barrier per slab), LDS fragments double-buffered in registers
// so a wave's ds_reads run under its own MFMAs.  Persistent over row blocks; the weight stream is block-independent, so the slab
// prefetch runs across block boundaries.
#include <stdlib.h>

#include "common.h"

namespace {

struct FFP {
    const bf16_t* x;
    const bf16_t* W1;
    const bf16_t* W2;
    const float* b1;
    const float* b2;
    const bf16_t* res1;
    const bf16_t* res2;
    const float* coef;
    bf16_t* out;
    long long M, ldx, ldr1, ldr2, ldo, coef_rpg;
    float c_acc, c_res1, c_res2;
    int hidden;
    unsigned w1_bytes, w2_bytes;
};

__device__ __forceinline__ int ff_swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // 64-byte LDS rows, see gemm.hip

__device__ unsigned long long g_ff_dbg[4 * 16 * 8];   // [wave][slab 8..23][stamp] of block 0 (timeline build only)

template <int C, bool DBG = false>
__global__ __launch_bounds__(256, 1) void ff_fused_kernel(FFP p) {
    constexpr int NT = C / 32;                 // k steps of phase A
    constexpr int NO = C / 16;                 // output channel fragments
    constexpr int W1_STAGE = 64 * 64;          // 64 packed rows x 64 B (32 k)
    constexpr int W1_BYTES = NT * W1_STAGE;
    constexpr int W2_BYTES = C * 64;           // C output rows x 64 B (the slab's 32 hidden channels)
    constexpr int SLAB_BYTES = W1_BYTES + W2_BYTES;
    constexpr int NP1 = NT * 4, NP2 = C / 16;
    static_assert((NP1 + NP2) % 4 == 0 && NP1 % 4 == 0, "pieces per wave");
    constexpr int PPW1 = NP1 / 4, PPW2 = NP2 / 4;
    constexpr int HALF = NO / 2;               // output fragments per staging pass
    constexpr int SROW = HALF * 32 + 16;
    constexpr int STAGE_REGION = 16 * SROW;
    static_assert(NO % 2 == 0 && (16 * HALF * 2) % 64 == 0, "staging geometry");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * SLAB_BYTES + 4 * STAGE_REGION + (DBG ? 4096 : 0)];
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(lds + 2 * SLAB_BYTES + 4 * STAGE_REGION);
    auto stamp = [&](int s_, int k) __attribute__((always_inline)) {
        if (DBG && blockIdx.x == 0 && s_ >= 8 && s_ < 24 && (threadIdx.x & 63) == 0) dbg[((threadIdx.x >> 6) * 16 + (s_ - 8)) * 8 + k] = __builtin_amdgcn_s_memtime();
    };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    unsigned char* stage = lds + 2 * SLAB_BYTES + wave * STAGE_REGION;

    // ---- weight-slab loader: 1-KiB pieces (16 rows x 64 B), lane l -> row l >> 2, 16-byte chunk (l & 3) ^ swz(row)
    const bufrsrc_t rsW1 = make_rsrc(p.W1, p.w1_bytes);
    const bufrsrc_t rsW2 = make_rsrc(p.W2, p.w2_bytes);
    const int prow = lane >> 2;
    const unsigned kchunk_b = (unsigned)(((lane & 3) ^ ff_swz(prow)) * 16);
    unsigned voff1[PPW1], voff2[PPW2];
#pragma unroll
    for (int i = 0; i < PPW1; ++i) {
        const int q = wave + 4 * i, t = q >> 2, r4 = q & 3;
        voff1[i] = (unsigned)(((r4 * 16 + prow) * C + t * 32) * 2) + kchunk_b;
    }
#pragma unroll
    for (int i = 0; i < PPW2; ++i) {
        const int r16 = wave + 4 * i;
        voff2[i] = (unsigned)(((r16 * 16 + prow) * p.hidden) * 2) + kchunk_b;
    }
    const int nslab = p.hidden / 32;
    int ld_slab = 0;
    auto issue_slab = [&](int buf) __attribute__((always_inline)) {
        unsigned char* sb = lds + buf * SLAB_BYTES;
        const int so1 = ld_slab * 64 * C * 2, so2 = ld_slab * 64;
#pragma unroll
        for (int i = 0; i < PPW1; ++i) {
            const int q = wave + 4 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (__attribute__((address_space(3))) void*)(sb + (q >> 2) * W1_STAGE + (q & 3) * 1024), 16,
                                                     (int)voff1[i], so1, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PPW2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (__attribute__((address_space(3))) void*)(sb + W1_BYTES + (wave + 4 * i) * 1024), 16,
                                                     (int)voff2[i], so2, 0, 0);
        if (++ld_slab == nslab) ld_slab = 0;
    };

    const int frag_off = fr * 64 + ((fq ^ ff_swz(fr)) * 16);   // fragment row fr, logical chunk fq
    const long long nblocks = p.M / 128;

    f32x4 acc2[2][NO];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int o = 0; o < NO; ++o) acc2[f][o] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue_slab(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int buf = 0;

    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const long long mw0 = blk * 128 + wave * 32;
        // the wave's 32 input rows as MFMA operand fragments, resident for the whole block
        bf16x8 xr[2][NT];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                xr[f][t] = *reinterpret_cast<const bf16x8*>(p.x + (mw0 + f * 16 + fr) * p.ldx + t * 32 + fq * 8);

        for (int s = 0; s < nslab; ++s) {
            stamp(s, 0);
            // bias first: vmcnt retires in order, a bias load issued behind the 15 slab DMAs would make the GEGLU wait for all of them
            float4 bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const float4*>(p.b1 + s * 64 + j * 16 + fq * 4);
            __builtin_amdgcn_sched_barrier(0);
            issue_slab(buf ^ 1);   // the other buffer was released by the barrier that ended the previous slab
            const unsigned char* sb = lds + buf * SLAB_BYTES;
            stamp(s, 1);

            // ---- phase A: S = W1 slab . x^T, fragments of step t+1 read while step t's MFMAs run
            f32x4 acc1[2][4];
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc1[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // W1 fragments run PD steps ahead of their MFMAs in a register ring (one wave per SIMD: nothing else hides the LDS latency)
            constexpr int PD = 2;
            bf16x8 wf[PD + 1][4];
#pragma unroll
            for (int t0 = 0; t0 < PD; ++t0)
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[t0][j] = *reinterpret_cast<const bf16x8*>(sb + t0 * W1_STAGE + frag_off + j * 1024);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t + PD < NT) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        wf[(t + PD) % (PD + 1)][j] = *reinterpret_cast<const bf16x8*>(sb + (t + PD) * W1_STAGE + frag_off + j * 1024);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        acc1[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t % (PD + 1)][j], xr[f][t], acc1[f][j], 0, 0, 0);
            }
            stamp(s, 2);
            // first W2 fragments on their way while the GELUs run
            constexpr int G = 5;   // W2 fragments per register batch
            static_assert(NO % G == 0, "phase B batching");
            bf16x8 w2[2][G];
#pragma unroll
            for (int g = 0; g < G; ++g) w2[0][g] = *reinterpret_cast<const bf16x8*>(sb + W1_BYTES + frag_off + g * 1024);

            // ---- GEGLU: fragments (0,1) = value / gate of hidden channels 0..15 of the slab, (2,3) = 16..31; a lane owns channels
            //      4 fq .. 4 fq + 3 of each -> 8 hidden values of pixel fr = one k-slice of phase B in the packed K order
            bf16x8 hB[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                float h[8];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const float4 bvv = bv[2 * pr], bgg = bv[2 * pr + 1];
                    const float vb[4] = {bvv.x, bvv.y, bvv.z, bvv.w}, gb[4] = {bgg.x, bgg.y, bgg.z, bgg.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[pr * 4 + r] = (acc1[f][2 * pr][r] + vb[r]) * gelu_erf_tight(acc1[f][2 * pr + 1][r] + gb[r]);
                }
                const u32x4 u = {pack2bf(h[0], h[1]), pack2bf(h[2], h[3]), pack2bf(h[4], h[5]), pack2bf(h[6], h[7])};
                hB[f] = __builtin_bit_cast(bf16x8, u);
            }

            stamp(s, 3);
            // ---- phase B: out += W2[:, slab] . h
#pragma unroll
            for (int ob = 0; ob < NO / G; ++ob) {
                if (ob + 1 < NO / G) {
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        w2[(ob + 1) & 1][g] = *reinterpret_cast<const bf16x8*>(sb + W1_BYTES + frag_off + ((ob + 1) * G + g) * 1024);
                }
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        acc2[f][ob * G + g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[ob & 1][g], hB[f], acc2[f][ob * G + g], 0, 0, 0);
            }
            stamp(s, 4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next slab has landed (and every LDS read of this one was consumed)
            stamp(s, 5);
            __builtin_amdgcn_s_barrier();
            stamp(s, 6);
            buf ^= 1;
        }

        // ---- block epilogue: out = c_acc (acc + b2) + c1 res1 + c2 res2, 16 rows x C/2 channels per staging pass
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const long long m = mw0 + f * 16 + fr;
            float ca = p.c_acc, c1 = p.c_res1, c2 = p.c_res2;
            if (p.coef) {
                const float* cf = p.coef + (m / p.coef_rpg) * 3;
                ca = cf[0]; c1 = cf[1]; c2 = cf[2];
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                constexpr int CPRO = HALF * 2;                 // 16-byte chunks per staged row
                const int col0 = hf * HALF * 16;
                if (p.res1) {                                  // coalesced residual rows -> LDS -> MFMA layout
                    const bf16_t* rz = p.res1 + (mw0 + f * 16) * p.ldr1 + col0;
#pragma unroll
                    for (int c0 = 0; c0 < 16 * CPRO; c0 += 64) {
                        const int c = c0 + lane;
                        *reinterpret_cast<uint4*>(stage + (c / CPRO) * SROW + (c % CPRO) * 16) =
                            *reinterpret_cast<const uint4*>(rz + (long long)(c / CPRO) * p.ldr1 + (c % CPRO) * 8);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
#pragma unroll
                for (int oo = 0; oo < HALF; ++oo) {
                    const int o = hf * HALF + oo;
                    const float4 b = *reinterpret_cast<const float4*>(p.b2 + o * 16 + fq * 4);
                    float v[4] = {ca * (acc2[f][o][0] + b.x), ca * (acc2[f][o][1] + b.y), ca * (acc2[f][o][2] + b.z), ca * (acc2[f][o][3] + b.w)};
                    if (p.res1) {
                        const uint2 rr = *reinterpret_cast<const uint2*>(stage + fr * SROW + oo * 32 + fq * 8);
                        v[0] += c1 * bflo(rr.x); v[1] += c1 * bfhi(rr.x); v[2] += c1 * bflo(rr.y); v[3] += c1 * bfhi(rr.y);
                    }
                    if (p.res2) {
                        const uint2 rr = *reinterpret_cast<const uint2*>(p.res2 + m * p.ldr2 + o * 16 + fq * 4);
                        v[0] += c2 * bflo(rr.x); v[1] += c2 * bfhi(rr.x); v[2] += c2 * bflo(rr.y); v[3] += c2 * bfhi(rr.y);
                    }
                    *reinterpret_cast<uint2*>(stage + fr * SROW + oo * 32 + fq * 8) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                    acc2[f][o] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bf16_t* oz = p.out + (mw0 + f * 16) * p.ldo + col0;
#pragma unroll
                for (int c0 = 0; c0 < 16 * CPRO; c0 += 64) {
                    const int c = c0 + lane;
                    *reinterpret_cast<uint4*>(oz + (long long)(c / CPRO) * p.ldo + (c % CPRO) * 8) =
                        *reinterpret_cast<const uint4*>(stage + (c / CPRO) * SROW + (c % CPRO) * 16);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (DBG && blockIdx.x == 0 && (threadIdx.x & 63) < 16)
        for (int k = 0; k < 8; ++k) g_ff_dbg[((threadIdx.x >> 6) * 16 + (threadIdx.x & 63)) * 8 + k] = dbg[((threadIdx.x >> 6) * 16 + (threadIdx.x & 63)) * 8 + k];
}

}  // namespace

extern "C" int v3d_debug_ff_timeline(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ff_dbg), sizeof(g_ff_dbg)) == hipSuccess ? 0 : -1;
}

extern "C" int v3d_ff_fused(const void* x, int64_t ldx, const void* W1p, const float* b1, const void* W2p, const float* b2,
                            const void* res1, int64_t ldr1, const void* res2, int64_t ldr2, const float* coef, int64_t coef_rpg,
                            float c_acc, float c_res1, float c_res2, void* out, int64_t ldo, int64_t M, int32_t C, int32_t hidden,
                            v3d_stream_t stream) {
    V3D_REQUIRE(x && W1p && b1 && W2p && b2 && out, "v3d_ff_fused: null pointer");
    V3D_REQUIRE(C == 320, "v3d_ff_fused: built for C = 320 (got %d); wider levels use the two-GEMM path", C);
    V3D_REQUIRE(hidden > 0 && hidden % 32 == 0 && hidden * (long long)C * 4 < (1ll << 31), "v3d_ff_fused: bad hidden size %d", hidden);
    V3D_REQUIRE(M > 0 && M % 128 == 0, "v3d_ff_fused: M must be a multiple of 128 (got %lld)", (long long)M);
    V3D_REQUIRE(ldx % 8 == 0 && ldo % 8 == 0 && (!res1 || ldr1 % 8 == 0) && (!res2 || ldr2 % 4 == 0), "v3d_ff_fused: row strides");
    V3D_REQUIRE(((((uintptr_t)x | (uintptr_t)W1p | (uintptr_t)W2p | (uintptr_t)out | (uintptr_t)b1 | (uintptr_t)b2) & 15) == 0) &&
                    (!res1 || ((uintptr_t)res1 & 15) == 0) && (!res2 || ((uintptr_t)res2 & 7) == 0),
                "v3d_ff_fused: misaligned pointer");
    V3D_REQUIRE(!coef || coef_rpg > 0, "v3d_ff_fused: coef_rpg must be > 0");
    FFP p;
    p.x = (const bf16_t*)x; p.W1 = (const bf16_t*)W1p; p.W2 = (const bf16_t*)W2p; p.b1 = b1; p.b2 = b2;
    p.res1 = (const bf16_t*)res1; p.res2 = (const bf16_t*)res2; p.coef = coef; p.out = (bf16_t*)out;
    p.M = M; p.ldx = ldx; p.ldr1 = ldr1; p.ldr2 = ldr2; p.ldo = ldo; p.coef_rpg = coef_rpg;
    p.c_acc = c_acc; p.c_res1 = c_res1; p.c_res2 = c_res2; p.hidden = hidden;
    p.w1_bytes = (unsigned)(2ll * hidden * C * 2);
    p.w2_bytes = (unsigned)((long long)C * hidden * 2);
    const long long nblocks = M / 128;
    const int grid = (int)(nblocks < v3d_num_cus() ? nblocks : v3d_num_cus());
    static int dbg = -1;
    if (dbg < 0) {
        const char* e = getenv("V3D_FF_TIMELINE");
        dbg = e ? atoi(e) : 0;
    }
    if (dbg)
        hipLaunchKernelGGL((ff_fused_kernel<320, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((ff_fused_kernel<320>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    return v3d_check_launch("v3d_ff_fused");
}
