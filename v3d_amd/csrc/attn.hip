// Attention kernels for gfx950: spatial (streamed-softmax MFMA, d_head 64) and temporal (T <= 32 frames).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------------
// Spatial self-attention.  Block = 4 waves = 128 queries of one (image, head); each wave owns 32 queries.
// Per 32-key sub-tile a wave computes S^T = K . Q^T with v_mfma_f32_32x32x16_bf16 (A = K rows from LDS,
// B = Q held in registers), so each lane owns one query column and 16 of its 32 scores: the row max / sum need
// one cross-lane exchange (lane <-> lane+32) and P^T is already in B-operand position for O^T += V^T . P^T.
// The contraction index of that second MFMA is a permutation of the keys that is applied identically to the
// A operand (V^T read from LDS as two 8-byte pieces), so no transpose or lane shuffle of P is needed.
// V arrives pre-transposed ([C][S], produced directly by the value projection GEMM).
// ------------------------------------------------------------------------------------------------------
constexpr int AKT = 64;          // keys per LDS tile
constexpr int AROW = 64 + 8;     // padded LDS row (bf16)

__global__ __launch_bounds__(256, 2) void attn_spatial_kernel(const bf16_t* __restrict__ q, long long ldq,
                                                              const bf16_t* __restrict__ k, long long ldk,
                                                              const bf16_t* __restrict__ vT, bf16_t* __restrict__ out,
                                                              long long ldo, long long S, int heads, float scale2) {
    __shared__ __attribute__((aligned(16))) bf16_t sK[2][AKT * AROW];
    __shared__ __attribute__((aligned(16))) bf16_t sV[2][64 * AROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = lane & 31, hi = lane >> 5;
    const long long n = blockIdx.z;
    const int h = blockIdx.y;
    const long long C = (long long)heads * 64;
    const long long qrow = (long long)blockIdx.x * 128 + wave * 32 + qi;
    const bool q_ok = qrow < S;

    // per-(image, head) buffer descriptors: rows / keys past S read as zeros in hardware
    const bufrsrc_t rsQ = make_rsrc(q + n * S * ldq + h * 64, (unsigned)(((S - 1) * ldq + 64) * 2));
    const bufrsrc_t rsK = make_rsrc(k + n * S * ldk + h * 64, (unsigned)(((S - 1) * ldk + 64) * 2));
    const bufrsrc_t rsV = make_rsrc(vT + (n * C + h * 64) * S, (unsigned)(64 * S * 2));

    bf16x8 qf[4];
    {
        const unsigned qoff = q_ok ? (unsigned)((qrow * ldq + hi * 8) * 2) : kInvalid;
#pragma unroll
        for (int st = 0; st < 4; ++st)
            qf[st] = __builtin_bit_cast(bf16x8, buf_load16(rsQ, q_ok ? qoff + st * 32 : kInvalid));
    }

    // staging: thread owns chunks c = tid + 256*i, row = c>>3, kc = c&7
    const int srow0 = tid >> 3, skc = tid & 7;
    u32x4 rk[2], rv[2];
    const int ntiles = (int)((S + AKT - 1) / AKT);
    auto gload = [&](int t) {
        const long long k0 = (long long)t * AKT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = srow0 + 32 * i;
            rk[i] = buf_load16(rsK, (k0 + row < S) ? (unsigned)(((k0 + row) * ldk + skc * 8) * 2) : kInvalid);
            rv[i] = buf_load16(rsV, (k0 + skc * 8 < S) ? (unsigned)(((long long)row * S + k0 + skc * 8) * 2) : kInvalid);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = srow0 + 32 * i;
            *reinterpret_cast<u32x4*>(&sK[buf][row * AROW + skc * 8]) = rk[i];
            *reinterpret_cast<u32x4*>(&sV[buf][row * AROW + skc * 8]) = rv[i];
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    gload(0);
    lstore(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        const long long k0 = (long long)t * AKT;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x16 sT;
#pragma unroll
            for (int r = 0; r < 16; ++r) sT[r] = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&sK[buf][(sub * 32 + qi) * AROW + st * 16 + hi * 8]);
                sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], sT, 0, 0, 0);
            }
            // sT[r] = score(key = sub*32 + (r&3) + 8*(r>>2) + 4*hi, query = qi)
            float mx = -INFINITY;
            const bool tail = (k0 + AKT > S);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s = sT[r] * scale2;
                if (tail) {
                    const long long key = k0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= S) s = -INFINITY;
                }
                sT[r] = s;
                mx = fmaxf(mx, s);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            // m_new == -inf only if every key so far is masked (cannot happen for sub-tile 0 of tile 0)
            const float alpha = (m_new == -INFINITY) ? 1.f : exp2f(m_run - m_new);
            float psum = 0.f;
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = (m_new == -INFINITY) ? 0.f : exp2f(sT[r] - m_new);
                psum += p[r];
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[0][r] *= alpha;
                o[1][r] *= alpha;
            }
            bf16x8 pf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 u = make_uint4(pack2bf(p[ks * 8 + 0], p[ks * 8 + 1]), pack2bf(p[ks * 8 + 2], p[ks * 8 + 3]),
                                     pack2bf(p[ks * 8 + 4], p[ks * 8 + 5]), pack2bf(p[ks * 8 + 6], p[ks * 8 + 7]));
                pf[ks] = __builtin_bit_cast(bf16x8, u);
            }
#pragma unroll
            for (int db = 0; db < 2; ++db) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16_t* vp = &sV[buf][(db * 32 + qi) * AROW + sub * 32 + ks * 16 + hi * 4];
                    const uint2 lo = *reinterpret_cast<const uint2*>(vp);
                    const uint2 hi2 = *reinterpret_cast<const uint2*>(vp + 8);
                    const bf16x8 vf = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi2.x, hi2.y));
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[ks], o[db], 0, 0, 0);
                }
            }
        }
        if (t + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_ok) {
        bf16_t* op = out + (n * S + qrow) * ldo + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + 8 * g + 4 * hi;
                const uint2 w = make_uint2(pack2bf(o[db][g * 4 + 0] * inv, o[db][g * 4 + 1] * inv),
                                           pack2bf(o[db][g * 4 + 2] * inv, o[db][g * 4 + 3] * inv));
                *reinterpret_cast<uint2*>(op + d0) = w;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Temporal self-attention (frame axis).  A "problem" is one (sample b, position s, head h): Tq x Tk scores,
// d_head 64.  G = 64 / Tq problems share a wave; lane (slot, i) owns query frame i of its slot's problem, keeps
// q and the output row in registers and walks the Tk keys/values staged in LDS (all lanes of a slot read the same
// LDS address -> broadcast).  Loads follow the frame stride of the channels-last layout directly, so the
// "(b t) s c -> (b s) t c" transposes of the reference never happen.
// ------------------------------------------------------------------------------------------------------
constexpr int TMAX = 32;

struct TP {
    const bf16_t* q; long long q_sb, q_st, q_ss;
    const bf16_t* k; const bf16_t* v; long long kv_sb, kv_st, kv_ss;
    bf16_t* out; long long o_sb, o_st, o_ss;
    long long P, S; int heads, Tq, Tk, G; float scale;
};

__global__ __launch_bounds__(256) void attn_temporal_kernel(TP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_wave = p.G * p.Tk * 64;  // bf16 elements of K (and of V) per wave
    bf16_t* sK = reinterpret_cast<bf16_t*>(smem_raw) + (size_t)wave * 2 * per_wave;
    bf16_t* sV = sK + per_wave;

    const long long wave_id = (long long)blockIdx.x * 4 + wave;
    const long long p0 = wave_id * p.G;

    // stage K and V rows: chunk c -> (row = c>>3, kc = c&7), row -> (slot, j)
    const int nchunks = p.G * p.Tk * 8;
    for (int c = lane; c < nchunks; c += 64) {
        const int row = c >> 3, kc = c & 7;
        const int slot = row / p.Tk, j = row - slot * p.Tk;
        const long long pp = p0 + slot;
        uint4 uk = make_uint4(0, 0, 0, 0), uv = uk;
        if (pp < p.P) {
            const int h = (int)(pp % p.heads);
            const long long bs = pp / p.heads;
            const long long s = bs % p.S, b = bs / p.S;
            const long long off = b * p.kv_sb + (long long)j * p.kv_st + s * p.kv_ss + h * 64 + kc * 8;
            uk = *reinterpret_cast<const uint4*>(p.k + off);
            uv = *reinterpret_cast<const uint4*>(p.v + off);
        }
        *reinterpret_cast<uint4*>(sK + row * 64 + kc * 8) = uk;
        *reinterpret_cast<uint4*>(sV + row * 64 + kc * 8) = uv;
    }
    __syncthreads();

    const int slot = lane / p.Tq, i = lane - slot * p.Tq;
    const long long pp = p0 + slot;
    if (slot >= p.G || pp >= p.P) return;
    const int h = (int)(pp % p.heads);
    const long long bs = pp / p.heads;
    const long long s = bs % p.S, b = bs / p.S;

    float qv[64];
    {
        const bf16_t* qp = p.q + b * p.q_sb + (long long)i * p.q_st + s * p.q_ss + h * 64;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint4 u = *reinterpret_cast<const uint4*>(qp + c * 8);
            qv[c * 8 + 0] = bflo(u.x) * p.scale; qv[c * 8 + 1] = bfhi(u.x) * p.scale;
            qv[c * 8 + 2] = bflo(u.y) * p.scale; qv[c * 8 + 3] = bfhi(u.y) * p.scale;
            qv[c * 8 + 4] = bflo(u.z) * p.scale; qv[c * 8 + 5] = bfhi(u.z) * p.scale;
            qv[c * 8 + 6] = bflo(u.w) * p.scale; qv[c * 8 + 7] = bfhi(u.w) * p.scale;
        }
    }
    const bf16_t* kk = sK + slot * p.Tk * 64;
    const bf16_t* vv = sV + slot * p.Tk * 64;
    float sc[TMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
        sc[j] = -INFINITY;
        if (j < p.Tk) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint4 u = *reinterpret_cast<const uint4*>(kk + j * 64 + c * 8);
                acc += qv[c * 8 + 0] * bflo(u.x) + qv[c * 8 + 1] * bfhi(u.x) + qv[c * 8 + 2] * bflo(u.y) + qv[c * 8 + 3] * bfhi(u.y) +
                       qv[c * 8 + 4] * bflo(u.z) + qv[c * 8 + 5] * bfhi(u.z) + qv[c * 8 + 6] * bflo(u.w) + qv[c * 8 + 7] * bfhi(u.w);
            }
            sc[j] = acc;
            mx = fmaxf(mx, acc);
        }
    }
    float ov[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) ov[d] = 0.f;
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
        if (j < p.Tk) {
            const float pj = __expf(sc[j] - mx);
            l += pj;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint4 u = *reinterpret_cast<const uint4*>(vv + j * 64 + c * 8);
                ov[c * 8 + 0] += pj * bflo(u.x); ov[c * 8 + 1] += pj * bfhi(u.x);
                ov[c * 8 + 2] += pj * bflo(u.y); ov[c * 8 + 3] += pj * bfhi(u.y);
                ov[c * 8 + 4] += pj * bflo(u.z); ov[c * 8 + 5] += pj * bfhi(u.z);
                ov[c * 8 + 6] += pj * bflo(u.w); ov[c * 8 + 7] += pj * bfhi(u.w);
            }
        }
    }
    const float inv = 1.0f / l;
    bf16_t* op = p.out + b * p.o_sb + (long long)i * p.o_st + s * p.o_ss + h * 64;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint4 w = make_uint4(pack2bf(ov[c * 8 + 0] * inv, ov[c * 8 + 1] * inv), pack2bf(ov[c * 8 + 2] * inv, ov[c * 8 + 3] * inv),
                                   pack2bf(ov[c * 8 + 4] * inv, ov[c * 8 + 5] * inv), pack2bf(ov[c * 8 + 6] * inv, ov[c * 8 + 7] * inv));
        *reinterpret_cast<uint4*>(op + c * 8) = w;
    }
}

}  // namespace

extern "C" int v3d_attn_spatial(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vT, void* out,
                                int64_t ldo, int64_t n_img, int64_t S, int32_t heads, float scale, v3d_stream_t stream) {
    V3D_REQUIRE(q && k && vT && out, "v3d_attn_spatial: null pointer");
    V3D_REQUIRE(n_img > 0 && n_img <= 65535 && heads > 0 && heads <= 65535 && S > 0, "v3d_attn_spatial: bad sizes");
    V3D_REQUIRE(S % 8 == 0, "v3d_attn_spatial: S must be a multiple of 8 (got %lld)", (long long)S);
    V3D_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 4 == 0, "v3d_attn_spatial: ldq/ldk must be multiples of 8, ldo of 4");
    V3D_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vT) & 15) == 0 && ((uintptr_t)out & 7) == 0, "v3d_attn_spatial: misaligned pointer");
    V3D_REQUIRE((unsigned long long)S * (ldq > ldk ? ldq : ldk) * 2ull <= kMaxBufBytes, "v3d_attn_spatial: per-image q/k slab exceeds 4 GiB");
    dim3 grid((unsigned)((S + 127) / 128), (unsigned)heads, (unsigned)n_img);
    hipLaunchKernelGGL(attn_spatial_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, (long long)ldq,
                       (const bf16_t*)k, (long long)ldk, (const bf16_t*)vT, (bf16_t*)out, (long long)ldo, (long long)S, heads,
                       scale * 1.44269504088896340736f);
    return v3d_check_launch("v3d_attn_spatial");
}

extern "C" int v3d_attn_temporal(const void* q, int64_t q_sb, int64_t q_st, int64_t q_ss,
                                 const void* k, const void* v, int64_t kv_sb, int64_t kv_st, int64_t kv_ss,
                                 void* out, int64_t o_sb, int64_t o_st, int64_t o_ss,
                                 int64_t B, int32_t Tq, int32_t Tk, int64_t S, int32_t heads, float scale,
                                 v3d_stream_t stream) {
    V3D_REQUIRE(q && k && v && out, "v3d_attn_temporal: null pointer");
    V3D_REQUIRE(Tq >= 1 && Tq <= TMAX && Tk >= 1 && Tk <= TMAX, "v3d_attn_temporal: Tq/Tk must be in [1,%d] (got %d,%d)", TMAX, Tq, Tk);
    V3D_REQUIRE(B > 0 && S > 0 && heads > 0, "v3d_attn_temporal: bad sizes");
    V3D_REQUIRE((q_sb | q_st | q_ss | kv_sb | kv_st | kv_ss | o_sb | o_st | o_ss) % 8 == 0, "v3d_attn_temporal: strides must be multiples of 8 elements");
    V3D_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0, "v3d_attn_temporal: misaligned pointer");
    TP p;
    p.q = (const bf16_t*)q; p.q_sb = q_sb; p.q_st = q_st; p.q_ss = q_ss;
    p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.kv_sb = kv_sb; p.kv_st = kv_st; p.kv_ss = kv_ss;
    p.out = (bf16_t*)out; p.o_sb = o_sb; p.o_st = o_st; p.o_ss = o_ss;
    p.P = (long long)B * S * heads; p.S = S; p.heads = heads; p.Tq = Tq; p.Tk = Tk; p.scale = scale;
    {   // problems per wave: as many Tq-lane slots as fit in a wave, capped so K+V staging stays <= 16 KiB per wave
        int g = 64 / Tq;
        const int cap = 16384 / (Tk * 256);
        if (g > cap) g = cap;
        if (g < 1) g = 1;
        p.G = g;
    }
    const long long nwaves = (p.P + p.G - 1) / p.G;
    const long long blocks = (nwaves + 3) / 4;
    V3D_REQUIRE(blocks < (1ll << 31), "v3d_attn_temporal: grid too large");
    const size_t shmem = (size_t)4 * 2 * p.G * Tk * 64 * sizeof(bf16_t);
    V3D_REQUIRE(shmem <= 64 * 1024, "v3d_attn_temporal: LDS request %zu too large", shmem);
    hipLaunchKernelGGL(attn_temporal_kernel, dim3((unsigned)blocks), dim3(256), shmem, (hipStream_t)stream, p);
    return v3d_check_launch("v3d_attn_temporal");
}
