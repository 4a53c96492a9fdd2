// v3d_gemm: multi-tap bf16 MFMA contraction for gfx950 (see include/v3d_hip.h for the op contract).
//
// One kernel family covers nn.Linear / 1x1 conv (1 tap), Conv2d 3x3 incl. stride-2 and fused nearest-2x
// upsample (9 taps, implicit GEMM over channels-last pixels) and the (3,1,1) temporal conv (3 taps along the
// frame stride).  Tile: BM x BN x 64, 256 threads = 4 waves (2 x 2), v_mfma_f32_16x16x32_bf16 with the WEIGHT
// fragment as the A operand so that each lane ends up with 4 consecutive output channels of one pixel
// (8-byte bf16 stores, float4 bias / per-image vector loads).  Operands are staged HBM -> VGPR -> LDS with
// the next tile's global loads in flight during the MFMAs of the current one; LDS rows are padded by 16 B.
#include "common.h"

namespace {

constexpr int BK = 64;
constexpr int LROW = BK + 8;  // padded LDS row (bf16 elements) = 144 B, keeps 16-B alignment

struct GP {
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
    int T, tmin, tmax;
    long long S;
    long long sA, sW, sO;
    int mt, nt;  // tile counts
};

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
        iy0 = oy * p.stride - 1;
        ix0 = ox * p.stride - 1;
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
    int t_;
    bool ok;
    __device__ void init(const GP& p, long long m) {
        ok = m < p.M;
        m_ = m;
        long long frame = m / p.S;
        t_ = (int)(frame % p.T);
    }
    __device__ bool tap(const GP& p, int t, long long& s) const {
        int tt = t_ + t - 1;
        s = m_ + (long long)(t - 1) * p.S;
        return ok && tt >= p.tmin && tt <= p.tmax;
    }
};

template <int MODE>
constexpr int ntaps() {
    return MODE == V3D_GEMM_LINEAR ? 1 : (MODE == V3D_GEMM_CONV3X3 ? 9 : 3);
}

template <int BM, int BN, int MODE, bool GEGLU>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GP p) {
    constexpr int AI = BM / 32;      // 16-B chunks of A per thread per stage
    constexpr int BI = BN / 32;
    constexpr int WM = BM / 2;       // wave tile rows (pixels)
    constexpr int WN = BN / 2;       // wave tile cols (out channels)
    constexpr int MF = WM / 16;      // m fragments per wave
    constexpr int NF = WN / 16;      // n fragments per wave
    __shared__ __attribute__((aligned(16))) bf16_t sA[2][BM * LROW];
    __shared__ __attribute__((aligned(16))) bf16_t sB[2][BN * LROW];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap of the 1-D grid: consecutive logical tiles share an XCD (and its L2)
    const int nblk = p.mt * p.nt;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tile_n = bid % p.nt;
    const int tile_m = bid / p.nt;
    const long long m0 = (long long)tile_m * BM;
    const long long n0 = (long long)tile_n * BN;

    const long long z = blockIdx.y;
    // Buffer descriptors: out-of-range offsets (kInvalid) return zeros in hardware, so conv zero padding, the
    // M/N/K tails and the frame-boundary taps need no branches and no select on the loaded data.
    const bufrsrc_t rsA = make_rsrc(p.A + z * p.sA, p.a_bytes);
    const bufrsrc_t rsW = make_rsrc(p.W + z * p.sW, p.w_bytes);

    // global->LDS staging assignment: thread owns 16-B chunk column kc of rows r0 + 32*i
    const int kc = tid & 7;
    const int r0 = tid >> 3;

    RowInfo<MODE> ri[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) ri[i].init(p, m0 + r0 + 32 * i);

    u32x4 ra[AI], rb[BI];
    unsigned aoff[AI], boff[BI];   // byte offsets of this thread's chunks for the current tap (k0 = 0), or kInvalid

    const int ksteps = (int)((p.K + BK - 1) / BK);
    const int nsteps = ksteps * ntaps<MODE>();

    int ld_tap = 0, ld_k0 = 0;  // position of the NEXT tile to load
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
        for (int i = 0; i < AI; ++i)
            *reinterpret_cast<u32x4*>(&sA[buf][(r0 + 32 * i) * LROW + kc * 8]) = ra[i];
#pragma unroll
        for (int i = 0; i < BI; ++i)
            *reinterpret_cast<u32x4*>(&sB[buf][(r0 + 32 * i) * LROW + kc * 8]) = rb[i];
    };

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    gload();
    lstore(0);
    __syncthreads();

    const int frow = lane & 15;        // row inside a 16-row fragment
    const int fk = (lane >> 4) * 8;    // k offset (8 bf16 = 16 B) inside a 32-wide k slice

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

    // ---- epilogue: lane holds, per fragment, 4 consecutive n (= (lane>>4)*4 + r) of pixel m = lane&15 ----
    const long long Nout = GEGLU ? p.N / 2 : p.N;
    const bool vec_ok = (p.ldo % 4 == 0) && (!p.res1 || p.ldr1 % 4 == 0) && (!p.res2 || p.ldr2 % 4 == 0);
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const long long m = m0 + wm * WM + i * 16 + (lane & 15);
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
            const long long np = n0 + wn * WN + j * 16 + (lane >> 4) * 4;  // packed weight-row index
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
            if (np < p.N) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (np + r < p.N) {
                        if (p.bias) v[r] += p.bias[np + r];
                        if (addv) v[r] += addv[np + r];
                    }
                }
            }
            long long col = np;
            if (GEGLU) {
                // j even = value rows, j+1 = the matching gate rows (same lane, same r)
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
            const bool full = vec_ok && (col + 3 < Nout);
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = ca * v[r];
            if (p.res1) {
                if (full) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(p.res1 + m * p.ldr1 + col);
                    o[0] += c1 * bflo(rr.x); o[1] += c1 * bfhi(rr.x); o[2] += c1 * bflo(rr.y); o[3] += c1 * bfhi(rr.y);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < Nout) o[r] += c1 * bf2f(p.res1[m * p.ldr1 + col + r]);
                }
            }
            if (p.res2) {
                if (full) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(p.res2 + m * p.ldr2 + col);
                    o[0] += c2 * bflo(rr.x); o[1] += c2 * bfhi(rr.x); o[2] += c2 * bflo(rr.y); o[3] += c2 * bfhi(rr.y);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < Nout) o[r] += c2 * bf2f(p.res2[m * p.ldr2 + col + r]);
                }
            }
            if (p.out_fp32) {
                float* op = reinterpret_cast<float*>(p.out) + z * p.sO + m * p.ldo + col;
                if (full) {
                    *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < Nout) op[r] = o[r];
                }
            } else {
                bf16_t* op = reinterpret_cast<bf16_t*>(p.out) + z * p.sO + m * p.ldo + col;
                if (full) {
                    *reinterpret_cast<uint2*>(op) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < Nout) op[r] = f2bf(o[r]);
                }
            }
        }
    }
}

template <int BM, int BN, int MODE, bool GEGLU>
int launch(const GP& p0, int batch, hipStream_t st) {
    GP p = p0;
    p.mt = (int)((p.M + BM - 1) / BM);
    p.nt = (int)((p.N + BN - 1) / BN);
    dim3 grid((unsigned)(p.mt * p.nt), (unsigned)batch, 1);
    hipLaunchKernelGGL((gemm_kernel<BM, BN, MODE, GEGLU>), grid, dim3(256), 0, st, p);
    return v3d_check_launch("v3d_gemm");
}

template <int MODE, bool GEGLU>
int dispatch(const GP& p, int batch, hipStream_t st) {
    // N tile: 128 unless a 64-wide tile wastes less (e.g. N = 320: 5 x 64 exact vs 3 x 128 = 17 % padding)
    const long long w128 = ((p.N + 127) / 128) * 128, w64 = ((p.N + 63) / 64) * 64;
    if (w64 < w128) return launch<128, 64, MODE, GEGLU>(p, batch, st);
    return launch<128, 128, MODE, GEGLU>(p, batch, st);
}

}  // namespace

extern "C" int v3d_gemm(const v3d_gemm_args* a, v3d_stream_t stream) {
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
    const long long mtiles = (a->M + 127) / 128, ntiles = (a->N + 63) / 64;
    V3D_REQUIRE(mtiles * ntiles < (1ll << 31), "v3d_gemm: grid too large");
    const int taps = a->mode == V3D_GEMM_LINEAR ? 1 : (a->mode == V3D_GEMM_CONV3X3 ? 9 : 3);
    const unsigned long long a_bytes = (unsigned long long)a->a_rows * a->lda * 2ull;
    const long long ldw = a->ldw ? a->ldw : a->K;
    V3D_REQUIRE(ldw >= a->K && ldw % 8 == 0, "v3d_gemm: ldw must be >= K and a multiple of 8");
    const unsigned long long w_bytes = ((unsigned long long)(taps * a->N - 1) * ldw + a->K) * 2ull;
    V3D_REQUIRE(a_bytes <= kMaxBufBytes && w_bytes <= kMaxBufBytes, "v3d_gemm: operand larger than 4 GiB - 256 B (A %llu B, W %llu B)", a_bytes, w_bytes);
    GP p;
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
    p.T = a->T; p.tmin = a->tmin; p.tmax = a->tmax; p.S = a->S;
    p.sA = a->sA; p.sW = a->sW; p.sO = a->sO;
    p.mt = p.nt = 0;
    hipStream_t st = (hipStream_t)stream;
    switch (a->mode) {
        case V3D_GEMM_LINEAR:
            if (a->geglu) return dispatch<V3D_GEMM_LINEAR, true>(p, a->batch, st);
            return dispatch<V3D_GEMM_LINEAR, false>(p, a->batch, st);
        case V3D_GEMM_CONV3X3:
            V3D_REQUIRE(a->up == 1 || a->up == 2, "v3d_gemm: up must be 1 or 2");
            V3D_REQUIRE(a->stride == 1 || a->stride == 2, "v3d_gemm: stride must be 1 or 2");
            V3D_REQUIRE(a->Hin > 0 && a->Win > 0 && a->Hout > 0 && a->Wout > 0, "v3d_gemm: bad conv geometry");
            V3D_REQUIRE(a->M % ((long long)a->Hout * a->Wout) == 0, "v3d_gemm: M must be n_img*Hout*Wout");
            p.upshift = a->up - 1;
            return dispatch<V3D_GEMM_CONV3X3, false>(p, a->batch, st);
        case V3D_GEMM_CONVT3:
            V3D_REQUIRE(a->T > 0 && a->S > 0, "v3d_gemm: convt3 needs T,S");
            return dispatch<V3D_GEMM_CONVT3, false>(p, a->batch, st);
        default:
            v3d_set_error("v3d_gemm: unknown mode %d", a->mode);
            return V3D_ERR_ARG;
    }
}
