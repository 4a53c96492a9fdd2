// Shared device helpers for the gfx950 kernels (wave64, bf16 storage as uint16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/v3d_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short bf16_t;  // raw storage
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t bufrsrc_t;

// Raw buffer descriptor over [base, base+bytes): loads at byte offsets >= bytes (e.g. kInvalid) return 0.
constexpr unsigned kInvalid = 0xFFFFFF00u;
constexpr unsigned long long kMaxBufBytes = 0xFFFFFF00ull;
__device__ __forceinline__ bufrsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(bufrsrc_t r, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
}



// LDS stores the compiler's wait-count pass cannot see.  With LDS-DMA (buffer_load ... lds) in flight it puts s_waitcnt vmcnt(0) in front of
// every ds_write to the same LDS object (a DMA piece might still be landing there).  The callers write wave-private regions that no DMA ever
// targets.  LDS executes a wave's operations in order, so later ds_reads of the same addresses need no extra wait.
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void lds_store16_nowait(void* p, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory"); }

#define V3D_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
// round-to-nearest-even, NaN kept quiet
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// two floats -> packed bf16 pair, round-to-nearest-even in hardware (one v_cvt_pk_bf16_f32 on gfx950)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// LayerNorm (no affine) of a wave's 32 rows held as MFMA operand fragments, in place: lane (l31 = lane & 31, hi = lane >> 5) holds channels
// 16 k + 8 hi .. + 7 of row l31 in xr[k], its partner lane (lane ^ 32) the rest.  gamma / beta are folded into the consumer's weights / bias at
// pack time (W diag(gamma), W beta), so this is (x - mean) * rstd only.  Sums and sums of squares by v_dot2_f32_bf16 on the packed pairs (1
// instruction per 2 elements, exact bf16 products, fp32 accumulation); variance as E[x^2] - mean^2 in fp32; the result is rounded to bf16 where
// the stand-alone v3d_layernorm stores it.  (Element pairs are picked with shufflevector: the u32x4 bit_cast + subscript form of the first loop
// was miscompiled by ROCm 7.2's clang - every dot2 read dword 0 of the fragment.)
template <int NK>
__device__ __forceinline__ void ln_rows_inplace(bf16x8 (&xr)[NK], float eps) {
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    constexpr int C = NK * 16;
    const bf16x2v ones = __builtin_bit_cast(bf16x2v, 0x3f803f80u);
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const bf16x2v a = __builtin_shufflevector(xr[k], xr[k], 0, 1), b = __builtin_shufflevector(xr[k], xr[k], 2, 3);
        const bf16x2v c = __builtin_shufflevector(xr[k], xr[k], 4, 5), d = __builtin_shufflevector(xr[k], xr[k], 6, 7);
        s0 = __builtin_amdgcn_fdot2_f32_bf16(a, ones, s0, false);
        s1 = __builtin_amdgcn_fdot2_f32_bf16(b, ones, s1, false);
        q0 = __builtin_amdgcn_fdot2_f32_bf16(a, a, q0, false);
        q1 = __builtin_amdgcn_fdot2_f32_bf16(b, b, q1, false);
        s0 = __builtin_amdgcn_fdot2_f32_bf16(c, ones, s0, false);
        s1 = __builtin_amdgcn_fdot2_f32_bf16(d, ones, s1, false);
        q0 = __builtin_amdgcn_fdot2_f32_bf16(c, c, q0, false);
        q1 = __builtin_amdgcn_fdot2_f32_bf16(d, d, q1, false);
    }
    float s = s0 + s1, q = q0 + q1;
    s += __shfl_xor(s, 32, 64);
    q += __shfl_xor(q, 32, 64);
    const float mean = s * (1.0f / C);
    const float var = fmaxf(q * (1.0f / C) - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    const float sh = -mean * rstd;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const u32x4 u = __builtin_bit_cast(u32x4, xr[k]);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2bf(__builtin_fmaf(bflo(u[e]), rstd, sh), __builtin_fmaf(bfhi(u[e]), rstd, sh));
        xr[k] = __builtin_bit_cast(bf16x8, o);
    }
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact-erf GELU (F.gelu default, reference attention.py:97-99) with erf from Abramowitz-Stegun 7.1.26
// (|abs err| <= 1.5e-7, far below bf16 resolution): 1 rcp + 1 exp + 6 fma instead of libm erff's ~40 instructions —
// the GEGLU epilogue of the K=320 feed-forward GEMMs was as long as their main loop.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    float p = 1.061405429f;
    p = p * t + -1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t + -0.284496736f;
    p = p * t + 0.254829592f;
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
// GEGLU epilogue form: erf(z) = 1 - 2^(-z Q(z)) on z = min(|x| / sqrt2, 4) with Q a degree-5 minimax fit of
// -log2(erfc(z)) / z (tools: scipy erfc, weighted Remez iterations; |erf error| <= 3.3e-7, |gelu error| <= 7.4e-7 in fp32
// Horner, i.e. the accuracy of the Abramowitz-Stegun form above) - ONE transcendental (v_exp_f32) and 11 plain VALU per
// value instead of two transcendentals (quarter rate on CDNA4) and 13: the GEGLU epilogue is VALU-bound
// (tools/gemm_floor.py: the K = 320 projection spends 220 of its 410 us there).
__device__ __forceinline__ float gelu_erf_tight(float x) {
    const float z = fminf(fabsf(x) * 0.70710678118654752440f, 4.0f);
    float q = __builtin_fmaf(-1.420422594e-04f, z, 3.664275106e-03f);
    q = __builtin_fmaf(q, z, -3.089617980e-02f);
    q = __builtin_fmaf(q, z, 1.496994254e-01f);
    q = __builtin_fmaf(q, z, 9.181654723e-01f);
    q = __builtin_fmaf(q, z, 1.627925070e+00f);
    const float r = 1.0f - __builtin_amdgcn_exp2f(-(q * z));   // |erf|
    const float h = 0.5f * x;
    return __builtin_fmaf(h, copysignf(r, x), h);               // 0.5 x (1 + erf(x / sqrt2))
}

// value * gelu(gate) for the GEGLU epilogues, in the fewest issue slots the bf16 result allows (the form ff.hip spreads over its MFMA
// slots):  gelu(x) = relu(x) - |x| 2^-(z Q(z) + 1),  z = min(|x| / sqrt2, 4),  0.5 erfc(z) = 2^-(z Q(z) + 1),  Q = degree-3 fit of
// -log2(erfc(z)) / z weighted for the GELU error: |gelu error| <= 8.6e-6 (fp32 Horner) against 4e-3 relative of the bf16 rounding that
// follows.  10 VALU + 1 v_exp_f32 including the product with `value` (gelu_erf_tight above: 14 + 1).
__device__ __forceinline__ float geglu_mul(float value, float x) {
    const float z = fminf(fabsf(x) * 0.70710678118654752440f, 4.0f);
    float q = __builtin_fmaf(-1.664338751e-02f, z, 1.293501013e-01f);
    q = __builtin_fmaf(q, z, 9.298687989e-01f);
    q = __builtin_fmaf(q, z, 1.625731271e+00f);
    q = __builtin_fmaf(q, z, 1.0f);
    const float e = __builtin_amdgcn_exp2f(-q);
    return value * (fmaxf(x, 0.0f) - fabsf(x) * e);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// host-side error plumbing (defined in capi.hip)
void v3d_set_error(const char* fmt, ...);
int v3d_check_launch(const char* what);
int v3d_num_cus();   // compute units of the current device (cached)

#define V3D_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            v3d_set_error(__VA_ARGS__);   \
            return V3D_ERR_ARG;           \
        }                                 \
    } while (0)
