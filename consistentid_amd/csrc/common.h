// Shared device helpers for the ConsistentID MI355X (gfx950 / CDNA4) kernels.
// wave = 64 lanes; MFMA shape used everywhere: v_mfma_f32_32x32x16_f16.
//
// Fragment conventions (lane l, idx = l & 31, hi = l >> 5):
//   A operand: row idx of the A tile, 8 consecutive k at k-slot (hi, 0..7)
//   B operand: col idx of the B tile, 8 consecutive k at k-slot (hi, 0..7)
//   C/D      : col = idx, row(r) = (r & 3) + 8 * (r >> 2) + 4 * hi,  r in [0,16)
// Only the C/D map and "A rows / B cols = lane & 31" are relied upon; the k-slot
// numbering is the same function for A and B, so it cancels in every contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CID_DEVINL __device__ __forceinline__

CID_DEVINL f32x16 mfma32(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

CID_DEVINL int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

CID_DEVINL half8 ld_global_h8(const half_t* p) {
    return *reinterpret_cast<const half8*>(p);
}
CID_DEVINL half8 zero_h8() {
    half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return z;
}
CID_DEVINL f32x16 zero_f16v() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

CID_DEVINL float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
CID_DEVINL float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// x * sigmoid(x) with the hardware exp2 / reciprocal (1 ulp each; the result is rounded to fp16 by every caller)
CID_DEVINL float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }
// erf GELU (diffusers GEGLU uses F.gelu without the tanh approximation): gelu(g) = g Phi(g) = max(g, 0) - |g| / 2 * erfc(|g| / sqrt 2),
// with erfc(t / sqrt 2) = 2^P(t) for t >= 0, P an odd-free degree-6 minimax fit without constant term (tools/fit_gelu.py:
// |error of erf| <= 2.4e-7, of gelu <= 5.8e-7 in fp32 -- the class of the Abramowitz & Stegun 7.1.26 form used until round 5,
// 4.7e-7, and three orders below the fp16 rounding of the product).  Seven FMA-class operations and ONE transcendental per element
// instead of sixteen and two: the GEGLU epilogue is VALU-bound at K = 320, no cancellation on either side (the
// negative tail is -|g| / 2 * 2^P exactly).
#ifdef CID_GELU_AS7126      // experiment builds only (build.py --variant ...): the round-5 form, for same-call A/B timing
CID_DEVINL float gelu_erf_f(float g) {
    const float x = g * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, __builtin_fabsf(x), 1.f));
    float p = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    p = __builtin_fmaf(t, p, 1.421413741f);
    p = __builtin_fmaf(t, p, -0.284496736f);
    p = __builtin_fmaf(t, p, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(x * (x * -1.4426950408889634f));
    const float y = __builtin_fmaf(-p, e, 1.f);
    const float h = 0.5f * g;
    return __builtin_fmaf(h, __builtin_copysignf(y, x), h);
}
#else
CID_DEVINL float gelu_erf_f(float g) {
    const float t = __builtin_fabsf(g);
    float q = __builtin_fmaf(t, 1.775648707e-05f, -6.477678544e-04f);
    q = __builtin_fmaf(t, q, 7.724069990e-03f);
    q = __builtin_fmaf(t, q, -5.292676762e-02f);
    q = __builtin_fmaf(t, q, -4.590827227e-01f);
    q = __builtin_fmaf(t, q, -1.151116848e+00f);
    const float e = __builtin_amdgcn_exp2f(t * q);          // erfc(|g| / sqrt 2)
    return __builtin_fmaf(-0.5f * t, e, __builtin_fmaxf(g, 0.f));
}
#endif

// ---------------------------------------------------------------- host side
#include <stdio.h>
#include <string.h>
void cid_set_error(const char* fmt, ...);
#define CID_CHECK_ARG(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            cid_set_error(__VA_ARGS__);   \
            return -22;                   \
        }                                 \
    } while (0)
#define CID_CHECK_LAUNCH(name)                                              \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            cid_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return -5;                                                      \
        }                                                                   \
    } while (0)
