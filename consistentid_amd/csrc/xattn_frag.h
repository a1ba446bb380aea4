// Register-level helpers shared by the fused identity cross-attention kernels (xattn3.hip; first written for its deleted predecessor):
// v_mfma_f32_16x16x32_f16 fragments (lane l: l16 = l & 15, lq = l >> 4)
//   A operand a[j] = A[l16][8 lq + j],  B operand b[j] = B[8 lq + j][l16],  C / D c[i] = C[4 lq + i][l16]
// so an accumulator quad (4 consecutive rows of one column) is a legal half of a B operand's k-slots.
#pragma once
#include "common.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));
CID_DEVINL f32x4v mfma16(half8 a, half8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

CID_DEVINL half8 cat4(half4 a, half4 b) {
    half8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}
CID_DEVINL half4 cvt4(f32x4v v) {
    half4 r;
    r[0] = (half_t)v[0]; r[1] = (half_t)v[1]; r[2] = (half_t)v[2]; r[3] = (half_t)v[3];
    return r;
}

// materialise a value HERE: without it hipcc sinks the fp32 -> fp16 conversions to their far-away uses and keeps
// (spills) the twice as large fp32 accumulators instead
CID_DEVINL void pin(half4& v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 u = __builtin_bit_cast(u32x2, v);
    asm volatile("" : "+v"(u));
    v = __builtin_bit_cast(half4, u);
}

// reductions over the four 16-lane rows of a wave (the token is lane & 15, the rows hold different channels / keys)
CID_DEVINL float rows_max(float v) {
    unsigned u = __float_as_uint(v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    u = __float_as_uint(m);
    auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
CID_DEVINL float rows_sum(float v) {
    unsigned u = __float_as_uint(v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    float m = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    u = __float_as_uint(m);
    auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// key class of score register (kt, i) for a context layout known at compile time (NT text + NI ID keys; the
// reference's is 77 + 4, attention.py:241 with num_tokens = 4); key = 16 kt + 4 lq + i, lq = 0..3:
// 0 = text in every lane row, 1 = absent in every lane row, 2 = depends on the lane row (or layout only known at run time)
constexpr int key_class(int NT, int NI, int kt, int i) {
    if (NT == 0) return 2;
    const int lo = 16 * kt + i, hi = lo + 12;
    return hi < NT ? 0 : (lo >= NT + NI ? 1 : 2);
}
