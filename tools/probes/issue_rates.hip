// Issue / throughput cost of the instructions the softmax-bound attention kernels are made of, on ONE SIMD:
// cycles per instruction for long runs of independent instructions (s_memtime around an unrolled asm block), with
// W waves resident on the SIMD (W = 1, 2, 3), and the cost of k VALU / transcendental fillers between two MFMAs.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/issue_rates.hip -o tools/probes/issue_rates ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

enum Mode { M_FMA, M_EXP, M_PKFMA32, M_PKFMA16, M_CVTPK, M_MAX3, M_EXP16, M_MFMA16K32, M_MFMA16K16, M_MFMA32K16, M_MFMA32K8,
            M_MF32_E0, M_MF32_E2, M_MF32_E4, M_MF32_E6, M_MF32_E8, M_MF32_V4, M_MF32_V8, M_MF16_E2, M_MF16_E4, M_MF16_V2, M_MF16_V4,
            M_MF16_CVT2, M_MF16_CVT4, M_MF16_MAX2, M_MF16_MAX4, M_MF16_PKADD2, M_MF16_PKMUL2, M_MF16_PKFMA16_2, M_MF16_FMA2X2,
            M_HET_EXP, M_HET_CVT, M_HET_FMA, M_COUNT };
static const char* names[] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "v_pk_fma_f16", "v_cvt_pk_f16_f32", "v_max3_f32", "v_exp_f16",
                              "mfma 16x16x32 f16", "mfma 16x16x16 f16 (legacy)", "mfma 32x32x16 f16", "mfma 32x32x8 f16 (legacy)",
                              "mfma32 + 0 exp", "mfma32 + 2 exp", "mfma32 + 4 exp", "mfma32 + 6 exp", "mfma32 + 8 exp",
                              "mfma32 + 4 fma", "mfma32 + 8 fma", "mfma16 + 2 exp", "mfma16 + 4 exp", "mfma16 + 2 fma", "mfma16 + 4 fma",
                              "mfma16 + 2 cvt_pk_f16_f32", "mfma16 + 4 cvt_pk_f16_f32", "mfma16 + 2 max3_f32", "mfma16 + 4 max3_f32",
                              "mfma16 + 2 pk_add_f32", "mfma16 + 2 pk_mul_f32", "mfma16 + 2 pk_fma_f16", "mfma16 + 4 fma (= 2 pk unpacked)",
                              "het: mfma16 wave | exp wave", "het: mfma16 wave | cvt_pk wave", "het: mfma16 wave | fma wave"};

template <int MODE>
__global__ void __launch_bounds__(1024) probe(unsigned long long* out, int iters, float seed, int mfma_mult = 1) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + i * 0.001f + threadIdx.x * 1e-6f;
    half8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
    half4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d0, d1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { d0[i] = 0.f; d1[i] = 0.f; }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    // (heterogeneous modes with several VALU waves per SIMD: the MFMA waves run mfma_mult times as long, so that the VALU
    //  waves are measured with the matrix pipe busy from start to end)
    if (MODE >= M_HET_EXP && MODE <= M_HET_FMA && threadIdx.x < 256) iters *= mfma_mult;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == M_FMA) {
            asm volatile(REP16("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n")
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
        } else if constexpr (MODE == M_EXP) {
            asm volatile(REP16("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
        } else if constexpr (MODE == M_EXP16) {
            asm volatile(REP16("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n")
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
        } else if constexpr (MODE == M_PKFMA32) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {v[0], v[1]}, p1 = {v[2], v[3]}, p2 = {v[4], v[5]}, p3 = {v[6], v[7]};
            asm volatile(REP16("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            v[0] = p0[0] + p1[0] + p2[0] + p3[0];
        } else if constexpr (MODE == M_PKFMA16) {
            asm volatile(REP16("v_pk_fma_f16 %0, %0, %0, %0\n v_pk_fma_f16 %1, %1, %1, %1\n v_pk_fma_f16 %2, %2, %2, %2\n v_pk_fma_f16 %3, %3, %3, %3\n")
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
        } else if constexpr (MODE == M_CVTPK) {
            asm volatile(REP16("v_cvt_pk_f16_f32 %0, %4, %5\n v_cvt_pk_f16_f32 %1, %4, %5\n v_cvt_pk_f16_f32 %2, %4, %5\n v_cvt_pk_f16_f32 %3, %4, %5\n")
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(v[4]), "v"(v[5]));
        } else if constexpr (MODE == M_MAX3) {
            asm volatile(REP16("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5\n")
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(v[4]), "v"(v[5]));
        } else if constexpr (MODE == M_MFMA16K32) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
            }
        } else if constexpr (MODE == M_MFMA16K16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c3, 0, 0, 0);
            }
        } else if constexpr (MODE == M_MFMA32K16) {
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d1, 0, 0, 0);
            }
        } else if constexpr (MODE == M_MFMA32K8) {
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                d0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, d1, 0, 0, 0);
            }
        } else if constexpr (MODE >= M_MF32_E0 && MODE <= M_MF32_V8) {
            // 64 x [one 32x32x16 MFMA (two accumulators alternating) + k independent fillers]
            constexpr int K = MODE == M_MF32_E0 ? 0 : MODE == M_MF32_E2 ? 2 : MODE == M_MF32_E4 ? 4 : MODE == M_MF32_E6 ? 6 : MODE == M_MF32_E8 ? 8
                              : MODE == M_MF32_V4 ? 4 : 8;
            constexpr bool EXP = MODE <= M_MF32_E8;
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d0, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k]));
                    else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[k]));
                }
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d1, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[8 + k]));
                    else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[8 + k]));
                }
            }
        } else if constexpr (MODE >= M_MF16_CVT2 && MODE <= M_MF16_FMA2X2) {
            // 64 x [one 16x16x32 MFMA + k independent non-FMA fillers]: the classes the attention kernels' VALU is made of
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = f2{v[2 * i], v[2 * i + 1]};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#define FILL(OFF)                                                                                                              \
    if constexpr (MODE == M_MF16_CVT2 || MODE == M_MF16_CVT4) {                                                                 \
        _Pragma("unroll") for (int k = 0; k < (MODE == M_MF16_CVT2 ? 2 : 4); ++k)                                               \
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "+v"(v[(OFF + k) & 15]) : "v"(v[14]), "v"(v[15]));                   \
    } else if constexpr (MODE == M_MF16_MAX2 || MODE == M_MF16_MAX4) {                                                          \
        _Pragma("unroll") for (int k = 0; k < (MODE == M_MF16_MAX2 ? 2 : 4); ++k)                                               \
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[(OFF + k) & 15]) : "v"(v[14]), "v"(v[15]));                     \
    } else if constexpr (MODE == M_MF16_PKADD2) {                                                                               \
        _Pragma("unroll") for (int k = 0; k < 2; ++k) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[(OFF / 2 + k) & 7]));   \
    } else if constexpr (MODE == M_MF16_PKMUL2) {                                                                               \
        _Pragma("unroll") for (int k = 0; k < 2; ++k) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[(OFF / 2 + k) & 7]));   \
    } else if constexpr (MODE == M_MF16_PKFMA16_2) {                                                                            \
        _Pragma("unroll") for (int k = 0; k < 2; ++k) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(v[(OFF + k) & 15]));  \
    } else {                                                                                                                    \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[(OFF + k) & 15]));     \
    }
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0); FILL(0)
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0); FILL(4)
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0); FILL(8)
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0); FILL(10)
#undef FILL
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[0] += p[i][0] + p[i][1];
        } else if constexpr (MODE >= M_HET_EXP && MODE <= M_HET_FMA) {
            // heterogeneous pair on one SIMD: waves 0-3 issue only MFMAs, waves 4-7 (same SIMDs) only VALU of one class
            if (threadIdx.x < 256) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
                }
            } else if constexpr (MODE == M_HET_EXP) {
                asm volatile(REP16("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
            } else if constexpr (MODE == M_HET_CVT) {
                asm volatile(REP16("v_cvt_pk_f16_f32 %0, %4, %5\n v_cvt_pk_f16_f32 %1, %4, %5\n v_cvt_pk_f16_f32 %2, %4, %5\n v_cvt_pk_f16_f32 %3, %4, %5\n")
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(v[4]), "v"(v[5]));
            } else {
                asm volatile(REP16("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n")
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
            }
        } else {
            constexpr int K = (MODE == M_MF16_E2 || MODE == M_MF16_V2) ? 2 : 4;
            constexpr bool EXP = MODE == M_MF16_E2 || MODE == M_MF16_E4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#define ONE(C, OFF) C = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, C, 0, 0, 0);                                      \
    _Pragma("unroll") for (int k = 0; k < K; ++k) {                                                                        \
        if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[OFF + k]));                                                      \
        else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[OFF + k]));                                                  \
    }
                ONE(c0, 0) ONE(c1, 4) ONE(c2, 8) ONE(c3, 12)
#undef ONE
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + d0[i] + d1[i];
    s += c0[0] + c1[1] + c2[2] + c3[3];
    if (s == 12345.678f) out[1023] = 1;           // keep everything alive
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(unsigned long long* d, int waves_per_simd) {
    // one workgroup of 4 * W waves on one CU: W waves on every SIMD
    const int iters = 20;
    const int per_iter = 64;
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(256 * waves_per_simd), 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < 4 * waves_per_simd; ++w) mx = h[w] > mx ? h[w] : mx;
    // cycles for ALL the SIMD's waves to finish their iters * per_iter instructions (or MFMA+filler groups)
    printf("  %-28s W=%d  %8.1f cycles per instruction(group) per wave, %8.1f per SIMD-issue\n", names[MODE], waves_per_simd,
           (double)mx / (iters * per_iter), (double)mx / (iters * per_iter * waves_per_simd));
}

template <int MODE>
void run_all(unsigned long long* d) {
    run<MODE>(d, 1);
    run<MODE>(d, 2);
    run<MODE>(d, 3);
}

template <int MODE>
void run_het(unsigned long long* d) {
    // W = 2: waves 0-3 (one per SIMD) MFMA only, waves 4-7 (their SIMD partners) VALU only; both halves' times for 64 x 20 instructions
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(512), 0, 0, d, 20, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long m0 = 0, m1 = 0;
    for (int w = 0; w < 4; ++w) { m0 = h[w] > m0 ? h[w] : m0; m1 = h[4 + w] > m1 ? h[4 + w] : m1; }
    printf("  %-32s MFMA wave %6.1f cycles per MFMA, VALU wave %6.1f cycles per instruction (alone: 16.8 / see above)\n", names[MODE],
           (double)m0 / (20 * 64), (double)m1 / (20 * 64));
}

template <int MODE>
void run_het_n(unsigned long long* d, int vwaves) {
    // one MFMA-only wave + vwaves VALU-only waves on every SIMD (waves i, i + 4, i + 8, ... of a workgroup share SIMD i & 3)
    const int mult = 3;
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(256 * (1 + vwaves)), 0, 0, d, 20, 1.0f, mult);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long m0 = 0, m1 = 0;
    for (int w = 0; w < 4; ++w) m0 = h[w] > m0 ? h[w] : m0;
    for (int w = 4; w < 4 * (1 + vwaves); ++w) m1 = h[w] > m1 ? h[w] : m1;
    printf("  %-32s + %d VALU waves: MFMA wave %6.1f cycles per MFMA, each VALU wave %6.1f cycles per instruction = %5.1f per SIMD-issue%s\n",
           names[MODE], vwaves, (double)m0 / (20 * 64 * mult), (double)m1 / (20 * 64), (double)m1 / (20 * 64 * vwaves),
           m1 > m0 ? "  (VALU waves outlasted the MFMA wave!)" : "");
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 1024 * 8);
    printf("cycles = s_memtime / readcyclecounter ticks (shader clock); 64-instruction blocks x 20 iterations; W = waves per SIMD (one workgroup of 4 W waves)\n");
    run_all<M_FMA>(d); run_all<M_EXP>(d); run_all<M_EXP16>(d); run_all<M_PKFMA32>(d); run_all<M_PKFMA16>(d); run_all<M_CVTPK>(d); run_all<M_MAX3>(d);
    run_all<M_MFMA16K32>(d); run_all<M_MFMA16K16>(d); run_all<M_MFMA32K16>(d); run_all<M_MFMA32K8>(d);
    printf("one 32x32x16 MFMA + k independent fillers (group = MFMA + fillers):\n");
    run_all<M_MF32_E0>(d); run_all<M_MF32_E2>(d); run_all<M_MF32_E4>(d); run_all<M_MF32_E6>(d); run_all<M_MF32_E8>(d);
    run_all<M_MF32_V4>(d); run_all<M_MF32_V8>(d);
    printf("one 16x16x32 MFMA + k independent fillers:\n");
    run_all<M_MF16_E2>(d); run_all<M_MF16_E4>(d); run_all<M_MF16_V2>(d); run_all<M_MF16_V4>(d);
    printf("one 16x16x32 MFMA + k independent fillers of the classes of the attention kernels' softmax:\n");
    run_all<M_MF16_CVT2>(d); run_all<M_MF16_CVT4>(d); run_all<M_MF16_MAX2>(d); run_all<M_MF16_MAX4>(d);
    run_all<M_MF16_PKADD2>(d); run_all<M_MF16_PKMUL2>(d); run_all<M_MF16_PKFMA16_2>(d); run_all<M_MF16_FMA2X2>(d);
    printf("heterogeneous waves on one SIMD (an MFMA-only wave beside a VALU-only wave):\n");
    run_het<M_HET_EXP>(d); run_het<M_HET_CVT>(d); run_het<M_HET_FMA>(d);
    printf("one MFMA-only wave beside 1 / 2 / 3 VALU-only waves on the same SIMD (round 6):\n");
    for (int vw = 1; vw <= 3; ++vw) { run_het_n<M_HET_EXP>(d, vw); run_het_n<M_HET_CVT>(d, vw); run_het_n<M_HET_FMA>(d, vw); }
    return 0;
}
