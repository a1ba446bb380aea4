// Probe: the burst at the start of the fused cross-attention -- 512 workgroups (two per CU) each pull their
// own 40-KB token tile (64 rows x 640 B, contiguous in memory) HBM -> LDS at the same time.
// How long until the LAST workgroup has its tile, for
//   mode 0: LDS image slab-major [5][64][128 B]: every DMA instruction reads 8 rows x 128 B, 640 B apart (xattn2/3)
//   mode 1: LDS image row-major [64][640 B]: every DMA instruction reads 1 KiB of contiguous memory
//   mode 2: like 0, but only the first slab (8 KB per workgroup): the latency floor of the first k-slab
// Also prints the shader clock (s_memtime against the 100 MHz s_memrealtime).
// build: hipcc --offload-arch=gfx950 -O3 -o x_tile_rate x_tile_rate.hip ; run: ./x_tile_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((address_space(3))) void lds_void;

template <int MODE>
__global__ void __launch_bounds__(256) k(const char* x, unsigned long long* stamps, long tile0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = x + (tile0 + blockIdx.x) * 40960L;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 40960, 0x00020000);
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 10; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + (wave * 10 + j) * 1024), 16,
                                                     (unsigned)((wave * 10 + j) * 1024 + lane * 16), 0, 0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < (MODE == 2 ? 1 : 5); ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = (2 * wave + j) * 8 + (lane >> 3);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + s * 8192 + (2 * wave + j) * 1024), 16,
                                                         (unsigned)(r * 640 + (lane & 7) * 16), s * 128, 0, 0);
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;
        unsigned long long* p = stamps + blockIdx.x * 8;
        p[0] = t0; p[1] = t1; p[2] = r0; p[3] = r1; p[4] = xcc;
    }
}

__global__ void spin(unsigned long long* out) {        // shader clock against the 100 MHz real-time counter
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    float a = 1.f;
    for (int i = 0; i < 2000000; ++i) a = a * 1.0000001f + 1e-9f;
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)a; }
}

int main() {
    const int nwg = 512, rounds = 12;
    const size_t bytes = (size_t)nwg * 40960 * rounds;          // 252 MB: every round reads fresh memory
    char* x; unsigned long long* st;
    hipMalloc(&x, bytes); hipMalloc(&st, nwg * 8 * 8);
    hipMemset(x, 1, bytes);
    spin<<<1, 64>>>(st); hipDeviceSynchronize();
    unsigned long long c[3]; hipMemcpy(c, st, 24, hipMemcpyDeviceToHost);
    const double ghz = (double)c[0] / ((double)c[1] * 10.0);     // cycles per ns
    printf("shader clock (idle chip, one wave): %.2f GHz\n", ghz);
    std::vector<unsigned long long> h(nwg * 8);
    auto run = [&](auto kern, const char* name, double kb) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
        double worst_sum = 0, mean_sum = 0, rt_sum = 0; int n = 0;
        for (int it = 0; it < rounds; ++it) {
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 81920, 0, (const char*)x, st, (long)it * nwg);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), st, nwg * 64, hipMemcpyDeviceToHost);
            if (it < 2) continue;
            unsigned long long first[16], r_first = ~0ull, r_last = 0;
            for (int q = 0; q < 16; ++q) first[q] = ~0ull;
            for (int w = 0; w < nwg; ++w) {
                first[h[w * 8 + 4]] = std::min(first[h[w * 8 + 4]], h[w * 8]);
                r_first = std::min(r_first, h[w * 8 + 2]); r_last = std::max(r_last, h[w * 8 + 3]);
            }
            double worst = 0, mean = 0;
            for (int w = 0; w < nwg; ++w) {
                const double d = (double)(h[w * 8 + 1] - first[h[w * 8 + 4]]);
                worst = std::max(worst, d); mean += d / nwg;
            }
            worst_sum += worst; mean_sum += mean; rt_sum += (double)(r_last - r_first) * 10.0; ++n;
        }
        const double rt_ns = rt_sum / n;
        printf("%-46s last tile in after %7.0f cycles (mean %6.0f); real time first start -> last end %6.2f us = %5.2f TB/s\n",
               name, worst_sum / n, mean_sum / n, rt_ns * 1e-3, nwg * kb * 1024.0 / rt_ns * 1e-3);
    };
    run(k<0>, "slab-major image, 128-B pieces 640 B apart", 40);
    run(k<1>, "row-major image, contiguous 1-KiB pieces", 40);
    run(k<2>, "first 64-channel slab only (8 KB / workgroup)", 8);
    return 0;
}
