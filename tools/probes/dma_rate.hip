// Probe: how fast can every CU pull the SAME weight matrix out of L2 (the access pattern of the fused
// cross-attention's weight slabs)?  Variants: LDS-DMA (buffer_load ... lds) vs ordinary 16-B loads,
// free running vs one barrier per 40-KB slab, same matrix for all workgroups vs a private one each.
// build: hipcc --offload-arch=gfx950 -O3 -o dma_rate dma_rate.hip ; run: ./dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: DMA free running, 1: DMA + barrier per slab, 2: plain loads to registers
__global__ void __launch_bounds__(512) k(const char* w, float* sink, int nslab, long wg_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = w + (long)blockIdx.x * wg_stride;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    f4 acc = {0, 0, 0, 0};
    for (int s = 0; s < nslab; ++s) {
        const int slot = s & 1;
        if (MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const unsigned off = (unsigned)(((s % 5) * 40 + wave + 8 * j) * 1024 + lane * 16);
            if (MODE == 2) acc += *reinterpret_cast<const f4*>(base + off);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + slot * 40960 + (wave + 8 * j) * 1024), 16, off, 0, 0, 0);
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 2) sink[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    const int nwg = 256, nslab = 200;          // 200 slabs x 40 KB = 8 MB per workgroup
    char* w; float* sink;
    hipMalloc(&w, (size_t)nwg * 204800 + 4096); hipMalloc(&sink, nwg * 512 * 4);
    hipMemset(w, 1, (size_t)nwg * 204800);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name, long stride) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), 81920, 0, (const char*)w, sink, nslab, stride);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)nwg * nslab * 40960;
        printf("%-46s %8.1f us  %7.2f TB/s chip  %6.1f GB/s per CU  (%.1f B/clk/CU at 2.1 GHz)\n", name, ms * 1e3,
               bytes / ms / 1e9, bytes / nwg / ms / 1e6, bytes / nwg / (ms * 1e-3) / 2.1e9);
    };
    run(k<0>, "DMA free-running, same 200 KB for all WGs", 0);
    run(k<1>, "DMA + barrier per 40 KB slab, same matrix", 0);
    run(k<2>, "plain 16-B loads to VGPRs, same matrix", 0);
    run(k<0>, "DMA free-running, private 200 KB per WG", 204800);
    run(k<1>, "DMA + barrier per slab, private per WG", 204800);
    run(k<2>, "plain loads, private per WG", 204800);
    return 0;
}
