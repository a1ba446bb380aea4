// Which SIMD does wave i of an 8-wave workgroup land on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8] ...)
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/simd_map.hip -o tools/probes/simd_map ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(512) probe(unsigned* out, int lds_bytes) {
    extern __shared__ char smem[];
    if (lds_bytes < 0) smem[threadIdx.x] = 1;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
}
int main() {
    unsigned* d; hipMalloc(&d, 4096 * 8 * 4);
    for (int lds : {0, 64 * 1024, 160 * 1024}) {
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(probe, dim3(512), dim3(512), lds, 0, d, lds);
        hipDeviceSynchronize();
        unsigned h[512 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("LDS %d KB: simd of waves 0..7 for the first 6 workgroups:\n", lds / 1024);
        for (int b = 0; b < 6; ++b) { for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("   (cu %u)\n", (h[b * 8] >> 8) & 15); }
        int same = 0; for (int b = 0; b < 512; ++b) for (int w = 0; w < 4; ++w) same += ((h[b*8+w] >> 4) & 3) == ((h[b*8+w+4] >> 4) & 3);
        printf("  waves i and i+4 on the same SIMD: %d of %d\n", same, 512 * 4);
        int par = 0; for (int b = 0; b < 512; ++b) for (int w = 0; w < 8; w += 2) par += ((h[b*8+w] >> 4) & 3) == ((h[b*8+w+1] >> 4) & 3);
        printf("  waves 2i and 2i+1 on the same SIMD: %d of %d\n", par, 512 * 4);
    }
    return 0;
}
