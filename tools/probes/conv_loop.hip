// Prototype of the 3x3 stride-1 halo convolution on v_mfma_f32_32x32x16_f16 with ONE compute wave per SIMD:
// tile 256 tokens x 160 channels, wave tile 64 tokens x 160 channels (2 x 5 accumulator tiles of 32 x 32), halo tile per
// 64-channel slab in LDS (two buffers), weights through a three-stage LDS ring by DMA (buffer_load ... lds) with counted
// vmcnt.  Variants timed in one run (same data, same check against a CPU reference on sampled outputs):
//   MODE 0: four waves, every wave computes AND issues its share of the DMA pieces
//   MODE 1: eight waves = four compute waves + four loader waves (wave i + 4 shares the SIMD of wave i)
//   ILV   : 1 = ds_reads interleaved with the MFMAs by sched_group_barrier, 0 = reads first, then the MFMA block
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/conv_loop.hip -o tools/probes/conv_loop ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
#define DEVINL __device__ __forceinline__

struct Args {
    const half_t* x; const half_t* w; half_t* out; unsigned long long* cyc;
    int M, N, C, H, W;
    unsigned bytes_x, bytes_w;
    int nostore;
    int abl;       // ablation: 1 = no weight DMA in the loop, 2 = no halo DMA in the loop (stale LDS is multiplied; timing only)
};

constexpr int BM = 256, BN = 160, BK = 64;
constexpr int WST = BN * 128;            // bytes of one weight stage: 160 rows x 64 halfs
constexpr int NSTG = 3;
constexpr int HQ = 13;                   // halo piece slots per loader wave (4 x 13 = 52 >= 50 pieces)
constexpr int WQ = 5;                    // weight pieces per loader wave and slab (20 pieces)

DEVINL int key(int r) { return (r >> 1) & 7; }

DEVINL void wait_vmcnt(int n) {
#define VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        VM(0) VM(1) VM(2) VM(3) VM(4) VM(5) VM(6) VM(7) VM(8) VM(9) VM(10) VM(11) VM(12) VM(13) VM(14) VM(15) VM(16)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef VM
}

template <int N>
DEVINL void pin(half8 (&f)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i]));
}

template <int MODE, int ILV, int CABL = 0>      // CABL (timing only): 1 = no fragment reads in the loop, 2 = no barrier in the loop
__global__ void __launch_bounds__(MODE == 1 ? 512 : 256, 1)
conv_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r8 = lane >> 3, c8 = lane & 7;
    const int l32 = lane & 31, lh = lane >> 5;
    int bid = blockIdx.y * gridDim.x + blockIdx.x;
    {
        const int nwg = gridDim.x * gridDim.y;
        if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    }
    const int n0 = (bid % (int)gridDim.x) * BN;
    const int m0 = (bid / (int)gridDim.x) * BM;
    const int W = a.W, H = a.H, HW = H * W, C = a.C;
    const int seg_tok = BM < HW ? BM : HW;
    const int rs = seg_tok / W;
    const int hs = (rs + 2) * (W + 2);
    const int nh = (BM / seg_tok) * hs;
    const int HP = (nh + 7) >> 3;                 // halo pieces
    const int HS = HP * 1024;                     // bytes of one halo buffer
    const int img0 = m0 / HW;
    const int y0 = (m0 - img0 * HW) / W;
    char* hbuf = smem;
    char* wbuf = smem + 2 * HS;
    const int ncs = C / BK;
    const int S = ncs * 9;
    const int K9 = 9 * C;

    const bool is_loader = MODE == 1 ? wave >= 4 : true;
    const bool is_compute = MODE == 1 ? wave < 4 : true;
    const int lw = MODE == 1 ? (wave & 3) : wave;           // loader index 0..3

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.bytes_x, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.bytes_w, 0x00020000);

    // ---- loader state ------------------------------------------------------------------------------------------------
    unsigned hoff[HQ], woff[WQ];
    if (is_loader) {
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int p = lw + 4 * q;
            const int hr = p * 8 + r8;
            const int seg = hr / hs, rem = hr - seg * hs;
            const int hy = rem / (W + 2), hx = rem - hy * (W + 2);
            const int img = img0 + seg;
            const int yy = y0 + hy - 1, xx = hx - 1;
            const bool ok = (hr < nh) && (yy >= 0) && (yy < H) && (xx >= 0) && (xx < W) && ((long)img * HW < a.M);
            const long row = ((long)img * H + yy) * W + xx;
            hoff[q] = ok ? (unsigned)((row * C + (c8 ^ key(hr)) * 8) * 2) : OOB;
        }
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int R = (lw + 4 * q) * 8 + r8;
            woff[q] = (unsigned)(((long)(n0 + R) * K9 + (c8 ^ key(R)) * 8) * 2);
        }
    }
    const int nq = is_loader ? ((HP - lw + 3) >> 2) : 0;          // valid halo slots of this loader: pieces lw, lw + 4, ... < HP
    auto hcount = [&](int tap) { const int c = nq - 2 * tap; return tap > 6 ? 0 : (c < 0 ? 0 : (c > 2 ? 2 : c)); };
    auto issue_w = [&](int k, int stage) {
        const int cs = k / 9, tap = k - cs * 9;
        const unsigned soff = (unsigned)((tap * C + cs * BK) * 2);
        char* dst = wbuf + stage * WST;
#pragma unroll
        for (int q = 0; q < WQ; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(dst + (lw + 4 * q) * 1024), 16, woff[q], soff, 0, 0);
    };
    auto issue_h1 = [&](int cs, int q) {      // one halo piece slot of channel slab cs (q is a compile-time constant at every call)
        if (q < HQ && q < nq)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(hbuf + (cs & 1) * HS + (lw + 4 * q) * 1024), 16, hoff[q < HQ ? q : 0],
                                                     (unsigned)(cs * BK * 2), 0, 0);
    };
    // window (cs, tap) (behind the barrier that publishes slab k = 9 cs + tap): W(k + 2) into its stage, two halo slots of cs + 1
    auto window = [&](int cs, int tap) {
        const int k = cs * 9 + tap;
        if (k + 2 < S && !(a.abl & 1)) issue_w(k + 2, (k + 2) % NSTG);
        if (cs + 1 < ncs && tap <= 6 && !(a.abl & 2)) { issue_h1(cs + 1, 2 * tap); issue_h1(cs + 1, 2 * tap + 1); }
    };
    // before the barrier that publishes slab k + 1: W(k + 1) has landed; younger issues: the halo pieces of window k - 1, all of window k
    auto wait_for = [&](int cs, int tap) {
        const int k = cs * 9 + tap;
        int n = (k + 2 < S ? WQ : 0) + ((cs + 1 < ncs) ? hcount(tap) : 0);
        if (tap >= 1) n += (cs + 1 < ncs) ? hcount(tap - 1) : 0;      // (tap 0: the previous window is tap 8 of cs - 1, no halo pieces)
        if (a.abl) n = 0;
        wait_vmcnt(n);
    };

    // ---- compute state -----------------------------------------------------------------------------------------------
    int hbase[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ml = ((MODE == 1 ? wave & 3 : wave) * 2 + t) * 32 + l32;
        const int seg = ml / seg_tok, rem = ml - seg * seg_tok;
        const int y = rem / W, x = rem - y * W;
        hbase[t] = seg * hs + (y + 1) * (W + 2) + (x + 1);
        asm volatile("" : "+v"(hbase[t]));
    }
    const int wlane = l32 * 128 + ((lh ^ key(l32)) << 4);
    f32x16 acc[2][5];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][c][i] = 0.f;
    int xaddr[2];
    auto set_slab = [&](int cs, int tap) {
        const int ty = tap / 3;
        const int shift = (ty - 1) * (W + 2) + (tap - ty * 3 - 1);
        const int hsel = (cs & 1) * HS;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int hb = hbase[t];
            asm volatile("" : "+v"(hb));             // (keeps the nine taps' addresses from being hoisted out of the channel-slab loop: 72 registers)
            const int row = hb + shift;
            xaddr[t] = hsel + row * 128 + ((lh ^ key(row)) << 4);
        }
    };
    auto read_frags = [&](int stage, int j, half8 (&xf)[2], half8 (&wf)[5], bool in_loop = true) {
        if ((CABL & 1) && in_loop) return;
        const char* wsb = wbuf + stage * WST + (wlane ^ (j << 5));
#pragma unroll
        for (int t = 0; t < 2; ++t) xf[t] = *reinterpret_cast<const half8*>(hbuf + (xaddr[t] ^ (j << 5)));
#pragma unroll
        for (int c = 0; c < 5; ++c) wf[c] = *reinterpret_cast<const half8*>(wsb + c * 4096);
    };
    auto mma = [&](const half8 (&xf)[2], const half8 (&wf)[5]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[c], xf[t], acc[t][c], 0, 0, 0);
    };
    auto interleave = [&]() {
        if (ILV) {
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
        }
    };

    // ---- prologue ----------------------------------------------------------------------------------------------------
    if (is_loader) {
#pragma unroll
        for (int q = 0; q < HQ; ++q) issue_h1(0, q);
        issue_w(0, 0);
        if (1 < S) issue_w(1, 1);
        wait_vmcnt(1 < S ? WQ : 0);
    }
    __builtin_amdgcn_s_barrier();                 // barrier 0: halo(0), W(0) visible
    if (MODE == 1 && !is_compute) {
        // ---- loader waves ----
        for (int cs = 0; cs < ncs; ++cs) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                window(cs, tap);
                wait_for(cs, tap);
                if (!(CABL & 2)) __builtin_amdgcn_s_barrier();         // publishes slab 9 cs + tap + 1
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    const unsigned long long t_start = __builtin_readcyclecounter();
    half8 xf0[2], wf0[5], xf1[2], wf1[5];
    set_slab(0, 0);
    read_frags(0, 0, xf0, wf0, false);
    read_frags(0, 1, xf1, wf1, false);
    if (MODE == 0) window(0, 0);
    for (int cs = 0; cs < ncs; ++cs) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int st = tap % NSTG, stn = (tap + 1) % NSTG;      // (9 slabs per channel slab, three stages: static)
            // j = 0
            read_frags(st, 1, xf1, wf1);
            mma(xf0, wf0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            // j = 1
            read_frags(st, 2, xf0, wf0);
            mma(xf1, wf1);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            // j = 2
            read_frags(st, 3, xf1, wf1);
            mma(xf0, wf0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            // barrier: every wave's reads of this slab have landed, the next slab's weights are visible
            if (MODE == 0) wait_for(cs, tap);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(CABL & 2)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 0) { if (tap < 8) window(cs, tap + 1); else if (cs + 1 < ncs) window(cs + 1, 0); }
            // j = 3, reading the first fragments of the next slab meanwhile (behind the last slab: stale LDS, never used)
            if (tap < 8) set_slab(cs, tap + 1); else set_slab(cs + 1, 0);
            read_frags(stn, 0, xf0, wf0);
            mma(xf1, wf1);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.cyc && lane == 0 && (MODE == 1 ? wave & 3 : wave) == 0) a.cyc[bid] = __builtin_readcyclecounter() - t_start;

    // ---- epilogue (probe: direct 8-byte stores, no bias) -------------------------------------------------------------
    const int cw = MODE == 1 ? wave & 3 : wave;
    if (a.nostore) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 5; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) s += acc[t][c][i];
        if (s == 123.456f) a.out[0] = (half_t)s;
        return;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = m0 + (cw * 2 + t) * 32 + l32;
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + c * 32 + g * 8 + lh * 4;
                half4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (half_t)acc[t][c][g * 4 + i];
                if (m < a.M) *reinterpret_cast<half4*>(a.out + (long)m * a.N + n) = o;
            }
    }
}

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; }

// `conv_loop long`: the loader / compute variant on the level-0 shape back to back for ~4 s, printing the clock the cycle counter
// implies for every batch of launches (compare with rocm-smi's sclk sampled meanwhile: is the reported clock the throttled one?)
static void run_long() {
    const int B = 8, side = 64, C = 320, N = 320, M = B * side * side, K9 = 9 * C;
    std::vector<half_t> hx((size_t)M * C), hw((size_t)N * K9);
    unsigned seed = 12345;
    for (auto& v : hx) v = (half_t)(frand(seed) * 0.5f);
    for (auto& v : hw) v = (half_t)(frand(seed) * 0.05f);
    half_t *dx, *dw, *dout; unsigned long long* dcyc;
    hipMalloc(&dx, hx.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&dout, (size_t)M * N * 2); hipMalloc(&dcyc, 4096 * 8);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    Args a{dx, dw, dout, dcyc, M, N, C, side, side, (unsigned)(hx.size() * 2), (unsigned)(hw.size() * 2), 1, 0};
    const int smem = 2 * 50 * 1024 + NSTG * WST;
    auto kern = conv_kernel<1, 1>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int batch = 0; batch < 16; ++batch) {
        hipEventRecord(e0);
        for (int i = 0; i < 5000; ++i) hipLaunchKernelGGL(kern, dim3(N / BN, M / BM), dim3(512), smem, 0, a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> hc(256);
        hipMemcpy(hc.data(), dcyc, hc.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (auto c : hc) sum += (double)c;
        const double us = ms * 1e3 / 5000, cyc = sum / 256;
        printf("batch %2d: %6.1f us per launch, loop %6.0f cycles -> >= %.2f GHz (loop cycles / whole launch time), %.0f TF/s\n", batch, us, cyc,
               cyc / us / 1e3, 2.0 * M * N * K9 / us / 1e6);
        fflush(stdout);
    }
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'l') { run_long(); return 0; }
    struct Shape { const char* name; int B, side, C, N; };
    const Shape shapes[] = {{"L0 320->320", 8, 64, 320, 320}, {"L0 640->320", 8, 64, 640, 320}, {"L1 640->640", 8, 32, 640, 640},
                            {"L2 1280->1280", 8, 16, 1280, 1280}};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (const Shape& sh : shapes) {
        const int M = sh.B * sh.side * sh.side, C = sh.C, N = sh.N, K9 = 9 * C;
        std::vector<half_t> hx((size_t)M * C), hw((size_t)N * K9);
        unsigned seed = 12345;
        for (auto& v : hx) v = (half_t)(frand(seed) * 0.5f);
        for (auto& v : hw) v = (half_t)(frand(seed) * 0.05f);
        half_t *dx, *dw, *dout;
        hipMalloc(&dx, hx.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&dout, (size_t)M * N * 2);
        hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
        unsigned long long* dcyc; hipMalloc(&dcyc, 4096 * 8);
        Args a{dx, dw, dout, dcyc, M, N, C, sh.side, sh.side, (unsigned)(hx.size() * 2), (unsigned)(hw.size() * 2), 0, 0};
        const int HW = sh.side * sh.side, seg = 256 < HW ? 256 : HW;
        const int nh = (256 / seg) * (seg / sh.side + 2) * (sh.side + 2);
        const int smem = 2 * ((nh + 7) / 8) * 1024 + NSTG * WST;
        const double fl = 2.0 * M * N * K9;
        auto run = [&](auto kern, int threads, const char* name) {
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            dim3 grid(N / BN, M / BM);
            for (int ns = 0; ns < 2; ++ns) {
                a.nostore = ns;
                hipMemset(dout, 0, (size_t)M * N * 2);
                for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(threads), smem, 0, a);
                hipEventRecord(e0);
                for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, grid, dim3(threads), smem, 0, a);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double us = ms * 1e3 / 20;
                hipError_t err = hipGetLastError();
                if (ns == 1) {
                    std::vector<unsigned long long> hc(grid.x * grid.y);
                    hipMemcpy(hc.data(), dcyc, hc.size() * 8, hipMemcpyDeviceToHost);
                    double sum = 0, mx = 0; for (auto c : hc) { sum += (double)c; mx = mx > (double)c ? mx : (double)c; }
                    const int S = 9 * C / 64;
                    printf("    %-34s no-store: %7.1f us %7.1f TF/s   loop cycles avg %.0f max %.0f = %.0f per slab (40 MFMA = 1280); if the loop were the whole kernel: %.2f GHz\n",
                           "", us, fl / us / 1e6, sum / hc.size(), mx, sum / hc.size() / S, mx / us / 1e3);
                    continue;
                }
                // check 256 sampled outputs against fp64 on the host
                std::vector<half_t> ho((size_t)M * N);
                hipMemcpy(ho.data(), dout, ho.size() * 2, hipMemcpyDeviceToHost);
                double maxerr = 0; unsigned s2 = 777;
                for (int it = 0; it < 256; ++it) {
                    s2 = s2 * 1664525u + 1013904223u; const int m = (s2 >> 4) % M;
                    s2 = s2 * 1664525u + 1013904223u; const int n = (s2 >> 4) % N;
                    const int b = m / HW, y = (m % HW) / sh.side, x = m % sh.side;
                    double ref = 0;
                    for (int tap = 0; tap < 9; ++tap) {
                        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                        if (yy < 0 || yy >= sh.side || xx < 0 || xx >= sh.side) continue;
                        const half_t* xr = &hx[((size_t)(b * sh.side + yy) * sh.side + xx) * C];
                        const half_t* wr = &hw[(size_t)n * K9 + tap * C];
                        for (int c = 0; c < C; ++c) ref += (double)xr[c] * (double)wr[c];
                    }
                    const double e = fabs((double)ho[(size_t)m * N + n] - ref) / (fabs(ref) + 0.05);
                    if (e > maxerr) maxerr = e;
                }
                printf("  %-12s %-22s %7.1f us %7.1f TF/s  (lds %d B, grid %d, max rel err %.2e%s, %s)\n", sh.name, name, us, fl / us / 1e6, smem,
                       grid.x * grid.y, maxerr, maxerr < 5e-3 ? "" : " MISMATCH", hipGetErrorString(err));
            }
        };
        run(conv_kernel<0, 0>, 256, "4 waves, reads first");
        run(conv_kernel<0, 1>, 256, "4 waves, interleaved");
        run(conv_kernel<1, 0>, 512, "4+4 loaders, reads first");
        run(conv_kernel<1, 1>, 512, "4+4 loaders, interleaved");
        a.abl = 1; run(conv_kernel<1, 1>, 512, "  same, no W DMA (abl)");
        a.abl = 2; run(conv_kernel<1, 1>, 512, "  same, no halo DMA (abl)");
        a.abl = 3; run(conv_kernel<1, 1>, 512, "  same, no DMA at all (abl)");
        a.abl = 0;
        run(conv_kernel<1, 1, 1>, 512, "  same, no frag reads (abl)");
        run(conv_kernel<1, 1, 2>, 512, "  same, no barrier (abl)");
        run(conv_kernel<1, 1, 3>, 512, "  same, no reads, no barrier");
        hipFree(dx); hipFree(dw); hipFree(dout);
    }
    return 0;
}
