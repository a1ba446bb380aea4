// FeedForward's GEGLU projection (diffusers FeedForward / GEGLU inside BasicTransformerBlock as driven
// from /root/reference/attention.py:128-167 through the UNet call of pipline_StableDiffusion_ConsistentID.py:552-557) on
// v_mfma_f32_32x32x16_f16 with the wave roles of conv3x3.hip:
//   out[m][o] = (x W_v^T + b_v)[m][o] * gelu_erf((x W_g^T + b_g)[m][o]),   W = [W_v ; W_g] interleaved by the host in 16-row
//   value / gate blocks (weights.py: interleaved column n -> group n / 32, rows 0..15 of a group the value rows of outputs
//   16 (n / 32) .. + 15, rows 16..31 their gate rows).
// Why a second kernel for this op: the 16x16x32 form of gemm.hip (128 x 128 tiles, wave tile 32 x 64) reads six 1-KiB LDS
// fragments per eight MFMAs -- 192 B per clock and CU at the matrix pipe's issue rate against the 128 B per clock the LDS
// delivers, DMA writes into the same LDS not counted.  With DMA, MFMA, erf and stores all ablated that kernel keeps 58 of its
// 74 us at the SD1.5 level-0 shape (profiles/r06_gemm_ablation.txt): it is LDS-bound.  A 64 x 160 wave tile of 32 x 32 x 16
// MFMAs needs seven fragments per ten MFMAs of twice the length: 90 B per clock.
//   * four COMPUTE waves (one per SIMD): 64 tokens x 160 interleaved columns = 160 accumulator registers, ds_read_b128 + MFMA
//     interleaved one to one; at the end of an n-tile they run the GEGLU epilogue on their own accumulators: value and gate of
//     an output sit in the SAME lane (accumulator registers 4 g + i and 4 (g + 2) + i of a 32-column tile), the two 4-output
//     runs of a lane are traded with the lane 32 further on (v_permlane32_swap) so that a lane stores 16 contiguous bytes;
//   * four LOADER waves: nothing but DMA (buffer_load ... lds) -- per 64-deep slab the 256 x 64 activation slab and the
//     160 x 64 weight slab, three LDS stages, counted vmcnt -- plus, once per n-tile, the tile's 160 bias values;
//   * N-loop (as in gemm.hip): a workgroup walks nloop consecutive n-tiles of its 256-token tile as ONE slab sequence, the ring
//     runs ahead across tile boundaries;
//   * ONE barrier per slab, between the third and the fourth k-step of the compute waves.
// The erf epilogue (about 900 VALU per 64 x 160 tile and wave) is NOT overlapped inside a workgroup: beside a saturated MFMA
// wave a SIMD issues one VALU instruction per 11.7 cycles however many waves offer them (profiles/r06_issue_rates.txt), so the
// loader waves cannot take it; and a second accumulator set to drain in the MFMAs' own shadow only fits a 32 x 160 wave tile,
// whose loop is LDS-bound again (built and measured: profiles/r06_geglu_h32.txt, 112 vs 74 us at level 0).  plan_gemm
// therefore routes this kernel for deep K only (>= 1024 channels: one exposed epilogue per twenty slabs): SD1.5's 16 x 16
// level 65 vs 75 us, SDXL's 32 x 32 level (60 of its 70 layers) 103 vs 118 us hot, 104 vs 126 us inside the step.
// Roofline: MFMA; algorithmic work 2 M N K flop, bytes (M K + N K + M N / 2) * 2.  Inside a denoise step the shorter kernel
// is paid back as a lower chip clock at lower power, not as time (profiles/r06_dvfs_ab.txt).
#include "gemm_args.h"
#include "../../include/cid.h"

namespace {
using cidg::GemmArgs;
using cidg::wait_vmcnt;

constexpr int BN = 160, BK = 64;
constexpr int WST = BN * 128;            // bytes of one weight slab: 160 rows x 64 halfs
constexpr int NSTG = 3;
constexpr int WQ = 5;                    // weight pieces (8 rows, 1 KiB) per loader wave and slab
constexpr int BIAS_IMG = 512;            // bytes of one bias image in LDS: 160 halfs, written by two 256-byte DMA pieces

typedef __attribute__((address_space(3))) void lds_void;

CID_DEVINL int key(int r) { return (r >> 1) & 7; }      // bank swizzle of conv3x3.hip: chunk c of row r sits at c ^ key(r)

template <int TM>
__global__ void __launch_bounds__(512, 1)
geglu_h32_kernel(GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128 * TM;
    constexpr int XST = BM * 128;                 // bytes of one activation slab
    constexpr int STG = XST + WST;
    constexpr int XQ = BM / 32;                   // activation pieces per loader wave and slab
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* bias_img = smem + NSTG * STG;           // [2][BIAS_IMG]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r8 = lane >> 3, c8 = lane & 7;
    const int l32 = lane & 31, lh = lane >> 5;
    int nb_, mb_;
    cidg::xcd_tile(a, nb_, mb_);                  // XCD-aware tile order (gemm_args.h)
    const int ncs = a.cslabs, nloop = a.nloop;
    const int F = ncs * nloop;                    // slabs of the flattened (n-tile, channel slab) sequence
    const int n0 = nb_ * BN * nloop;
    const int m0 = mb_ * BM;
    const bool is_loader = wave >= 4;
    const int lw = wave & 3;

    if (is_loader) {
        // =========================================== loader waves ===========================================================
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x1, 0, a.bytes_x1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.bytes_w, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? (unsigned)(a.N * 2) : 0u, 0x00020000);
        unsigned xoff[XQ], woff[WQ];
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int R = (lw + 4 * q) * 8 + r8;
            xoff[q] = (unsigned)(((long)(m0 + R) * a.ld1 + (c8 ^ key(R)) * 8) * 2);
        }
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int R = (lw + 4 * q) * 8 + r8;
            woff[q] = (unsigned)(((long)(n0 + R) * a.ktot + (c8 ^ key(R)) * 8) * 2);
        }
        const unsigned wstep = (unsigned)((long)BN * a.ktot * 2);      // bytes from one n-tile's weight rows to the next
        // issue cursor: slab `is_f` = channel slab is_cs of n-tile is_j, into stage is_st
        int is_cs = 0, is_j = 0, is_st = 0;
        unsigned is_w = 0u;
        // one slab: (the bias image of a new n-tile first -- it is older than the slab's pieces, so it has landed when they have)
        auto issue_next = [&]() -> int {
            int n = XQ + WQ;
            if (is_cs == 0 && lw == 0) {
                char* dst = bias_img + (is_j & 1) * BIAS_IMG;
                const unsigned boff = (unsigned)((n0 + is_j * BN) * 2 + lane * 4);
                // 160 halfs = 320 bytes: lanes 0..63 of the first piece, lanes 0..15 of the second (the others read past the
                // tile's values: harmless, nobody reads them back)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)dst, 4, boff, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(dst + 256), 4, boff + 256u, 0, 0, 0);
                n += 2;
            }
            char* xs = smem + is_st * STG;
            char* ws = xs + XST;
            const unsigned coff = (unsigned)(is_cs * BK * 2);
#pragma unroll
            for (int q = 0; q < XQ; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(xs + (lw + 4 * q) * 1024), 16, xoff[q], coff, 0, 0);
#pragma unroll
            for (int q = 0; q < WQ; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(ws + (lw + 4 * q) * 1024), 16, woff[q], coff + is_w, 0, 0);
            if (++is_cs == ncs) { is_cs = 0; ++is_j; is_w += wstep; }
            is_st = is_st + 1 == NSTG ? 0 : is_st + 1;
            return n;
        };
        issue_next();
        const int n1 = F > 1 ? issue_next() : 0;
        wait_vmcnt(n1);
        __builtin_amdgcn_s_barrier();                 // publishes slab 0
        for (int f = 0; f < F; ++f) {
            // behind the barrier that publishes slab f: slab f + 2 goes into the stage slab f - 1 has left; slab f + 1 must
            // have landed at the next barrier, what this window issues may keep flying
            const int n = f + 2 < F ? issue_next() : 0;
            wait_vmcnt(n);
            __builtin_amdgcn_s_barrier();             // publishes slab f + 1
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // =========================================== compute waves ==========================================================
        int xaddr[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int row = (lw * TM + t) * 32 + l32;
            xaddr[t] = row * 128 + ((lh ^ key(row)) << 4);
        }
        const int wlane = XST + l32 * 128 + ((lh ^ key(l32)) << 4);
        f32x16 acc[TM][5];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int c = 0; c < 5; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][c][i] = 0.f;
        // k-step j of a slab: chunk (2 j + lh) ^ key == ((lh ^ key) ^ (2 j)): one address per tile, XOR j << 5
        auto read_frags = [&](const char* stg, int j, half8 (&xf)[TM], half8 (&wf)[5]) {
#pragma unroll
            for (int t = 0; t < TM; ++t) xf[t] = *reinterpret_cast<const half8*>(stg + (xaddr[t] ^ (j << 5)));
            const char* wsb = stg + (wlane ^ (j << 5));
#pragma unroll
            for (int c = 0; c < 5; ++c) wf[c] = *reinterpret_cast<const half8*>(wsb + c * 4096);
        };
        // operands "swapped" (MFMA A = weight rows, B = token rows): a lane owns ONE token and, per 32-column tile, four quads of
        // consecutive interleaved columns (8 g + 4 lh + i): g = 0, 1 value columns, g = 2, 3 the gate columns of the same outputs
        auto mma = [&](const half8 (&xf)[TM], const half8 (&wf)[5]) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int c = 0; c < 5; ++c) acc[t][c] = mfma32(wf[c], xf[t], acc[t][c]);
        };
        auto interleave = [&]() {                     // MFMA, read, MFMA, read ... : the reads issue in the MFMAs' shadow
            if constexpr (TM == 2) {
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        };
        __builtin_amdgcn_s_barrier();                 // slab 0 is published
        half8 xf0[TM], wf0[5], xf1[TM], wf1[5];
        read_frags(smem, 0, xf0, wf0);
        int cs = 0, jt = 0, st = 0;
        for (int f = 0; f < F; ++f) {
            const char* stg = smem + st * STG;
            const int stn = st + 1 == NSTG ? 0 : st + 1;
            read_frags(stg, 1, xf1, wf1);
            mma(xf0, wf0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            read_frags(stg, 2, xf0, wf0);
            mma(xf1, wf1);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            read_frags(stg, 3, xf1, wf1);
            mma(xf0, wf0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            // every wave's reads of this slab have landed (its stage may be refilled); the next slab is published
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // fourth k-step, the next slab's first fragments travel under it (behind the last slab: stale LDS, unused)
            read_frags(smem + stn * STG, 0, xf0, wf0);
            mma(xf1, wf1);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            st = stn;
            if (++cs == ncs) {
                // ---- n-tile complete: GEGLU on the wave's own accumulators, registers -> HBM ----------------------------------
                const half_t* bimg = reinterpret_cast<const half_t*>(bias_img + (jt & 1) * BIAS_IMG);
                const bool has_bias = a.bias != nullptr;
                const int ocol = ((n0 + jt * BN) >> 1) + 8 * lh;
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    half_t* orow = a.out + (long)(m0 + (lw * TM + t) * 32 + l32) * a.ldo + ocol;
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
                        unsigned pk[2][2];        // [g][pair]: outputs 8 g + 4 lh + (0..3) of the tile's 16, as packed halfs
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            float bv[4] = {0.f, 0.f, 0.f, 0.f}, bg[4] = {0.f, 0.f, 0.f, 0.f};
                            if (has_bias) {
                                const half4 hv = *reinterpret_cast<const half4*>(bimg + c * 32 + 8 * g + 4 * lh);
                                const half4 hg = *reinterpret_cast<const half4*>(bimg + c * 32 + 16 + 8 * g + 4 * lh);
#pragma unroll
                                for (int i = 0; i < 4; ++i) { bv[i] = (float)hv[i]; bg[i] = (float)hg[i]; }
                            }
                            float o[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                o[i] = (acc[t][c][4 * g + i] + bv[i]) * gelu_erf_f(acc[t][c][4 * (g + 2) + i] + bg[i]);
                            const half2v p0 = {(half_t)o[0], (half_t)o[1]}, p1 = {(half_t)o[2], (half_t)o[3]};
                            pk[g][0] = __builtin_bit_cast(unsigned, p0);
                            pk[g][1] = __builtin_bit_cast(unsigned, p1);
                        }
                        // lane (token, lh) holds outputs {4 lh .. + 3} (g = 0) and {8 + 4 lh .. + 3} (g = 1): swap the upper
                        // lanes' g = 0 run with the lower lanes' g = 1 run -> lh = 0 holds outputs 0..7, lh = 1 outputs 8..15
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                        const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
                        *reinterpret_cast<u32x4*>(orow + c * 16) = v;
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[t][c][i] = 0.f;
                    }
                }
                cs = 0; ++jt;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#endif
}

}  // namespace

template <int TM>
static int launch_geglu_tm(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 128 * TM;
    constexpr int SMEM = NSTG * (BM * 128 + WST) + 2 * BIAS_IMG;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static bool configured = false;
    if (!configured) {
        hipError_t herr = hipFuncSetAttribute((const void*)geglu_h32_kernel<TM>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (herr != hipSuccess) {
            cid_set_error("cid_gemm_f16: cannot reserve %d bytes of LDS (%s)", SMEM, hipGetErrorString(herr));
            return -5;
        }
        configured = true;
    }
    dim3 grid(a.N / BN / a.nloop, a.M / BM, 1);
    GemmArgs b = a;
    b.xcd_pn = cidg::choose_xcd_pn((int)grid.x, (int)grid.y, 2.0 * a.N * a.ktot, (double)a.bytes_x1);
    hipLaunchKernelGGL(geglu_h32_kernel<TM>, grid, dim3(512), SMEM, s, b);
    return 0;
}

int cidg::launch_geglu_h32(const GemmArgs& a, hipStream_t s) { return launch_geglu_tm<2>(a, s); }
