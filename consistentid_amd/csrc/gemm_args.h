// Launch arguments shared by the implicit-GEMM kernels (gemm.hip) and the 3x3 halo convolution on 32 x 32 MFMA tiles
// (conv3x3.hip): filled by plan_gemm (gemm.hip) from a cid_gemm_desc.
#pragma once
#include "common.h"

namespace cidg {

struct GemmArgs {
    const half_t* x1; const half_t* x2;
    int c1, c2, ld1, ld2;
    const half_t* w;
    half_t* out; int ldo;
    half_t* out2;        // second destination of every output row (same pitch), or nullptr
    const half_t* bias;
    const half_t* rowbias; int ld_rowbias; int rows_per_sample;
    const half_t* res; int ldr;
    int M, N, taps;
    int Hi, Wi, Ho, Wo, stride, up;
    int mode;
    half_t* vt; int n_vt0, heads, dhead, dvp, ntok;
    int n_begin, n_end;  // column range covered by this launch
    int ktot;            // taps * (c1 + c2)
    int nslab;           // ktot / 64
    int cslabs;          // (c1 + c2) / 64
    int splitk;          // gridDim.z
    int nloop;           // consecutive n-tiles walked by ONE workgroup (GEGLU launches; 1 = one tile per workgroup)
    int nbuf;            // LDS stages of the DMA ring (2, or 3 where plan_gemm finds the launch latency-bound)
    // mode 3 (query projection with the identity cross-attention as its epilogue): packed K / V^T of the context rows
    // (cid_kv_pack_f16), context row of every sample, halfs per packed row, context layout, ID-stream scale
    const half_t* att_kp; const half_t* att_vp; const int* att_kvrow;
    long att_krow, att_vrow;
    int att_n_txt, att_n_ip;
    float att_scale;
    unsigned bytes_x1, bytes_x2, bytes_w;   // buffer-descriptor ranges
    float* ws;           // [splitk][M][N] fp32 partials when splitk > 1
    const float* ln_s;   // LayerNorm folded into the projection: row sums of W' = W diag(gamma) ...
    const float* ln_b;   // ... and W beta + bias; out = rstd * (acc - mean * ln_s[n]) + ln_b[n]
    float ln_eps;
    float* gn_stats;     // [M / BM][N / gn_unit][2] partial (sum, sum of squares) of the fp16 outputs, or nullptr
    int gn_unit;         // channels per statistics unit (N / 32: the tensor's own GroupNorm group width)
    int ablate;          // profiling knob (CID_GEMM_ABLATE): 1 = no global loads in the loop,
                         // 2 = no MFMA, 3 = no LDS fragment reads / MFMA
    int xcd_pn;          // tile -> XCD partition (xcd_tile below): n-blocks of the 2-D partition (1 | 2 | 4 | 8), 0 = linear runs
};

// Tile of a workgroup.  The hardware deals consecutive workgroup ids round-robin over the 8 XCDs (a private 4-MB L2 each): the
// tile grid [gridDim.y token tiles] x [gridDim.x n-tiles] is cut into 8 / pn row blocks x pn column blocks, one block per XCD
// (inside a block n runs fastest).  Every weight byte then crosses the fabric 8 / pn times and every activation byte pn times;
// choose_xcd_pn picks pn for the launch's byte counts.  pn = 1 is the partition of rounds 2-5 (an XCD owns whole token tiles:
// right while the activations outweigh the weights); at the 16 x 16 / 8 x 8 levels the weights outweigh them 6 : 1 and a
// launch fetched its 26-30 MB of weights EIGHT times (profiles/r06_pmc_gemm_family.txt: 210 MB read for 31 MB algorithmic).
CID_DEVINL void xcd_tile(const GemmArgs& a, int& nb, int& mb) {
    const int gx = gridDim.x, gy = gridDim.y;
    int bid = blockIdx.y * gx + blockIdx.x;
    if (a.xcd_pn > 0) {
        const int pn = a.xcd_pn;
        const int x = bid & 7, l = bid >> 3;
        const int bnc = gx / pn;                       // n-tiles per block ((gy * pn / 8) token tiles per block)
        nb = (x % pn) * bnc + l % bnc;
        mb = (x / pn) * (gy * pn >> 3) + l / bnc;
    } else {
        const int nwg = gx * gy;
        if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
        nb = bid % gx;
        mb = bid / gx;
    }
}

// host side: pn in {1, 2, 4, 8} (pm = 8 / pn) minimising pm * weight bytes + pn * activation bytes among the partitions the
// grid divides into; 0 (linear runs, the old order) when none does.  CID_XCD_2D=0 pins the old order (A/B switch).
int choose_xcd_pn(int gx, int gy, double w_bytes, double x_bytes);

// s_waitcnt vmcnt(N) with a run-time (wave-uniform) N
CID_DEVINL void wait_vmcnt(int n) {
#define CID_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        CID_VM(0) CID_VM(1) CID_VM(2) CID_VM(3) CID_VM(4) CID_VM(5) CID_VM(6) CID_VM(7) CID_VM(8) CID_VM(9)
        CID_VM(10) CID_VM(11) CID_VM(12) CID_VM(13) CID_VM(14) CID_VM(15) CID_VM(16) CID_VM(17) CID_VM(18)
        CID_VM(19) CID_VM(20) CID_VM(21) CID_VM(22) CID_VM(23) CID_VM(24) CID_VM(25) CID_VM(26) CID_VM(27)
        CID_VM(28) CID_VM(29) CID_VM(30) CID_VM(31) CID_VM(32)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef CID_VM
}

// 3x3 stride-1 halo convolution, bm-token (256 | 128) x 160-channel tiles on v_mfma_f32_32x32x16_f16 (conv3x3.hip); plan_gemm
// has checked: a tile = whole image rows of one image or whole images, halo <= 400 rows, N % 160 == 0, M % bm == 0
int launch_conv_h32(const GemmArgs& a, int bm, hipStream_t s);

// GEGLU projection on 32 x 32 x 16 MFMA tiles with loader / compute wave roles (linear_h32.hip); plan_gemm has checked: one
// source, taps == 1, no LayerNorm fold, M % 256 == 0, N % 160 == 0, a.nloop divides N / 160, at least three channel slabs
int launch_geglu_h32(const GemmArgs& a, hipStream_t s);

}  // namespace cidg
