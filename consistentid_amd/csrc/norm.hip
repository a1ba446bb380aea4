// LayerNorm and GroupNorm(+SiLU) over token-major fp16 activations.
// HBM-bound: algorithmic bytes = read x once + write once.  GroupNorm: one read + one write when the GEMM that produced
// x emitted the statistics from its epilogue (cid_groupnorm_stats_f16) or when a (sample, group) slice fits a workgroup's
// registers (gn_small_kernel); otherwise a statistics pass + an apply pass (the second read is an L2 / Infinity-Cache
// hit for the tensor sizes of this UNet).  All reductions are deterministic (no atomics).
#include "common.h"
#include "../../include/cid.h"

namespace {

// ------------------------------------------------------------------ LayerNorm
// one wave per row; lanes hold up to 3 chunks of 8 channels (C <= 1536)
constexpr int LN_MAXCH = 3;

__global__ void __launch_bounds__(256)
layernorm_kernel(const half_t* __restrict__ x, half_t* __restrict__ out,
                 const half_t* __restrict__ gamma, const half_t* __restrict__ beta,
                 int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = C >> 3;
    const half_t* xr = x + (long)row * C;
    float v[LN_MAXCH][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXCH; ++k) {
        const int c = lane + k * 64;
        if (c < nch) {
            const half8 h = ld_global_h8(xr + c * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[k][i] = (float)h[i]; s += v[k][i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[k][i] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXCH; ++k) {
        const int c = lane + k * 64;
        if (c < nch) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = v[k][i] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    half_t* orow = out + (long)row * C;
#pragma unroll
    for (int k = 0; k < LN_MAXCH; ++k) {
        const int c = lane + k * 64;
        if (c < nch) {
            const half8 g = ld_global_h8(gamma + c * 8), b = ld_global_h8(beta + c * 8);
            half8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (half_t)((v[k][i] - mean) * rstd * (float)g[i] + (float)b[i]);
            *reinterpret_cast<half8*>(orow + c * 8) = o;
        }
    }
}

// Several rows per wave: LPR lanes share a row (8 at 320 / 640 channels, 16 at 1280), each lane holds up to ten 16-byte chunks
// of it and requests them all before anything else.  The one-row-per-wave form above keeps 40 of 64 lanes busy at 320 channels
// and has ONE load in flight per lane: 32 768 waves of two dependent round trips each -- the kernel was latency x occupancy
// (12.7 us for 42 MB at SD1.5's 64 x 64 level).  Same arithmetic (fp32, two passes over the registers), fixed reduction order.
constexpr int LNR_MAXCH = 10;
template <int LPR>
__global__ void __launch_bounds__(256)
layernorm_rows_kernel(const half_t* __restrict__ x, half_t* __restrict__ out,
                      const half_t* __restrict__ gamma, const half_t* __restrict__ beta,
                      int M, int C, float eps) {
    constexpr int RPW = 64 / LPR;                       // rows per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const int nk = (C >> 3) / LPR;                      // chunks per lane (the host checked: exact, <= LNR_MAXCH)
    const bool live = row < M;
    const half_t* xr = x + (long)(live ? row : 0) * C;
    half8 h[LNR_MAXCH], g[LNR_MAXCH], b[LNR_MAXCH];
#pragma unroll
    for (int k = 0; k < LNR_MAXCH; ++k)
        if (k < nk) h[k] = ld_global_h8(xr + (sub + k * LPR) * 8);
#pragma unroll
    for (int k = 0; k < LNR_MAXCH; ++k)
        if (k < nk) { g[k] = ld_global_h8(gamma + (sub + k * LPR) * 8); b[k] = ld_global_h8(beta + (sub + k * LPR) * 8); }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LNR_MAXCH; ++k)
        if (k < nk)
#pragma unroll
            for (int i = 0; i < 8; ++i) s += (float)h[k][i];
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LNR_MAXCH; ++k)
        if (k < nk)
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = (float)h[k][i] - mean; q += d * d; }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)C + eps);
    if (!live) return;
    half_t* orow = out + (long)row * C;
#pragma unroll
    for (int k = 0; k < LNR_MAXCH; ++k)
        if (k < nk) {
            half8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (half_t)(((float)h[k][i] - mean) * rstd * (float)g[k][i] + (float)b[k][i]);
            *reinterpret_cast<half8*>(orow + (sub + k * LPR) * 8) = o;
        }
}

// ------------------------------------------------------------------ GroupNorm
// pass 1: per (sample, row-chunk) block -> per-group (sum, sumsq) partials (deterministic, no atomics)
// pass 2: every block first folds the partials of its sample into per-channel scale/shift in LDS
//         (double accumulation), then streams y = x * scale + shift (+ SiLU)
constexpr int GN_ROWS_PER_BLOCK = 32;
constexpr int GN_MAXC = 2560;
constexpr int GN_MAXBLK = 512;   // partial blocks per sample: 32 rows each up to 128 x 128 tokens, more rows beyond

__global__ void __launch_bounds__(256)
gn_partial_kernel(const half_t* __restrict__ x1, const half_t* __restrict__ x2, int c1, int c2,
                  int HW, int groups, int rows_per_block, float* __restrict__ part /*[B][nblk][groups][2]*/) {
    // [phase][C] per-channel partials; phases * C <= 2560 by construction
    __shared__ float ssum[GN_MAXC], ssq[GN_MAXC];
    const int C = c1 + c2, nch = C >> 3;
    const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    const int r0 = blk * rows_per_block;
    const int r1 = min(HW, r0 + rows_per_block);
    // thread -> (chunk column, row phase): `phases` rows are processed concurrently
    const int phases = nch <= 256 ? 256 / nch : 1;
    for (int cc0 = 0; cc0 < nch; cc0 += 256) {          // one trip unless C > 2048
        const int t = threadIdx.x;
        const int col = nch <= 256 ? t % nch : cc0 + t;
        const int ph = nch <= 256 ? t / nch : 0;
        if (col < nch && ph < phases) {
            float s[8], q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { s[i] = 0.f; q[i] = 0.f; }
            const int ch0 = col * 8;
            const half_t* base; int ld, off;
            if (ch0 < c1) { base = x1; ld = c1; off = ch0; } else { base = x2; ld = c2; off = ch0 - c1; }
            const half_t* p = base + ((long)b * HW) * ld + off;
            int r = r0 + ph;
            // four independent loads in flight per thread
            for (; r + 7 * phases < r1; r += 8 * phases) {
                half8 h[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) h[u] = ld_global_h8(p + (long)(r + u * phases) * ld);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) { const float f = (float)h[u][i]; s[i] += f; q[i] += f * f; }
            }
            for (; r < r1; r += phases) {
                const half8 h = ld_global_h8(p + (long)r * ld);
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float f = (float)h[i]; s[i] += f; q[i] += f * f; }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { ssum[ph * C + ch0 + i] = s[i]; ssq[ph * C + ch0 + i] = q[i]; }
        }
    }
    __syncthreads();
    const int cg = C / groups;
    if ((int)threadIdx.x < groups) {
        float s = 0.f, q = 0.f;
        for (int ph = 0; ph < phases; ++ph)
            for (int c = threadIdx.x * cg; c < (threadIdx.x + 1) * cg; ++c) { s += ssum[ph * C + c]; q += ssq[ph * C + c]; }
        float* p = part + (((long)b * nblk + blk) * groups + threadIdx.x) * 2;
        p[0] = s; p[1] = q;
    }
}

// `part` != nullptr: per-(row block, group) partials of gn_partial_kernel.  Otherwise the statistics come from the GEMM
// epilogues that WROTE the input tensor(s) (cid_gemm_desc.gn_stats): st1 / st2 = fp32 [B * nb][32][2] per source, one
// (sum, sumsq) pair per block of tokens and unit of c / 32 channels; a group is a whole number of units of ONE source.
template <bool SILU>
__global__ void __launch_bounds__(256)
gn_apply_kernel(const half_t* __restrict__ x1, const half_t* __restrict__ x2, int c1, int c2,
                half_t* __restrict__ out, const float* __restrict__ part, int nblk, int groups, float eps,
                const half_t* __restrict__ gamma, const half_t* __restrict__ beta, int HW, int rows_per_block,
                const float* __restrict__ st1, int nb1, const float* __restrict__ st2, int nb2) {
    __shared__ float sscale[GN_MAXC], sshift[GN_MAXC];
    __shared__ float smean[64], srstd[64];
    __shared__ double dsum[256], dsq[256];
    const int C = c1 + c2, nch = C >> 3;
    const int b = blockIdx.y;
    const int cg = C / groups;
    if (part == nullptr) {
        const int slices = 256 / groups;
        const int g = threadIdx.x % groups, sl = threadIdx.x / groups;
        double s = 0.0, q = 0.0;
        if (sl < slices) {
            const bool first = g * cg < c1;
            const int u = (first ? c1 : c2) >> 5;                 // channels per unit of that source
            const int upg = cg / u;                               // units per group
            const int u0 = (first ? g * cg : g * cg - c1) / u;    // first unit of the group
            const int nb = first ? nb1 : nb2;
            const float* st = (first ? st1 : st2) + (long)b * nb * 64;
            for (int k = sl; k < nb * upg; k += slices) {
                const int blk = k / upg, uu = k - blk * upg;
                const float* p = st + ((long)blk * 32 + u0 + uu) * 2;
                s += (double)p[0]; q += (double)p[1];
            }
        }
        dsum[threadIdx.x] = s; dsq[threadIdx.x] = q;
    } else {
        // 256 threads fold the partials: thread -> (group, slice of the row blocks)
        const int slices = 256 / groups;
        const int g = threadIdx.x % groups, sl = threadIdx.x / groups;
        double s = 0.0, q = 0.0;
        if (sl < slices)
            for (int k = sl; k < nblk; k += slices) {
                const float* p = part + (((long)b * nblk + k) * groups + g) * 2;
                s += (double)p[0]; q += (double)p[1];
            }
        dsum[threadIdx.x] = s; dsq[threadIdx.x] = q;
    }
    __syncthreads();
    if ((int)threadIdx.x < groups) {
        const int slices = 256 / groups;
        double s = 0.0, q = 0.0;
        for (int sl = 0; sl < slices; ++sl) { s += dsum[sl * groups + threadIdx.x]; q += dsq[sl * groups + threadIdx.x]; }
        const double n = (double)HW * (double)cg;
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        smean[threadIdx.x] = (float)mean;
        srstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c / cg;
        const float sc = srstd[g] * (float)gamma[c];
        sscale[c] = sc;
        sshift[c] = (float)beta[c] - smean[g] * sc;
    }
    __syncthreads();
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(HW, r0 + rows_per_block);
    const long total = (long)(r1 - r0) * nch;
    for (long q = threadIdx.x; q < total; q += 256) {
        const int rr = (int)(q / nch);
        const int cc = (int)(q - (long)rr * nch);
        const long row = (long)b * HW + r0 + rr;
        const int ch0 = cc * 8;
        half8 h;
        if (ch0 < c1) h = ld_global_h8(x1 + row * c1 + ch0);
        else          h = ld_global_h8(x2 + row * c2 + (ch0 - c1));
        half8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float y = (float)h[i] * sscale[ch0 + i] + sshift[ch0 + i];
            if (SILU) y = silu_f(y);
            o[i] = (half_t)y;
        }
        *reinterpret_cast<half8*>(out + row * C + ch0) = o;
    }
}


// ------------------------------------------------------------------ GroupNorm, small tensors: ONE launch
// At the 8 x 8 / 16 x 16 levels a (sample, group) slice is at most 40 KB: one workgroup keeps it
// in registers -- read once, reduce, normalise, write -- instead of a statistics launch plus an apply launch that are
// both pure latency at that size (30 of the 61 GroupNorms of an SD1.5 forward).  Fixed reduction order (no atomics).
constexpr int GNS_MAXPER = 20;                       // 4-channel pieces per thread
constexpr long GNS_MAXELEM = 256L * GNS_MAXPER * 4;  // elements of a (sample, group) slice: 20480
#ifndef GNS_MAXHW
#define GNS_MAXHW 256                                // pixels per sample up to which the single launch is used
#endif

template <bool SILU>
__global__ void __launch_bounds__(256)
gn_small_kernel(const half_t* __restrict__ x1, const half_t* __restrict__ x2, int c1, int c2, half_t* __restrict__ out,
                const half_t* __restrict__ gamma, const half_t* __restrict__ beta, int B, int HW, int groups, float eps,
                int step_r, int step_c) {          // 256 / (cg / 4), 256 % (cg / 4): piece index -> (pixel, piece) without divisions
    __shared__ float wsum[4], wsq[4];
    __shared__ float sstat[2];
    const int C = c1 + c2, cg = C / groups, cq = cg >> 2;
    // workgroup -> (sample, group); with a multiple of 8 samples an XCD owns whole samples: the groups whose 40..160-byte
    // row pieces share cache lines meet in ONE L2 (consecutive workgroups go to consecutive XCDs)
    int b, g;
    {
        const int i = blockIdx.x;
        if ((B & 7) == 0) { const int j = i >> 3; b = (i & 7) + 8 * (j / groups); g = j % groups; }
        else { b = i / groups; g = i % groups; }
    }
    const int nchunk = HW * cq;
    const int r_first = (int)threadIdx.x / cq, c_first = (int)threadIdx.x - r_first * cq;
    half4 v[GNS_MAXPER];
    float s = 0.f, q = 0.f;
    int r = r_first, cc = c_first;
#pragma unroll
    for (int k = 0; k < GNS_MAXPER; ++k) {
        const int idx = threadIdx.x + k * 256;
        if (idx < nchunk) {
            const int ch0 = g * cg + 4 * cc;
            const long row = (long)b * HW + r;
            v[k] = ch0 < c1 ? *reinterpret_cast<const half4*>(x1 + row * c1 + ch0)
                            : *reinterpret_cast<const half4*>(x2 + row * c2 + (ch0 - c1));
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float f = (float)v[k][i]; s += f; q += f * f; }
        }
        r += step_r; cc += step_c;
        if (cc >= cq) { cc -= cq; ++r; }
    }
    s = wave_sum(s); q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = s; wsq[threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n = (double)HW * (double)cg;
        const double S = (double)wsum[0] + (double)wsum[1] + (double)wsum[2] + (double)wsum[3];
        const double Q = (double)wsq[0] + (double)wsq[1] + (double)wsq[2] + (double)wsq[3];
        const double mean = S / n;
        double var = Q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        sstat[0] = (float)mean;
        sstat[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const float mean = sstat[0], rstd = sstat[1];
    r = r_first; cc = c_first;
#pragma unroll
    for (int k = 0; k < GNS_MAXPER; ++k) {
        const int idx = threadIdx.x + k * 256;
        if (idx < nchunk) {
            const int ch0 = g * cg + 4 * cc;
            const half4 gm = *reinterpret_cast<const half4*>(gamma + ch0), bt = *reinterpret_cast<const half4*>(beta + ch0);
            half4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float sc = rstd * (float)gm[i];
                float y = (float)v[k][i] * sc + ((float)bt[i] - mean * sc);
                if (SILU) y = silu_f(y);
                o[i] = (half_t)y;
            }
            *reinterpret_cast<half4*>(out + ((long)b * HW + r) * C + ch0) = o;
        }
        r += step_r; cc += step_c;
        if (cc >= cq) { cc -= cq; ++r; }
    }
}

// ------------------------------------------------------------------ row softmax (VAE mid-block attention)
// in place over fp16 rows of base-2 logits (the q projection carries scale * log2 e): one wave per row,
// three L2-resident sweeps (max, sum, write) in fp32.
__global__ void __launch_bounds__(256)
softmax_rows_kernel(half_t* __restrict__ x, int rows, int cols, long ld) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    half_t* p = x + row * ld;
    const int nch = cols >> 3;
    float mx = -INFINITY;
    for (int c = lane; c < nch; c += 64) {
        const half8 h = ld_global_h8(p + c * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) mx = fmaxf(mx, (float)h[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float s = 0.f;
    for (int c = lane; c < nch; c += 64) {
        const half8 h = ld_global_h8(p + c * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += __builtin_amdgcn_exp2f((float)h[i] - mx);
    }
    const float inv = 1.f / wave_sum(s);
    for (int c = lane; c < nch; c += 64) {
        const half8 h = ld_global_h8(p + c * 8);
        half8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (half_t)(__builtin_amdgcn_exp2f((float)h[i] - mx) * inv);
        *reinterpret_cast<half8*>(p + c * 8) = o;
    }
}

// rows per partial block: 32 while that keeps a sample within GN_MAXBLK blocks (UNet levels), a multiple beyond (VAE)
inline int gn_rows(int HW) { return GN_ROWS_PER_BLOCK * ((HW + GN_ROWS_PER_BLOCK * GN_MAXBLK - 1) / (GN_ROWS_PER_BLOCK * GN_MAXBLK)); }
inline int gn_nblk(int HW) { return (HW + gn_rows(HW) - 1) / gn_rows(HW); }

}  // namespace

extern "C" int cid_layernorm_f16(const cid_half* x, cid_half* out, const cid_half* gamma, const cid_half* beta,
                                 int32_t M, int32_t C, float eps, cid_stream_t stream) {
    CID_CHECK_ARG(x && out && gamma && beta, "cid_layernorm_f16: null pointer");
    CID_CHECK_ARG(M > 0 && C > 0 && C % 8 == 0 && C <= 8 * 64 * LN_MAXCH, "cid_layernorm_f16: bad shape M=%d C=%d", M, C);
    const int nch = C >> 3;
    // several rows per wave only where the launch is large enough to fill the chip that way (measured: 12.2 -> 10.2 us at
    // 32768 x 320, but 8.2 -> 8.7 us at 8192 x 640 and 2048 x 1280, where one row per wave already is one round of waves)
    const bool big = (long)M * C >= (8L << 20);
    if (!big)
        hipLaunchKernelGGL(layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                           (const half_t*)x, (half_t*)out, (const half_t*)gamma, (const half_t*)beta, M, C, eps);
    else if (nch % 8 == 0 && nch / 8 <= LNR_MAXCH)
        hipLaunchKernelGGL(layernorm_rows_kernel<8>, dim3((M + 31) / 32), dim3(256), 0, (hipStream_t)stream,
                           (const half_t*)x, (half_t*)out, (const half_t*)gamma, (const half_t*)beta, M, C, eps);
    else if (nch % 16 == 0 && nch / 16 <= LNR_MAXCH)
        hipLaunchKernelGGL(layernorm_rows_kernel<16>, dim3((M + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                           (const half_t*)x, (half_t*)out, (const half_t*)gamma, (const half_t*)beta, M, C, eps);
    else
        hipLaunchKernelGGL(layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                           (const half_t*)x, (half_t*)out, (const half_t*)gamma, (const half_t*)beta, M, C, eps);
    CID_CHECK_LAUNCH("cid_layernorm_f16");
    return 0;
}


extern "C" int cid_softmax_rows_f16(cid_half* x, int32_t rows, int32_t cols, int64_t ld, cid_stream_t stream) {
    CID_CHECK_ARG(x && rows > 0 && cols > 0 && cols % 8 == 0 && ld >= cols && ld % 8 == 0, "cid_softmax_rows_f16: bad shape");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (half_t*)x, rows, cols, (long)ld);
    CID_CHECK_LAUNCH("cid_softmax_rows_f16");
    return 0;
}

extern "C" int64_t cid_groupnorm_ws_bytes(int32_t B, int32_t C) {
    (void)C;
    return (int64_t)B * GN_MAXBLK * 64 * 2 * 4;   // at most GN_MAXBLK partial blocks per sample, 64 groups
}

extern "C" int cid_groupnorm_f16(const cid_half* x1, const cid_half* x2, int32_t c1, int32_t c2,
                                 cid_half* out, const cid_half* gamma, const cid_half* beta,
                                 int32_t B, int32_t HW, int32_t groups, float eps, int32_t silu,
                                 void* ws, cid_stream_t stream) {
    CID_CHECK_ARG(x1 && out && gamma && beta && ws, "cid_groupnorm_f16: null pointer");
    const int C = c1 + c2;
    CID_CHECK_ARG(c1 > 0 && c1 % 8 == 0 && c2 >= 0 && c2 % 8 == 0 && (c2 == 0 || x2), "cid_groupnorm_f16: bad channels");
    CID_CHECK_ARG(groups > 0 && groups <= 64 && C % groups == 0 && C <= GN_MAXC, "cid_groupnorm_f16: bad groups/C");
    CID_CHECK_ARG(B > 0 && HW > 0 && HW <= 1024 * 1024, "cid_groupnorm_f16: bad B/HW");
    const int nblk = gn_nblk(HW);
    float* part = (float*)ws;
    hipStream_t s = (hipStream_t)stream;
    const int cg = C / groups;
    // a (sample, group) slice fits one workgroup's registers; measured faster than the two launches up to 16 x 16 pixels
    // (tools/kbench.py --only norm: 20.2 -> 18 us at 256 pixels x 2560 channels, 12.1 -> 9.5 us at 64 x 1280; slower at 1024 x 640)
    if (cg % 4 == 0 && HW <= GNS_MAXHW && (long)HW * cg <= GNS_MAXELEM) {
        const int cq = cg / 4;
        if (silu)
            hipLaunchKernelGGL(gn_small_kernel<true>, dim3(groups * B), dim3(256), 0, s, (const half_t*)x1, (const half_t*)x2, c1, c2,
                               (half_t*)out, (const half_t*)gamma, (const half_t*)beta, B, HW, groups, eps, 256 / cq, 256 % cq);
        else
            hipLaunchKernelGGL(gn_small_kernel<false>, dim3(groups * B), dim3(256), 0, s, (const half_t*)x1, (const half_t*)x2, c1, c2,
                               (half_t*)out, (const half_t*)gamma, (const half_t*)beta, B, HW, groups, eps, 256 / cq, 256 % cq);
        CID_CHECK_LAUNCH("cid_groupnorm_f16");
        return 0;
    }
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nblk, B), dim3(256), 0, s,
                       (const half_t*)x1, (const half_t*)x2, c1, c2, HW, groups, gn_rows(HW), part);
    // apply: ~16 KB of activations per block
    int rpb = (16384 / (C * 2)) > 0 ? 16384 / (C * 2) : 1;
    if ((HW + rpb - 1) / rpb > 1024) rpb = (HW + 1023) / 1024;   // every apply block re-folds the partials: bound their number
    if (rpb > HW) rpb = HW;
    const int ablk = (HW + rpb - 1) / rpb;
    if (silu)
        hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(ablk, B), dim3(256), 0, s, (const half_t*)x1, (const half_t*)x2,
                           c1, c2, (half_t*)out, part, nblk, groups, eps, (const half_t*)gamma, (const half_t*)beta, HW, rpb,
                           (const float*)nullptr, 0, (const float*)nullptr, 0);
    else
        hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(ablk, B), dim3(256), 0, s, (const half_t*)x1, (const half_t*)x2,
                           c1, c2, (half_t*)out, part, nblk, groups, eps, (const half_t*)gamma, (const half_t*)beta, HW, rpb,
                           (const float*)nullptr, 0, (const float*)nullptr, 0);
    CID_CHECK_LAUNCH("cid_groupnorm_f16");
    return 0;
}

extern "C" int cid_groupnorm_stats_ok(int32_t c1, int32_t c2, int32_t groups) {
    // every group must be a whole number of statistics units (c / 32 channels) of ONE source
    const int C = c1 + c2;
    if (groups <= 0 || groups > 64 || C % groups || c1 <= 0 || c1 % 32 || c2 < 0 || c2 % 32) return 0;
    const int cg = C / groups;
    if (c1 % cg) return 0;
    if (cg % (c1 / 32)) return 0;
    if (c2 && cg % (c2 / 32)) return 0;
    return 1;
}

extern "C" int cid_groupnorm_stats_f16(const cid_half* x1, const cid_half* x2, int32_t c1, int32_t c2,
                                       cid_half* out, const cid_half* gamma, const cid_half* beta,
                                       int32_t B, int32_t HW, int32_t groups, float eps, int32_t silu,
                                       const float* stats1, int32_t rows1, const float* stats2, int32_t rows2,
                                       cid_stream_t stream) {
    CID_CHECK_ARG(x1 && out && gamma && beta && stats1, "cid_groupnorm_stats_f16: null pointer");
    const int C = c1 + c2;
    CID_CHECK_ARG(c1 > 0 && c1 % 8 == 0 && c2 >= 0 && c2 % 8 == 0 && (c2 == 0 || (x2 && stats2)), "cid_groupnorm_stats_f16: bad channels");
    CID_CHECK_ARG(C <= GN_MAXC && cid_groupnorm_stats_ok(c1, c2, groups),
                  "cid_groupnorm_stats_f16: groups of %d channels are not whole statistics units of one source (c1=%d c2=%d)",
                  groups ? C / groups : 0, c1, c2);
    CID_CHECK_ARG(B > 0 && HW > 0 && rows1 > 0 && HW % rows1 == 0 && (c2 == 0 || (rows2 > 0 && HW % rows2 == 0)),
                  "cid_groupnorm_stats_f16: a statistics block must lie inside one sample (HW=%d rows %d / %d)", HW, rows1, rows2);
    hipStream_t s = (hipStream_t)stream;
    int rpb = (16384 / (C * 2)) > 0 ? 16384 / (C * 2) : 1;
    if ((HW + rpb - 1) / rpb > 1024) rpb = (HW + 1023) / 1024;
    if (rpb > HW) rpb = HW;
    const int ablk = (HW + rpb - 1) / rpb;
    const int nb1 = HW / rows1, nb2 = c2 ? HW / rows2 : 0;
    if (silu)
        hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(ablk, B), dim3(256), 0, s, (const half_t*)x1, (const half_t*)x2,
                           c1, c2, (half_t*)out, (const float*)nullptr, 0, groups, eps, (const half_t*)gamma, (const half_t*)beta,
                           HW, rpb, stats1, nb1, stats2, nb2);
    else
        hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(ablk, B), dim3(256), 0, s, (const half_t*)x1, (const half_t*)x2,
                           c1, c2, (half_t*)out, (const float*)nullptr, 0, groups, eps, (const half_t*)gamma, (const half_t*)beta,
                           HW, rpb, stats1, nb1, stats2, nb2);
    CID_CHECK_LAUNCH("cid_groupnorm_stats_f16");
    return 0;
}
