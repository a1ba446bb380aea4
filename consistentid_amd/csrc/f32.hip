// fp32 kernels for the SDXL VAE decode: the reference upcasts that VAE to float32 before decoding
// (/root/reference/pipline_StableDiffusionXL_ConsistentID.py:670-676: `self.upcast_vae()`, then
// `self.vae.decode(latents / scaling_factor)`), because its activations leave the fp16 range.  Once per image, not
// per denoising step: built for exactness and simplicity on the f32-input matrix cores, not tuned to their roofline.
//
//   cid_gemm_f32           implicit GEMM on v_mfma_f32_32x32x2_f32 (exact f32: a k-ordered fmaf chain, 157 TF/s peak):
//                          Linear / 1x1 / 3x3 conv (pad 1) with fused nearest-2x upsample, bias, residual
//   cid_groupnorm_f32      GroupNorm (+SiLU): per-channel partial sums -> per-group statistics -> streaming apply
//   cid_softmax_rows_f32   in-place row softmax of base-2 logits (the mid block's single-head attention)
// Roofline: the GEMM is MFMA(f32)-bound (2 M N K flop), the two others HBM-bound (GroupNorm: 2 reads + 1 write of the
// tensor; softmax: 3 reads + 1 write of the score matrix).
#include "common.h"
#include "../../include/cid.h"

namespace {

constexpr int G_BM = 64, G_BN = 64, G_BK = 16, G_PAD = 4;

struct Gemm32Args {
    const float* x; const float* w; const float* bias; const float* res; float* out;
    int M, N, c, taps, ldx, ldo, ldr;
    int Hi, Wi, Ho, Wo, up;
};

// out[m][n] = sum_k A(m, k) W[n][k] + bias[n] + res[m][n],  k = tap * c + ch,
// A(m, k) = x[pixel(m, tap)][ch] of the (optionally 2x nearest-upsampled) input, zero outside the image
__global__ void __launch_bounds__(256)
gemm_f32_kernel(Gemm32Args a) {
    __shared__ float As[G_BK][G_BM + G_PAD], Bs[G_BK][G_BN + G_PAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * G_BM, n0 = blockIdx.y * G_BN;
    const int K = a.taps * a.c;
    // loader role: thread -> (row, 4 consecutive k)
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    // the output pixel of this thread's A row
    const int m = m0 + lrow;
    int pb = 0, py = 0, px = 0;
    if (a.taps == 9 && m < a.M) {
        const int hw = a.Ho * a.Wo;
        pb = m / hw;
        const int r = m - pb * hw;
        py = r / a.Wo;
        px = r - py * a.Wo;
    }
    f32x16 acc = zero_f16v();
    const bool vec = (a.c & 3) == 0;
    for (int k0 = 0; k0 < K; k0 += G_BK) {
        float av[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
        // ---- A: gather
        if (m < a.M) {
            if (vec) {
                const int k = k0 + lk;
                if (k < K) {
                    const int tap = k / a.c, ch = k - tap * a.c;
                    long row = m;
                    bool ok = true;
                    if (a.taps == 9) {
                        int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
                        ok = yy >= 0 && yy < a.Ho && xx >= 0 && xx < a.Wo;
                        if (a.up) { yy >>= 1; xx >>= 1; }
                        row = ((long)pb * a.Hi + yy) * a.Wi + xx;
                    }
                    if (ok) {
                        const float4 v = *reinterpret_cast<const float4*>(a.x + row * a.ldx + ch);
                        av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + lk + j;
                    if (k < K) {
                        const int tap = k / a.c, ch = k - tap * a.c;
                        long row = m;
                        bool ok = true;
                        if (a.taps == 9) {
                            int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
                            ok = yy >= 0 && yy < a.Ho && xx >= 0 && xx < a.Wo;
                            if (a.up) { yy >>= 1; xx >>= 1; }
                            row = ((long)pb * a.Hi + yy) * a.Wi + xx;
                        }
                        if (ok) av[j] = a.x[row * a.ldx + ch];
                    }
                }
            }
        }
        // ---- W rows
        const int n = n0 + lrow;
        if (n < a.N) {
            if (vec && k0 + lk < K) {
                const float4 v = *reinterpret_cast<const float4*>(a.w + (long)n * K + k0 + lk);
                bv[0] = v.x; bv[1] = v.y; bv[2] = v.z; bv[3] = v.w;
            } else if (!vec) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k0 + lk + j < K) bv[j] = a.w[(long)n * K + k0 + lk + j];
            }
        }
        __syncthreads();            // the previous step's fragments have been consumed
#pragma unroll
        for (int j = 0; j < 4; ++j) { As[lk + j][lrow] = av[j]; Bs[lk + j][lrow] = bv[j]; }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < G_BK / 2; ++s) {
            const float fa = As[2 * s + (lane >> 5)][wm * 32 + (lane & 31)];
            const float fb = Bs[2 * s + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        }
    }
    // D layout: column (n) = lane & 31, row (m) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int n = n0 + wn * 32 + (lane & 31);
    if (n < a.N) {
        const float b = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = m0 + wm * 32 + crow(r, lane >> 5);
            if (mm < a.M) {
                float v = acc[r] + b;
                if (a.res) v += a.res[(long)mm * a.ldr + n];
                a.out[(long)mm * a.ldo + n] = v;
            }
        }
    }
}

// ---------------------------------------------------------------- GroupNorm (fp32)
// rows per statistics block: 64, more for large token maps so that a sample has at most 256 partial blocks
inline int gn32_rows(int HW) { const int r = (HW + 255) / 256; return r < 64 ? 64 : r; }

__global__ void __launch_bounds__(256)
gn32_partial_kernel(const float* __restrict__ x, int HW, int C, int rows, float* __restrict__ part /*[B][nblk][C][2]*/) {
    const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    const int r0 = blk * rows, r1 = min(HW, r0 + rows);
    for (int c = threadIdx.x; c < C; c += 256) {            // a thread owns a channel: consecutive threads, consecutive addresses
        float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
        const float* p = x + ((long)b * HW + r0) * C + c;
        int r = r0;
        for (; r + 3 < r1; r += 4, p += 4 * (long)C) {       // four independent loads in flight
            const float v0 = p[0], v1 = p[C], v2 = p[2 * (long)C], v3 = p[3 * (long)C];
            s[0] += v0; q[0] += v0 * v0; s[1] += v1; q[1] += v1 * v1; s[2] += v2; q[2] += v2 * v2; s[3] += v3; q[3] += v3 * v3;
        }
        for (; r < r1; ++r, p += C) { const float v = *p; s[0] += v; q[0] += v * v; }
        float* o = part + (((long)b * nblk + blk) * C + c) * 2;
        o[0] = (s[0] + s[1]) + (s[2] + s[3]); o[1] = (q[0] + q[1]) + (q[2] + q[3]);
    }
}

__global__ void __launch_bounds__(256)
gn32_finalize_kernel(const float* __restrict__ part, int nblk, int HW, int C, int groups, float eps,
                     const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ ss /*[B][C][2]*/) {
    __shared__ double gs[64], gq[64];
    __shared__ double cs[2048], cq[2048];
    const int b = blockIdx.x, cg = C / groups;
    for (int c = threadIdx.x; c < C; c += 256) {
        double s = 0.0, q = 0.0;
        for (int k = 0; k < nblk; ++k) {
            const float* p = part + (((long)b * nblk + k) * C + c) * 2;
            s += (double)p[0]; q += (double)p[1];
        }
        cs[c] = s; cq[c] = q;
    }
    __syncthreads();
    if ((int)threadIdx.x < groups) {
        double s = 0.0, q = 0.0;
        for (int c = threadIdx.x * cg; c < (threadIdx.x + 1) * cg; ++c) { s += cs[c]; q += cq[c]; }
        const double n = (double)HW * cg, mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        gs[threadIdx.x] = mean;
        gq[threadIdx.x] = 1.0 / sqrt(var + (double)eps);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c / cg;
        const float sc = (float)gq[g] * gamma[c];
        ss[((long)b * C + c) * 2 + 0] = sc;
        ss[((long)b * C + c) * 2 + 1] = beta[c] - (float)gs[g] * sc;
    }
}

__global__ void __launch_bounds__(256)
gn32_apply_kernel(const float* __restrict__ x, float* __restrict__ out, const float* __restrict__ ss, long per_sample, int C,
                  int silu) {
    const int b = blockIdx.y;
    const float* xb = x + (long)b * per_sample;
    float* ob = out + (long)b * per_sample;
    const float* sb = ss + (long)b * C * 2;
    for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < per_sample; e += (long)gridDim.x * 1024) {
        const float4 v = *reinterpret_cast<const float4*>(xb + e);
        const int c = (int)(e % C);                          // C % 4 == 0: the four values share a row
        float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float y = r[i] * sb[(c + i) * 2] + sb[(c + i) * 2 + 1];
            if (silu) y = y / (1.f + __expf(-y));
            r[i] = y;
        }
        *reinterpret_cast<float4*>(ob + e) = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// ---------------------------------------------------------------- row softmax (fp32, base-2 logits, in place)
__global__ void __launch_bounds__(256)
softmax_rows_f32_kernel(float* __restrict__ x, int cols, long ld) {
    __shared__ float red[4];
    float* p = x + (long)blockIdx.x * ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, p[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) s += __builtin_amdgcn_exp2f(p[c] - mx);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
    for (int c = threadIdx.x; c < cols; c += 256) p[c] = __builtin_amdgcn_exp2f(p[c] - mx) * inv;
}

}  // namespace

extern "C" int cid_gemm_f32(const float* x, const float* w, const float* bias, const float* res, float* out,
                            int32_t M, int32_t N, int32_t c, int32_t taps, int32_t ldx, int32_t ldo, int32_t ldr,
                            int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t up, cid_stream_t stream) {
    CID_CHECK_ARG(x && w && out, "cid_gemm_f32: null pointer");
    CID_CHECK_ARG(M > 0 && N > 0 && c > 0 && (taps == 1 || taps == 9) && ldx >= c && ldo >= N, "cid_gemm_f32: bad shape");
    CID_CHECK_ARG((c & 3) != 0 || (ldx & 3) == 0, "cid_gemm_f32: ldx must be a multiple of 4 when c is");
    CID_CHECK_ARG(!res || ldr >= N, "cid_gemm_f32: bad residual pitch");
    if (taps == 9) {
        CID_CHECK_ARG(Hi > 0 && Wi > 0 && Ho == (Hi << (up ? 1 : 0)) && Wo == (Wi << (up ? 1 : 0)) && M % (Ho * Wo) == 0,
                      "cid_gemm_f32: 3x3 geometry (stride 1, pad 1, optional nearest-2x upsample)");
    }
    Gemm32Args a{x, w, bias, res, out, M, N, c, taps, ldx, ldo, ldr, Hi, Wi, Ho, Wo, up ? 1 : 0};
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((M + G_BM - 1) / G_BM, (N + G_BN - 1) / G_BN), dim3(256), 0, (hipStream_t)stream, a);
    CID_CHECK_LAUNCH("cid_gemm_f32");
    return 0;
}

extern "C" int64_t cid_groupnorm_f32_ws_bytes(int32_t B, int32_t HW, int32_t C) {
    const int64_t rows = gn32_rows(HW), nblk = (HW + rows - 1) / rows;
    return ((int64_t)B * nblk * C * 2 + (int64_t)B * C * 2) * 4;
}

extern "C" int cid_groupnorm_f32(const float* x, float* out, const float* gamma, const float* beta, int32_t B, int32_t HW,
                                 int32_t C, int32_t groups, float eps, int32_t silu, void* ws, cid_stream_t stream) {
    CID_CHECK_ARG(x && out && gamma && beta && ws, "cid_groupnorm_f32: null pointer");
    CID_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 2048 && groups > 0 && groups <= 64 && C % groups == 0,
                  "cid_groupnorm_f32: bad shape");
    const int rows = gn32_rows(HW), nblk = (HW + rows - 1) / rows;
    float* part = (float*)ws;
    float* ss = part + (size_t)B * nblk * C * 2;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gn32_partial_kernel, dim3(nblk, B), dim3(256), 0, s, x, HW, C, rows, part);
    hipLaunchKernelGGL(gn32_finalize_kernel, dim3(B), dim3(256), 0, s, (const float*)part, nblk, HW, C, groups, eps, gamma, beta, ss);
    const long per_sample = (long)HW * C;
    long blocks = (per_sample / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gn32_apply_kernel, dim3((int)blocks, B), dim3(256), 0, s, x, out, (const float*)ss, per_sample, C, silu);
    CID_CHECK_LAUNCH("cid_groupnorm_f32");
    return 0;
}

extern "C" int cid_softmax_rows_f32(float* x, int32_t rows, int32_t cols, int64_t ld, cid_stream_t stream) {
    CID_CHECK_ARG(x && rows > 0 && cols > 0 && ld >= cols, "cid_softmax_rows_f32: bad shape");
    hipLaunchKernelGGL(softmax_rows_f32_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, cols, (long)ld);
    CID_CHECK_LAUNCH("cid_softmax_rows_f32");
    return 0;
}
