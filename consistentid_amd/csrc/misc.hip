// HBM-/latency-bound ends of the UNet step: conv_in, conv_out, timestep embedding,
// the tiny-M linears of the time path, CFG + DDIM update, residual adds.
// None of these is matrix-core work: they are vectorised (16 B per lane) streaming
// kernels; algorithmic bytes = tensors read once + written once.
#include "common.h"
#include "../../include/cid.h"
#include <stdarg.h>

// ---------------------------------------------------------------- error plumbing
static thread_local char g_err[512] = "";
void cid_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* cid_last_error(void) { return g_err; }
extern "C" int cid_version(void) { return 100; }

namespace {

// ---------------------------------------------------------------- conv_in
// sample NCHW [Bin][cin][H][W] -> token-major [B][H*W][cout], 3x3 pad 1 (diffusers UNet2DConditionModel.conv_in as called from
// /root/reference/pipline_StableDiffusion_ConsistentID.py:552-557 on torch.cat([latents] * 2), :537-539).
// HBM-bound on its output (8 input channels, 320 output channels).  Eight threads share a pixel: each gathers the pixel's
// 9 x cin inputs ONCE as packed half pairs and walks every eighth octet of output channels with v_dot2_f32_f16 against
// weights staged in LDS as [k pair][cout] half2 (read in the weights' own [cout][K] order, transposed on the LDS side).  Batch
// row b reads sample b % Bin: the rows that share a sample are COMPUTED once and stored B / Bin times (the CFG duplication).
// (Rounds 1-5: one thread per (pixel, channel octet) re-gathering the 36 inputs with 2-byte loads and converting every
//  weight it touched -- 50 us at the SD1.5 level-0 shape for 10 MB of output.)
constexpr int CIN_MAXKP = 42;    // half pairs of the contraction: 9 taps x cin1 and 9 taps x cin2, each rounded up to a pair
constexpr int CIN_MAXCO = 320;
constexpr int CIN_TPP = 8;       // threads per pixel

// C1 / C2 > 0: the channel split is a compile-time constant (4 + 0 plain, 9 + 0 and 4 + 5 inpainting): every (tap, channel) of the
// gather is then known at compile time and its loads are unconditional (coordinates clamped, the value zeroed outside the image),
// so that all of a pixel's inputs are in flight together.  C1 = 0: any split <= 9, runtime divisions (slow, correct).
template <int C1, int C2>
__global__ void __launch_bounds__(256)
conv_in_kernel(const half_t* __restrict__ sample, const half_t* __restrict__ extra, half_t* __restrict__ out,
               const half_t* __restrict__ w, const half_t* __restrict__ bias, int B, int Bin, int cin1_, int cin2_, int H, int W,
               int cout, const float* __restrict__ in_scale) {
    const int cin1 = C1 > 0 ? C1 : cin1_, cin2 = C1 > 0 ? C2 : cin2_;
    // channels [0, cin1) come from `sample` (the latents, scaled by in_scale), channels [cin1, cin1 + cin2) from `extra`
    // (mask | masked-image latents of a 9-channel inpainting UNet, inpaint ref :320-321 -- concatenated AFTER
    // scale_model_input there, so they are not scaled); both NCHW, batch row b reads sample b % Bin of either
    const int cin = cin1 + cin2;
    // scheduler.scale_model_input (ref :540): a scalar on the latents, read from device memory so one captured graph
    // serves every step; the convolution is linear, so it is applied to the latent part's tap sum (fp32)
    const float isc = in_scale ? in_scale[0] : 1.f;
    __shared__ half2v wl[CIN_MAXKP * CIN_MAXCO];           // [k pair][cout]: part 1 = (tap, ci < cin1), then part 2
    const int K = 9 * cin, K1 = 9 * cin1, K2 = 9 * cin2;
    const int KP1 = (K1 + 1) >> 1, KP2 = (K2 + 1) >> 1;
    half_t* wl1 = reinterpret_cast<half_t*>(wl);
    // staging: thread co copies weight row co ([tap][ci], 2 K bytes) -- its loads are independent and in flight together, its
    // 2-byte LDS stores land 4 bytes from the neighbouring thread's (conflict free).  (An element-per-thread transpose walked
    // K * cout / 256 = 45 dependent load -> store rounds with two runtime divisions each: 50 of this kernel's 70 us.)
    for (int co = threadIdx.x; co < cout; co += 256) {
        const half_t* wr = w + (long)co * K;
        int tap = 0, ci = 0;
        for (int k = 0; k < K; ++k) {
            const int kk = ci < cin1 ? tap * cin1 + ci : 2 * KP1 + tap * cin2 + (ci - cin1);   // position in the reordered contraction
            wl1[((kk >> 1) * cout + co) * 2 + (kk & 1)] = wr[k];
            if (++ci == cin) { ci = 0; ++tap; }
        }
        if (K1 & 1) wl1[((KP1 - 1) * cout + co) * 2 + 1] = (half_t)0.f;                         // the odd pairs' second halves
        if (K2 & 1) wl1[((KP1 + KP2 - 1) * cout + co) * 2 + 1] = (half_t)0.f;
    }
    __syncthreads();
    const int nco = cout >> 3;
    const int HW = H * W, rep = B / Bin;
    const long npix = (long)Bin * HW;
    const int sub = threadIdx.x & (CIN_TPP - 1);
    for (long p = (long)blockIdx.x * (256 / CIN_TPP) + (threadIdx.x / CIN_TPP); p < npix; p += (long)gridDim.x * (256 / CIN_TPP)) {
        const int b = (int)(p / HW);
        const int rem = (int)(p - (long)b * HW);
        const int y = rem / W, x = rem - y * W;
        const half_t* src = sample + (long)b * cin1 * HW;
        const half_t* src2 = extra + (long)b * cin2 * HW;   // not dereferenced when cin2 == 0
        half2v in[CIN_MAXKP];
#pragma unroll
        for (int kp = 0; kp < CIN_MAXKP; ++kp) {
            half_t v[2] = {(half_t)0.f, (half_t)0.f};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int kk = 2 * kp + h;
                const bool part2 = kk >= 2 * KP1;
                if (part2) kk -= 2 * KP1;
                const int cpart = part2 ? cin2 : cin1;
                const bool live = kp < KP1 + KP2 && kk < 9 * cpart;
                if (C1 > 0 && !live) continue;                  // (compile-time in the specialised instances)
                const int tap = live ? kk / cpart : 0, ci = live ? kk - tap * cpart : 0;
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                const bool ok = live && yy >= 0 && yy < H && xx >= 0 && xx < W;
                const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                const half_t* sp = (part2 && cin2 > 0) ? src2 : src;
                const half_t t = sp[((long)ci * H + yc) * W + xc];      // always a valid address: no branch around the load
                v[h] = ok ? t : (half_t)0.f;
            }
            in[kp] = half2v{v[0], v[1]};
        }
        for (int cc = sub; cc < nco; cc += CIN_TPP) {
            float a1[8], a2[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { a1[i] = 0.f; a2[i] = 0.f; }
#pragma unroll
            for (int kp = 0; kp < CIN_MAXKP; ++kp) {
                if (kp < KP1 + KP2) {
                    const half2v* wp = wl + kp * cout + cc * 8;
                    const bool p2 = kp >= KP1;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (p2) a2[i] = __builtin_amdgcn_fdot2(in[kp], wp[i], a2[i], false);
                        else a1[i] = __builtin_amdgcn_fdot2(in[kp], wp[i], a1[i], false);
                    }
                }
            }
            const half8 bb = ld_global_h8(bias + cc * 8);
            half8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (half_t)(a1[i] * isc + a2[i] + (float)bb[i]);
            for (int r = 0; r < rep; ++r)
                *reinterpret_cast<half8*>(out + ((long)(b + r * Bin) * HW + rem) * cout + cc * 8) = o;
        }
    }
}


// ---------------------------------------------------------------- small 3x3 conv (ControlNet condition embedding)
// token-major [B][Hi*Wi][cin] -> token-major [B][Ho*Wo][cout], pad 1, stride 1 or 2, optional SiLU.
// Channel counts below the MFMA kernel's 64-multiple rule (3, 16, 32, 96): runs once per generation (the
// control image is constant over the denoising steps), so a plain fp32-accumulating direct conv is enough.
__global__ void __launch_bounds__(256)
conv3x3_small_kernel(const half_t* __restrict__ x, half_t* __restrict__ out, const half_t* __restrict__ w,
                     const half_t* __restrict__ bias, int B, int Hi, int Wi, int Ho, int Wo, int cin, int cout,
                     int stride, int silu) {
    const int nco = cout / 8;
    const long total = (long)B * Ho * Wo * nco;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long pix = q / nco;
        const int cc = (int)(q - pix * nco);
        const int b = (int)(pix / (Ho * Wo));
        const int rem = (int)(pix - (long)b * Ho * Wo);
        const int y = rem / Wo, xo = rem - y * Wo;
        float acc[8];
        const half8 bb = ld_global_h8(bias + cc * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (float)bb[i];
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y * stride + tap / 3 - 1, xx = xo * stride + tap % 3 - 1;
            if (yy < 0 || yy >= Hi || xx < 0 || xx >= Wi) continue;
            const half_t* xp = x + (((long)b * Hi + yy) * Wi + xx) * cin;
            const half_t* wp = w + ((long)(cc * 8) * 9 + tap) * cin;
            if ((cin & 7) == 0) {
                for (int ci = 0; ci < cin; ci += 8) {
                    const half8 xv = ld_global_h8(xp + ci);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const half8 wv = ld_global_h8(wp + (long)i * 9 * cin + ci);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[i] += (float)xv[j] * (float)wv[j];
                    }
                }
            } else {
                for (int ci = 0; ci < cin; ++ci) {
                    const float v = (float)xp[ci];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += v * (float)wp[(long)i * 9 * cin + ci];
                }
            }
        }
        half8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (half_t)(silu ? silu_f(acc[i]) : acc[i]);
        *reinterpret_cast<half8*>(out + pix * cout + cc * 8) = o;
    }
}

// ---------------------------------------------------------------- identity-conditioning stack helpers (once per image)
// exact-erf GELU in place (nn.GELU() of functions.py:395, :499 and attention.py:58)
__global__ void __launch_bounds__(256)
gelu_kernel(half_t* __restrict__ x, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        half8 h = ld_global_h8(x + i * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) h[k] = (half_t)gelu_erf_f((float)h[k]);
        *reinterpret_cast<half8*>(x + i * 8) = h;
    }
}

// PerceiverAttention core (functions.py:439-447): a handful of latent queries attend to [image tokens ; latents].
// One wave per (sample, head, query); head width 64 (dim_head is 64 throughout the reference).  Keys live in two row
// blocks (the torch.cat((x, latents)) of :433 without the copy); a row holds [K | V] (to_kv(...).chunk(2), :434).
constexpr int SA_MAXK = 16;     // 64 lanes x 16 = up to 1024 keys
__global__ void __launch_bounds__(256)
small_attn_kernel(const half_t* __restrict__ q, int ldq, const half_t* __restrict__ kv1, int n1,
                  const half_t* __restrict__ kv2, int n2, int ldkv, half_t* __restrict__ out, int ldo,
                  int Lq, int heads, float scale2, int total) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= total) return;
    const int qi = w % Lq, h = (w / Lq) % heads, b = w / (Lq * heads);
    const int inner = heads * 64, n = n1 + n2;
    float qv[64];
    {
        const half_t* qp = q + (long)(b * Lq + qi) * ldq + h * 64;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const half8 t = ld_global_h8(qp + c * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) qv[c * 8 + i] = (float)t[i];
        }
    }
    auto row = [&](int j) -> const half_t* {
        return j < n1 ? kv1 + (long)(b * n1 + j) * ldkv : kv2 + (long)(b * n2 + (j - n1)) * ldkv;
    };
    float sc[SA_MAXK];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < SA_MAXK; ++t) {
        const int j = t * 64 + lane;
        float s = -INFINITY;
        if (j < n) {
            const half_t* kp = row(j) + h * 64;
            s = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const half8 kk = ld_global_h8(kp + c * 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) s += qv[c * 8 + i] * (float)kk[i];
            }
            s *= scale2;
        }
        sc[t] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < SA_MAXK; ++t) { sc[t] = (t * 64 + lane < n) ? __expf(sc[t] - mx) : 0.f; sum += sc[t]; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    // out[c = lane] = sum_j p_j V[j][c]; p_j is fetched from its owner lane
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < SA_MAXK; ++t) {
        if (t * 64 >= n) break;
        for (int src = 0; src < 64; ++src) {
            const int j = t * 64 + src;
            if (j >= n) break;
            const float p = __shfl(sc[t], src, 64);
            acc += p * (float)row(j)[inner + h * 64 + lane];
        }
    }
    out[(long)(b * Lq + qi) * ldo + h * 64 + lane] = (half_t)(acc * inv);
}

// ---------------------------------------------------------------- conv_out
// token-major [B][H*W][cin] -> NCHW [B][cout<=4][H][W], 3x3 pad 1 (diffusers UNet2DConditionModel.conv_out behind
// conv_norm_out + SiLU, as called from /root/reference/pipline_StableDiffusion_ConsistentID.py:552-557).
// The activation is read once from HBM; the nine taps of a pixel hit in L2 -- 190 MB of L2 -> L1 traffic at the SD1.5 level-0
// shape, which is what its 27 us are (7 TB/s; unconditional clamped loads made it 38 us: more bytes, not fewer round trips).  Eight lanes
// share a pixel pair (each lane every eighth 16-byte channel chunk), the weights [9][cin / 8][4 outputs] sit in LDS and one
// read of a (tap, chunk) serves both pixels; v_dot2_f32_f16 accumulates in fp32; the eight partial sums meet by DPP-style
// shuffles.  (The first version ran one wave per pixel with the weights re-read from L1 per wave and four 64-lane
// reductions per pixel: 63.6 us at the SD1.5 level-0 shape, 0.33 TB/s.)
constexpr int CO_PPL = 2;        // pixels per lane
constexpr int CO_PPW = 8 * CO_PPL;   // pixels per wave

CID_DEVINL float dot8(half8 a, half8 b, float c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const half2v x = {a[2 * j], a[2 * j + 1]}, y = {b[2 * j], b[2 * j + 1]};
        c = __builtin_amdgcn_fdot2(x, y, c, false);
    }
    return c;
}

__global__ void __launch_bounds__(256)
conv_out_kernel(const half_t* __restrict__ x, half_t* __restrict__ out, const half_t* __restrict__ w,
                const half_t* __restrict__ bias, int B, int H, int W, int cin, int cout) {
    extern __shared__ __attribute__((aligned(16))) char co_smem[];
    half8* wl = reinterpret_cast<half8*>(co_smem);          // [9][nch][4]
    const int nch = cin / 8;
    for (int e = threadIdx.x; e < 9 * nch * 4; e += 256) {
        const int co = e & 3, r = e >> 2;
        const int tap = r / nch, c8 = r - tap * nch;
        wl[e] = co < cout ? ld_global_h8(w + ((long)co * 9 + tap) * cin + c8 * 8) : zero_h8();
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pg = lane >> 3, cl = lane & 7;
    const long npix = (long)B * H * W;
    const int HW = H * W;
    for (long p0 = ((long)blockIdx.x * 4 + wave) * CO_PPW; p0 < npix; p0 += (long)gridDim.x * 4 * CO_PPW) {
        int pb[CO_PPL], py[CO_PPL], px[CO_PPL];
        bool pok[CO_PPL];
#pragma unroll
        for (int j = 0; j < CO_PPL; ++j) {
            const long p = p0 + pg * CO_PPL + j;
            pok[j] = p < npix;
            const long q = pok[j] ? p : 0;
            pb[j] = (int)(q / HW);
            const int rem = (int)(q - (long)pb[j] * HW);
            py[j] = rem / W; px[j] = rem - py[j] * W;
        }
        float acc[CO_PPL][4];
#pragma unroll
        for (int j = 0; j < CO_PPL; ++j)
#pragma unroll
            for (int co = 0; co < 4; ++co) acc[j][co] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const half_t* src[CO_PPL];
            bool ok[CO_PPL];
#pragma unroll
            for (int j = 0; j < CO_PPL; ++j) {
                const int yy = py[j] + dy, xx = px[j] + dx;
                ok[j] = pok[j] && yy >= 0 && yy < H && xx >= 0 && xx < W;
                src[j] = x + (((long)pb[j] * H + (ok[j] ? yy : 0)) * W + (ok[j] ? xx : 0)) * cin;
            }
            for (int c8 = cl; c8 < nch; c8 += 8) {
                half8 xv[CO_PPL];
#pragma unroll
                for (int j = 0; j < CO_PPL; ++j) xv[j] = ok[j] ? ld_global_h8(src[j] + c8 * 8) : zero_h8();
                const half8* wp = wl + (tap * nch + c8) * 4;
#pragma unroll
                for (int co = 0; co < 4; ++co) {
                    const half8 wv = wp[co];
#pragma unroll
                    for (int j = 0; j < CO_PPL; ++j) acc[j][co] = dot8(xv[j], wv, acc[j][co]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CO_PPL; ++j)
#pragma unroll
            for (int co = 0; co < 4; ++co) {
                float v = acc[j][co];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
                acc[j][co] = v;
            }
        if (cl == 0) {
#pragma unroll
            for (int j = 0; j < CO_PPL; ++j)
                if (pok[j])
                    for (int co = 0; co < cout; ++co)
                        out[(((long)pb[j] * cout + co) * H + py[j]) * W + px[j]] = (half_t)(acc[j][co] + (float)bias[co]);
        }
    }
}

// ---------------------------------------------------------------- timestep path
__global__ void sincos_kernel(const float* __restrict__ v, half_t* __restrict__ out, int rows, int dim) {
    const int half_dim = dim / 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= rows * half_dim) return;
    const int r = q / half_dim, i = q - r * half_dim;
    const float f = expf(-9.210340371976184f * (float)i / (float)half_dim);   // ln(10000)
    const float a = v[r] * f;
    out[(long)r * dim + i] = (half_t)cosf(a);
    out[(long)r * dim + half_dim + i] = (half_t)sinf(a);
}

constexpr int LS_NPW = 4;   // outputs per wave
constexpr int LS_MS = 8;    // rows per register slab

__global__ void __launch_bounds__(256)
linear_small_kernel(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ w, const half_t* __restrict__ b,
                    const half_t* __restrict__ add, int ldadd, half_t* __restrict__ out, int ldo,
                    int M, int N, int K, int act_in, int act_out) {
    const int lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LS_NPW;
    if (n0 >= N) return;
    const int nk = K / 8;
    for (int m0 = 0; m0 < M; m0 += LS_MS) {
        float acc[LS_NPW][LS_MS];
#pragma unroll
        for (int j = 0; j < LS_NPW; ++j)
#pragma unroll
            for (int m = 0; m < LS_MS; ++m) acc[j][m] = 0.f;
        for (int kc = lane; kc < nk; kc += 64) {
            float xv[LS_MS][8];
#pragma unroll
            for (int m = 0; m < LS_MS; ++m) {
                if (m0 + m < M) {
                    const half8 hh = ld_global_h8(x + (long)(m0 + m) * ldx + kc * 8);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float f = (float)hh[i];
                        if (act_in == 1) f = silu_f(f);
                        xv[m][i] = f;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) xv[m][i] = 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < LS_NPW; ++j) {
                if (n0 + j < N) {
                    const half8 wv = ld_global_h8(w + (long)(n0 + j) * K + kc * 8);
#pragma unroll
                    for (int m = 0; m < LS_MS; ++m)
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[j][m] += xv[m][i] * (float)wv[i];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < LS_NPW; ++j)
#pragma unroll
            for (int m = 0; m < LS_MS; ++m) {
                const float r = wave_sum(acc[j][m]);
                if (lane == 0 && n0 + j < N && m0 + m < M) {
                    float v = r + (b ? (float)b[n0 + j] : 0.f);
                    if (add) v += (float)add[(long)(m0 + m) * ldadd + n0 + j];
                    if (act_out == 1) v = silu_f(v);
                    out[(long)(m0 + m) * ldo + n0 + j] = (half_t)v;
                }
            }
    }
}

// ---------------------------------------------------------------- loop glue
__global__ void __launch_bounds__(256)
cfg_ddim_kernel(const half_t* __restrict__ eps, half_t* __restrict__ lat, const float* __restrict__ coef,
                float g, const half_t* __restrict__ mask, const half_t* __restrict__ init,
                const half_t* __restrict__ noise, long n /* B * per_sample, multiple of 8 */) {
    const float cx = coef[0], ce = coef[1];
    const float ci = mask ? coef[2] : 0.f, cn = mask ? coef[3] : 0.f;
    for (long q = ((long)blockIdx.x * 256 + threadIdx.x) * 8; q < n; q += (long)gridDim.x * 256 * 8) {
        const half8 eu = ld_global_h8(eps + q), ec = ld_global_h8(eps + n + q), xl = ld_global_h8(lat + q);
        half8 mk, in0, nz;
        if (mask) { mk = ld_global_h8(mask + q); in0 = ld_global_h8(init + q); nz = ld_global_h8(noise + q); }
        half8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float u = (float)eu[i], c = (float)ec[i];
            const float e = u + g * (c - u);
            float v = cx * (float)xl[i] + ce * e;
            if (mask) {
                const float m = (float)mk[i];
                v = (1.f - m) * (ci * (float)in0[i] + cn * (float)nz[i]) + m * v;
            }
            o[i] = (half_t)v;
        }
        *reinterpret_cast<half8*>(lat + q) = o;
    }
}

__global__ void __launch_bounds__(256)
add_inplace_kernel(half_t* __restrict__ y, const half_t* __restrict__ a, long n, long na) {
    for (long q = ((long)blockIdx.x * 256 + threadIdx.x) * 8; q < n; q += (long)gridDim.x * 256 * 8) {
        half8 yv = ld_global_h8(y + q);
        const half8 av = ld_global_h8(a + (q % na));
#pragma unroll
        for (int i = 0; i < 8; ++i) yv[i] = (half_t)((float)yv[i] + (float)av[i]);
        *reinterpret_cast<half8*>(y + q) = yv;
    }
}

inline int grid_for(long items, int per_block, int cap) {
    long g = (items + per_block - 1) / per_block;
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int cid_conv_in_f16(const cid_half* sample, cid_half* out, const cid_half* w, const cid_half* bias,
                               int32_t B, int32_t Bin, int32_t cin, int32_t H, int32_t W, int32_t cout,
                               const float* in_scale, cid_stream_t stream) {
    CID_CHECK_ARG(sample && out && w && bias, "cid_conv_in_f16: null pointer");
    CID_CHECK_ARG(B > 0 && Bin > 0 && cin > 0 && cin <= 9 && cout % 8 == 0 && cout <= CIN_MAXCO && H > 0 && W > 0,
                  "cid_conv_in_f16: bad shape (cin <= 9, cout <= 320)");
    CID_CHECK_ARG(B % Bin == 0, "cid_conv_in_f16: B must be a multiple of Bin (batch row b reads sample b %% Bin)");
    const long items = (long)Bin * H * W;                   // pixels computed (rows that share a sample are stored B / Bin times)
    const dim3 grid(grid_for(items, 256 / CIN_TPP, 512));
#define CID_CONV_IN(A, Bc) hipLaunchKernelGGL((conv_in_kernel<A, Bc>), grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)sample, \
                       (const half_t*)sample, (half_t*)out, (const half_t*)w, (const half_t*)bias, B, Bin, cin, 0, H, W, cout, in_scale)
    if (cin == 4) CID_CONV_IN(4, 0);
    else if (cin == 9) CID_CONV_IN(9, 0);
    else CID_CONV_IN(0, 0);
#undef CID_CONV_IN
    CID_CHECK_LAUNCH("cid_conv_in_f16");
    return 0;
}

extern "C" int cid_conv_in_cat_f16(const cid_half* sample, int32_t cin1, const cid_half* extra, int32_t cin2, cid_half* out,
                                   const cid_half* w, const cid_half* bias, int32_t B, int32_t Bin, int32_t H, int32_t W,
                                   int32_t cout, const float* in_scale, cid_stream_t stream) {
    CID_CHECK_ARG(sample && extra && out && w && bias, "cid_conv_in_cat_f16: null pointer");
    CID_CHECK_ARG(B > 0 && Bin > 0 && cin1 > 0 && cin2 > 0 && cin1 + cin2 <= 9 && cout % 8 == 0 && cout <= CIN_MAXCO && H > 0 &&
                  W > 0, "cid_conv_in_cat_f16: bad shape (cin1 + cin2 <= 9, cout <= 320)");
    CID_CHECK_ARG(B % Bin == 0, "cid_conv_in_cat_f16: B must be a multiple of Bin (batch row b reads sample b %% Bin)");
    const long items = (long)Bin * H * W;
    const dim3 grid(grid_for(items, 256 / CIN_TPP, 512));
    if (cin1 == 4 && cin2 == 5)
        hipLaunchKernelGGL((conv_in_kernel<4, 5>), grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)sample, (const half_t*)extra,
                           (half_t*)out, (const half_t*)w, (const half_t*)bias, B, Bin, cin1, cin2, H, W, cout, in_scale);
    else
        hipLaunchKernelGGL((conv_in_kernel<0, 0>), grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)sample, (const half_t*)extra,
                           (half_t*)out, (const half_t*)w, (const half_t*)bias, B, Bin, cin1, cin2, H, W, cout, in_scale);
    CID_CHECK_LAUNCH("cid_conv_in_cat_f16");
    return 0;
}


extern "C" int cid_conv3x3_small_f16(const cid_half* x, cid_half* out, const cid_half* w, const cid_half* bias,
                                     int32_t B, int32_t Hi, int32_t Wi, int32_t cin, int32_t cout, int32_t stride,
                                     int32_t silu, cid_stream_t stream) {
    CID_CHECK_ARG(x && out && w && bias, "cid_conv3x3_small_f16: null pointer");
    CID_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && cin > 0 && cout > 0 && cout % 8 == 0 && (stride == 1 || stride == 2),
                  "cid_conv3x3_small_f16: bad shape (cout %% 8, stride 1 or 2)");
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;     // kernel 3, pad 1
    const long items = (long)B * Ho * Wo * (cout / 8);
    hipLaunchKernelGGL(conv3x3_small_kernel, dim3(grid_for(items, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)x, (half_t*)out, (const half_t*)w, (const half_t*)bias, B, Hi, Wi, Ho, Wo, cin, cout,
                       stride, silu);
    CID_CHECK_LAUNCH("cid_conv3x3_small_f16");
    return 0;
}


extern "C" int cid_gelu_f16(cid_half* x, int64_t n, cid_stream_t stream) {
    CID_CHECK_ARG(x && n > 0 && n % 8 == 0, "cid_gelu_f16: n must be a positive multiple of 8");
    hipLaunchKernelGGL(gelu_kernel, dim3(grid_for(n / 8, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (half_t*)x, (long)(n / 8));
    CID_CHECK_LAUNCH("cid_gelu_f16");
    return 0;
}

extern "C" int cid_small_attn_f16(const cid_half* q, int32_t ldq, const cid_half* kv1, int32_t n1, const cid_half* kv2,
                                  int32_t n2, int32_t ldkv, cid_half* out, int32_t ldo, int32_t B, int32_t Lq,
                                  int32_t heads, int32_t dim_head, float scale2, cid_stream_t stream) {
    CID_CHECK_ARG(q && kv1 && out && (n2 == 0 || kv2), "cid_small_attn_f16: null pointer");
    CID_CHECK_ARG(dim_head == 64, "cid_small_attn_f16: head width must be 64 (got %d)", dim_head);
    CID_CHECK_ARG(B > 0 && Lq > 0 && heads > 0 && n1 > 0 && n2 >= 0 && n1 + n2 <= 64 * SA_MAXK && ldq % 8 == 0 && ldkv % 8 == 0,
                  "cid_small_attn_f16: bad shape (at most %d keys)", 64 * SA_MAXK);
    const int total = B * heads * Lq;
    hipLaunchKernelGGL(small_attn_kernel, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const half_t*)q, ldq,
                       (const half_t*)kv1, n1, (const half_t*)kv2, n2, ldkv, (half_t*)out, ldo, Lq, heads, scale2, total);
    CID_CHECK_LAUNCH("cid_small_attn_f16");
    return 0;
}

extern "C" int cid_conv_out_f16(const cid_half* x, cid_half* out, const cid_half* w, const cid_half* bias,
                                int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout, cid_stream_t stream) {
    CID_CHECK_ARG(x && out && w && bias, "cid_conv_out_f16: null pointer");
    CID_CHECK_ARG(B > 0 && H > 0 && W > 0 && cin % 8 == 0 && cout > 0 && cout <= 4, "cid_conv_out_f16: bad shape (cout <= 4)");
    const long pix = (long)B * H * W;
    const int smem = 9 * (cin / 8) * 4 * 16;                 // weights [9][cin / 8][4] x 16 B
    CID_CHECK_ARG(smem <= 64 * 1024, "cid_conv_out_f16: cin = %d is too wide (the weights are held in 64 KB of LDS)", cin);
    hipLaunchKernelGGL(conv_out_kernel, dim3(grid_for(pix, 4 * CO_PPW, 2048)), dim3(256), smem, (hipStream_t)stream,
                       (const half_t*)x, (half_t*)out, (const half_t*)w, (const half_t*)bias, B, H, W, cin, cout);
    CID_CHECK_LAUNCH("cid_conv_out_f16");
    return 0;
}

extern "C" int cid_sincos_embed_f16(const float* v, cid_half* out, int32_t rows, int32_t dim, cid_stream_t stream) {
    CID_CHECK_ARG(v && out && rows > 0 && dim > 0 && dim % 2 == 0, "cid_sincos_embed_f16: bad arguments");
    const int n = rows * (dim / 2);
    hipLaunchKernelGGL(sincos_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, v, (half_t*)out, rows, dim);
    CID_CHECK_LAUNCH("cid_sincos_embed_f16");
    return 0;
}

extern "C" int cid_linear_small_f16(const cid_half* x, int32_t ldx, const cid_half* w, const cid_half* b,
                                    const cid_half* add, int32_t ldadd, cid_half* out, int32_t ldo,
                                    int32_t M, int32_t N, int32_t K, int32_t act_in, int32_t act_out,
                                    cid_stream_t stream) {
    CID_CHECK_ARG(x && w && out, "cid_linear_small_f16: null pointer");
    CID_CHECK_ARG(M > 0 && M <= 64 && N > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0, "cid_linear_small_f16: bad shape (M <= 64, K %% 8)");
    const int waves = (N + LS_NPW - 1) / LS_NPW;
    hipLaunchKernelGGL(linear_small_kernel, dim3((waves + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)x, ldx, (const half_t*)w, (const half_t*)b, (const half_t*)add, ldadd,
                       (half_t*)out, ldo, M, N, K, act_in, act_out);
    CID_CHECK_LAUNCH("cid_linear_small_f16");
    return 0;
}

extern "C" int cid_cfg_ddim_step_f16(const cid_half* eps, cid_half* latents, const float* coef, float guidance,
                                     const cid_half* mask, const cid_half* init, const cid_half* noise,
                                     int32_t B, int32_t per_sample, cid_stream_t stream) {
    CID_CHECK_ARG(eps && latents && coef, "cid_cfg_ddim_step_f16: null pointer");
    CID_CHECK_ARG((mask == nullptr) == (init == nullptr) && (mask == nullptr) == (noise == nullptr),
                  "cid_cfg_ddim_step_f16: mask/init/noise must come together");
    const long n = (long)B * per_sample;
    CID_CHECK_ARG(B > 0 && per_sample > 0 && n % 8 == 0, "cid_cfg_ddim_step_f16: B * per_sample must be a multiple of 8");
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid_for(n / 8, 256, 1024)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)eps, (half_t*)latents, coef, guidance, (const half_t*)mask, (const half_t*)init,
                       (const half_t*)noise, n);
    CID_CHECK_LAUNCH("cid_cfg_ddim_step_f16");
    return 0;
}

extern "C" int cid_add_inplace_f16(cid_half* y, const cid_half* a, int64_t n, int64_t na, cid_stream_t stream) {
    CID_CHECK_ARG(y && a && n > 0 && na > 0 && n % 8 == 0 && na % 8 == 0 && n % na == 0, "cid_add_inplace_f16: sizes must be multiples of 8");
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n / 8, 256, 2048)), dim3(256), 0, (hipStream_t)stream,
                       (half_t*)y, (const half_t*)a, (long)n, (long)na);
    CID_CHECK_LAUNCH("cid_add_inplace_f16");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Per-step values of the denoise loop (t, scheduler coefficients, embed-set rows, time-embedding row, SDXL pooled embeds):
// row *counter of a device table -> the buffers the captured step reads; then ++*counter.  One workgroup.
namespace {
struct StepSegs { cid_step_seg s[8]; int n; };
__global__ void __launch_bounds__(1024)
step_select_kernel(const unsigned* __restrict__ table, long row_words, int n_rows, int* counter, StepSegs segs) {
    int row = *counter;
    if (row < 0) row = 0;
    if (row > n_rows - 1) row = n_rows - 1;
    __syncthreads();                           // every thread has read the counter before it moves on
    const unsigned* src = table + (long)row * row_words;
    // one workgroup (the counter update needs no second launch), 1024 lanes, 16 bytes per lane where the segment and its
    // place in the row are 16-byte aligned: a 40-KB time-embedding row is 2.5 load rounds instead of 40 dependent ones
    for (int k = 0; k < segs.n; ++k) {
        unsigned* dst = reinterpret_cast<unsigned*>(segs.s[k].dst);
        const long w0 = segs.s[k].offset >> 2, nw = segs.s[k].nbytes >> 2;
        const bool wide = (((uintptr_t)dst | (uintptr_t)(src + w0)) & 15) == 0;
        long done = 0;
        if (wide) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const long nq = nw >> 2;
            for (long i = threadIdx.x; i < nq; i += 1024)
                reinterpret_cast<u32x4*>(dst)[i] = reinterpret_cast<const u32x4*>(src + w0)[i];
            done = nq << 2;
        }
        for (long i = done + threadIdx.x; i < nw; i += 1024) dst[i] = src[w0 + i];
    }
    if (threadIdx.x == 0) *counter = row + 1;
}
}  // namespace

extern "C" int cid_step_select(const void* table, int64_t row_bytes, int32_t n_rows, int32_t* counter, const cid_step_seg* segs,
                               int32_t n_segs, cid_stream_t stream) {
    CID_CHECK_ARG(table && counter && segs, "cid_step_select: null pointer");
    CID_CHECK_ARG(n_rows > 0 && row_bytes > 0 && row_bytes % 4 == 0 && n_segs > 0 && n_segs <= 8,
                  "cid_step_select: bad table (%d rows of %ld bytes, %d segments; at most 8)", n_rows, (long)row_bytes, n_segs);
    StepSegs sg;
    sg.n = n_segs;
    for (int k = 0; k < n_segs; ++k) {
        CID_CHECK_ARG(segs[k].dst && segs[k].offset >= 0 && segs[k].nbytes > 0 && segs[k].offset % 4 == 0 && segs[k].nbytes % 4 == 0 &&
                      segs[k].offset + segs[k].nbytes <= row_bytes, "cid_step_select: segment %d does not fit the row", k);
        sg.s[k] = segs[k];
    }
    hipLaunchKernelGGL(step_select_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const unsigned*)table,
                       (long)(row_bytes >> 2), n_rows, counter, sg);
    CID_CHECK_LAUNCH("cid_step_select");
    return 0;
}
