// Implicit-GEMM kernel for every Linear / 1x1 / 3x3 convolution of the UNet
// (SURVEY.md 8a rows a5-a9).  out[m][n] = epi(sum_k A(m,k) W[n][k]),  fp16 in,
// fp32 MFMA accumulate, fp16 out.
//
// Roofline: MFMA-bound (v_mfma_f32_16x16x32_f16, dense fp16 2.5 PFLOP/s); algorithmic
// work 2*M*N*K flop, bytes (M*K + N*K + M*N) * 2.
//
// Why 16x16x32 tiles: every channel count of the SD / SDXL UNet is 320 * 2^k = 5 * 64 * 2^k,
// so a wave tile of 80 channels (5 x 16) x 64 tokens (4 x 16) tiles N exactly with 2 or 4
// waves across N; 32-wide MFMA tiles cannot split 160 channels over an even wave count.
//
// Structure (workgroup = WM x WN waves, wave tile = TM x TN tiles of 16 x 16):
//   * activations / weights are staged global -> VGPR -> LDS in BK = 64 slabs (128-B rows),
//     double buffered, ONE barrier per slab; loads of slab t+1 are issued before the MFMAs of
//     slab t and written to LDS after them (issue-early / write-late);
//   * 16-B chunk c of LDS row r is stored at chunk c ^ ((r >> 1) & 7): a ds_read_b128 of 16
//     consecutive rows at one k-chunk touches 16 distinct bank slots (conflict free);
//   * operands are fed "swapped" (MFMA A = weight rows, B = token rows) so a lane owns 4
//     consecutive CHANNELS of one token -> 8-byte stores; the 3x3 taps are shifted token
//     rows of the same token-major image (zero outside), nearest-2x upsampling and stride 2
//     are folded into the row gather, a skip concat is two source pointers;
//   * the V third of the fused QKV projection flips the operand roles so a lane owns 4
//     consecutive TOKENS of one channel and writes V transposed for the attention kernel;
//   * small-M / deep-K problems (the 8x8 and 16x16 levels) are split along K over
//     gridDim.z with fp32 partial tiles + a fused reduce/epilogue kernel, so that every
//     launch puts >= ~2 waves on every SIMD of the 256 CUs.
#include "common.h"
#include "../../include/cid.h"

namespace {

struct GemmArgs {
    const half_t* x1; const half_t* x2;
    int c1, c2, ld1, ld2;
    const half_t* w;
    half_t* out; int ldo;
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
    float* ws;           // [splitk][M][N] fp32 partials when splitk > 1
};

constexpr int BK = 64;

typedef float f32x4v __attribute__((ext_vector_type(4)));

CID_DEVINL f32x4v mfma16(half8 a, half8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// byte offset of 16-B chunk c (0..7) of row r in a [rows][64] fp16 LDS tile
CID_DEVINL int lds_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

template <int TM, int TN, int WM, int WN, bool VMODE>
__global__ void __launch_bounds__(64 * WM * WN, (WM * WN >= 8) ? 2 : 1)
igemm_kernel(GemmArgs a) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 16 * TM * WM;
    constexpr int BN = 16 * TN * WN;
    constexpr int XCH = (BM * 8 + NT - 1) / NT;   // 16-B chunks per thread, activations
    constexpr int WCH = (BN * 8 + NT - 1) / NT;   // weights
    constexpr int XBYTES = BM * 128, WBYTES = BN * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, lq = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = a.n_begin + blockIdx.x * BN;
    const int m0 = blockIdx.y * BM;

    // ---- per-thread staging metadata ------------------------------------------
    const int cchunk = tid & 7;  // NT % 8 == 0, so every chunk of a thread has the same c
    int xrow[XCH], xb[XCH], xy[XCH], xx[XCH];
    bool xok[XCH];
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
        const int q = tid + j * NT;
        xrow[j] = q >> 3;
        const int m = m0 + xrow[j];
        xok[j] = (xrow[j] < BM) && (m < a.M);
        if (a.taps == 9) {
            const int hw = a.Ho * a.Wo;
            const int b = m / hw, rem = m - b * hw;
            xb[j] = b; xy[j] = rem / a.Wo; xx[j] = rem - xy[j] * a.Wo;
        } else {
            xb[j] = 0; xy[j] = 0; xx[j] = m;
        }
    }
    int wrow[WCH]; bool wok[WCH];
#pragma unroll
    for (int j = 0; j < WCH; ++j) {
        const int q = tid + j * NT;
        wrow[j] = q >> 3;
        wok[j] = (wrow[j] < BN) && (n0 + wrow[j] < a.n_end);
    }

    half8 xreg[XCH], wreg[WCH];

    auto stage_load = [&](int slab) {
        const int tap = slab / a.cslabs;
        const int cs = slab - tap * a.cslabs;
        const int cbase = cs * BK;
        const half_t* src; int ld, coff;
        if (cbase < a.c1) { src = a.x1; ld = a.ld1; coff = cbase; }
        else              { src = a.x2; ld = a.ld2; coff = cbase - a.c1; }
        const int dy = (a.taps == 9) ? tap / 3 - 1 : 0;
        const int dx = (a.taps == 9) ? tap - (tap / 3) * 3 - 1 : 0;
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            half8 v = zero_h8();
            if (xok[j]) {
                long row;
                bool ok = true;
                if (a.taps == 9) {
                    int yy = xy[j] * a.stride + dy, xs = xx[j] * a.stride + dx;
                    const int Hv = a.Hi << a.up, Wv = a.Wi << a.up;  // virtual (upsampled) input
                    ok = (yy >= 0) && (yy < Hv) && (xs >= 0) && (xs < Wv);
                    yy >>= a.up; xs >>= a.up;
                    row = ((long)xb[j] * a.Hi + yy) * a.Wi + xs;
                } else {
                    row = xx[j];
                }
                if (ok) v = ld_global_h8(src + row * ld + coff + cchunk * 8);
            }
            xreg[j] = v;
        }
#pragma unroll
        for (int j = 0; j < WCH; ++j) {
            half8 v = zero_h8();
            if (wok[j]) v = ld_global_h8(a.w + (long)(n0 + wrow[j]) * a.ktot + slab * BK + cchunk * 8);
            wreg[j] = v;
        }
    };
    auto stage_write = [&](int buf) {
        char* xs = smem + buf * (XBYTES + WBYTES);
        char* ws = xs + XBYTES;
#pragma unroll
        for (int j = 0; j < XCH; ++j)
            if (xrow[j] < BM) *reinterpret_cast<half8*>(xs + lds_off(xrow[j], cchunk)) = xreg[j];
#pragma unroll
        for (int j = 0; j < WCH; ++j)
            if (wrow[j] < BN) *reinterpret_cast<half8*>(ws + lds_off(wrow[j], cchunk)) = wreg[j];
    };

    f32x4v acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int c = 0; c < TN; ++c) acc[t][c] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // K range of this split
    const int s_begin = (int)((long)a.nslab * blockIdx.z / a.splitk);
    const int s_end = (int)((long)a.nslab * (blockIdx.z + 1) / a.splitk);

    stage_load(s_begin);
    stage_write(0);
    __syncthreads();

    int cur = 0;
    for (int slab = s_begin; slab < s_end; ++slab) {
        const bool more = slab + 1 < s_end;
        if (more) stage_load(slab + 1);
        const char* xs = smem + cur * (XBYTES + WBYTES);
        const char* ws = xs + XBYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 xf[TM], wf[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t)
                xf[t] = *reinterpret_cast<const half8*>(xs + lds_off((wm * TM + t) * 16 + l16, ks * 4 + lq));
#pragma unroll
            for (int c = 0; c < TN; ++c)
                wf[c] = *reinterpret_cast<const half8*>(ws + lds_off((wn * TN + c) * 16 + l16, ks * 4 + lq));
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int c = 0; c < TN; ++c)
                    acc[t][c] = VMODE ? mfma16(xf[t], wf[c], acc[t][c]) : mfma16(wf[c], xf[t], acc[t][c]);
        }
        if (more) stage_write(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue ---------------------------------------------------------------
    if constexpr (VMODE) {
        // D rows = tokens (4 lq + i), cols = channels (l16): lane owns channel n, 4 consecutive tokens
#pragma unroll
        for (int c = 0; c < TN; ++c) {
            const int n = n0 + (wn * TN + c) * 16 + l16;
            if (n >= a.n_end) continue;
            const int cg = n - a.n_vt0;
            const int head = cg / a.dhead, dd = cg - head * a.dhead;
            const float bv = a.bias ? (float)a.bias[n] : 0.f;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int mt = m0 + (wm * TM + t) * 16;   // 16-token tile base
                if (mt >= a.M) continue;
                const int b = mt / a.ntok, tok0 = mt - b * a.ntok;
                half_t* dst = a.vt + ((long)(b * a.heads + head) * a.dvp + dd) * a.ntok + tok0;
                half4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (half_t)(acc[t][c][i] + bv);
                // token 4 lq + i of the 16-group -> pos = 8 (lq & 1) + 4 (lq >> 1) + i
                *reinterpret_cast<half4*>(dst + 8 * (lq & 1) + 4 * (lq >> 1)) = o;
            }
        }
        return;
    } else {
        if (a.splitk > 1) {
            // fp32 partial tile; bias / residual / conversion happen in splitk_epilogue_kernel
            float* wsp = a.ws + (long)blockIdx.z * a.M * a.N;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int m = m0 + (wm * TM + t) * 16 + l16;
                if (m >= a.M) continue;
#pragma unroll
                for (int c = 0; c < TN; ++c) {
                    const int n = n0 + (wn * TN + c) * 16 + 4 * lq;
                    if (n >= a.n_end) continue;
                    *reinterpret_cast<f32x4v*>(wsp + (long)m * a.N + n) = acc[t][c];
                }
            }
            return;
        }
        if (a.mode == 1) {
            // GEGLU: even 16-row tile = value, odd = gate (weights interleaved by the host)
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int m = m0 + (wm * TM + t) * 16 + l16;
                if (m >= a.M) continue;
#pragma unroll
                for (int c = 0; c + 1 < TN; c += 2) {
                    const int nt = n0 + (wn * TN + c) * 16;   // interleaved column of the value tile
                    if (nt >= a.n_end) continue;
                    const int no = (nt >> 1) + 4 * lq;        // output column
                    half4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[t][c][i], g = acc[t][c + 1][i];
                        if (a.bias) { v += (float)a.bias[nt + 4 * lq + i]; g += (float)a.bias[nt + 16 + 4 * lq + i]; }
                        o[i] = (half_t)(v * gelu_erf_f(g));
                    }
                    *reinterpret_cast<half4*>(a.out + (long)m * a.ldo + no) = o;
                }
            }
            return;
        }
        // plain: lane owns token m, 4 consecutive channels per tile
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int m = m0 + (wm * TM + t) * 16 + l16;
            if (m >= a.M) continue;
            const half_t* rb = a.rowbias ? a.rowbias + (long)(m / a.rows_per_sample) * a.ld_rowbias : nullptr;
            const half_t* rs = a.res ? a.res + (long)m * a.ldr : nullptr;
            half_t* op = a.out + (long)m * a.ldo;
#pragma unroll
            for (int c = 0; c < TN; ++c) {
                const int n = n0 + (wn * TN + c) * 16 + 4 * lq;
                if (n >= a.n_end) continue;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[t][c][i];
                if (a.bias) {
                    const half4 bb = *reinterpret_cast<const half4*>(a.bias + n);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)bb[i];
                }
                if (rb) {
                    const half4 bb = *reinterpret_cast<const half4*>(rb + n);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)bb[i];
                }
                if (rs) {
                    const half4 bb = *reinterpret_cast<const half4*>(rs + n);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)bb[i];
                }
                half4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (half_t)v[i];
                *reinterpret_cast<half4*>(op + n) = o;
            }
        }
    }
}

// sum the split-K partials and apply the plain epilogue; one thread per 4 output channels
__global__ void __launch_bounds__(256)
splitk_epilogue_kernel(GemmArgs a) {
    const int n4 = a.N >> 2;
    const long total = (long)a.M * n4;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const int m = (int)(q / n4);
        const int n = (int)(q - (long)m * n4) * 4;
        f32x4v v = *reinterpret_cast<const f32x4v*>(a.ws + (long)m * a.N + n);
        for (int z = 1; z < a.splitk; ++z) {
            const f32x4v p = *reinterpret_cast<const f32x4v*>(a.ws + ((long)z * a.M + m) * a.N + n);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += p[i];
        }
        if (a.bias) {
            const half4 bb = *reinterpret_cast<const half4*>(a.bias + n);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += (float)bb[i];
        }
        if (a.rowbias) {
            const half4 bb = *reinterpret_cast<const half4*>(a.rowbias + (long)(m / a.rows_per_sample) * a.ld_rowbias + n);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += (float)bb[i];
        }
        if (a.res) {
            const half4 bb = *reinterpret_cast<const half4*>(a.res + (long)m * a.ldr + n);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += (float)bb[i];
        }
        half4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (half_t)v[i];
        *reinterpret_cast<half4*>(a.out + (long)m * a.ldo + n) = o;
    }
}

template <int TM, int TN, int WM, int WN, bool VMODE>
int launch_one(const GemmArgs& a, int ncols, hipStream_t s) {
    constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
    constexpr int SMEM = 2 * (BM + BN) * 128;
    auto kern = igemm_kernel<TM, TN, WM, WN, VMODE>;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) {
            cid_set_error("cid_gemm_f16: cannot reserve %d bytes of LDS", SMEM);
            return -5;
        }
        configured = true;
    }
    dim3 grid((ncols + BN - 1) / BN, (a.M + BM - 1) / BM, VMODE ? 1 : a.splitk);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), SMEM, s, a);
    return 0;
}

template <int TM, int TN, int WM, int WN>
int launch(GemmArgs a, hipStream_t s) {
    const int n_plain = (a.mode == 2) ? a.n_vt0 : a.N;
    a.n_begin = 0; a.n_end = n_plain;
    int rc = launch_one<TM, TN, WM, WN, false>(a, n_plain, s);
    if (rc) return rc;
    if (a.splitk > 1) {
        const long items = (long)a.M * (a.N >> 2);
        const int grid = (int)((items + 255) / 256 > 2048 ? 2048 : (items + 255) / 256);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(grid), dim3(256), 0, s, a);
    }
    if (a.mode == 2 && a.N > a.n_vt0) {
        // the V third: flipped operand roles, transposed store
        GemmArgs v = a;
        v.n_begin = a.n_vt0; v.n_end = a.N; v.splitk = 1;
        rc = launch_one<TM, TN, WM, WN, true>(v, a.N - a.n_vt0, s);
    }
    return rc;
}

}  // namespace

extern "C" int cid_gemm_f16(const cid_gemm_desc* d, cid_stream_t stream) {
    CID_CHECK_ARG(d && d->x1 && d->w && d->out, "cid_gemm_f16: null pointer");
    CID_CHECK_ARG(d->taps == 1 || d->taps == 9, "cid_gemm_f16: taps must be 1 or 9 (got %d)", d->taps);
    CID_CHECK_ARG(d->c1 > 0 && d->c1 % 32 == 0 && d->c2 >= 0 && d->c2 % 32 == 0 && (d->c1 + d->c2) % 64 == 0 &&
                  (d->c2 == 0 || d->c1 % 64 == 0),
                  "cid_gemm_f16: channel counts must be multiples of 64 (c1=%d c2=%d)", d->c1, d->c2);
    CID_CHECK_ARG(d->c2 == 0 || d->x2, "cid_gemm_f16: c2 > 0 needs x2");
    CID_CHECK_ARG(d->N > 0 && d->N % 32 == 0 && d->M > 0, "cid_gemm_f16: bad M/N (%d, %d)", d->M, d->N);
    CID_CHECK_ARG(d->mode >= 0 && d->mode <= 2, "cid_gemm_f16: bad mode %d", d->mode);
    CID_CHECK_ARG(d->ld1 % 8 == 0 && d->ldo % 4 == 0 && (d->c2 == 0 || d->ld2 % 8 == 0),
                  "cid_gemm_f16: row pitches must keep 16-byte alignment");
    GemmArgs a;
    a.x1 = (const half_t*)d->x1; a.x2 = (const half_t*)d->x2;
    a.c1 = d->c1; a.c2 = d->c2; a.ld1 = d->ld1; a.ld2 = d->ld2;
    a.w = (const half_t*)d->w; a.out = (half_t*)d->out; a.ldo = d->ldo;
    a.bias = (const half_t*)d->bias;
    a.rowbias = (const half_t*)d->rowbias; a.ld_rowbias = d->ld_rowbias;
    a.rows_per_sample = d->rows_per_sample > 0 ? d->rows_per_sample : 1;
    a.res = (const half_t*)d->res; a.ldr = d->ldr;
    a.M = d->M; a.N = d->N; a.taps = d->taps;
    a.Hi = d->Hi; a.Wi = d->Wi; a.Ho = d->Ho; a.Wo = d->Wo; a.stride = d->stride; a.up = d->up;
    a.mode = d->mode;
    a.vt = (half_t*)d->vt; a.n_vt0 = d->n_vt0; a.heads = d->heads; a.dhead = d->dhead;
    a.dvp = d->dvp; a.ntok = d->ntok;
    a.cslabs = (d->c1 + d->c2) / BK;
    a.ktot = d->taps * (d->c1 + d->c2);
    a.nslab = a.ktot / BK;
    a.splitk = 1;
    a.ws = (float*)d->ws;
    a.n_begin = 0; a.n_end = a.N;
    if (d->taps == 9) {
        CID_CHECK_ARG(d->Hi > 0 && d->Wi > 0 && d->Ho > 0 && d->Wo > 0 && (d->stride == 1 || d->stride == 2)
                      && (d->up == 0 || d->up == 1), "cid_gemm_f16: bad conv geometry");
        CID_CHECK_ARG(d->M % (d->Ho * d->Wo) == 0, "cid_gemm_f16: M is not batch * Ho * Wo");
    }

    // ---- tile choice: aim for >= 2 waves on each of the 1024 SIMDs ------------------------
    enum { A256x160, B128x160, C64x160, G256x128, G128x128, O64x64, O128x32 } cfg;
    const int n_plain = (d->mode == 2) ? d->n_vt0 : d->N;
    const long target = 2048;
    auto waves = [&](int bm_, int bn_, int w) {
        return (long)((a.M + bm_ - 1) / bm_) * ((n_plain + bn_ - 1) / bn_) * w;
    };
    int bm = 0, bn = 0, nw = 0;
    if (d->mode == 1) {
        if (d->N % 128 != 0) { cfg = O64x64; bm = 64; bn = 64; nw = 4; }
        else if (waves(256, 128, 8) >= target) { cfg = G256x128; bm = 256; bn = 128; nw = 8; }
        else { cfg = G128x128; bm = 128; bn = 128; nw = 8; }
    } else if (n_plain % 160 == 0) {
        if (waves(256, 160, 8) >= target)      { cfg = A256x160; bm = 256; bn = 160; nw = 8; }
        else if (waves(128, 160, 8) >= target) { cfg = B128x160; bm = 128; bn = 160; nw = 8; }
        else                                   { cfg = C64x160;  bm = 64;  bn = 160; nw = 4; }
    } else if (n_plain % 64 == 0) { cfg = O64x64; bm = 64; bn = 64; nw = 4; }
    else { cfg = O128x32; bm = 128; bn = 32; nw = 4; }
    if (d->mode == 1) CID_CHECK_ARG(d->N % 64 == 0, "cid_gemm_f16: GEGLU needs N %% 64 == 0");
    if (d->mode == 2) {
        CID_CHECK_ARG(d->vt && d->ntok % 16 == 0 && d->M % d->ntok == 0 && d->n_vt0 % bn == 0 && d->dhead > 0
                      && d->heads > 0 && d->dvp >= d->dhead && (d->N - d->n_vt0) % 16 == 0,
                      "cid_gemm_f16: bad QKV/V^T description");
    }
    // split-K for small-M / deep-K problems (plain epilogue only)
    if (d->mode == 0 && a.ws && a.nslab >= 16) {
        const long w = waves(bm, bn, nw);
        if (w < target) {
            int sk = (int)((target + w - 1) / w);
            if (sk > 8) sk = 8;
            while (sk > 1 && a.nslab / sk < 8) --sk;
            while (sk > 1 && (int64_t)sk * a.M * a.N * 4 > d->ws_bytes) --sk;
            a.splitk = sk;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    int rc = 0;
    switch (cfg) {
        case A256x160: rc = launch<4, 5, 4, 2>(a, s); break;
        case B128x160: rc = launch<2, 5, 4, 2>(a, s); break;
        case C64x160:  rc = launch<2, 5, 2, 2>(a, s); break;
        case G256x128: rc = launch<4, 4, 4, 2>(a, s); break;
        case G128x128: rc = launch<2, 4, 4, 2>(a, s); break;
        case O64x64:   rc = launch<2, 2, 2, 2>(a, s); break;
        case O128x32:  rc = launch<2, 2, 4, 1>(a, s); break;
    }
    if (rc) return rc;
    CID_CHECK_LAUNCH("cid_gemm_f16");
    return 0;
}
