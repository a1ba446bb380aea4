// Implicit-GEMM kernel for every Linear / 1x1 / 3x3 convolution of the UNet
// (SURVEY.md 8a rows a5-a9).  out[m][n] = epi(sum_k A(m,k) W[n][k]),  fp16 in,
// fp32 MFMA accumulate, fp16 out.
//
// Roofline: MFMA-bound (v_mfma_f32_32x32x16_f16, 2.5 PFLOP/s dense); algorithmic
// work 2*M*N*K flop, bytes (M*K + N*K + M*N) * 2.
//
// Structure (one workgroup = WT x WC waves, wave tile = TTW token tiles x NCT
// channel tiles of 32x32):
//   * activations and weights are staged global -> VGPR -> LDS in BK = 32 slabs,
//     double buffered, ONE barrier per slab; the loads of slab t+1 are issued before
//     the MFMAs of slab t and written to LDS after them (issue-early / write-late);
//   * LDS rows are 64 B with the 16-B chunk index XOR-swizzled by (row >> 2) & 3 so a
//     ds_read_b128 of 16 consecutive rows touches 16 distinct bank slots;
//   * operands are fed "swapped" (A = weight rows, B = token rows) so that a lane ends
//     up with 4 consecutive CHANNELS of one token -> 8-byte coalesced stores, and the
//     3x3 taps are shifted token rows of the same token-major image (zero outside);
//   * the V third of the fused QKV projection flips the operand roles instead, so a
//     lane holds 4 consecutive TOKENS of one channel and writes V transposed.
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
    int ktot;    // taps * (c1 + c2)
    int nslab;   // ktot / 32
    int cslabs;  // (c1 + c2) / 32
};

constexpr int BK = 32;

// byte offset of 16-B chunk c (0..3) of row r in a [rows][32] fp16 LDS tile
CID_DEVINL int lds_off(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

template <int TTW, int NCT, int WT, int WC, bool VMODE>
__global__ void __launch_bounds__(64 * WT * WC)
igemm_kernel(GemmArgs a) {
    constexpr int NT = 64 * WT * WC;
    constexpr int BM = 32 * TTW * WT;
    constexpr int BN = 32 * NCT * WC;
    constexpr int XCH = (BM * 4 + NT - 1) / NT;   // 16-B chunks per thread, activations
    constexpr int WCH = (BN * 4 + NT - 1) / NT;   // weights
    constexpr int XBYTES = BM * 64, WBYTES = BN * 64;
    __shared__ __attribute__((aligned(16))) char smem[2 * (XBYTES + WBYTES)];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int idx = lane & 31, hi = lane >> 5;
    const int wt = wave / WC, wc = wave % WC;
    // grid: x = channel tiles (fast, so consecutive blocks share the activation slab in L2)
    const int n0 = a.n_begin + blockIdx.x * BN;
    const int m0 = blockIdx.y * BM;
    constexpr bool vmode = VMODE;

    // ---- per-thread staging metadata ------------------------------------------
    const int cchunk = tid & 3;  // NT % 4 == 0, so every chunk of a thread has the same c
    int xrow[XCH];               // local row
    int xb[XCH], xy[XCH], xx[XCH];
    bool xok[XCH];
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
        const int q = tid + j * NT;
        xrow[j] = q >> 2;
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
        wrow[j] = q >> 2;
        wok[j] = (wrow[j] < BN) && (n0 + wrow[j] < a.n_end);
    }

    half8 xreg[XCH], wreg[WCH];

    auto stage_load = [&](int slab) {
        // slab -> (tap, channel slab)
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

    f32x16 acc[TTW][NCT];
#pragma unroll
    for (int t = 0; t < TTW; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[t][c] = zero_f16v();

    stage_load(0);
    stage_write(0);
    __syncthreads();

    int cur = 0;
    for (int slab = 0; slab < a.nslab; ++slab) {
        const bool more = slab + 1 < a.nslab;
        if (more) stage_load(slab + 1);
        const char* xs = smem + cur * (XBYTES + WBYTES);
        const char* ws = xs + XBYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 xf[TTW], wf[NCT];
#pragma unroll
            for (int t = 0; t < TTW; ++t)
                xf[t] = *reinterpret_cast<const half8*>(xs + lds_off((wt * TTW + t) * 32 + idx, ks * 2 + hi));
#pragma unroll
            for (int c = 0; c < NCT; ++c)
                wf[c] = *reinterpret_cast<const half8*>(ws + lds_off((wc * NCT + c) * 32 + idx, ks * 2 + hi));
#pragma unroll
            for (int t = 0; t < TTW; ++t)
#pragma unroll
                for (int c = 0; c < NCT; ++c)
                    acc[t][c] = vmode ? mfma32(xf[t], wf[c], acc[t][c]) : mfma32(wf[c], xf[t], acc[t][c]);
        }
        if (more) stage_write(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue ---------------------------------------------------------------
    if constexpr (vmode) {
        // D rows = tokens, cols = channels: lane owns channel n, 4 consecutive tokens per quad
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const int n = n0 + (wc * NCT + c) * 32 + idx;
            if (n >= a.n_end) continue;
            const int cg = n - a.n_vt0;
            const int head = cg / a.dhead, dd = cg - head * a.dhead;
            const float bv = a.bias ? (float)a.bias[n] : 0.f;
#pragma unroll
            for (int t = 0; t < TTW; ++t) {
                const int mt = m0 + (wt * TTW + t) * 32;  // token tile base (multiple of 32)
                if (mt >= a.M) continue;
                const int b = mt / a.ntok, tok0 = mt - b * a.ntok;
                half_t* dst = a.vt + ((long)(b * a.heads + head) * a.dvp + dd) * a.ntok + tok0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    half4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (half_t)(acc[t][c][j * 4 + i] + bv);
                    // tokens 8j + 4hi + i  ->  pos = 16*(j>>1) + 8hi + 4(j&1) + i
                    *reinterpret_cast<half4*>(dst + 16 * (j >> 1) + 8 * hi + 4 * (j & 1)) = o;
                }
            }
        }
        return;
    }

    if (a.mode == 1) {
        // GEGLU: even channel tile = value, odd = gate (weights interleaved by the host)
#pragma unroll
        for (int t = 0; t < TTW; ++t) {
            const int m = m0 + (wt * TTW + t) * 32 + idx;
            if (m >= a.M) continue;
#pragma unroll
            for (int c = 0; c + 1 < NCT; c += 2) {
                const int nt = n0 + (wc * NCT + c) * 32;  // interleaved column of the value tile
                if (nt >= a.n_end) continue;
                const int no = nt >> 1;                   // output column base
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r0 = 8 * j + 4 * hi;
                    half4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[t][c][j * 4 + i], g = acc[t][c + 1][j * 4 + i];
                        if (a.bias) { v += (float)a.bias[nt + r0 + i]; g += (float)a.bias[nt + 32 + r0 + i]; }
                        o[i] = (half_t)(v * gelu_erf_f(g));
                    }
                    *reinterpret_cast<half4*>(a.out + (long)m * a.ldo + no + r0) = o;
                }
            }
        }
        return;
    }

    // plain: lane owns token m (column), 4 consecutive channels per quad
#pragma unroll
    for (int t = 0; t < TTW; ++t) {
        const int m = m0 + (wt * TTW + t) * 32 + idx;
        if (m >= a.M) continue;
        const half_t* rb = a.rowbias ? a.rowbias + (long)(m / a.rows_per_sample) * a.ld_rowbias : nullptr;
        const half_t* rs = a.res ? a.res + (long)m * a.ldr : nullptr;
        half_t* op = a.out + (long)m * a.ldo;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const int nt = n0 + (wc * NCT + c) * 32;
            if (nt >= a.n_end) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = nt + 8 * j + 4 * hi;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[t][c][j * 4 + i];
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

template <int TTW, int NCT, int WT, int WC>
int launch(GemmArgs a, hipStream_t s) {
    constexpr int BM = 32 * TTW * WT, BN = 32 * NCT * WC;
    const int n_plain = (a.mode == 2) ? a.n_vt0 : a.N;
    a.n_begin = 0; a.n_end = n_plain;
    dim3 grid((n_plain + BN - 1) / BN, (a.M + BM - 1) / BM);
    hipLaunchKernelGGL((igemm_kernel<TTW, NCT, WT, WC, false>), grid, dim3(64 * WT * WC), 0, s, a);
    if (a.mode == 2 && a.N > a.n_vt0) {
        // the V third: flipped operand roles, transposed store
        a.n_begin = a.n_vt0; a.n_end = a.N;
        dim3 gridv((a.N - a.n_vt0 + BN - 1) / BN, (a.M + BM - 1) / BM);
        hipLaunchKernelGGL((igemm_kernel<TTW, NCT, WT, WC, true>), gridv, dim3(64 * WT * WC), 0, s, a);
    }
    return 0;
}

}  // namespace

extern "C" int cid_gemm_f16(const cid_gemm_desc* d, cid_stream_t stream) {
    CID_CHECK_ARG(d && d->x1 && d->w && d->out, "cid_gemm_f16: null pointer");
    CID_CHECK_ARG(d->taps == 1 || d->taps == 9, "cid_gemm_f16: taps must be 1 or 9 (got %d)", d->taps);
    CID_CHECK_ARG(d->c1 > 0 && d->c1 % 32 == 0 && d->c2 >= 0 && d->c2 % 32 == 0,
                  "cid_gemm_f16: channel counts must be multiples of 32 (c1=%d c2=%d)", d->c1, d->c2);
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
    a.cslabs = (d->c1 + d->c2) / 32;
    a.ktot = d->taps * (d->c1 + d->c2);
    a.nslab = a.ktot / 32;
    if (d->taps == 9) {
        CID_CHECK_ARG(d->Hi > 0 && d->Wi > 0 && d->Ho > 0 && d->Wo > 0 && (d->stride == 1 || d->stride == 2)
                      && (d->up == 0 || d->up == 1), "cid_gemm_f16: bad conv geometry");
        CID_CHECK_ARG(d->M % (d->Ho * d->Wo) == 0, "cid_gemm_f16: M is not batch * Ho * Wo");
    }
    if (d->mode == 1) CID_CHECK_ARG(d->N % 64 == 0, "cid_gemm_f16: GEGLU needs N %% 64 == 0");
    // Tile choice: fill >= ~1 block per CU when the problem allows it.
    enum { T128x320, T128x160, T64x160, T64x64, G128x256, G64x128 } cfg;
    int bn;
    if (d->mode == 1) {
        // GEGLU needs an even channel-tile count per wave
        const long tb = (long)((a.N + 255) / 256) * ((a.M + 127) / 128);
        if (tb >= 200) { cfg = G128x256; bn = 256; } else { cfg = G64x128; bn = 128; }
    } else if (a.N % 160 != 0) {
        cfg = T64x64; bn = 64;      // odd widths (tests, tiny UNet)
    } else {
        const long tiles_big = (long)(a.N / 320) * ((a.M + 127) / 128);
        const long tiles_mid = (long)(a.N / 160) * ((a.M + 127) / 128);
        if (a.N % 320 == 0 && tiles_big >= 200) { cfg = T128x320; bn = 320; }
        else if (tiles_mid >= 200)              { cfg = T128x160; bn = 160; }
        else                                    { cfg = T64x160;  bn = 160; }
    }
    if (d->mode == 2) {
        CID_CHECK_ARG(d->vt && d->ntok % 32 == 0 && d->M % d->ntok == 0 && d->n_vt0 % bn == 0 && d->dhead > 0
                      && d->heads > 0 && d->dvp >= d->dhead,
                      "cid_gemm_f16: bad QKV/V^T description");
    }
    hipStream_t s = (hipStream_t)stream;
    switch (cfg) {
        case T128x320: launch<2, 5, 2, 2>(a, s); break;
        case T128x160: launch<2, 5, 2, 1>(a, s); break;
        case T64x160:  launch<1, 5, 2, 1>(a, s); break;
        case T64x64:   launch<1, 2, 2, 1>(a, s); break;
        case G128x256: launch<2, 4, 2, 2>(a, s); break;
        case G64x128:  launch<1, 4, 2, 1>(a, s); break;
    }
    CID_CHECK_LAUNCH("cid_gemm_f16");
    return 0;
}
