// Fused identity cross-attention: Consistent_IPAttProcessor.__call__
// (/root/reference/attention.py:207-294) in ONE launch, LoRA merged:
//
//   q   = LN?(x) Wq'^T                                   (:236)
//   o_h = softmax_{text keys}(q_h K_h^T) V_h
//       + ip_scale * softmax_{ID keys}(q_h Kip_h^T) Vip_h (:259-279, two separate softmaxes)
//   out = o Wo'^T + b_o (+ residual)                     (:282)
//
// Roofline: MFMA-bound (dense fp16 2.5 PFLOP/s).  Algorithmic work per (sample, layer):
// 4 N C^2 + 4 N L C flop, bytes 4 N C (x in, out) + 4 C^2 (weights) + 4 L C (K, V).
//
// One workgroup (4 waves) owns BT tokens of one sample; the [BT][C] tile lives in LDS
// and is reused in place for x -> q -> o, so x is read from HBM once and out written
// once.  All three contractions run "transposed" (MFMA A = weight / K / V^T rows,
// B = token rows) so the token axis is always the lane axis:
//   stage 1  Q^T = Wq' T^T        A: Wq' fragments (pre-packed, one coalesced 1 KiB load
//                                    per fragment, L2-resident), B: T rows (LDS)
//   stage 2  per head: S^T = K_h Q_h^T over 96 key slots (77 text + 4 ID + pad), the two
//            softmaxes are taken over their own key ranges inside the SAME score tile
//            (lane-local + one lane^32 exchange), P^T feeds O^T = V_h^T P^T straight from
//            the accumulator registers (V^T is packed with the matching key permutation)
//   stage 3  out^T = Wo' T^T + bias (+ residual), 8-byte coalesced stores.
// K / V^T of both streams are pre-projected and pre-packed once per (layer, embed set)
// by cid_kv_pack_f16: encoder_hidden_states does not change across denoising steps.
#include "common.h"
#include "../../include/cid.h"

namespace {

constexpr int KTILES = 3;          // 96 key slots
constexpr int PV_KSTEPS = 6;       // 96 / 16

template <int C, int D, int BT, int TTW>
struct XCfg {
    static constexpr int NH = C / D;
    static constexpr int WT = BT / (32 * TTW);
    static constexpr int WC = 4 / WT;
    static constexpr int NCT = (C / 32) / WC;
    static constexpr int KK = C / 16;                     // k-steps of the projections
    static constexpr int DKP = (D + 15) / 16 * 16;
    static constexpr int QKS = DKP / 16;                  // k-steps of Q K^T
    static constexpr int DVT = (D + 31) / 32;
    static constexpr int TP = C + 8;                      // LDS row pitch (halfs): odd slot count
    static constexpr int SMEM = BT * TP * 2;
    static constexpr long KROW = (long)NH * KTILES * QKS * 512;      // halfs per packed K row
    static constexpr long VROW = (long)NH * DVT * PV_KSTEPS * 512;   // halfs per packed V row
    static_assert(WT * WC == 4 && NCT * WC * 32 == C, "bad wave layout");
};

template <int C, int D, int BT, int TTW>
__global__ void __launch_bounds__(256)
id_xattn_kernel(const half_t* __restrict__ x, half_t* __restrict__ out, const half_t* __restrict__ residual,
                const half_t* __restrict__ ln_g, const half_t* __restrict__ ln_b, float ln_eps,
                const half_t* __restrict__ wq, const half_t* __restrict__ wo, const half_t* __restrict__ bo,
                const half_t* __restrict__ kp, const half_t* __restrict__ vp, const int* __restrict__ kvrow,
                int N, int n_txt, int n_ip, float ip_scale) {
    using Cfg = XCfg<C, D, BT, TTW>;
    constexpr int NCT = Cfg::NCT, KK = Cfg::KK, TP = Cfg::TP, WC = Cfg::WC, QKS = Cfg::QKS, DVT = Cfg::DVT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* T = reinterpret_cast<half_t*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = lane & 31, hi = lane >> 5;
    const int wt = wave / WC, wc = wave % WC;
    const int sample = blockIdx.y;
    const long tok0 = (long)sample * N + (long)blockIdx.x * BT;   // first token row of this tile

    // ------------------------------------------------ stage 0: x (-> LayerNorm) -> T
    {
        constexpr int NCH = C / 8;
        constexpr int PER = (NCH + 63) / 64;
        for (int r = wave; r < BT; r += 4) {
            const half_t* xr = x + (tok0 + r) * C;
            float v[PER][8];
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int c = lane + k * 64;
                if (c < NCH) {
                    const half8 hh = ld_global_h8(xr + c * 8);
#pragma unroll
                    for (int i = 0; i < 8; ++i) { v[k][i] = (float)hh[i]; s += v[k][i]; }
                }
            }
            if (ln_g) {
                const float mean = wave_sum(s) * (1.f / C);
                float qq = 0.f;
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int c = lane + k * 64;
                    if (c < NCH) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { const float dlt = v[k][i] - mean; qq += dlt * dlt; }
                    }
                }
                const float rstd = rsqrtf(wave_sum(qq) * (1.f / C) + ln_eps);
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int c = lane + k * 64;
                    if (c < NCH) {
                        const half8 g = ld_global_h8(ln_g + c * 8), bb = ld_global_h8(ln_b + c * 8);
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[k][i] = (v[k][i] - mean) * rstd * (float)g[i] + (float)bb[i];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int c = lane + k * 64;
                if (c < NCH) {
                    half8 o;
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = (half_t)v[k][i];
                    *reinterpret_cast<half8*>(T + r * TP + c * 8) = o;
                }
            }
            if (lane == 0) *reinterpret_cast<half8*>(T + r * TP + C) = zero_h8();   // pad columns
        }
    }
    __syncthreads();

    // ------------------------------------------------ projection: acc^T = W' T^T
    f32x16 acc[TTW][NCT];
    auto project = [&](const half_t* __restrict__ wpk) {
#pragma unroll
        for (int t = 0; t < TTW; ++t)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[t][c] = zero_f16v();
        const half_t* wl = wpk + ((long)(wc * NCT) * KK) * 512 + lane * 8;
        half8 anext[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) anext[c] = ld_global_h8(wl + ((long)c * KK) * 512);
        for (int kk = 0; kk < KK; ++kk) {
            half8 acur[NCT];
#pragma unroll
            for (int c = 0; c < NCT; ++c) acur[c] = anext[c];
            if (kk + 1 < KK) {
#pragma unroll
                for (int c = 0; c < NCT; ++c) anext[c] = ld_global_h8(wl + ((long)c * KK + kk + 1) * 512);
            }
            half8 bf[TTW];
#pragma unroll
            for (int t = 0; t < TTW; ++t)
                bf[t] = *reinterpret_cast<const half8*>(T + ((wt * TTW + t) * 32 + idx) * TP + kk * 16 + hi * 8);
#pragma unroll
            for (int t = 0; t < TTW; ++t)
#pragma unroll
                for (int c = 0; c < NCT; ++c) acc[t][c] = mfma32(acur[c], bf[t], acc[t][c]);
        }
    };

    // ------------------------------------------------ stage 1: Q^T = Wq' T^T, Q -> T
    project(wq);
    __syncthreads();   // every wave is done reading x from T
#pragma unroll
    for (int t = 0; t < TTW; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                half4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (half_t)acc[t][c][j * 4 + i];
                *reinterpret_cast<half4*>(T + ((wt * TTW + t) * 32 + idx) * TP + (wc * NCT + c) * 32 + 8 * j + 4 * hi) = o;
            }
    __syncthreads();

    // ------------------------------------------------ stage 2: two-stream attention per head
    {
        const long row = kvrow[sample];
        const half_t* kpr = kp + row * Cfg::KROW + lane * 8;
        const half_t* vpr = vp + row * Cfg::VROW + lane * 8;
        const int n_all = n_txt + n_ip;
        for (int h = wc; h < Cfg::NH; h += WC) {
            // Q_h^T fragments (B operand); columns beyond D hit K's zero padding
            half8 qf[TTW][QKS];
#pragma unroll
            for (int t = 0; t < TTW; ++t)
#pragma unroll
                for (int kk = 0; kk < QKS; ++kk)
                    qf[t][kk] = *reinterpret_cast<const half8*>(T + ((wt * TTW + t) * 32 + idx) * TP + h * D + kk * 16 + hi * 8);
            f32x16 s[KTILES][TTW];
#pragma unroll
            for (int kt = 0; kt < KTILES; ++kt)
#pragma unroll
                for (int t = 0; t < TTW; ++t) s[kt][t] = zero_f16v();
#pragma unroll
            for (int kt = 0; kt < KTILES; ++kt)
#pragma unroll
                for (int kk = 0; kk < QKS; ++kk) {
                    const half8 kf = ld_global_h8(kpr + ((long)(h * KTILES + kt) * QKS + kk) * 512);
#pragma unroll
                    for (int t = 0; t < TTW; ++t) s[kt][t] = mfma32(kf, qf[t][kk], s[kt][t]);
                }
            // two independent softmaxes over [0, n_txt) and [n_txt, n_all)
            half8 pf[TTW][PV_KSTEPS];
#pragma unroll
            for (int t = 0; t < TTW; ++t) {
                float mt = -INFINITY, mi = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < KTILES; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * 32 + crow(r, hi);
                        const float v = s[kt][t][r];
                        if (key < n_txt) mt = fmaxf(mt, v);
                        else if (key < n_all) mi = fmaxf(mi, v);
                    }
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
                mi = fmaxf(mi, __shfl_xor(mi, 32, 64));
                float lt = 0.f, li = 0.f;
#pragma unroll
                for (int kt = 0; kt < KTILES; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * 32 + crow(r, hi);
                        float p = 0.f;
                        if (key < n_txt) { p = __builtin_amdgcn_exp2f(s[kt][t][r] - mt); lt += p; }
                        else if (key < n_all) { p = __builtin_amdgcn_exp2f(s[kt][t][r] - mi); li += p; }
                        s[kt][t][r] = p;
                    }
                lt += __shfl_xor(lt, 32, 64);
                li += __shfl_xor(li, 32, 64);
                const float it = 1.f / lt;
                const float ii = (n_ip > 0) ? ip_scale / li : 0.f;
#pragma unroll
                for (int kt = 0; kt < KTILES; ++kt)
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        half8 pv;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int r = g * 8 + i;
                            const int key = kt * 32 + crow(r, hi);
                            pv[i] = (half_t)(s[kt][t][r] * (key < n_txt ? it : ii));
                        }
                        pf[t][kt * 2 + g] = pv;
                    }
            }
            // O_h^T = V_h^T P^T
            f32x16 o[DVT][TTW];
#pragma unroll
            for (int d = 0; d < DVT; ++d)
#pragma unroll
                for (int t = 0; t < TTW; ++t) o[d][t] = zero_f16v();
#pragma unroll
            for (int d = 0; d < DVT; ++d)
#pragma unroll
                for (int ks = 0; ks < PV_KSTEPS; ++ks) {
                    const half8 vf = ld_global_h8(vpr + ((long)(h * DVT + d) * PV_KSTEPS + ks) * 512);
#pragma unroll
                    for (int t = 0; t < TTW; ++t) o[d][t] = mfma32(vf, pf[t][ks], o[d][t]);
                }
            // O_h -> T (over Q_h; only this wave touches these rows x columns)
#pragma unroll
            for (int t = 0; t < TTW; ++t)
#pragma unroll
                for (int d = 0; d < DVT; ++d)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int dd = d * 32 + 8 * j + 4 * hi;
                        if (dd < D) {
                            half4 ov;
#pragma unroll
                            for (int i = 0; i < 4; ++i) ov[i] = (half_t)o[d][t][j * 4 + i];
                            *reinterpret_cast<half4*>(T + ((wt * TTW + t) * 32 + idx) * TP + h * D + dd) = ov;
                        }
                    }
        }
    }
    __syncthreads();

    // ------------------------------------------------ stage 3: out^T = Wo' T^T + b (+ res)
    project(wo);
#pragma unroll
    for (int t = 0; t < TTW; ++t) {
        const long m = tok0 + (wt * TTW + t) * 32 + idx;
        half_t* op = out + m * C;
        const half_t* rp = residual ? residual + m * C : nullptr;
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = (wc * NCT + c) * 32 + 8 * j + 4 * hi;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[t][c][j * 4 + i];
                if (bo) {
                    const half4 bb = *reinterpret_cast<const half4*>(bo + n);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)bb[i];
                }
                if (rp) {
                    const half4 rr = *reinterpret_cast<const half4*>(rp + n);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)rr[i];
                }
                half4 ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[i] = (half_t)v[i];
                *reinterpret_cast<half4*>(op + n) = ov;
            }
    }
}

template <int C, int D, int BT, int TTW>
int launch_xattn(const half_t* x, half_t* out, const half_t* residual, const half_t* g, const half_t* bta, float eps,
                 const half_t* wq, const half_t* wo, const half_t* bo, const half_t* kp, const half_t* vp,
                 const int* kvrow, int B, int N, int n_txt, int n_ip, float ip_scale, hipStream_t s) {
    using Cfg = XCfg<C, D, BT, TTW>;
    if (N % BT != 0) {
        cid_set_error("cid_id_xattn_f16: N=%d is not a multiple of the token tile %d (C=%d)", N, BT, C);
        return -22;
    }
    auto kern = id_xattn_kernel<C, D, BT, TTW>;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM) != hipSuccess) {
            cid_set_error("cid_id_xattn_f16: cannot reserve %d bytes of LDS", Cfg::SMEM);
            return -5;
        }
        configured = true;
    }
    hipLaunchKernelGGL(kern, dim3(N / BT, B), dim3(256), Cfg::SMEM, s, x, out, residual, g, bta, eps, wq, wo, bo,
                       kp, vp, kvrow, N, n_txt, n_ip, ip_scale);
    return 0;
}

// (C, heads) -> instantiation table
#define CID_XATTN_CONFIGS(X)   \
    X(320, 40, 128, 2)         \
    X(640, 80, 64, 2)          \
    X(1280, 160, 32, 1)        \
    X(640, 64, 64, 2)          \
    X(1280, 64, 32, 1)         \
    X(64, 32, 64, 1)           \
    X(128, 64, 64, 1)

__global__ void __launch_bounds__(256)
kv_pack_kernel(const half_t* __restrict__ kv_txt, const half_t* __restrict__ kv_ip, half_t* __restrict__ kp,
               half_t* __restrict__ vp, int R, int C, int heads, int n_txt, int n_ip, long krow, long vrow) {
    const int D = C / heads, L = n_txt + n_ip;
    const int QKS = (D + 15) / 16, DVT = (D + 31) / 32;
    const long kchunks = krow / 8, vchunks = vrow / 8;            // 16-B chunks per packed row
    const long total = (long)R * (kchunks + vchunks);
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const int r = (int)(q / (kchunks + vchunks));
        long e = q - (long)r * (kchunks + vchunks);
        half8 o = zero_h8();
        if (e < kchunks) {
            // [h][kt][kk][lane]
            const int lane = (int)(e & 63); e >>= 6;
            const int kk = (int)(e % QKS); e /= QKS;
            const int kt = (int)(e % KTILES); const int h = (int)(e / KTILES);
            const int key = kt * 32 + (lane & 31);
            const int dc = kk * 16 + (lane >> 5) * 8;
            if (key < L && dc < D) {
                const half_t* src = (key < n_txt ? kv_txt : kv_ip) + ((long)r * L + key) * 2 * C + h * D + dc;
                o = ld_global_h8(src);   // D % 8 == 0, so a chunk never straddles the head
            }
            *reinterpret_cast<half8*>(kp + (long)r * krow + (q - (long)r * (kchunks + vchunks)) * 8) = o;
        } else {
            e -= kchunks;
            const long e0 = e;
            // [h][dt][ks][lane]
            const int lane = (int)(e & 63); e >>= 6;
            const int ks = (int)(e % PV_KSTEPS); e /= PV_KSTEPS;
            const int dt = (int)(e % DVT); const int h = (int)(e / DVT);
            const int d = dt * 32 + (lane & 31), hi = lane >> 5;
            if (d < D) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int key = ks * 16 + 4 * hi + (i & 3) + 8 * (i >> 2);
                    if (key < L)
                        o[i] = ((key < n_txt ? kv_txt : kv_ip) + ((long)r * L + key) * 2 * C + C + h * D)[d];
                }
            }
            *reinterpret_cast<half8*>(vp + (long)r * vrow + e0 * 8) = o;
        }
    }
}

__global__ void __launch_bounds__(256)
pack_wfrag_kernel(const half_t* __restrict__ w, half_t* __restrict__ wp, int rows, int K) {
    const long total = (long)rows * K / 8;
    const int KK = K / 16;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        long e = q;
        const int lane = (int)(e & 63); e >>= 6;
        const int kk = (int)(e % KK); const int rt = (int)(e / KK);
        *reinterpret_cast<half8*>(wp + q * 8) =
            ld_global_h8(w + (long)(rt * 32 + (lane & 31)) * K + kk * 16 + (lane >> 5) * 8);
    }
}

}  // namespace

extern "C" int64_t cid_kv_pack_elems(int32_t C, int32_t heads, int32_t which) {
    if (heads <= 0 || C % heads) return -22;
    const int D = C / heads;
    const int QKS = (D + 15) / 16, DVT = (D + 31) / 32;
    return which == 0 ? (int64_t)heads * KTILES * QKS * 512 : (int64_t)heads * DVT * PV_KSTEPS * 512;
}

extern "C" int cid_kv_pack_f16(const cid_half* kv_txt, const cid_half* kv_ip, cid_half* kp, cid_half* vp,
                               int32_t R, int32_t C, int32_t heads, int32_t n_txt, int32_t n_ip, cid_stream_t stream) {
    CID_CHECK_ARG(kv_txt && kv_ip && kp && vp, "cid_kv_pack_f16: null pointer");
    CID_CHECK_ARG(R > 0 && heads > 0 && C % heads == 0 && (C / heads) % 8 == 0, "cid_kv_pack_f16: bad C/heads");
    CID_CHECK_ARG(n_txt > 0 && n_ip >= 0 && n_txt + n_ip <= 32 * KTILES, "cid_kv_pack_f16: at most %d context rows", 32 * KTILES);
    const long krow = cid_kv_pack_elems(C, heads, 0), vrow = cid_kv_pack_elems(C, heads, 1);
    const long total = (long)R * (krow + vrow) / 8;
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(kv_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const half_t*)kv_txt,
                       (const half_t*)kv_ip, (half_t*)kp, (half_t*)vp, R, C, heads, n_txt, n_ip, krow, vrow);
    CID_CHECK_LAUNCH("cid_kv_pack_f16");
    return 0;
}

extern "C" int cid_pack_wfrag_f16(const cid_half* w, cid_half* wp, int32_t rows, int32_t K, cid_stream_t stream) {
    CID_CHECK_ARG(w && wp && rows > 0 && rows % 32 == 0 && K > 0 && K % 16 == 0, "cid_pack_wfrag_f16: rows %% 32, K %% 16 required");
    const long total = (long)rows * K / 8;
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_wfrag_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const half_t*)w, (half_t*)wp, rows, K);
    CID_CHECK_LAUNCH("cid_pack_wfrag_f16");
    return 0;
}

extern "C" int cid_id_xattn_f16(const cid_half* x, cid_half* out, const cid_half* residual,
                                const cid_half* ln_gamma, const cid_half* ln_beta, float ln_eps,
                                const cid_half* wq, const cid_half* wo, const cid_half* bo,
                                const cid_half* kp, const cid_half* vp, const int32_t* kvrow,
                                int32_t B, int32_t N, int32_t C, int32_t heads,
                                int32_t n_txt, int32_t n_ip, float ip_scale, cid_stream_t stream) {
    CID_CHECK_ARG(x && out && wq && wo && kp && vp && kvrow, "cid_id_xattn_f16: null pointer");
    CID_CHECK_ARG((ln_gamma == nullptr) == (ln_beta == nullptr), "cid_id_xattn_f16: gamma/beta must come together");
    CID_CHECK_ARG(B > 0 && N > 0 && heads > 0 && C % heads == 0, "cid_id_xattn_f16: bad shape");
    CID_CHECK_ARG(n_txt > 0 && n_ip >= 0 && n_txt + n_ip <= 32 * KTILES, "cid_id_xattn_f16: at most %d context rows", 32 * KTILES);
    const int D = C / heads;
    int rc = -22;
    bool found = false;
#define CID_X(CC, DD, BT, TTW)                                                                              \
    if (!found && C == CC && D == DD) {                                                                     \
        found = true;                                                                                       \
        rc = launch_xattn<CC, DD, BT, TTW>((const half_t*)x, (half_t*)out, (const half_t*)residual,         \
                                           (const half_t*)ln_gamma, (const half_t*)ln_beta, ln_eps,         \
                                           (const half_t*)wq, (const half_t*)wo, (const half_t*)bo,         \
                                           (const half_t*)kp, (const half_t*)vp, kvrow, B, N, n_txt, n_ip,  \
                                           ip_scale, (hipStream_t)stream);                                  \
    }
    CID_XATTN_CONFIGS(CID_X)
#undef CID_X
    if (!found) {
        cid_set_error("cid_id_xattn_f16: no kernel for C=%d head_dim=%d", C, D);
        return -22;
    }
    if (rc) return rc;
    CID_CHECK_LAUNCH("cid_id_xattn_f16");
    return 0;
}
