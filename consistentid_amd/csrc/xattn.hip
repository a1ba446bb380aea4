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
// One workgroup (8 waves) owns BT tokens of one sample; the [BT][C] tile lives in LDS
// and is reused in place for x -> q -> o, so x is read from HBM once and out written
// once.  All three contractions run "transposed" (MFMA A = weight / K / V^T rows,
// B = token rows) so the token axis is always the lane axis:
//   stage 1  Q^T = Wq' T^T        A: [320 x 32] slabs of Wq' streamed L2 -> LDS by DMA through a
//                                    3-stage ring (16x16x32 MFMA, wave tile 64 tok x 80 ch),
//                                    B: T rows (LDS); Q stays in registers until all of x is consumed
//   stage 2  per head: S^T = K_h Q_h^T over 96 key slots (77 text + 4 ID + pad), the two
//            softmaxes are taken over their own key ranges inside the SAME score tile
//            (lane-local + one lane^32 exchange), P^T feeds O^T = V_h^T P^T straight from
//            the accumulator registers (V^T is packed with the matching key permutation)
//   stage 3  out^T = Wo' T^T + bias (+ residual), 8-byte coalesced stores.
// K / V^T of both streams are pre-projected and pre-packed once per (layer, embed set)
// by cid_kv_pack_f16: encoder_hidden_states does not change across denoising steps.
#include "xattn_core.h"
#include "../../include/cid.h"
#include <stdlib.h>

namespace {

constexpr int KTILES = XC_KTILES;          // 96 key slots
constexpr int PV_KSTEPS = XC_PV_KSTEPS;    // 96 / 16

typedef float f32x4v __attribute__((ext_vector_type(4)));
CID_DEVINL f32x4v mfma16(half8 a, half8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <int C, int D, int BT>
struct XCfg {
    static constexpr int NH = C / D;
    static constexpr int CHUNK = C < 320 ? C : 320;       // output channels per projection pass
    static constexpr int NCHUNK = C / CHUNK;
    static constexpr int TM = BT / 32;                    // 16-token tiles per wave (2 wave rows)
    static constexpr int TN = CHUNK / 64;                 // 16-channel tiles per wave (4 wave columns)
    static constexpr int KS = C / 32;                     // 32-deep weight slabs per pass
    static constexpr int WPW = (CHUNK / 16 + 7) / 8;      // 1-KiB DMA pieces (16 rows x 64 B) per wave per slab
    static constexpr int WSTAGE = WPW * 8 * 1024;         // bytes per ring stage (incl. zero-filled scratch rows)
    static constexpr int TP = C + 8;                      // T row pitch (halfs): odd number of 16-B slots
    static constexpr int TBYTES = BT * TP * 2;
    static constexpr int NSTG_FIT = (160 * 1024 - TBYTES) / WSTAGE;       // ring depth: what LDS leaves
    static constexpr int NSTG = NSTG_FIT > 6 ? 6 : NSTG_FIT;
    static constexpr int SMEM = TBYTES + NSTG * WSTAGE;
    static constexpr int TTW = BT >= 64 ? 2 : 1;          // 32-token tiles per attention unit
    static constexpr int NTG = BT / (32 * TTW);
    static constexpr int DKP = (D + 15) / 16 * 16;
    static constexpr int QKS = DKP / 16;                  // k-steps of Q K^T
    static constexpr int DVT = (D + 31) / 32;
    static constexpr long KROW = (long)NH * KTILES * QKS * 512;      // halfs per packed K row
    static constexpr long VROW = (long)NH * DVT * PV_KSTEPS * 512;   // halfs per packed V row
    static_assert(C % CHUNK == 0 && CHUNK % 64 == 0 && BT % 32 == 0 && SMEM <= 160 * 1024 && NSTG >= 3 && 4 * WPW <= 63, "bad tiling");
};

// STD: the context layout is the reference's (77 text + 4 ID tokens, attention.py:241 with
// num_tokens = 4), known at compile time so the key-range predicates of the softmax fold away.
template <int C, int D, int BT, bool STD>
__global__ void __launch_bounds__(512, 2)
id_xattn_kernel(const half_t* __restrict__ x, half_t* __restrict__ out, const half_t* __restrict__ residual,
                const half_t* __restrict__ ln_g, const half_t* __restrict__ ln_b, float ln_eps,
                const half_t* __restrict__ wq, const half_t* __restrict__ wo, const half_t* __restrict__ bo,
                const half_t* __restrict__ kp, const half_t* __restrict__ vp, const int* __restrict__ kvrow,
                int N, int n_txt_rt, int n_ip_rt, float ip_scale, int core_only_flag, int ablate) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = XCfg<C, D, BT>;
    const int n_txt = STD ? 77 : n_txt_rt;
    const int n_ip = STD ? 4 : n_ip_rt;
    constexpr int CHUNK = Cfg::CHUNK, NCHUNK = Cfg::NCHUNK, TM = Cfg::TM, TN = Cfg::TN, KS = Cfg::KS;
    constexpr int WPW = Cfg::WPW, WSTAGE = Cfg::WSTAGE, NSTG = Cfg::NSTG, TP = Cfg::TP;
    constexpr int TTW = Cfg::TTW, NTG = Cfg::NTG, QKS = Cfg::QKS, DVT = Cfg::DVT;
    constexpr int G = 2 * NCHUNK * KS;                    // weight slabs of both projections
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    half_t* T = reinterpret_cast<half_t*>(smem);
    char* ring = smem + Cfg::TBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int idx = lane & 31, hi = lane >> 5;            // 32x32x16 fragment coordinates (attention)
    const int l16 = lane & 15, lq = lane >> 4;            // 16x16x32 fragment coordinates (projections)
    const int wm = wave >> 2, wn = wave & 3;
    const int sample = blockIdx.y;
    const long tok0 = (long)sample * N + (long)blockIdx.x * BT;   // first token row of this tile

    // ---- weight streaming: [CHUNK x 32] slabs of Wq' then Wo' through a 3-stage LDS ring by DMA --------
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc((void*)wq, 0, C * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)wo, 0, C * C * 2, 0x00020000);
    unsigned wbase[WPW];
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int R = (j * 8 + wave) * 16 + (lane >> 2);          // row inside the chunk
        const int cl = (lane & 3) ^ ((R >> 2) & 3);               // source-side swizzle (see lds_w)
        wbase[j] = (R < CHUNK) ? (unsigned)((R * C + cl * 8) * 2) : OOB;
    }
    auto issue_w = [&](int g) {
        const int P = g / (NCHUNK * KS);
        const int rem = g - P * (NCHUNK * KS);
        const int ch = rem / KS, kk = rem - ch * KS;
        char* st = ring + (g % NSTG) * WSTAGE;
        const unsigned off = (unsigned)((ch * CHUNK * C + kk * 32) * 2);
        if (P == 0) {
#pragma unroll
            for (int j = 0; j < WPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (lds_void*)(st + (j * 8 + wave) * 1024), 16, wbase[j] + off, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < WPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_o, (lds_void*)(st + (j * 8 + wave) * 1024), 16, wbase[j] + off, 0, 0, 0);
        }
    };
    // ring row r (64 B, 4 chunks): logical chunk c lives at chunk c ^ ((r >> 2) & 3)
    auto lds_w = [&](int stage, int r, int c) -> const half8* {
        return reinterpret_cast<const half8*>(ring + stage * WSTAGE + r * 64 + ((c ^ ((r >> 2) & 3)) << 4));
    };

    const bool core_only = core_only_flag != 0;   // x already holds Q; write O (no projections)
    // profiling knobs exist in -DCID_XATTN_ABLATION experiment builds only; the product kernel has no such branches
#if defined(CID_XATTN_ABLATION)
#define CID_XABL(bit) ((ablate & (bit)) != 0)
#else
#define CID_XABL(bit) false
#endif
    if (!core_only) {
    #pragma unroll
        for (int i = 0; i < NSTG - 1; ++i)
            if (i < G) issue_w(i);
    }

    // ------------------------------------------------ stage 0: x (-> LayerNorm) -> T
    // each wave owns BT/8 rows; ALL of their 16-B chunks are requested before the first one is
    // consumed (one round trip to HBM per wave instead of one per row)
    {
        constexpr int NCH = C / 8;
        constexpr int PER = (NCH + 63) / 64;
        constexpr int RPW = BT / 8;
        half8 raw[RPW][PER];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const half_t* xr = x + (tok0 + wave * RPW + i) * C;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int c = lane + k * 64;
                raw[i][k] = (c < NCH) ? ld_global_h8(xr + c * 8) : zero_h8();
            }
        }
        half8 gam[PER], bet[PER];
        if (ln_g) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int c = lane + k * 64;
                gam[k] = (c < NCH) ? ld_global_h8(ln_g + c * 8) : zero_h8();
                bet[k] = (c < NCH) ? ld_global_h8(ln_b + c * 8) : zero_h8();
            }
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int r = wave * RPW + i;
            float v[PER][8];
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < PER; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[k][j] = (float)raw[i][k][j]; s += v[k][j]; }   // absent chunks are 0
            if (ln_g) {
                const float mean = wave_sum(s) * (1.f / C);
                float qq = 0.f;
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    if (lane + k * 64 < NCH) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { const float dlt = v[k][j] - mean; qq += dlt * dlt; }
                    }
                }
                const float rstd = rsqrtf(wave_sum(qq) * (1.f / C) + ln_eps);
#pragma unroll
                for (int k = 0; k < PER; ++k)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[k][j] = (v[k][j] - mean) * rstd * (float)gam[k][j] + (float)bet[k][j];
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int c = lane + k * 64;
                if (c < NCH) {
                    half8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (half_t)v[k][j];
                    *reinterpret_cast<half8*>(T + r * TP + c * 8) = o;
                }
            }
            if (lane == 0) *reinterpret_cast<half8*>(T + r * TP + C) = zero_h8();   // pad columns
        }
    }

    // ------------------------------------------------ projection pass: acc^T[ch] = W'[ch] T^T
    // one barrier per 32-deep slab: it publishes slab g (every wave waited for its own DMA pieces)
    // and retires slab g-1, whose ring stage the DMA of slab g+2 then overwrites.
    int g = 0;
    auto project_chunk = [&](f32x4v (&acc)[TM][TN]) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int c = 0; c < TN; ++c) acc[t][c] = f32x4v{0.f, 0.f, 0.f, 0.f};
        for (int kk = 0; kk < KS; ++kk, ++g) {
            // (lgkmcnt: a raw s_barrier does not drain this wave's own LDS writes of stage 0 / reads of slab g-1)
            {
                // slab g has landed once only the (at most NSTG-2) younger slabs' pieces are outstanding
                const int younger = min(NSTG - 2, G - 1 - g);
                switch (younger) {
                    case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
                    case 1: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(1 * WPW) : "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * WPW) : "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * WPW) : "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * WPW) : "memory"); break;
                }
            }
            __builtin_amdgcn_s_barrier();
            if (g + NSTG - 1 < G) issue_w(g + NSTG - 1);
            const int stage = g % NSTG;
            half8 tf[TM], wf[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t)
                tf[t] = *reinterpret_cast<const half8*>(T + ((wm * TM + t) * 16 + l16) * TP + kk * 32 + lq * 8);
#pragma unroll
            for (int c = 0; c < TN; ++c) wf[c] = *lds_w(stage, (wn * TN + c) * 16 + l16, lq);
            if (CID_XABL(2)) {   // profiling knob: keep the loads, drop the matrix work
#pragma unroll
                for (int t = 0; t < TM; ++t) asm volatile("" ::"v"(tf[t]));
#pragma unroll
                for (int c = 0; c < TN; ++c) asm volatile("" ::"v"(wf[c]));
                continue;
            }
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int c = 0; c < TN; ++c) acc[t][c] = mfma16(wf[c], tf[t], acc[t][c]);
        }
    };

    // ------------------------------------------------ stage 1: Q^T = Wq' T^T, Q -> T (in place)
    if (!core_only) {
        f32x4v qacc[NCHUNK][TM][TN];
#pragma unroll
        for (int ch = 0; ch < NCHUNK; ++ch) project_chunk(qacc[ch]);
        __syncthreads();   // every wave is done reading x from T
#pragma unroll
        for (int ch = 0; ch < NCHUNK; ++ch)
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int c = 0; c < TN; ++c) {
                    half4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (half_t)qacc[ch][t][c][i];
                    *reinterpret_cast<half4*>(T + ((wm * TM + t) * 16 + l16) * TP + ch * CHUNK + (wn * TN + c) * 16 + 4 * lq) = o;
                }
    }
    __syncthreads();

    // ------------------------------------------------ stage 2: two-stream attention, unit = (head, token group)
    {
        const long row = kvrow[sample];
        const half_t* kpr = kp + row * Cfg::KROW + lane * 8;
        const half_t* vpr = vp + row * Cfg::VROW + lane * 8;
        const int n_all = n_txt + n_ip;
        for (int u = wave; u < Cfg::NH * NTG; u += 8) {
            if (CID_XABL(1)) break;   // profiling knob: skip the attention core
            const int h = u / NTG, tg = u - h * NTG;
            xattn_core_unit<D, TTW, STD>(T, TP, tg * 32 * TTW, h * D, kpr + (long)h * KTILES * QKS * 512,
                                         vpr + (long)h * DVT * PV_KSTEPS * 512, n_txt, n_ip, ip_scale, idx, hi);
        }
    }
    __syncthreads();

    if (core_only) {
        // attention core only: O (in T) -> out, whole rows, 16 B per lane
        constexpr int NCH = C / 8;
        for (int e = tid; e < BT * NCH; e += 512) {
            const int r = e / NCH, c = e - r * NCH;
            *reinterpret_cast<half8*>(out + (tok0 + r) * C + c * 8) = *reinterpret_cast<const half8*>(T + r * TP + c * 8);
        }
        return;
    }
    // ------------------------------------------------ stage 3: out^T = Wo' T^T + b (+ res)
    if constexpr (NCHUNK == 1) {
        // One output chunk (C <= 320): the residual rows are requested BEFORE the projection loop (their HBM latency
        // hides under it) and the finished tile is transposed through the now idle LDS so that every lane stores 16 B
        // of a whole output row (the MFMA layout would give 8 B per lane in 32-B row segments).
        half4 resv[TM][TN];
        if (residual) {
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const half_t* rp = residual + (tok0 + (wm * TM + t) * 16 + l16) * C;
#pragma unroll
                for (int c = 0; c < TN; ++c) resv[t][c] = *reinterpret_cast<const half4*>(rp + (wn * TN + c) * 16 + 4 * lq);
            }
        }
        f32x4v acc[TM][TN];
        project_chunk(acc);
        __syncthreads();      // every wave is done with T and the weight ring: LDS is free
        constexpr int P = TN * 16 + 8;
        half_t* stg = reinterpret_cast<half_t*>(smem) + wave * (TM * 16 * P);
        static_assert(8 * TM * 16 * P * 2 <= Cfg::SMEM, "epilogue staging fits");
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int c = 0; c < TN; ++c) {
                const int n = (wn * TN + c) * 16 + 4 * lq;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[t][c][i];
                if (bo) {
                    const half4 bb = *reinterpret_cast<const half4*>(bo + n);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)bb[i];
                }
                if (residual) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)resv[t][c][i];
                }
                half4 ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[i] = (half_t)v[i];
                *reinterpret_cast<half4*>(stg + (t * 16 + l16) * P + c * 16 + 4 * lq) = ov;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the staging tile is private to the wave
        constexpr int CPR = TN * 2;
        constexpr int PER = (TM * 16 * CPR + 63) / 64;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = i * 64 + lane;
            if (e < TM * 16 * CPR) {
                const int r = e / CPR, cc = e - r * CPR;
                *reinterpret_cast<half8*>(out + (tok0 + wm * TM * 16 + r) * C + wn * TN * 16 + cc * 8) =
                    *reinterpret_cast<const half8*>(stg + r * P + cc * 8);
            }
        }
        return;
    }
#pragma unroll 1
    for (int ch = 0; ch < NCHUNK; ++ch) {
        f32x4v acc[TM][TN];
        project_chunk(acc);
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const long m = tok0 + (wm * TM + t) * 16 + l16;
            half_t* op = out + m * C;
            const half_t* rp = residual ? residual + m * C : nullptr;
#pragma unroll
            for (int c = 0; c < TN; ++c) {
                const int n = ch * CHUNK + (wn * TN + c) * 16 + 4 * lq;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[t][c][i];
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
#endif
}

template <int C, int D, int BT>
int launch_xattn(const half_t* x, half_t* out, const half_t* residual, const half_t* g, const half_t* bta, float eps,
                 const half_t* wq, const half_t* wo, const half_t* bo, const half_t* kp, const half_t* vp,
                 const int* kvrow, int B, int N, int n_txt, int n_ip, float ip_scale, hipStream_t s, bool core = false) {
    using Cfg = XCfg<C, D, BT>;
    if (N % BT != 0) {
        cid_set_error("cid_id_xattn_f16: N=%d is not a multiple of the token tile %d (C=%d)", N, BT, C);
        return -22;
    }
    const bool std_ctx = (n_txt == 77 && n_ip == 4);
    auto kern = std_ctx ? id_xattn_kernel<C, D, BT, true> : id_xattn_kernel<C, D, BT, false>;
    static bool configured[2] = {false, false};
    if (!configured[std_ctx]) {
        hipError_t herr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
        if (herr != hipSuccess) {
            cid_set_error("cid_id_xattn_f16: cannot reserve %d bytes of LDS (%s)", Cfg::SMEM, hipGetErrorString(herr));
            return -5;
        }
        configured[std_ctx] = true;
    }
    int ablate = 0;
#if defined(CID_XATTN_ABLATION)
    static int ablate_env = -1;
    if (ablate_env < 0) { const char* e = getenv("CID_XATTN_ABLATE"); ablate_env = e ? atoi(e) : 0; }
    ablate = ablate_env;
#endif
    hipLaunchKernelGGL(kern, dim3(N / BT, B), dim3(512), Cfg::SMEM, s, x, out, residual, g, bta, eps, wq, wo, bo,
                       kp, vp, kvrow, N, n_txt, n_ip, ip_scale, core ? 1 : 0, ablate);
    return 0;
}

// (C, head dim, tokens per workgroup) -> instantiation table
#define CID_XATTN_CONFIGS(X)   \
    X(320, 40, 128)            \
    X(640, 80, 64)             \
    X(1280, 160, 32)           \
    X(640, 64, 64)             \
    X(1280, 64, 32)            \
    X(64, 32, 64)              \
    X(128, 64, 64)

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

extern "C" int cid_id_xattn_core_f16(const cid_half* q, cid_half* out, const cid_half* kp, const cid_half* vp,
                                     const int32_t* kvrow, int32_t B, int32_t N, int32_t C, int32_t heads,
                                     int32_t n_txt, int32_t n_ip, float ip_scale, cid_stream_t stream) {
    CID_CHECK_ARG(q && out && kp && vp && kvrow, "cid_id_xattn_core_f16: null pointer");
    CID_CHECK_ARG(B > 0 && N > 0 && heads > 0 && C % heads == 0, "cid_id_xattn_core_f16: bad shape");
    CID_CHECK_ARG(n_txt > 0 && n_ip >= 0 && n_txt + n_ip <= 32 * KTILES, "cid_id_xattn_core_f16: at most %d context rows", 32 * KTILES);
    const int D = C / heads;
    int rc = -22;
    bool found = false;
#define CID_X(CC, DD, BT)                                                                                   \
    if (!found && C == CC && D == DD) {                                                                     \
        found = true;                                                                                       \
        rc = launch_xattn<CC, DD, BT>((const half_t*)q, (half_t*)out, nullptr, nullptr, nullptr, 0.f,       \
                                      (const half_t*)q, (const half_t*)q, nullptr,                          \
                                      (const half_t*)kp, (const half_t*)vp, kvrow, B, N, n_txt, n_ip,       \
                                      ip_scale, (hipStream_t)stream, true);                                 \
    }
    CID_XATTN_CONFIGS(CID_X)
#undef CID_X
    if (!found) {
        cid_set_error("cid_id_xattn_core_f16: no kernel for C=%d head_dim=%d", C, D);
        return -22;
    }
    if (rc) return rc;
    CID_CHECK_LAUNCH("cid_id_xattn_core_f16");
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
#define CID_X(CC, DD, BT)                                                                                   \
    if (!found && C == CC && D == DD) {                                                                     \
        found = true;                                                                                       \
        rc = launch_xattn<CC, DD, BT>((const half_t*)x, (half_t*)out, (const half_t*)residual,              \
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
