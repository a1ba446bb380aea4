// Two-stream attention of the identity cross-attention (Consistent_IPAttProcessor.__call__, /root/reference/attention.py:
// 259-279) for ONE unit = (head, TTW x 32 tokens) whose Q sits in an LDS tile -- shared by the first-generation fused kernel
// (xattn.hip: tile = [BT tokens][C channels]) and by the query projection with attention epilogue (gemm.hip, mode 3: tile =
// the GEMM's own [BM tokens][BN channels] output tile, BN = whole heads).
//   S^T = K_h Q_h^T over 96 key slots (n_txt text + n_ip ID + pad), the two softmaxes over their own key ranges inside the
//   SAME score tile (lane-local + one lane^32 exchange), P^T feeds O^T = V_h^T P^T straight from the accumulator registers
//   (V^T is packed with the matching key permutation by cid_kv_pack_f16); O_h overwrites Q_h in the tile.
// One wave executes a unit; K / V^T fragments come from the packed per-head images (L2-resident, 1 KiB per fragment).
#pragma once
#include "common.h"

constexpr int XC_KTILES = 3;          // 96 key slots
constexpr int XC_PV_KSTEPS = 6;       // 96 / 16

// T: the LDS tile (row pitch TP halfs); trow: first token row of the unit; col0: first column of the head inside the tile;
// kph / vph: this head's packed K / V^T fragments, already offset by lane * 8; STD: the reference's 77 + 4 context (the key
// range predicates fold away); idx = lane & 31, hi = lane >> 5.
template <int D, int TTW, bool STD>
CID_DEVINL void xattn_core_unit(half_t* T, int TP, int trow, int col0, const half_t* kph, const half_t* vph,
                                int n_txt, int n_ip, float ip_scale, int idx, int hi) {
    constexpr int QKS = (D + 15) / 16;           // k-steps of Q K^T
    constexpr int DVT = (D + 31) / 32;           // 32-row slices of the head dim
    const int n_all = n_txt + n_ip;
    // Q_h^T fragments (B operand); columns beyond D hit K's zero padding
    half8 qf[TTW][QKS];
#pragma unroll
    for (int t = 0; t < TTW; ++t)
#pragma unroll
        for (int kk = 0; kk < QKS; ++kk)
            qf[t][kk] = *reinterpret_cast<const half8*>(T + (trow + t * 32 + idx) * TP + col0 + kk * 16 + hi * 8);
    // all K fragments of the head are requested up front (L2-resident, 1 KiB each)
    half8 kf[XC_KTILES][QKS];
#pragma unroll
    for (int kt = 0; kt < XC_KTILES; ++kt)
#pragma unroll
        for (int kk = 0; kk < QKS; ++kk)
            kf[kt][kk] = ld_global_h8(kph + (long)(kt * QKS + kk) * 512);
    f32x16 s[XC_KTILES][TTW];
#pragma unroll
    for (int kt = 0; kt < XC_KTILES; ++kt)
#pragma unroll
        for (int t = 0; t < TTW; ++t) s[kt][t] = zero_f16v();
#pragma unroll
    for (int kt = 0; kt < XC_KTILES; ++kt)
#pragma unroll
        for (int kk = 0; kk < QKS; ++kk)
#pragma unroll
            for (int t = 0; t < TTW; ++t) s[kt][t] = mfma32(kf[kt][kk], qf[t][kk], s[kt][t]);
    // V fragments of the first head-dim slice travel while the softmax runs
    half8 vf[XC_PV_KSTEPS];
#pragma unroll
    for (int ks = 0; ks < XC_PV_KSTEPS; ++ks)
        vf[ks] = ld_global_h8(vph + (long)ks * 512);
    // two independent softmaxes over [0, n_txt) and [n_txt, n_all)
    half8 pf[TTW][XC_PV_KSTEPS];
#pragma unroll
    for (int t = 0; t < TTW; ++t) {
        float mt = -INFINITY, mi = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < XC_KTILES; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + crow(r, hi);
                const float v = s[kt][t][r];
                if (STD && kt * 32 + crow(r, 1) < 77) { mt = fmaxf(mt, v); continue; }      // text for both halves
                if (STD && kt * 32 + crow(r, 0) >= 81) continue;                             // padding for both halves
                if (key < n_txt) mt = fmaxf(mt, v);
                else if (key < n_all) mi = fmaxf(mi, v);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        mi = fmaxf(mi, __shfl_xor(mi, 32, 64));
        float lt = 0.f, li = 0.f;
#pragma unroll
        for (int kt = 0; kt < XC_KTILES; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + crow(r, hi);
                float p;
                if (STD && kt * 32 + crow(r, 1) < 77) { p = __builtin_amdgcn_exp2f(s[kt][t][r] - mt); lt += p; }
                else if (STD && kt * 32 + crow(r, 0) >= 81) { p = 0.f; }
                else if (key < n_txt) { p = __builtin_amdgcn_exp2f(s[kt][t][r] - mt); lt += p; }
                else if (key < n_all) { p = __builtin_amdgcn_exp2f(s[kt][t][r] - mi); li += p; }
                else p = 0.f;
                s[kt][t][r] = p;
            }
        lt += __shfl_xor(lt, 32, 64);
        li += __shfl_xor(li, 32, 64);
        const float it = 1.f / lt;
        const float ii = (n_ip > 0) ? ip_scale / li : 0.f;
#pragma unroll
        for (int kt = 0; kt < XC_KTILES; ++kt)
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                half8 pv;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = gq * 8 + i;
                    const int key = kt * 32 + crow(r, hi);
                    float f;
                    if (STD && kt * 32 + crow(r, 1) < 77) f = it;
                    else if (STD && kt * 32 + crow(r, 0) >= 81) f = 0.f;
                    else f = (key < n_txt) ? it : ii;
                    pv[i] = (half_t)(s[kt][t][r] * f);
                }
                pf[t][kt * 2 + gq] = pv;
            }
    }
    // O_h^T = V_h^T P^T, one 32-row slice of the head dim at a time; O_h -> T over Q_h
#pragma unroll
    for (int d = 0; d < DVT; ++d) {
        f32x16 o[TTW];
#pragma unroll
        for (int t = 0; t < TTW; ++t) o[t] = zero_f16v();
        half8 vn[XC_PV_KSTEPS];
        if (d + 1 < DVT) {   // next slice's fragments are requested before this slice's MFMAs
#pragma unroll
            for (int ks = 0; ks < XC_PV_KSTEPS; ++ks)
                vn[ks] = ld_global_h8(vph + (long)((d + 1) * XC_PV_KSTEPS + ks) * 512);
        }
#pragma unroll
        for (int ks = 0; ks < XC_PV_KSTEPS; ++ks)
#pragma unroll
            for (int t = 0; t < TTW; ++t) o[t] = mfma32(vf[ks], pf[t][ks], o[t]);
        if (d + 1 < DVT) {
#pragma unroll
            for (int ks = 0; ks < XC_PV_KSTEPS; ++ks) vf[ks] = vn[ks];
        }
#pragma unroll
        for (int t = 0; t < TTW; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int dd = d * 32 + 8 * j + 4 * hi;
                if (dd < D) {
                    half4 ov;
#pragma unroll
                    for (int i = 0; i < 4; ++i) ov[i] = (half_t)o[t][j * 4 + i];
                    *reinterpret_cast<half4*>(T + (trow + t * 32 + idx) * TP + col0 + dd) = ov;
                }
            }
    }
}
