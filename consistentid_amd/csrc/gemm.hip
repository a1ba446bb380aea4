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
//   * activations / weights go L2 -> LDS by DMA (buffer_load ... lds, 1-KiB pieces, no VGPR round trip) in BK = 64
//     slabs (128-B rows), two or three LDS stages + two fragment register sets, ONE barrier per slab: DMA of slab
//     t + stages is issued behind the barrier that publishes slab t+1 (counted vmcnt: the younger issues keep flying),
//     every MFMA batch runs while the next batch's ds_reads are in flight.  Three stages where they cost no co-resident
//     workgroup (the 256-token tile; launches of <= 256 workgroups of the smaller tiles): inside a denoise step a layer's
//     weights are first-touch HBM reads, and with few token tiles per weight slab every workgroup pays that latency;
//   * GEGLU launches walk several n-tiles of one token tile per workgroup (N-loop): one flattened slab sequence, the erf
//     epilogue drains the accumulators between two tiles while the ring runs on -- at K = 320 a 128 x 128 tile is five
//     slabs, and one workgroup per tile meant 5 120 workgroups of launch / prologue / epilogue structure;
//   * 16-B chunk c of LDS row r is stored at chunk c ^ swz_key(r) (the XOR sits on the DMA's source address): a
//     ds_read_b128 of 16 consecutive rows at one k-chunk touches 16 distinct bank slots (conflict free);
//   * operands are fed "swapped" (MFMA A = weight rows, B = token rows) so a lane owns 4 consecutive CHANNELS of one
//     token; the plain epilogue transposes the finished fp16 tile through the idle LDS and stores whole rows, 16 B per
//     lane; the 3x3 taps are shifted token rows of the same token-major image (zero outside), nearest-2x upsampling and
//     stride 2 are folded into the row gather, a skip concat is two source pointers;
//   * stride-1 3x3 convolutions keep an LDS HALO tile per 64-channel slab (igemm_halo_kernel): nine taps = nine shifted
//     views of one DMA; its eight waves run the half-slab OFFSET pipeline (second four waves half a slab out of phase,
//     static priority) -- DESIGN.md 4.2;
//   * epilogue extras: GEGLU (erf), LayerNorm fold (row statistics from the A fragments on their way to the MFMAs),
//     GroupNorm statistics of the tensor being written (for the GroupNorm that consumes it), residual quads requested
//     at kernel start on the 256-row tiles;
//   * the V third of the fused QKV projection flips the operand roles so a lane owns 4
//     consecutive TOKENS of one channel and writes V transposed for the attention kernel;
//   * small-M / deep-K problems (the 8x8 and 16x16 levels) are split along K over
//     gridDim.z with fp32 partial tiles + a fused reduce/epilogue kernel, so that every
//     launch puts >= ~2 waves on every SIMD of the 256 CUs.
#include "xattn_core.h"
#include "gemm_args.h"
#include "../../include/cid.h"
#include <stdlib.h>

// Profiling knobs (CID_GEMM_ABLATE bits: 1 no DMA in the loop, 2 no MFMA, 4 no halo DMA, 8 no halo fragment
// reads) exist only in -DCID_GEMM_ABLATION builds: a runtime branch around the fragment reads splits the
// basic block and makes the compiler drain lgkmcnt to 0 before the MFMA batch the reads should overlap.
// Pipeline form of the halo kernel (one comparator knob, default = what ships):
//   CID_HALO_STAGGER  1 = the second four waves run half a slab out of phase with static priority (see igemm_halo_kernel),
//                     0 = all eight waves in lock step (the comparator build of DESIGN.md 4.2).
// Closed experiments are not in the tree any more (per-batch priority flips, the same offset pipeline for the plain
// eight-wave GEMMs, a three-stage weight ring in the halo kernel -- measured twice, hot and inside the step --, fragment
// reads ahead of the DMA issue): DESIGN.md 5.3 has their numbers.
#ifndef CID_HALO_STAGGER
#define CID_HALO_STAGGER 1
#endif
// Experiment builds only (--variant ctr CID_CONV_TRACE): phase stamps of waves 0 and NW/2 of one workgroup (tools/conv_trace.py)
#ifdef CID_CONV_TRACE
__device__ unsigned long long g_conv_trace[2 * 4096];
#define CONV_STAMP(k) do { if (tr_on) tr_ts[k] = __builtin_readcyclecounter(); } while (0)
#define CONV_FLUSH() do { if (tr_on && lane == 0 && tr_n + 8 <= 4096) { for (int q_ = 0; q_ < 8; ++q_) g_conv_trace[tr_slot * 4096 + tr_n + q_] = tr_ts[q_]; } tr_n += 8; } while (0)
#else
#define CONV_STAMP(k) do { } while (0)
#define CONV_FLUSH() do { } while (0)
#endif
#if defined(CID_GEMM_ABLATION)
#define CID_ABL(bit) ((a.ablate & (bit)) != 0)
#else
#define CID_ABL(bit) false
#endif

namespace {
using cidg::GemmArgs;
using cidg::wait_vmcnt;

// Pins fragment registers to "loaded": the compiler has to place its lgkmcnt wait for the ds_reads that
// produced them HERE (after the MFMA batch issued just before), not in front of the next MFMA batch where it
// would also drain the reads that batch is supposed to overlap (loop-carried fragments otherwise get a
// conservative s_waitcnt lgkmcnt(0) at the loop header).
template <int N>
__device__ __forceinline__ void frags_landed(half8 (&f)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i]));
}


constexpr int BK = 64;

typedef float f32x4v __attribute__((ext_vector_type(4)));

CID_DEVINL f32x4v mfma16(half8 a, half8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// byte offset of 16-B chunk c (0..7) of row r in a [rows][64] fp16 LDS tile
// Bank swizzle key of an LDS row (XORed onto the 16-B chunk index).  ds_read_b128 is served in groups of 16 lanes
// that mix two lq values -- lanes {0-3, 12-15} of one quarter with lanes {4-11} of the next (MI355X_MICROARCH
// LDS table) -- i.e. rows {0-3, 12-15} read chunk k and rows {4-11} chunk k ^ 1 of a 16-row fragment.  The
// 256-B bank line holds two rows, so the eight same-parity rows of a group must land on eight different chunks
// for EVERY start row (the halo taps shift the start row by arbitrary amounts): key = 2 * ((row / 2) mod 4) does
// that -- rows {0-3, 12-15} take the four even keys, rows {4-11} the same four keys, made odd by their k ^ 1.
// (The earlier key (row / 2) mod 8 was conflict free only for fragments starting at a multiple of 16 rows:
// 24 % of all LDS cycles of the halo kernel were bank conflicts, rocprofv3 SQ_LDS_BANK_CONFLICT.)
CID_DEVINL int swz_key(int r) { return ((r >> 1) & 3) << 1; }
CID_DEVINL int lds_off(int r, int c) { return r * 128 + ((c ^ swz_key(r)) << 4); }

// sum over the 16 lanes of a DPP row (all lanes get the total): four row rotations, v_add_f32 with a DPP operand
CID_DEVINL float row16_sum(float v) {
#define CID_ROR(N) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false))
    CID_ROR(8); CID_ROR(4); CID_ROR(2); CID_ROR(1);
#undef CID_ROR
    return v;
}

// shared epilogue: VMODE transposed-V store, split-K partials, GEGLU, or bias / time-row / residual.
// LN: the A operand was the RAW residual stream and W carries gamma -- lmean / lrstd are the LayerNorm statistics of this
// lane's token (row l16 of 16-token tile t), out = rstd * (acc - mean * ln_s[n]) + ln_b[n]  (ln_b includes the bias).
template <int TM, int TN, bool VMODE, bool LN, bool NLOOP = false>
CID_DEVINL void igemm_epilogue(const GemmArgs& a, f32x4v (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                               int l16, int lq, char* smem, int wave, const float (&lmean)[TM], const float (&lrstd)[TM],
                               int nwaves, int wn_count, const half4 (&rpre)[TM][TN], bool rpre_valid) {
    // ---- epilogue ---------------------------------------------------------------
    if constexpr (VMODE) {
        // D rows = tokens (4 lq + i), cols = channels (l16): lane owns channel n, 4 consecutive tokens
#pragma unroll
        for (int c = 0; c < TN; ++c) {
            const int n = n0 + (wn * TN + c) * 16 + l16;
            if (n >= a.n_end) continue;
            const int cg = n - a.n_vt0;
            const int head = cg / a.dhead, dd = cg - head * a.dhead;
            const float bv = LN ? a.ln_b[n] : (a.bias ? (float)a.bias[n] : 0.f);
            const float sv = LN ? a.ln_s[n] : 0.f;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int mt = m0 + (wm * TM + t) * 16;   // 16-token tile base
                if (mt >= a.M) continue;
                const int b = mt / a.ntok, tok0 = mt - b * a.ntok;
                half_t* dst = a.vt + ((long)(b * a.heads + head) * a.dvp + dd) * a.ntok + tok0;
                half4 o;
                if constexpr (LN) {
                    // this lane's accumulator rows are tokens 4 lq + i; their statistics live in the lanes l16 = 4 lq + i
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float mu = __shfl(lmean[t], 4 * lq + i, 64), rs = __shfl(lrstd[t], 4 * lq + i, 64);
                        o[i] = (half_t)(rs * (acc[t][c][i] - mu * sv) + bv);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (half_t)(acc[t][c][i] + bv);
                }
                // token 4 lq + i of the 16-group -> pos = 8 (lq & 1) + 4 (lq >> 1) + i
                *reinterpret_cast<half4*>(dst + 8 * (lq & 1) + 4 * (lq >> 1)) = o;
            }
        }
        return;
    } else {
        if (!NLOOP && a.splitk > 1) {
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
        if (NLOOP || a.mode == 1) {         // (the N-loop form exists for GEGLU launches only)
            // GEGLU: even 16-row tile = value, odd = gate (weights interleaved by the host)
#pragma unroll
            for (int c = 0; c + 1 < TN; c += 2) {
                const int nt = n0 + (wn * TN + c) * 16;   // interleaved column of the value tile
                if (nt >= a.n_end) continue;
                float bv[4] = {0.f, 0.f, 0.f, 0.f}, bg[4] = {0.f, 0.f, 0.f, 0.f};
                float sv[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (LN) {
                    const f32x4v fv = *reinterpret_cast<const f32x4v*>(a.ln_b + nt + 4 * lq);
                    const f32x4v fg = *reinterpret_cast<const f32x4v*>(a.ln_b + nt + 16 + 4 * lq);
                    const f32x4v qv = *reinterpret_cast<const f32x4v*>(a.ln_s + nt + 4 * lq);
                    const f32x4v qg = *reinterpret_cast<const f32x4v*>(a.ln_s + nt + 16 + 4 * lq);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { bv[i] = fv[i]; bg[i] = fg[i]; sv[i] = qv[i]; sg[i] = qg[i]; }
                } else if (a.bias && !CID_ABL(512)) {
                    const half4 hv = *reinterpret_cast<const half4*>(a.bias + nt + 4 * lq);
                    const half4 hg = *reinterpret_cast<const half4*>(a.bias + nt + 16 + 4 * lq);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { bv[i] = (float)hv[i]; bg[i] = (float)hg[i]; }
                }
                half_t* op = a.out + (long)(m0 + wm * TM * 16 + l16) * a.ldo + (nt >> 1) + 4 * lq;
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    if (m0 + (wm * TM + t) * 16 + l16 >= a.M) continue;
                    half4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float val = acc[t][c][i], gate = acc[t][c + 1][i];
                        if constexpr (LN) {
                            val = lrstd[t] * (val - lmean[t] * sv[i]);
                            gate = lrstd[t] * (gate - lmean[t] * sg[i]);
                        }
                        o[i] = CID_ABL(32) ? (half_t)((val + bv[i]) * (gate + bg[i])) : (half_t)((val + bv[i]) * gelu_erf_f(gate + bg[i]));
                    }
                    if (CID_ABL(64)) { if (o[0] == (half_t)123.25f && o[1] == (half_t)77.5f) a.out[0] = o[2]; }   // profiling knob: no GEGLU stores
                    else *reinterpret_cast<half4*>(op + (long)t * 16 * a.ldo) = o;
                }
            }
            return;
        }
        if constexpr (NLOOP) return;
        if (CID_ABL(16)) {   // profiling knob: no epilogue traffic (keeps the accumulators alive)
            float sacc = 0.f;
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int c = 0; c < TN; ++c) sacc += acc[t][c][0] + acc[t][c][1] + acc[t][c][2] + acc[t][c][3];
            if (sacc == 123.456f) a.out[0] = (half_t)sacc;
            return;
        }
        // plain: the MFMA layout gives a lane ONE token and 4 consecutive channels per 16x16 tile -- stored
        // directly that is 8 B per lane in 32-B row segments (store-issue bound, partial lines).  Instead the
        // finished fp16 values (bias / time row / residual added in fp32, rounded once) are transposed through
        // the now idle LDS: each wave parks its [16 TM tokens][16 TN channels] tile and writes it back out as
        // whole rows, 16 B per lane.
        constexpr int P = TN * 16 + 8;                     // staging row pitch (halfs), 16-B aligned
        half_t* stg = reinterpret_cast<half_t*>(smem) + wave * (TM * 16 * P);
        __builtin_amdgcn_s_barrier();                      // every wave has finished reading the pipeline stages
        // GroupNorm statistics of the tensor being written (its consumer is a GroupNorm): per-lane partial sums of the
        // fp16-ROUNDED outputs -- what the consumer will read -- over this lane's TM tokens, 4 channels per tile
        const bool gstat = a.gn_stats != nullptr;
        float gs[TN][4], gq[TN][4];
#pragma unroll
        for (int c = 0; c < TN; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) { gs[c][i] = 0.f; gq[c][i] = 0.f; }
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int m = m0 + (wm * TM + t) * 16 + l16;
            const bool mok = m < a.M;
            const half_t* rb = a.rowbias ? a.rowbias + (long)((mok ? m : 0) / a.rows_per_sample) * a.ld_rowbias : nullptr;
            const half_t* rs = (a.res && mok) ? a.res + (long)m * a.ldr : nullptr;
#pragma unroll
            for (int c = 0; c < TN; ++c) {
                const int n = n0 + (wn * TN + c) * 16 + 4 * lq;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[t][c][i];
                if constexpr (LN) {
                    if (n < a.n_end) {
                        const f32x4v qs = *reinterpret_cast<const f32x4v*>(a.ln_s + n);
                        const f32x4v qb = *reinterpret_cast<const f32x4v*>(a.ln_b + n);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = lrstd[t] * (v[i] - lmean[t] * qs[i]) + qb[i];
                    }
                }
                if (n < a.n_end && !CID_ABL(512)) {      // (profiling knob 512: no bias / time-row / residual loads)
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
                        const half4 bb = rpre_valid ? rpre[t][c] : *reinterpret_cast<const half4*>(rs + n);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += (float)bb[i];
                    }
                }
                half4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (half_t)v[i];
                *reinterpret_cast<half4*>(stg + (t * 16 + l16) * P + c * 16 + 4 * lq) = o;
                if (gstat && mok) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float f = (float)o[i]; gs[c][i] += f; gq[c][i] = __builtin_fmaf(f, f, gq[c][i]); }
                }
            }
        }
        float* gst = reinterpret_cast<float*>(smem + nwaves * (TM * 16 * P) * 2);    // [wave][TN * 16 channels][2], behind the staging tiles
        if (gstat) {
            // fold the 16 tokens of a DPP row (lanes of one lq hold the same channels), park the wave's per-channel sums
#pragma unroll
            for (int c = 0; c < TN; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) { gs[c][i] = row16_sum(gs[c][i]); gq[c][i] = row16_sum(gq[c][i]); }
            if (l16 == 0) {
#pragma unroll
                for (int c = 0; c < TN; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float* p = gst + (wave * (TN * 16) + c * 16 + 4 * lq + i) * 2;
                        p[0] = gs[c][i]; p[1] = gq[c][i];
                    }
            }
        }
        // the tile is private to the wave: its own LDS writes only need lgkmcnt(0), no barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int CPR = TN * 2;                        // 16-B chunks per tile row
        constexpr int PER = TM * 16 * CPR / 64;
        static_assert((TM * 16 * CPR) % 64 == 0, "tile chunks must divide over the wave");
        const int lane = lq * 16 + l16;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = i * 64 + lane;
            const int r = e / CPR, ch = e - r * CPR;
            const int m = m0 + wm * TM * 16 + r;
            const int n = n0 + wn * TN * 16 + ch * 8;
            const half8 val = *reinterpret_cast<const half8*>(stg + r * P + ch * 8);
            if (m < a.M && n < a.n_end) {
                *reinterpret_cast<half8*>(a.out + (long)m * a.ldo + n) = val;
                if (a.out2) *reinterpret_cast<half8*>(a.out2 + (long)m * a.ldo + n) = val;
            }
        }
        if (gstat) {
            // one (sum, sumsq) pair per statistics unit of this workgroup's [BM tokens] x [BN channels] tile: thread u adds the
            // WM waves that share its channels and the gn_unit channels of unit u, in a fixed order (bit-reproducible)
            __builtin_amdgcn_s_barrier();
            const int bn = wn_count * TN * 16, wm_count = nwaves / wn_count;
            const int nu = bn / a.gn_unit;
            const int u = wave * 64 + lq * 16 + l16;
            if (u < nu && n0 + u * a.gn_unit < a.n_end) {
                float S = 0.f, Q = 0.f;
                for (int w_ = 0; w_ < wm_count; ++w_)
                    for (int ch = u * a.gn_unit; ch < (u + 1) * a.gn_unit; ++ch) {
                        const int wv = w_ * wn_count + ch / (TN * 16);
                        const float* p = gst + (wv * (TN * 16) + ch % (TN * 16)) * 2;
                        S += p[0]; Q += p[1];
                    }
                float* dst = a.gn_stats + ((long)(m0 / (wm_count * TM * 16)) * (a.N / a.gn_unit) + (n0 - a.n_begin) / a.gn_unit + u) * 2;
                dst[0] = S; dst[1] = Q;
            }
        }
    }
}

// The residual operand of the plain epilogue (x + f(x) of every attention / feed-forward / resnet block), fetched at
// kernel START into registers: issued in front of the first slab's DMA, the loads travel under the pipeline prologue's
// latency instead of standing exposed in the epilogue (lin 320 -> 320 at M = 32768: 27.2 us with the residual read in the
// epilogue, 19.9 us without a residual -- tools/abl.py).  8-byte quads in the accumulator layout: token l16, 4 channels.
template <int TM, int TN>
CID_DEVINL bool prefetch_residual(const GemmArgs& a, half4 (&rpre)[TM][TN], int m0, int n0, int wm, int wn, int l16, int lq) {
    // only the 256-token tiles prefetch: they run two waves per SIMD whatever they do (256-register budget, 40 of them for
    // the quads); the smaller tiles would drop a wave per SIMD -- and with it a whole workgroup per CU
    const bool on = TM >= 4 && a.res != nullptr && a.mode == 0 && a.splitk == 1;
    if constexpr (TM < 4) return false;
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int c = 0; c < TN; ++c) {
            const int m = m0 + (wm * TM + t) * 16 + l16, n = n0 + (wn * TN + c) * 16 + 4 * lq;
            half4 z = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
            rpre[t][c] = (on && m < a.M && n < a.n_end) ? *reinterpret_cast<const half4*>(a.res + (long)m * a.ldr + n) : z;
        }
    return on;
}

// (N-loop instances keep the 128-register budget of the one-tile form: two eight-wave workgroups per CU, so that one's erf
//  epilogue runs beside the other's MFMAs)
// ATT_D > 0 (mode 3): the launch is the QUERY PROJECTION of an identity cross-attention whose tile spans whole heads of
// ATT_D channels; its epilogue keeps the fp16 Q tile in LDS, runs the two-stream attention of those heads on it
// (xattn_core_unit, one (head, 32-token) unit per wave) and writes O -- q never goes to HBM, one launch less per layer.
template <int TM, int TN, int WM, int WN, bool VMODE, int NBUF, bool LN, bool NLOOP = false, int ATT_D = 0>
// (two workgroups per CU asked for even of the four-wave tiles: with a 512-register budget the compiler parks the accumulators
//  in AGPRs and rotates them through VGPRs at the head of every slab -- 60 v_accvgpr moves beside 20 MFMAs, tools/isa_mix.py)
__global__ void __launch_bounds__(64 * WM * WN, (NLOOP && TM == 2) ? 4 : 2)
igemm_kernel(GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // device-only builtins below; the host pass only needs the stub
    constexpr int NW = WM * WN;
    constexpr int BM = 16 * TM * WM;
    constexpr int BN = 16 * TN * WN;
    constexpr int XPW = (BM / 8 + NW - 1) / NW, WPW = (BN / 8 + NW - 1) / NW;   // 1-KiB DMA pieces per wave
    constexpr int XI = XPW * NW, WI = WPW * NW;           // every wave issues the same number of pieces;
    // (surplus x pieces are zero-filled scratch rows; surplus weight pieces lie wholly past the tile and are never issued,
    //  so the weight part of a stage is exactly BN rows)
    constexpr int XBYTES = XI * 1024, WBYTES = BN * 128;
    constexpr int SBYTES = XBYTES + WBYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lq = lane >> 4;
    const int r8 = lane >> 3, c8 = lane & 7;
    const int wm = wave / WN, wn = wave % WN;
    int nb_, mb_;
    cidg::xcd_tile(a, nb_, mb_);      // XCD-aware tile order (gemm_args.h)
    // N-loop (GEGLU launches, a.nloop > 1): the workgroup walks a.nloop consecutive n-tiles of ONE token tile as a single
    // flattened slab sequence -- the DMA ring runs ahead across tile boundaries, the accumulators are drained by the GEGLU
    // epilogue (registers -> HBM, no LDS) between two tiles.  At K = 320 a 128 x 128 tile is five slabs: as one workgroup
    // per tile (5 120 workgroups, 40 960 waves at SD1.5 level 0) the launch is wave-dispatch and prologue structure.
    const int n0 = a.n_begin + nb_ * BN * a.nloop;
    const int m0 = mb_ * BM;

    // ---- staging: global -> LDS by DMA (buffer_load ... lds), no VGPR round trip ----------
    // One wave instruction moves 8 tile rows x 128 B = 1 KiB: lane (r8, c8) fetches 16 B and the
    // hardware writes them lane-linearly, i.e. to LDS chunk c8 of row r8.  The bank-conflict
    // swizzle therefore sits on the SOURCE side: lane c8 fetches logical chunk c8 ^ swz_key(R)
    // (same 128-B line, coalescing unchanged) and readers use the same XOR (lds_off).
    // Out-of-range offsets (padding taps, ragged rows) return zeros through the descriptor.
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x1, 0, a.bytes_x1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 ? a.x2 : a.x1), 0,
                                                                           a.x2 ? a.bytes_x2 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.bytes_w, 0x00020000);
    int xb[XPW], xy[XPW], xx[XPW], xsw[XPW];
    bool xok[XPW];
#pragma unroll
    for (int j = 0; j < XPW; ++j) {
        const int R = (j * NW + wave) * 8 + r8;
        const int m = m0 + R;
        xok[j] = (R < BM) && (m < a.M);
        xsw[j] = (c8 ^ swz_key(R)) * 8;
        if (a.taps == 9) {
            const int hw = a.Ho * a.Wo;
            const int b = m / hw, rem = m - b * hw;
            xb[j] = b; xy[j] = rem / a.Wo; xx[j] = rem - xy[j] * a.Wo;
        } else {
            xb[j] = 0; xy[j] = 0; xx[j] = m;
        }
    }
    unsigned woff[WPW];   // byte offset of this lane's weight chunk at k = 0 (or OOB)
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int R = (j * NW + wave) * 8 + r8;
        const bool ok = (R < BN) && (n0 + R < a.n_end);
        woff[j] = ok ? (unsigned)(((long)(n0 + R) * a.ktot + (c8 ^ swz_key(R)) * 8) * 2) : OOB;
    }
    // row byte offsets (per source pitch) of the current tap, recomputed only when the tap changes
    unsigned xoff1[XPW], xoff2[XPW];
    auto set_tap = [&](int tap) {
        const int dy = (a.taps == 9) ? tap / 3 - 1 : 0;
        const int dx = (a.taps == 9) ? tap - (tap / 3) * 3 - 1 : 0;
#pragma unroll
        for (int j = 0; j < XPW; ++j) {
            bool ok = xok[j];
            long row = xx[j];
            if (a.taps == 9) {
                int yy = xy[j] * a.stride + dy, xs = xx[j] * a.stride + dx;
                const int Hv = a.Hi << a.up, Wv = a.Wi << a.up;  // virtual (upsampled) input
                ok = ok && (yy >= 0) && (yy < Hv) && (xs >= 0) && (xs < Wv);
                yy >>= a.up; xs >>= a.up;
                row = ((long)xb[j] * a.Hi + yy) * a.Wi + xs;
            }
            xoff1[j] = ok ? (unsigned)((row * a.ld1 + xsw[j]) * 2) : OOB;
            xoff2[j] = ok ? (unsigned)((row * a.ld2 + xsw[j]) * 2) : OOB;
        }
    };
    int ld_tap = -1;

    auto issue = [&](int slab, int buf, unsigned wtile = 0u) {      // wtile: byte offset of the n-tile's weight rows (N-loop)
        // K order: channel slab major, tap minor -- the 9 shifted views of one 64-channel slab of
        // the activation tile are fetched back to back, so 8 of the 9 hit in L1/L2
        const int cs = slab / a.taps;
        const int tap = slab - cs * a.taps;
        const int cbase = cs * BK;
        if (tap != ld_tap) { set_tap(tap); ld_tap = tap; }
        char* xs = smem + buf * SBYTES;
        char* ws = xs + XBYTES;
        const bool first = cbase < a.c1;
        const unsigned coff = (unsigned)((first ? cbase : cbase - a.c1) * 2);
        if (first) {
#pragma unroll
            for (int j = 0; j < XPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x1, (lds_void*)(xs + (j * NW + wave) * 1024), 16, xoff1[j] + coff, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < XPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x2, (lds_void*)(xs + (j * NW + wave) * 1024), 16, xoff2[j] + coff, 0, 0, 0);
        }
        const unsigned koff = (unsigned)((tap * (a.c1 + a.c2) + cbase) * 2) + wtile;
#pragma unroll
        for (int j = 0; j < WPW; ++j)
            if ((j + 1) * NW * 8 <= BN || (j * NW + wave) * 8 < BN)      // (pieces wholly past the tile's rows: nobody reads them)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(ws + (j * NW + wave) * 1024), 16, woff[j] + koff, 0, 0, 0);
    };

    f32x4v acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int c = 0; c < TN; ++c) acc[t][c] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // K range of this split
    const int s_begin = (int)((long)a.nslab * blockIdx.z / a.splitk);
    const int s_end = (int)((long)a.nslab * (blockIdx.z + 1) / a.splitk);

    // ---- pipeline ----------------------------------------------------------------------
    // Two LDS stages + two fragment register sets.  Per 64-deep slab t:
    //   read F1(t) | MFMA F0(t) | wait DMA(t+1) + barrier | issue DMA(t+2) | read F0(t+1) | MFMA F1(t)
    // so every MFMA batch runs while the next batch's ds_reads are in flight, the barrier sits
    // between two MFMA batches (the SIMD's partner wave keeps the matrix pipe busy), and the
    // stage being overwritten by DMA(t+2) has already been pulled into registers by every wave.
    static_assert(NBUF >= 2 && NBUF <= 4, "two to four LDS stages");
    auto read_frags = [&](const char* xs, const char* ws, int ks, half8 (&xf)[TM], half8 (&wf)[TN]) {
        if (CID_ABL(256)) return;     // profiling knob: no LDS fragment reads
#pragma unroll
        for (int t = 0; t < TM; ++t)
            xf[t] = *reinterpret_cast<const half8*>(xs + lds_off((wm * TM + t) * 16 + l16, ks * 4 + lq));
#pragma unroll
        for (int c = 0; c < TN; ++c)
            wf[c] = *reinterpret_cast<const half8*>(ws + lds_off((wn * TN + c) * 16 + l16, ks * 4 + lq));
    };
    auto mma = [&](const half8 (&xf)[TM], const half8 (&wf)[TN]) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int c = 0; c < TN; ++c)
                acc[t][c] = VMODE ? mfma16(xf[t], wf[c], acc[t][c]) : mfma16(wf[c], xf[t], acc[t][c]);
    };

    // LN: LayerNorm statistics of this lane's token of every 16-token tile, from the activation fragments on their way to
    // the MFMAs (v_dot2_f32_f16 against ones / against itself); a lane sees a quarter of a row's k-chunks, the four lane
    // rows are added up after the loop.  Every wave column (wn) computes them for itself: no exchange, no barrier.
    float lsum[TM], lsq[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) { lsum[t] = 0.f; lsq[t] = 0.f; }
    bool ln_on = true;                 // (N-loop: the row statistics are taken during the first n-tile only)
    auto ln_acc = [&](const half8 (&xf)[TM]) {
        if constexpr (LN) {
            if (!ln_on) return;
            const half2v one = {(half_t)1.f, (half_t)1.f};
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const half2v h = {xf[t][2 * j], xf[t][2 * j + 1]};
                    lsum[t] = __builtin_amdgcn_fdot2(h, one, lsum[t], false);
                    lsq[t] = __builtin_amdgcn_fdot2(h, h, lsq[t], false);
                }
        }
    };

    float lmean[TM], lrstd[TM];
    auto ln_finish = [&]() {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            if constexpr (LN) {
                float sm = lsum[t], sq = lsq[t];
                sm += __shfl_xor(sm, 16, 64); sq += __shfl_xor(sq, 16, 64);
                sm += __shfl_xor(sm, 32, 64); sq += __shfl_xor(sq, 32, 64);
                const float inv = 1.f / (float)a.ktot;
                const float mu = sm * inv;
                lmean[t] = mu;
                lrstd[t] = rsqrtf(fmaxf(sq * inv - mu * mu, 0.f) + a.ln_eps);
            } else { lmean[t] = 0.f; lrstd[t] = 1.f; }
        }
    };

    half4 rpre[TM][TN];
    const bool rpre_valid = VMODE ? false : prefetch_residual<TM, TN>(a, rpre, m0, n0, wm, wn, l16, lq);
    if constexpr (NLOOP) {
        // ---- N-loop: a.nloop n-tiles x a.nslab slabs as ONE slab sequence (taps == 1, no split-K, GEGLU epilogue) -----------
        // same two-stage ring and barrier algebra as below; slab f of the sequence = slab f % nslab of n-tile f / nslab, its
        // weights a.ktot * BN halfs further on per tile.  Between two tiles the accumulators go through the GEGLU epilogue
        // (registers -> HBM; it touches no LDS, so the ring keeps running: the first two slabs of the next tile are in flight).
        static_assert(NBUF == 2, "the N-loop form runs on two stages");
        const int ns = a.nslab, T = ns * a.nloop;
        const unsigned wstep = (unsigned)((long)BN * a.ktot * 2);
        int is_ks = 0; unsigned is_w = 0u;            // issue cursor (advances one slab per call)
        set_tap(0);                                   // one source, one tap: the row offsets never change
        auto issue_next = [&](int buf) {
            char* xs = smem + buf * SBYTES;
            char* ws = xs + XBYTES;
            const unsigned coff = (unsigned)(is_ks * BK * 2);
#pragma unroll
            for (int j = 0; j < XPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x1, (lds_void*)(xs + (j * NW + wave) * 1024), 16, xoff1[j] + coff, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < WPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(ws + (j * NW + wave) * 1024), 16, woff[j] + coff + is_w, 0, 0, 0);
            if (++is_ks == ns) { is_ks = 0; is_w += wstep; }
        };
        half8 xf0[TM], wf0[TN], xf1[TM], wf1[TN];
        issue_next(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        read_frags(smem, smem + XBYTES, 0, xf0, wf0);
        if (1 < T) issue_next(1);
        frags_landed(xf0); frags_landed(wf0);
        int cur = 0, ks = 0, nt = 0;
        for (int f = 0; f < T; ++f) {
            const char* xs = smem + cur * SBYTES;
            read_frags(xs, xs + XBYTES, 1, xf1, wf1);
            mma(xf0, wf0);
            ln_acc(xf0);
            __builtin_amdgcn_sched_barrier(0);
            frags_landed(xf1); frags_landed(wf1);
            if (f + 1 < T) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (f + 2 < T) issue_next(cur);
                const char* xn = smem + (cur ^ 1) * SBYTES;
                read_frags(xn, xn + XBYTES, 0, xf0, wf0);
            }
            mma(xf1, wf1);
            ln_acc(xf1);
            __builtin_amdgcn_sched_barrier(0);
            frags_landed(xf0); frags_landed(wf0);
            cur ^= 1;
            if (++ks == ns) {           // n-tile complete
                if (nt == 0) { ln_finish(); ln_on = false; }
                igemm_epilogue<TM, TN, false, LN, true>(a, acc, m0, n0 + nt * BN, wm, wn, l16, lq, smem, wave, lmean, lrstd, NW, WN, rpre, false);
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int c = 0; c < TN; ++c) acc[t][c] = f32x4v{0.f, 0.f, 0.f, 0.f};
                ks = 0; ++nt;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    {
    // NBUF LDS stages: NBUF - 1 slabs of DMA in flight while one is multiplied.  Two stages hide one slab time of DMA
    // latency; launches with one or two waves per SIMD and a first-touch weight stream (the small-M levels) spend a slab
    // time of ~330 MFMA cycles waiting ~2 000 cycles for the next slab -- a deeper ring hides that (NBUF = 3, 4 where the
    // LDS holds it).  The wait in front of a barrier leaves the DMA issues YOUNGER than the slab it publishes in flight:
    // counted vmcnt, pw = pieces this wave issues per slab.
    int pw = XPW;
#pragma unroll
    for (int j = 0; j < WPW; ++j)
        if ((j + 1) * NW * 8 <= BN || (j * NW + wave) * 8 < BN) ++pw;
    half8 xf0[TM], wf0[TN], xf1[TM], wf1[TN];
    const int nsl = s_end - s_begin;
    int pre = 0;
    if (!CID_ABL(128))      // (profiling knob 128: no prologue DMA)
        for (; pre < NBUF - 1 && pre < nsl; ++pre) issue(s_begin + pre, pre);
    if (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else wait_vmcnt((pre > 0 ? pre - 1 : 0) * pw);
    __builtin_amdgcn_s_barrier();
    read_frags(smem, smem + XBYTES, 0, xf0, wf0);
    if (NBUF - 1 < nsl && !CID_ABL(1)) issue(s_begin + NBUF - 1, NBUF - 1);      // (the last stage is free from the start)
    frags_landed(xf0); frags_landed(wf0);

    int cur = 0;
    for (int slab = s_begin; slab < s_end; ++slab) {
        const char* xs = smem + cur * SBYTES;
        const int nxt = (cur + 1 == NBUF) ? 0 : cur + 1;
        read_frags(xs, xs + XBYTES, 1, xf1, wf1);
        if (!CID_ABL(2)) mma(xf0, wf0);
        ln_acc(xf0);
        __builtin_amdgcn_sched_barrier(0);
        frags_landed(xf1); frags_landed(wf1);
        if (slab + 1 < s_end) {
            // DMA(slab+1) landed (the issues behind it may still fly) and our reads of stage `cur` are done
            if (NBUF == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else {
                int young = s_end - 2 - slab;            // slabs issued behind slab + 1
                if (young > NBUF - 2) young = NBUF - 2;
                if (young < 0 || CID_ABL(1)) young = 0;
                wait_vmcnt(young * pw);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (slab + NBUF < s_end && !CID_ABL(1)) issue(slab + NBUF, cur);
            const char* xn = smem + nxt * SBYTES;
            read_frags(xn, xn + XBYTES, 0, xf0, wf0);
        }
        if (!CID_ABL(2)) mma(xf1, wf1);
        ln_acc(xf1);
        __builtin_amdgcn_sched_barrier(0);
        frags_landed(xf0); frags_landed(wf0);
        cur = nxt;
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ln_finish();

    static_assert(NW * TM * 16 * (TN * 16 + 8) * 2 + NW * TN * 16 * 8 <= NBUF * SBYTES, "epilogue staging fits the pipeline stages");
    if constexpr (ATT_D > 0) {
        // ---- mode 3: Q tile -> LDS, attention of the tile's heads, O -> HBM -------------------------------------------------
        static_assert(!VMODE && !NLOOP && BN % ATT_D == 0 && BM % 32 == 0, "the tile spans whole heads and 32-token groups");
        constexpr int TP = BN + 8;                         // tile row pitch (halfs): rows 16-byte aligned, odd number of 16-B slots
        static_assert(BM * TP * 2 <= NBUF * SBYTES, "the Q / O tile fits the pipeline stages");
        half_t* T = reinterpret_cast<half_t*>(smem);
        __builtin_amdgcn_s_barrier();                      // every wave has finished reading the pipeline stages
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int c = 0; c < TN; ++c) {
                const int nl = (wn * TN + c) * 16 + 4 * lq;          // column inside the tile
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[t][c][i];
                if constexpr (LN) {
                    const f32x4v qs = *reinterpret_cast<const f32x4v*>(a.ln_s + n0 + nl);
                    const f32x4v qb = *reinterpret_cast<const f32x4v*>(a.ln_b + n0 + nl);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = lrstd[t] * (v[i] - lmean[t] * qs[i]) + qb[i];
                }
                half4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (half_t)v[i];
                *reinterpret_cast<half4*>(T + ((wm * TM + t) * 16 + l16) * TP + nl) = o;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            const int idx = lane & 31, hi = lane >> 5;
            const long row = a.att_kvrow[m0 / a.ntok];                 // (a tile lies inside one sample: plan_gemm)
            constexpr int QKS = (ATT_D + 15) / 16, DVT = (ATT_D + 31) / 32;
            constexpr int NHL = BN / ATT_D, NTG = BM / 32;
            const int h0 = n0 / ATT_D;
            for (int u = wave; u < NHL * NTG; u += NW) {
                const int hl = u / NTG, tg = u - hl * NTG;
                const half_t* kph = a.att_kp + row * a.att_krow + (long)(h0 + hl) * XC_KTILES * QKS * 512 + lane * 8;
                const half_t* vph = a.att_vp + row * a.att_vrow + (long)(h0 + hl) * DVT * XC_PV_KSTEPS * 512 + lane * 8;
                xattn_core_unit<ATT_D, 1, true>(T, TP, tg * 32, hl * ATT_D, kph, vph, a.att_n_txt, a.att_n_ip, a.att_scale, idx, hi);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        constexpr int CPR = BN / 8;                        // 16-byte chunks per tile row
        for (int e = tid; e < BM * CPR; e += 64 * NW) {
            const int r = e / CPR, cc = e - r * CPR;
            if (m0 + r < a.M)
                *reinterpret_cast<half8*>(a.out + (long)(m0 + r) * a.ldo + n0 + cc * 8) = *reinterpret_cast<const half8*>(T + r * TP + cc * 8);
        }
        return;
    }
    igemm_epilogue<TM, TN, VMODE, LN>(a, acc, m0, n0, wm, wn, l16, lq, smem, wave, lmean, lrstd, NW, WN, rpre, rpre_valid);
#endif
}

// ---------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution with an LDS HALO tile: the activation tile (whole image rows + a one
// pixel frame, 64 channels deep) is brought into LDS ONCE per channel slab and the nine taps are
// nine shifted row views of it, instead of nine separate gathers from L2.  Per channel slab the
// DMA traffic drops from 9 x [BM x 64] to 1 x [(rows+2) x (W+2) x 64]; the weight slabs still
// stream one per tap through a 2-stage ring.  Same pipeline as igemm_kernel otherwise.
template <int TM, int TN, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN, 2)
igemm_halo_kernel(GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WM * WN;
    constexpr int BM = 16 * TM * WM;
    constexpr int BN = 16 * TN * WN;
    constexpr int HPW = 7;                                   // halo DMA pieces per wave (8 rows each)
    constexpr int HROWS = HPW * NW * 8;                      // 448 halo rows max
    constexpr int WPW = (BN / 8 + NW - 1) / NW;
    constexpr int HBYTES = HROWS * 128, WBYTES = WPW * NW * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    char* hbuf = smem;                     // [2][HBYTES]
    char* wbuf = smem + 2 * HBYTES;        // [2][WBYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lq = lane >> 4;
    const int r8 = lane >> 3, c8 = lane & 7;
    const int wm = wave / WN, wn = wave % WN;
    int nb_, mb_;
    cidg::xcd_tile(a, nb_, mb_);
    const int n0 = a.n_begin + nb_ * BN;
    const int m0 = mb_ * BM;

    // geometry: the tile is `nseg` segments of `rs` whole image rows (one segment = part of one image)
    const int W = a.Wo, H = a.Ho, HW = H * W;
    const int seg_tok = BM < HW ? BM : HW;                  // tokens per segment
    const int rs = seg_tok / W;                             // image rows per segment
    const int hs = (rs + 2) * (W + 2);                      // halo rows per segment
    const int nh = (BM / seg_tok) * hs;                     // halo rows of the tile
    const int img0 = m0 / HW;
    const int y0 = (m0 - img0 * HW) / W;                    // first image row of the tile (0 when BM >= HW)

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x1, 0, a.bytes_x1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 ? a.x2 : a.x1), 0,
                                                                           a.x2 ? a.bytes_x2 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.bytes_w, 0x00020000);

    // halo DMA: lane (r8, c8) of piece p fills physical chunk c8 of halo row p*8 + r8 (source-side swizzle)
    unsigned hoff1[HPW], hoff2[HPW];
#pragma unroll
    for (int j = 0; j < HPW; ++j) {
        const int hr = (j * NW + wave) * 8 + r8;
        const int seg = hr / hs, rem = hr - seg * hs;
        const int hy = rem / (W + 2), hx = rem - hy * (W + 2);
        const int img = img0 + seg;
        const int yy = y0 + hy - 1, xx = hx - 1;
        const bool ok = (hr < nh) && (yy >= 0) && (yy < H) && (xx >= 0) && (xx < W) && ((long)img * HW < a.M);
        const long row = ((long)img * H + yy) * W + xx;
        const int swz = (c8 ^ swz_key(hr)) * 8;
        hoff1[j] = ok ? (unsigned)((row * a.ld1 + swz) * 2) : OOB;
        hoff2[j] = ok ? (unsigned)((row * a.ld2 + swz) * 2) : OOB;
    }
    unsigned woff[WPW];
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int R = (j * NW + wave) * 8 + r8;
        const bool ok = (R < BN) && (n0 + R < a.n_end);
        woff[j] = ok ? (unsigned)(((long)(n0 + R) * a.ktot + (c8 ^ swz_key(R)) * 8) * 2) : OOB;
    }
    // halo row of each of this wave's 16-token tiles (this lane's token), before the tap shift
    int hbase[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int ml = (wm * TM + t) * 16 + l16;
        const int seg = ml / seg_tok, rem = ml - seg * seg_tok;
        const int y = rem / W, x = rem - y * W;
        hbase[t] = seg * hs + (y + 1) * (W + 2) + (x + 1);
        // pinned in a VGPR: left to itself the compiler re-derives it in EVERY slab from (y, x) -- a v_mul_lo_u32 and four more
        // VALU per tile, 256 VGPRs and a spill; pinned the slab's address block is 4 VALU per tile (tools/isa_mix.py)
        asm volatile("" : "+v"(hbase[t]));
    }

    const int ctot = a.c1 + a.c2;
    auto issue_halo = [&](int cs, int hb) {
        const int cbase = cs * BK;
        char* dst = hbuf + hb * HBYTES;
        if (cbase < a.c1) {
#pragma unroll
            for (int j = 0; j < HPW; ++j)
                if ((j * NW + wave) * 8 < nh)                           // (pieces wholly past the halo: nobody reads them)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x1, (lds_void*)(dst + (j * NW + wave) * 1024), 16, hoff1[j] + cbase * 2, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < HPW; ++j)
                if ((j * NW + wave) * 8 < nh)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x2, (lds_void*)(dst + (j * NW + wave) * 1024), 16, hoff2[j] + (cbase - a.c1) * 2, 0, 0, 0);
        }
    };
    auto issue_w = [&](int slab, int wb) {       // slab = cs * 9 + tap
        const int cs = slab / 9, tap = slab - cs * 9;
        const unsigned koff = (unsigned)((tap * ctot + cs * BK) * 2);
        char* dst = wbuf + wb * WBYTES;
#pragma unroll
        for (int j = 0; j < WPW; ++j)
            if ((j + 1) * NW * 8 <= BN || (j * NW + wave) * 8 < BN)      // (pieces wholly past the tile's rows: nobody reads them)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(dst + (j * NW + wave) * 1024), 16, woff[j] + koff, 0, 0, 0);
    };
    // Fragment addresses.  Chunk (4 ks + lq) ^ swz == (lq ^ swz) ^ (4 ks): the ks = 1 address of a row is the
    // ks = 0 address XOR 64, so a slab needs ONE address per 16-token tile (tap shift added to the lane's halo
    // row, swizzle key re-derived from the shifted row) plus one XOR -- about 6 VALU per tile and slab.  The
    // weight rows of a wave share the swizzle key of l16 ((80 wn + 16 c) / 2 is a multiple of 4), so their
    // addresses are one per-lane base plus immediates.
    const int wlane = lds_off(wn * TN * 16 + l16, lq);
    int xaddr[TM];
    auto read_frags = [&](int slab, int wb, int ks, half8 (&xf)[TM], half8 (&wf)[TN]) {
        if (CID_ABL(8)) return;   // profiling knob: no LDS fragment reads (MFMAs run on stale registers)
        const char* ws = wbuf + wb * WBYTES + (ks ? (wlane ^ 64) : wlane);
        if (ks == 0) {
            const int cs = slab / 9, tap = slab - cs * 9;
            const int ty = tap / 3;
            const int shift = (ty - 1) * (W + 2) + (tap - ty * 3 - 1);
            const int hsel = (cs & 1) * HBYTES;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int row = hbase[t] + shift;
                xaddr[t] = hsel + row * 128 + ((lq ^ swz_key(row)) << 4);
            }
        }
#pragma unroll
        for (int t = 0; t < TM; ++t) xf[t] = *reinterpret_cast<const half8*>(hbuf + (ks ? (xaddr[t] ^ 64) : xaddr[t]));
#pragma unroll
        for (int c = 0; c < TN; ++c) wf[c] = *reinterpret_cast<const half8*>(ws + c * 2048);
    };

    f32x4v acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int c = 0; c < TN; ++c) acc[t][c] = f32x4v{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](const half8 (&xf)[TM], const half8 (&wf)[TN]) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int c = 0; c < TN; ++c) acc[t][c] = mfma16(wf[c], xf[t], acc[t][c]);
    };

    // split-K over whole channel slabs (9 taps stay together)
    const int ncs = a.cslabs;
    const int cs_begin = (int)((long)ncs * blockIdx.z / a.splitk);
    const int cs_end = (int)((long)ncs * (blockIdx.z + 1) / a.splitk);
    const int s_begin = cs_begin * 9, s_end = cs_end * 9;

    half4 rpre[TM][TN];
    const bool rpre_valid = prefetch_residual<TM, TN>(a, rpre, m0, n0, wm, wn, l16, lq);
#ifdef CID_CONV_TRACE
    const bool tr_on = (blockIdx.y * gridDim.x + blockIdx.x == 100) && blockIdx.z == 0 && (wave == 0 || wave == NW / 2);
    const int tr_slot = wave == 0 ? 0 : 1;
    int tr_n = 0;
    unsigned long long tr_ts[8] = {};
#endif
#if CID_HALO_STAGGER
    // ---- half-slab offset pipeline ------------------------------------------------------------------------------------
    // The second half of the workgroup (waves NW/2 .., hosted on the SAME four SIMDs as waves 0 .. NW/2-1) runs the same
    // one-barrier-per-slab loop HALF A SLAB out of phase.  With BAR(s) the barrier that publishes slab s:
    //   first half :  BAR(s+1) | issue DMA(s+2) | read k0(s+1) | MFMA k1(s) | read k1(s+1) | MFMA k0(s+1) | BAR(s+2)
    //   second half:  BAR(s+1) | MFMA k0(s) | issue DMA(s+2) | read k0(s+1) | MFMA k1(s) | read k1(s+1) | BAR(s+2)
    // i.e. the second half arrives at a barrier with BOTH fragment sets of the previous slab still unmultiplied: right
    // behind the barrier it feeds the matrix pipe while the first half issues DMA and LDS reads, and vice versa half a
    // slab later.  Both halves have pulled slab s out of LDS before BAR(s+1), so DMA(s+2) may overwrite its stage; both
    // read slab s+1 only behind BAR(s+1).  Same LDS, same registers, same barrier count as the lock-step form.
    const bool second = wave >= NW / 2;
    half8 xf0[TM], wf0[TN], xf1[TM], wf1[TN];
    issue_halo(cs_begin, cs_begin & 1);
    issue_w(s_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (second) __builtin_amdgcn_s_setprio(1);      // the later-dispatched half loses every arbitration otherwise
    auto issue_next = [&](int slab, int stage) {   // behind BAR(slab+1): W(slab+2) into the stage of `slab`, a new channel slab's halo
        if (slab + 2 < s_end && !CID_ABL(1)) issue_w(slab + 2, stage);
        const int cs = (slab + 1) / 9;
        if ((slab + 1) - cs * 9 == 0 && cs + 1 < cs_end && !CID_ABL(4)) issue_halo(cs + 1, (cs + 1) & 1);
    };
    if (!second) {
        read_frags(s_begin, 0, 0, xf0, wf0);
        if (s_begin + 1 < s_end) issue_w(s_begin + 1, 1);
        if (cs_begin + 1 < cs_end) issue_halo(cs_begin + 1, (cs_begin + 1) & 1);   // second halo buffer is free from the start
        frags_landed(xf0); frags_landed(wf0);
        int cur = 0;
        for (int slab = s_begin; slab < s_end; ++slab) {
            CONV_STAMP(0);
            read_frags(slab, cur, 1, xf1, wf1);
            CONV_STAMP(1);
            mma(xf0, wf0);
            __builtin_amdgcn_sched_barrier(0);
            CONV_STAMP(2);
            frags_landed(xf1); frags_landed(wf1);
            if (slab + 1 < s_end) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                CONV_STAMP(3);
                __builtin_amdgcn_s_barrier();
                CONV_STAMP(4);
                issue_next(slab, cur);
                CONV_STAMP(5);
                read_frags(slab + 1, cur ^ 1, 0, xf0, wf0);
                CONV_STAMP(6);
            }
            mma(xf1, wf1);
            __builtin_amdgcn_sched_barrier(0);
            CONV_STAMP(7);
            CONV_FLUSH();
            frags_landed(xf0); frags_landed(wf0);
            cur ^= 1;
        }
    } else {
        if (s_begin + 1 < s_end) issue_w(s_begin + 1, 1);
        if (cs_begin + 1 < cs_end) issue_halo(cs_begin + 1, (cs_begin + 1) & 1);
        read_frags(s_begin, 0, 0, xf0, wf0);
        read_frags(s_begin, 0, 1, xf1, wf1);
        int cur = 0;
        for (int slab = s_begin; slab < s_end; ++slab) {
            const bool more = slab + 1 < s_end;
            CONV_STAMP(0);
            if (more) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                CONV_STAMP(1);
                __builtin_amdgcn_s_barrier();
            }
            CONV_STAMP(2);
            frags_landed(xf0); frags_landed(wf0);
            mma(xf0, wf0);
            __builtin_amdgcn_sched_barrier(0);
            CONV_STAMP(3);
            if (more) {
                issue_next(slab, cur);
                CONV_STAMP(4);
                read_frags(slab + 1, cur ^ 1, 0, xf0, wf0);
            }
            CONV_STAMP(5);
            frags_landed(xf1); frags_landed(wf1);
            mma(xf1, wf1);
            __builtin_amdgcn_sched_barrier(0);
            CONV_STAMP(6);
            if (more) read_frags(slab + 1, cur ^ 1, 1, xf1, wf1);
            CONV_STAMP(7);
            CONV_FLUSH();
            cur ^= 1;
        }
    }
    __builtin_amdgcn_s_setprio(0);
#else
    half8 xf0[TM], wf0[TN], xf1[TM], wf1[TN];
    issue_halo(cs_begin, cs_begin & 1);
    issue_w(s_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(s_begin, 0, 0, xf0, wf0);
    if (s_begin + 1 < s_end) issue_w(s_begin + 1, 1);
    if (cs_begin + 1 < cs_end) issue_halo(cs_begin + 1, (cs_begin + 1) & 1);   // second halo buffer is free from the start
    frags_landed(xf0); frags_landed(wf0);

    int cur = 0;
    for (int slab = s_begin; slab < s_end; ++slab) {
        CONV_STAMP(0);                                                 // 0: slab begins (behind the barrier)
        read_frags(slab, cur, 1, xf1, wf1);
        CONV_STAMP(1);                                                 // 1: reads of k-step 1 issued
        mma(xf0, wf0);
        __builtin_amdgcn_sched_barrier(0);
        CONV_STAMP(2);                                                 // 2: MFMAs of k-step 0 issued
        frags_landed(xf1); frags_landed(wf1);
        if (slab + 1 < s_end) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            CONV_STAMP(3);                                             // 3: next slab landed, own reads landed
            __builtin_amdgcn_s_barrier();
            CONV_STAMP(4);                                             // 4: behind the barrier
            if (slab + 2 < s_end && !CID_ABL(1)) issue_w(slab + 2, cur);
            // next channel slab's halo: requested as soon as its buffer is free (the previous slab's last
            // tap has been read by every wave), i.e. right after the barrier that starts a new channel slab
            const int cs = (slab + 1) / 9;
            if ((slab + 1) - cs * 9 == 0 && cs + 1 < cs_end && !CID_ABL(4)) issue_halo(cs + 1, (cs + 1) & 1);
            CONV_STAMP(5);                                             // 5: DMA issued
            read_frags(slab + 1, cur ^ 1, 0, xf0, wf0);
            CONV_STAMP(6);                                             // 6: reads of the next k-step 0 issued
        }
        mma(xf1, wf1);
        __builtin_amdgcn_sched_barrier(0);
        CONV_STAMP(7);                                                 // 7: MFMAs of k-step 1 issued
        CONV_FLUSH();
        frags_landed(xf0); frags_landed(wf0);
        cur ^= 1;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float lmean[TM] = {}, lrstd[TM] = {};
    igemm_epilogue<TM, TN, false, false>(a, acc, m0, n0, wm, wn, l16, lq, smem, wave, lmean, lrstd, NW, WN, rpre, rpre_valid);
#endif
}

#ifdef CID_CONV_TRACE
extern "C" int cid_debug_conv_trace(unsigned long long* host, int64_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_trace), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

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
        if (a.out2) *reinterpret_cast<half4*>(a.out2 + (long)m * a.ldo + n) = o;
    }
}

template <int TM, int TN, int WM, int WN, bool VMODE, bool LN, bool NLOOP = false, int NBUF = 2, int ATT_D = 0>
int launch_one_ln(const GemmArgs& a, int ncols, hipStream_t s) {
    constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
    constexpr int NW = WM * WN;
    constexpr int STAGE = ((BM / 8 + NW - 1) / NW) * NW * 1024 + BN * 128;   // x rows incl. scratch pieces + exactly BN weight rows
    constexpr int SMEM = NBUF * STAGE;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    auto kern = igemm_kernel<TM, TN, WM, WN, VMODE, NBUF, LN, NLOOP, ATT_D>;
    static bool configured = false;
    if (!configured) {
        hipError_t herr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (herr != hipSuccess) {
            cid_set_error("cid_gemm_f16: cannot reserve %d bytes of LDS (%s)", SMEM, hipGetErrorString(herr));
            return -5;
        }
        configured = true;
    }
    dim3 grid((ncols + BN - 1) / BN / (NLOOP ? a.nloop : 1), (a.M + BM - 1) / BM, VMODE ? 1 : a.splitk);
    GemmArgs b = a;
    b.xcd_pn = cidg::choose_xcd_pn((int)grid.x, (int)grid.y, 2.0 * ncols * a.ktot, (double)a.bytes_x1 + a.bytes_x2);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), SMEM, s, b);
    return 0;
}

template <int TM, int TN, int WM, int WN, bool VMODE>
int launch_one(const GemmArgs& a, int ncols, hipStream_t s) {
    if constexpr (!VMODE && TN == 4 && WM * WN == 8) {      // the GEGLU tiles: N-loop form (plan_gemm sets a.nloop)
        if (a.nloop > 1)
            return a.ln_s ? launch_one_ln<TM, TN, WM, WN, false, true, true>(a, ncols, s)
                          : launch_one_ln<TM, TN, WM, WN, false, false, true>(a, ncols, s);
    }
    // deeper ring (plan_gemm sets a.nbuf) for the 160-wide family: 3 stages = 156 / 108 / 84 KB at 256 / 128 / 64 tokens
    if constexpr (!VMODE && TN == 5) {
        if (a.nbuf == 3)
            return a.ln_s ? launch_one_ln<TM, TN, WM, WN, false, true, false, 3>(a, ncols, s)
                          : launch_one_ln<TM, TN, WM, WN, false, false, false, 3>(a, ncols, s);
    }
    return a.ln_s ? launch_one_ln<TM, TN, WM, WN, VMODE, true>(a, ncols, s) : launch_one_ln<TM, TN, WM, WN, VMODE, false>(a, ncols, s);
}

// mode 3: query projection + attention epilogue on tiles of whole heads (two ring stages: the Q / O tile reuses them)
template <int TM, int TN, int WM, int WN, int ATT_D>
int launch_att(GemmArgs a, hipStream_t s) {
    a.n_begin = 0; a.n_end = a.N;
    return a.ln_s ? launch_one_ln<TM, TN, WM, WN, false, true, false, 2, ATT_D>(a, a.N, s)
                  : launch_one_ln<TM, TN, WM, WN, false, false, false, 2, ATT_D>(a, a.N, s);
}

template <int TM, int TN, int WM, int WN>
int launch_halo(GemmArgs a, hipStream_t s) {
    constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN, NW = WM * WN;
    constexpr int SMEM = 2 * (7 * NW * 1024) + 2 * (((BN / 8 + NW - 1) / NW) * NW * 1024);
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    auto kern = igemm_halo_kernel<TM, TN, WM, WN>;
    static bool configured = false;
    if (!configured) {
        hipError_t herr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (herr != hipSuccess) {
            cid_set_error("cid_gemm_f16: cannot reserve %d bytes of LDS (%s)", SMEM, hipGetErrorString(herr));
            return -5;
        }
        configured = true;
    }
    a.n_begin = 0; a.n_end = a.N;
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.splitk);
    a.xcd_pn = cidg::choose_xcd_pn((int)grid.x, (int)grid.y, 2.0 * a.N * a.ktot, (double)a.bytes_x1 + a.bytes_x2);
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), SMEM, s, a);
    if (a.splitk > 1) {
        const long items = (long)a.M * (a.N >> 2);
        const int g = (int)((items + 255) / 256 > 2048 ? 2048 : (items + 255) / 256);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(g), dim3(256), 0, s, a);
    }
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

int cidg::choose_xcd_pn(int gx, int gy, double w_bytes, double x_bytes) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("CID_XCD_2D"); on = e ? atoi(e) : 1; }
    if (!on || ((long)gx * gy) % 8 != 0) return 0;
    int best = 0;
    double cost = 0.0;
    for (int pn = 1; pn <= 8; pn *= 2) {
        const int pm = 8 / pn;
        if (gx % pn != 0 || gy % pm != 0) continue;
        const double c = pm * w_bytes + pn * x_bytes;
        if (best == 0 || c < cost) { best = pn; cost = c; }      // (ties keep the smaller pn: the order of rounds 2-5)
    }
    return best;
}

enum TileCfg { A256x160, B128x160, C64x160, G256x128, G128x128, O64x64, O128x32 };

// argument checks + tile / split-K choice of one cid_gemm_f16 call (no launch): shared by the call itself and by
// cid_gemm_stats_rows, which tells the host how the GroupNorm statistics of that call will be blocked
static int plan_gemm(const cid_gemm_desc* d, GemmArgs& a, TileCfg& cfg, bool& halo, int& bm_out, bool& h32, bool& g32) {
    h32 = false;
    g32 = false;
    CID_CHECK_ARG(d && d->x1 && d->w && d->out, "cid_gemm_f16: null pointer");
    CID_CHECK_ARG(d->taps == 1 || d->taps == 9, "cid_gemm_f16: taps must be 1 or 9 (got %d)", d->taps);
    CID_CHECK_ARG(d->c1 > 0 && d->c1 % 32 == 0 && d->c2 >= 0 && d->c2 % 32 == 0 && (d->c1 + d->c2) % 64 == 0 &&
                  (d->c2 == 0 || d->c1 % 64 == 0),
                  "cid_gemm_f16: channel counts must be multiples of 64 (c1=%d c2=%d)", d->c1, d->c2);
    CID_CHECK_ARG(d->c2 == 0 || d->x2, "cid_gemm_f16: c2 > 0 needs x2");
    CID_CHECK_ARG(d->N > 0 && d->N % 32 == 0 && d->M > 0, "cid_gemm_f16: bad M/N (%d, %d)", d->M, d->N);
    CID_CHECK_ARG(d->mode >= 0 && d->mode <= 3, "cid_gemm_f16: bad mode %d", d->mode);
    CID_CHECK_ARG(d->ld1 % 8 == 0 && d->ldo % 8 == 0 && (d->c2 == 0 || d->ld2 % 8 == 0) && (!d->res || d->ldr % 8 == 0),
                  "cid_gemm_f16: row pitches (ld1, ld2, ldo, ldr) must keep 16-byte alignment");
    CID_CHECK_ARG((((uintptr_t)d->out | (uintptr_t)d->res | (uintptr_t)d->out2) & 15) == 0,
                  "cid_gemm_f16: out / out2 / res must be 16-byte aligned (rows are stored and the residual is read in 16-byte chunks)");
    CID_CHECK_ARG((d->ln_s == nullptr) == (d->ln_b == nullptr), "cid_gemm_f16: ln_s and ln_b come together");
    CID_CHECK_ARG(!d->ln_s || (d->taps == 1 && d->c2 == 0 && !d->bias && d->ln_eps > 0.f),
                  "cid_gemm_f16: the LayerNorm fold applies to one-source linears; the bias belongs in ln_b");
    a.x1 = (const half_t*)d->x1; a.x2 = (const half_t*)d->x2;
    a.c1 = d->c1; a.c2 = d->c2; a.ld1 = d->ld1; a.ld2 = d->ld2;
    a.w = (const half_t*)d->w; a.out = (half_t*)d->out; a.ldo = d->ldo;
    a.out2 = (half_t*)d->out2;
    CID_CHECK_ARG(!d->out2 || d->mode == 0, "cid_gemm_f16: out2 (a second destination) goes with the plain epilogue, mode 0");
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
    a.nloop = 1;
    a.nbuf = 2;
    a.xcd_pn = 0;
    a.att_kp = (const half_t*)d->att_kp; a.att_vp = (const half_t*)d->att_vp; a.att_kvrow = (const int*)d->att_kvrow;
    a.att_n_txt = d->att_n_txt; a.att_n_ip = d->att_n_ip; a.att_scale = d->att_ip_scale;
    a.att_krow = a.att_vrow = 0;
    a.ws = (float*)d->ws;
    a.ln_s = d->ln_s; a.ln_b = d->ln_b; a.ln_eps = d->ln_eps;
    a.gn_stats = d->gn_stats; a.gn_unit = d->N / 32;
    {
#if defined(CID_GEMM_ABLATION)      // experiment builds only (build.py --variant ... CID_GEMM_ABLATION)
        static int ablate = -1;
        if (ablate < 0) { const char* e = getenv("CID_GEMM_ABLATE"); ablate = e ? atoi(e) : 0; }
        a.ablate = ablate;
#else
        a.ablate = 0;
#endif
    }
    {
        // rows addressable through x1 / x2: the input image for convs, M rows for linears
        const long rows_in = (d->taps == 9) ? (long)(d->M / (d->Ho * d->Wo)) * d->Hi * d->Wi : (long)d->M;
        const long b1 = ((rows_in - 1) * d->ld1 + d->c1) * 2, b2 = d->c2 ? ((rows_in - 1) * d->ld2 + d->c2) * 2 : 0;
        const long bw = (long)d->N * a.ktot * 2;
        CID_CHECK_ARG(b1 < 0x7fffffffL && b2 < 0x7fffffffL && bw < 0x7fffffffL, "cid_gemm_f16: tensor exceeds 2 GiB");
        a.bytes_x1 = (unsigned)b1; a.bytes_x2 = (unsigned)b2; a.bytes_w = (unsigned)bw;
    }
    a.n_begin = 0; a.n_end = a.N;
    if (d->taps == 9) {
        CID_CHECK_ARG(d->Hi > 0 && d->Wi > 0 && d->Ho > 0 && d->Wo > 0 && (d->stride == 1 || d->stride == 2)
                      && (d->up == 0 || d->up == 1), "cid_gemm_f16: bad conv geometry");
        CID_CHECK_ARG(d->M % (d->Ho * d->Wo) == 0, "cid_gemm_f16: M is not batch * Ho * Wo");
    }

    if (d->mode == 3) {
        // query projection with the identity cross-attention as its epilogue: tiles of whole heads inside one sample
        CID_CHECK_ARG(d->att_kp && d->att_vp && d->att_kvrow, "cid_gemm_f16: mode 3 needs att_kp / att_vp / att_kvrow");
        CID_CHECK_ARG(d->taps == 1 && d->c2 == 0 && !d->bias && !d->rowbias && !d->res && !d->gn_stats,
                      "cid_gemm_f16: mode 3 is a plain projection (no bias / residual / statistics)");
        CID_CHECK_ARG(d->heads > 0 && d->dhead > 0 && d->N == d->heads * d->dhead && (d->dhead == 64 || d->dhead == 80 || d->dhead == 160),
                      "cid_gemm_f16: mode 3 needs N = heads * dhead with dhead in {64, 80, 160} (got %d x %d, N = %d)", d->heads, d->dhead, d->N);
        CID_CHECK_ARG(d->att_n_txt == 77 && d->att_n_ip == 4, "cid_gemm_f16: mode 3 is built for the reference's 77 + 4 context (got %d + %d)",
                      d->att_n_txt, d->att_n_ip);
        CID_CHECK_ARG(d->ntok > 0 && d->M % d->ntok == 0 && d->ntok % 64 == 0, "cid_gemm_f16: mode 3 needs ntok (tokens per sample, a multiple of 64)");
        const int qks = (d->dhead + 15) / 16, dvt = (d->dhead + 31) / 32;
        a.att_krow = (long)d->heads * 3 * qks * 512;
        a.att_vrow = (long)d->heads * dvt * 6 * 512;
        if (d->dhead == 64) { cfg = G128x128; bm_out = 128; CID_CHECK_ARG(d->N % 128 == 0 && d->ntok % 128 == 0, "cid_gemm_f16: mode 3, dhead 64: N and ntok multiples of 128"); }
        else {
            // 160-wide tiles span whole heads (two of 80 channels, one of 160): N must be a whole number of them
            CID_CHECK_ARG(d->N % 160 == 0, "cid_gemm_f16: mode 3, dhead %d: N = %d is not a multiple of the 160-channel tile", d->dhead, d->N);
            if (d->ntok % 128 == 0 && (long)(d->M / 128) * (d->N / 160) >= 256) { cfg = B128x160; bm_out = 128; }
            else { cfg = C64x160; bm_out = 64; }
        }
        halo = false;
        return 0;
    }
    // ---- tile choice: aim for >= 2 waves on each of the 1024 SIMDs ------------------------
    const int n_plain = (d->mode == 2) ? d->n_vt0 : d->N;
    const long target = 2048;
    auto waves = [&](int bm_, int bn_, int w) {
        return (long)((a.M + bm_ - 1) / bm_) * ((n_plain + bn_ - 1) / bn_) * w;
    };
    int bm = 0, bn = 0, nw = 0;
    // GEGLU with short K (few slabs): the erf epilogue and the pipeline prologue dominate a tile's
    // life, so prefer the tile that lets two workgroups share a CU and overlap them (measured:
    // 156 -> 115 us at M=32768, N=2560, K=320; plain epilogues do not benefit)
    // 256-token tiles only for deep K AND at least 1024 of them (in situ: SDXL's M = 4096, K = 1280 level +0.9 % end to end
    // with the big tile, SD1.5's M = 2048 level +0.3 % with the small one)
    bool small_tiles = (d->mode == 1) && (a.nslab <= 10 || waves(256, 128, 1) < 1024);
    if (d->mode == 1) {
        static int g_tile = -1;
        if (g_tile < 0) { const char* e = getenv("CID_GEGLU_TILE"); g_tile = e ? atoi(e) : 0; }
        if (g_tile == 1) small_tiles = false;
        if (g_tile == 2) small_tiles = true;
        if (d->N % 128 != 0) { cfg = O64x64; bm = 64; bn = 64; nw = 4; }
        else if (waves(256, 128, 8) >= target && !small_tiles) { cfg = G256x128; bm = 256; bn = 128; nw = 8; }
        else { cfg = G128x128; bm = 128; bn = 128; nw = 8; }
    } else if (n_plain % 160 == 0) {
        // plain epilogue: the largest tile that still yields >= 256 workgroups, if necessary with the help
        // of split-K (small-M levels are weight-traffic bound: W is re-read once per token tile)
        static int f_tile = -1, f_sk = -1;
        if (f_tile < 0) { const char* e = getenv("CID_GEMM_TILE"); f_tile = e ? atoi(e) : 0; }
        if (f_sk < 0) { const char* e = getenv("CID_GEMM_SK"); f_sk = e ? atoi(e) : 0; }
        const bool can_split = (d->mode == 0) && a.ws != nullptr && !a.ln_s;
        auto tiles = [&](int bm_) { return (long)((a.M + bm_ - 1) / bm_) * (n_plain / 160); };
        auto sk_for = [&](int bm_) {
            long t = tiles(bm_);
            int sk = (int)((256 + t - 1) / t);
            if (!can_split) sk = 1;
            if (sk > 16) sk = 16;
            while (sk > 1 && a.nslab / sk < 6) --sk;
            while (sk > 1 && (int64_t)sk * a.M * a.N * 4 > d->ws_bytes) --sk;
            return sk;
        };
        int pick = 0;
        bool nosplit = false;
        static int prefer128 = -1;
        if (prefer128 < 0) { const char* e = getenv("CID_GEMM_PREFER128"); prefer128 = e ? atoi(e) : 1; }
        if (f_tile) pick = f_tile;
        else if (d->taps == 1 && tiles(256) >= 256) {
            // enough 256-token tiles without split-K; the fused QKV projection (no split possible, short K) prefers
            // twice as many half-size tiles when the big ones only just fill the chip (measured 63 -> 55 us at SDXL's
            // 32x32 level, 45 -> 42 us at SD1.5's 32x32 level)
            // (and 45 -> 41 us at SD1.5's 64x64 level, where the big tiles number exactly 512: CID_GEMM_TILE A/B, round 4)
            // Round 6 re-measured the rule for every linear (CID_GEMM_TILE A/B at SD1.5 CFG batch 8 / 16 and SDXL batch 4,
            // profiles/r06_tile_rule.txt): 128-token tiles win wherever they number >= 256 -- 320 -> 320 at 64 x 64 17.8 -> 14.9 us
            // (33.1 -> 26.9 at CFG batch 16), SDXL's 640 -> 640 24.4 -> 21.1, ff2 at 1280 channels 68.6 -> 63.2 (unsplit instead of
            // 256-token tiles + split-K 2).  One 256-token workgroup per CU loads, multiplies and stores in lock step with every
            // other CU; two half-size workgroups per CU are out of phase.  CID_GEMM_PREFER128=0: the rule of rounds 3-5.
            pick = (prefer128 || (d->mode == 2 && tiles(256) <= 512 && tiles(128) >= 512)) ? 2 : 1;
            nosplit = true;
        } else if (d->taps == 1 && tiles(128) >= 256 && (a.nslab <= 40 || prefer128)) { pick = 2; nosplit = true; }   // no fp32 partials
        else if (d->taps == 1 && tiles(64) >= 256 && a.nslab <= 20) { pick = 3; nosplit = true; }      // beats 256-tiles + split-K
        else if (tiles(256) * sk_for(256) >= 256) pick = 1;                                             // (tools/sweep_tiles*.sh)
        else if (tiles(128) * sk_for(128) >= 256) pick = 2;
        else pick = 3;
        if (pick == 1)      { cfg = A256x160; bm = 256; bn = 160; nw = 8; }
        else if (pick == 2) { cfg = B128x160; bm = 128; bn = 160; nw = 8; }
        else                { cfg = C64x160;  bm = 64;  bn = 160; nw = 4; }
        a.splitk = f_sk ? f_sk : (nosplit ? 1 : sk_for(bm));
        if (!can_split || (int64_t)a.splitk * a.M * a.N * 4 > d->ws_bytes || a.nslab < a.splitk) a.splitk = 1;
    } else if (d->mode == 0 && n_plain % 128 == 0) {
        // widths off the 160 grid (VAE decoder: 128 / 256 / 512 channels, attention score / value GEMMs)
        if (waves(256, 128, 8) >= target) { cfg = G256x128; bm = 256; bn = 128; nw = 8; }
        else { cfg = G128x128; bm = 128; bn = 128; nw = 8; }
    } else if (n_plain % 64 == 0) { cfg = O64x64; bm = 64; bn = 64; nw = 4; }
    else { cfg = O128x32; bm = 128; bn = 32; nw = 4; }
    if (d->mode == 1) CID_CHECK_ARG(d->N % 64 == 0, "cid_gemm_f16: GEGLU needs N %% 64 == 0");
    if (d->mode == 1 && (cfg == G128x128 || cfg == G256x128) && d->taps == 1 && d->c2 == 0 && a.M % bm == 0) {
        // N-loop: one workgroup walks several n-tiles of its token tile (igemm_kernel, NLOOP) -- as many as leave one round of
        // resident workgroups; the count must divide the n-tiles (the flattened slab sequence has no ragged tail)
        static int f_nl = -1;
        if (f_nl < 0) { const char* e = getenv("CID_GEGLU_NLOOP"); f_nl = e ? atoi(e) : 0; }      // A/B switch: 1 = off, n = force
        const int nt = d->N / bn;
        const long tiles = (long)(a.M / bm) * nt;
        int nl = f_nl > 0 ? f_nl : (int)(tiles / (cfg == G128x128 ? 512 : 256));      // (128-token tiles: two workgroups per CU)
        if (nl > nt) nl = nt;
        while (nl > 1 && nt % nl != 0) --nl;
        if (nl > 1 && (long)nl * bn * a.ktot * 2 < 0x7fffffffL) a.nloop = nl;
    }
    if (d->mode == 1 && d->taps == 1 && d->c2 == 0 && !a.ln_s && d->N % 160 == 0 && a.M % 256 == 0 && a.cslabs >= 16) {
        // linear_h32.hip: 256 x 160 tiles of 32 x 32 x 16 MFMAs, loader / compute wave roles, N-loop -- the 16 x 16 x 32 tiles
        // above are LDS-bandwidth-bound on this op (profiles/r06_gemm_ablation.txt).  Deep K only (>= 1024 channels): the erf
        // epilogue of that kernel is exposed once per n-tile (one compute wave per SIMD), which costs more than the leaner loop
        // gains at K = 320 (86 vs 73 us at SD1.5's 64 x 64 level), draws at K = 640 and wins at K = 1280 (64 vs 77 us).  One
        // round of 256 workgroups: every workgroup walks tiles / 256 n-tiles (a divisor of the n-tile count); launches that
        // cannot fill the chip stay above.
        static int f_g32 = -1;
        if (f_g32 < 0) { const char* e = getenv("CID_GEGLU_H32"); f_g32 = e ? atoi(e) : 1; }      // A/B switch: 0 = off
        const int nt = d->N / 160;
        const long tiles = (long)(a.M / 256) * nt;
        int nl = (int)(tiles / 256);
        if (nl > nt) nl = nt;
        while (nl > 1 && nt % nl != 0) --nl;
        if (f_g32 && tiles >= 256 && nl >= 1 && (long)nl * 160 * a.ktot * 2 < 0x7fffffffL) {
            g32 = true;
            a.nloop = nl;
            bm = 256;
        }
    }
    if (d->mode == 2) {
        CID_CHECK_ARG(d->vt && d->ntok % 16 == 0 && d->M % d->ntok == 0 && d->n_vt0 % bn == 0 && d->dhead > 0
                      && d->heads > 0 && d->dvp >= d->dhead && (d->N - d->n_vt0) % 16 == 0,
                      "cid_gemm_f16: bad QKV/V^T description");
    }
    // split-K for small-M / deep-K problems (plain epilogue only)
    if (d->mode == 0 && a.ws && !a.ln_s && a.nslab >= 16 && n_plain % 160 != 0) {
        const long w = waves(bm, bn, nw);
        if (w < target) {
            int sk = (int)((target + w - 1) / w);
            if (sk > 8) sk = 8;
            while (sk > 1 && a.nslab / sk < 8) --sk;
            while (sk > 1 && (int64_t)sk * a.M * a.N * 4 > d->ws_bytes) --sk;
            a.splitk = sk;
        }
    }
    {
        // three-stage ring: launches of the 128- / 64-token tiles that put at most one workgroup on a CU anyway (<= 256
        // workgroups) and walk enough slabs for the lookahead to matter
        static int f_nb = -1;
        if (f_nb < 0) { const char* e = getenv("CID_GEMM_NBUF"); f_nb = e ? atoi(e) : 0; }      // A/B switch: 2 = never, 3 = whenever legal
        const bool legal = (cfg == A256x160 || cfg == B128x160 || cfg == C64x160) && d->mode != 1;
        const long wgs = (long)((a.M + bm - 1) / bm) * ((n_plain + bn - 1) / bn) * a.splitk;
        // (the 256-token tile holds one workgroup per CU whatever its ring: three stages whenever there are slabs to look ahead;
        //  the smaller tiles only where a third stage does not cost a co-resident workgroup)
        const bool want = cfg == A256x160 ? (a.nslab / a.splitk >= 4) : (wgs <= 256 && a.nslab / a.splitk >= 8);
        if (legal && (f_nb == 3 || (f_nb == 0 && want))) a.nbuf = 3;
    }
    static int no_halo = -1;
    if (no_halo < 0) { const char* e = getenv("CID_GEMM_NOHALO"); no_halo = e ? atoi(e) : 0; }
    halo = false;
    if (cfg == A256x160 && !no_halo && d->mode == 0 && d->taps == 9 && d->stride == 1 && d->up == 0 && d->Wo == d->Wi &&
        d->Ho == d->Hi) {
        // halo kernel: the 256-token tile must be whole image rows of one image, or whole images
        const int HW = d->Ho * d->Wo;
        const int seg = 256 < HW ? 256 : HW;
        const bool rows_ok = (seg % d->Wo == 0) && (HW % seg == 0) && (256 % seg == 0) && (d->M % 256 == 0);
        const int nh = (256 / seg) * (seg / d->Wo + 2) * (d->Wo + 2);
        if (rows_ok && nh <= 448) {
            halo = true;
            if (a.splitk > a.cslabs) a.splitk = a.cslabs;     // split over whole channel slabs only
        }
    }
    {
        // conv3x3.hip (32 x 32 MFMA tiles, loader / compute wave roles) for the stride-1 3x3 convolutions on the 160-channel
        // grid, unsplit: 256-token tiles where they fill the chip; 128-token tiles where those do and K is short (they are
        // LDS-bound: measured faster than 256-token tiles + split-K up to 10 channel slabs -- the 32 x 32 level's 640 -> 640,
        // the first resnet of the CFG-deduplicated level 0 -- and slower beyond).  Three weight stages next to two halo buffers
        // need a halo of <= 400 rows.  Everything else stays on the halo kernel above (+ splitk_epilogue_kernel).
        static int no_h32 = -1;
        static int only256 = 0;
        if (no_h32 < 0) { const char* e = getenv("CID_CONV_H32"); no_h32 = (e && atoi(e) == 0) ? 1 : 0; only256 = (e && atoi(e) == 2) ? 1 : 0; }
        const int HW = d->taps == 9 ? d->Ho * d->Wo : 0;
        // (Upsample2D's convolution, up == 1: the halo holds input pixels; a tile must be an even number of whole output rows of
        //  one image, starting on an even row)
        const bool shape_ok = !no_h32 && d->mode == 0 && d->taps == 9 && d->stride == 1 && (d->up == 0 || d->up == 1) &&
                              d->Wo == (d->Wi << d->up) && d->Ho == (d->Hi << d->up) && d->N % 160 == 0 && HW >= 64 &&
                              (!d->rowbias || (a.rows_per_sample >= 64 && a.rows_per_sample % 64 == 0));
        for (int bm_try = 256; shape_ok && !h32 && bm_try >= (only256 ? 256 : 128); bm_try >>= 1) {
            const int seg = bm_try < HW ? bm_try : HW;
            if (seg % d->Wo != 0 || HW % seg != 0 || bm_try % seg != 0 || d->M % bm_try != 0) continue;
            if (d->up && (seg != bm_try || (seg / d->Wo) % 2 != 0)) continue;
            const int nh = (bm_try / seg) * (((seg / d->Wo) >> d->up) + 2) * ((d->Wo >> d->up) + 2);
            if (nh > 400) continue;
            const long tiles = (long)(d->M / bm_try) * (d->N / 160);
            if (tiles < 256 || (bm_try == 128 && a.cslabs > 10)) continue;
            h32 = true; halo = false;
            bm = bm_try;
            a.splitk = 1;
        }
    }
    bm_out = bm;
    // GroupNorm statistics come out of the plain, unsplit epilogue of the 160-wide tiles, whole tiles only
    if (a.gn_stats) {
        const bool ok = d->mode == 0 && a.splitk == 1 && (cfg == A256x160 || cfg == B128x160 || cfg == C64x160) &&
                        d->N % 32 == 0 && 80 % a.gn_unit == 0 && d->M % bm == 0;
        CID_CHECK_ARG(ok, "cid_gemm_f16: gn_stats requested for a launch that cannot emit them (ask cid_gemm_stats_rows first)");
    }
    return 0;
}

extern "C" int cid_gemm_stats_rows(const cid_gemm_desc* d) {
    if (!d) return 0;
    cid_gemm_desc q = *d;
    q.gn_stats = nullptr;
    GemmArgs a;
    TileCfg cfg;
    bool halo, h32, g32;
    int bm = 0;
    if (plan_gemm(&q, a, cfg, halo, bm, h32, g32) != 0) return 0;
    const int unit = d->N / 32;
    const bool ok = d->mode == 0 && a.splitk == 1 && (cfg == A256x160 || cfg == B128x160 || cfg == C64x160) &&
                    d->N % 32 == 0 && unit > 0 && 80 % unit == 0 && d->M % bm == 0;
    return ok ? bm : 0;
}

extern "C" int cid_gemm_f16(const cid_gemm_desc* d, cid_stream_t stream) {
    GemmArgs a;
    TileCfg cfg;
    bool halo, h32, g32;
    int bm = 0;
    int rc = plan_gemm(d, a, cfg, halo, bm, h32, g32);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (g32) {
        a.n_begin = 0; a.n_end = a.N;
        rc = cidg::launch_geglu_h32(a, s);
        if (rc) return rc;
        CID_CHECK_LAUNCH("cid_gemm_f16");
        return 0;
    }
    if (a.mode == 3) {
        if (cfg == G128x128) rc = launch_att<2, 4, 4, 2, 64>(a, s);
        else if (cfg == B128x160) rc = a.dhead == 80 ? launch_att<2, 5, 4, 2, 80>(a, s) : launch_att<2, 5, 4, 2, 160>(a, s);
        else rc = a.dhead == 80 ? launch_att<2, 5, 2, 2, 80>(a, s) : launch_att<2, 5, 2, 2, 160>(a, s);
        if (rc) return rc;
        CID_CHECK_LAUNCH("cid_gemm_f16");
        return 0;
    }
    if (h32) {
        a.n_begin = 0; a.n_end = a.N;
        rc = cidg::launch_conv_h32(a, bm, s);
        if (rc) return rc;
        CID_CHECK_LAUNCH("cid_gemm_f16");
        return 0;
    }
    if (halo) {
        rc = launch_halo<4, 5, 4, 2>(a, s);
        if (rc) return rc;
        CID_CHECK_LAUNCH("cid_gemm_f16");
        return 0;
    }
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
