// VARIANT BUILDS ONLY (python -m consistentid_amd.build --variant x2 CID_WITH_XATTN2): the second generation of the
// fused identity cross-attention, kept as an A/B comparator; the product library does not contain it.
// Fused identity cross-attention, second generation (SD1.5 level-0 geometry: C = 320, 8 heads of 40):
// Consistent_IPAttProcessor.__call__ (/root/reference/attention.py:207-294) wrapped in the
// BasicTransformerBlock's  x += attn2(LayerNorm(x), ehs)  -- ONE launch, LoRA merged, LayerNorm folded.
//
//   q   = LN(x) Wq^T = rstd * (x Wq'^T - mean * s) + b'          Wq' = Wq diag(gamma) (x d^-0.5 log2 e),
//                                                                 s = Wq' 1, b' = Wq beta          (:236)
//   o_h = softmax_text(q_h K_h^T) V_h + scale * softmax_ID(q_h Kip_h^T) Vip_h   two softmaxes    (:259-279)
//   out = o Wo^T + b_o (+ x)                                                                       (:282)
//
// Roofline: MFMA (dense fp16 2.5 PFLOP/s).  Algorithmic work per (sample, layer): 4 N C^2 + 4 N 81 C flop;
// bytes 4 N C (x in, out written once) + 4 C^2 (weights, per workgroup from L2) + 4 * 81 * C (K, V).
//
// Why this shape.  At B2 = 8, N = 4096 a CU owns 128 tokens: 66 MFLOP against 80 KB in, 80 KB out and
// ~570 KB of weights / K / V from L2.  Round 1 staged x, LayerNorm-ed it, wrote Q and O back to LDS and
// walked 40 weight slabs behind 40 barriers; 70 % of that kernel was skeleton.  Here:
//   * x arrives by LDS-DMA in five 64-channel slabs, each consumed by the Q projection as it lands
//     (LayerNorm is folded into the weights, the per-token mean / rstd are accumulated from the very
//     fragments the MFMAs consume), x is read from HBM ONCE and stays in LDS for the residual;
//   * a wave owns 64 tokens x 80 channels = 64 tokens x TWO WHOLE HEADS in both projections, so the
//     Q tile never leaves its registers: the 16x16x32 accumulator layout (lane = token, 4 consecutive
//     channels per register quad) IS a legal B-operand layout once K is packed with the matching
//     contraction order -- same for P -> P.V and, through LDS, for O -> out-projection;
//   * weights stream L2 -> LDS by DMA in 64-deep slabs (one barrier per 1280 MFMA cycles per SIMD),
//     the next slab in flight while one is consumed; the first two Wo slabs land during the attention;
//   * softmax denominators come out of the matrix pipe (a constant 0/1 A operand), the ID stream is
//     rescaled inside P so that ONE accumulator serves both streams.
// LDS: T[128][320] fp16 (x, later O) 80 KB + two 40 KB weight slabs = 160 KB, one workgroup per CU.
#include "xattn_frag.h"
#include "../../include/cid.h"

namespace {

#ifndef X2_VLOAD_AHEAD
#define X2_VLOAD_AHEAD 0      // V^T fragments requested this many token tiles before the end of pass 1 (0: at the start of pass 2)
#endif

constexpr int XC = 320, XD = 40, XNH = 8, XBT = 128;
constexpr int X_TSLAB = XBT * 128;            // bytes of one 64-channel slab of the token tile
constexpr int X_TBYTES = 5 * X_TSLAB;         // 81920
constexpr int X_WSLOT = XC * 128;             // bytes of one 64-deep weight slab (320 rows x 128 B)
constexpr int X_SMEM = X_TBYTES + 2 * X_WSLOT;  // 163840
constexpr int X_KF = 12, X_VF = 9;            // 1-KiB fragments per head: K 6 key tiles x 2 k-steps, V^T 3 row tiles x 3 k-steps
constexpr long X_KROW = (long)XNH * X_KF * 512, X_VROW = (long)XNH * X_VF * 512;   // halfs per context row
constexpr int X_STG_PITCH = 176;              // epilogue staging row pitch (bytes): 80 channels + pad

// Experiment builds only (python -m consistentid_amd.build --variant trace CID_X2_TRACE): every wave stamps the shader
// clock at its phase boundaries; tools/x2_trace.py turns the stamps into a per-phase timeline.  Not in the product build.
#ifdef CID_X2_TRACE
__device__ unsigned long long g_x2_trace[4096 * 8 * 32];
#define X2_STAMP(k) do { if (lane == 0) g_x2_trace[((long)blockIdx.x * 8 + wave) * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define X2_STAMP(k) do { } while (0)
#endif

// NT / NI: context layout fixed at compile time (the score predicates fold away), NT = 0: run-time layout
template <int NT, int NI>
__global__ void __launch_bounds__(512)
id_xattn2_kernel(const half_t* __restrict__ x, half_t* __restrict__ out,
                 const half_t* __restrict__ wq, const float* q_rowsum, const float* q_bias,
                 const half_t* __restrict__ wo, const half_t* bo,
                 const half_t* kp, const half_t* vp, const int* __restrict__ kvrow,
                 int N, int tiles_per_sample, int total_tiles, int n_txt_rt, int n_ip_rt,
                 float ip_scale, float ln_eps, int flags) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    const int n_txt = NT ? NT : n_txt_rt;
    const int n_all = NT ? NT + NI : n_txt_rt + n_ip_rt;
    const bool has_ln = (flags & 1) != 0, add_res = (flags & 2) != 0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lq = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;          // token half (64 tokens), head pair (80 channels)

    // workgroup -> (sample, token tile); consecutive tiles of a sample share an XCD (its L2 keeps that sample's K/V)
    int id = blockIdx.x;
    if ((total_tiles & 7) == 0) id = (id & 7) * (total_tiles >> 3) + (id >> 3);
    const int sample = id / tiles_per_sample;
    const long tok0 = (long)sample * N + (long)(id - sample * tiles_per_sample) * XBT;

    // ------------------------------------------------------------------ DMA plumbing
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(x + tok0 * XC), 0, XBT * XC * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc((void*)wq, 0, XC * XC * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)wo, 0, XC * XC * 2, 0x00020000);
    // a 1-KiB piece = 8 rows x 128 B of a slab; LDS image is linear, the XOR swizzle sits on the source address:
    // physical 16-B chunk pc of row r holds logical chunk pc ^ ((r >> 1) & 7)
    unsigned xoff[2], woff[5];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (2 * wave + j) * 8 + (lane >> 3);
        xoff[j] = (unsigned)((r * XC + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2);
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int r = (wave + 8 * j) * 8 + (lane >> 3);
        woff[j] = (unsigned)((r * XC + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2);
    }
    auto issue_x = [&](int s) {       // x[:, 64 s .. 64 s + 64) -> T slab s
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(smem + s * X_TSLAB + (2 * wave + j) * 1024), 16,
                                                     xoff[j], s * 128, 0, 0);
    };
    auto issue_w = [&](int G) {       // stage G: Wq' slab G (G < 5) or Wo slab G - 5, into ring slot G & 1
        char* st = smem + X_TBYTES + (G & 1) * X_WSLOT;
        if (G < 5) {
#pragma unroll
            for (int j = 0; j < 5; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (lds_void*)(st + (wave + 8 * j) * 1024), 16, woff[j], G * 128, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_o, (lds_void*)(st + (wave + 8 * j) * 1024), 16, woff[j], (G - 5) * 128, 0, 0);
        }
    };
    // fragment addresses (bytes): token tile row / weight row r, 16-B chunk c of the slab
    auto t_frag = [&](int slab, int r, int c) -> const half8* {
        return reinterpret_cast<const half8*>(smem + slab * X_TSLAB + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
    };
    auto w_frag = [&](int slot, int r, int c) -> const half8* {
        return reinterpret_cast<const half8*>(smem + X_TBYTES + slot * X_WSLOT + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
    };
    // 8-byte access to token-tile element (row r, channel ch), ch % 4 == 0
    auto t_quad = [&](int r, int ch) -> half4* {
        return reinterpret_cast<half4*>(smem + (ch >> 6) * X_TSLAB + r * 128 + ((((ch >> 3) & 7) ^ ((r >> 1) & 7)) << 4) + (ch & 4) * 2);
    };

    X2_STAMP(0);
    // prologue: the first two stages; later x slabs are requested two stages ahead of their use -- asking for all of
    // x at once makes every CU's first slab wait behind 21 MB of HBM traffic (slab 0 ready after 5.8k cycles instead of 4.3k)
    issue_x(0); issue_w(0); issue_x(1); issue_w(1);

    // ------------------------------------------------------------------ projection pass over five 64-deep slabs
    // acc[ct][tt]: channel tile ct (16 of the wave's 80 channels) x token tile tt (16 of its 64 tokens);
    // lane (l16, lq) holds token l16, channels 4 lq .. 4 lq + 3 of the tile.
    f32x4v acc[5][4];
    float ssum[4], ssq[4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int ct = 0; ct < 5; ++ct)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[ct][tt] = f32x4v{0.f, 0.f, 0.f, 0.f};
    };
    auto slab_mfma = [&](int slot, int tslab, bool stats) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 af[5], bf[4];
#pragma unroll
            for (int ct = 0; ct < 5; ++ct) af[ct] = *w_frag(slot, wn * 80 + ct * 16 + l16, ks * 4 + lq);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) bf[tt] = *t_frag(tslab, wm * 64 + tt * 16 + l16, ks * 4 + lq);
            if (stats) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const half2v h = {bf[tt][2 * j], bf[tt][2 * j + 1]};
                        const half2v one = {(half_t)1.f, (half_t)1.f};
                        ssum[tt] = __builtin_amdgcn_fdot2(h, one, ssum[tt], false);
                        ssq[tt] = __builtin_amdgcn_fdot2(h, h, ssq[tt], false);
                    }
                // (pinned: otherwise hipcc sinks all 320 dot products -- and the 160 registers of x fragments they
                // read -- into the `has_ln` branch after the loop)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) asm volatile("" : "+v"(ssum[tt]), "+v"(ssq[tt]));
            }
#pragma unroll
            for (int ct = 0; ct < 5; ++ct)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc[ct][tt] = mfma16(af[ct], bf[tt], acc[ct][tt]);
        }
    };

    // ------------------------------------------------------------------ phase A: Q^T = Wq' x^T while x streams in
    zero_acc();
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) { ssum[tt] = 0.f; ssq[tt] = 0.f; }
#pragma unroll
    for (int g = 0; g < 5; ++g) {
        // stage g = (x slab g, Wq' slab g) has landed once only the younger pieces are outstanding.  Issue order per wave:
        // x0 W0 x1 W1 | x2 | W2 x3 | W3 x4 | W4 | Wo0   (x: 2 pieces, W: 5 pieces, "|" = the barriers below)
        if (g == 3) X2_STAMP(14);
        if (g == 0) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");         // x1 W1 may be in flight
        else if (g == 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                // the x slab of stage g + 1
        if (g == 3) X2_STAMP(15);
        __builtin_amdgcn_s_barrier();       // publishes stage g, retires stage g - 1 (its ring slot is free)
        asm volatile("" ::: "memory");
        if (g == 0) X2_STAMP(1);
        if (g == 1) X2_STAMP(2);
        if (g == 3) X2_STAMP(16);
        if (g == 4) X2_STAMP(17);
        if (g >= 1) issue_w(g + 1);         // g = 4: the first Wo slab
        if (g + 2 < 5) issue_x(g + 2);
        slab_mfma(g & 1, g, true);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4v sv[5], bv[5];                    // fold vectors of the wave's 80 channels (zeros when there is no LayerNorm)
#pragma unroll
    for (int ct = 0; ct < 5; ++ct) {
        sv[ct] = *reinterpret_cast<const f32x4v*>(q_rowsum + wn * 80 + ct * 16 + 4 * lq);
        bv[ct] = *reinterpret_cast<const f32x4v*>(q_bias + wn * 80 + ct * 16 + 4 * lq);
    }

    X2_STAMP(3);
    asm volatile("; MARK finalize");
    // per-token LayerNorm statistics from the fragments the MFMAs consumed (each lane saw 80 of the 320 channels)
    float mean[4], rstd[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        if (has_ln) {
            const float s1 = rows_sum(ssum[tt]) * (1.f / XC);
            const float s2 = rows_sum(ssq[tt]) * (1.f / XC);
            mean[tt] = s1;
            rstd[tt] = rsqrtf(fmaxf(s2 - s1 * s1, 0.f) + ln_eps);
        } else {
            mean[tt] = 0.f;
            rstd[tt] = 1.f;
        }
    }
    // Q (fp16) in accumulator layout: qh[ct][tt] = channels 16 ct + 4 lq .. + 3 of token tt * 16 + l16
    half4 qh[5][4];
#pragma unroll
    for (int ct = 0; ct < 5; ++ct) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            f32x4v q;
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = fmaf(rstd[tt], fmaf(-mean[tt], sv[ct][i], acc[ct][tt][i]), bv[ct][i]);
            qh[ct][tt] = cvt4(q);
            pin(qh[ct][tt]);
        }
    }

    X2_STAMP(4);
    asm volatile("; MARK barrierE");
    const long ctx_row = kvrow[sample];
    const half_t* kpr = kp + ctx_row * X_KROW + lane * 8;
    const half_t* vpr = vp + ctx_row * X_VROW + lane * 8;
    half8 kf[6][2];
    auto load_k = [&](int h) {
#pragma unroll
        for (int kt = 0; kt < 6; ++kt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#ifdef X2_EXPERIMENT_NO_KV_TRAFFIC      // timing experiment only (wrong results): how much of phase B is K/V delivery?
                half8 c; for (int j = 0; j < 8; ++j) c[j] = (half_t)(0.01f * (float)((lane + kt + ks + j) & 7));
                kf[kt][ks] = c;
#else
                kf[kt][ks] = ld_global_h8(kpr + ((long)h * X_KF + kt * 2 + ks) * 512);
#endif
            }
    };
    __builtin_amdgcn_sched_barrier(0);      // (the accumulators are dead from here on)
    load_k(2 * wn);                         // first head's K fragments travel across the barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // every wave is done with x as an operand and with ring slot 0
    asm volatile("" ::: "memory");
    X2_STAMP(5);
    issue_w(6);                             // second Wo slab; both land while the attention runs

    // ------------------------------------------------------------------ phase B: two-stream attention on the wave's two heads
    // constant A operand that makes the matrix pipe emit the two softmax denominators:
    // row (4 q' + 0) = 1 on text keys, row (4 q' + 1) = 1 on ID keys  ->  every lane gets l_text in reg 0, l_id in reg 1
    half8 ones_a[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int key = 32 * ks + 16 * (j >> 2) + 4 * lq + (j & 3);
            const bool t = key < n_txt, ip = key >= n_txt && key < n_all;
            ones_a[ks][j] = ((l16 & 3) == 0 && t) || ((l16 & 3) == 1 && ip) ? (half_t)1.f : (half_t)0.f;
        }
    // residual rows of this wave's output tile: each head's part is fetched right before that head's O overwrites it
    half4 resid[5][4];
    {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = 2 * wn + hh;
    asm volatile("; MARK pass1");
            // pass 1: scores, softmax numerators and the per-stream rescale, for the four token tiles
            half8 pb[4][3];
            float inv_lt[4];
            half8 vf[3][3];
            auto load_v = [&]() {
#pragma unroll
                for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
#ifdef X2_EXPERIMENT_NO_KV_TRAFFIC
                        half8 c; for (int j = 0; j < 8; ++j) c[j] = (half_t)(0.01f * (float)((lane + dt + ks + j) & 7));
                        vf[dt][ks] = c;
#else
                        vf[dt][ks] = ld_global_h8(vpr + ((long)h * X_VF + dt * 3 + ks) * 512);
#endif
                    }
            };
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                // V^T fragments of this head are requested X2_VLOAD_AHEAD token tiles before pass 1 ends
                if (tt == 4 - X2_VLOAD_AHEAD) load_v();
                // head A = channel tiles 0, 1 and rows 0..7 of tile 2; head B = rows 8..15 of tile 2 and tiles 3, 4;
                // K is packed with zeros where a k-slot belongs to the other head (or to no channel)
                const half8 qb0 = hh == 0 ? cat4(qh[0][tt], qh[1][tt]) : cat4(qh[3][tt], qh[4][tt]);
                const half8 qb1 = cat4(qh[2][tt], qh[2][tt]);
                f32x4v s[6];
#pragma unroll
                for (int kt = 0; kt < 6; ++kt) {
                    s[kt] = mfma16(kf[kt][0], qb0, f32x4v{0.f, 0.f, 0.f, 0.f});
                    s[kt] = mfma16(kf[kt][1], qb1, s[kt]);
                }
                // the two row maxima (text keys / ID keys), lane-local then across the four lane rows
                float mt = -INFINITY, mi = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 6; ++kt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int key = 16 * kt + 4 * lq + i;
                        const int cls = key_class(NT, NI, kt, i);
                        if (cls == 0) mt = fmaxf(mt, s[kt][i]);
                        else if (cls == 2) {
                            mt = fmaxf(mt, key < n_txt ? s[kt][i] : -INFINITY);
                            mi = fmaxf(mi, (key >= n_txt && key < n_all) ? s[kt][i] : -INFINITY);
                        }
                    }
                mt = rows_max(mt);
                mi = rows_max(mi);
#pragma unroll
                for (int kt = 0; kt < 6; ++kt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int key = 16 * kt + 4 * lq + i;
                        const int cls = key_class(NT, NI, kt, i);
                        if (cls == 0) s[kt][i] = __builtin_amdgcn_exp2f(s[kt][i] - mt);
                        else if (cls == 1) s[kt][i] = 0.f;
                        else {
                            const bool t = key < n_txt, ip = key >= n_txt && key < n_all;
                            const float e = __builtin_amdgcn_exp2f(s[kt][i] - (t ? mt : mi));
                            s[kt][i] = (t || ip) ? e : 0.f;
                        }
                    }
                half8 p[3];
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) p[ks] = cat4(cvt4(s[2 * ks]), cvt4(s[2 * ks + 1]));
                f32x4v l = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) l = mfma16(ones_a[ks], p[ks], l);
                // o = (sum_text p v + rho sum_id p v) / l_text,  rho = scale * l_text / l_id
                const float rho = l[1] > 0.f ? ip_scale * l[0] / l[1] : 0.f;
                inv_lt[tt] = 1.f / l[0];
#pragma unroll
                for (int kt = 0; kt < 6; ++kt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int key = 16 * kt + 4 * lq + i;
                        const int cls = key_class(NT, NI, kt, i);
                        if (cls == 2) {
                            const bool ip = key >= n_txt && key < n_all;
                            p[kt >> 1][(kt & 1) * 4 + i] = (half_t)(s[kt][i] * (ip ? rho : 1.f));
                        }
                    }
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) pb[tt][ks] = p[ks];
                __builtin_amdgcn_sched_barrier(0);     // one token tile's scores at a time (register budget)
            }
            if (hh == 0) X2_STAMP(6);
            if (hh == 1) X2_STAMP(7);
    asm volatile("; MARK pass1end");
            __builtin_amdgcn_sched_barrier(0);     // keep pass 2's loads out of pass 1 (register budget)
            if (add_res) {
#pragma unroll
                for (int ct = (hh == 0 ? 0 : 3); ct < (hh == 0 ? 3 : 5); ++ct)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) resid[ct][tt] = *t_quad(wm * 64 + tt * 16 + l16, wn * 80 + ct * 16 + 4 * lq);
            }
    asm volatile("; MARK pass2");
            // pass 2: O^T = V^T P^T (rows of V^T follow the channel tiles of the wave), O -> T over x
            if (X2_VLOAD_AHEAD == 0) load_v();
            if (hh == 0) load_k(h + 1);         // the other head's K fragments travel under this head's P.V
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int r = wm * 64 + tt * 16 + l16;
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    f32x4v o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) o = mfma16(vf[dt][ks], pb[tt][ks], o);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] *= inv_lt[tt];
                    const int ct = dt + 2 * hh;
                    // the shared tile 2: rows 0..7 (lane rows 0, 1) are head A's, rows 8..15 head B's
                    if (ct != 2 || (hh == 0 ? lq < 2 : lq >= 2)) *t_quad(r, wn * 80 + ct * 16 + 4 * lq) = cvt4(o);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    X2_STAMP(8);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // O of every head is in T; both Wo slabs of the ring have landed
    asm volatile("" ::: "memory");

    X2_STAMP(9);
    asm volatile("; MARK phaseC");
    // ------------------------------------------------------------------ phase C: out^T = Wo O^T
    zero_acc();
#pragma unroll
    for (int g = 0; g < 5; ++g) {
        if (g >= 1) {
            if (g == 2) X2_STAMP(18);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (g == 2) X2_STAMP(19);
            __builtin_amdgcn_s_barrier();
            if (g == 2) X2_STAMP(20);
            if (g == 3) X2_STAMP(21);
            asm volatile("" ::: "memory");
            if (g + 1 < 5) issue_w(5 + g + 1);
        }
        slab_mfma((5 + g) & 1, g, false);
    }
    X2_STAMP(10);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // T and the ring are dead: LDS becomes the store staging area
    asm volatile("" ::: "memory");

    X2_STAMP(11);
    asm volatile("; MARK epilogue");
    // epilogue: + bias + residual, transposed through LDS so that every lane stores 16 B of a whole output row
    {
        char* stg = smem + wave * (64 * X_STG_PITCH);
#pragma unroll
        for (int ct = 0; ct < 5; ++ct) {
            const int ch = wn * 80 + ct * 16 + 4 * lq;
            f32x4v bb = {0.f, 0.f, 0.f, 0.f};
            if (bo) {
                const half4 b4 = *reinterpret_cast<const half4*>(bo + ch);
#pragma unroll
                for (int i = 0; i < 4; ++i) bb[i] = (float)b4[i];
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                f32x4v v = acc[ct][tt];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += bb[i];
                if (add_res) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)resid[ct][tt][i];
                }
                *reinterpret_cast<half4*>(stg + (tt * 16 + l16) * X_STG_PITCH + (ct * 16 + 4 * lq) * 2) = cvt4(v);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the staging tile is private to the wave
        half_t* ob = out + (tok0 + wm * 64) * XC + wn * 80;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int e = k * 64 + lane;
            const int r = e / 10, c = e - r * 10;
            *reinterpret_cast<half8*>(ob + (long)r * XC + c * 8) = *reinterpret_cast<const half8*>(stg + r * X_STG_PITCH + c * 16);
        }
        X2_STAMP(12);
#ifdef CID_X2_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        X2_STAMP(13);
    }
#endif
}

}  // namespace

#ifdef CID_X2_TRACE
extern "C" int cid_debug_x2_trace(unsigned long long* host, int64_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_x2_trace), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int cid_id_xattn2_supported(int32_t C, int32_t heads, int32_t n_txt, int32_t n_ip) {
    // geometry: SD1.5 level 0; context layouts: the reference's 77 + 4 (UNet) and 81 plain keys (ControlNet)
    return (C == XC && heads == XNH && ((n_txt == 77 && n_ip == 4) || (n_txt == 81 && n_ip == 0))) ? 1 : 0;
}

extern "C" int cid_id_xattn2_f16(const cid_half* x, cid_half* out, const cid_half* wq_folded, const float* q_rowsum,
                                 const float* q_bias, const cid_half* wo, const cid_half* bo, const cid_half* kp,
                                 const cid_half* vp, const int32_t* kvrow, int32_t B, int32_t N, int32_t C, int32_t heads,
                                 int32_t n_txt, int32_t n_ip, float ip_scale, float ln_eps, int32_t flags,
                                 cid_stream_t stream) {
    CID_CHECK_ARG(x && out && wq_folded && q_rowsum && q_bias && wo && kp && vp && kvrow, "cid_id_xattn2_f16: null pointer");
    CID_CHECK_ARG(cid_id_xattn2_supported(C, heads, n_txt, n_ip),
                  "cid_id_xattn2_f16: built for C=%d, %d heads and a 77+4 or 81+0 context (got C=%d heads=%d context %d+%d)",
                  XC, XNH, C, heads, n_txt, n_ip);
    CID_CHECK_ARG(B > 0 && N > 0 && N % XBT == 0, "cid_id_xattn2_f16: N=%d must be a positive multiple of %d", N, XBT);
    CID_CHECK_ARG(x != out, "cid_id_xattn2_f16: in-place operation is not supported");
    // 0: the reference's 77 + 4 (UNet), 1: 81 plain keys (ControlNet's default attention)
    const int kind = (n_txt == 77 && n_ip == 4) ? 0 : 1;
    auto kern = kind == 0 ? id_xattn2_kernel<77, 4> : id_xattn2_kernel<81, 0>;
    static bool configured[2] = {false, false};
    if (!configured[kind]) {
        hipError_t herr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, X_SMEM);
        if (herr != hipSuccess) {
            cid_set_error("cid_id_xattn2_f16: cannot reserve %d bytes of LDS (%s)", X_SMEM, hipGetErrorString(herr));
            return -5;
        }
        configured[kind] = true;
    }
    const int tiles = N / XBT, total = tiles * B;
    hipLaunchKernelGGL(kern, dim3(total), dim3(512), X_SMEM, (hipStream_t)stream, (const half_t*)x, (half_t*)out,
                       (const half_t*)wq_folded, q_rowsum, q_bias, (const half_t*)wo, (const half_t*)bo,
                       (const half_t*)kp, (const half_t*)vp, kvrow, N, tiles, total, n_txt, n_ip, ip_scale, ln_eps, flags);
    CID_CHECK_LAUNCH("cid_id_xattn2_f16");
    return 0;
}
