// Fused identity cross-attention, third generation (SD1.5 level-0 geometry: C = 320, 8 heads of 40):
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
// What the second generation's phase trace said (profiles/r02_x2_trace.txt): one 8-wave workgroup per CU walks
// load -> Q projection -> attention -> out projection -> store strictly in sequence, every CU in the same phase at
// the same time, 13 workgroup barriers, the two projections co-bound by LDS bandwidth (weights staged through LDS are
// read back by two token halves), 3 064 VALU instructions per wave against 592 MFMAs.
// Here the tile is 64 tokens and the workgroup 4 waves, TWO workgroups per CU:
//   * a wave owns 64 tokens x 80 channels = 64 tokens x two whole heads, as before (Q stays in registers, the
//     accumulator layout is reused as a B operand for Q -> S, P -> O);
//   * with one token group per workgroup every weight row is consumed by exactly ONE wave, so staging weights
//     in LDS buys nothing: the host packs Wq' and Wo in fragment order (xattn_pack.pack_w3) and each wave
//     streams its own A operands L2 -> registers, 1 KiB per load instruction, two 64-deep slabs in flight.
//     No weight ring, no barrier inside the projections except the five that publish the x slabs;
//   * x (40 KB) and O (40 KB) have their own LDS tiles, so the residual is read where the epilogue needs it
//     (not parked in 40 registers across the attention) and the result is transposed IN PLACE over the wave's
//     own x bytes: no staging area, no barrier before the 16-byte row stores; the bias is the accumulators' start value;
//   * LayerNorm statistics: a wave reduces only its own 16 tokens and the four waves trade mean / rstd through
//     512 bytes of LDS; the fold vectors s, b' are DMA-ed into LDS ahead of the first x slab;
//   * keys in register-major order (xattn_pack.slot_key "reg"): every score register of a lane is all-text, all-ID or
//     absent except one partly filled register per stream -- no per-element predicates in the softmax; with paired
//     subtractions, hardware reciprocals and no NaN canonicalisation (-fno-honor-nans for this file, build.py) the
//     kernel issues 1 510 VALU instructions per wave.
// LDS: x tile 40 KB + O tile 40 KB = 80 KB, two workgroups per CU.  Barriers per workgroup: 5 + 2 + 1.
// Measured at B2 = 8, N = 4096: 27.7-28.8 us = 0.24 of the MFMA roofline (second generation 31.3-32.4 us).
#include "xattn_frag.h"
#include "../../include/cid.h"

namespace {

constexpr int YC = 320, YNH = 8, YBT = 64;
constexpr int Y_TSLAB = YBT * 128;            // bytes of one 64-channel slab of a token tile
constexpr int Y_TBYTES = 5 * Y_TSLAB;         // 40960
constexpr int Y_SMEM = 2 * Y_TBYTES;          // x tile, O tile
constexpr int Y_KF = 12, Y_VF = 9;            // 1-KiB fragments per head (same packing as the second generation)
constexpr long Y_KROW = (long)YNH * Y_KF * 512, Y_VROW = (long)YNH * Y_VF * 512;
constexpr long Y_WWAVE = 10 * 5 * 512;        // halfs of one wave's packed weight slice: 10 k-steps x 5 row tiles x 1 KiB

#ifndef X3_XAHEAD
#define X3_XAHEAD 2           // x slabs requested before the first one is consumed (2, 3 or 5 = the whole tile)
#endif
// Issue order per wave (x slab: 2 DMA pieces, W slab: 10 loads), A = X3_XAHEAD:
//   x0 W0 x1 W1 x2 .. x(A-1) | W2 xA | W3 x(A+1) | W4 x(A+2) | - | -          "|" = the slab barriers
// vmcnt that leaves only the operations YOUNGER than both x_g and W_g in flight when slab g starts:
constexpr int x3_slab_vmcnt(int ahead, int g) {
    int n_issued = 0, last_needed = 0;       // operations issued before slab g's wait; index after the later of x_g, W_g
    auto x = [&](int s) { n_issued += 2; if (s == g) last_needed = n_issued; };
    auto w = [&](int s) { n_issued += 10; if (s == g) last_needed = n_issued; };
    x(0); w(0); x(1); w(1);
    for (int s = 2; s < ahead; ++s) x(s);
    for (int r = 0; r < g; ++r) {            // regions completed before slab g
        if (r + 2 < 5) w(r + 2);
        if (r + ahead < 5) x(r + ahead);
    }
    return n_issued - last_needed;
}
static_assert(x3_slab_vmcnt(2, 0) == 12 && x3_slab_vmcnt(2, 3) == 12 && x3_slab_vmcnt(2, 4) == 0, "x3 wait table");
static_assert(x3_slab_vmcnt(3, 0) == 14 && x3_slab_vmcnt(3, 2) == 14 && x3_slab_vmcnt(3, 3) == 12, "x3 wait table");
static_assert(x3_slab_vmcnt(5, 0) == 18 && x3_slab_vmcnt(5, 1) == 16 && x3_slab_vmcnt(5, 2) == 10, "x3 wait table");

// Register-major key order (xattn_pack.slot_key, order "reg"): score register rho = 4 kt + i holds keys 4 rho .. 4 rho + 3
// in its four lane rows; text keys first, the ID keys start at the next whole register.  Class of a register:
//   0 text in every lane row | 1 absent | 2 text in lane rows lq < NT % 4 only | 3 ID in every lane row | 4 ID in lq < NI % 4
constexpr int reg_class(int NT, int NI, int rho) {
    const int ip0 = (NT + 3) / 4;
    if (rho < NT / 4) return 0;
    if (rho == NT / 4 && NT % 4) return 2;
    if (rho >= ip0 && rho < ip0 + NI / 4) return 3;
    if (rho == ip0 + NI / 4 && NI % 4) return 4;
    return 1;
}

// Experiment builds only (python -m consistentid_amd.build --variant trace CID_X3_TRACE): phase stamps per wave
#ifdef CID_X3_TRACE
__device__ unsigned long long g_x3_trace[8192 * 4 * 32];
#define X3_STAMP(k) do { if (lane == 0) g_x3_trace[((long)blockIdx.x * 4 + wn) * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define X3_STAMP(k) do { } while (0)
#endif

// NT / NI: context layout, fixed at compile time
template <int NT, int NI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))      // 256 VGPRs: two workgroups per CU
id_xattn3_kernel(const half_t* __restrict__ x, half_t* __restrict__ out,
                 const half_t* __restrict__ wqp, const float* q_rowsum, const float* q_bias,
                 const half_t* __restrict__ wop, const half_t* bo,
                 const half_t* kp, const half_t* vp, const int* __restrict__ kvrow,
                 int N, int tiles_per_sample, int total_tiles, float ip_scale, float ln_eps, int flags) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    static_assert(NT > 0 && 4 * ((NT + 3) / 4) + NI <= 96, "context does not fit the six key tiles");
    const bool has_ln = (flags & 1) != 0, add_res = (flags & 2) != 0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);     // the wave = head pair (80 channels), all 64 tokens
    const int l16 = lane & 15, lq = lane >> 4;
    const bool row_t = lq < NT % 4, row_i = lq < NI % 4;       // lane rows that hold a key in the partly filled registers

    // workgroup -> (sample, token tile); consecutive tiles of a sample share an XCD (its L2 keeps that sample's K/V)
    int id = blockIdx.x;
    if ((total_tiles & 7) == 0) id = (id & 7) * (total_tiles >> 3) + (id >> 3);
    const int sample = id / tiles_per_sample;
    const long tok0 = (long)sample * N + (long)(id - sample * tiles_per_sample) * YBT;

    // ------------------------------------------------------------------ x by LDS-DMA, weights straight into registers
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(x + tok0 * YC), 0, YBT * YC * 2, 0x00020000);
    // a 1-KiB piece = 8 rows x 128 B of a slab; LDS image is linear, the XOR swizzle sits on the source address:
    // physical 16-B chunk pc of row r holds logical chunk pc ^ ((r >> 1) & 7)
    unsigned xoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (2 * wn + j) * 8 + (lane >> 3);
        xoff[j] = (unsigned)((r * YC + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2);
    }
    auto issue_x = [&](int s) {       // x[:, 64 s .. 64 s + 64) -> slab s of the x tile (2 pieces per wave)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(smem + s * Y_TSLAB + (2 * wn + j) * 1024), 16,
                                                     xoff[j], s * 128, 0, 0);
    };
    // packed weights: [wave][k-step][row tile][lane][8] -- one load instruction = one A operand
    const half_t* wq_l = wqp + (long)wn * Y_WWAVE + lane * 8;
    const half_t* wo_l = wop + (long)wn * Y_WWAVE + lane * 8;
    half8 wf[2][2][5];                // [ring slot][k-step of the slab][row tile]: two 64-deep slabs of the wave's 80 rows
    auto load_w = [&](const half_t* base, int kstep, half8 (&dst)[5]) {
#pragma unroll
        for (int ct = 0; ct < 5; ++ct) dst[ct] = ld_global_h8(base + (long)(kstep * 5 + ct) * 512);
    };
    // fragment addresses (bytes): row r, 16-B chunk c of a slab of tile `base` (0: x, Y_TBYTES: O)
    auto t_frag = [&](int base, int slab, int r, int c) -> const half8* {
        return reinterpret_cast<const half8*>(smem + base + slab * Y_TSLAB + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
    };
    // 8-byte access to this lane's accumulator quad (token 16 tt + l16, channels 80 wn + 16 ct + 4 lq ..) in a tile:
    // the swizzle key (row >> 1) & 7 does not depend on tt, so one offset per channel tile serves all four token tiles
    unsigned qoff[5];
#pragma unroll
    for (int ct = 0; ct < 5; ++ct) {
        const int ch = wn * 80 + ct * 16 + 4 * lq;
        qoff[ct] = (unsigned)((ch >> 6) * Y_TSLAB + l16 * 128 + ((((ch >> 3) & 7) ^ ((l16 >> 1) & 7)) << 4) + (ch & 4) * 2);
    }
    auto t_quad = [&](int base, int tt, int ct) -> half4* {
        return reinterpret_cast<half4*>(smem + base + tt * 2048 + qoff[ct]);
    };

    X3_STAMP(0);
#ifdef CID_X3_TRACE
    if (lane == 0) {                        // where the wave runs: HW_ID (cu / sh / se) and XCC_ID
        g_x3_trace[((long)blockIdx.x * 4 + wn) * 32 + 30] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        g_x3_trace[((long)blockIdx.x * 4 + wn) * 32 + 31] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        g_x3_trace[((long)blockIdx.x * 4 + wn) * 32 + 28] = wall_clock64();      // 100 MHz, one counter for the whole chip
    }
#endif
    // the LayerNorm fold vectors of the wave's 80 channels go to LDS first (the O tile is idle until the attention): at
    // the end of the projection they are an LDS read away instead of an L2 round trip.  Lanes past the 20 chunks read
    // out of the descriptor's range (zeros), into the wave's own 1-KiB landing area.
    constexpr int Y_FOLD = Y_TBYTES + 4096;          // behind the 512-byte statistics mailbox
    if (has_ln) {
        const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)(q_rowsum + wn * 80), 0, 320, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(q_bias + wn * 80), 0, 320, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, (lds_void*)(smem + Y_FOLD + wn * 2048), 16, lane * 16, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(smem + Y_FOLD + wn * 2048 + 1024), 16, lane * 16, 0, 0, 0);
    }
    issue_x(0);
    load_w(wq_l, 0, wf[0][0]); load_w(wq_l, 1, wf[0][1]);
    issue_x(1);
    load_w(wq_l, 2, wf[1][0]); load_w(wq_l, 3, wf[1][1]);
#pragma unroll
    for (int sx = 2; sx < X3_XAHEAD; ++sx) issue_x(sx);
    __builtin_amdgcn_sched_barrier(0);

    // acc[ct][tt]: channel tile ct (16 of the wave's 80 channels) x token tile tt (16 of the 64 tokens);
    // lane (l16, lq) holds token l16, channels 4 lq .. 4 lq + 3 of the tile.
    f32x4v acc[5][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int ct = 0; ct < 5; ++ct)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[ct][tt] = f32x4v{0.f, 0.f, 0.f, 0.f};
    };
    auto step_mfma = [&](int tbase, int slab, int ks, const half8 (&af)[5]) {
        half8 bf[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) bf[tt] = *t_frag(tbase, slab, tt * 16 + l16, ks * 4 + lq);
#pragma unroll
        for (int ct = 0; ct < 5; ++ct)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[ct][tt] = mfma16(af[ct], bf[tt], acc[ct][tt]);
    };

    // K fragments of the first head are requested when the last x slab is in: they travel under its MFMAs
    const long ctx_row = kvrow[sample];
    const half_t* kpr = kp + ctx_row * Y_KROW + lane * 8;
    const half_t* vpr = vp + ctx_row * Y_VROW + lane * 8;
    half8 kf[6][2];
    auto load_k = [&](int h) {
#pragma unroll
        for (int kt = 0; kt < 6; ++kt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kf[kt][ks] = ld_global_h8(kpr + ((long)h * Y_KF + kt * 2 + ks) * 512);
    };
    f32x4v sv[5], bv[5];                    // fold vectors of the wave's 80 channels (not used without LayerNorm)

    // ------------------------------------------------------------------ phase A: Q^T = Wq' x^T while x streams in
    zero_acc();
    float ssum = 0.f, ssq = 0.f;          // LayerNorm sums of token 16 wn + l16 over the channels this lane row sees
#pragma unroll
    for (int g = 0; g < 5; ++g) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(x3_slab_vmcnt(X3_XAHEAD, g)) : "memory");
        __builtin_amdgcn_s_barrier();       // publishes x slab g (every wave has landed its two pieces)
        asm volatile("" ::: "memory");
        if (g == 0) X3_STAMP(1);
        if (g == 1) X3_STAMP(2);
        if (g == 4) {
            X3_STAMP(3);
            load_k(2 * wn);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (has_ln) {
                const half8 sf = *t_frag(0, g, wn * 16 + l16, ks * 4 + lq);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const half2v h = {sf[2 * j], sf[2 * j + 1]};
                    const half2v one = {(half_t)1.f, (half_t)1.f};
                    ssum = __builtin_amdgcn_fdot2(h, one, ssum, false);
                    ssq = __builtin_amdgcn_fdot2(h, h, ssq, false);
                }
            }
            step_mfma(0, g, ks, wf[g & 1][ks]);
            if (g + 2 < 5) load_w(wq_l, 2 * (g + 2) + ks, wf[g & 1][ks]);       // refill the half slab just consumed
        }
        if (g + X3_XAHEAD < 5) issue_x(g + X3_XAHEAD);
        __builtin_amdgcn_sched_barrier(0);
    }
    X3_STAMP(4);
    // per-token LayerNorm statistics: wave wn owns tokens 16 wn .. 16 wn + 15, the O tile (still unused) is the mailbox
    float mean[4], rstd[4];
    if (has_ln) {
        const float s1 = rows_sum(ssum) * (1.f / YC);
        const float s2 = rows_sum(ssq) * (1.f / YC);
        typedef float f32x2v __attribute__((ext_vector_type(2)));
        f32x2v* box = reinterpret_cast<f32x2v*>(smem + Y_TBYTES);
        if (lq == 0) box[wn * 16 + l16] = f32x2v{s1, rsqrtf(fmaxf(s2 - s1 * s1, 0.f) + ln_eps)};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const f32x2v st = box[tt * 16 + l16];
            mean[tt] = st[0];
            rstd[tt] = st[1];
        }
#pragma unroll
        for (int ct = 0; ct < 5; ++ct) {    // (landed long ago: they were requested before the first x slab)
            sv[ct] = *reinterpret_cast<const f32x4v*>(smem + Y_FOLD + wn * 2048 + (ct * 16 + 4 * lq) * 4);
            bv[ct] = *reinterpret_cast<const f32x4v*>(smem + Y_FOLD + wn * 2048 + 1024 + (ct * 16 + 4 * lq) * 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();       // every wave has read the mailbox: the O tile may be written from here on
        asm volatile("" ::: "memory");
    } else {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) { mean[tt] = 0.f; rstd[tt] = 1.f; }
#pragma unroll
        for (int ct = 0; ct < 5; ++ct) { sv[ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; bv[ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
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
    X3_STAMP(5);
    __builtin_amdgcn_sched_barrier(0);      // (the accumulators are dead from here on)

    // ------------------------------------------------------------------ phase B: two-stream attention on the wave's two heads
    // constant A operand that makes the matrix pipe emit the two softmax denominators:
    // row (4 q' + 0) = 1 on text keys, row (4 q' + 1) = 1 on ID keys  ->  every lane gets l_text in reg 0, l_id in reg 1
    half4 bias4[5];
    half8 ones_a[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cls = reg_class(NT, NI, 4 * (2 * ks + (j >> 2)) + (j & 3));     // k-slot (lq, j) = register (kt, i), lane row lq
            const bool t = cls == 0 || (cls == 2 && row_t), ip = cls == 3 || (cls == 4 && row_i);
            ones_a[ks][j] = ((l16 & 3) == 0 && t) || ((l16 & 3) == 1 && ip) ? (half_t)1.f : (half_t)0.f;
        }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int h = 2 * wn + hh;
        // pass 1: scores, softmax numerators and the per-stream rescale, for the four token tiles
        half8 pb[4][3];
        float inv_lt[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
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
                    const int cls = reg_class(NT, NI, 4 * kt + i);
                    if (cls == 0) mt = fmaxf(mt, s[kt][i]);
                    else if (cls == 2) mt = fmaxf(mt, row_t ? s[kt][i] : -INFINITY);
                    else if (cls == 3) mi = fmaxf(mi, s[kt][i]);
                    else if (cls == 4) mi = fmaxf(mi, row_i ? s[kt][i] : -INFINITY);
                }
            mt = rows_max(mt);
            if (NI) mi = rows_max(mi);
            // numerators: same-class neighbours two at a time (v_pk_add_f32), the partly filled registers behind a select
            typedef float f32x2v __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int kt = 0; kt < 6; ++kt)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    const int c0 = reg_class(NT, NI, 4 * kt + i), c1 = reg_class(NT, NI, 4 * kt + i + 1);
                    if (c0 == c1 && (c0 == 0 || c0 == 3)) {
                        const float m = c0 == 0 ? mt : mi;
                        const f32x2v d = f32x2v{s[kt][i], s[kt][i + 1]} - f32x2v{m, m};
                        s[kt][i] = __builtin_amdgcn_exp2f(d[0]);
                        s[kt][i + 1] = __builtin_amdgcn_exp2f(d[1]);
                        continue;
                    }
#pragma unroll
                    for (int u = i; u < i + 2; ++u) {
                        const int cls = reg_class(NT, NI, 4 * kt + u);
                        if (cls == 0) s[kt][u] = __builtin_amdgcn_exp2f(s[kt][u] - mt);
                        else if (cls == 3) s[kt][u] = __builtin_amdgcn_exp2f(s[kt][u] - mi);
                        else if (cls == 2) s[kt][u] = row_t ? __builtin_amdgcn_exp2f(s[kt][u] - mt) : 0.f;
                        else if (cls == 4) s[kt][u] = row_i ? __builtin_amdgcn_exp2f(s[kt][u] - mi) : 0.f;
                        else s[kt][u] = 0.f;
                    }
                }
            half8 p[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) p[ks] = cat4(cvt4(s[2 * ks]), cvt4(s[2 * ks + 1]));
            f32x4v l = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) l = mfma16(ones_a[ks], p[ks], l);
            // o = (sum_text p v + rho sum_id p v) / l_text,  rho = scale * l_text / l_id
            // (hardware reciprocals, 1 ulp; without ID keys l[1] = 0 and rho multiplies nothing)
            const float rho = ip_scale * l[0] * __builtin_amdgcn_rcpf(fmaxf(l[1], 1e-30f));
            inv_lt[tt] = __builtin_amdgcn_rcpf(l[0]);
#pragma unroll
            for (int kt = 0; kt < 6; ++kt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cls = reg_class(NT, NI, 4 * kt + i);
                    if (cls == 3 || cls == 4) p[kt >> 1][(kt & 1) * 4 + i] = (half_t)(s[kt][i] * rho);     // the ID registers
                }
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) pb[tt][ks] = p[ks];
            __builtin_amdgcn_sched_barrier(0);     // one token tile's scores at a time (register budget)
        }
        if (hh == 0) X3_STAMP(6);
        if (hh == 1) X3_STAMP(7);
        __builtin_amdgcn_sched_barrier(0);     // keep pass 2's loads out of pass 1 (register budget)
        // pass 2: O^T = V^T P^T (rows of V^T follow the channel tiles of the wave), O -> its own LDS tile
        half8 vf[3][3];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) vf[dt][ks] = ld_global_h8(vpr + ((long)h * Y_VF + dt * 3 + ks) * 512);
        if (hh == 0) load_k(h + 1);             // the other head's K fragments travel under this head's P.V
        else {                                  // ... and the first two Wo slabs (and the bias) under the last one
            load_w(wo_l, 0, wf[0][0]); load_w(wo_l, 1, wf[0][1]);
            load_w(wo_l, 2, wf[1][0]); load_w(wo_l, 3, wf[1][1]);
            // (a null bias reads q_bias' bytes instead and is masked later: one straight-line batch of loads, no branch)
            const half_t* bsrc = bo ? bo : reinterpret_cast<const half_t*>(q_bias);
#pragma unroll
            for (int ct = 0; ct < 5; ++ct) bias4[ct] = *reinterpret_cast<const half4*>(bsrc + wn * 80 + ct * 16 + 4 * lq);
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            typedef float f32x2v __attribute__((ext_vector_type(2)));
            const f32x2v inv2 = {inv_lt[tt], inv_lt[tt]};
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                f32x4v o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) o = mfma16(vf[dt][ks], pb[tt][ks], o);
                const f32x2v a = f32x2v{o[0], o[1]} * inv2, b = f32x2v{o[2], o[3]} * inv2;
                const half4 oh = {(half_t)a[0], (half_t)a[1], (half_t)b[0], (half_t)b[1]};
                const int ct = dt + 2 * hh;
                // the shared tile 2: rows 0..7 (lane rows 0, 1) are head A's, rows 8..15 head B's
                if (ct != 2 || (hh == 0 ? lq < 2 : lq >= 2)) *t_quad(Y_TBYTES, tt, ct) = oh;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    X3_STAMP(8);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // O of every head is in its tile (the Wo loads keep flying)
    asm volatile("" ::: "memory");
    X3_STAMP(9);

    // ------------------------------------------------------------------ phase C: out^T = Wo O^T, no barrier inside
    // the accumulators start from the bias (row 4 lq + i of a tile = channel, the same for every token column)
#pragma unroll
    for (int ct = 0; ct < 5; ++ct) {
        f32x4v bb;
#pragma unroll
        for (int i = 0; i < 4; ++i) bb[i] = bo ? (float)bias4[ct][i] : 0.f;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[ct][tt] = bb;
    }
#pragma unroll
    for (int g = 0; g < 5; ++g) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            step_mfma(Y_TBYTES, g, ks, wf[g & 1][ks]);
            if (g + 2 < 5) load_w(wo_l, 2 * (g + 2) + ks, wf[g & 1][ks]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    X3_STAMP(10);

    // ------------------------------------------------------------------ epilogue: + residual, transposed in place
    // the result quad of (token, 4 channels) overwrites the x quad it just consumed; afterwards every lane stores
    // 16 B of a whole output row from the wave's own bytes of the x tile (no other wave touches them)
    {
        half4 xr[5][4];                     // all residual quads first: one LDS round trip, not twenty
        if (add_res) {
#pragma unroll
            for (int ct = 0; ct < 5; ++ct)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) xr[ct][tt] = *t_quad(0, tt, ct);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ct = 0; ct < 5; ++ct)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                // attn(x) is rounded to fp16 and the residual added in fp16 -- the reference's own arithmetic
                // (hidden_states = attn2(...) + hidden_states on fp16 tensors), and two packed adds instead of eight
                // conversions and four adds per quad
                half4 r = cvt4(acc[ct][tt]);
                if (add_res) r = r + xr[ct][tt];
                *t_quad(0, tt, ct) = r;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        X3_STAMP(11);
        half_t* ob = out + tok0 * YC + wn * 80;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int e = k * 64 + lane;
            const int r = e / 10, c = e - r * 10;
            const int ch = wn * 80 + c * 8;
            *reinterpret_cast<half8*>(ob + (long)r * YC + c * 8) = *t_frag(0, ch >> 6, r, (ch >> 3) & 7);
        }
        X3_STAMP(12);
#ifdef CID_X3_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        X3_STAMP(13);
#ifdef CID_X3_TRACE
        if (lane == 0) g_x3_trace[((long)blockIdx.x * 4 + wn) * 32 + 29] = wall_clock64();
#endif
    }
#endif
}

// dst[r][e] = idx[e] < 0 ? 0 : (bit 30 of idx[e] ? src_b : src_a)[r * src_row + (idx[e] & 0x3fffffff)]
__global__ void __launch_bounds__(256)
gather_pack_kernel(const half_t* __restrict__ src_a, const half_t* __restrict__ src_b, const int* __restrict__ idx,
                   half_t* __restrict__ dst, int R, long src_row, long n_idx) {
    const long total = (long)R * n_idx;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long r = q / n_idx, e = q - r * n_idx;
        const int i = idx[e];
        half_t v = (half_t)0.f;
        if (i >= 0) v = ((i & 0x40000000) ? src_b : src_a)[r * src_row + (i & 0x3fffffff)];
        dst[q] = v;
    }
}

}  // namespace

#ifdef CID_X3_TRACE
extern "C" int cid_debug_x3_trace(unsigned long long* host, int64_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_x3_trace), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int cid_id_xattn3_supported(int32_t C, int32_t heads, int32_t n_txt, int32_t n_ip) {
    // geometry: SD1.5 level 0; context layouts: the reference's 77 + 4 (UNet) and 81 plain keys (ControlNet)
    return (C == YC && heads == YNH && ((n_txt == 77 && n_ip == 4) || (n_txt == 81 && n_ip == 0))) ? 1 : 0;
}

extern "C" int64_t cid_kv_pack2_elems(int32_t C, int32_t heads, int32_t which) {
    if (C != YC || heads != YNH) return -22;
    return which == 0 ? Y_KROW : Y_VROW;
}

extern "C" int cid_gather_pack_f16(const cid_half* src_a, const cid_half* src_b, const int32_t* idx, cid_half* dst,
                                   int32_t R, int64_t src_row_elems, int64_t n_idx, cid_stream_t stream) {
    CID_CHECK_ARG(src_a && src_b && idx && dst, "cid_gather_pack_f16: null pointer");
    CID_CHECK_ARG(R > 0 && src_row_elems > 0 && src_row_elems < (1 << 30) && n_idx > 0, "cid_gather_pack_f16: bad sizes");
    const long total = (long)R * n_idx;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(gather_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const half_t*)src_a,
                       (const half_t*)src_b, (const int*)idx, (half_t*)dst, R, (long)src_row_elems, (long)n_idx);
    CID_CHECK_LAUNCH("cid_gather_pack_f16");
    return 0;
}

extern "C" int cid_id_xattn3_f16(const cid_half* x, cid_half* out, const cid_half* wq_packed, const float* q_rowsum,
                                 const float* q_bias, const cid_half* wo_packed, const cid_half* bo, const cid_half* kp,
                                 const cid_half* vp, const int32_t* kvrow, int32_t B, int32_t N, int32_t C, int32_t heads,
                                 int32_t n_txt, int32_t n_ip, float ip_scale, float ln_eps, int32_t flags,
                                 cid_stream_t stream) {
    CID_CHECK_ARG(x && out && wq_packed && q_rowsum && q_bias && wo_packed && kp && vp && kvrow, "cid_id_xattn3_f16: null pointer");
    CID_CHECK_ARG(cid_id_xattn3_supported(C, heads, n_txt, n_ip),
                  "cid_id_xattn3_f16: built for C=%d, %d heads and a 77+4 or 81+0 context (got C=%d heads=%d context %d+%d)",
                  YC, YNH, C, heads, n_txt, n_ip);
    CID_CHECK_ARG(B > 0 && N > 0 && N % YBT == 0, "cid_id_xattn3_f16: N=%d must be a positive multiple of %d", N, YBT);
    CID_CHECK_ARG(x != out, "cid_id_xattn3_f16: in-place operation is not supported");
    const int kind = (n_txt == 77 && n_ip == 4) ? 0 : 1;
    auto kern = kind == 0 ? id_xattn3_kernel<77, 4> : id_xattn3_kernel<81, 0>;
    static bool configured[2] = {false, false};
    if (!configured[kind]) {
        hipError_t herr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Y_SMEM);
        if (herr != hipSuccess) {
            cid_set_error("cid_id_xattn3_f16: cannot reserve %d bytes of LDS (%s)", Y_SMEM, hipGetErrorString(herr));
            return -5;
        }
        configured[kind] = true;
    }
    const int tiles = N / YBT, total = tiles * B;
    hipLaunchKernelGGL(kern, dim3(total), dim3(256), Y_SMEM, (hipStream_t)stream, (const half_t*)x, (half_t*)out,
                       (const half_t*)wq_packed, q_rowsum, q_bias, (const half_t*)wo_packed, (const half_t*)bo,
                       (const half_t*)kp, (const half_t*)vp, kvrow, N, tiles, total, ip_scale, ln_eps, flags);
    CID_CHECK_LAUNCH("cid_id_xattn3_f16");
    return 0;
}
