// Flash-style self-attention core for Consistent_AttProcessor (attention.py:149-159):
// O = softmax(Q K^T) V per (sample, head), never materialising the N x N scores.
//
// Roofline: MFMA-bound; algorithmic work 4 * N^2 * d flop per (sample, head).
//
// Formulation (everything "transposed" so the softmax axis is lane-local):
//   S^T[key][q] = K[key][:] . Q[q][:]         MFMA A = K rows (LDS), B = Q rows (registers)
//   lane (q = lane & 31, hi) holds 16 keys of its query per 32-key tile: row max / sum
//   are in-lane loops plus ONE cross-half exchange (lane ^ 32).
//   O^T[d][q] += Vt[d][key] . P^T[key][q]     MFMA A = V^T rows (LDS), B = P^T taken
//   straight from the S^T accumulator registers: the C-layout of S^T puts keys
//   {4hi+0..3, 4hi+8..11} of every 16-key group in a lane, and the V^T image is stored
//   with exactly that permutation of the key axis (written by the QKV GEMM epilogue),
//   so no cross-lane shuffle or LDS round trip is needed between the two MFMAs.
// Q arrives pre-multiplied by d^-0.5 * log2(e): softmax uses raw v_exp_f32 (2^x).
// K / V^T tiles (64 keys) are double buffered in LDS and arrive by DMA (buffer_load ... lds, no VGPR round trip) as
// chunk-major images -- 16-B chunk c of every row of the tile is contiguous, so a ds_read_b128 of 16 consecutive rows
// at one chunk is bank-conflict free and the pad chunk / ones rows are written once; the next tile's DMA is in flight
// during the current tile's MFMAs.  Workgroups are numbered so that one XCD owns whole (sample, head) pairs: their
// K / V stay in ONE L2.
// Where the time goes at d = 40: per 64-key step a wave issues 14 32x32x16 MFMAs (455 cycles) and ~75 VALU
// instructions of which 32 v_exp_f32 (8.5 cycles each; 16 v_cvt_pk and 16 v_max3 at 4.4) -- the two do not overlap on a SIMD,
// across waves or inside one: 455 + ~400 = the measured ~850 cycles per step (MFMA-only build 120 us, VALU-only build
// 169 us, shipped 283 us in the same call: profiles/r04_attn_ablation.txt).
#include "common.h"
#include "../../include/cid.h"
#include <stdlib.h>
#include <type_traits>

// Profiling knobs, experiment builds only (python -m consistentid_amd.build --variant attnabl CID_ATTN_ABL=<bits>):
// 1 no exp2 (a multiply instead), 2 no P.V MFMAs, 4 no Q.K^T MFMAs, 8 no K / V staging in the loop, 16 no row-max pass
#ifndef CID_ATTN_ABL
#define CID_ATTN_ABL 0
#endif
// comparator build (--variant attnhead CID_ATTN_MAX_AT_HEAD): the row max of a score tile is taken at the head of the step
// that consumes it (round 3's form) instead of beside the previous step's P.V MFMAs
#ifdef CID_ATTN_MAX_AT_HEAD
#define CID_ATTN_HEADMAX true
#else
#define CID_ATTN_HEADMAX false
#endif

namespace {

// max of a 16-float accumulator tile and a carry-in: plain fmaxf chains (v_max3_f32 -- the file is built with
// -fno-honor-nans, so no canonicalising v_max x, x in front of every operand; scores are finite or -inf).  Plain code, not
// an asm block: the compiler knows the MFMA -> VALU wait states and can spread the chain between the P.V MFMAs.
__device__ __forceinline__ float max17f(float m, const f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, v[r]), v[r + 1]);
    return m;
}

template <int D, int QT, int NWV>
struct AttnCfg {
    static constexpr int NT = 64 * NWV;
    static constexpr int BQ = 32 * QT * NWV;
    static constexpr int DKP = (D + 15) / 16 * 16;    // contraction dim padded to the MFMA k step
    static constexpr int KSTEPS = DKP / 16;
    static constexpr int DVT = (D + 31) / 32;          // 32-row output tiles of O^T
    // LDS images, chunk-major so that one LDS-DMA piece (64 lanes x 16 B, written lane-linearly) is one 16-byte chunk
    // column: K tile = [DKP / 8 chunks][64 keys][16 B] (chunk c of key r at c * 1024 + r * 16; the pad chunk behind the
    // D / 8 data chunks is written once and never touched by the DMA), V^T tile = [8 key chunks][D rows][16 B] followed by
    // the constant rows D .. 32 DVT - 1 ([8][CR][16 B]: the ones row, then don't-care rows).  A 32-row fragment read is 32
    // consecutive 16-byte slots: conflict free without padding.
    static constexpr int NCH = D / 8;                  // data chunks per key = DMA pieces per K tile = pieces per V^T tile
    static constexpr int CR = DVT * 32 - D;            // constant rows of the V^T image
    static constexpr int KBYTES = (DKP / 8) * 1024;
    static constexpr int VDATA = D * 128;
    static constexpr int VBYTES = VDATA + CR * 128;
    static constexpr int BUF = KBYTES + VBYTES;
    static constexpr int PPW = (NCH + NWV - 1) / NWV;  // pieces per wave and tile (K and V^T alike)
};

template <int D, int QT, int NWV, bool MASK>
__global__ void __launch_bounds__(64 * NWV, (QT == 1 && D <= 40 && NWV == 4) ? 3 : ((QT == 1 && D <= 80) ? 2 : 1))
self_attn_kernel(const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ vt,
                 half_t* __restrict__ out, int N, int heads, int ldq, int ldk, int dvp, int ldo, int n_keys, int xcd_remap) {
#if defined(__HIP_DEVICE_COMPILE__)   // device-only builtins below; the host pass only needs the stub
    using Cfg = AttnCfg<D, QT, NWV>;
    constexpr int NT = Cfg::NT, KSTEPS = Cfg::KSTEPS, DVT = Cfg::DVT;
    constexpr int NCH = Cfg::NCH, CR = Cfg::CR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int idx = lane & 31, hi = lane >> 5;
    // XCD-aware order: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs (private L2 each).  The
    // query tiles of ONE (sample, head) all stream the same K / V^T (N x d x 2 x 2 bytes: 655 KB at 4096 tokens, d = 40);
    // dealt round-robin every XCD sees every (sample, head) in flight and its 4 MB L2 thrashes.  Remapped, an XCD owns a
    // contiguous run of (sample, head) pairs: ~3 in flight per XCD, K / V^T stay L2-resident for all their query tiles.
    int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    {
        const int nwg = gridDim.x * gridDim.y * gridDim.z;
        if (xcd_remap && (nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    }
    const int bx = bid % (int)gridDim.x;
    const int h = (bid / (int)gridDim.x) % (int)gridDim.y, b = bid / (int)(gridDim.x * gridDim.y);
    const int q0 = bx * Cfg::BQ + wave * (32 * QT);

    const half_t* kbase = k + (long)b * N * ldk + h * D;
    const half_t* vbase = vt + ((long)(b * heads + h) * dvp) * N;

    // The K pad chunk [D, DKP) of both buffers is written once (the DMA never touches it): zeros, except
    // column D = 1.  With Q's pad slot D holding -m (the running row max, kept fp16-representable) the QK^T MFMA
    // itself delivers s - m, which removes the 32 v_sub per tile from the VALU-bound softmax (d = 40: the k axis
    // is padded 40 -> 48 anyway, so the bias slot is free).
#if defined(CID_ATTN_NOBIAS)
    constexpr bool BIAS = false;
#else
    constexpr bool BIAS = Cfg::DKP > D;
#endif
    static_assert(!BIAS || Cfg::DKP - D == 8, "pad is one 16-byte slot");
    if (Cfg::DKP > D) {
        half8 pad = zero_h8();
        if (BIAS) pad[0] = (half_t)1.f;
        for (int r = tid; r < 2 * 64; r += NT)
            *reinterpret_cast<half8*>(smem + (r >> 6) * Cfg::BUF + NCH * 1024 + (r & 63) * 16) = pad;
    }
    const bool bias_lane = BIAS && ((KSTEPS - 1) * 16 + hi * 8 == D);   // lanes whose last Q fragment starts at column D

    // Q fragments (B operand), zero beyond the head dim
    half8 qf[QT][KSTEPS];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const half_t* qrow = q + ((long)b * N + q0 + t * 32 + idx) * ldq + h * D;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const int c = kk * 16 + hi * 8;
            qf[t][kk] = (c < D) ? ld_global_h8(qrow + c) : zero_h8();
        }
    }

    // ones row: when the head dim leaves spare rows in the last 32-row tile of V^T (d = 40, 80), row D of
    // the LDS image is all ones, so O^T row D accumulates sum_k P[k][q] -- the softmax denominator comes
    // out of the matrix pipe (with the same rescaling as O) instead of 32 VALU adds per tile.  Constant row 0 of the
    // V^T image (written once, never touched by the DMA); the other constant rows feed output rows nobody reads.
    constexpr bool ONES = (D % 32) != 0;
    constexpr int ONES_REG = ((D % 32) & 3) + 4 * ((D % 32) >> 3);   // accumulator slot of row D (lanes hi = ((D%32)>>2)&1)
    constexpr int ONES_HI = ((D % 32) >> 2) & 1;
    if (ONES) {
        half8 one;
#pragma unroll
        for (int i = 0; i < 8; ++i) one[i] = (half_t)1.f;
        for (int e = tid; e < 2 * 8 * CR; e += NT) {           // every constant row gets ones: finite values everywhere
            const int bf = e / (8 * CR), rem = e - bf * (8 * CR);
            *reinterpret_cast<half8*>(smem + bf * Cfg::BUF + Cfg::KBYTES + Cfg::VDATA + rem * 16) = one;
        }
    }

    // ---- staging by LDS-DMA (buffer_load ... lds, no VGPR round trip, no ds_write): piece p of a K tile = chunk p of
    //      the 64 keys (lane = key), piece p of a V^T tile = slots 64 p .. 64 p + 63 of the [8 key chunks][D rows] image
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (unsigned)(((long)(N - 1) * ldk + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (unsigned)((long)D * N * 2), 0x00020000);
    const unsigned kvo = (unsigned)(lane * ldk * 2);
    unsigned vvo[Cfg::PPW];
#pragma unroll
    for (int j = 0; j < Cfg::PPW; ++j) {
        const int sl = (j * NWV + wave) * 64 + lane;              // slot = chunk * D + row
        const int c = sl / D, r = sl - c * D;
        vvo[j] = (unsigned)(((long)r * N + c * 8) * 2);
    }
    auto dma_k = [&](int key0, int buf) {
#pragma unroll
        for (int j = 0; j < Cfg::PPW; ++j) {
            const int p = j * NWV + wave;
            if (p < NCH)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (lds_void*)(smem + buf * Cfg::BUF + p * 1024), 16, kvo,
                                                         key0 * ldk * 2 + p * 16, 0, 0);
        }
    };
    auto dma_v = [&](int key0, int buf) {
#pragma unroll
        for (int j = 0; j < Cfg::PPW; ++j) {
            const int p = j * NWV + wave;
            if (p < NCH)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_void*)(smem + buf * Cfg::BUF + Cfg::KBYTES + p * 1024), 16,
                                                         vvo[j], key0 * 2, 0, 0);
        }
    };
    // S^T = K Q^T for the 64 keys of K buffer `buf`
    auto qk = [&](int buf, f32x16 (&s)[2][QT]) {
        const char* kb = smem + buf * Cfg::BUF;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int t = 0; t < QT; ++t) s[kt][t] = zero_f16v();
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            half8 kf[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
                kf[kt] = *reinterpret_cast<const half8*>(kb + (kk * 2 + hi) * 1024 + (kt * 32 + idx) * 16);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int t = 0; t < QT; ++t) { if (!(CID_ATTN_ABL & 4)) s[kt][t] = mfma32(kf[kt], qf[t][kk], s[kt][t]); else s[kt][t][kk] += (float)kf[kt][0]; }
        }
    };
    // byte offset of this lane's V^T fragment rows: data rows (chunk stride D slots) / constant rows (chunk stride CR slots)
    const int vlane = (hi * D + idx) * 16;
    const int vlane_c = Cfg::VDATA + (hi * CR + idx + (DVT - 1) * 32 - D) * 16;
    const bool vconst_lane = (DVT - 1) * 32 + idx >= D;

    f32x16 oacc[DVT][QT];
#pragma unroll
    for (int d = 0; d < DVT; ++d)
#pragma unroll
        for (int t = 0; t < QT; ++t) oacc[d][t] = zero_f16v();
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) { m_run[t] = -INFINITY; l_run[t] = 0.f; }

    // ---- software pipeline: K tiles run one tile ahead of V tiles, so that the QK^T MFMAs of tile t+1
    //      are independent of (and interleave with) the softmax VALU work of tile t
    const int ntiles = N / 64;
    dma_k(0, 0); dma_v(0, 0);
    if (ntiles > 1) dma_k(64, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 s_cur[2][QT], s_nxt[2][QT];
    qk(0, s_cur);
    // row max of a 64-key tile: in-lane max3 chains + one cross-half exchange
    auto tile_max = [&](f32x16 (&sc)[2][QT], int t) {
        float mx = max17f(sc[0][t][0], sc[0][t]);
        mx = max17f(mx, sc[1][t]);
        return fmaxf(mx, __shfl_xor(mx, 32, 64));
    };
    // install a new (fp16-representable) row max: into Q's bias slot, and into the scores already computed
    auto rebias = [&](f32x16 (&sc)[2][QT], int t, float m_new, float delta) {
        if (bias_lane) qf[t][KSTEPS - 1][0] = (half_t)(-m_new);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt][t][r] -= delta;
    };
    // mx_a / mx_b: row max of the score tile a step is about to consume, handed from step to step (taken beside the
    // previous step's P.V MFMAs); tile 0's comes from here
    float mx_a[QT], mx_b[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float mx = tile_max(s_cur, t);
        if (BIAS) {
            const float m0 = (float)(half_t)fminf(fmaxf(mx, -60000.f), 60000.f);
            rebias(s_cur, t, m0, m0);
            m_run[t] = m0;
            mx_a[t] = mx - m0;          // relative to m_run: the fp16 rounding of m0, far below the 2^8 threshold
        } else {
            mx_a[t] = mx;
        }
        mx_b[t] = 0.f;
    }
    __syncthreads();      // every wave has read K tile 0: step 0 refills its buffer

    // one pipeline step; HAS_NEXT is a compile-time flag so that the next tile's QK^T MFMAs sit in the
    // same basic block as this tile's exp / convert work and the scheduler can interleave the two pipes.
    // mx_cur: row max of s_cur (BIAS: relative to m_run), taken ONE STEP EARLIER behind that step's P.V MFMAs; mx_nxt: the
    // same for s_nxt.  (Measured: the launch does not care where the chain sits -- 283 vs 282 us at level 0 -- because
    // non-FMA VALU work and MFMAs do not overlap on a gfx950 SIMD; kept because the chain is plain code now,
    // with the compiler's own MFMA -> VALU wait states instead of hand-counted s_nops.)
    auto step = [&](int tile, auto has_next_tag, f32x16 (&s_cur)[2][QT], f32x16 (&s_nxt)[2][QT], float (&mx_cur)[QT],
                    float (&mx_nxt)[QT]) {
        constexpr bool HAS_NEXT = decltype(has_next_tag)::value;
        const int cur = tile & 1;
        // staging: K(tile+2) -> K buffer `cur` (its tile was multiplied one step ago), V(tile+1) -> V buffer `cur^1` (read
        // by the previous step); both have the whole step to land, the wait sits in front of the closing barrier
        if (!(CID_ATTN_ABL & 8)) {
            if (tile + 2 < ntiles) dma_k((tile + 2) * 64, cur);
            if (HAS_NEXT) dma_v((tile + 1) * 64, cur ^ 1);
        }
        // keys beyond n_keys are padding (token counts that are not a multiple of 64, e.g. CLIP's 257): score -inf
        if (MASK && (tile + 1) * 64 > n_keys) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int t = 0; t < QT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (tile * 64 + kt * 32 + crow(r, hi) >= n_keys) s_cur[kt][t][r] = -INFINITY;
        }
        // ---- running max of tile `tile` (per query column).
        //      It is only raised when a score exceeds it by more than 2^8 ("defer max"): p <= 256 keeps
        //      full fp16 relative precision and the O / l rescale is skipped almost always.
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            // (masked launches take the max here, behind the mask; the others get it from the previous step)
            const float mx = (CID_ATTN_ABL & 16) ? s_cur[0][t][0] : ((MASK || CID_ATTN_HEADMAX) ? tile_max(s_cur, t) : mx_cur[t]);   // BIAS: relative to m_run
            if (__any(mx > (BIAS ? 8.f : m_run[t] + 8.f))) {
                float m_new, alpha;
                if (BIAS) {
                    m_new = (float)(half_t)fminf(m_run[t] + fmaxf(mx, 0.f), 60000.f);
                    const float delta = m_new - m_run[t];
                    alpha = __builtin_amdgcn_exp2f(-delta);
                    rebias(s_cur, t, m_new, delta);   // before qk(next): S^T of the next tile is born with the new max
                } else {
                    m_new = fmaxf(m_run[t], mx);
                    alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);
                }
                m_run[t] = m_new;
                l_run[t] *= alpha;
#pragma unroll
                for (int d = 0; d < DVT; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[d][t][r] *= alpha;
            }
        }
        // ---- S^T of the next tile (MFMA) || exp2 / convert of this tile (VALU)
        if (HAS_NEXT) qk(cur ^ 1, s_nxt);
        half8 pf[QT][4];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const float m = m_run[t];
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    half8 pv;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float a_ = BIAS ? s_cur[kt][t][g * 8 + i] : s_cur[kt][t][g * 8 + i] - m;
                        const float p = (CID_ATTN_ABL & 1) ? a_ * 0.001f : __builtin_amdgcn_exp2f(a_);
                        if (!ONES) rs += p;
                        pv[i] = (half_t)p;
                    }
                    pf[t][kt * 2 + g] = pv;
                }
            if (!ONES) l_run[t] += rs;
        }
        // ---- O^T += V^T P^T : 4 k-steps of 16 keys; the row max of the NEXT tile's scores rides between these MFMAs
        const char* vb = smem + cur * Cfg::BUF + Cfg::KBYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            half8 vf[DVT];
#pragma unroll
            for (int d = 0; d < DVT; ++d) {
                int off = vlane + (ks * 2 * D + d * 32) * 16;
                if (CR > 0 && d == DVT - 1) off = vconst_lane ? vlane_c + ks * 2 * CR * 16 : off;
                vf[d] = *reinterpret_cast<const half8*>(vb + off);
            }
#pragma unroll
            for (int d = 0; d < DVT; ++d)
#pragma unroll
                for (int t = 0; t < QT; ++t) { if (!(CID_ATTN_ABL & 2)) oacc[d][t] = mfma32(vf[d], pf[t][ks], oacc[d][t]); else oacc[d][t][ks] += (float)vf[d][0] * (float)pf[t][ks][0]; }
        }
        if (HAS_NEXT && !MASK && !CID_ATTN_HEADMAX) {
#pragma unroll
            for (int t = 0; t < QT; ++t) mx_nxt[t] = tile_max(s_nxt, t);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces have landed ...
        __syncthreads();                                      // ... and so have everybody's
    };
    // two steps per trip with the score tiles ping-ponging between two register sets (no tile copies)
    int tile = 0;
    for (; tile + 2 < ntiles; tile += 2) {
        step(tile, std::true_type{}, s_cur, s_nxt, mx_a, mx_b);
        step(tile + 1, std::true_type{}, s_nxt, s_cur, mx_b, mx_a);
    }
    if (ntiles - tile == 2) {
        step(tile, std::true_type{}, s_cur, s_nxt, mx_a, mx_b);
        step(tile + 1, std::false_type{}, s_nxt, s_cur, mx_b, mx_a);
    } else {
        step(tile, std::false_type{}, s_cur, s_nxt, mx_a, mx_b);
    }

    // ---- epilogue: O = O^T / l, lane owns query q, 4 consecutive head-dims per quad
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float l;
        if (ONES) {
            const float mine = oacc[DVT - 1][t][ONES_REG];          // valid in lanes with hi == ONES_HI
            const float other = __shfl_xor(mine, 32, 64);
            l = (hi == ONES_HI) ? mine : other;
        } else {
            l = l_run[t] + __shfl_xor(l_run[t], 32, 64);
        }
        const float inv = 1.f / l;
        half_t* orow = out + ((long)b * N + q0 + t * 32 + idx) * ldo + h * D;
#pragma unroll
        for (int d = 0; d < DVT; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int dd = d * 32 + 8 * j + 4 * hi;
                if (dd < D) {
                    half4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (half_t)(oacc[d][t][j * 4 + i] * inv);
                    *reinterpret_cast<half4*>(orow + dd) = o;
                }
            }
    }
#endif
}

template <int D, int QT, int NWV, bool MASK = false>
int launch_attn(const half_t* q, const half_t* k, const half_t* vt, half_t* out, int B, int N, int heads,
                int ldq, int ldk, int dvp, int ldo, int n_keys, hipStream_t s) {
    using Cfg = AttnCfg<D, QT, NWV>;
    constexpr int smem = 2 * Cfg::BUF;
    static bool configured = false;
    auto kern = self_attn_kernel<D, QT, NWV, MASK>;
    if (!configured) {
        hipError_t herr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (herr != hipSuccess) {
            cid_set_error("cid_self_attn_f16: cannot reserve %d bytes of LDS (%s)", smem, hipGetErrorString(herr));
            return -5;
        }
        configured = true;
    }
    dim3 grid(N / Cfg::BQ, heads, B);
    static int remap = -1;
    if (remap < 0) { const char* e = getenv("CID_ATTN_XCD"); remap = e ? atoi(e) : 1; }      // A/B switch
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), smem, s, q, k, vt, out, N, heads, ldq, ldk, dvp, ldo, n_keys, remap);
    return 0;
}

}  // namespace

extern "C" int cid_self_attn_f16(const cid_half* q, const cid_half* k, const cid_half* vt, cid_half* out,
                                 int32_t B, int32_t N, int32_t heads, int32_t d,
                                 int32_t ldq, int32_t ldk, int32_t dvp, int32_t ldo, cid_stream_t stream) {
    return cid_self_attn_keys_f16(q, k, vt, out, B, N, heads, d, ldq, ldk, dvp, ldo, N, stream);
}

extern "C" int cid_self_attn_keys_f16(const cid_half* q, const cid_half* k, const cid_half* vt, cid_half* out,
                                      int32_t B, int32_t N, int32_t heads, int32_t d, int32_t ldq, int32_t ldk,
                                      int32_t dvp, int32_t ldo, int32_t n_keys, cid_stream_t stream) {
    CID_CHECK_ARG(q && k && vt && out, "cid_self_attn_f16: null pointer");
    CID_CHECK_ARG(n_keys > 0 && n_keys <= N, "cid_self_attn_keys_f16: n_keys must be in (0, N] (got %d, N = %d)", n_keys, N);
    CID_CHECK_ARG(B > 0 && heads > 0 && N > 0 && N % 64 == 0, "cid_self_attn_f16: N must be a positive multiple of 64 (got %d)", N);
    CID_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 4 == 0 && dvp >= d, "cid_self_attn_f16: bad pitches");
    const half_t* Q = (const half_t*)q; const half_t* K = (const half_t*)k; const half_t* V = (const half_t*)vt;
    half_t* O = (half_t*)out;
    hipStream_t s = (hipStream_t)stream;
    int rc = -22;
#define CID_ATTN(DD, QT, NWV) rc = launch_attn<DD, QT, NWV>(Q, K, V, O, B, N, heads, ldq, ldk, dvp, ldo, n_keys, s)
#define CID_ATTN_MASKED(DD) \
    rc = (N % 128 == 0) ? launch_attn<DD, 1, 4, true>(Q, K, V, O, B, N, heads, ldq, ldk, dvp, ldo, n_keys, s) \
                        : launch_attn<DD, 1, 2, true>(Q, K, V, O, B, N, heads, ldq, ldk, dvp, ldo, n_keys, s)
    if (n_keys != N) {
        // padded key axis (CLIP towers with 257 tokens; UNet levels whose token count is not a multiple of 64 at
        // resolutions other than 512^2 / 1024^2): one 32-query tile per wave
        if (d == 80) CID_ATTN_MASKED(80);
        else if (d == 64) CID_ATTN_MASKED(64);
        else if (d == 40) CID_ATTN_MASKED(40);
        else if (d == 160) CID_ATTN_MASKED(160);
        else if (d == 32) CID_ATTN_MASKED(32);
        else { cid_set_error("cid_self_attn_keys_f16: unsupported head dim %d (40, 64, 80, 160, 32)", d); return -22; }
        if (rc) return rc;
        CID_CHECK_LAUNCH("cid_self_attn_keys_f16");
        return 0;
    }
    if (d == 40) {
        // one 32-query tile per wave: ~3 waves per SIMD, so one wave's softmax (VALU) overlaps another's MFMAs
        if (N % 128 == 0) CID_ATTN(40, 1, 4); else CID_ATTN(40, 1, 2);
    } else if (d == 64 || d == 80) {
        // one 32-query tile per wave by default: two tiles per wave (K/V fragments reused, 256 registers, one wave per
        // SIMD) measured slower end to end at every level that occurs (SDXL 1024^2: 1.43 vs 1.51 images/s; SD1.5 equal)
        static int force_qt = -1;
        if (force_qt < 0) { const char* e = getenv("CID_ATTN_QT"); force_qt = e ? atoi(e) : 0; }
        const bool qt2 = (N % 256 == 0) && force_qt == 2;
        if (d == 64) {
            if (qt2) CID_ATTN(64, 2, 4); else if (N % 128 == 0) CID_ATTN(64, 1, 4); else CID_ATTN(64, 1, 2);
        } else {
            if (qt2) CID_ATTN(80, 2, 4); else if (N % 128 == 0) CID_ATTN(80, 1, 4); else CID_ATTN(80, 1, 2);
        }
    } else if (d == 160) {
        if (N % 128 == 0) CID_ATTN(160, 1, 4); else CID_ATTN(160, 1, 2);
    } else if (d == 32) {
        if (N % 128 == 0) CID_ATTN(32, 1, 4); else CID_ATTN(32, 1, 2);
    } else {
        cid_set_error("cid_self_attn_f16: unsupported head dim %d (40, 64, 80, 160, 32)", d);
        return -22;
    }
#undef CID_ATTN
#undef CID_ATTN_MASKED
    if (rc) return rc;
    CID_CHECK_LAUNCH("cid_self_attn_f16");
    return 0;
}
