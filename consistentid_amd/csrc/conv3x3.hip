// 3x3 stride-1 convolution of ResnetBlock2D (diffusers ResnetBlock2D.conv1 / conv2 as called from
// /root/reference/pipline_StableDiffusion_ConsistentID.py:552-557) on v_mfma_f32_32x32x16_f16, with the two roles of a GEMM
// pipeline given to DIFFERENT waves:
//   * four COMPUTE waves (one per SIMD) own a 64-token x 160-channel accumulator tile each (2 x 5 tiles of 32 x 32: 160
//     registers) and execute nothing but ds_read_b128 + MFMA, the reads interleaved one to one with the MFMAs of the previous
//     16-deep k-step (7 reads per 10 MFMAs, sched_group_barrier): 1 330 cycles per 64-channel slab against 1 280 of MFMA issue
//     (tools/probes/conv_loop.hip, the prototype of this kernel);
//   * four LOADER waves (waves 4-7: the SIMD partners of waves 0-3) execute nothing but DMA (buffer_load ... lds): the halo
//     tile of the NEXT 64-channel slab ((rows + 2) x (W + 2) tokens, two buffers) two 1-KiB pieces per weight slab, the weight
//     slab two slabs ahead (three LDS stages, counted vmcnt), and -- in the last channel slab -- the residual tile, into their
//     otherwise idle registers, as whole 16-byte row chunks;
//   * ONE barrier per 64-deep slab, placed between the third and the fourth k-step of the compute waves: the fragment reads
//     of the next slab's first k-step are issued right behind it and land under the fourth k-step's MFMAs.
// Tile = 256 tokens (whole image rows of one image, or whole images) x 160 channels; LDS = 2 halo buffers + 3 weight stages
// (<= 160 KB).  Same operand layout as igemm_halo_kernel (gemm.hip): token-major activations (optionally two sources = skip
// concat), W[n][tap][c].  Bank swizzle: 16-B chunk c of LDS row r sits at chunk c ^ ((r >> 1) & 7) (XOR on the DMA's source
// address): a ds_read_b128 of 32 consecutive rows at one k-chunk is conflict free for every start row.
//
// Epilogue: the loaders park the residual rows and the per-sample bias + time-embedding vector in LDS, the compute waves add
// them to the accumulators in fp32, round once and transpose the fp16 tile through LDS, ALL eight waves write it out as
// 16-byte row chunks; the GroupNorm statistics of the written tensor (for the GroupNorm that consumes it) are column sums of
// the staged fp16 tile.
//
// No split-K in this kernel.  An in-launch reduction (fp32 slabs, one ticket, the last slice adds the others) was built and
// measured: the slab round trip (160 KB per workgroup, 41 MB per launch) costs what the separate reduce launch costs, written
// through (sc1) or not -- 32 x 32 level, 640 -> 640: 70-77 us against 66 us for the halo kernel of gemm.hip + its reduce
// launch.  Levels with fewer than 256 tiles of 256 tokens run 128-token tiles here while K is short (LDS-bound: six fragment
// reads per five MFMAs), else stay on gemm.hip's halo kernel + splitk_epilogue_kernel (plan_gemm).
//
// Roofline: MFMA-bound in cycles (96 % of the issue rate); in time the chip is POWER-bound under it -- all 256 CUs in this
// loop sustain 1.35-1.4 GHz (cycle counter against wall time, tools/probes/conv_loop.hip), i.e. 1.4 PFLOP/s of the 2.5 PFLOP/s
// the 2.4 GHz peak clock would give.  Algorithmic work 2 * M * N * 9 C flop; bytes (M C + 9 C N + M N) * 2.
#include "gemm_args.h"
#include "../../include/cid.h"

namespace {
using cidg::GemmArgs;
using cidg::wait_vmcnt;

constexpr int BN = 160, BK = 64;          // (token tile BM = 128 TM: 256 tokens, or 128 where 256-token tiles would not fill the chip)
constexpr int WST = BN * 128;            // bytes of one weight stage: 160 rows x 64 halfs
constexpr int NSTG = 3;
constexpr int HQ = 13;                   // halo piece slots per loader wave (4 x 13 = 52 >= 50 pieces of 8 rows)
constexpr int WQ = 5;                    // weight pieces per loader wave and slab (20 pieces)
constexpr int TP = BN + 8;               // staging tile row pitch (halfs): 336 B, rows 16-byte aligned
// epilogue images in LDS: staging tile [BM][TP] fp16 | combined bias [4 samples][160] fp32 | GroupNorm partial sums
// [BM / 32 token segments][160 channels][2] fp32 | channel sums [160][2] fp64
constexpr int t_bytes(int bm) { return bm * TP * 2; }
constexpr int cb_off(int bm) { return t_bytes(bm); }
constexpr int gs_off(int bm) { return cb_off(bm) + 4 * BN * 4; }
constexpr int epi_bytes(int bm) { return gs_off(bm) + (bm / 32) * BN * 2 * 4 + BN * 2 * 8; }

typedef __attribute__((address_space(3))) void lds_void;

CID_DEVINL int key(int r) { return (r >> 1) & 7; }

template <int TM>
__global__ void __launch_bounds__(512, 1)
conv_h32_kernel(GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128 * TM;
    constexpr int RQ = BM * 20 / 256;            // residual row chunks (16 B) per loader lane
    constexpr int NSEG = BM / 32;                // 32-token segments of the tile (GroupNorm statistics)
    constexpr int CB_OFF = cb_off(BM), GS_OFF = gs_off(BM);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r8 = lane >> 3, c8 = lane & 7;
    const int l32 = lane & 31, lh = lane >> 5;
    int nb_, mb_;
    cidg::xcd_tile(a, nb_, mb_);                  // XCD-aware tile order (gemm_args.h)
    const int n0 = nb_ * BN;
    const int m0 = mb_ * BM;
    const int W = a.Wo, H = a.Ho, HW = H * W;
    const int seg_tok = BM < HW ? BM : HW;
    const int rs = seg_tok / W;
    // up = 1 (Upsample2D: nearest-2x, then the convolution): the halo holds INPUT pixels -- output pixel (y, x), tap (ty, tx)
    // reads input pixel ((y + ty - 1) >> 1, (x + tx - 1) >> 1) -- rs / 2 + 2 input rows of Wi + 2 columns per tile (a tile is
    // whole output rows of one image there: plan_gemm)
    const int up = a.up;
    const int Wi = W >> up, Hi = H >> up;
    const int hs = ((rs >> up) + 2) * (Wi + 2);
    const int nh = (BM / seg_tok) * hs;
    const int HP = (nh + 7) >> 3;                 // halo pieces (8 rows, 1 KiB)
    const int HS = HP * 1024;                     // bytes of one halo buffer
    const int img0 = m0 / HW;
    const int y0 = (m0 - img0 * HW) / W;
    char* hbuf = smem;
    char* wbuf = smem + 2 * HS;
    const int ctot = a.c1 + a.c2;
    const int ncs = a.cslabs;
    const int cs_begin = 0, cs_end = ncs;         // (no split-K here: see the header)

    const bool is_loader = wave >= 4;
    const int lw = wave & 3;                      // loader / compute index 0..3

    constexpr unsigned OOB = 0x80000000u;
    // ---- epilogue protocol, the same barrier sequence in both roles ------------------------------------------------------------
    half_t* T = reinterpret_cast<half_t*>(smem);
    float* cb = reinterpret_cast<float*>(smem + CB_OFF);
    const int rps = a.rowbias ? a.rows_per_sample : 0x40000000;      // (no time row: one "sample", index 0)
    const int smp0 = m0 / rps;

    if (is_loader) {
        // =========================================== loader waves ===========================================================
        const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x1, 0, a.bytes_x1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 ? a.x2 : a.x1), 0,
                                                                               a.x2 ? a.bytes_x2 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.bytes_w, 0x00020000);
        half8 resv[RQ];
        unsigned hoff1[HQ], hoff2[HQ], woff[WQ];
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int hr = (lw + 4 * q) * 8 + r8;
            const int seg = hr / hs, rem = hr - seg * hs;
            const int hy = rem / (Wi + 2), hx = rem - hy * (Wi + 2);
            const int img = img0 + seg;
            const int yy = (y0 >> up) + hy - 1, xx = hx - 1;          // input pixel of halo row hr
            const bool ok = (hr < nh) && (yy >= 0) && (yy < Hi) && (xx >= 0) && (xx < Wi) && ((long)img * HW < a.M);
            const long row = ((long)img * Hi + yy) * Wi + xx;
            const int swz = (c8 ^ key(hr)) * 8;
            hoff1[q] = ok ? (unsigned)((row * a.ld1 + swz) * 2) : OOB;
            hoff2[q] = ok ? (unsigned)((row * a.ld2 + swz) * 2) : OOB;
        }
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int R = (lw + 4 * q) * 8 + r8;
            woff[q] = (unsigned)(((long)(n0 + R) * a.ktot + (c8 ^ key(R)) * 8) * 2);
        }
        const int nq = (HP - lw + 3) >> 2;            // valid halo slots of this loader: pieces lw, lw + 4, ... < HP
        auto hcount = [&](int tap) { const int c = nq - 2 * tap; return tap > 6 ? 0 : (c < 0 ? 0 : (c > 2 ? 2 : c)); };
        auto issue_w = [&](int cs, int tap) {         // weight slab (cs, tap) into stage tap % 3
            const unsigned soff = (unsigned)((tap * ctot + cs * BK) * 2);
            char* dst = wbuf + (tap % NSTG) * WST;
#pragma unroll
            for (int q = 0; q < WQ; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(dst + (lw + 4 * q) * 1024), 16, woff[q], soff, 0, 0);
        };
        auto issue_h1 = [&](int cs, int q) {          // one halo piece slot of channel slab cs (q: compile-time at every call)
            if (q < HQ && q < nq) {
                char* dst = hbuf + (cs & 1) * HS + (lw + 4 * q) * 1024;
                const int cbase = cs * BK;
                if (cbase < a.c1)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x1, (lds_void*)dst, 16, hoff1[q < HQ ? q : 0], (unsigned)(cbase * 2), 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x2, (lds_void*)dst, 16, hoff2[q < HQ ? q : 0], (unsigned)((cbase - a.c1) * 2), 0, 0);
            }
        };
        // residual rows: chunk e = it * 256 + lw * 64 + lane of the [256 rows][20 chunks] tile, requested in the LAST channel
        // slab (three per window, windows 1..7) so that they travel under the last slabs' MFMAs
        const bool res_on = a.res != nullptr;
        auto rcount = [&](int tap, bool last) {
            if (!(last && res_on && tap >= 1 && tap <= 7)) return 0;
            const int left = RQ - (tap - 1) * 3;
            return left <= 0 ? 0 : (left > 3 ? 3 : left);
        };
        auto issue_res = [&](int tap) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int it = (tap - 1) * 3 + j;
                if (it >= 0 && it < RQ) {
                    const int e = it * 256 + lw * 64 + lane;
                    const int row = e / 20, ch = e - row * 20;
                    resv[it] = *reinterpret_cast<const half8*>(a.res + (long)(m0 + row) * a.ldr + n0 + ch * 8);
                }
            }
        };
        // window (cs, tap), behind the barrier that publishes slab (cs, tap): the weight slab two ahead, two halo slots of
        // cs + 1, residual chunks in the last channel slab
        auto window = [&](int cs, int tap) {
            const bool last = cs + 1 >= cs_end;
            if (tap + 2 < 9) issue_w(cs, tap + 2);
            else if (!last) issue_w(cs + 1, tap + 2 - 9);
            if (!last && tap <= 6) { issue_h1(cs + 1, 2 * tap); issue_h1(cs + 1, 2 * tap + 1); }
            // the counted vmcnt below relies on the issue ORDER weights, halo, residual: the residual rows are ordinary
            // global loads the compiler could otherwise hoist above the (non-aliasing) DMA intrinsics of this window
            asm volatile("" ::: "memory");
            if (rcount(tap, last)) issue_res(tap);
        };
        auto wcount = [&](int tap, bool last) {       // loads issued by window (cs, tap)
            return ((tap + 2 < 9 || !last) ? WQ : 0) + (last ? 0 : hcount(tap)) + rcount(tap, last);
        };
        // ---- prologue: halo of the first channel slab, weight slabs 0 and 1
#pragma unroll
        for (int q = 0; q < HQ; ++q) issue_h1(cs_begin, q);
        issue_w(cs_begin, 0);
        issue_w(cs_begin, 1);
        wait_vmcnt(WQ);
        __builtin_amdgcn_s_barrier();                 // publishes slab (cs_begin, 0)
        for (int cs = cs_begin; cs < cs_end; ++cs) {
            const bool last = cs + 1 >= cs_end;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                window(cs, tap);
                // the next slab's weights have landed; younger issues may fly: everything of this window, and what the
                // previous window issued behind its weight pieces (tap 0: the previous window is tap 8, weights only)
                int n = wcount(tap, last);
                if (tap >= 1) n += (last ? 0 : hcount(tap - 1)) + rcount(tap - 1, last);
                wait_vmcnt(n);
                __builtin_amdgcn_s_barrier();         // publishes the next slab
            }
        }
        // ---- loader epilogue: residual rows -> staging tile; bias[n] + time row[sample][n] of the (<= 4) samples of the tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // the ring and the halo buffers are free
        if (res_on) {
#pragma unroll
            for (int it = 0; it < RQ; ++it) {
                const int e = it * 256 + lw * 64 + lane;
                const int row = e / 20, ch = e - row * 20;
                *reinterpret_cast<half8*>(T + row * TP + ch * 8) = resv[it];
            }
        }
        {
            const int j = lw * 64 + lane;
            if (j < BN) {
                const float b0 = a.bias ? (float)a.bias[n0 + j] : 0.f;
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const long smp = smp0 + sidx;
                    const bool ok = a.rowbias != nullptr && smp * rps < a.M;
                    cb[sidx * BN + j] = b0 + (ok ? (float)a.rowbias[smp * a.ld_rowbias + n0 + j] : 0.f);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // residual + bias images are in LDS
        __builtin_amdgcn_s_barrier();                 // the compute waves have rounded the tile
    } else {
        // =========================================== compute waves ==========================================================
        int hbase[TM], ty_[TM], tx_[TM];            // halo row of the lane's token before the tap shift; its (row in the tile, column)
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int ml = (lw * TM + t) * 32 + l32;
            const int seg = ml / seg_tok, rem = ml - seg * seg_tok;
            const int y = rem / W, x = rem - y * W;
            hbase[t] = seg * hs + (y + 1) * (W + 2) + (x + 1);
            ty_[t] = y + (y0 & up); tx_[t] = x;      // (up: row parity of the tile's first row -- 0, tiles are whole even rows)
            asm volatile("" : "+v"(hbase[t]));
        }
        const int wlane = l32 * 128 + ((lh ^ key(l32)) << 4);
        f32x16 acc[TM][5];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int c = 0; c < 5; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][c][i] = 0.f;
        int xaddr[TM];
        auto set_slab = [&](int cs, int tap) {
            const int ty = tap / 3;
            const int shift = (ty - 1) * (W + 2) + (tap - ty * 3 - 1);
            const int hsel = (cs & 1) * HS;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                int hb = hbase[t];
                asm volatile("" : "+v"(hb));          // (keeps the nine taps' addresses from being hoisted out of the channel-slab loop: 72 registers)
                int row = hb + shift;
                if (up) {
                    int yy = ty_[t], xx = tx_[t];
                    asm volatile("" : "+v"(yy), "+v"(xx));
                    // input pixel ((y + ty - 1) >> 1, (x + tx - 1) >> 1), halo origin = input row (y0 >> 1) - 1, column -1
                    row = (((yy + ty - 1) >> 1) + 1) * (Wi + 2) + ((xx + (tap - ty * 3) - 1) >> 1) + 1;
                }
                xaddr[t] = hsel + row * 128 + ((lh ^ key(row)) << 4);
            }
        };
        // k-step j of a slab: chunk (2 j + lh) ^ key == ((lh ^ key) ^ (2 j)): one address per tile, XOR j << 5
        auto read_frags = [&](int stage, int j, half8 (&xf)[TM], half8 (&wf)[5]) {
            const char* wsb = wbuf + stage * WST + (wlane ^ (j << 5));
#pragma unroll
            for (int t = 0; t < TM; ++t) xf[t] = *reinterpret_cast<const half8*>(hbuf + (xaddr[t] ^ (j << 5)));
#pragma unroll
            for (int c = 0; c < 5; ++c) wf[c] = *reinterpret_cast<const half8*>(wsb + c * 4096);
        };
        // operands "swapped" (MFMA A = weight rows, B = token rows): a lane owns ONE token and, per 32-channel tile, four
        // quads of consecutive channels (8 g + 4 lh + i)
        auto mma = [&](const half8 (&xf)[TM], const half8 (&wf)[5]) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int c = 0; c < 5; ++c) acc[t][c] = mfma32(wf[c], xf[t], acc[t][c]);
        };
        auto interleave = [&]() {                     // MFMA, read, MFMA, read ... : the reads issue in the MFMAs' shadow
            if constexpr (TM == 2) {
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
            } else {                                  // five MFMAs, six reads
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        };
        __builtin_amdgcn_s_barrier();                 // slab (cs_begin, 0) is published
        half8 xf0[TM], wf0[5], xf1[TM], wf1[5];
        set_slab(cs_begin, 0);
        read_frags(0, 0, xf0, wf0);
        for (int cs = cs_begin; cs < cs_end; ++cs) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int st = tap % NSTG, stn = (tap + 1) % NSTG;      // (nine slabs per channel slab, three stages: static)
                read_frags(st, 1, xf1, wf1);
                mma(xf0, wf0);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
                read_frags(st, 2, xf0, wf0);
                mma(xf1, wf1);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
                read_frags(st, 3, xf1, wf1);
                mma(xf0, wf0);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
                // every wave's reads of this slab have landed (its stage and, at tap 8, its halo buffer may be refilled);
                // the next slab is published
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // fourth k-step, the next slab's first fragments travel under it (behind the last slab: stale LDS, unused)
                if (tap < 8) set_slab(cs, tap + 1); else set_slab(cs + 1, 0);
                read_frags(stn, 0, xf0, wf0);
                mma(xf1, wf1);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- compute epilogue ---------------------------------------------------------------------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // the ring and the halo buffers are free
        __builtin_amdgcn_s_barrier();                 // residual + bias images are in LDS
        {
            const bool has_res = a.res != nullptr;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int rowt = (lw * TM + t) * 32 + l32;
                const int sidx = (m0 + rowt) / rps - smp0;
                const float* cbs = cb + sidx * BN + 4 * lh;
                half_t* trow = T + rowt * TP + 4 * lh;
#pragma unroll
                for (int c = 0; c < 5; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nl = c * 32 + g * 8;
                        const f32x4 b = *reinterpret_cast<const f32x4*>(cbs + nl);
                        float v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = acc[t][c][4 * g + i] + b[i];
                        if (has_res) {
                            const half4 r = *reinterpret_cast<const half4*>(trow + nl);
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] += (float)r[i];
                        }
                        half4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = (half_t)v[i];
                        *reinterpret_cast<half4*>(trow + nl) = o;
                    }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // the fp16 tile is complete
    }

    // ---- whole rows out, 16 B per lane, all eight waves
#pragma unroll
    for (int it = 0; it < BM * 20 / 512; ++it) {
        const int e = it * 512 + tid;
        const int row = e / 20, ch = e - row * 20;
        const half8 v = *reinterpret_cast<const half8*>(T + row * TP + ch * 8);
        *reinterpret_cast<half8*>(a.out + (long)(m0 + row) * a.ldo + n0 + ch * 8) = v;
        if (a.out2) *reinterpret_cast<half8*>(a.out2 + (long)(m0 + row) * a.ldo + n0 + ch * 8) = v;
    }
    if (a.gn_stats != nullptr) {
        // GroupNorm statistics of the tensor just written: (sum, sum of squares) of the fp16 outputs per statistics unit of
        // this [256 tokens] x [160 channels] tile -- thread (segment of 32 tokens, channel quad) sums a column strip of the
        // staged tile, thread u adds the eight segments and the gn_unit channels of unit u in a fixed order
        float* gs = reinterpret_cast<float*>(smem + GS_OFF);
        if (tid < 40 * NSEG) {
            const int seg = tid / 40, q = tid - seg * 40;
            float s[4][4], ss[4][4];                   // [row mod 4][channel]: four independent chains, added pairwise
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) { s[k][i] = 0.f; ss[k][i] = 0.f; }
            const half_t* col = T + (seg * 32) * TP + q * 4;
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const half4 h = *reinterpret_cast<const half4*>(col + r * TP);
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float f = (float)h[i]; s[r & 3][i] += f; ss[r & 3][i] = __builtin_fmaf(f, f, ss[r & 3][i]); }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                gs[(seg * BN + q * 4 + i) * 2] = (s[0][i] + s[1][i]) + (s[2][i] + s[3][i]);
                gs[(seg * BN + q * 4 + i) * 2 + 1] = (ss[0][i] + ss[1][i]) + (ss[2][i] + ss[3][i]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // thread (channel, sum | sum of squares): the segments in order.  From here on in double: the strip sums are small, the
        // totals of 2 560+ values cancel from +-50 down to ~0.1 -- fp32 adds at that magnitude cost 1e-4 of absolute error
        double* gc = reinterpret_cast<double*>(gs + NSEG * BN * 2);      // [160 channels][2]
        if (tid < 2 * BN) {
            float v[NSEG];
#pragma unroll
            for (int seg = 0; seg < NSEG; ++seg) v[seg] = gs[seg * BN * 2 + tid];
            double sum = (double)v[0];
#pragma unroll
            for (int seg = 1; seg < NSEG; ++seg) sum += (double)v[seg];
            gc[tid] = sum;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // thread (unit, sum | sum of squares): gn_unit (10 / 20 / 40) channels in order, five independent reads at a time
        const int nu = BN / a.gn_unit;
        if (tid < 2 * nu) {
            const int u = tid >> 1, which = tid & 1;
            double sum = 0.0;
            for (int c0 = 0; c0 < a.gn_unit; c0 += 5) {       // (N / 32 with N a multiple of 160: a multiple of 5)
                double v[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) v[i] = gc[(u * a.gn_unit + c0 + i) * 2 + which];
#pragma unroll
                for (int i = 0; i < 5; ++i) sum += v[i];
            }
            a.gn_stats[((long)(m0 / BM) * (a.N / a.gn_unit) + n0 / a.gn_unit + u) * 2 + which] = (float)sum;
        }
    }
#endif
}

}  // namespace

template <int TM>
static int launch_tm(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 128 * TM;
    const int W = a.Wo, HW = a.Ho * a.Wo;
    const int seg = BM < HW ? BM : HW;
    const int nh = (BM / seg) * (((seg / W) >> a.up) + 2) * ((W >> a.up) + 2);
    int smem = 2 * ((nh + 7) / 8) * 1024 + NSTG * WST;
    if (smem < epi_bytes(BM)) smem = epi_bytes(BM);      // (an input-resolution halo is small: the epilogue images set the size)
    if (smem > 160 * 1024) {
        cid_set_error("cid_gemm_f16: conv_h32 needs %d bytes of LDS (epilogue images: %d)", smem, epi_bytes(BM));
        return -22;
    }
    static bool configured = false;
    if (!configured) {
        hipError_t herr = hipFuncSetAttribute((const void*)conv_h32_kernel<TM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (herr != hipSuccess) {
            cid_set_error("cid_gemm_f16: cannot reserve 160 KB of LDS (%s)", hipGetErrorString(herr));
            return -5;
        }
        configured = true;
    }
    dim3 grid(a.N / BN, a.M / BM, 1);
    GemmArgs b = a;
    b.xcd_pn = cidg::choose_xcd_pn((int)grid.x, (int)grid.y, 2.0 * a.N * a.ktot, (double)a.bytes_x1 + a.bytes_x2);
    hipLaunchKernelGGL(conv_h32_kernel<TM>, grid, dim3(512), smem, s, b);
    return 0;
}

int cidg::launch_conv_h32(const GemmArgs& a, int bm, hipStream_t s) {
    return bm == 256 ? launch_tm<2>(a, s) : launch_tm<1>(a, s);
}
