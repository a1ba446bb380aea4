/* cid.h -- C ABI of libcid.so: hand-written HIP kernels (gfx950 / MI355X) for the
 * ConsistentID UNet-denoise hot path.
 *
 * The reference (JackAILab/ConsistentID) has no FFI of its own: its plugin boundary
 * is diffusers' Python attention-processor protocol plus the UNet call signature
 * (SURVEY.md section 8b).  Each entry point below names the reference interface it
 * replaces (file:line relative to /root/reference; "D:" marks diffusers==0.23.0
 * internals that the reference calls but does not vendor).  INTEGRATION.md shows the
 * ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host"; fp16 tensors are
 *     IEEE binary16, row-major, token-major ("NHWC": [batch, H*W, C]);
 *   - no allocation, no retained pointers, no host synchronisation inside;
 *     everything is enqueued on `stream` (a hipStream_t) and is graph-capturable;
 *   - return 0 on success, negative errno-style code on failure
 *     (-22 bad argument, -5 launch failure); cid_last_error() gives the text;
 *   - 16-byte alignment is required for every tensor base and row pitch.
 */
#ifndef CID_H
#define CID_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* cid_stream_t;      /* hipStream_t */
typedef uint16_t cid_half;       /* IEEE binary16 bit pattern */

int cid_version(void);
const char* cid_last_error(void);

/* ---------------------------------------------------------------------------
 * Implicit GEMM:  out[m][n] = epilogue( sum_k A(m,k) * W[n][k] )
 * Replaces (D:) nn.Linear / nn.Conv2d 1x1 / 3x3 (stride 1|2, pad 1) / nearest-2x
 * upsample + conv, i.e. ResnetBlock2D.conv1/conv2/conv_shortcut, Downsample2D,
 * Upsample2D, Transformer2DModel.proj_in/out, Attention.to_q/to_k/to_v/to_out,
 * FeedForward (GEGLU) -- SURVEY.md 8a rows a5-a9; LoRA of attention.py:139-162,
 * :236-282 is merged into W by the host.
 *   c1 + c2 must be a multiple of 64 (and c1 too when c2 > 0); N a multiple of 32;
 *   row pitches ld1, ld2, ldo multiples of 8 halfs (rows start 16-byte aligned).
 *   A(m, k): k = tap * (c1 + c2) + c ; for taps == 9 the row m = (b, y, x) of the
 *   OUTPUT grid reads input pixel (y*stride + dy - 1, x*stride + dx - 1) of the
 *   (optionally 2x nearest-upsampled) input, zero outside.  Channels [0, c1) come
 *   from x1, [c1, c1 + c2) from x2 (skip concat without a copy).
 *   mode 0: out[m][n] = acc + bias[n] + rowbias[m / rows_per_sample][n] + res[m][n]
 *           (taps == 9, stride 1, up 0|1, N % 160 == 0: ResnetBlock2D.conv1 / conv2 and Upsample2D.conv run on csrc/conv3x3.hip
 *            -- 32 x 32 MFMA tiles, loader / compute wave roles -- wherever its tiles fill the chip unsplit; same arithmetic)
 *   mode 1: GEGLU. W rows are interleaved in blocks of 16 (value block, gate block);
 *           out[m][j] = (val + bias_v) * gelu_erf(gate + bias_g), out width N / 2;
 *           bias is interleaved the same way.
 *   mode 2: fused QKV for self-attention. Columns [0, n_vt0) are written row-major
 *           to out; columns [n_vt0, N) (the V third) are written TRANSPOSED to
 *           vt[b][head][d][pos(token)] (see cid_self_attn_f16).
 *   mode 3: the QUERY projection of the identity cross-attention with the attention as its epilogue
 *           (Consistent_IPAttProcessor.__call__, attention.py:236 `query = attn.to_q(hidden_states) + lora` followed by
 *           :259-279, the two softmaxes over the text and the ID keys): W = the merged to_q weight pre-multiplied by
 *           d^-0.5 * log2(e), N = heads * dhead; out[m][h * dhead + j] = the attention output of head h BEFORE to_out
 *           (what cid_id_xattn_core_f16 writes) -- q never goes to memory.  att_kp / att_vp: the packed K / V^T of
 *           cid_kv_pack_f16, att_kvrow[b] the context row of sample b, ntok tokens per sample (a multiple of 64; of 128 for dhead 64),
 *           the reference's 77 + 4 context, dhead in {64, 80, 160}.  No bias / rowbias / res / split-K; the LayerNorm
 *           fold applies (norm2 in front of to_q).
 */
typedef struct cid_gemm_desc {
    const cid_half* x1; const cid_half* x2;
    int32_t c1, c2, ld1, ld2;
    const cid_half* w;            /* [N][taps * (c1 + c2)] */
    cid_half* out; int32_t ldo;
    const cid_half* bias;         /* [N] or NULL */
    const cid_half* rowbias; int32_t ld_rowbias; int32_t rows_per_sample; /* or NULL */
    const cid_half* res; int32_t ldr;                                      /* or NULL; 16-byte aligned, ldr % 8 == 0 (like out / ldo) */
    int32_t M, N;
    int32_t taps;                 /* 1 or 9 */
    int32_t Hi, Wi, Ho, Wo, stride, up; /* conv geometry (taps == 9) */
    int32_t mode;
    cid_half* vt; int32_t n_vt0, heads, dhead, dvp, ntok; /* mode 2 */
    void* ws; int64_t ws_bytes;   /* optional fp32 scratch for split-K (small M, deep K); NULL => never split */
    /* LayerNorm folded into the projection (D: BasicTransformerBlock.norm1 / norm2 / norm3 in front of
     * Attention.to_q/to_k/to_v, to_q and FeedForward's GEGLU; attention.py:133-147, :236): x1 is the RAW residual stream,
     * W = W0 diag(gamma) rounded to fp16, ln_s[n] = sum_k W[n][k] (fp32, of the rounded W), ln_b[n] = (W0 beta)[n] +
     * bias[n] (fp32); the kernel computes mean / rstd of every row of x1 on the fly and writes
     * rstd * (acc - mean * ln_s[n]) + ln_b[n] through the mode's epilogue.  taps == 1, c2 == 0, bias == NULL. */
    const float* ln_s; const float* ln_b; float ln_eps;   /* both NULL => no LayerNorm */
    /* GroupNorm statistics of `out` for its consumer (D: ResnetBlock2D.norm1 / norm2, Transformer2DModel.norm,
     * conv_norm_out read what a conv / proj_out just wrote): fp32 [M / rows][32][2] = (sum, sum of squares) of the fp16
     * outputs over `rows` consecutive tokens x N / 32 consecutive channels, rows = cid_gemm_stats_rows(d) (> 0 required);
     * consumed by cid_groupnorm_stats_f16.  mode 0 only. */
    float* gn_stats;
    /* mode 3 */
    const cid_half* att_kp; const cid_half* att_vp; const int32_t* att_kvrow;
    int32_t att_n_txt, att_n_ip; float att_ip_scale;
    /* Second destination (or NULL): every output row is stored to out2 + m * ldo as well -- the CFG `torch.cat([x] * 2)`
     * of a tensor both halves of the batch share (ref pipline_StableDiffusion_ConsistentID.py:537-539) written by its
     * producer instead of by a copy launch.  mode 0 only. */
    cid_half* out2;
} cid_gemm_desc;
int cid_gemm_f16(const cid_gemm_desc* d, cid_stream_t stream);
/* Token rows per statistics block if cid_gemm_f16(d) can emit gn_stats (its tile height), 0 if it cannot (split-K,
 * tile widths off the 160-channel grid, ragged M): the caller then lets cid_groupnorm_f16 take its own statistics. */
int cid_gemm_stats_rows(const cid_gemm_desc* d);

/* ---------------------------------------------------------------------------
 * Self-attention core (replaces the softmax(QK^T)V of Consistent_AttProcessor,
 * attention.py:149-159; q/k/v/out projections incl. LoRA are cid_gemm_f16 calls).
 *   q : [B][N][ldq] slice, head h at columns h*d .. ; pre-multiplied by
 *       scale * log2(e) (folded into the merged to_q weight by the host)
 *   k : same layout;  vt : [B][heads][dvp][N] fp16 with the token axis permuted
 *       inside every group of 16:  pos = (t & ~15) | (8*((t>>2)&1) + 4*((t>>3)&1) + (t&3))
 *   out[b][n][h*d + j].   N % 64 == 0;  d in {40, 64, 80, 160}.
 */
int cid_self_attn_f16(const cid_half* q, const cid_half* k, const cid_half* vt, cid_half* out,
                      int32_t B, int32_t N, int32_t heads, int32_t d,
                      int32_t ldq, int32_t ldk, int32_t dvp, int32_t ldo, cid_stream_t stream);
/* Same, with only the first n_keys (<= N) keys of every sample real and the rest padding (scores -inf): token counts
 * that are not a multiple of 64 -- the 257 tokens of the CLIP-ViT-H vision tower whose hidden_states[-2] the reference
 * feeds to ProjPlusModel / FacialEncoder (pipline_StableDiffusion_ConsistentID.py:182-183, :200-201).  Rows beyond n_keys
 * of q / out are computed like any other and are for the caller to ignore. */
int cid_self_attn_keys_f16(const cid_half* q, const cid_half* k, const cid_half* vt, cid_half* out,
                           int32_t B, int32_t N, int32_t heads, int32_t d, int32_t ldq, int32_t ldk,
                           int32_t dvp, int32_t ldo, int32_t n_keys, cid_stream_t stream);

/* ---------------------------------------------------------------------------
 * Fused identity cross-attention = Consistent_IPAttProcessor.__call__
 * (attention.py:207-294) with LoRA merged:
 *   q   = LN?(x) Wq'^T                               (:236)
 *   o   = softmax(q Kt^T) Vt + ip_scale * softmax(q Kip^T) Vip   (:259-279)
 *   out = o Wo'^T + bo (+ residual)                  (:282)
 * K/V of both streams come pre-projected and pre-packed (cid_kv_pack_f16) because
 * encoder_hidden_states is step-invariant.  One launch, x read once, out written
 * once.  `kvrow[s]` selects the packed K/V row used by sample s.
 *   x, out, residual : [B][N][C] fp16 (ld = C).  ln_gamma/ln_beta NULL => no LN.
 *   wq, wo  : merged [C][C] weights, row-major ([out][in]); wq additionally
 *             pre-scaled by d^-0.5 * log2(e).
 */
int cid_id_xattn_f16(const cid_half* x, cid_half* out, const cid_half* residual,
                     const cid_half* ln_gamma, const cid_half* ln_beta, float ln_eps,
                     const cid_half* wq, const cid_half* wo, const cid_half* bo,
                     const cid_half* kp, const cid_half* vp, const int32_t* kvrow,
                     int32_t B, int32_t N, int32_t C, int32_t heads,
                     int32_t n_txt, int32_t n_ip, float ip_scale, cid_stream_t stream);
/* Attention core only (same two-stream softmax.V, no projections): q already projected and
 * pre-scaled, out = o.  Used for the 1280-channel levels, where [tokens x C] tiles are too
 * large to keep resident and the projections run as cid_gemm_f16 calls instead. */
int cid_id_xattn_core_f16(const cid_half* q, cid_half* out, const cid_half* kp, const cid_half* vp,
                          const int32_t* kvrow, int32_t B, int32_t N, int32_t C, int32_t heads,
                          int32_t n_txt, int32_t n_ip, float ip_scale, cid_stream_t stream);
/* bytes of one packed K row / V row (per embed row) for (C, heads) */
int64_t cid_kv_pack_elems(int32_t C, int32_t heads, int32_t which /*0=K,1=V*/);
/* kv_txt, kv_ip: [R][L][2C] = [K | V] projections of all L = n_txt + n_ip context
 * rows with the text weights resp. the to_k_ip/to_v_ip weights (attention.py:249-250,
 * :266-269); packs rows < n_txt from kv_txt and rows >= n_txt from kv_ip. */
int cid_kv_pack_f16(const cid_half* kv_txt, const cid_half* kv_ip, cid_half* kp, cid_half* vp,
                    int32_t R, int32_t C, int32_t heads, int32_t n_txt, int32_t n_ip,
                    cid_stream_t stream);
/* [rows][K] fp16 row-major -> MFMA A-fragment order [rows/32][K/16][64 lanes][8]. */
int cid_pack_wfrag_f16(const cid_half* w, cid_half* wp, int32_t rows, int32_t K, cid_stream_t stream);

/* ---------------------------------------------------------------------------
 * Fused identity cross-attention with the BasicTransformerBlock wrapper, ONE launch (csrc/xattn3.hip):
 * Consistent_IPAttProcessor.__call__ (attention.py:207-294) INCLUDING  x + attn2(LayerNorm(x), ehs)
 * (D: diffusers BasicTransformerBlock norm2 / residual; called from pipline_StableDiffusion_ConsistentID.py:552-557
 * through the UNet):
 *   q   = rstd * (x Wq'^T - mean * s) + b'     LayerNorm folded: Wq' = Wq diag(gamma) * d^-0.5 * log2(e)
 *                                              (fp16), s[n] = sum_k Wq'[n][k] (fp32), b' = Wq beta (fp32)
 *   o   = softmax(q Kt^T) Vt + ip_scale * softmax(q Kip^T) Vip            (:259-279)
 *   out = o Wo^T + bo (+ x)                                                (:282)
 * Built for the SD1.5 level-0 geometry (C = 320, 8 heads of 40; cid_id_xattn3_supported tells) with the
 * reference's 77 + 4 context or ControlNet's 81 + 0; N % 64 == 0.  x is read from HBM once, out written once.
 * 64-token tiles, two 4-wave workgroups per CU, weights streamed L2 -> registers in MFMA A-operand order.
 *   q_rowsum, q_bias : fp32 [C] (zeros when there is no LayerNorm);
 *   kp, vp           : context rows in MFMA-fragment order, cid_kv_pack2_elems halfs per row, produced by
 *                      cid_gather_pack_f16 with the host tables of consistentid_amd/xattn_pack.py (key order "reg");
 *   wq_packed, wo_packed : the [C][C] matrices Wq' / Wo re-ordered as [wave 4][k-step 10][row tile 5][lane 64][8]:
 *                          element (wave w, k-step s, tile t, lane l, j) = W[80 w + 16 t + (l & 15)][32 s + 8 (l >> 4) + j]
 *                          (consistentid_amd/xattn_pack.pack_w3; same element count as the plain matrix);
 *   flags            : bit 0 = LayerNorm folded (mean / rstd are computed in-kernel), bit 1 = add x (residual). */
int64_t cid_kv_pack2_elems(int32_t C, int32_t heads, int32_t which /*0=K,1=V*/);
int cid_id_xattn3_supported(int32_t C, int32_t heads, int32_t n_txt, int32_t n_ip);
int cid_id_xattn3_f16(const cid_half* x, cid_half* out, const cid_half* wq_packed, const float* q_rowsum,
                      const float* q_bias, const cid_half* wo_packed, const cid_half* bo, const cid_half* kp,
                      const cid_half* vp, const int32_t* kvrow, int32_t B, int32_t N, int32_t C, int32_t heads,
                      int32_t n_txt, int32_t n_ip, float ip_scale, float ln_eps, int32_t flags,
                      cid_stream_t stream);
/* dst[r][e] = idx[e] < 0 ? 0 : (bit 30 of idx[e] ? src_b : src_a)[r * src_row_elems + (idx[e] & 0x3fffffff)]
 * for r < R, e < n_idx: the generic "put projected K / V rows into fragment order" step (idx on the device). */
int cid_gather_pack_f16(const cid_half* src_a, const cid_half* src_b, const int32_t* idx, cid_half* dst,
                        int32_t R, int64_t src_row_elems, int64_t n_idx, cid_stream_t stream);

/* ---------------------------------------------------------------------------
 * Normalisation (D: nn.LayerNorm eps 1e-5 in BasicTransformerBlock; nn.GroupNorm
 * 32 groups, eps 1e-5 in ResnetBlock2D / conv_norm_out, 1e-6 in Transformer2DModel),
 * SiLU optionally fused (D: ResnetBlock2D.nonlinearity).  SURVEY.md 8a a5-a7.
 */
int cid_layernorm_f16(const cid_half* x, cid_half* out, const cid_half* gamma, const cid_half* beta,
                      int32_t M, int32_t C, float eps, cid_stream_t stream);
/* In-place softmax over fp16 rows of BASE-2 logits (fp32 math): replaces the softmax of diffusers' default
 * attention in the VAE mid block (single head of width 512 over all latent pixels), which the reference reaches
 * through self.decode_latents / vae.decode (pipline_StableDiffusion_ConsistentID.py:587,:597).  cols % 8 == 0. */
int cid_softmax_rows_f16(cid_half* x, int32_t rows, int32_t cols, int64_t ld, cid_stream_t stream);

/* x = concat(x1[.., c1], x2[.., c2]) per token; ws: >= cid_groupnorm_ws_bytes() scratch. */
int64_t cid_groupnorm_ws_bytes(int32_t B, int32_t C);
int cid_groupnorm_f16(const cid_half* x1, const cid_half* x2, int32_t c1, int32_t c2,
                      cid_half* out, const cid_half* gamma, const cid_half* beta,
                      int32_t B, int32_t HW, int32_t groups, float eps, int32_t silu,
                      void* ws, cid_stream_t stream);
/* GroupNorm (+ SiLU) whose statistics were emitted by the GEMM epilogue(s) that wrote the input (cid_gemm_desc.gn_stats):
 * ONE launch, x read once -- the statistics pass of cid_groupnorm_f16 is gone.  stats1 / stats2 belong to x1 / x2 (skip
 * concat: two tensors, two producers), rows1 / rows2 = the cid_gemm_stats_rows of their producers; cid_groupnorm_stats_ok
 * tells whether every group is a whole number of c / 32-channel units of one source (it is for every GroupNorm of the
 * SD1.5 / SDXL UNets except the 1280+640 and 640+320 concatenations). */
int cid_groupnorm_stats_ok(int32_t c1, int32_t c2, int32_t groups);
int cid_groupnorm_stats_f16(const cid_half* x1, const cid_half* x2, int32_t c1, int32_t c2,
                            cid_half* out, const cid_half* gamma, const cid_half* beta,
                            int32_t B, int32_t HW, int32_t groups, float eps, int32_t silu,
                            const float* stats1, int32_t rows1, const float* stats2, int32_t rows2,
                            cid_stream_t stream);

/* ---------------------------------------------------------------------------
 * fp32 kernels of the SDXL VAE decode.  The reference upcasts that VAE to float32 before decoding
 * (pipline_StableDiffusionXL_ConsistentID.py:670-676: upcast_vae(), vae.decode(latents / scaling_factor)); these
 * three entry points are the fp32 counterparts of cid_gemm_f16 / cid_groupnorm_f16 / cid_softmax_rows_f16 that the
 * fp32 decoder engine (consistentid_amd/vae.py::HipVAEDecoderF32) is built on.  All tensors fp32, token-major.
 *   cid_gemm_f32: out[m][n] = sum_k A(m,k) W[n][k] + bias[n] + res[m][n], k = tap * c + ch; taps 1 (Linear / 1x1) or
 *     9 (3x3, stride 1, pad 1, `up` = input nearest-upsampled 2x first: Ho = Hi << up); exact f32 accumulation on
 *     v_mfma_f32_32x32x2_f32.  ldx % 4 == 0 when c % 4 == 0.
 *   cid_groupnorm_f32: GroupNorm over [B][HW][C] (+ SiLU); ws >= cid_groupnorm_f32_ws_bytes(B, HW, C).
 *   cid_softmax_rows_f32: in place, base-2 logits. */
int cid_gemm_f32(const float* x, const float* w, const float* bias, const float* res, float* out,
                 int32_t M, int32_t N, int32_t c, int32_t taps, int32_t ldx, int32_t ldo, int32_t ldr,
                 int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t up, cid_stream_t stream);
int64_t cid_groupnorm_f32_ws_bytes(int32_t B, int32_t HW, int32_t C);
int cid_groupnorm_f32(const float* x, float* out, const float* gamma, const float* beta, int32_t B, int32_t HW,
                      int32_t C, int32_t groups, float eps, int32_t silu, void* ws, cid_stream_t stream);
int cid_softmax_rows_f32(float* x, int32_t rows, int32_t cols, int64_t ld, cid_stream_t stream);

/* ---------------------------------------------------------------------------
 * UNet ends (D: UNet2DConditionModel.conv_in / conv_out), HBM-bound direct kernels.
 * conv_in : sample NCHW fp16 [Bin][cin][H][W] -> token-major [B][H*W][cout];
 *           batch b reads sample (b % Bin)  (the CFG torch.cat([latents]*2),
 *           pipline_StableDiffusion_ConsistentID.py:537-539, without the copy);
 *           B % Bin == 0: rows that share a sample are computed once and stored B / Bin times.
 *           w: [cout][9][cin], bias [cout]; cin <= 9, cout % 8 == 0, cout <= 320.
 * conv_out: token-major [B][H*W][cin] -> NCHW fp16 [B][cout<=4][H][W]; w [cout][9][cin].
 */
int cid_conv_in_f16(const cid_half* sample, cid_half* out, const cid_half* w, const cid_half* bias,
                    int32_t B, int32_t Bin, int32_t cin, int32_t H, int32_t W, int32_t cout,
                    const float* in_scale /* device scalar multiplying the sample (scheduler.scale_model_input,
                                             pipline_StableDiffusion_ConsistentID.py:540), or NULL */,
                    cid_stream_t stream);
/* conv_in of a 9-channel inpainting UNet: input channels [0, cin1) from `sample` (latents, scaled by in_scale), channels
 * [cin1, cin1 + cin2) from `extra` (cat([mask, masked_image_latents]) NCHW [Bin][cin2][H][W], NOT scaled) -- the
 * torch.cat([latent_model_input, mask, masked_image_latents], dim=1) of
 * pipelines/StableDIffusionInpaint_ConsistentID.py:320-321 and StableDIffusionControlNetInpaint_ConsistentID.py:415-416
 * without materialising it.  w: [cout][9][cin1 + cin2]. */
int cid_conv_in_cat_f16(const cid_half* sample, int32_t cin1, const cid_half* extra, int32_t cin2, cid_half* out,
                        const cid_half* w, const cid_half* bias, int32_t B, int32_t Bin, int32_t H, int32_t W,
                        int32_t cout, const float* in_scale, cid_stream_t stream);
/* Small-channel 3x3 convolution (pad 1, stride 1 or 2, optional SiLU), token-major in and out; w [cout][9][cin].
 * Replaces the nn.Conv2d + F.silu chain of diffusers' ControlNetConditioningEmbedding (3 -> 16 -> ... -> 256 -> C0)
 * that the reference reaches through self.controlnet(..., controlnet_cond=control_image, ...)
 * (pipelines/StableDIffusionControlNetInpaint_ConsistentID.py:405-412).  cout % 8 == 0; any cin. */
int cid_conv3x3_small_f16(const cid_half* x, cid_half* out, const cid_half* w, const cid_half* bias,
                          int32_t B, int32_t Hi, int32_t Wi, int32_t cin, int32_t cout, int32_t stride,
                          int32_t silu, cid_stream_t stream);
/* Identity-conditioning stack (once per image; ProjPlusModel functions.py:490-522, FacialEncoder attention.py:72-88):
 * its Linears / LayerNorms are cid_gemm_f16 / cid_layernorm_f16; these two fill the gaps.
 *   cid_gelu_f16        in-place exact-erf GELU (nn.GELU(), functions.py:395,:499, attention.py:58); n % 8 == 0
 *   cid_small_attn_f16  PerceiverAttention core (functions.py:439-447): Lq latent queries per sample attend to the keys of
 *                       two row blocks (n1 image tokens, n2 latents: the torch.cat of :433 without the copy); a key row is
 *                       [K | V] as produced by to_kv (:434); out = softmax(scale2 * q k^T) v, head width 64,
 *                       n1 + n2 <= 1024.  q [B*Lq][ldq], kv* [B*n*][ldkv], out [B*Lq][ldo]. */
int cid_gelu_f16(cid_half* x, int64_t n, cid_stream_t stream);
int cid_small_attn_f16(const cid_half* q, int32_t ldq, const cid_half* kv1, int32_t n1, const cid_half* kv2, int32_t n2,
                       int32_t ldkv, cid_half* out, int32_t ldo, int32_t B, int32_t Lq, int32_t heads, int32_t dim_head,
                       float scale2, cid_stream_t stream);
int cid_conv_out_f16(const cid_half* x, cid_half* out, const cid_half* w, const cid_half* bias,
                     int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout, cid_stream_t stream);

/* ---------------------------------------------------------------------------
 * Timestep path (D: get_timestep_embedding flip_sin_to_cos, TimestepEmbedding,
 * ResnetBlock2D.time_emb_proj).  SURVEY.md 8a a4.
 * cid_sincos_embed: out[r][0:dim/2] = cos(v[r] * f_i), out[r][dim/2:] = sin(..),
 *                   f_i = exp(-ln(10000) * i / (dim/2)); v is a DEVICE fp32 array.
 * cid_linear_small: out[m][n] = act_out( sum_k act_in(x[m][k]) W[n][k] + b[n] (+ add[m][n]) )
 *                   for M <= 32 rows; act: 0 none, 1 SiLU.
 */
int cid_sincos_embed_f16(const float* v, cid_half* out, int32_t rows, int32_t dim, cid_stream_t stream);
int cid_linear_small_f16(const cid_half* x, int32_t ldx, const cid_half* w, const cid_half* b,
                         const cid_half* add, int32_t ldadd, cid_half* out, int32_t ldo,
                         int32_t M, int32_t N, int32_t K, int32_t act_in, int32_t act_out,
                         cid_stream_t stream);

/* ---------------------------------------------------------------------------
 * Loop glue (pipline_StableDiffusion_ConsistentID.py:561-571; D: DDIMScheduler.step,
 * eta = 0):  eps = eps_u + g (eps_c - eps_u);
 *            x_prev = c_x * x + c_eps * eps   with c_x = sqrt(a_prev)/sqrt(a_t),
 *            c_eps = sqrt(1-a_prev) - sqrt(a_prev) sqrt(1-a_t)/sqrt(a_t).
 * eps: [2B][...] (uncond first), latents [B][...] updated in place.  coef: DEVICE
 * fp32 {c_x, c_eps} so the launch is graph-replayable across timesteps.
 * Optional inpaint blend (pipelines/StableDIffusionControlNetInpaint_ConsistentID.py:
 * 437-449): latents = (1-m) * (c_init*init + c_noise*noise) + m * latents, with
 * coef[2], coef[3] = {c_init, c_noise}; mask/init/noise NULL => skipped.
 */
int cid_cfg_ddim_step_f16(const cid_half* eps, cid_half* latents, const float* coef, float guidance,
                          const cid_half* mask, const cid_half* init, const cid_half* noise,
                          int32_t B, int32_t per_sample, cid_stream_t stream);
/* y[i] += a[i mod na]  (ControlNet residual adds, CN :418-425) */
int cid_add_inplace_f16(cid_half* y, const cid_half* a, int64_t n, int64_t na, cid_stream_t stream);

/* The reference's  `for i, t in enumerate(timesteps):`  (pipline_StableDiffusion_ConsistentID.py:535, SDXL :611, CN :375)
 * turns every per-step host value -- t, the scheduler coefficients of step i, the embed set selected by
 * `i <= start_merge_step` (:542-549), SDXL's pooled embeds (:620-631), the time-embedding row -- into a kernel argument.
 * Here one row per step of all of them sits in a DEVICE table ([n_rows][row_bytes], built once per generation) and this
 * launch, the first node of the captured step, copies row *counter into the buffers the step's kernels read (segment k:
 * segs[k].nbytes bytes at table + row * row_bytes + segs[k].offset -> segs[k].dst; sizes and offsets multiples of 4
 * (16-byte aligned segments, rows and destinations move as 16-byte words: ops.StepTable pads its columns so),
 * n_segs <= 8), then increments *counter: a DDIM step is ONE graph replay with no host-side copy around it.
 * row = min(*counter, n_rows - 1). */
typedef struct cid_step_seg {
    void* dst;
    int64_t offset;
    int64_t nbytes;
} cid_step_seg;
int cid_step_select(const void* table, int64_t row_bytes, int32_t n_rows, int32_t* counter, const cid_step_seg* segs,
                    int32_t n_segs, cid_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
