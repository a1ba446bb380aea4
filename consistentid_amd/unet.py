"""HIP execution engine for the UNet denoiser with ConsistentID identity injection.

Drop-in for the object the reference pipelines call as
``self.unet(latent_model_input, t, encoder_hidden_states=..., cross_attention_kwargs=...,
added_cond_kwargs=..., down_block_additional_residuals=..., mid_block_additional_residual=...)
.sample``  (pipline_StableDiffusion_ConsistentID.py:552-557, SDXL :634-641,
pipelines/StableDIffusionControlNetInpaint_ConsistentID.py:418-425).

Design (MI355X-first, not a module tree):
  * activations are token-major fp16 ``[B, H*W, C]`` end to end, so ResNet <-> Transformer
    transitions are free (no NCHW permutes) and skip concatenation is done by giving the
    consumer kernels two source pointers instead of copying;
  * every op is a hand-written HIP kernel behind the C ABI (ops.py -> libcid.so);
  * weights are packed once (weights.py); cross-attention K/V of each embed set are
    projected + packed once per generation (``set_context``), not once per step;
  * the whole step is allocation-stable and sync-free, so one hipGraph (torch.cuda.CUDAGraph
    capturing the same HIP stream) replays it for every timestep: the timestep value,
    DDIM coefficients and the K/V row selector live in device buffers.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch

from . import ops
from .unet_spec import BlockSpec, ResnetSpec, TransformerSpec, UNetConfig, walk
from .weights import PackedUNet


@dataclass
class UNetOutput:
    sample: torch.Tensor


class _Ctx:
    """packed K/V of one set of encoder_hidden_states rows, per cross-attention layer"""

    def __init__(self):
        self.kp: Dict[str, torch.Tensor] = {}
        self.vp: Dict[str, torch.Tensor] = {}
        self.v2: Dict[str, int] = {}       # generation (0 / 2 / 3) of the fused kernel whose K/V operand layout the layer holds
        self.rows = 0
        self.n_txt = 0
        self.n_ip = 0
        # implicit cache of __call__: (address, version, shape) of the caller's tensor.  The tensor itself is kept alive in
        # key_ref: without the strong reference the caching allocator may hand the same address (version 0, same shape)
        # to the NEXT prompt's embeddings and the stale K/V would be served
        self.key = None
        self.key_ref = None


class HipUNet:
    def __init__(self, cfg: UNetConfig, unet_sd: Optional[Dict[str, torch.Tensor]] = None,
                 adapter_sd: Optional[Dict[str, torch.Tensor]] = None, device="cuda:0",
                 num_tokens: int = 4, lora_scale: float = 1.0, packed: Optional[PackedUNet] = None,
                 encoder_only: bool = False, keep_base: bool = False):
        self.config = cfg
        self.device = torch.device(device)
        self.dtype = torch.float16
        self.num_tokens = num_tokens
        if packed is not None:          # weights received through distributed.broadcast_weights
            self.packed = packed
        else:
            with torch.cuda.device(self.device):
                self.packed = PackedUNet(cfg, unet_sd, adapter_sd, self.device, lora_scale, encoder_only=encoder_only,
                                         keep_base=keep_base)
        self.W = self.packed.w
        self.downs, self.mid, self.ups = walk(cfg)
        if self.packed.encoder_only:
            self.ups = []
        self._ctx = _Ctx()
        self._gn_ws: Optional[torch.Tensor] = None
        self._gemm_ws = torch.empty(64 << 20, dtype=torch.uint8, device=self.device)   # split-K partials
        # widest level that runs the one-launch fused ID cross-attention (wider levels: GEMMs around the core)
        self._xattn_fused_max_c = int(os.environ.get("CID_XATTN_FUSED_MAX_C", "320"))
        self._cfg_dedup = os.environ.get("CID_CFG_DEDUP", "1") != "0"
        self._qattn = os.environ.get("CID_QATTN", "1") != "0"      # A/B switch: query projection with the attention epilogue
        # A/B switch: generation of the fused cross-attention kernel at the SD1.5 level-0 geometry (3: xattn3.hip,
        # 1: the first-generation xattn.hip); CID_XATTN_V2=0 is the older spelling of generation 1
        self._xattn_gen = ops.xattn_generation()
        self._t_buf = torch.zeros(1, dtype=torch.float32, device=self.device)

    def load_adapter_modules(self, adapter_sd: Dict[str, torch.Tensor], lora_scale: Optional[float] = None):
        """``adapter_modules`` of a ConsistentID checkpoint -> packed weights, in place (needs ``keep_base=True``);
        the cached cross-attention K/V are dropped because the K/V projections changed."""
        with torch.cuda.device(self.device):
            self.packed.load_adapter_modules(adapter_sd, lora_scale)
        self._ctx.key = self._ctx.key_ref = None
        return self

    # -- attributes the reference pipelines read
    @property
    def in_channels(self) -> int:
        return self.config.in_channels

    def to(self, *a, **k):
        return self

    # ------------------------------------------------------------------ context (K/V cache)
    def set_context(self, ehs: torch.Tensor, num_tokens: Optional[int] = None):
        """Project + pack cross-attention K/V for ``ehs`` [R, L, Dc] (L = text + ID tokens).
        Replaces the per-step to_k/to_v/to_k_ip/to_v_ip GEMMs of attention.py:249-250,266-269."""
        nt = self.num_tokens if num_tokens is None else num_tokens
        ehs = ehs.to(device=self.device, dtype=torch.float16).contiguous()
        R, L, Dc = ehs.shape
        ctx = self._ctx
        ctx.rows, ctx.n_txt, ctx.n_ip = R, L - nt, nt
        M = R * L
        cache: Dict[int, torch.Tensor] = {}
        for b in self.packed.xattn_layers:
            wt = self.W[f"{b}.attn2.kv_txt.w"]
            C2 = wt.shape[0]
            C_ = C2 // 2
            heads = self._heads_of(b)
            kv_txt = torch.empty(M, C2, dtype=torch.float16, device=self.device)
            kv_ip = torch.empty(M, C2, dtype=torch.float16, device=self.device)
            ops.gemm(ehs, wt, kv_txt, M=M, N=C2, c1=Dc)
            ops.gemm(ehs, self.W[f"{b}.attn2.kv_ip.w"], kv_ip, M=M, N=C2, c1=Dc)
            v3 = (self._xattn_gen >= 3 and f"{b}.attn2.wq_p" in self.W
                  and ops.id_xattn3_supported(C_, heads, ctx.n_txt, ctx.n_ip))
            ke, ve = ops.kv_pack2_elems(C_, heads) if v3 else ops.kv_pack_elems(C_, heads)
            kp, vp = ctx.kp.get(b), ctx.vp.get(b)
            if kp is None or kp.numel() != R * ke:   # keep addresses stable across generations
                kp = torch.empty(R * ke, dtype=torch.float16, device=self.device)
                vp = torch.empty(R * ve, dtype=torch.float16, device=self.device)
            if v3:      # fragment order of the third-generation fused kernel (SD1.5 level 0)
                ops.kv_pack2(kv_txt, kv_ip, kp, vp, R=R, L=L, C_=C_, heads=heads, n_txt=ctx.n_txt, n_ip=ctx.n_ip, order="reg")
            else:
                ops.kv_pack(kv_txt, kv_ip, kp, vp, R=R, C_=C_, heads=heads, n_txt=ctx.n_txt, n_ip=ctx.n_ip)
            ctx.kp[b], ctx.vp[b] = kp, vp
            ctx.v2[b] = 3 if v3 else 0
        ctx.key, ctx.key_ref = (ehs.data_ptr(), ehs._version, tuple(ehs.shape)), ehs
        return self

    def context_addresses(self):
        return tuple(t.data_ptr() for t in self._ctx.kp.values()) + (self._ctx.n_txt, self._ctx.n_ip)

    def _heads_of(self, block_name: str) -> int:
        for blk in self.downs + [self.mid] + self.ups:
            for t in blk.attentions:
                if block_name.startswith(t.name + "."):
                    return t.heads
        raise KeyError(block_name)

    # ------------------------------------------------------------------ forward
    def _ws(self, B: int) -> torch.Tensor:
        need = ops.groupnorm_ws_bytes(B, 2560)
        if self._gn_ws is None or self._gn_ws.numel() < need:
            self._gn_ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
        return self._gn_ws

    def _empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float16, device=self.device)

    def _ln(self, b: str, which: str):
        """(s, b', eps) of a LayerNorm folded into projection ``which`` of transformer block ``b`` (weights.fold_ln)"""
        return (self.W[f"{b}.{which}s"].view(torch.float32), self.W[f"{b}.{which}b"].view(torch.float32), ops.LN_EPS)

    def _gn(self, x1, c1, B, HW, g, b, eps, silu, x2=None, c2=0):
        out = self._empty(B * HW, c1 + c2)
        ops.groupnorm(x1, out, g, b, self._ws(B), B=B, HW=HW, c1=c1, x2=x2, c2=c2,
                      groups=self.config.norm_num_groups, eps=eps, silu=silu)
        return out

    def _resnet(self, r: ResnetSpec, x, skip, c_x, c_skip, B, H, Wd, temb_all, temb_rows, rep: int = 1):
        """``rep`` = 2: the output is wanted twice (the CFG batch's two halves share it so far) -- conv2 writes both copies
        (cid_gemm_desc.out2) and the result is [2 * B * HW, cout], its first half being the B-sample tensor"""
        W, n = self.W, r.name
        HW = H * Wd
        M = B * HW
        cin = c_x + c_skip
        assert cin == r.cin, (n, cin, r.cin)
        h = self._gn(x, c_x, B, HW, W[f"{n}.norm1.g"], W[f"{n}.norm1.b"], self.config.norm_eps, True, skip, c_skip)
        h1 = self._empty(M, r.cout)
        off = self.packed.temb_offsets[n]
        rps = HW if temb_rows > 1 else M   # shared timestep row vs per-sample rows (SDXL)
        ops.gemm(h, W[f"{n}.conv1.w"], h1, M=M, N=r.cout, c1=cin, bias=W[f"{n}.conv1.b"],
                 rowbias=temb_all[:, off:], ld_rowbias=self.packed.temb_total, rows_per_sample=rps,
                 taps=9, Hi=H, Wi=Wd, Ho=H, Wo=Wd, ws=self._gemm_ws, gn_hw=HW)      # (norm2 reads h1)
        h2 = self._gn(h1, r.cout, B, HW, W[f"{n}.norm2.g"], W[f"{n}.norm2.b"], self.config.norm_eps, True)
        if r.cin != r.cout:
            sc = self._empty(M, r.cout)
            ops.gemm(x, W[f"{n}.short.w"], sc, M=M, N=r.cout, c1=c_x, x2=skip, c2=c_skip, bias=W[f"{n}.short.b"],
                     ws=self._gemm_ws)
        else:
            assert skip is None
            sc = x
        full = self._empty(rep * M, r.cout)
        out = full[:M] if rep == 2 else full
        ops.gemm(h2, W[f"{n}.conv2.w"], out, M=M, N=r.cout, c1=r.cout, bias=W[f"{n}.conv2.b"], res=sc, ldr=r.cout,
                 taps=9, Hi=H, Wi=Wd, Ho=H, Wo=Wd, ws=self._gemm_ws, gn_hw=HW,      # (a GroupNorm reads every resnet output)
                 out2=full[M:] if rep == 2 else None)
        if rep == 2:
            out._cfg_full = full
        return out

    def _dup(self, t: torch.Tensor, rep: int) -> torch.Tensor:
        """[rows, c] -> [rep * rows, c], the CFG ``torch.cat([x] * 2)`` of a tensor both halves share"""
        out = self._empty(rep * t.shape[0], t.shape[1])
        out.view(rep, -1).copy_(t.reshape(1, -1).expand(rep, -1))
        return out

    def _transformer(self, t: TransformerSpec, x, B, H, Wd, kvrow, Bp: Optional[int] = None):
        """``Bp`` < B: the rows of ``x`` are the Bp distinct samples of a CFG batch B = rep * Bp whose halves were identical
        so far (same latents, same timestep).  Everything up to the first cross-attention -- the first place the two
        halves see different inputs -- runs once on Bp samples and is then repeated; the result is the same tensor the
        full batch would produce, row for row."""
        W, n, c = self.W, t.name, t.channels
        N = H * Wd
        d = c // t.heads
        ctx = self._ctx
        Bc = B if Bp is None else Bp          # batch of the part computed so far
        g = self._gn(x, c, Bc, N, W[f"{n}.norm.g"], W[f"{n}.norm.b"], 1e-6, False)
        # The attention kernels tile the token axis (64 keys per step; 128 / 64 tokens per workgroup in the fused
        # cross-attention).  512^2 and 1024^2 images give multiples at every level; other sizes run the block on a
        # zero-padded token axis: every other op of the block is row-wise, the self-attention masks the pad keys
        # (cid_self_attn_keys_f16), and the pad rows are dropped at the end.
        gens = {ctx.v2.get(f"{n}.transformer_blocks.{k}", 0) for k in range(t.n_layers)}
        need = 128 if (2 in gens or (c <= self._xattn_fused_max_c and c > 128 and not gens <= {3})) else 64
        N_real = N
        if N % need:
            N = (N + need - 1) // need * need

            def pad(tn):
                o = torch.zeros(tn.shape[0] // N_real, N, c, dtype=torch.float16, device=self.device)
                o[:, :N_real] = tn.view(-1, N_real, c)
                return o.view(-1, c)
            g, x = pad(g), pad(x)
        M = Bc * N
        h = self._empty(M, c)
        ops.gemm(g, W[f"{n}.proj_in.w"], h, M=M, N=c, c1=c, bias=W[f"{n}.proj_in.b"])
        for k in range(t.n_layers):
            b = f"{n}.transformer_blocks.{k}"
            # --- self attention (Consistent_AttProcessor, attention.py:110-174)
            qk = self._empty(M, 2 * c)
            vt = self._empty(Bc * t.heads * ops.dvp_of(d) * N)
            if ops.ln_fold(M):     # norm1 folded into the projection: the GEMM reads the residual stream itself
                ops.gemm(h, W[f"{b}.attn1.qkv.wl"], qk, M=M, N=3 * c, c1=c, mode=2, vt=vt, n_vt0=2 * c,
                         heads=t.heads, dhead=d, ntok=N, ln=self._ln(b, "attn1.qkv.ln"))
            else:
                ln = self._empty(M, c)
                ops.layernorm(h, ln, W[f"{b}.norm1.g"], W[f"{b}.norm1.b"], M=M, C_=c)
                ops.gemm(ln, W[f"{b}.attn1.qkv.w"], qk, M=M, N=3 * c, c1=c, mode=2, vt=vt, n_vt0=2 * c,
                         heads=t.heads, dhead=d, ntok=N)
            ao = self._empty(M, c)
            ops.self_attn(qk, qk[:, c:], vt, ao, B=Bc, N=N, heads=t.heads, d=d, ldq=2 * c, ldk=2 * c, ldo=c,
                          n_keys=N_real)
            rep = B // Bc
            if rep == 2:    # the halves part ways at the cross-attention: the out projection writes the shared stream twice
                h2 = self._empty(2 * M, c)
                ops.gemm(ao, W[f"{b}.attn1.out.w"], h2[:M], M=M, N=c, c1=c, bias=W[f"{b}.attn1.out.b"], res=h, ldr=c, out2=h2[M:])
            else:
                h2 = self._empty(M, c)
                ops.gemm(ao, W[f"{b}.attn1.out.w"], h2, M=M, N=c, c1=c, bias=W[f"{b}.attn1.out.b"], res=h, ldr=c)
            if Bc != B:     # ... and the block input (the residual of proj_out) exists twice already where its producer wrote it so
                if rep != 2:
                    h2 = self._dup(h2, rep)
                full = getattr(x, "_cfg_full", None)
                x = full if full is not None and full.shape[0] == rep * x.shape[0] else self._dup(x, rep)
                Bc, M = B, B * N
            # --- identity cross attention (Consistent_IPAttProcessor, attention.py:207-294) inside x + attn2(LN(x), ehs)
            h3 = self.cross_attention(b, h2, B, N, c, t.heads, kvrow)
            # --- feed forward (GEGLU)
            ff = self._empty(M, 4 * c)
            if ops.ln_fold_geglu(M, c):     # norm3 folded into the GEGLU projection
                ops.gemm(h3, W[f"{b}.ff1.wl"], ff, M=M, N=8 * c, c1=c, mode=1, ln=self._ln(b, "ff1.ln"))
            else:
                ln3 = self._empty(M, c)
                ops.layernorm(h3, ln3, W[f"{b}.norm3.g"], W[f"{b}.norm3.b"], M=M, C_=c)
                ops.gemm(ln3, W[f"{b}.ff1.w"], ff, M=M, N=8 * c, c1=c, bias=W[f"{b}.ff1.b"], mode=1)
            h = self._empty(M, c)
            ops.gemm(ff, W[f"{b}.ff2.w"], h, M=M, N=c, c1=4 * c, bias=W[f"{b}.ff2.b"], res=h3, ldr=c,
                     ws=self._gemm_ws)
        out = self._empty(M, c)
        ops.gemm(h, W[f"{n}.proj_out.w"], out, M=M, N=c, c1=c, bias=W[f"{n}.proj_out.b"], res=x, ldr=c,
                 gn_hw=N if N == N_real else 0)      # (the next resnet's norm1 / a skip consumer reads it)
        if N != N_real:
            out = out.view(-1, N, c)[:, :N_real].reshape(-1, c)      # reshape of a sliced view: one copy
        return out

    def cross_attention(self, b: str, h2: torch.Tensor, B: int, N: int, c: int, heads: int, kvrow: torch.Tensor) -> torch.Tensor:
        """``h2 + attn2(LayerNorm(h2), context)`` of transformer block ``b`` on token-major ``h2`` [B * N, c]: the launch
        sequence the denoise step uses for this layer (bench.py times exactly this for the roofline block).
          * SD1.5 level 0 (C = 320, 8 heads): ONE launch of the third-generation fused kernel (csrc/xattn3.hip):
            LayerNorm folded into Wq, x read from HBM once;
          * C <= CID_XATTN_FUSED_MAX_C otherwise: one launch of the first-generation fused kernel;
          * wider levels: LayerNorm + q GEMM + two-stream attention core + out GEMM (+ bias + residual) -- a
            [tokens x C] tile does not fit in LDS next to the weight slabs there."""
        W, ctx = self.W, self._ctx
        M = B * N
        h3 = self._empty(M, c)
        if ctx.v2.get(b) == 3:
            ops.id_xattn3(h2, h3, wq_p=W[f"{b}.attn2.wq_p"], q_rowsum=W[f"{b}.attn2.qs"].view(torch.float32),
                          q_bias=W[f"{b}.attn2.qb"].view(torch.float32), wo_p=W[f"{b}.attn2.wo_p"], bo=W[f"{b}.attn2.bo"],
                          kp=ctx.kp[b], vp=ctx.vp[b], kvrow=kvrow, B=B, N=N, C_=c, heads=heads, n_txt=ctx.n_txt,
                          n_ip=ctx.n_ip, ip_scale=self.packed.ip_scale[b], has_ln=True, add_residual=True, ln_eps=ops.LN_EPS)
        elif self._fused_gen1(c, B * N):
            ops.id_xattn(h2, h3, wq=W[f"{b}.attn2.wq"], wo=W[f"{b}.attn2.wo"], bo=W[f"{b}.attn2.bo"],
                         kp=ctx.kp[b], vp=ctx.vp[b], kvrow=kvrow, B=B, N=N, C_=c, heads=heads,
                         n_txt=ctx.n_txt, n_ip=ctx.n_ip, ip_scale=self.packed.ip_scale[b], residual=h2,
                         ln_gamma=W[f"{b}.norm2.g"], ln_beta=W[f"{b}.norm2.b"], ln_eps=ops.LN_EPS)
        elif self._qattn and ops.qattn_supported(c, heads, N, ctx.n_txt, ctx.n_ip):
            # wider levels: the (LayerNorm-folded) query projection runs the attention as its epilogue (cid_gemm_f16 mode 3:
            # tiles of whole heads, Q never leaves the CU), then the out projection (+ bias + residual) -- two launches
            o2 = self._empty(M, c)
            att = (ctx.kp[b], ctx.vp[b], kvrow, ctx.n_txt, ctx.n_ip, self.packed.ip_scale[b])
            if ops.ln_fold_q(M):
                ops.gemm(h2, W[f"{b}.attn2.wql"], o2, M=M, N=c, c1=c, mode=3, heads=heads, dhead=c // heads, ntok=N,
                         ln=self._ln(b, "attn2.wq_ln"), att=att)
            else:
                ln2 = self._empty(M, c)
                ops.layernorm(h2, ln2, W[f"{b}.norm2.g"], W[f"{b}.norm2.b"], M=M, C_=c)
                ops.gemm(ln2, W[f"{b}.attn2.wq"], o2, M=M, N=c, c1=c, mode=3, heads=heads, dhead=c // heads, ntok=N, att=att)
            ops.gemm(o2, W[f"{b}.attn2.wo"], h3, M=M, N=c, c1=c, bias=W[f"{b}.attn2.bo"], res=h2, ldr=c)
        else:
            q2 = self._empty(M, c)
            if ops.ln_fold(M):     # norm2 folded into the query projection
                ops.gemm(h2, W[f"{b}.attn2.wql"], q2, M=M, N=c, c1=c, ln=self._ln(b, "attn2.wq_ln"))
            else:
                ln2 = self._empty(M, c)
                ops.layernorm(h2, ln2, W[f"{b}.norm2.g"], W[f"{b}.norm2.b"], M=M, C_=c)
                ops.gemm(ln2, W[f"{b}.attn2.wq"], q2, M=M, N=c, c1=c)
            o2 = self._empty(M, c)
            ops.id_xattn_core(q2, o2, kp=ctx.kp[b], vp=ctx.vp[b], kvrow=kvrow, B=B, N=N, C_=c, heads=heads,
                              n_txt=ctx.n_txt, n_ip=ctx.n_ip, ip_scale=self.packed.ip_scale[b])
            ops.gemm(o2, W[f"{b}.attn2.wo"], h3, M=M, N=c, c1=c, bias=W[f"{b}.attn2.bo"], res=h2, ldr=c)
        return h3

    def _fused_gen1(self, c: int, tokens: int) -> bool:
        """one launch of the first-generation fused kernel instead of LN + GEMM + core + GEMM?  Measured per level
        (tools/xattn_levels.py, profiles/r02_xattn_levels.txt): always up to CID_XATTN_FUSED_MAX_C = 320 channels; at 640
        channels only with >= 16 k tokens in flight (SDXL's 64 x 64 level at CFG batch 4: 68.5 vs 78.2 us; SD1.5's
        32 x 32 level at CFG batch 8 has 8 k: 61.8 vs 53.3 us); never at 1280 (144-149 vs 54-67 us)."""
        return c <= self._xattn_fused_max_c or (c <= 640 and tokens >= 16384 and self._xattn_fused_max_c >= 320)

    def cross_attention_path(self, b: str, c: int, B: int, N: int) -> str:
        """the launch sequence :meth:`cross_attention` runs for block ``b`` on B samples of N tokens, as text (bench.py's
        roofline block, tools/xattn_levels.py) -- the same predicates on the same arguments as the method itself"""
        assert B > 0 and N > 0, "cross_attention_path: B samples of N tokens (the shape decides the launch sequence)"
        tokens = B * N
        if self._ctx.v2.get(b):
            gen = self._ctx.v2[b]
            return f"id_xattn{gen}_kernel<{self._ctx.n_txt},{self._ctx.n_ip}> (one launch)"
        if self._fused_gen1(c, tokens):
            return "id_xattn_kernel (one launch, first generation)"
        heads = self._heads_of(b)
        if self._qattn and ops.qattn_supported(c, heads, N, self._ctx.n_txt, self._ctx.n_ip):
            return ("q GEMM with attention epilogue + out GEMM (two launches)" if ops.ln_fold_q(tokens)
                    else "layernorm + q GEMM with attention epilogue + out GEMM (three launches)")
        return "layernorm + q GEMM + id_xattn core + out GEMM (four launches)"

    def time_embed(self, t_dev: torch.Tensor, B: int, added_cond_kwargs=None) -> torch.Tensor:
        """sinusoid -> MLP (-> + SDXL text_time embedding) -> all ResnetBlock2D.time_emb_proj
        at once.  Returns [rows, sum(Cout)] with rows = 1 (shared timestep) or B (SDXL)."""
        cfg, W = self.config, self.W
        c0, ted = cfg.block_out_channels[0], cfg.time_embed_dim
        sc = self._empty(1, c0)
        ops.sincos_embed(t_dev, sc, rows=1, dim=c0)
        e1 = self._empty(1, ted)
        ops.linear_small(sc, W["time_embedding.linear_1.w"], W["time_embedding.linear_1.b"], e1,
                         M=1, N=ted, K=c0, act_out=1)
        emb = self._empty(1, ted)
        ops.linear_small(e1, W["time_embedding.linear_2.w"], W["time_embedding.linear_2.b"], emb, M=1, N=ted, K=ted)
        rows = 1
        if cfg.addition_embed_type == "text_time":
            text_embeds = added_cond_kwargs["text_embeds"].to(device=self.device, dtype=torch.float16).contiguous()
            time_ids = added_cond_kwargs["time_ids"].to(device=self.device, dtype=torch.float32).contiguous()
            assert text_embeds.shape[0] == B and time_ids.shape[0] == B
            ad = cfg.addition_time_embed_dim
            nid = time_ids.shape[1]
            pdim = text_embeds.shape[1] + nid * ad
            assert pdim == cfg.projection_class_embeddings_input_dim
            cat = self._empty(B, pdim)
            cat[:, :text_embeds.shape[1]].copy_(text_embeds)
            tid = self._empty(B * nid, ad)
            ops.sincos_embed(time_ids.reshape(-1), tid, rows=B * nid, dim=ad)
            cat[:, text_embeds.shape[1]:].copy_(tid.reshape(B, nid * ad))
            a1 = self._empty(B, ted)
            ops.linear_small(cat, W["add_embedding.linear_1.w"], W["add_embedding.linear_1.b"], a1,
                             M=B, N=ted, K=pdim, act_out=1)
            embB = self._empty(B, ted)
            ops.linear_small(a1, W["add_embedding.linear_2.w"], W["add_embedding.linear_2.b"], embB,
                             M=B, N=ted, K=ted, add=emb, ldadd=0)
            emb, rows = embB, B
        out = self._empty(rows, self.packed.temb_total)
        ops.linear_small(emb, W["temb_all.w"], W["temb_all.b"], out, M=rows, N=self.packed.temb_total, K=ted, act_in=1)
        return out

    @torch.no_grad()
    def time_embed_table(self, tvals: torch.Tensor) -> torch.Tensor:
        """The time path of EVERY step of a generation at once: [S] timesteps -> [S, sum(Cout)].  Without SDXL's
        text_time conditioning the per-resnet time rows depend on the timestep only, so the three GEMVs that would
        otherwise re-read 50 MB of time-projection weights in every step run once per generation; a step then takes
        its row from the table (``forward_tokens(temb=...)``)."""
        cfg, W = self.config, self.W
        assert cfg.addition_embed_type is None
        c0, ted = cfg.block_out_channels[0], cfg.time_embed_dim
        tv = tvals.to(device=self.device, dtype=torch.float32).contiguous()
        S = tv.numel()
        out = self._empty(S, self.packed.temb_total)
        for lo in range(0, S, 64):                              # cid_linear_small_f16 takes up to 64 rows
            n = min(64, S - lo)
            sc, e1, emb = self._empty(n, c0), self._empty(n, ted), self._empty(n, ted)
            ops.sincos_embed(tv[lo:lo + n], sc, rows=n, dim=c0)
            ops.linear_small(sc, W["time_embedding.linear_1.w"], W["time_embedding.linear_1.b"], e1,
                             M=n, N=ted, K=c0, act_out=1)
            ops.linear_small(e1, W["time_embedding.linear_2.w"], W["time_embedding.linear_2.b"], emb, M=n, N=ted, K=ted)
            ops.linear_small(emb, W["temb_all.w"], W["temb_all.b"], out[lo:lo + n], M=n, N=self.packed.temb_total,
                             K=ted, act_in=1)
        return out

    @torch.no_grad()
    def forward_tokens(self, sample: torch.Tensor, t_dev: torch.Tensor, kvrow: torch.Tensor, B: int,
                       added_cond_kwargs=None, down_residuals: Optional[Sequence[torch.Tensor]] = None,
                       mid_residual: Optional[torch.Tensor] = None, temb: Optional[torch.Tensor] = None,
                       in_scale: Optional[torch.Tensor] = None, extra: Optional[torch.Tensor] = None) -> torch.Tensor:
        """sample: NCHW fp16 [Bin, cin, H, W] with B % Bin == 0 (batch row b reads sample b % Bin,
        i.e. the CFG duplication of ref :537-539 costs no copy).  Returns NCHW fp16 [B, cout, H, W].
        ``extra`` [Bin, cin2, H, W]: the trailing input channels of a 9-channel inpainting UNet (cat([mask,
        masked_image_latents]), inpaint ref :320-321), read by conv_in beside the latents instead of a concatenated copy
        and not touched by ``in_scale``."""
        cfg, W = self.config, self.W
        Bin, cin, H, Wd = sample.shape
        if extra is not None:
            cin += extra.shape[1]
        assert B % Bin == 0 and cin == cfg.in_channels, (B, Bin, cin, cfg.in_channels)
        if temb is None:
            temb = self.time_embed(t_dev, B, added_cond_kwargs)
        trows = temb.shape[0]
        c0 = cfg.block_out_channels[0]
        # CFG batches arrive as B = rep * Bin with batch row b reading latent b % Bin: until the first cross-attention the
        # rep copies are the same computation on the same data (same latents, same timestep row), so conv_in, the first
        # ResnetBlock2D and the first transformer's GroupNorm / proj_in / self-attention run on the Bin distinct samples
        # only and are repeated where the halves start to differ (their encoder_hidden_states).  Row-for-row the same
        # tensors as the full batch; CID_CFG_DEDUP=0 disables it.
        Bp = B
        if (self._cfg_dedup and B > Bin and trows == 1 and self.downs and self.downs[0].attentions
                and self.downs[0].attentions[0].n_layers >= 1):
            Bp = Bin
        # (conv_in reads latent b % Bin for batch row b: run on all B rows it writes the duplicated skip tensor itself, and the
        #  first Bp samples of that buffer are the deduplicated stream)
        xfull = self._empty(B * H * Wd, c0)
        ops.conv_in(sample, xfull, W["conv_in.w"], W["conv_in.b"], B=B, Bin=Bin, cin=cin, H=H, W=Wd, cout=c0, in_scale=in_scale,
                    extra=extra)
        x = xfull[:Bp * H * Wd]
        skips = [(xfull, c0, H, Wd)]
        c = c0
        for bi, blk in enumerate(self.downs):
            for j, r in enumerate(blk.resnets):
                first = Bp != B and bi == 0 and j == 0
                x = self._resnet(r, x, None, c, 0, Bp if first else B, H, Wd, temb, trows, rep=2 if first and B == 2 * Bp else 1)
                c = r.cout
                if blk.attentions:
                    x = self._transformer(blk.attentions[j], x, B, H, Wd, kvrow, Bp=Bp if first else None)
                skips.append((x, c, H, Wd))
            if blk.sampler:
                n = f"{blk.name}.{blk.sampler}.conv"
                Ho, Wo = H // 2, Wd // 2
                y = self._empty(B * Ho * Wo, c)
                ops.gemm(x, W[f"{n}.w"], y, M=B * Ho * Wo, N=c, c1=c, bias=W[f"{n}.b"], taps=9,
                         Hi=H, Wi=Wd, Ho=Ho, Wo=Wo, stride=2, ws=self._gemm_ws, gn_hw=Ho * Wo)
                x, H, Wd = y, Ho, Wo
                skips.append((x, c, H, Wd))
        if down_residuals is not None:
            # ControlNet residuals (CN :418-425), given token-major [B or B/2, HW, C]
            assert len(down_residuals) == len(skips)
            # The down path is complete: every skip tensor has been read by whatever followed it there, the up blocks are the
            # only readers left -- the residual is added IN PLACE.  Two exceptions get a copy: the last skip is also the
            # mid block's input (which takes no down residual, CN :418-425 add to the skip list only), and a tensor that
            # sits in the list twice (none today) must not be added to twice.
            new, seen = [], set()
            for (s, sc_, sh, sw), r in zip(skips, down_residuals):
                if s is x or s.data_ptr() in seen:
                    s = s.clone()
                seen.add(s.data_ptr())
                ops.add_inplace(s, r.to(device=self.device, dtype=torch.float16).contiguous())
                new.append((s, sc_, sh, sw))
            skips = new
        x = self._resnet(self.mid.resnets[0], x, None, c, 0, B, H, Wd, temb, trows)
        x = self._transformer(self.mid.attentions[0], x, B, H, Wd, kvrow)
        x = self._resnet(self.mid.resnets[1], x, None, c, 0, B, H, Wd, temb, trows)
        if mid_residual is not None:     # (x is the mid block's own output here: nobody else holds it)
            ops.add_inplace(x, mid_residual.to(device=self.device, dtype=torch.float16).contiguous())
        for blk in self.ups:
            for j, r in enumerate(blk.resnets):
                s, sc_, sh, sw = skips.pop()
                assert (sh, sw) == (H, Wd)
                x = self._resnet(r, x, s, c, sc_, B, H, Wd, temb, trows)
                c = r.cout
                if blk.attentions:
                    x = self._transformer(blk.attentions[j], x, B, H, Wd, kvrow)
            if blk.sampler:
                n = f"{blk.name}.{blk.sampler}.conv"
                Ho, Wo = H * 2, Wd * 2
                y = self._empty(B * Ho * Wo, c)
                ops.gemm(x, W[f"{n}.w"], y, M=B * Ho * Wo, N=c, c1=c, bias=W[f"{n}.b"], taps=9,
                         Hi=H, Wi=Wd, Ho=Ho, Wo=Wo, stride=1, up=1, ws=self._gemm_ws, gn_hw=Ho * Wo)
                x, H, Wd = y, Ho, Wo
        g = self._gn(x, c, B, H * Wd, W["conv_norm_out.g"], W["conv_norm_out.b"], cfg.norm_eps, True)
        out = self._empty(B, cfg.out_channels, H, Wd)
        ops.conv_out(g, out, W["conv_out.w"], W["conv_out.b"], B=B, H=H, W=Wd, cin=c, cout=cfg.out_channels)
        return out

    # ------------------------------------------------------------------ diffusers-style call
    @torch.no_grad()
    def __call__(self, sample, timestep, encoder_hidden_states=None, cross_attention_kwargs=None,
                 added_cond_kwargs=None, down_block_additional_residuals=None,
                 mid_block_additional_residual=None, return_dict=True):
        sample = sample.to(device=self.device, dtype=torch.float16).contiguous()
        B = sample.shape[0]
        ehs = encoder_hidden_states
        key = (ehs.data_ptr(), ehs._version, tuple(ehs.shape))
        if self._ctx.key != key or self._ctx.key_ref is not ehs:
            self.set_context(ehs)
            self._ctx.key, self._ctx.key_ref = key, ehs
        kvrow = torch.arange(B, dtype=torch.int32, device=self.device)
        self._t_buf.fill_(float(timestep))

        def tok(r):  # NCHW -> token-major
            r = r.to(device=self.device, dtype=torch.float16)
            return r.permute(0, 2, 3, 1).reshape(r.shape[0], -1, r.shape[1]).contiguous()

        dres = [tok(r) for r in down_block_additional_residuals] if down_block_additional_residuals is not None else None
        mres = tok(mid_block_additional_residual) if mid_block_additional_residual is not None else None
        out = self.forward_tokens(sample, self._t_buf, kvrow, B, added_cond_kwargs, dres, mres)
        return UNetOutput(sample=out) if return_dict else (out,)
