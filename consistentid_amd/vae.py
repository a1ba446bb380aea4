"""HIP execution engine for the VAE decoder.

Replaces what the reference reaches through ``self.decode_latents(latents)``
(pipline_StableDiffusion_ConsistentID.py:587,:597 -> diffusers ``StableDiffusionPipeline.decode_latents``:
``latents / scaling_factor`` -> ``vae.decode`` -> ``(image / 2 + 0.5).clamp(0, 1)``).

Same design as the UNet engine: activations token-major fp16 ``[B, H*W, C]`` from the first conv to the last, every
op a kernel behind the C ABI (3x3 convs / linears / nearest-2x upsample fused into the following conv:
``cid_gemm_f16``; GroupNorm(+SiLU): ``cid_groupnorm_f16``).  The mid block's attention is ONE head of width C over
all latent pixels -- too wide for a register-resident flash kernel, and run once per image -- so it is four GEMMs
around a row softmax: S = Q K^T (scores, base-2 logits), softmax, V^T by a GEMM with the operand roles swapped,
O = P V.  Weight preparation at load: the 1/scaling_factor latent scale is folded into ``post_quant_conv``, the
attention scale and log2(e) into ``to_q``, the K bias is dropped (softmax is shift-invariant per row) and the V bias
moves through the softmax (rows of P sum to 1) into the output projection's bias.

SDXL's VAE sets ``force_upcast``: the reference decodes it in float32 (``upcast_vae()``,
pipline_StableDiffusionXL_ConsistentID.py:670-676).  ``HipVAEDecoderF32`` below is that decoder on the fp32 kernels
(``cid_gemm_f32`` / ``cid_groupnorm_f32`` / ``cid_softmax_rows_f32``, csrc/f32.hip); ``make_vae_decoder`` picks the
engine from the config like the reference's ``needs_upcasting`` test does.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from .vae_spec import VAEConfig, decoder_blocks
from .weights import LOG2E, _conv3, _f, _h


class HipVAEDecoder:
    def __init__(self, cfg: VAEConfig, vae_sd: Dict[str, torch.Tensor], device="cuda:0"):
        if cfg.force_upcast:
            raise NotImplementedError("force_upcast VAEs (SDXL) decode in fp32 in the reference: use HipVAEDecoderF32 "
                                      "(make_vae_decoder picks it)")
        self.config = cfg
        self.device = torch.device(device)
        self.dtype = torch.float16
        dev, sd = self.device, vae_sd
        W: Dict[str, torch.Tensor] = {}
        self.W = W
        L, boc = cfg.latent_channels, cfg.block_out_channels
        c_mid = boc[-1]
        with torch.cuda.device(self.device):
            # post_quant_conv (1x1, L -> L) with the latent scale folded in, carried as the centre tap of a 3x3 conv
            # whose output is padded to 8 channels (the small-conv kernel's granule); conv_in reads those 8 channels
            pq = _f(sd["post_quant_conv.weight"], dev).reshape(L, L) / cfg.scaling_factor
            w = torch.zeros(8, 9, L, device=dev)
            w[:L, 4, :] = pq
            W["pq.w"] = _h(w.reshape(8, 9 * L), dev)
            b = torch.zeros(8, device=dev)
            b[:L] = _f(sd["post_quant_conv.bias"], dev)
            W["pq.b"] = _h(b, dev)
            ci = _f(sd["decoder.conv_in.weight"], dev)                         # [C, L, 3, 3]
            w = torch.zeros(c_mid, 9, 8, device=dev)
            w[:, :, :L] = ci.permute(0, 2, 3, 1).reshape(c_mid, 9, L)
            W["conv_in.w"] = _h(w.reshape(c_mid, 72), dev)
            W["conv_in.b"] = _h(sd["decoder.conv_in.bias"], dev)

            def resnet(n):
                for k in ("norm1", "norm2"):
                    W[f"{n}.{k}.g"], W[f"{n}.{k}.b"] = _h(sd[f"{n}.{k}.weight"], dev), _h(sd[f"{n}.{k}.bias"], dev)
                for k in ("conv1", "conv2"):
                    W[f"{n}.{k}.w"], W[f"{n}.{k}.b"] = _conv3(sd[f"{n}.{k}.weight"], dev), _h(sd[f"{n}.{k}.bias"], dev)
                if f"{n}.conv_shortcut.weight" in sd:
                    sw = sd[f"{n}.conv_shortcut.weight"]
                    W[f"{n}.short.w"] = _h(sw.reshape(sw.shape[0], sw.shape[1]), dev)
                    W[f"{n}.short.b"] = _h(sd[f"{n}.conv_shortcut.bias"], dev)

            m = "decoder.mid_block"
            resnet(f"{m}.resnets.0")
            resnet(f"{m}.resnets.1")
            a = f"{m}.attentions.0"
            W["attn.gn.g"], W["attn.gn.b"] = _h(sd[f"{a}.group_norm.weight"], dev), _h(sd[f"{a}.group_norm.bias"], dev)
            qs = (c_mid ** -0.5) * LOG2E
            W["attn.q.w"] = _h(_f(sd[f"{a}.to_q.weight"], dev) * qs, dev)
            W["attn.q.b"] = _h(_f(sd[f"{a}.to_q.bias"], dev) * qs, dev)
            W["attn.k.w"] = _h(sd[f"{a}.to_k.weight"], dev)                     # bias dropped: constant per score row
            W["attn.v.w"] = _h(sd[f"{a}.to_v.weight"], dev)
            wo = _f(sd[f"{a}.to_out.0.weight"], dev)
            W["attn.o.w"] = _h(wo, dev)
            W["attn.o.b"] = _h(_f(sd[f"{a}.to_out.0.bias"], dev) + wo @ _f(sd[f"{a}.to_v.bias"], dev), dev)
            self.blocks = decoder_blocks(cfg)
            for name, cin, cout, n, up in self.blocks:
                for j in range(n):
                    resnet(f"{name}.resnets.{j}")
                if up:
                    u = f"{name}.upsamplers.0.conv"
                    W[f"{u}.w"], W[f"{u}.b"] = _conv3(sd[f"{u}.weight"], dev), _h(sd[f"{u}.bias"], dev)
            W["norm_out.g"], W["norm_out.b"] = _h(sd["decoder.conv_norm_out.weight"], dev), _h(sd["decoder.conv_norm_out.bias"], dev)
            co = sd["decoder.conv_out.weight"]
            W["conv_out.w"] = _h(co.permute(0, 2, 3, 1).reshape(co.shape[0], -1), dev)
            W["conv_out.b"] = _h(sd["decoder.conv_out.bias"], dev)
        self._gn_ws: Optional[torch.Tensor] = None
        self._gemm_ws = torch.empty(64 << 20, dtype=torch.uint8, device=self.device)

    # ------------------------------------------------------------------ helpers
    def _empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float16, device=self.device)

    def _gn(self, x, c, B, HW, g, b, silu):
        need = ops.groupnorm_ws_bytes(B, c)
        if self._gn_ws is None or self._gn_ws.numel() < need:
            self._gn_ws = torch.zeros(need, dtype=torch.uint8, device=self.device)   # arrival counters start at zero
        out = self._empty(B * HW, c)
        ops.groupnorm(x, out, g, b, self._gn_ws, B=B, HW=HW, c1=c, groups=self.config.norm_num_groups, eps=1e-6, silu=silu)
        return out

    def _conv(self, x, n, cin, cout, B, H, Wd, res=None, up=0):
        Ho, Wo = H << up, Wd << up
        out = self._empty(B * Ho * Wo, cout)
        ops.gemm(x, self.W[f"{n}.w"], out, M=B * Ho * Wo, N=cout, c1=cin, bias=self.W[f"{n}.b"], res=res,
                 ldr=cout if res is not None else 0, taps=9, Hi=H, Wi=Wd, Ho=Ho, Wo=Wo, stride=1, up=up, ws=self._gemm_ws)
        return out

    def _resnet(self, n, x, cin, cout, B, H, Wd):
        W, HW = self.W, H * Wd
        h = self._gn(x, cin, B, HW, W[f"{n}.norm1.g"], W[f"{n}.norm1.b"], True)
        h = self._conv(h, f"{n}.conv1", cin, cout, B, H, Wd)
        h = self._gn(h, cout, B, HW, W[f"{n}.norm2.g"], W[f"{n}.norm2.b"], True)
        if cin != cout:
            sc = self._empty(B * HW, cout)
            ops.gemm(x, W[f"{n}.short.w"], sc, M=B * HW, N=cout, c1=cin, bias=W[f"{n}.short.b"], ws=self._gemm_ws)
        else:
            sc = x
        return self._conv(h, f"{n}.conv2", cout, cout, B, H, Wd, res=sc)

    def _attention(self, x, c, B, HW):
        """single-head attention over all HW positions of each sample (diffusers Attention, residual_connection)"""
        W = self.W
        M = B * HW
        t = self._gn(x, c, B, HW, W["attn.gn.g"], W["attn.gn.b"], False)
        q, k = self._empty(M, c), self._empty(M, c)
        ops.gemm(t, W["attn.q.w"], q, M=M, N=c, c1=c, bias=W["attn.q.b"])
        ops.gemm(t, W["attn.k.w"], k, M=M, N=c, c1=c)
        o = self._empty(M, c)
        s = self._empty(HW, HW)
        vt = self._empty(c, HW)
        for b in range(B):
            rows = slice(b * HW, (b + 1) * HW)
            ops.gemm(q[rows], k[rows], s, M=HW, N=HW, c1=c)                      # S = Q K^T      [HW, HW]
            ops.softmax_rows(s, rows=HW, cols=HW, ld=HW)
            ops.gemm(W["attn.v.w"], t[rows], vt, M=c, N=HW, c1=c)                # V^T = Wv T^T   [C, HW]
            ops.gemm(s, vt, o[rows], M=HW, N=c, c1=HW, ws=self._gemm_ws)         # O = P V        [HW, C]
        out = self._empty(M, c)
        ops.gemm(o, W["attn.o.w"], out, M=M, N=c, c1=c, bias=W["attn.o.b"], res=x, ldr=c)
        return out

    # ------------------------------------------------------------------ decode
    @torch.no_grad()
    def decode_tokens(self, latents: torch.Tensor):
        """latents [B, L, h, w] (UNSCALED, as the denoise loop leaves them) -> image token-major is internal; returns
        the decoded image [B, 3, 8h, 8w] fp16 NCHW in the VAE's [-1, 1] range."""
        cfg, W = self.config, self.W
        lat = latents.to(device=self.device, dtype=torch.float16)
        B, L, H, Wd = lat.shape
        assert L == cfg.latent_channels
        x = lat.permute(0, 2, 3, 1).reshape(B * H * Wd, L).contiguous()        # 32 KB per image: host-side plumbing
        z = self._empty(B * H * Wd, 8)
        ops.conv3x3_small(x, z, W["pq.w"], W["pq.b"], B=B, Hi=H, Wi=Wd, cin=L, cout=8)
        c = cfg.block_out_channels[-1]
        x = self._empty(B * H * Wd, c)
        ops.conv3x3_small(z, x, W["conv_in.w"], W["conv_in.b"], B=B, Hi=H, Wi=Wd, cin=8, cout=c)
        m = "decoder.mid_block"
        x = self._resnet(f"{m}.resnets.0", x, c, c, B, H, Wd)
        x = self._attention(x, c, B, H * Wd)
        x = self._resnet(f"{m}.resnets.1", x, c, c, B, H, Wd)
        for name, cin, cout, n, up in self.blocks:
            for j in range(n):
                x = self._resnet(f"{name}.resnets.{j}", x, cin if j == 0 else cout, cout, B, H, Wd)
            if up:
                x = self._conv(x, f"{name}.upsamplers.0.conv", cout, cout, B, H, Wd, up=1)
                H, Wd = 2 * H, 2 * Wd
            c = cout
        g = self._gn(x, c, B, H * Wd, W["norm_out.g"], W["norm_out.b"], True)
        out = self._empty(B, cfg.out_channels, H, Wd)
        ops.conv_out(g, out, W["conv_out.w"], W["conv_out.b"], B=B, H=H, W=Wd, cin=c, cout=cfg.out_channels)
        return out

    def decode(self, z: torch.Tensor, return_dict: bool = False):
        """diffusers ``vae.decode(z)`` protocol: ``z`` is ALREADY divided by scaling_factor by the caller
        (pipline_StableDiffusionXL_ConsistentID.py:676); the fold in post_quant_conv is undone here."""
        if return_dict:
            raise NotImplementedError("return_dict=True (the reference passes return_dict=False)")
        return (self.decode_tokens(z * self.config.scaling_factor),)

    def decode_latents(self, latents: torch.Tensor) -> torch.Tensor:
        """``StableDiffusionPipeline.decode_latents`` up to the device tensor: [B, 3, H, W] fp16 in [0, 1]."""
        return (self.decode_tokens(latents) / 2 + 0.5).clamp(0, 1)



class HipVAEDecoderF32:
    """The same decoder in float32, for VAEs with ``force_upcast`` (SDXL): what the reference runs after
    ``upcast_vae()`` (pipline_StableDiffusionXL_ConsistentID.py:670-676).  Token-major fp32 activations, every op one of
    the three fp32 kernels of csrc/f32.hip; the weight folds (latent scale into post_quant_conv, attention scale and
    log2 e into to_q, K bias dropped, V bias through the softmax into the output bias) are the fp16 engine's."""

    def __init__(self, cfg: VAEConfig, vae_sd: Dict[str, torch.Tensor], device="cuda:0"):
        self.config = cfg
        self.device = torch.device(device)
        self.dtype = torch.float32
        dev, sd = self.device, vae_sd
        W: Dict[str, torch.Tensor] = {}
        self.W = W
        L, c_mid = cfg.latent_channels, cfg.block_out_channels[-1]
        f = lambda t: _f(t, dev).contiguous()
        conv = lambda w: f(w).permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()       # [Cout, 9 * Cin], tap-major
        with torch.cuda.device(self.device):
            W["pq.w"] = (f(sd["post_quant_conv.weight"]).reshape(L, L) / cfg.scaling_factor).contiguous()
            W["pq.b"] = f(sd["post_quant_conv.bias"])
            W["conv_in.w"], W["conv_in.b"] = conv(sd["decoder.conv_in.weight"]), f(sd["decoder.conv_in.bias"])

            def resnet(n):
                for k in ("norm1", "norm2"):
                    W[f"{n}.{k}.g"], W[f"{n}.{k}.b"] = f(sd[f"{n}.{k}.weight"]), f(sd[f"{n}.{k}.bias"])
                for k in ("conv1", "conv2"):
                    W[f"{n}.{k}.w"], W[f"{n}.{k}.b"] = conv(sd[f"{n}.{k}.weight"]), f(sd[f"{n}.{k}.bias"])
                if f"{n}.conv_shortcut.weight" in sd:
                    sw = sd[f"{n}.conv_shortcut.weight"]
                    W[f"{n}.short.w"], W[f"{n}.short.b"] = f(sw).reshape(sw.shape[0], sw.shape[1]).contiguous(), f(sd[f"{n}.conv_shortcut.bias"])

            m = "decoder.mid_block"
            resnet(f"{m}.resnets.0")
            resnet(f"{m}.resnets.1")
            a = f"{m}.attentions.0"
            W["attn.gn.g"], W["attn.gn.b"] = f(sd[f"{a}.group_norm.weight"]), f(sd[f"{a}.group_norm.bias"])
            qs = (c_mid ** -0.5) * LOG2E
            W["attn.q.w"], W["attn.q.b"] = (f(sd[f"{a}.to_q.weight"]) * qs).contiguous(), (f(sd[f"{a}.to_q.bias"]) * qs).contiguous()
            W["attn.k.w"], W["attn.v.w"] = f(sd[f"{a}.to_k.weight"]), f(sd[f"{a}.to_v.weight"])
            wo = f(sd[f"{a}.to_out.0.weight"])
            W["attn.o.w"] = wo
            W["attn.o.b"] = (f(sd[f"{a}.to_out.0.bias"]) + wo @ f(sd[f"{a}.to_v.bias"])).contiguous()
            self.blocks = decoder_blocks(cfg)
            for name, cin, cout, n, up in self.blocks:
                for j in range(n):
                    resnet(f"{name}.resnets.{j}")
                if up:
                    u = f"{name}.upsamplers.0.conv"
                    W[f"{u}.w"], W[f"{u}.b"] = conv(sd[f"{u}.weight"]), f(sd[f"{u}.bias"])
            W["norm_out.g"], W["norm_out.b"] = f(sd["decoder.conv_norm_out.weight"]), f(sd["decoder.conv_norm_out.bias"])
            W["conv_out.w"], W["conv_out.b"] = conv(sd["decoder.conv_out.weight"]), f(sd["decoder.conv_out.bias"])
        self._gn_ws: Optional[torch.Tensor] = None

    def _empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def _gn(self, x, c, B, HW, g, b, silu):
        need = ops.groupnorm_f32_ws_bytes(B, HW, c)
        if self._gn_ws is None or self._gn_ws.numel() < need:
            self._gn_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = self._empty(B * HW, c)
        ops.groupnorm_f32(x, out, g, b, self._gn_ws, B=B, HW=HW, C_=c, groups=self.config.norm_num_groups, eps=1e-6, silu=silu)
        return out

    def _conv(self, x, n, cin, cout, B, H, Wd, res=None, up=0):
        out = self._empty(B * (H << up) * (Wd << up), cout)
        ops.gemm_f32(x, self.W[f"{n}.w"], out, M=out.shape[0], N=cout, c=cin, bias=self.W[f"{n}.b"], res=res, taps=9, Hi=H, Wi=Wd, up=up)
        return out

    def _linear(self, x, w, b, M, N, c, res=None):
        out = self._empty(M, N)
        ops.gemm_f32(x, w, out, M=M, N=N, c=c, bias=b, res=res)
        return out

    def _resnet(self, n, x, cin, cout, B, H, Wd):
        W, HW = self.W, H * Wd
        h = self._gn(x, cin, B, HW, W[f"{n}.norm1.g"], W[f"{n}.norm1.b"], True)
        h = self._conv(h, f"{n}.conv1", cin, cout, B, H, Wd)
        h = self._gn(h, cout, B, HW, W[f"{n}.norm2.g"], W[f"{n}.norm2.b"], True)
        sc = x if cin == cout else self._linear(x, W[f"{n}.short.w"], W[f"{n}.short.b"], B * HW, cout, cin)
        return self._conv(h, f"{n}.conv2", cout, cout, B, H, Wd, res=sc)

    def _attention(self, x, c, B, HW):
        W, M = self.W, B * HW
        t = self._gn(x, c, B, HW, W["attn.gn.g"], W["attn.gn.b"], False)
        q = self._linear(t, W["attn.q.w"], W["attn.q.b"], M, c, c)
        k = self._linear(t, W["attn.k.w"], None, M, c, c)
        o = self._empty(M, c)
        s, vt = self._empty(HW, HW), self._empty(c, HW)
        for b in range(B):
            rows = slice(b * HW, (b + 1) * HW)
            ops.gemm_f32(q[rows], k[rows], s, M=HW, N=HW, c=c)                   # S = Q K^T    [HW, HW]
            ops.softmax_rows_f32(s, rows=HW, cols=HW, ld=HW)
            ops.gemm_f32(W["attn.v.w"], t[rows], vt, M=c, N=HW, c=c)             # V^T = Wv T^T [C, HW]
            ops.gemm_f32(s, vt, o[rows], M=HW, N=c, c=HW)                        # O = P V      [HW, C]
        return self._linear(o, W["attn.o.w"], W["attn.o.b"], M, c, c, res=x)

    @torch.no_grad()
    def decode_tokens(self, latents: torch.Tensor):
        """latents [B, L, h, w] (UNSCALED) -> decoded image [B, 3, 8h, 8w] fp32 NCHW in the VAE's [-1, 1] range"""
        cfg, W = self.config, self.W
        lat = latents.to(device=self.device, dtype=torch.float32)
        B, L, H, Wd = lat.shape
        assert L == cfg.latent_channels
        x = lat.permute(0, 2, 3, 1).reshape(B * H * Wd, L).contiguous()
        z = self._linear(x, W["pq.w"], W["pq.b"], B * H * Wd, L, L)
        c = cfg.block_out_channels[-1]
        x = self._conv(z, "conv_in", L, c, B, H, Wd)
        m = "decoder.mid_block"
        x = self._resnet(f"{m}.resnets.0", x, c, c, B, H, Wd)
        x = self._attention(x, c, B, H * Wd)
        x = self._resnet(f"{m}.resnets.1", x, c, c, B, H, Wd)
        for name, cin, cout, n, up in self.blocks:
            for j in range(n):
                x = self._resnet(f"{name}.resnets.{j}", x, cin if j == 0 else cout, cout, B, H, Wd)
            if up:
                x = self._conv(x, f"{name}.upsamplers.0.conv", cout, cout, B, H, Wd, up=1)
                H, Wd = 2 * H, 2 * Wd
            c = cout
        g = self._gn(x, c, B, H * Wd, W["norm_out.g"], W["norm_out.b"], True)
        img = self._conv(g, "conv_out", c, cfg.out_channels, B, H, Wd)              # [B * H * W, 3]
        return img.view(B, H, Wd, cfg.out_channels).permute(0, 3, 1, 2).contiguous()   # NCHW for the caller (plumbing)

    def decode(self, z: torch.Tensor, return_dict: bool = False):
        if return_dict:
            raise NotImplementedError("return_dict=True (the reference passes return_dict=False)")
        return (self.decode_tokens(z * self.config.scaling_factor),)

    def decode_latents(self, latents: torch.Tensor) -> torch.Tensor:
        return (self.decode_tokens(latents) / 2 + 0.5).clamp(0, 1)


def make_vae_decoder(cfg: VAEConfig, vae_sd: Dict[str, torch.Tensor], device="cuda:0"):
    """fp32 engine for ``force_upcast`` VAEs (the reference's ``needs_upcasting`` test, SDXL :670), fp16 engine otherwise"""
    return (HipVAEDecoderF32 if cfg.force_upcast else HipVAEDecoder)(cfg, vae_sd, device)
