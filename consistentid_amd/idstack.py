"""HIP execution engine for the identity-conditioning modules (once per image).

Replaces ``self.image_proj_model`` (ProjPlusModel, functions.py:490-522) and ``self.FacialEncoder`` (attention.py:72-88)
as the reference calls them in ``get_image_embeds`` / ``get_facial_embeds`` (pipline_StableDiffusion_ConsistentID.py:176-209)
and assembles ``prompt_embeds = cat([null, augmented, text_only])`` the way ``__call__`` does (:479-507) -- the tensor the
denoising loop ``.chunk(3)``s.  Inputs are what the encoders upstream produce: CLIP-ViT-H penultimate hidden states of the
face image / the facial crops / zero images, the 512-d FaceID vector, the text encoder outputs and the trigger-token masks.
(Those encoders -- CLIP vision and text towers, insightface, BiSeNet -- are still outside this repository.)

Every Linear is ``cid_gemm_f16``, LayerNorm ``cid_layernorm_f16``; GELU and the tiny latent-query attention are
``cid_gelu_f16`` / ``cid_small_attn_f16``.  Row gathers / scatters of the handful of trigger-token rows are index
plumbing done with torch on the device.  State-dict keys are the reference's (``state_dict["image_proj"]`` and
``state_dict["FacialEncoder"]`` of the checkpoint, see checkpoint.py).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops


def _h(t: torch.Tensor, dev) -> torch.Tensor:
    return t.to(device=dev, dtype=torch.float16).contiguous()


class _Mod:
    """weights of one reference module under a key prefix, fp16 on the device"""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str, dev):
        self.dev = dev
        self.w = {k[len(prefix):]: _h(v, dev) for k, v in sd.items() if k.startswith(prefix)}

    def __getitem__(self, k):
        return self.w[k]

    def has(self, k):
        return k in self.w


class _Engine:
    def __init__(self, device):
        self.device = torch.device(device)

    def _empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float16, device=self.device)

    def linear(self, x, w, b=None, res=None):
        M, K = x.shape
        N = w.shape[0]
        out = self._empty(M, N)
        ops.gemm(x, w, out, M=M, N=N, c1=K, bias=b, res=res, ldr=N if res is not None else 0)
        return out

    def layernorm(self, x, g, b):
        out = torch.empty_like(x)
        ops.layernorm(x, out, g, b, M=x.shape[0], C_=x.shape[1])
        return out

    def perceiver_stack(self, W: _Mod, x, latents, B: int, T: int, L: int, heads: int, depth: int):
        """``for attn, ff in layers: latents = attn(x, latents) + latents; latents = ff(latents) + latents``
        (functions.py:485-487 / :588-590); x [B*T, D], latents [B*L, D]"""
        inner = heads * 64
        for i in range(depth):
            a, f = f"layers.{i}.0.", f"layers.{i}.1."
            xn = self.layernorm(x, W[a + "norm1.weight"], W[a + "norm1.bias"])
            ln = self.layernorm(latents, W[a + "norm2.weight"], W[a + "norm2.bias"])
            q = self.linear(ln, W[a + "to_q.weight"])
            kv_x = self.linear(xn, W[a + "to_kv.weight"])
            kv_l = self.linear(ln, W[a + "to_kv.weight"])
            att = self._empty(B * L, inner)
            ops.small_attn(q, kv_x, kv_l, att, B=B, Lq=L, n1=T, n2=L, heads=heads)
            latents = self.linear(att, W[a + "to_out.weight"], res=latents)
            h = self.layernorm(latents, W[f + "0.weight"], W[f + "0.bias"])
            h = ops.gelu_(self.linear(h, W[f + "1.weight"]))
            latents = self.linear(h, W[f + "3.weight"], res=latents)
        return latents


class HipProjPlusModel(_Engine):
    """``image_proj_model(id_embeds, clip_embeds, shortcut=False, scale=1.0)`` -> [B, num_tokens, cross_attention_dim]"""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda:0"):
        super().__init__(device)
        self.W = _Mod(state_dict, "", self.device)
        self.cross_attention_dim = self.W["norm.weight"].shape[0]
        self.num_tokens = self.W["proj.2.weight"].shape[0] // self.cross_attention_dim
        self.depth = 1 + max(int(k.split(".")[2]) for k in self.W.w if k.startswith("perceiver_resampler.layers."))
        self.heads = self.W["perceiver_resampler.layers.0.0.to_q.weight"].shape[0] // 64
        self.R = _Mod(state_dict, "perceiver_resampler.", self.device)

    @torch.no_grad()
    def __call__(self, id_embeds, clip_embeds, shortcut: bool = False, scale: float = 1.0):
        W, R, D, nt = self.W, self.R, self.cross_attention_dim, self.num_tokens
        ide = _h(id_embeds, self.device)
        clip = _h(clip_embeds, self.device)
        B, T, _ = clip.shape
        h = ops.gelu_(self.linear(ide, W["proj.0.weight"], W["proj.0.bias"]))
        x = self.linear(h, W["proj.2.weight"], W["proj.2.bias"]).view(B * nt, D)
        x = self.layernorm(x, W["norm.weight"], W["norm.bias"])                           # the 4 latent tokens
        feats = self.linear(clip.view(B * T, -1), R["proj_in.weight"], R["proj_in.bias"])
        lat = self.perceiver_stack(R, feats, x, B, T, nt, self.heads, self.depth)
        out = self.linear(lat, R["proj_out.weight"], R["proj_out.bias"])
        out = self.layernorm(out, R["norm_out.weight"], R["norm_out.bias"])
        if shortcut:
            out = (x.float() + scale * out.float()).half()      # functions.py:520 (on in the SDXL pipeline, ref SDXL :567)
        return out.view(B, nt, D)


class HipFacialEncoder(_Engine):
    """``FacialEncoder(prompt_embeds, multi_image_embeds, class_tokens_mask, valid_id_mask)`` -> updated prompt_embeds"""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda:0"):
        super().__init__(device)
        self.V = _Mod(state_dict, "visual_projection.", self.device)
        self.F = _Mod(state_dict, "fuse_module.", self.device)
        self.depth = 1 + max(int(k.split(".")[1]) for k in self.V.w if k.startswith("layers."))
        self.heads = self.V["layers.0.0.to_q.weight"].shape[0] // 64

    def _mlp(self, pre: str, x, residual: bool):
        F = self.F
        h = self.layernorm(x, F[pre + "layernorm.weight"], F[pre + "layernorm.bias"])
        h = ops.gelu_(self.linear(h, F[pre + "fc1.weight"], F[pre + "fc1.bias"]))
        return self.linear(h, F[pre + "fc2.weight"], F[pre + "fc2.bias"], res=x if residual else None)

    @torch.no_grad()
    def __call__(self, prompt_embeds, multi_image_embeds, class_tokens_mask, valid_id_mask):
        V, F, dev = self.V, self.F, self.device
        pe = _h(prompt_embeds, dev)
        mi = _h(multi_image_embeds, dev)
        bs, n, T, d = mi.shape
        # visual_projection = AttentionMLP (functions.py:572-593): one learned latent per crop
        D = V["latents"].shape[-1]
        lat = V["latents"].view(1, D).repeat(bs * n, 1).contiguous()
        feats = self.linear(mi.view(bs * n * T, d), V["proj_in.weight"], V["proj_in.bias"])
        lat = self.perceiver_stack(V, feats, lat, bs * n, T, 1, self.heads, self.depth)
        ide = self.linear(lat, V["proj_out.weight"], V["proj_out.bias"])
        ide = self.layernorm(ide, V["norm_out.weight"], V["norm_out.bias"])               # [bs*n, E]
        # fuse_module (attention.py:24-48): rows of the valid crops meet the trigger-token rows of the prompt
        vmask = valid_id_mask.to(dev).flatten().bool()
        cmask = class_tokens_mask.to(dev).reshape(-1).bool()
        valid = ide[vmask].contiguous()
        flat = pe.reshape(-1, pe.shape[-1]).clone()
        tok = flat[cmask].contiguous()
        assert tok.shape[0] == valid.shape[0], f"{tok.shape[0]} trigger tokens vs {valid.shape[0]} valid crops"
        if tok.shape[0]:
            s = self._mlp("mlp1.", torch.cat([tok, valid], dim=-1).contiguous(), residual=False)
            s = (s.float() + tok.float()).half()          # mlp1(...) + prompt_embeds (attention.py:19)
            s = self._mlp("mlp2.", s, residual=True)
            s = self.layernorm(s, F["layer_norm.weight"], F["layer_norm.bias"])
            flat[cmask] = s
        return flat.view(bs, pe.shape[1], -1)


class HipIDConditioner:
    """ProjPlusModel + FacialEncoder + the concatenation order of the reference's ``__call__`` (:479-507)."""

    def __init__(self, image_proj_sd: Dict[str, torch.Tensor], facial_encoder_sd: Dict[str, torch.Tensor], device="cuda:0"):
        self.device = torch.device(device)
        self.image_proj_model = HipProjPlusModel(image_proj_sd, device)
        self.FacialEncoder = HipFacialEncoder(facial_encoder_sd, device)

    @torch.no_grad()
    def __call__(self, *, text_embeds, negative_embeds, text_only_embeds, faceid_embeds, clip_embeds, uncond_clip_embeds,
                 facial_embeds, uncond_facial_embeds, facial_token_mask, valid_facial_mask, s_scale: float = 1.0,
                 shortcut: bool = False, sdxl: bool = False) -> torch.Tensor:
        """-> ``prompt_embeds`` [3B, 77 + num_tokens, Dc] = cat([null, augmented, text_only]): what
        ``ConsistentIDStableDiffusionPipeline.__call__(prompt_embeds=...)`` takes.
        ``sdxl=True``: the SDXL pipeline's four sets [4B, ...] = cat([null_text_only, augmented, text_only, null_facial])
        (ref SDXL :586-590) -- it keeps the raw negative embeds for the steps up to start_merge_step -- and it calls
        get_image_embeds with ``shortcut=True`` (ref SDXL :567): pass shortcut=True there."""
        ip, fe, dev = self.image_proj_model, self.FacialEncoder, self.device
        fid = _h(faceid_embeds, dev)
        tok = ip(fid, clip_embeds, shortcut=shortcut, scale=s_scale)                                   # :197
        utok = ip(torch.zeros_like(fid), uncond_clip_embeds, shortcut=shortcut, scale=s_scale)         # :198
        facial = fe(text_embeds, facial_embeds, facial_token_mask, valid_facial_mask)                  # :190
        ufacial = fe(negative_embeds, uncond_facial_embeds, facial_token_mask, valid_facial_mask)      # :193
        augmented = torch.cat([facial, tok], dim=1)                                                    # :492
        null = torch.cat([ufacial, utok], dim=1)                                                       # :493
        text_only = torch.cat([_h(text_only_embeds, dev), tok], dim=1)                                 # :504
        if sdxl:
            null_text_only = torch.cat([_h(negative_embeds, dev), utok], dim=1)                        # SDXL :590
            return torch.cat([null_text_only, augmented, text_only, null], dim=0)
        return torch.cat([null, augmented, text_only], dim=0)
