"""HIP execution engine for the CLIP vision tower as the reference uses it (once per image):

    self.image_encoder(clip_image, output_hidden_states=True).hidden_states[-2]
    (pipline_StableDiffusion_ConsistentID.py:182-183, :200-201; ``CLIPVisionModelWithProjection`` of
     laion/CLIP-ViT-H-14-laion2B-s32B-b79K, loaded at :54-56)

i.e. the token states [B, 257, 1280] after all but the last encoder layer.  Weights: the ``transformers`` state_dict
(``vision_model.*``).  The 14x14/14 patch embedding is an unfold (index plumbing) + ``cid_gemm_f16`` (K padded 588 -> 640),
every encoder layer is LayerNorm -> fused QKV GEMM (V written transposed) -> ``cid_self_attn_keys_f16`` (257 real keys on a
320-token padded axis) -> out-proj GEMM (+residual) -> LayerNorm -> fc1 GEMM -> ``cid_gelu_f16`` -> fc2 GEMM (+residual).
The pooled / projected image embedding (post_layernorm, visual_projection) is never read by the reference and is not computed.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops
from .weights import LOG2E


def _h(t, dev):
    return t.to(device=dev, dtype=torch.float16).contiguous()


class HipCLIPVision:
    def __init__(self, state_dict: Dict[str, torch.Tensor], num_heads: int, patch_size: int = 14, device="cuda:0",
                 hidden_act: str = "gelu", layer_norm_eps: float = 1e-5):
        if hidden_act != "gelu":
            raise NotImplementedError(f"hidden_act {hidden_act!r}: the ViT-H tower of the reference uses exact GELU")
        self.device = dev = torch.device(device)
        sd = {k[len("vision_model."):]: v for k, v in state_dict.items() if k.startswith("vision_model.")}
        pw = sd["embeddings.patch_embedding.weight"]                     # [C, 3, P, P], no bias
        self.C, self.P, self.heads, self.eps = pw.shape[0], patch_size, num_heads, layer_norm_eps
        assert pw.shape[-1] == patch_size and self.C % num_heads == 0
        self.d = self.C // num_heads
        kp = pw[0].numel()
        self.kpad = (kp + 63) // 64 * 64
        w = torch.zeros(self.C, self.kpad)
        w[:, :kp] = pw.reshape(self.C, kp).float()
        W: Dict[str, torch.Tensor] = {"patch.w": _h(w, dev)}
        self.cls = _h(sd["embeddings.class_embedding"], dev)
        self.pos = _h(sd["embeddings.position_embedding.weight"], dev)   # [1 + n_patches, C]
        for n in ("pre_layrnorm",):
            W[f"{n}.g"], W[f"{n}.b"] = _h(sd[f"{n}.weight"], dev), _h(sd[f"{n}.bias"], dev)
        self.n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
        qs = (self.d ** -0.5) * LOG2E                                    # softmax scale (CLIPAttention.scale) and log2 e
        for i in range(self.n_layers):
            p = f"encoder.layers.{i}."
            a = p + "self_attn."
            W[f"{i}.qkv.w"] = _h(torch.cat([sd[a + "q_proj.weight"].float() * qs, sd[a + "k_proj.weight"].float(),
                                            sd[a + "v_proj.weight"].float()], 0), dev)
            W[f"{i}.qkv.b"] = _h(torch.cat([sd[a + "q_proj.bias"].float() * qs, sd[a + "k_proj.bias"].float(),
                                            sd[a + "v_proj.bias"].float()], 0), dev)
            W[f"{i}.o.w"], W[f"{i}.o.b"] = _h(sd[a + "out_proj.weight"], dev), _h(sd[a + "out_proj.bias"], dev)
            for ln in ("layer_norm1", "layer_norm2"):
                W[f"{i}.{ln}.g"], W[f"{i}.{ln}.b"] = _h(sd[p + ln + ".weight"], dev), _h(sd[p + ln + ".bias"], dev)
            for fc in ("fc1", "fc2"):
                W[f"{i}.{fc}.w"], W[f"{i}.{fc}.b"] = _h(sd[p + f"mlp.{fc}.weight"], dev), _h(sd[p + f"mlp.{fc}.bias"], dev)
        self.W = W

    def _empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float16, device=self.device)

    @torch.no_grad()
    def hidden_states(self, pixel_values: torch.Tensor, index: int = -2) -> torch.Tensor:
        """``image_encoder(pixel_values, output_hidden_states=True).hidden_states[index]`` -> [B, 1 + n_patches, C]"""
        W, C, P, dev = self.W, self.C, self.P, self.device
        n_run = self.n_layers + 1 + index if index < 0 else index      # hidden_states[k] = state after k layers
        assert 0 <= n_run <= self.n_layers
        x = pixel_values.to(device=dev, dtype=torch.float16)
        B, _, H, Wd = x.shape
        gh, gw = H // P, Wd // P
        ntok = 1 + gh * gw
        assert ntok == self.pos.shape[0], "image size does not match the position embedding"
        Np = (ntok + 63) // 64 * 64                                     # padded token axis
        # patch embedding: unfold (c, ky, kx) like Conv2d's weight layout, GEMM
        pt = x.unfold(2, P, P).unfold(3, P, P).permute(0, 2, 3, 1, 4, 5).reshape(B * gh * gw, -1)
        pin = torch.zeros(B * gh * gw, self.kpad, dtype=torch.float16, device=dev)
        pin[:, :pt.shape[1]] = pt
        pe = self._empty(B * gh * gw, C)
        ops.gemm(pin, W["patch.w"], pe, M=B * gh * gw, N=C, c1=self.kpad)
        h = torch.zeros(B, Np, C, dtype=torch.float16, device=dev)      # pad rows start at zero and stay row-local
        h[:, 0] = self.cls
        h[:, 1:ntok] = pe.view(B, gh * gw, C)
        h[:, :ntok] += self.pos                                         # embeddings = patches (+cls) + position (fp16 add)
        h = h.view(B * Np, C)
        M = B * Np
        x = self._empty(M, C)
        ops.layernorm(h, x, W["pre_layrnorm.g"], W["pre_layrnorm.b"], M=M, C_=C, eps=self.eps)
        for i in range(n_run):
            ln = self._empty(M, C)
            ops.layernorm(x, ln, W[f"{i}.layer_norm1.g"], W[f"{i}.layer_norm1.b"], M=M, C_=C, eps=self.eps)
            qk = self._empty(M, 2 * C)
            vt = self._empty(B * self.heads * ops.dvp_of(self.d) * Np)
            ops.gemm(ln, W[f"{i}.qkv.w"], qk, M=M, N=3 * C, c1=C, bias=W[f"{i}.qkv.b"], mode=2, vt=vt, n_vt0=2 * C,
                     heads=self.heads, dhead=self.d, ntok=Np)
            ao = self._empty(M, C)
            ops.self_attn(qk, qk[:, C:], vt, ao, B=B, N=Np, heads=self.heads, d=self.d, ldq=2 * C, ldk=2 * C, ldo=C,
                          n_keys=ntok)
            x2 = self._empty(M, C)
            ops.gemm(ao, W[f"{i}.o.w"], x2, M=M, N=C, c1=C, bias=W[f"{i}.o.b"], res=x, ldr=C)
            ln2 = self._empty(M, C)
            ops.layernorm(x2, ln2, W[f"{i}.layer_norm2.g"], W[f"{i}.layer_norm2.b"], M=M, C_=C, eps=self.eps)
            f = self._empty(M, W[f"{i}.fc1.w"].shape[0])
            ops.gemm(ln2, W[f"{i}.fc1.w"], f, M=M, N=f.shape[1], c1=C, bias=W[f"{i}.fc1.b"])
            ops.gelu_(f)
            x = self._empty(M, C)
            ops.gemm(f, W[f"{i}.fc2.w"], x, M=M, N=C, c1=f.shape[1], bias=W[f"{i}.fc2.b"], res=x2, ldr=C)
        return x.view(B, Np, C)[:, :ntok].contiguous()

    def __call__(self, pixel_values, output_hidden_states: bool = True):
        raise NotImplementedError("only hidden_states(pixel_values, -2) is implemented: the reference reads nothing else "
                                  "from the image encoder (ref :182-183, :200-201)")
