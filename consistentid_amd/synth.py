"""Synthetic weights and inputs for tests and benchmarks (there are no checkpoints or
datasets in the build/bench environment).  Weight tensors follow the torch default
init bounds (U(-1/sqrt(fan_in), 1/sqrt(fan_in))) with three deliberate deviations so
that the numerics tests are not vacuous:
  * LoRA ``up`` ~ N(0, 0.02^2) instead of the zero init of diffusers' LoRALinearLayer,
  * ``to_{k,v}_ip`` = ``to_{k,v}`` + N(0, 0.01^2)  (mirrors reference train.py:168-174),
  * ``to_q`` / ``to_k`` scaled x3 and norm affine params perturbed, so softmaxes are not
    uniform and every affine path is exercised.
Everything is rounded to fp16 so the fp32 oracle and the fp16 HIP path see identical values.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

from .unet_spec import UNetConfig, adapter_param_shapes, unet_param_shapes


def _gen(device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def random_controlnet_state_dict(cfg: UNetConfig, seed: int = 3, device="cpu") -> Dict[str, torch.Tensor]:
    """diffusers ControlNetModel state_dict for ``cfg`` (the "zero convs" get ordinary random weights: a trained
    ControlNet's are not zero, and zeros would make the parity test vacuous)"""
    from .unet_spec import controlnet_param_shapes
    return _random_state_dict(controlnet_param_shapes(cfg), seed, device)


def random_unet_state_dict(cfg: UNetConfig, seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    return _random_state_dict(unet_param_shapes(cfg), seed, device)


def random_vae_state_dict(cfg, seed: int = 5, device="cpu", decoder_only: bool = False) -> Dict[str, torch.Tensor]:
    """diffusers AutoencoderKL state_dict for a ``vae_spec.VAEConfig``"""
    from .vae_spec import vae_param_shapes
    return _random_state_dict(vae_param_shapes(cfg, decoder_only), seed, device)


def _random_state_dict(shapes, seed: int, device) -> Dict[str, torch.Tensor]:
    g = _gen(device, seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in shapes.items():
        leaf = name.rsplit(".", 2)[-2] if name.count(".") >= 1 else name
        is_norm = (".norm" in name or "conv_norm_out" in name or "group_norm" in name) and len(shape) == 1
        if is_norm:
            t = torch.randn(shape, generator=g, device=device) * 0.1
            if name.endswith(".weight"):
                t = t + 1.0
        else:
            if name.endswith(".weight"):
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
            else:
                # bias bound uses the fan_in of the matching weight; approximate with a small uniform
                fan_in = 256
            bound = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g, device=device) * 2 - 1) * bound
            if leaf in ("to_q", "to_k"):
                t = t * 3.0
        sd[name] = t.to(torch.float16)
    return sd


def random_adapter_state_dict(cfg: UNetConfig, unet_sd: Dict[str, torch.Tensor], rank: int = 128, seed: int = 1,
                              device="cpu") -> Dict[str, torch.Tensor]:
    from .unet_spec import attn_processor_names
    g = _gen(device, seed)
    names = attn_processor_names(cfg)
    sd: Dict[str, torch.Tensor] = {}
    for key, shape in adapter_param_shapes(cfg, rank).items():
        idx = int(key.split(".", 1)[0])
        if key.endswith("lora.down.weight"):
            t = torch.randn(shape, generator=g, device=device) * (1.0 / rank)
        elif key.endswith("lora.up.weight"):
            t = torch.randn(shape, generator=g, device=device) * 0.02
        else:
            which = "to_k" if ".to_k_ip." in key else "to_v"
            base = names[idx][: -len(".processor")]
            t = unet_sd[f"{base}.{which}.weight"].to(device=device, dtype=torch.float32) \
                + torch.randn(shape, generator=g, device=device) * 0.01
        sd[key] = t.to(torch.float16)
    return sd


def random_inputs(cfg: UNetConfig, batch: int, height: int, width: int, seed_latents: int = 2024,
                  seed_embeds: int = 1, num_tokens: int = 4, text_len: int = 77, device="cpu"):
    """latents [B,4,h/8,w/8] ~ N(0,1) (infer.py:59 uses seed 2024) and the three embed sets
    (null / augmented / text-only) [B, 77+4, Dc] ~ N(0,1); SDXL pooled embeds + time ids."""
    g = _gen(device, seed_latents)
    lat = torch.randn(batch, cfg.in_channels, height // 8, width // 8, generator=g, device=device)
    g = _gen(device, seed_embeds)
    L = text_len + num_tokens
    emb = torch.randn(3, batch, L, cfg.cross_attention_dim, generator=g, device=device)
    out = {"latents": lat.to(torch.float16), "null": emb[0].to(torch.float16),
           "augmented": emb[1].to(torch.float16), "text": emb[2].to(torch.float16)}
    if cfg.addition_embed_type == "text_time":
        pdim = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim
        pooled = torch.randn(3, batch, pdim, generator=g, device=device).to(torch.float16)
        out.update({"pooled_null": pooled[0], "pooled_augmented": pooled[1], "pooled_text": pooled[2],
                    "time_ids": torch.tensor([[height, width, 0, 0, height, width]], dtype=torch.float32,
                                             device=device).repeat(2 * batch, 1)})
    return out
