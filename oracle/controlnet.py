"""Oracle (test infrastructure): fp32 CPU restatement of diffusers==0.23.0 ``ControlNetModel``
as the reference's ControlNet-inpaint pipeline drives it.

PARITY UNPINNED: diffusers is not vendored under /root/reference and not installed here
(see oracle/unet.py).  This file restates the published algorithm of
``diffusers/models/controlnet.py`` (0.23.0) with diffusers' parameter names, anchored on
the reference's call site:

  pipelines/StableDIffusionControlNetInpaint_ConsistentID.py:405-412
      down_block_res_samples, mid_block_res_sample = self.controlnet(
          control_model_input, t, encoder_hidden_states=controlnet_prompt_embeds,
          controlnet_cond=control_image, conditioning_scale=cond_scale, return_dict=False)

The ControlNet keeps diffusers' DEFAULT attention processors (the reference only swaps the
UNet's, pipline_StableDiffusion_ConsistentID.py:152-174), so all 81 context tokens -- the 77
text tokens and the 4 identity tokens -- are ordinary keys of one softmax.  ``guess_mode`` is
not forwarded by the reference, so every residual is scaled by ``conditioning_scale`` only.
Structural pin: the SD1.5 configuration has 361,279,120 parameters (tests/test_oracle_invariants.py).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet import DownBlock, MidBlock, TimestepEmbedding, UNetConfig, timestep_embedding


class ControlNetConditioningEmbedding(nn.Module):
    """image-space condition (3 x 8h x 8w) -> latent-space feature map (C0 x h x w): conv_in, then per
    stage a same-width conv and a stride-2 widening conv, SiLU after each, and a zero-initialised conv_out"""

    def __init__(self, conditioning_embedding_channels: int, conditioning_channels: int = 3,
                 block_out_channels: Sequence[int] = (16, 32, 96, 256)):
        super().__init__()
        boc = list(block_out_channels)
        self.conv_in = nn.Conv2d(conditioning_channels, boc[0], 3, padding=1)
        self.blocks = nn.ModuleList()
        for i in range(len(boc) - 1):
            self.blocks.append(nn.Conv2d(boc[i], boc[i], 3, padding=1))
            self.blocks.append(nn.Conv2d(boc[i], boc[i + 1], 3, padding=1, stride=2))
        self.conv_out = nn.Conv2d(boc[-1], conditioning_embedding_channels, 3, padding=1)

    def forward(self, cond):
        e = F.silu(self.conv_in(cond))
        for blk in self.blocks:
            e = F.silu(blk(e))
        return self.conv_out(e)


class ControlNetModel(nn.Module):
    """Encoder half of the UNet + one 1x1 "zero conv" per skip tensor and one for the mid block."""

    def __init__(self, cfg: UNetConfig, conditioning_embedding_out_channels: Sequence[int] = (16, 32, 96, 256),
                 conditioning_channels: int = 3):
        super().__init__()
        self.config = cfg
        boc = cfg.block_out_channels
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], cfg.time_embed_dim)
        self.controlnet_cond_embedding = ControlNetConditioningEmbedding(
            boc[0], conditioning_channels, conditioning_embedding_out_channels)
        self.down_blocks = nn.ModuleList()
        self.controlnet_down_blocks = nn.ModuleList([nn.Conv2d(boc[0], boc[0], 1)])
        out = boc[0]
        for i, typ in enumerate(cfg.down_block_types):
            cin, out = out, boc[i]
            last = i == len(boc) - 1
            self.down_blocks.append(DownBlock(cfg, cin, out, typ.startswith("CrossAttn"), cfg.num_attention_heads[i],
                                              cfg.transformer_layers_per_block[i], add_down=not last))
            for _ in range(cfg.layers_per_block):
                self.controlnet_down_blocks.append(nn.Conv2d(out, out, 1))
            if not last:
                self.controlnet_down_blocks.append(nn.Conv2d(out, out, 1))
        self.controlnet_mid_block = nn.Conv2d(boc[-1], boc[-1], 1)
        self.mid_block = MidBlock(cfg, boc[-1], cfg.num_attention_heads[-1], cfg.transformer_layers_per_block[-1])

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale: float = 1.0,
                return_dict: bool = False) -> Tuple[List[torch.Tensor], torch.Tensor]:
        cfg = self.config
        b = sample.shape[0]
        t = torch.as_tensor(timestep, device=sample.device)
        t = t.reshape(-1).expand(b) if t.numel() == 1 else t
        emb = self.time_embedding(timestep_embedding(t, cfg.block_out_channels[0]).to(sample.dtype))
        x = self.conv_in(sample)
        x = x + self.controlnet_cond_embedding(controlnet_cond)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, encoder_hidden_states, None)
            skips += outs
        x = self.mid_block(x, emb, encoder_hidden_states, None)
        down = [conv(s) * conditioning_scale for s, conv in zip(skips, self.controlnet_down_blocks)]
        mid = self.controlnet_mid_block(x) * conditioning_scale
        return down, mid
