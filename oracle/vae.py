"""Oracle (test infrastructure): fp32 CPU restatement of diffusers==0.23.0 ``AutoencoderKL`` (the SD1.5 VAE) as the
reference decodes with it.

PARITY UNPINNED: diffusers is not vendored under /root/reference and not installed here (see oracle/unet.py).  This
file restates the published algorithm of ``diffusers/models/autoencoder_kl.py`` + ``vae.py`` + ``unet_2d_blocks.py``
(``UNetMidBlock2D``, ``UpDecoderBlock2D``, ``DownEncoderBlock2D``) with diffusers' parameter names, anchored on the
reference's call sites:

  pipline_StableDiffusion_ConsistentID.py:584-598   image = self.decode_latents(latents)
      (diffusers' StableDiffusionPipeline.decode_latents: latents / scaling_factor -> vae.decode ->
       (image / 2 + 0.5).clamp(0, 1) -> NHWC float32 numpy)
  pipline_StableDiffusionXL_ConsistentID.py:670-684 vae.decode(latents / scaling_factor) + image_processor.postprocess

Structural pin: the SD configuration has the published 83,653,863 parameters (tests/test_oracle_invariants.py).
The encoder is restated for that count and for producing test latents; the product only decodes.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


def sd_vae_config() -> VAEConfig:
    return VAEConfig()


def tiny_vae_config() -> VAEConfig:
    return VAEConfig(block_out_channels=(64, 128), layers_per_block=1)


class ResnetBlock(nn.Module):
    """ResnetBlock2D without a time embedding (temb_channels=None), eps 1e-6"""

    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class AttnBlock(nn.Module):
    """diffusers ``Attention`` as built by UNetMidBlock2D for the VAE: one head of width C, GroupNorm on the
    input, biased q/k/v/out projections, residual connection, softmax in fp32"""

    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x.reshape(b, c, h * w)).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (c ** -0.5), dim=-1)
        o = self.to_out[0](torch.bmm(p, v))
        return o.transpose(1, 2).reshape(b, c, h, w) + x


class MidBlock(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.attentions = nn.ModuleList([AttnBlock(c, groups)])
        self.resnets = nn.ModuleList([ResnetBlock(c, c, groups), ResnetBlock(c, c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Conv(nn.Module):
    def __init__(self, c, stride=1, padding=1):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=stride, padding=padding)


class Upsample(_Conv):
    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Downsample(_Conv):
    """Downsample2D(padding=0): asymmetric (0,1,0,1) zero pad, then a stride-2 conv without padding"""

    def __init__(self, c):
        super().__init__(c, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class UpDecoderBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, add_upsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(cin if j == 0 else cout, cout, groups) for j in range(n)])
        self.upsamplers = nn.ModuleList([Upsample(cout)]) if add_upsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.upsamplers is None else self.upsamplers[0](x)


class DownEncoderBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(cin if j == 0 else cout, cout, groups) for j in range(n)])
        self.downsamplers = nn.ModuleList([Downsample(cout)]) if add_downsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.downsamplers is None else self.downsamplers[0](x)


class Decoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.latent_channels, boc[-1], 3, padding=1)
        self.mid_block = MidBlock(boc[-1], g)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out = rev[0]
        for i in range(len(rev)):
            prev, out = out, rev[i]
            self.up_blocks.append(UpDecoderBlock(prev, out, cfg.layers_per_block + 1, g, i != len(rev) - 1))
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for blk in self.up_blocks:
            x = blk(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Encoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i in range(len(boc)):
            cin, out = out, boc[i]
            self.down_blocks.append(DownEncoderBlock(cin, out, cfg.layers_per_block, g, i != len(boc) - 1))
        self.mid_block = MidBlock(boc[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for blk in self.down_blocks:
            x = blk(x)
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class AutoencoderKL(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        self.config = cfg
        self.encoder = Encoder(cfg)
        self.decoder = Decoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)

    def encode_mean(self, x):
        """mean of the latent posterior (the mode used by inpaint pre-processing)"""
        return self.quant_conv(self.encoder(x)).chunk(2, dim=1)[0]

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


def decode_latents(vae: AutoencoderKL, latents: torch.Tensor) -> torch.Tensor:
    """diffusers StableDiffusionPipeline.decode_latents as the reference calls it (SD :587,:597), kept as a
    [B, 3, H, W] tensor in [0, 1] (the reference then moves it to NHWC numpy)."""
    image = vae.decode(latents / vae.config.scaling_factor)
    return (image / 2 + 0.5).clamp(0, 1)
