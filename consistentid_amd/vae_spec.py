"""VAE topology + parameter inventory (names/shapes of diffusers==0.23.0 ``AutoencoderKL``'s state_dict, which the
reference's ``from_pretrained`` loads into ``pipe.vae``).  Pure metadata for the decoder engine (vae.py)."""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple


@dataclass(frozen=True)
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215
    force_upcast: bool = False        # SDXL's VAE sets this: fp32 decode (SDXL :670-676) -> vae.HipVAEDecoderF32


def sd_vae_config(**kw) -> VAEConfig:
    return VAEConfig(**kw)


def sdxl_vae_config(**kw) -> VAEConfig:
    """stabilityai/stable-diffusion-xl-base-1.0 vae/config.json: the SD architecture, scaling 0.13025, force_upcast"""
    return VAEConfig(**{**dict(scaling_factor=0.13025, force_upcast=True), **kw})


def tiny_vae_config() -> VAEConfig:
    return VAEConfig(block_out_channels=(64, 128), layers_per_block=1)


def decoder_blocks(cfg: VAEConfig):
    """[(name, cin, cout, n_resnets, has_upsampler)] of ``decoder.up_blocks`` in execution order"""
    rev = list(reversed(cfg.block_out_channels))
    out, blocks = rev[0], []
    for i in range(len(rev)):
        prev, out = out, rev[i]
        blocks.append((f"decoder.up_blocks.{i}", prev, out, cfg.layers_per_block + 1, i != len(rev) - 1))
    return blocks


def vae_param_shapes(cfg: VAEConfig, decoder_only: bool = False) -> "OrderedDict[str, Tuple[int, ...]]":
    P: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc, L = cfg.block_out_channels, cfg.latent_channels

    def conv(n, o, i, k):
        P[f"{n}.weight"] = (o, i, k, k)
        P[f"{n}.bias"] = (o,)

    def vec(n, c):
        P[f"{n}.weight"] = (c,)
        P[f"{n}.bias"] = (c,)

    def lin(n, o, i):
        P[f"{n}.weight"] = (o, i)
        P[f"{n}.bias"] = (o,)

    def resnet(n, cin, cout):
        vec(f"{n}.norm1", cin)
        conv(f"{n}.conv1", cout, cin, 3)
        vec(f"{n}.norm2", cout)
        conv(f"{n}.conv2", cout, cout, 3)
        if cin != cout:
            conv(f"{n}.conv_shortcut", cout, cin, 1)

    def mid(n, c):
        a = f"{n}.attentions.0"
        vec(f"{a}.group_norm", c)
        for w in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(f"{a}.{w}", c, c)
        resnet(f"{n}.resnets.0", c, c)
        resnet(f"{n}.resnets.1", c, c)

    if not decoder_only:
        conv("encoder.conv_in", boc[0], cfg.in_channels, 3)
        out = boc[0]
        for i in range(len(boc)):
            cin, out = out, boc[i]
            for j in range(cfg.layers_per_block):
                resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else out, out)
            if i != len(boc) - 1:
                conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", out, out, 3)
        mid("encoder.mid_block", boc[-1])
        vec("encoder.conv_norm_out", boc[-1])
        conv("encoder.conv_out", 2 * L, boc[-1], 3)
        conv("quant_conv", 2 * L, 2 * L, 1)
    conv("decoder.conv_in", boc[-1], L, 3)
    mid("decoder.mid_block", boc[-1])
    for name, cin, cout, n, up in decoder_blocks(cfg):
        for j in range(n):
            resnet(f"{name}.resnets.{j}", cin if j == 0 else cout, cout)
        if up:
            conv(f"{name}.upsamplers.0.conv", cout, cout, 3)
    vec("decoder.conv_norm_out", boc[0])
    conv("decoder.conv_out", cfg.out_channels, boc[0], 3)
    conv("post_quant_conv", L, L, 1)
    return P
