"""Oracle (test infrastructure): fp32 CPU restatement of diffusers==0.23.0
``UNet2DConditionModel`` as the reference drives it.

PARITY UNPINNED: diffusers is a third-party dependency of the reference
(requirements.txt:36) that is neither vendored under /root/reference nor installed
here.  This file restates its published algorithm; parameter names are kept
identical to diffusers' so a real ``unet/diffusion_pytorch_model`` state_dict loads
``strict=True``.  Reference call sites this serves:
  pipline_StableDiffusion_ConsistentID.py:552-557      (SD1.5 UNet call)
  pipline_StableDiffusionXL_ConsistentID.py:634-641    (SDXL, added_cond_kwargs)
  pipelines/StableDIffusionControlNetInpaint_ConsistentID.py:418-425 (extra residuals)
  attention.py:110-174 / :207-294                      (the processors it invokes)

Every diffusers-ism lives in this one file.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- config
@dataclass
class UNetConfig:
    """Subset of diffusers' UNet2DConditionModel config that the hot path reads
    (pipline_StableDiffusion_ConsistentID.py:90,156-164,406)."""
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",)
    up_block_types: Tuple[str, ...] = ("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3
    layers_per_block: int = 2
    transformer_layers_per_block: Tuple[int, ...] = (1, 1, 1, 1)
    # SD1.5 quirk: "attention_head_dim" is used as the HEAD COUNT.
    num_attention_heads: Tuple[int, ...] = (8, 8, 8, 8)
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    use_linear_projection: bool = False
    addition_embed_type: Optional[str] = None
    addition_time_embed_dim: Optional[int] = None
    projection_class_embeddings_input_dim: Optional[int] = None
    family: str = "sd15"

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4


def sd15_config() -> UNetConfig:
    return UNetConfig()


def sdxl_config() -> UNetConfig:
    return UNetConfig(
        sample_size=128,
        block_out_channels=(320, 640, 1280),
        down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
        up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
        transformer_layers_per_block=(1, 2, 10),
        num_attention_heads=(5, 10, 20),
        cross_attention_dim=2048,
        use_linear_projection=True,
        addition_embed_type="text_time",
        addition_time_embed_dim=256,
        projection_class_embeddings_input_dim=2816,
        family="sdxl",
    )


def tiny_config(family: str = "sd15") -> UNetConfig:
    """A shrunken UNet with the same topology (every block type, skip concat,
    up/down-sampling, cross-attn with ID tokens) for fast CPU tests."""
    if family == "sdxl":
        return UNetConfig(
            sample_size=16, block_out_channels=(64, 128), layers_per_block=1,
            down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"),
            up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"),
            transformer_layers_per_block=(1, 2), num_attention_heads=(1, 2),
            cross_attention_dim=128, use_linear_projection=True,
            addition_embed_type="text_time", addition_time_embed_dim=32,
            projection_class_embeddings_input_dim=64 + 6 * 32, family="sdxl")
    return UNetConfig(
        sample_size=32, block_out_channels=(64, 128, 128), layers_per_block=1,
        down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
        up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
        transformer_layers_per_block=(1, 1, 1), num_attention_heads=(2, 2, 2),
        cross_attention_dim=128, family="sd15")


# --------------------------------------------------------------------------- leaves
def timestep_embedding(t: torch.Tensor, dim: int, flip_sin_to_cos: bool = True,
                       freq_shift: float = 0.0, max_period: float = 10000.0) -> torch.Tensor:
    """diffusers ``get_timestep_embedding``: [cos, sin] when flip_sin_to_cos."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device)
    exponent = exponent / (half - freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim: int, out_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, out_dim)
        self.linear_2 = nn.Linear(out_dim, out_dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class LoRALinearLayer(nn.Module):
    """diffusers/models/lora.py (imported by the reference at attention.py:4):
    ``up(down(x))`` (* alpha/rank when network_alpha is set; None in the reference)."""

    def __init__(self, in_features, out_features, rank=4, network_alpha=None):
        super().__init__()
        self.down = nn.Linear(in_features, rank, bias=False)
        self.up = nn.Linear(rank, out_features, bias=False)
        self.network_alpha = network_alpha
        self.rank = rank
        nn.init.normal_(self.down.weight, std=1 / rank)
        nn.init.zeros_(self.up.weight)

    def forward(self, x):
        orig = x.dtype
        y = self.up(self.down(x.to(self.down.weight.dtype)))
        if self.network_alpha is not None:
            y = y * (self.network_alpha / self.rank)
        return y.to(orig)


class AttnProcessor:
    """diffusers' default processor (``AttnProcessor`` / ``AttnProcessor2_0`` arithmetic): the one a model keeps
    when nobody calls ``set_attn_processor`` on it -- here the ControlNet (the reference swaps processors on
    the UNet only, pipline_StableDiffusion_ConsistentID.py:152-174)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q = attn.head_to_batch_dim(attn.to_q(hidden_states))
        k = attn.head_to_batch_dim(attn.to_k(ctx))
        v = attn.head_to_batch_dim(attn.to_v(ctx))
        p = attn.get_attention_scores(q, k, attention_mask)
        out = attn.batch_to_head_dim(torch.bmm(p, v))
        return attn.to_out[1](attn.to_out[0](out))


class Attention(nn.Module):
    """The members of diffusers' ``Attention`` that attention.py:110-294 touches."""

    def __init__(self, query_dim: int, cross_attention_dim: Optional[int], heads: int):
        super().__init__()
        dim_head = query_dim // heads
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(cross_attention_dim or query_dim, inner, bias=False)
        self.to_v = nn.Linear(cross_attention_dim or query_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])
        # attributes the processors test (attention.py:120-136,218-234,289-292)
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.processor = AttnProcessor()

    def set_processor(self, proc):
        self.processor = proc

    def prepare_attention_mask(self, attention_mask, target_length, batch_size):
        assert attention_mask is None, "hot path never passes an attention mask"
        return None

    def head_to_batch_dim(self, t):
        b, n, c = t.shape
        h = self.heads
        return t.reshape(b, n, h, c // h).permute(0, 2, 1, 3).reshape(b * h, n, c // h)

    def batch_to_head_dim(self, t):
        bh, n, d = t.shape
        h = self.heads
        return t.reshape(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, d * h)

    def get_attention_scores(self, q, k, attention_mask=None):
        s = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype, device=q.device),
                          q, k.transpose(-1, -2), beta=0, alpha=self.scale)
        return s.softmax(dim=-1).to(q.dtype)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)  # exact (erf) GELU


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, cross_attention_dim, heads)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, encoder_hidden_states, cross_attention_kwargs=None):
        kw = cross_attention_kwargs or {}
        x = self.attn1(self.norm1(x), encoder_hidden_states=None, **kw) + x
        x = self.attn2(self.norm2(x), encoder_hidden_states=encoder_hidden_states, **kw) + x
        x = self.ff(self.norm3(x)) + x
        return x


class Transformer2DModel(nn.Module):
    def __init__(self, channels, heads, n_layers, cross_attention_dim, groups, use_linear_projection):
        super().__init__()
        self.use_linear_projection = use_linear_projection
        self.norm = nn.GroupNorm(groups, channels, eps=1e-6, affine=True)
        if use_linear_projection:
            self.proj_in = nn.Linear(channels, channels)
            self.proj_out = nn.Linear(channels, channels)
        else:
            self.proj_in = nn.Conv2d(channels, channels, 1)
            self.proj_out = nn.Conv2d(channels, channels, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(channels, heads, cross_attention_dim) for _ in range(n_layers)])

    def forward(self, x, encoder_hidden_states, cross_attention_kwargs=None):
        b, c, h, w = x.shape
        residual = x
        y = self.norm(x)
        if self.use_linear_projection:
            y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
            y = self.proj_in(y)
        else:
            y = self.proj_in(y)
            y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
        for blk in self.transformer_blocks:
            y = blk(y, encoder_hidden_states, cross_attention_kwargs)
        if self.use_linear_projection:
            y = self.proj_out(y)
            y = y.reshape(b, h, w, c).permute(0, 3, 1, 2)
        else:
            y = y.reshape(b, h, w, c).permute(0, 3, 1, 2)
            y = self.proj_out(y)
        return y + residual


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h  # output_scale_factor == 1.0


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cfg: UNetConfig, cin, cout, has_attn, heads, n_tfm, add_down):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.attentions = nn.ModuleList() if has_attn else None
        for j in range(cfg.layers_per_block):
            self.resnets.append(ResnetBlock2D(cin if j == 0 else cout, cout, cfg.time_embed_dim,
                                              cfg.norm_num_groups, cfg.norm_eps))
            if has_attn:
                self.attentions.append(Transformer2DModel(cout, heads, n_tfm, cfg.cross_attention_dim,
                                                          cfg.norm_num_groups, cfg.use_linear_projection))
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x, temb, ehs, cak):
        outs = []
        for j, res in enumerate(self.resnets):
            x = res(x, temb)
            if self.attentions is not None:
                x = self.attentions[j](x, ehs, cak)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, cfg: UNetConfig, c, heads, n_tfm):
        super().__init__()
        mk = lambda: ResnetBlock2D(c, c, cfg.time_embed_dim, cfg.norm_num_groups, cfg.norm_eps)
        self.attentions = nn.ModuleList([Transformer2DModel(
            c, heads, n_tfm, cfg.cross_attention_dim, cfg.norm_num_groups, cfg.use_linear_projection)])
        self.resnets = nn.ModuleList([mk(), mk()])

    def forward(self, x, temb, ehs, cak):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ehs, cak)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, cfg: UNetConfig, cin, cout, cprev, has_attn, heads, n_tfm, add_up):
        super().__init__()
        n = cfg.layers_per_block + 1
        self.resnets = nn.ModuleList()
        self.attentions = nn.ModuleList() if has_attn else None
        for j in range(n):
            skip = cin if j == n - 1 else cout
            rin = cprev if j == 0 else cout
            self.resnets.append(ResnetBlock2D(rin + skip, cout, cfg.time_embed_dim,
                                              cfg.norm_num_groups, cfg.norm_eps))
            if has_attn:
                self.attentions.append(Transformer2DModel(cout, heads, n_tfm, cfg.cross_attention_dim,
                                                          cfg.norm_num_groups, cfg.use_linear_projection))
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x, skips: List[torch.Tensor], temb, ehs, cak):
        for j, res in enumerate(self.resnets):
            x = torch.cat([x, skips.pop()], dim=1)
            x = res(x, temb)
            if self.attentions is not None:
                x = self.attentions[j](x, ehs, cak)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


@dataclass
class UNetOutput:
    sample: torch.Tensor


class UNet2DConditionModel(nn.Module):
    """Registration order matters: ``attn_processors`` enumerates down_blocks,
    up_blocks, mid_block and the reference loads adapter
    weights by that index (pipline_StableDiffusion_ConsistentID.py:143-144)."""

    def __init__(self, cfg: UNetConfig):
        super().__init__()
        self.config = cfg
        boc = cfg.block_out_channels
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], cfg.time_embed_dim)
        if cfg.addition_embed_type == "text_time":
            self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim,
                                                   cfg.time_embed_dim)
        self.down_blocks = nn.ModuleList()
        self.up_blocks = nn.ModuleList()
        out = boc[0]
        for i, typ in enumerate(cfg.down_block_types):
            cin, out = out, boc[i]
            self.down_blocks.append(DownBlock(cfg, cin, out, typ.startswith("CrossAttn"),
                                              cfg.num_attention_heads[i],
                                              cfg.transformer_layers_per_block[i],
                                              add_down=i != len(boc) - 1))
        self.mid_block = MidBlock(cfg, boc[-1], cfg.num_attention_heads[-1],
                                  cfg.transformer_layers_per_block[-1])
        rev = list(reversed(boc))
        rheads = list(reversed(cfg.num_attention_heads))
        rtfm = list(reversed(cfg.transformer_layers_per_block))
        out = rev[0]
        for i, typ in enumerate(cfg.up_block_types):
            prev, out = out, rev[i]
            cin = rev[min(i + 1, len(boc) - 1)]
            self.up_blocks.append(UpBlock(cfg, cin, out, prev, typ.startswith("CrossAttn"),
                                          rheads[i], rtfm[i], add_up=i != len(boc) - 1))
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, boc[0], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    # -- processor plumbing (diffusers protocol used by set_ip_adapter, ref :152-174)
    @property
    def attn_processors(self) -> Dict[str, nn.Module]:
        procs: Dict[str, nn.Module] = {}

        def walk(name, mod):
            if isinstance(mod, Attention):
                procs[f"{name}.processor"] = mod.processor
                return
            for sub, child in mod.named_children():
                walk(f"{name}.{sub}", child)

        for name, child in self.named_children():
            walk(name, child)
        return procs

    def set_attn_processor(self, procs: Dict[str, nn.Module]):
        def walk(name, mod):
            if isinstance(mod, Attention):
                mod.set_processor(procs[f"{name}.processor"])
                return
            for sub, child in mod.named_children():
                walk(f"{name}.{sub}", child)

        for name, child in self.named_children():
            walk(name, child)

    @property
    def in_channels(self):
        return self.config.in_channels

    def forward(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None,
                added_cond_kwargs=None, down_block_additional_residuals=None,
                mid_block_additional_residual=None) -> UNetOutput:
        cfg = self.config
        b = sample.shape[0]
        t = torch.as_tensor(timestep, device=sample.device)
        t = t.reshape(-1).expand(b) if t.numel() == 1 else t
        emb = self.time_embedding(timestep_embedding(t, cfg.block_out_channels[0]).to(sample.dtype))
        if cfg.addition_embed_type == "text_time":
            text_embeds = added_cond_kwargs["text_embeds"]
            time_ids = added_cond_kwargs["time_ids"]
            te = timestep_embedding(time_ids.flatten(), cfg.addition_time_embed_dim)
            te = te.reshape(text_embeds.shape[0], -1)
            emb = emb + self.add_embedding(torch.cat([text_embeds, te], dim=-1).to(emb.dtype))
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, encoder_hidden_states, cross_attention_kwargs)
            skips += outs
        if down_block_additional_residuals is not None:
            assert len(down_block_additional_residuals) == len(skips)
            skips = [s + r for s, r in zip(skips, down_block_additional_residuals)]
        x = self.mid_block(x, emb, encoder_hidden_states, cross_attention_kwargs)
        if mid_block_additional_residual is not None:
            x = x + mid_block_additional_residual
        for blk in self.up_blocks:
            x = blk(x, skips, emb, encoder_hidden_states, cross_attention_kwargs)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return UNetOutput(sample=x)
