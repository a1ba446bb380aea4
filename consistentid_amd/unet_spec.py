"""UNet topology + parameter inventory (names/shapes identical to diffusers==0.23.0's
``UNet2DConditionModel`` state_dict, which is what the reference's ``from_pretrained``
loads -- infer.py:17, pipline_StableDiffusion_ConsistentID.py:33).  Pure metadata: the
HIP engine (unet.py) and the weight packer (weights.py) are driven by this."""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple


@dataclass(frozen=True)
class UNetConfig:
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",)
    up_block_types: Tuple[str, ...] = ("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3
    layers_per_block: int = 2
    transformer_layers_per_block: Tuple[int, ...] = (1, 1, 1, 1)
    num_attention_heads: Tuple[int, ...] = (8, 8, 8, 8)   # SD1.5: "attention_head_dim" = head COUNT
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


def sd15_config(**kw) -> UNetConfig:
    return UNetConfig(**kw)


def sdxl_config(**kw) -> UNetConfig:
    base = dict(
        sample_size=128, block_out_channels=(320, 640, 1280),
        down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
        up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
        transformer_layers_per_block=(1, 2, 10), num_attention_heads=(5, 10, 20),
        cross_attention_dim=2048, use_linear_projection=True, addition_embed_type="text_time",
        addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816, family="sdxl")
    base.update(kw)
    return UNetConfig(**base)


def tiny_config(family: str = "sd15") -> UNetConfig:
    """Same topology, toy widths: used by the fast parity tests."""
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


# ------------------------------------------------------------------ structural walk
@dataclass
class ResnetSpec:
    name: str
    cin: int
    cout: int


@dataclass
class TransformerSpec:
    name: str
    channels: int
    heads: int
    n_layers: int


@dataclass
class BlockSpec:
    name: str
    resnets: List[ResnetSpec]
    attentions: List[TransformerSpec]          # empty when the block has no attention
    sampler: Optional[str]                     # "downsamplers.0" / "upsamplers.0" / None
    channels: int


def walk(cfg: UNetConfig):
    """(down blocks, mid block, up blocks) in diffusers' construction order."""
    boc = cfg.block_out_channels
    downs: List[BlockSpec] = []
    out = boc[0]
    for i, typ in enumerate(cfg.down_block_types):
        cin, out = out, boc[i]
        name = f"down_blocks.{i}"
        res, att = [], []
        for j in range(cfg.layers_per_block):
            res.append(ResnetSpec(f"{name}.resnets.{j}", cin if j == 0 else out, out))
            if typ.startswith("CrossAttn"):
                att.append(TransformerSpec(f"{name}.attentions.{j}", out, cfg.num_attention_heads[i],
                                           cfg.transformer_layers_per_block[i]))
        downs.append(BlockSpec(name, res, att, "downsamplers.0" if i != len(boc) - 1 else None, out))
    c = boc[-1]
    mid = BlockSpec("mid_block",
                    [ResnetSpec("mid_block.resnets.0", c, c), ResnetSpec("mid_block.resnets.1", c, c)],
                    [TransformerSpec("mid_block.attentions.0", c, cfg.num_attention_heads[-1],
                                     cfg.transformer_layers_per_block[-1])], None, c)
    rev = list(reversed(boc))
    rheads = list(reversed(cfg.num_attention_heads))
    rtfm = list(reversed(cfg.transformer_layers_per_block))
    ups: List[BlockSpec] = []
    out = rev[0]
    for i, typ in enumerate(cfg.up_block_types):
        prev, out = out, rev[i]
        cin = rev[min(i + 1, len(boc) - 1)]
        name = f"up_blocks.{i}"
        n = cfg.layers_per_block + 1
        res, att = [], []
        for j in range(n):
            skip = cin if j == n - 1 else out
            rin = prev if j == 0 else out
            res.append(ResnetSpec(f"{name}.resnets.{j}", rin + skip, out))
            if typ.startswith("CrossAttn"):
                att.append(TransformerSpec(f"{name}.attentions.{j}", out, rheads[i], rtfm[i]))
        ups.append(BlockSpec(name, res, att, "upsamplers.0" if i != len(boc) - 1 else None, out))
    return downs, mid, ups


def attn_processor_names(cfg: UNetConfig) -> List[str]:
    """``unet.attn_processors.keys()`` order: down_blocks, up_blocks, mid_block (the reference
    indexes adapter weights by it: pipline_StableDiffusion_ConsistentID.py:143-144,155-164)."""
    downs, mid, ups = walk(cfg)
    names = []
    for blk in downs + ups + [mid]:
        for t in blk.attentions:
            for k in range(t.n_layers):
                for a in ("attn1", "attn2"):
                    names.append(f"{t.name}.transformer_blocks.{k}.{a}.processor")
    return names


def hidden_size_of(cfg: UNetConfig, proc_name: str) -> int:
    """set_ip_adapter's rule (pipline_StableDiffusion_ConsistentID.py:157-164)."""
    if proc_name.startswith("mid_block"):
        return cfg.block_out_channels[-1]
    if proc_name.startswith("up_blocks"):
        return list(reversed(cfg.block_out_channels))[int(proc_name[len("up_blocks.")])]
    return cfg.block_out_channels[int(proc_name[len("down_blocks.")])]


CONTROLNET_COND_CHANNELS = (16, 32, 96, 256)   # diffusers ControlNetModel.conditioning_embedding_out_channels


def controlnet_zero_conv_channels(cfg: UNetConfig) -> List[int]:
    """channel count of each entry of ``controlnet_down_blocks`` (one 1x1 conv per UNet skip tensor)"""
    boc = cfg.block_out_channels
    ch = [boc[0]]
    for i in range(len(boc)):
        ch += [boc[i]] * cfg.layers_per_block
        if i != len(boc) - 1:
            ch.append(boc[i])
    return ch


def controlnet_param_shapes(cfg: UNetConfig, conditioning_channels: int = 3,
                            cond_channels: Tuple[int, ...] = CONTROLNET_COND_CHANNELS):
    """state_dict of diffusers==0.23.0 ``ControlNetModel`` for the UNet config ``cfg``: the UNet's encoder half
    plus the condition embedding and the zero convs (what the reference loads at demo/controlnet_demo.py:44-47
    and calls at pipelines/StableDIffusionControlNetInpaint_ConsistentID.py:405-412).  SD1.5: 361,279,120 parameters."""
    P = unet_param_shapes(cfg, encoder_only=True)
    boc = cfg.block_out_channels

    def conv(n, o, i, k):
        P[f"{n}.weight"] = (o, i, k, k)
        P[f"{n}.bias"] = (o,)

    e = "controlnet_cond_embedding"
    conv(f"{e}.conv_in", cond_channels[0], conditioning_channels, 3)
    for i in range(len(cond_channels) - 1):
        conv(f"{e}.blocks.{2 * i}", cond_channels[i], cond_channels[i], 3)
        conv(f"{e}.blocks.{2 * i + 1}", cond_channels[i + 1], cond_channels[i], 3)
    conv(f"{e}.conv_out", boc[0], cond_channels[-1], 3)
    for i, c in enumerate(controlnet_zero_conv_channels(cfg)):
        conv(f"controlnet_down_blocks.{i}", c, c, 1)
    conv("controlnet_mid_block", boc[-1], boc[-1], 1)
    return P


def unet_param_shapes(cfg: UNetConfig, encoder_only: bool = False) -> "OrderedDict[str, Tuple[int, ...]]":
    P: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = cfg.block_out_channels
    ted = cfg.time_embed_dim

    def lin(n, o, i, bias=True):
        P[f"{n}.weight"] = (o, i)
        if bias:
            P[f"{n}.bias"] = (o,)

    def conv(n, o, i, k):
        P[f"{n}.weight"] = (o, i, k, k)
        P[f"{n}.bias"] = (o,)

    def norm(n, c):
        P[f"{n}.weight"] = (c,)
        P[f"{n}.bias"] = (c,)

    def resnet(r: ResnetSpec):
        norm(f"{r.name}.norm1", r.cin)
        conv(f"{r.name}.conv1", r.cout, r.cin, 3)
        lin(f"{r.name}.time_emb_proj", r.cout, ted)
        norm(f"{r.name}.norm2", r.cout)
        conv(f"{r.name}.conv2", r.cout, r.cout, 3)
        if r.cin != r.cout:
            conv(f"{r.name}.conv_shortcut", r.cout, r.cin, 1)

    def transformer(t: TransformerSpec):
        c = t.channels
        norm(f"{t.name}.norm", c)
        if cfg.use_linear_projection:
            lin(f"{t.name}.proj_in", c, c)
        else:
            conv(f"{t.name}.proj_in", c, c, 1)
        for k in range(t.n_layers):
            b = f"{t.name}.transformer_blocks.{k}"
            norm(f"{b}.norm1", c)
            for a, kv in (("attn1", c), ("attn2", cfg.cross_attention_dim)):
                lin(f"{b}.{a}.to_q", c, c, bias=False)
                lin(f"{b}.{a}.to_k", c, kv, bias=False)
                lin(f"{b}.{a}.to_v", c, kv, bias=False)
                lin(f"{b}.{a}.to_out.0", c, c)
                if a == "attn1":
                    norm(f"{b}.norm2", c)
            norm(f"{b}.norm3", c)
            lin(f"{b}.ff.net.0.proj", 8 * c, c)
            lin(f"{b}.ff.net.2", c, 4 * c)
        if cfg.use_linear_projection:
            lin(f"{t.name}.proj_out", c, c)
        else:
            conv(f"{t.name}.proj_out", c, c, 1)

    conv("conv_in", boc[0], cfg.in_channels, 3)
    lin("time_embedding.linear_1", ted, boc[0])
    lin("time_embedding.linear_2", ted, ted)
    if cfg.addition_embed_type == "text_time":
        lin("add_embedding.linear_1", ted, cfg.projection_class_embeddings_input_dim)
        lin("add_embedding.linear_2", ted, ted)
    downs, mid, ups = walk(cfg)
    for blk in downs:
        # diffusers registers attentions before resnets in CrossAttn blocks; order is irrelevant
        # for a state_dict (a mapping), only names and shapes matter.
        for r in blk.resnets:
            resnet(r)
        for t in blk.attentions:
            transformer(t)
        if blk.sampler:
            conv(f"{blk.name}.{blk.sampler}.conv", blk.channels, blk.channels, 3)
    for blk in ([] if encoder_only else ups):
        for r in blk.resnets:
            resnet(r)
        for t in blk.attentions:
            transformer(t)
        if blk.sampler:
            conv(f"{blk.name}.{blk.sampler}.conv", blk.channels, blk.channels, 3)
    for t in mid.attentions:
        transformer(t)
    for r in mid.resnets:
        resnet(r)
    if not encoder_only:
        norm("conv_norm_out", boc[0])
        conv("conv_out", cfg.out_channels, boc[0], 3)
    return P


def adapter_param_shapes(cfg: UNetConfig, rank: int = 128) -> "OrderedDict[str, Tuple[int, ...]]":
    """Keys of the checkpoint's ``adapter_modules`` entry:
    ``{idx}.to_{q,k,v,out}_lora.{down,up}.weight`` and ``{idx}.to_{k,v}_ip.weight``
    (attention.py:103-106, :195-205; loader pipline_StableDiffusion_ConsistentID.py:143-144)."""
    P: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for idx, name in enumerate(attn_processor_names(cfg)):
        c = hidden_size_of(cfg, name)
        cross = None if name.endswith("attn1.processor") else cfg.cross_attention_dim
        kv = cross or c
        for which, i in (("q", c), ("k", kv), ("v", kv), ("out", c)):
            P[f"{idx}.to_{which}_lora.down.weight"] = (rank, i)
            P[f"{idx}.to_{which}_lora.up.weight"] = (c, rank)
        if cross is not None:
            P[f"{idx}.to_k_ip.weight"] = (c, kv)
            P[f"{idx}.to_v_ip.weight"] = (c, kv)
    return P


def count_params(shapes: Dict[str, Tuple[int, ...]]) -> int:
    n = 0
    for s in shapes.values():
        k = 1
        for d in s:
            k *= d
        n += k
    return n
