"""state_dict (diffusers names) + ``adapter_modules`` -> device-resident fp16 weights in
the layouts the HIP kernels consume.

Done once at load time (the reference's UNet and adapters are frozen at inference):
  * LoRA merge  W' = W + lora_scale * up @ down   (attention.py:139,146,147,162 / :236,249,250,282)
  * softmax scale d^-0.5 (diffusers Attention.scale) and log2(e) folded into W'_q
  * self-attn q/k/v concatenated to one [3C, C] matrix (one GEMM launch pair)
  * cross-attn text and ID key/value projections concatenated to [2C, Dc] each
  * 3x3 conv weights [Cout, Cin, 3, 3] -> [Cout, 9, Cin] (tap-major, channel-contiguous K axis)
  * GEGLU projection rows interleaved in blocks of 16 (value block, gate block)
  * all ResnetBlock2D.time_emb_proj stacked into one [sum(Cout), 4*C0] matrix
  * cross-attn Wq'/Wo' stay row-major [C, C]: the fused kernel streams 32-deep slabs of them by DMA
All arithmetic for the merge is fp32 on the target device, rounded once to fp16.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import ops
from .unet_spec import UNetConfig, attn_processor_names, walk

LOG2E = 1.4426950408889634


def _h(t: torch.Tensor, dev) -> torch.Tensor:
    return t.to(device=dev, dtype=torch.float16).contiguous()


def _f(t: torch.Tensor, dev) -> torch.Tensor:
    return t.to(device=dev, dtype=torch.float32)


def _conv3(w: torch.Tensor, dev) -> torch.Tensor:
    co, ci, kh, kw = w.shape
    return _h(w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci), dev)


GEGLU_BLOCK = 16   # one MFMA tile of output channels


def _geglu_interleave(t: torch.Tensor) -> torch.Tensor:
    """rows [value(4C) | gate(4C)] -> blocks of 16: v0 g0 v1 g1 ... (a lane of the GEMM then holds
    a value and its gate in the same accumulator slot of two adjacent tiles)"""
    n2 = t.shape[0]
    half = n2 // 2
    v = t[:half].reshape(half // GEGLU_BLOCK, GEGLU_BLOCK, *t.shape[1:])
    g = t[half:].reshape(half // GEGLU_BLOCK, GEGLU_BLOCK, *t.shape[1:])
    return torch.stack([v, g], dim=1).reshape(n2, *t.shape[1:])


def fold_ln(w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """LayerNorm in front of a projection, folded (cid_gemm_desc.ln_s / ln_b):  LN(x) W^T + bias =
    rstd * (x W'^T - mean * s) + b'  with  W' = W diag(gamma) rounded to fp16, s = row sums of the ROUNDED W' (the mean
    term then cancels exactly), b' = W beta + bias.  ``w`` fp32 [N, C].  Returns (W' fp16, s, b') with the two fp32 vectors
    stored as raw bits in fp16-typed tensors (the broadcast arena is one dtype; use ``.view(torch.float32)``)."""
    w = w.float()
    wf = (w * gamma.float()[None, :]).half().contiguous()
    s = wf.float().sum(1).contiguous()
    b = w @ beta.float()
    if bias is not None:
        b = b + bias.float()
    return wf, s.view(torch.float16), b.contiguous().view(torch.float16)


class PackedUNet:
    """Holds every packed tensor; ``w[name]`` lookups use diffusers-style prefixes."""

    def __init__(self, cfg: UNetConfig, unet_sd: Dict[str, torch.Tensor],
                 adapter_sd: Optional[Dict[str, torch.Tensor]], device, lora_scale: float = 1.0,
                 encoder_only: bool = False, keep_base: bool = False):
        """``keep_base``: keep the un-merged attention projection weights (fp16, ~10 % of the model) so that
        ``load_adapter_modules`` can merge a ConsistentID checkpoint's LoRA / ID projections later, in place.
        ``encoder_only``: ``unet_sd`` is a diffusers ControlNetModel state_dict -- the UNet's encoder half plus
        ``controlnet_cond_embedding.*`` / ``controlnet_down_blocks.*`` / ``controlnet_mid_block.*``."""
        self.cfg = cfg
        self.encoder_only = encoder_only
        self._base_attn: Dict[str, torch.Tensor] = {}
        self._lora_scale = lora_scale
        self.device = device
        self.w: Dict[str, torch.Tensor] = {}
        self.temb_offsets: Dict[str, int] = {}
        self.ip_scale: Dict[str, float] = {}
        dev = device
        sd = unet_sd
        W = self.w
        downs, mid, ups = walk(cfg)
        if encoder_only:
            ups = []
        proc_index = {n: i for i, n in enumerate(attn_processor_names(cfg))}

        # ---- ends + time path
        W["conv_in.w"] = _h(sd["conv_in.weight"].permute(0, 2, 3, 1).reshape(sd["conv_in.weight"].shape[0], -1), dev)
        W["conv_in.b"] = _h(sd["conv_in.bias"], dev)
        if not encoder_only:
            W["conv_out.w"] = _h(sd["conv_out.weight"].permute(0, 2, 3, 1).reshape(sd["conv_out.weight"].shape[0], -1), dev)
            W["conv_out.b"] = _h(sd["conv_out.bias"], dev)
            for n in ("conv_norm_out",):
                W[f"{n}.g"], W[f"{n}.b"] = _h(sd[f"{n}.weight"], dev), _h(sd[f"{n}.bias"], dev)
        else:
            # condition embedding: (name, cin, cout, stride, silu) in execution order; zero convs as [C, C] matrices
            self.cond_convs = []
            e = "controlnet_cond_embedding"
            names = [f"{e}.conv_in"]
            i = 0
            while f"{e}.blocks.{i}.weight" in sd:
                names.append(f"{e}.blocks.{i}")
                i += 1
            names.append(f"{e}.conv_out")
            for n in names:
                w = sd[f"{n}.weight"]
                stride = 2 if (n.startswith(f"{e}.blocks.") and int(n.rsplit(".", 1)[1]) % 2 == 1) else 1
                W[f"{n}.w"], W[f"{n}.b"] = _conv3(w, dev), _h(sd[f"{n}.bias"], dev)
                self.cond_convs.append((n, w.shape[1], w.shape[0], stride, n != f"{e}.conv_out"))
            self.n_zero = 0
            while f"controlnet_down_blocks.{self.n_zero}.weight" in sd:
                n = f"controlnet_down_blocks.{self.n_zero}"
                c = sd[f"{n}.weight"].shape[0]
                W[f"{n}.w"], W[f"{n}.b"] = _h(sd[f"{n}.weight"].reshape(c, c), dev), _h(sd[f"{n}.bias"], dev)
                self.n_zero += 1
            c = sd["controlnet_mid_block.weight"].shape[0]
            W["controlnet_mid_block.w"] = _h(sd["controlnet_mid_block.weight"].reshape(c, c), dev)
            W["controlnet_mid_block.b"] = _h(sd["controlnet_mid_block.bias"], dev)
        for n in ("time_embedding.linear_1", "time_embedding.linear_2") + (
                ("add_embedding.linear_1", "add_embedding.linear_2") if cfg.addition_embed_type else ()):
            W[f"{n}.w"], W[f"{n}.b"] = _h(sd[f"{n}.weight"], dev), _h(sd[f"{n}.bias"], dev)

        # ---- resnets (+ stacked time_emb_proj)
        tw, tb, off = [], [], 0
        for blk in downs + [mid] + ups:
            for r in blk.resnets:
                n = r.name
                W[f"{n}.norm1.g"], W[f"{n}.norm1.b"] = _h(sd[f"{n}.norm1.weight"], dev), _h(sd[f"{n}.norm1.bias"], dev)
                W[f"{n}.norm2.g"], W[f"{n}.norm2.b"] = _h(sd[f"{n}.norm2.weight"], dev), _h(sd[f"{n}.norm2.bias"], dev)
                W[f"{n}.conv1.w"], W[f"{n}.conv1.b"] = _conv3(sd[f"{n}.conv1.weight"], dev), _h(sd[f"{n}.conv1.bias"], dev)
                W[f"{n}.conv2.w"], W[f"{n}.conv2.b"] = _conv3(sd[f"{n}.conv2.weight"], dev), _h(sd[f"{n}.conv2.bias"], dev)
                if r.cin != r.cout:
                    W[f"{n}.short.w"] = _h(sd[f"{n}.conv_shortcut.weight"].reshape(r.cout, r.cin), dev)
                    W[f"{n}.short.b"] = _h(sd[f"{n}.conv_shortcut.bias"], dev)
                tw.append(sd[f"{n}.time_emb_proj.weight"]); tb.append(sd[f"{n}.time_emb_proj.bias"])
                self.temb_offsets[n] = off
                off += r.cout
            if blk.sampler:
                n = f"{blk.name}.{blk.sampler}.conv"
                W[f"{n}.w"], W[f"{n}.b"] = _conv3(sd[f"{n}.weight"], dev), _h(sd[f"{n}.bias"], dev)
        self.temb_total = off
        W["temb_all.w"] = _h(torch.cat([t.to(torch.float32) for t in tw], 0), dev)
        W["temb_all.b"] = _h(torch.cat([t.to(torch.float32) for t in tb], 0), dev)

        # ---- transformers
        self.xattn_layers: List[str] = []
        for blk in downs + [mid] + ups:
            for t in blk.attentions:
                n, c = t.name, t.channels
                d = c // t.heads
                W[f"{n}.norm.g"], W[f"{n}.norm.b"] = _h(sd[f"{n}.norm.weight"], dev), _h(sd[f"{n}.norm.bias"], dev)
                for p in ("proj_in", "proj_out"):
                    W[f"{n}.{p}.w"] = _h(sd[f"{n}.{p}.weight"].reshape(c, c), dev)
                    W[f"{n}.{p}.b"] = _h(sd[f"{n}.{p}.bias"], dev)
                for k in range(t.n_layers):
                    b = f"{n}.transformer_blocks.{k}"
                    for ln in ("norm1", "norm2", "norm3"):
                        W[f"{b}.{ln}.g"], W[f"{b}.{ln}.b"] = _h(sd[f"{b}.{ln}.weight"], dev), _h(sd[f"{b}.{ln}.bias"], dev)
                    for k_, v_ in self._attention_weights(b, d, sd, adapter_sd, proc_index, lora_scale,
                                                          ln2=(W[f"{b}.norm2.g"], W[f"{b}.norm2.b"]),
                                                          ln1=(W[f"{b}.norm1.g"], W[f"{b}.norm1.b"])).items():
                        W[k_] = v_
                    W[f"{b}.attn1.out.b"] = _h(sd[f"{b}.attn1.to_out.0.bias"], dev)
                    W[f"{b}.attn2.bo"] = _h(sd[f"{b}.attn2.to_out.0.bias"], dev)
                    self.ip_scale[b] = 1.0 if adapter_sd is not None else 0.0
                    if keep_base:
                        for a in ("attn1", "attn2"):
                            for w_ in ("to_q", "to_k", "to_v", "to_out.0"):
                                self._base_attn[f"{b}.{a}.{w_}.weight"] = _h(sd[f"{b}.{a}.{w_}.weight"], dev)
                    self.xattn_layers.append(b)
                    # feed forward
                    W[f"{b}.ff1.w"] = _h(_geglu_interleave(_f(sd[f"{b}.ff.net.0.proj.weight"], dev)), dev)
                    W[f"{b}.ff1.b"] = _h(_geglu_interleave(_f(sd[f"{b}.ff.net.0.proj.bias"], dev)), dev)
                    # norm3 folded into the GEGLU projection (same row interleave for W', s and b')
                    W[f"{b}.ff1.wl"], W[f"{b}.ff1.lns"], W[f"{b}.ff1.lnb"] = fold_ln(
                        _geglu_interleave(_f(sd[f"{b}.ff.net.0.proj.weight"], dev)), W[f"{b}.norm3.g"], W[f"{b}.norm3.b"],
                        _geglu_interleave(_f(sd[f"{b}.ff.net.0.proj.bias"], dev)))
                    W[f"{b}.ff2.w"] = _h(sd[f"{b}.ff.net.2.weight"], dev)
                    W[f"{b}.ff2.b"] = _h(sd[f"{b}.ff.net.2.bias"], dev)

    def _attention_weights(self, b: str, d: int, sd, adapter_sd, proc_index, lora_scale, ln2=None, ln1=None) -> Dict[str, torch.Tensor]:
        """packed projection weights of transformer block ``b`` (head dim ``d``): LoRA merged in fp32 (attention.py
        :139-162 / :236-282), softmax scale and log2(e) folded into to_q, q/k/v and K/V pairs concatenated.
        ``ln2`` = (gamma, beta) of the block's norm2: where the second-generation fused cross-attention applies
        (cid_id_xattn3_supported), LayerNorm is folded into the query projection (xattn_pack.fold_layernorm); the two
        fp32 fold vectors are stored as raw bits in fp16-typed tensors so that the weight arena stays one dtype."""
        dev = self.device
        out: Dict[str, torch.Tensor] = {}

        def merged(base: str, idx: int, which: str) -> torch.Tensor:
            w = _f(sd[f"{base}.to_{which}.weight" if which != "out" else f"{base}.to_out.0.weight"], dev)
            if adapter_sd is None:
                return w
            up = _f(adapter_sd[f"{idx}.to_{which}_lora.up.weight"], dev)
            down = _f(adapter_sd[f"{idx}.to_{which}_lora.down.weight"], dev)
            return w + lora_scale * (up @ down)

        qscale = (d ** -0.5) * LOG2E
        i1 = proc_index[f"{b}.attn1.processor"]
        qkv = torch.cat([merged(f"{b}.attn1", i1, "q") * qscale, merged(f"{b}.attn1", i1, "k"), merged(f"{b}.attn1", i1, "v")], 0)
        out[f"{b}.attn1.qkv.w"] = _h(qkv, dev)
        if ln1 is not None:     # norm1 folded into the fused q / k / v projection
            out[f"{b}.attn1.qkv.wl"], out[f"{b}.attn1.qkv.lns"], out[f"{b}.attn1.qkv.lnb"] = fold_ln(qkv, ln1[0], ln1[1])
        out[f"{b}.attn1.out.w"] = _h(merged(f"{b}.attn1", i1, "out"), dev)
        i2 = proc_index[f"{b}.attn2.processor"]
        wq2 = merged(f"{b}.attn2", i2, "q") * qscale
        out[f"{b}.attn2.wq"] = _h(wq2, dev)
        if ln2 is not None:     # norm2 folded into the query projection (levels where LN + GEMM + core + GEMM runs)
            out[f"{b}.attn2.wql"], out[f"{b}.attn2.wq_lns"], out[f"{b}.attn2.wq_lnb"] = fold_ln(wq2, ln2[0], ln2[1])
        heads = wq2.shape[0] // d
        if ln2 is not None and ops.id_xattn3_supported(wq2.shape[0], heads, 77, 4):
            from .xattn_pack import fold_layernorm, pack_w3
            wf, qs, qb = fold_layernorm(wq2, ln2[0], ln2[1])
            out[f"{b}.attn2.wq_f"] = wf                                   # (row-major: the comparator build's operand)
            out[f"{b}.attn2.wq_p"] = pack_w3(wf)                          # A-operand streams of the fused kernel
            out[f"{b}.attn2.wo_p"] = pack_w3(_h(merged(f"{b}.attn2", i2, "out"), dev))
            out[f"{b}.attn2.qs"] = qs.view(torch.float16)
            out[f"{b}.attn2.qb"] = qb.view(torch.float16)
        out[f"{b}.attn2.wo"] = _h(merged(f"{b}.attn2", i2, "out"), dev)
        out[f"{b}.attn2.kv_txt.w"] = _h(torch.cat([merged(f"{b}.attn2", i2, "k"), merged(f"{b}.attn2", i2, "v")], 0), dev)
        if adapter_sd is not None:
            kip, vip = adapter_sd[f"{i2}.to_k_ip.weight"], adapter_sd[f"{i2}.to_v_ip.weight"]
        else:  # no adapter: the ID stream is disabled (ip_scale 0); keep shapes valid
            kip, vip = sd[f"{b}.attn2.to_k.weight"], sd[f"{b}.attn2.to_v.weight"]
        out[f"{b}.attn2.kv_ip.w"] = _h(torch.cat([_f(kip, dev), _f(vip, dev)], 0), dev)
        return out

    def load_adapter_modules(self, adapter_sd: Dict[str, torch.Tensor], lora_scale: Optional[float] = None):
        """Merge the ``adapter_modules`` entry of a ConsistentID checkpoint (what the reference does with
        ``ip_layers.load_state_dict(state_dict["adapter_modules"])``, pipline_StableDiffusion_ConsistentID.py:143-144)
        into the packed attention weights IN PLACE: device addresses do not change, so captured step graphs stay
        valid.  Needs ``keep_base=True`` at construction."""
        if not self._base_attn:
            raise RuntimeError("load_adapter_modules needs the base attention weights: construct with keep_base=True")
        from .unet_spec import adapter_param_shapes
        rank = adapter_sd["0.to_q_lora.down.weight"].shape[0] if "0.to_q_lora.down.weight" in adapter_sd else 128
        want = adapter_param_shapes(self.cfg, rank=rank)
        missing = [k for k in want if k not in adapter_sd]
        unexpected = [k for k in adapter_sd if k not in want]
        bad = [k for k in want if k in adapter_sd and tuple(adapter_sd[k].shape) != tuple(want[k])]
        if missing or unexpected or bad:     # the reference loads strict=True
            raise RuntimeError(f"adapter_modules mismatch: missing {missing[:3]} unexpected {unexpected[:3]} shape {bad[:3]}")
        scale = self._lora_scale if lora_scale is None else lora_scale
        downs, mid, ups = walk(self.cfg)
        proc_index = {n: i for i, n in enumerate(attn_processor_names(self.cfg))}
        for blk in downs + [mid] + ups:
            for t in blk.attentions:
                for k in range(t.n_layers):
                    b = f"{t.name}.transformer_blocks.{k}"
                    for name, val in self._attention_weights(b, t.channels // t.heads, self._base_attn, adapter_sd,
                                                             proc_index, scale,
                                                             ln2=(self.w[f"{b}.norm2.g"], self.w[f"{b}.norm2.b"]),
                                                             ln1=(self.w[f"{b}.norm1.g"], self.w[f"{b}.norm1.b"])).items():
                        self.w[name].copy_(val)
                    self.ip_scale[b] = 1.0
        return self

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.w.values())

    # ---- export / import for the one-off RCCL weight broadcast (distributed.broadcast_weights)
    def meta(self) -> dict:
        m = dict(temb_offsets=self.temb_offsets, temb_total=self.temb_total, ip_scale=self.ip_scale,
                 xattn_layers=self.xattn_layers, encoder_only=self.encoder_only, lora_scale=self._lora_scale)
        if self.encoder_only:
            m.update(cond_convs=self.cond_convs, n_zero=self.n_zero)
        return m

    @classmethod
    def from_tensors(cls, cfg: UNetConfig, w: Dict[str, torch.Tensor], meta: dict, device) -> "PackedUNet":
        self = cls.__new__(cls)
        self.cfg, self.device, self.w = cfg, device, w
        self._base_attn, self._lora_scale = {}, meta.get("lora_scale", 1.0)   # (un-merged weights are not broadcast)
        self.temb_offsets, self.temb_total = meta["temb_offsets"], meta["temb_total"]
        self.ip_scale, self.xattn_layers = meta["ip_scale"], meta["xattn_layers"]
        self.encoder_only = meta.get("encoder_only", False)
        if self.encoder_only:
            self.cond_convs, self.n_zero = meta["cond_convs"], meta["n_zero"]
        return self
