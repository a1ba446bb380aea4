"""Oracle (test infrastructure): restatement of the reference's two attention
processors and their wiring.

  Consistent_AttProcessor    <- /root/reference/attention.py:90-174
  Consistent_IPAttProcessor  <- /root/reference/attention.py:177-294
  set_ip_adapter             <- /root/reference/pipline_StableDiffusion_ConsistentID.py:152-174

PINNED: tests/golden/make_golden.py runs the real reference classes (imported from
/root/reference with shims for diffusers' LoRALinearLayer / xformers probe) and
tests/test_oracle_golden.py replays the stored vectors through these classes.

Constructor arguments, attribute names and state_dict keys are the reference's
(``to_{q,k,v,out}_lora.{down,up}.weight``, ``to_{k,v}_ip.weight``).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .unet import Attention, LoRALinearLayer, UNet2DConditionModel


def _sdpa(q, k, v, scale):
    """softmax(q k^T * scale) v with [B, heads, n, d] operands (ref :259-261, :272-274;
    identical to get_attention_scores + bmm of :157-158)."""
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    return torch.matmul(s.softmax(dim=-1), v)


class Consistent_AttProcessor(nn.Module):
    """Self-attention with LoRA on q/k/v/out (attention.py:90-174)."""

    def __init__(self, hidden_size=None, cross_attention_dim=None, rank=4, network_alpha=None,
                 lora_scale=1.0):
        super().__init__()
        self.rank = rank
        self.lora_scale = lora_scale
        kv_in = cross_attention_dim or hidden_size
        self.to_q_lora = LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)
        self.to_k_lora = LoRALinearLayer(kv_in, hidden_size, rank, network_alpha)
        self.to_v_lora = LoRALinearLayer(kv_in, hidden_size, rank, network_alpha)
        self.to_out_lora = LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)

    def __call__(self, attn: Attention, hidden_states, encoder_hidden_states=None,
                 attention_mask=None, temb=None):
        assert hidden_states.ndim == 3 and attn.spatial_norm is None and attn.group_norm is None
        x = hidden_states
        ctx = x if encoder_hidden_states is None else encoder_hidden_states  # :141-144
        ls = self.lora_scale
        q = attn.to_q(x) + ls * self.to_q_lora(x)                            # :139
        k = attn.to_k(ctx) + ls * self.to_k_lora(ctx)                        # :146
        v = attn.to_v(ctx) + ls * self.to_v_lora(ctx)                        # :147
        b, n, c = q.shape
        h = attn.heads
        split = lambda t: t.reshape(b, -1, h, c // h).transpose(1, 2)        # :149-151
        o = _sdpa(split(q), split(k), split(v), attn.scale)                  # :157-158
        o = o.transpose(1, 2).reshape(b, n, c)                               # :159
        o = attn.to_out[0](o) + ls * self.to_out_lora(o)                     # :162
        o = attn.to_out[1](o)                                                # :164 (dropout 0)
        if attn.residual_connection:                                         # :169
            o = o + hidden_states
        return o / attn.rescale_output_factor                                # :172


class Consistent_IPAttProcessor(nn.Module):
    """Cross-attention with a second (ID) key/value stream: two independently
    normalised softmaxes whose outputs are summed (attention.py:177-294)."""

    def __init__(self, hidden_size, cross_attention_dim=None, rank=4, network_alpha=None,
                 lora_scale=1.0, scale=1.0, num_tokens=4):
        super().__init__()
        self.rank = rank
        self.lora_scale = lora_scale
        self.num_tokens = num_tokens
        kv_in = cross_attention_dim or hidden_size
        self.to_q_lora = LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)
        self.to_k_lora = LoRALinearLayer(kv_in, hidden_size, rank, network_alpha)
        self.to_v_lora = LoRALinearLayer(kv_in, hidden_size, rank, network_alpha)
        self.to_out_lora = LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        self.scale = scale
        self.to_k_ip = nn.Linear(kv_in, hidden_size, bias=False)
        self.to_v_ip = nn.Linear(kv_in, hidden_size, bias=False)

    def __call__(self, attn: Attention, hidden_states, encoder_hidden_states=None,
                 attention_mask=None, scale=1.0, temb=None):
        assert hidden_states.ndim == 3 and attn.spatial_norm is None and attn.group_norm is None
        x = hidden_states
        ls = self.lora_scale
        q = attn.to_q(x) + ls * self.to_q_lora(x)                            # :236
        end = encoder_hidden_states.shape[1] - self.num_tokens               # :241
        txt, ip = encoder_hidden_states[:, :end], encoder_hidden_states[:, end:]  # :242-245
        k = attn.to_k(txt) + ls * self.to_k_lora(txt)                        # :249
        v = attn.to_v(txt) + ls * self.to_v_lora(txt)                        # :250
        b, n, c = q.shape
        h = attn.heads
        split = lambda t: t.reshape(b, -1, h, c // h).transpose(1, 2)        # :255-257
        qh = split(q)
        o_txt = _sdpa(qh, split(k), split(v), attn.scale)                    # :259-261
        o_ip = _sdpa(qh, split(self.to_k_ip(ip)), split(self.to_v_ip(ip)), attn.scale)  # :266-274
        o = (o_txt + self.scale * o_ip).transpose(1, 2).reshape(b, n, c)     # :263,276,279
        o = attn.to_out[0](o) + ls * self.to_out_lora(o)                     # :282
        o = attn.to_out[1](o)                                                # :284
        if attn.residual_connection:                                         # :289
            o = o + hidden_states
        return o / attn.rescale_output_factor                                # :292


def set_ip_adapter(unet: UNet2DConditionModel, lora_rank: int = 128, num_tokens: int = 4):
    """pipline_StableDiffusion_ConsistentID.py:152-174: attn1 -> Consistent_AttProcessor,
    everything else -> Consistent_IPAttProcessor; hidden_size from the block name."""
    cfg = unet.config
    procs = {}
    for name in unet.attn_processors.keys():
        cross = None if name.endswith("attn1.processor") else cfg.cross_attention_dim
        if name.startswith("mid_block"):
            hidden = cfg.block_out_channels[-1]
        elif name.startswith("up_blocks"):
            hidden = list(reversed(cfg.block_out_channels))[int(name[len("up_blocks.")])]
        else:
            hidden = cfg.block_out_channels[int(name[len("down_blocks.")])]
        if cross is None:
            procs[name] = Consistent_AttProcessor(hidden_size=hidden, cross_attention_dim=None,
                                                  rank=lora_rank)
        else:
            procs[name] = Consistent_IPAttProcessor(hidden_size=hidden, cross_attention_dim=cross,
                                                    scale=1.0, rank=lora_rank, num_tokens=num_tokens)
    unet.set_attn_processor(procs)
    return procs


def adapter_modules(unet: UNet2DConditionModel) -> nn.ModuleList:
    """The ``ModuleList(unet.attn_processors.values())`` whose state_dict is the
    checkpoint's ``adapter_modules`` entry (ref :143-144)."""
    return nn.ModuleList(list(unet.attn_processors.values()))
