"""The reference's attention-processor plugin boundary, backed by the HIP kernels.

  Consistent_AttProcessor    mirrors /root/reference/attention.py:90-174
  Consistent_IPAttProcessor  mirrors /root/reference/attention.py:177-294

Same constructor arguments, same public attributes (``scale``, ``num_tokens``,
``lora_scale``, ``rank``, ``hidden_size``, ``cross_attention_dim``), same ``state_dict``
keys (``to_{q,k,v,out}_lora.{down,up}.weight``, ``to_{k,v}_ip.weight``) and the same call
protocol ``proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, ...)``
where ``attn`` is a diffusers-style ``Attention`` module (``to_q/to_k/to_v`` bias-free
Linear, ``to_out[0]`` Linear with bias, ``heads``, ``scale``).  They can therefore be
installed with ``unet.set_attn_processor`` exactly like the reference's
(pipline_StableDiffusion_ConsistentID.py:152-174) and loaded with
``ModuleList(procs).load_state_dict(ckpt["adapter_modules"], strict=True)`` (:143-144).

Inputs must be fp16 CUDA tensors; there is no CPU path.  The frozen weights are merged
(W + lora_scale * up @ down) and packed on first use and cached until a parameter changes.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .weights import LOG2E


class LoRALinearLayer(nn.Module):
    """Parameter holder with diffusers' LoRALinearLayer layout (attention.py:4)."""

    def __init__(self, in_features, out_features, rank=4, network_alpha=None):
        super().__init__()
        self.down = nn.Linear(in_features, rank, bias=False)
        self.up = nn.Linear(rank, out_features, bias=False)
        self.network_alpha = network_alpha
        self.rank = rank
        nn.init.normal_(self.down.weight, std=1 / rank)
        nn.init.zeros_(self.up.weight)

    def delta(self) -> torch.Tensor:
        d = self.up.weight.float() @ self.down.weight.float()
        if self.network_alpha is not None:
            d = d * (self.network_alpha / self.rank)
        return d


def _ver(*params):
    return tuple((p.data_ptr(), p._version) for p in params)


def _check_unsupported(attn, hidden_states, attention_mask):
    if hidden_states.ndim != 3:
        raise NotImplementedError("4-D hidden_states branch (attention.py:126-128) is dead for UNet transformer blocks")
    if attention_mask is not None:
        raise NotImplementedError("attention_mask is never passed on the ConsistentID hot path")
    for name in ("spatial_norm", "group_norm", "norm_cross"):
        if getattr(attn, name, None):
            raise NotImplementedError(f"attn.{name} is not used by the SD / SDXL UNet (attention.py:120-136)")


class Consistent_AttProcessor(nn.Module):
    def __init__(self, hidden_size=None, cross_attention_dim=None, rank=4, network_alpha=None, lora_scale=1.0):
        super().__init__()
        self.rank = rank
        self.lora_scale = lora_scale
        kv_in = cross_attention_dim or hidden_size
        self.to_q_lora = LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)
        self.to_k_lora = LoRALinearLayer(kv_in, hidden_size, rank, network_alpha)
        self.to_v_lora = LoRALinearLayer(kv_in, hidden_size, rank, network_alpha)
        self.to_out_lora = LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)
        self._cache_key = None
        self._w = None

    def _weights(self, attn):
        ps = [attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, attn.to_out[0].weight,
              self.to_q_lora.up.weight, self.to_q_lora.down.weight, self.to_k_lora.up.weight,
              self.to_k_lora.down.weight, self.to_v_lora.up.weight, self.to_v_lora.down.weight,
              self.to_out_lora.up.weight, self.to_out_lora.down.weight]
        key = (_ver(*ps), self.lora_scale)
        if key != self._cache_key:
            ls = self.lora_scale
            c = attn.to_q.weight.shape[0]
            d = c // attn.heads
            wq = (attn.to_q.weight.float() + ls * self.to_q_lora.delta()) * (float(attn.scale) * LOG2E)
            wk = attn.to_k.weight.float() + ls * self.to_k_lora.delta()
            wv = attn.to_v.weight.float() + ls * self.to_v_lora.delta()
            wo = attn.to_out[0].weight.float() + ls * self.to_out_lora.delta()
            self._w = (torch.cat([wq, wk, wv], 0).half().contiguous(), wo.half().contiguous(),
                       attn.to_out[0].bias.detach().half().contiguous())
            self._cache_key = key
        return self._w

    @torch.no_grad()
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        _check_unsupported(attn, hidden_states, attention_mask)
        if encoder_hidden_states is not None:
            raise NotImplementedError("Consistent_AttProcessor is installed on attn1 (self-attention) only "
                                      "(pipline_StableDiffusion_ConsistentID.py:156,165)")
        x = hidden_states.contiguous()
        B, N, c = x.shape
        heads = attn.heads
        d = c // heads
        wqkv, wo, bo = self._weights(attn)
        M = B * N
        dev = x.device
        qk = torch.empty(M, 2 * c, dtype=torch.float16, device=dev)
        vt = torch.empty(B * heads * ops.dvp_of(d) * N, dtype=torch.float16, device=dev)
        ops.gemm(x, wqkv, qk, M=M, N=3 * c, c1=c, mode=2, vt=vt, n_vt0=2 * c, heads=heads, dhead=d, ntok=N)
        ao = torch.empty(M, c, dtype=torch.float16, device=dev)
        ops.self_attn(qk, qk[:, c:], vt, ao, B=B, N=N, heads=heads, d=d, ldq=2 * c, ldk=2 * c, ldo=c)
        out = torch.empty(B, N, c, dtype=torch.float16, device=dev)
        res = x if getattr(attn, "residual_connection", False) else None
        ops.gemm(ao, wo, out, M=M, N=c, c1=c, bias=bo, res=res, ldr=c)
        f = getattr(attn, "rescale_output_factor", 1.0)
        return out if f == 1.0 else out / f


class Consistent_IPAttProcessor(nn.Module):
    def __init__(self, hidden_size, cross_attention_dim=None, rank=4, network_alpha=None, lora_scale=1.0,
                 scale=1.0, num_tokens=4):
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
        self._cache_key = None
        self._w = None
        self._kv_key = None
        self._kv = None

    def _weights(self, attn):
        ps = [attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, attn.to_out[0].weight,
              self.to_q_lora.up.weight, self.to_q_lora.down.weight, self.to_k_lora.up.weight,
              self.to_k_lora.down.weight, self.to_v_lora.up.weight, self.to_v_lora.down.weight,
              self.to_out_lora.up.weight, self.to_out_lora.down.weight, self.to_k_ip.weight, self.to_v_ip.weight]
        key = (_ver(*ps), self.lora_scale)
        if key != self._cache_key:
            ls = self.lora_scale
            wq = (attn.to_q.weight.float() + ls * self.to_q_lora.delta()) * (float(attn.scale) * LOG2E)
            wk = attn.to_k.weight.float() + ls * self.to_k_lora.delta()
            wv = attn.to_v.weight.float() + ls * self.to_v_lora.delta()
            wo = attn.to_out[0].weight.float() + ls * self.to_out_lora.delta()
            self._w = dict(
                wq=wq.half().contiguous(), wo=wo.half().contiguous(),
                bo=attn.to_out[0].bias.detach().half().contiguous(),
                kv_txt=torch.cat([wk, wv], 0).half().contiguous(),
                kv_ip=torch.cat([self.to_k_ip.weight.float(), self.to_v_ip.weight.float()], 0).half().contiguous())
            self._cache_key = key
            self._kv_key = None
        return self._w

    @torch.no_grad()
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, scale=1.0, temb=None):
        _check_unsupported(attn, hidden_states, attention_mask)
        if encoder_hidden_states is None:
            raise NotImplementedError("Consistent_IPAttProcessor is installed on attn2 (cross-attention) only")
        x = hidden_states.contiguous()
        B, N, c = x.shape
        heads = attn.heads
        w = self._weights(attn)
        ehs = encoder_hidden_states.contiguous()
        R, L, Dc = ehs.shape
        assert R == B
        n_ip = self.num_tokens
        n_txt = L - n_ip                                         # attention.py:241
        # the one-launch fused kernel where its geometry applies (SD1.5 level 0), else the first-generation one
        gen = ops.xattn_generation()
        v3 = gen >= 3 and N % 64 == 0 and ops.id_xattn3_supported(c, heads, n_txt, n_ip)
        kvk = (ehs.data_ptr(), ehs._version, tuple(ehs.shape), self._cache_key, v3)
        # (the keyed tensor is kept in _kv[3]: a live tensor's address cannot be recycled for another prompt's embeddings;
        # a caller that passes a fresh temporary every step simply recomputes K/V every step, like the reference does)
        if kvk != self._kv_key or self._kv[3] is not ehs:
            dev = x.device
            kv_txt = torch.empty(R * L, 2 * c, dtype=torch.float16, device=dev)
            kv_ip = torch.empty(R * L, 2 * c, dtype=torch.float16, device=dev)
            ops.gemm(ehs, w["kv_txt"], kv_txt, M=R * L, N=2 * c, c1=Dc)       # :249-250
            ops.gemm(ehs, w["kv_ip"], kv_ip, M=R * L, N=2 * c, c1=Dc)         # :266-267
            ke, ve = ops.kv_pack2_elems(c, heads) if v3 else ops.kv_pack_elems(c, heads)
            kp = torch.empty(R * ke, dtype=torch.float16, device=dev)
            vp = torch.empty(R * ve, dtype=torch.float16, device=dev)
            if v3:
                ops.kv_pack2(kv_txt, kv_ip, kp, vp, R=R, L=L, C_=c, heads=heads, n_txt=n_txt, n_ip=n_ip, order="reg")
            else:
                ops.kv_pack(kv_txt, kv_ip, kp, vp, R=R, C_=c, heads=heads, n_txt=n_txt, n_ip=n_ip)
            self._kv = (kp, vp, torch.arange(R, dtype=torch.int32, device=dev), ehs)   # (ehs: see _kv_key below)
            self._kv_key = kvk
        kp, vp, kvrow = self._kv[:3]
        out = torch.empty_like(x)
        has_res = bool(getattr(attn, "residual_connection", False))
        if v3 and "zeros" not in w:
            w["zeros"] = torch.zeros(c, dtype=torch.float32, device=x.device)
        if v3:
            if "wq_p" not in w:
                from .xattn_pack import pack_w3
                w["wq_p"], w["wo_p"] = pack_w3(w["wq"]), pack_w3(w["wo"])
            ops.id_xattn3(x, out, wq_p=w["wq_p"], q_rowsum=w["zeros"], q_bias=w["zeros"], wo_p=w["wo_p"], bo=w["bo"], kp=kp,
                          vp=vp, kvrow=kvrow, B=B, N=N, C_=c, heads=heads, n_txt=n_txt, n_ip=n_ip,
                          ip_scale=float(self.scale), has_ln=False, add_residual=has_res)
        else:
            ops.id_xattn(x, out, wq=w["wq"], wo=w["wo"], bo=w["bo"], kp=kp, vp=vp, kvrow=kvrow, B=B, N=N, C_=c,
                         heads=heads, n_txt=n_txt, n_ip=n_ip, ip_scale=float(self.scale), residual=x if has_res else None)
        f = getattr(attn, "rescale_output_factor", 1.0)
        return out if f == 1.0 else out / f
