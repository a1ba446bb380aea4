#!/usr/bin/env python
"""Generate golden input/output vectors for the two attention processors by running the
REAL reference code (/root/reference/attention.py) in the build container.

The reference file imports three third-party symbols that are not installed here:
  diffusers.models.lora.LoRALinearLayer, diffusers.utils.import_utils.is_xformers_available,
  functions.AttentionMLP (functions.py needs cv2).
They are shimmed below (LoRALinearLayer per diffusers==0.23.0: up(down(x)), no alpha;
xformers reported absent; AttentionMLP unused by the processors).  ``attn`` is a minimal
stand-in for diffusers' Attention with exactly the members attention.py:110-294 touches.
Everything the processors compute is therefore the reference's own code path.

Run (only possible where /root/reference exists):  python tests/golden/make_golden.py
Outputs: tests/golden/processors_*.npz (inputs fp16-representable, outputs fp32 computed in fp64).
"""
import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn

REF = Path("/root/reference/attention.py")
OUT = Path(__file__).resolve().parent


def _install_shims():
    class LoRALinearLayer(nn.Module):
        def __init__(self, in_features, out_features, rank=4, network_alpha=None):
            super().__init__()
            self.down = nn.Linear(in_features, rank, bias=False)
            self.up = nn.Linear(rank, out_features, bias=False)
            self.network_alpha = network_alpha
            self.rank = rank

        def forward(self, x):
            y = self.up(self.down(x.to(self.down.weight.dtype)))
            if self.network_alpha is not None:
                y = y * (self.network_alpha / self.rank)
            return y.to(x.dtype)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("diffusers"); mod("diffusers.models"); mod("diffusers.utils")
    mod("diffusers.models.lora", LoRALinearLayer=LoRALinearLayer)
    mod("diffusers.utils.import_utils", is_xformers_available=lambda: False)
    mod("functions", AttentionMLP=type("AttentionMLP", (nn.Module,), {}))


class StandInAttention(nn.Module):
    """Members of diffusers' Attention used by attention.py:110-294."""

    def __init__(self, query_dim, cross_dim, heads):
        super().__init__()
        self.heads = heads
        self.scale = (query_dim // heads) ** -0.5
        self.to_q = nn.Linear(query_dim, query_dim, bias=False)
        self.to_k = nn.Linear(cross_dim or query_dim, query_dim, bias=False)
        self.to_v = nn.Linear(cross_dim or query_dim, query_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim), nn.Dropout(0.0)])
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0

    def prepare_attention_mask(self, mask, *a, **k):
        return mask

    def head_to_batch_dim(self, t):
        b, n, c = t.shape
        h = self.heads
        return t.reshape(b, n, h, c // h).permute(0, 2, 1, 3).reshape(b * h, n, c // h)

    def batch_to_head_dim(self, t):
        bh, n, d = t.shape
        h = self.heads
        return t.reshape(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, d * h)

    def get_attention_scores(self, q, k, attention_mask=None):
        s = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype), q,
                          k.transpose(-1, -2), beta=0, alpha=self.scale)
        return s.softmax(dim=-1)


def _round16(t):
    return t.half().double()


def make_case(tag, B, N, C, heads, Dc, L, rank, seed):
    ref = sys.modules["ref_attention"]
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s, scale=1.0: _round16(torch.randn(*s, generator=g) * scale)
    torch.set_default_dtype(torch.float64)
    try:
        attn1 = StandInAttention(C, None, heads)
        attn2 = StandInAttention(C, Dc, heads)
        p1 = ref.Consistent_AttProcessor(hidden_size=C, cross_attention_dim=None, rank=rank)
        p2 = ref.Consistent_IPAttProcessor(hidden_size=C, cross_attention_dim=Dc, rank=rank, scale=0.8, num_tokens=4)
    finally:
        torch.set_default_dtype(torch.float32)
    store = {}
    with torch.no_grad():
        for prefix, m in (("attn1", attn1), ("attn2", attn2), ("proc1", p1), ("proc2", p2)):
            for name, prm in m.named_parameters():
                fan_in = prm.shape[-1]
                s = 3.0 / fan_in ** 0.5 if name.startswith(("to_q", "to_k.")) else 1.0 / fan_in ** 0.5
                if "lora.up" in name:
                    s = 0.05
                prm.copy_(rnd(*prm.shape, scale=s))
                store[f"{prefix}.{name}"] = prm.detach().float().numpy()
        hidden = rnd(B, N, C)
        ehs = rnd(B, L, Dc)
        out1 = p1(attn1, hidden)
        out2 = p2(attn2, hidden, encoder_hidden_states=ehs)
    store.update(hidden=hidden.float().numpy(), ehs=ehs.float().numpy(),
                 out_self=out1.float().numpy(), out_ip=out2.float().numpy(),
                 meta=np.array([B, N, C, heads, Dc, L, rank], dtype=np.int64), ip_scale=np.float32(0.8))
    np.savez_compressed(OUT / f"processors_{tag}.npz", **store)
    print(tag, "self", float(out1.abs().mean()), "ip", float(out2.abs().mean()))


def make_real_case(tag, B, N, C, heads, Dc, rank, seed, L=81):
    """The real SD1.5 / SDXL widths and head counts.  Weights come from tests/oracle_utils.seeded_processor_weights
    (seed stored), so the file holds only inputs and the reference's outputs."""
    for q in (str(OUT.parent), str(OUT.parent.parent)):      # tests/ (oracle_utils) and the repository root
        if q not in sys.path:
            sys.path.insert(0, q)
    from oracle_utils import seeded_processor_weights
    ref = sys.modules["ref_attention"]
    torch.set_default_dtype(torch.float64)
    try:
        attn1 = StandInAttention(C, None, heads)
        attn2 = StandInAttention(C, Dc, heads)
        p1 = ref.Consistent_AttProcessor(hidden_size=C, cross_attention_dim=None, rank=rank)
        p2 = ref.Consistent_IPAttProcessor(hidden_size=C, cross_attention_dim=Dc, rank=rank, scale=0.8, num_tokens=4)
    finally:
        torch.set_default_dtype(torch.float32)
    seeded_processor_weights({"attn1": attn1, "attn2": attn2, "proc1": p1, "proc2": p2}, seed)
    g = torch.Generator().manual_seed(seed + 7)
    hidden = _round16(torch.randn(B, N, C, generator=g))
    ehs = _round16(torch.randn(B, L, Dc, generator=g))
    with torch.no_grad():
        out1 = p1(attn1, hidden)
        out2 = p2(attn2, hidden, encoder_hidden_states=ehs)
    np.savez_compressed(OUT / f"processors_real_{tag}.npz", hidden=hidden.half().numpy(), ehs=ehs.half().numpy(),
                        out_self=out1.float().numpy(), out_ip=out2.float().numpy(), seed=np.int64(seed),
                        meta=np.array([B, N, C, heads, Dc, L, rank], dtype=np.int64), ip_scale=np.float32(0.8))
    print(tag, "self", float(out1.abs().mean()), "ip", float(out2.abs().mean()))


def main():
    _install_shims()
    spec = importlib.util.spec_from_file_location("ref_attention", REF)
    m = importlib.util.module_from_spec(spec)
    sys.modules["ref_attention"] = m
    spec.loader.exec_module(m)
    make_case("c64_h2", B=2, N=256, C=64, heads=2, Dc=64, L=81, rank=8, seed=0)
    make_case("c128_h2", B=1, N=128, C=128, heads=2, Dc=128, L=81, rank=4, seed=1)
    # the widths / head counts the UNets really use (SD1.5 d = 40 / 80 / 160, SDXL d = 64 with 10 / 20 heads)
    make_real_case("c320_h8", B=1, N=128, C=320, heads=8, Dc=768, rank=16, seed=10)
    make_real_case("c640_h8", B=1, N=64, C=640, heads=8, Dc=768, rank=16, seed=11)
    make_real_case("c1280_h8", B=1, N=64, C=1280, heads=8, Dc=768, rank=16, seed=12)
    make_real_case("c640_h10", B=1, N=64, C=640, heads=10, Dc=2048, rank=16, seed=13)
    make_real_case("c1280_h20", B=1, N=64, C=1280, heads=20, Dc=2048, rank=16, seed=14)


if __name__ == "__main__":
    main()
