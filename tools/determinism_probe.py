#!/usr/bin/env python
"""Run every hot kernel twice on the same inputs and compare bit for bit (a data race shows up as run-to-run noise)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistentid_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def rnd(*s, seed=0, scale=0.5):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(*s, generator=g, device=dev) * scale).half()


def attn(N, c, heads, B=2, reps=6):
    d = c // heads
    x, w = rnd(B * N, c, seed=1), rnd(3 * c, c, seed=2, scale=0.08)
    qk = torch.empty(B * N, 2 * c, dtype=torch.float16, device=dev)
    vt = torch.empty(B * heads * ops.dvp_of(d) * N, dtype=torch.float16, device=dev)
    ops.gemm(x, w, qk, M=B * N, N=3 * c, c1=c, mode=2, vt=vt, n_vt0=2 * c, heads=heads, dhead=d, ntok=N)
    outs = []
    for _ in range(reps):
        o = torch.empty(B * N, c, dtype=torch.float16, device=dev)
        ops.self_attn(qk, qk[:, c:], vt, o, B=B, N=N, heads=heads, d=d, ldq=2 * c, ldk=2 * c, ldo=c)
        outs.append(o)
    torch.cuda.synchronize()
    bad = sum(not torch.equal(outs[0], o) for o in outs[1:])
    md = max((outs[0].float() - o.float()).abs().max().item() for o in outs[1:])
    print(f"self_attn N={N} d={d}: {bad}/{reps - 1} reruns differ, max diff {md:.3g}")


for N, c, h in ((4096, 320, 8), (1024, 640, 8), (256, 1280, 8), (64, 1280, 8), (1024, 640, 10), (1024, 1280, 20), (256, 64, 2)):
    attn(N, c, h)
