#!/usr/bin/env python
"""Ablation timing of the GEMM kernels (experiment build: python -m consistentid_amd.build --variant abl CID_GEMM_ABLATION,
CID_LIBRARY=consistentid_amd/libcid_abl.so).  CID_GEMM_ABLATE bits: 1 no DMA in the loop, 2 no MFMA, 4 no halo DMA,
8 no halo fragment reads, 16 no plain-epilogue traffic, 32 GEGLU without the erf.  One process per bit set (the knob is read once)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from consistentid_amd import ops
sys.path.insert(0, %r + "/tools")
from kbench import timeit
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).half()
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
B2 = 8
res = []
for label, side, cin, cout, taps in (("conv3 L0 320", 64, 320, 320, 9), ("conv3 L1 640", 32, 640, 640, 9), ("conv3 L2 1280", 16, 1280, 1280, 9),
                                     ("lin L0 320", 64, 320, 320, 1), ("ff2 L0", 64, 1280, 320, 1)):
    M = B2 * side * side
    x, w, b = rnd(M, cin), rnd(cout, taps * cin), rnd(cout)
    out = torch.empty(M, cout, dtype=torch.float16, device=dev)
    kw = dict(taps=9, Hi=side, Wi=side, Ho=side, Wo=side) if taps == 9 else {}
    res.append((label, timeit(lambda: ops.gemm(x, w, out, M=M, N=cout, c1=cin, bias=b, ws=ws, **kw)) * 1e6))
    if taps == 1 and cin == cout:
        r_ = rnd(M, cout)
        res.append((label + "+res", timeit(lambda: ops.gemm(x, w, out, M=M, N=cout, c1=cin, bias=b, res=r_, ldr=cout, ws=ws, **kw)) * 1e6))
for label, side, c in (("geglu L0", 64, 320), ("geglu L1", 32, 640)):
    M = B2 * side * side
    x, w, b = rnd(M, c), rnd(8 * c, c), rnd(8 * c)
    out = torch.empty(M, 4 * c, dtype=torch.float16, device=dev)
    res.append((label, timeit(lambda: ops.gemm(x, w, out, M=M, N=8 * c, c1=c, bias=b, mode=1)) * 1e6))
print(" ".join(f"{l}={t:.1f}" for l, t in res))
''' % (ROOT, ROOT)

for bits in [int(b) for b in sys.argv[1:]] or (0, 1, 2, 3, 4, 8, 12, 16, 32, 2 | 32, 1 | 2 | 32 | 16):
    env = dict(os.environ, CID_GEMM_ABLATE=str(bits), CID_LIBRARY=os.path.join(ROOT, "consistentid_amd", "libcid_abl.so"))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"ablate={bits:2d}: {r.stdout.strip()} {r.stderr.strip()[-200:] if r.returncode else ''}", flush=True)
