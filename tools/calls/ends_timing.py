#!/usr/bin/env python
"""conv_in / conv_out at the SD1.5 step's shape (CFG batch 8 of 4 samples, 64 x 64 latents), HIP-event timing, back to back."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from consistentid_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).half()
B, Bin, H, W, c = 8, 4, 64, 64, 320
lat, w, b = rnd(Bin, 4, H, W), rnd(c, 36), rnd(c)
out = torch.empty(B * H * W, c, dtype=torch.float16, device=dev)
sc = torch.ones(1, device=dev)
t = timeit(lambda: ops.conv_in(lat, out, w, b, B=B, Bin=Bin, cin=4, H=H, W=W, cout=c, in_scale=sc), iters=50)
print(f"conv_in  B={B} (Bin={Bin}) 4 -> {c}: {t * 1e6:6.1f} us  ({out.numel() * 2 / t / 1e9:.0f} GB/s written)")
x, w2, b2 = rnd(B * H * W, c), rnd(4, 9 * c), rnd(4)
o2 = torch.empty(B, 4, H, W, dtype=torch.float16, device=dev)
t = timeit(lambda: ops.conv_out(x, o2, w2, b2, B=B, H=H, W=W, cin=c, cout=4), iters=50)
print(f"conv_out B={B} {c} -> 4: {t * 1e6:6.1f} us  ({x.numel() * 2 / t / 1e9:.0f} GB/s read)")
