"""conv3x3 variants (bias only / + time row / + residual / + statistics) at the SD1.5 levels: where does the in-step time go?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from consistentid_amd import ops
from tools.kbench import timeit
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).half()
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
B2 = 8
for label, side, cin, cout in (("L0 320->320", 64, 320, 320), ("L1 640->640", 32, 640, 640), ("L2 1280->1280", 16, 1280, 1280), ("L0 B4 320->320", 64, 320, 320)):
    b2 = 4 if "B4" in label else B2
    M = b2 * side * side
    x, w, b = rnd(M, cin), rnd(cout, 9 * cin), rnd(cout)
    res, temb = rnd(M, cout), rnd(b2, cout)
    out = torch.empty(M, cout, dtype=torch.float16, device=dev)
    kw = dict(taps=9, Hi=side, Wi=side, Ho=side, Wo=side)
    base = dict(M=M, N=cout, c1=cin, bias=b, ws=ws, **kw)
    for name, extra in (("bias", {}), ("+temb", dict(rowbias=temb, ld_rowbias=cout, rows_per_sample=side * side)),
                        ("+res", dict(res=res, ldr=cout)), ("+stats", dict(gn_hw=side * side)),
                        ("+temb+stats", dict(rowbias=temb, ld_rowbias=cout, rows_per_sample=side * side, gn_hw=side * side)),
                        ("+res+stats", dict(res=res, ldr=cout, gn_hw=side * side))):
        t = timeit(lambda: ops.gemm(x, w, out, **base, **extra))
        print(f"{label:18s} {name:12s} {t * 1e6:7.1f} us")
