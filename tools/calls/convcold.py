"""conv3x3 with COLD weights: rotate through enough distinct weight tensors to exceed L2 + MALL (as inside a denoise step)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from consistentid_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).half()
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
B2 = 8
for label, side, cin, cout in (("L0 320->320", 64, 320, 320), ("L1 640->640", 32, 640, 640), ("L2 1280->1280", 16, 1280, 1280)):
    M = B2 * side * side
    x, b = rnd(M, cin), rnd(cout)
    nbytes = cout * 9 * cin * 2
    nw = max(2, min(200, (600 << 20) // nbytes))
    wts = [rnd(cout, 9 * cin) for _ in range(nw)]
    out = torch.empty(M, cout, dtype=torch.float16, device=dev)
    kw = dict(taps=9, Hi=side, Wi=side, Ho=side, Wo=side)
    for name, rot in (("hot", False), ("cold", True)):
        for i in range(3):
            ops.gemm(x, wts[0], out, M=M, N=cout, c1=cin, bias=b, ws=ws, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 3 * nw if rot else 40
        e0.record()
        for i in range(iters):
            ops.gemm(x, wts[i % nw] if rot else wts[0], out, M=M, N=cout, c1=cin, bias=b, ws=ws, **kw)
        e1.record(); torch.cuda.synchronize()
        print(f"{label:16s} {name:5s} ({nw} weight tensors of {nbytes / 1e6:.1f} MB) {e0.elapsed_time(e1) / iters * 1e3:7.1f} us")
