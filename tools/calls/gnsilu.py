import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.kbench import timeit
from consistentid_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).half()
B2 = 8
gws = torch.zeros(ops.groupnorm_ws_bytes(B2, 2560), dtype=torch.uint8, device=dev)
for side, c in ((64, 320), (32, 640)):
    HW = side * side
    x = rnd(B2 * HW, c); gm, bt = rnd(c), rnd(c); out = torch.empty_like(x)
    x._gn_stats = (torch.rand(B2 * HW // 256, 32, 2, device=dev) + 1.0, 256)
    for silu in (True, False):
        t = timeit(lambda: ops.groupnorm(x, out, gm, bt, gws, B=B2, HW=HW, c1=c, silu=silu), iters=50)
        print(f"gn_apply (epilogue stats) HW={HW} C={c} silu={silu}: {t*1e6:6.1f} us  {2*x.numel()*2/t/1e9:7.1f} GB/s")
    y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x), iters=50)
    print(f"   torch copy_ same bytes: {t*1e6:6.1f} us")
    t = timeit(lambda: torch.nn.functional.silu(x), iters=50)
    print(f"   torch silu (alloc + kernel): {t*1e6:6.1f} us")
