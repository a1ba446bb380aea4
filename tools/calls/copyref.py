import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.kbench import timeit
dev = torch.device("cuda:0")
for M, C in ((32768, 320), (8192, 640), (2048, 1280), (32768, 960), (262144, 320)):
    x = torch.randn(M, C, device=dev).half(); y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x), iters=50)
    t2 = timeit(lambda: torch.add(x, 1.0, out=y), iters=50)
    print(f"copy [{M} x {C}] fp16 ({2*M*C*2/1e6:.1f} MB moved): copy_ {t*1e6:6.1f} us = {2*M*C*2/t/1e9:7.1f} GB/s   add {t2*1e6:6.1f} us = {2*M*C*2/t2/1e9:7.1f} GB/s")
