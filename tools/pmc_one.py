#!/usr/bin/env python
"""Run ONE kernel shape a few times (for rocprofv3 --pmc passes).
  python tools/pmc_one.py conv0|lin0|ff20|qkv0|lin2|geglu0|geglu2|attn0|xattn0|xattn3 [iters]
  (geglu2 = the GEGLU projection of SD1.5's 16 x 16 level: csrc/linear_h32.hip, or gemm.hip's form with CID_GEGLU_H32=0)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistentid_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "conv0"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).half()
B2 = 8
if which in ("conv0", "conv2", "conv3"):
    side, c = {"conv0": (64, 320), "conv2": (16, 1280), "conv3": (8, 1280)}[which]
    M = B2 * side * side
    x, w, b = rnd(M, c), rnd(c, 9 * c), rnd(c)
    out = torch.empty(M, c, dtype=torch.float16, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    fn = lambda: ops.gemm(x, w, out, M=M, N=c, c1=c, bias=b, taps=9, Hi=side, Wi=side, Ho=side, Wo=side, ws=ws)
elif which == "lin0":
    M, c = B2 * 4096, 320
    x, w, b = rnd(M, c), rnd(c, c), rnd(c)
    out = torch.empty(M, c, dtype=torch.float16, device=dev)
    fn = lambda: ops.gemm(x, w, out, M=M, N=c, c1=c, bias=b)
elif which == "ff20":
    M, c = B2 * 4096, 320
    x, w, b, r = rnd(M, 4 * c), rnd(c, 4 * c), rnd(c), rnd(M, c)
    out = torch.empty(M, c, dtype=torch.float16, device=dev)
    fn = lambda: ops.gemm(x, w, out, M=M, N=c, c1=4 * c, bias=b, res=r, ldr=c)
elif which == "lin2":
    M, c = B2 * 256, 1280
    x, w, b, r = rnd(M, c), rnd(c, c), rnd(c), rnd(M, c)
    out = torch.empty(M, c, dtype=torch.float16, device=dev)
    fn = lambda: ops.gemm(x, w, out, M=M, N=c, c1=c, bias=b, res=r, ldr=c)
elif which == "qkv0":
    N, c, heads = 4096, 320, 8
    M, d = B2 * N, c // heads
    x, w = rnd(M, c), rnd(3 * c, c)
    qk = torch.empty(M, 2 * c, dtype=torch.float16, device=dev)
    vt = torch.empty(B2 * heads * ops.dvp_of(d) * N, dtype=torch.float16, device=dev)
    fn = lambda: ops.gemm(x, w, qk, M=M, N=3 * c, c1=c, mode=2, vt=vt, n_vt0=2 * c, heads=heads, dhead=d, ntok=N)
elif which in ("geglu0", "geglu2"):
    M, c = (B2 * 4096, 320) if which == "geglu0" else (B2 * 256, 1280)
    x, w, b = rnd(M, c), rnd(8 * c, c), rnd(8 * c)
    out = torch.empty(M, 4 * c, dtype=torch.float16, device=dev)
    fn = lambda: ops.gemm(x, w, out, M=M, N=8 * c, c1=c, bias=b, mode=1)
elif which == "attn0":
    N, c, heads = 4096, 320, 8
    d = c // heads
    qk, vt = rnd(B2 * N, 2 * c), rnd(B2 * heads * ops.dvp_of(d) * N)
    out = torch.empty(B2 * N, c, dtype=torch.float16, device=dev)
    fn = lambda: ops.self_attn(qk, qk[:, c:], vt, out, B=B2, N=N, heads=heads, d=d, ldq=2 * c, ldk=2 * c, ldo=c)
elif which == "xattn0":
    N, c, heads = 4096, 320, 8
    x = rnd(B2, N, c)
    out = torch.empty_like(x)
    wq, wo, bo = rnd(c, c), rnd(c, c), rnd(c)
    ke, ve = ops.kv_pack_elems(c, heads)
    kp, vp = rnd(B2 * ke), rnd(B2 * ve)
    kvrow = torch.arange(B2, dtype=torch.int32, device=dev)
    lg, lb = rnd(c), rnd(c)
    fn = lambda: ops.id_xattn(x, out, wq=wq, wo=wo, bo=bo, kp=kp, vp=vp, kvrow=kvrow, B=B2, N=N, C_=c, heads=heads,
                              n_txt=77, n_ip=4, ip_scale=1.0, residual=x, ln_gamma=lg, ln_beta=lb)
elif which == "xattn3":
    from consistentid_amd import xattn_pack
    N, c, heads = 4096, 320, 8
    x = rnd(B2, N, c)
    out = torch.empty_like(x)
    wo, bo = rnd(c, c), rnd(c)
    wq_f, qs, qb = xattn_pack.fold_layernorm(rnd(c, c).float(), rnd(c).float() + 1, rnd(c).float())
    wq_p, wo_p = xattn_pack.pack_w3(wq_f), xattn_pack.pack_w3(wo)
    ke, ve = ops.kv_pack2_elems(c, heads)
    kp, vp = rnd(B2 * ke), rnd(B2 * ve)
    kvrow = torch.arange(B2, dtype=torch.int32, device=dev)
    fn = lambda: ops.id_xattn3(x, out, wq_p=wq_p, q_rowsum=qs, q_bias=qb, wo_p=wo_p, bo=bo, kp=kp, vp=vp, kvrow=kvrow, B=B2,
                               N=N, C_=c, heads=heads, n_txt=77, n_ip=4, ip_scale=1.0, has_ln=True, add_residual=True)
else:
    raise SystemExit(f"unknown kernel {which}")
for _ in range(iters):
    fn()
torch.cuda.synchronize()
