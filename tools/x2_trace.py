#!/usr/bin/env python
"""Phase timeline of the second-generation fused ID cross-attention (experiment build with -DCID_X2_TRACE).
  python -m consistentid_amd.build --variant trace CID_X2_TRACE
  CID_LIBRARY=consistentid_amd/libcid_trace.so python tools/x2_trace.py
Prints, per stamp, the shader-clock cycles since the wave's own start (mean / min / max over all waves)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistentid_amd import _lib, ops, xattn_pack  # noqa: E402

NAMES = ["start", "slab0 ready", "slab1 ready", "A mfma done", "Q final (pre-barrier E)", "barrier E passed",
         "head0 pass1 done", "head1 pass1 done", "B done (pre-barrier F)", "barrier F passed", "C mfma done",
         "final barrier passed", "stores issued", "stores drained",
         "A slab3: before DMA wait", "A slab3: DMA landed", "A slab3: barrier passed", "A slab4: barrier passed",
         "C slab2: before DMA wait", "C slab2: DMA landed", "C slab2: barrier passed", "C slab3: barrier passed"]


def main():
    dev = torch.device("cuda:0")
    B2, N, c, heads = 8, 4096, 320, 8
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).half()
    x = rnd(B2, N, c)
    out = torch.empty_like(x)
    wo, bo = rnd(c, c), rnd(c)
    wq_f, qs, qb = xattn_pack.fold_layernorm(rnd(c, c).float(), rnd(c).float() + 1, rnd(c).float())
    ke, ve = ops.kv_pack2_elems(c, heads)
    kp, vp = rnd(B2 * ke), rnd(B2 * ve)
    kvrow = torch.arange(B2, dtype=torch.int32, device=dev)
    run = lambda: ops.id_xattn2(x, out, wq_f=wq_f, q_rowsum=qs, q_bias=qb, wo=wo, bo=bo, kp=kp, vp=vp, kvrow=kvrow, B=B2,
                                N=N, C_=c, heads=heads, n_txt=77, n_ip=4, ip_scale=1.0, has_ln=True, add_residual=True)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    lib = _lib.load()
    nwg = B2 * N // 128
    buf = np.zeros(nwg * 8 * 32, dtype=np.uint64)
    lib.cid_debug_x2_trace.argtypes = [C.c_void_p, C.c_int64]
    rc = lib.cid_debug_x2_trace(buf.ctypes.data, buf.size)
    assert rc == 0, rc
    t = buf.reshape(nwg, 8, 32)[:, :, :len(NAMES)].astype(np.int64)
    rel = t - t[:, :, :1]
    print(f"launch wall (events, traced build): {e0.elapsed_time(e1) * 1e3:.1f} us; {nwg} workgroups")
    print(f"{'stamp':28s} {'mean':>8s} {'min':>8s} {'max':>8s} {'delta(mean)':>12s}")
    order = np.argsort(rel.mean((0, 1)))
    prev = 0.0
    for k in order:
        m = rel[:, :, k].mean()
        print(f"{NAMES[k]:28s} {m:8.0f} {rel[:, :, k].min():8d} {rel[:, :, k].max():8d} {m - prev:12.0f}")
        prev = m


if __name__ == "__main__":
    main()
