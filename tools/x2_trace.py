#!/usr/bin/env python
"""Phase timeline of the fused ID cross-attention kernels (experiment build with -DCID_X2_TRACE -DCID_X3_TRACE).
  python -m consistentid_amd.build --variant trace CID_X2_TRACE CID_X3_TRACE
  CID_LIBRARY=consistentid_amd/libcid_trace.so python tools/x2_trace.py [--gen 2|3]
Prints, per stamp, the shader-clock cycles since the wave's own start (mean / min / max over all waves), and how far
the workgroups' start times are spread (all stamps come from one clock)."""
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


NAMES3 = ["start", "x slab0 published", "x slab1 published", "x slab4 published", "A mfma done", "Q final",
          "head0 pass1 done", "head1 pass1 done", "B done (pre-barrier F)", "barrier F passed", "C mfma done",
          "result in LDS", "stores issued", "stores drained"]


def main():
    gen = 3 if "--gen" in sys.argv and sys.argv[sys.argv.index("--gen") + 1] == "3" else 2
    names, tile, waves = (NAMES3, 64, 4) if gen == 3 else (NAMES, 128, 8)
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
    if gen == 3:
        wq_p, wo_p = xattn_pack.pack_w3(wq_f), xattn_pack.pack_w3(wo)
        run = lambda: ops.id_xattn3(x, out, wq_p=wq_p, q_rowsum=qs, q_bias=qb, wo_p=wo_p, bo=bo, kp=kp, vp=vp, kvrow=kvrow,
                                    B=B2, N=N, C_=c, heads=heads, n_txt=77, n_ip=4, ip_scale=1.0, has_ln=True,
                                    add_residual=True)
    else:
        raise SystemExit("the second generation (csrc/xattn2.hip) was removed in round 5: run with --gen 3")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    lib = _lib.load()
    nwg = B2 * N // tile
    buf = np.zeros(nwg * waves * 32, dtype=np.uint64)
    fetch = lib.cid_debug_x3_trace if gen == 3 else lib.cid_debug_x2_trace
    fetch.argtypes = [C.c_void_p, C.c_int64]
    rc = fetch(buf.ctypes.data, buf.size)
    assert rc == 0, rc
    t = buf.reshape(nwg, waves, 32)[:, :, :len(names)].astype(np.int64)
    rel = t - t[:, :, :1]
    print(f"generation {gen}: launch wall (events, traced build): {e0.elapsed_time(e1) * 1e3:.1f} us; {nwg} workgroups of {waves} waves")
    st = t[:, :, 0] - t[:, :, 0].min()
    print(f"workgroup start spread: mean {st.mean():.0f}, max {st.max()} cycles; first start -> last stamp: {t.max() - t[:, :, 0].min()} cycles")
    if gen == 3:        # where the time differs: per XCD, and how the two workgroups of a CU compare
        raw = buf.reshape(nwg, waves, 32)
        r0, r1 = raw[:, :, 28].astype(np.int64), raw[:, :, 29].astype(np.int64)       # 100 MHz real-time counter
        span_us = (r1.max() - r0.min()) / 100.0
        ghz = ((t[:, :, len(names) - 1] - t[:, :, 0]) / np.maximum(r1 - r0, 1)).mean() / 10.0
        print(f"real time: first wave start -> last wave end {span_us:.2f} us; wave starts spread over {(r0.max() - r0.min()) / 100.0:.2f} us; "
              f"shader clock while the kernel runs {ghz:.2f} GHz")
        hw, xcc = raw[:, 0, 30].astype(np.int64), raw[:, 0, 31].astype(np.int64) & 15
        cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)
        dur = (t[:, :, len(names) - 1] - t[:, :, 0]).max(1)
        t0 = np.array([t[xcc == k, :, 0].min() if (xcc == k).any() else 0 for k in range(8)])
        end = t[:, :, len(names) - 1].max(1) - t0[xcc]
        print("XCD   workgroups  mean duration   max   last end (since the XCD's first start)")
        for k in range(8):
            m = xcc == k
            if m.any():
                print(f"{k:3d} {m.sum():10d} {dur[m].mean():12.0f} {dur[m].max():8d} {end[m].max():10d}")
        key = xcc * 1000 + cu
        pairs = [dur[key == q] for q in np.unique(key)]
        sizes = np.bincount([len(q) for q in pairs])
        print("workgroups per CU (count of CUs):", {n: int(c) for n, c in enumerate(sizes) if c})
        two = np.array([q for q in pairs if len(q) == 2])
        if len(two):
            print(f"CUs with two workgroups: |duration difference| mean {np.abs(two[:, 0] - two[:, 1]).mean():.0f}, "
                  f"correlation {np.corrcoef(two[:, 0], two[:, 1])[0, 1]:.2f}")
        st_rel = t[:, 0, 0] - t0[xcc]
        print(f"start offsets inside an XCD: mean {st_rel.mean():.0f}, max {st_rel.max()} cycles")
    print(f"{'stamp':28s} {'mean':>8s} {'min':>8s} {'max':>8s} {'delta(mean)':>12s}")
    order = np.argsort(rel.mean((0, 1)))
    prev = 0.0
    for k in order:
        m = rel[:, :, k].mean()
        print(f"{names[k]:28s} {m:8.0f} {rel[:, :, k].min():8d} {rel[:, :, k].max():8d} {m - prev:12.0f}")
        prev = m


if __name__ == "__main__":
    main()
