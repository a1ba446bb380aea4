#!/usr/bin/env python
"""Per-kernel micro-benchmarks at the shapes of the SD1.5 (or SDXL) UNet step: achieved
TFLOP/s (MFMA-bound kernels) or GB/s (HBM-bound kernels) with HIP-event timing.
Usage: python tools/kbench.py [--b2 8] [--family sd15] [--only gemm,attn,xattn,norm]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistentid_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b2", type=int, default=8)
    ap.add_argument("--only", default="gemm,attn,xattn,norm")
    ap.add_argument("--family", default="sd15", choices=["sd15", "sdxl"], help="sdxl: GEMM shapes of the SDXL UNet (use --b2 4)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B2 = a.b2
    only = set(a.only.split(","))
    rows = []
    alg_bytes = {}      # label -> algorithmic HBM bytes per launch (activations + weights + outputs once): GB/s column
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).half()
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)

    if "gemm" in only:
        # (label, HW side, Cin, Cout, taps, count per UNet forward)
        shapes = [
            ("conv3 L0 320->320", 64, 320, 320, 9, 7), ("conv3 L0 960->320", 64, 960, 320, 9, 1),
            ("conv3 L0 640->320", 64, 640, 320, 9, 2), ("conv3 L0up 640->640", 64, 640, 640, 9, 1),
            ("conv3 L1 640->640", 32, 640, 640, 9, 7), ("conv3 L1 1920->640", 32, 1920, 640, 9, 1),
            ("conv3 L1 1280->640", 32, 1280, 640, 9, 1), ("conv3 L1up 1280->1280", 32, 1280, 1280, 9, 1),
            ("conv3 L2 1280->1280", 16, 1280, 1280, 9, 8), ("conv3 L2 2560->1280", 16, 2560, 1280, 9, 2),
            ("conv3 L2 1920->1280", 16, 1920, 1280, 9, 1),
            ("conv3 L3 1280->1280", 8, 1280, 1280, 9, 11), ("conv3 L3 2560->1280", 8, 2560, 1280, 9, 3),
            ("lin L0 320->320", 64, 320, 320, 1, 20), ("lin L0 1280->320 (ff2)", 64, 1280, 320, 1, 5),
            ("lin L1 640->640", 32, 640, 640, 1, 20), ("lin L1 2560->640 (ff2)", 32, 2560, 640, 1, 5),
            ("lin L2 1280->1280", 16, 1280, 1280, 1, 20), ("lin L2 5120->1280 (ff2)", 16, 5120, 1280, 1, 5),
        ]
        sdxl_shapes = a.family == "sdxl"
        if a.family == "sdxl":      # 1024^2: levels 128x128 (320, no attention), 64x64 (640, 2 layers), 32x32 (1280, 10 layers)
            shapes = [
                ("conv3 X0 320->320", 128, 320, 320, 9, 7), ("conv3 X1 640->640", 64, 640, 640, 9, 7),
                ("conv3 X2 1280->1280", 32, 1280, 1280, 9, 9), ("conv3 X2 2560->1280", 32, 2560, 1280, 9, 2),
                ("lin X1 640->640", 64, 640, 640, 1, 40), ("lin X1 2560->640 (ff2)", 64, 2560, 640, 1, 10),
                ("lin X2 1280->1280", 32, 1280, 1280, 1, 240), ("lin X2 5120->1280 (ff2)", 32, 5120, 1280, 1, 60),
            ]
        if not sdxl_shapes:      # Upsample2D convolutions (nearest-2x folded into the gather): side = INPUT side
            shapes += [("conv3 up 32->64 640->640", -32, 640, 640, 9, 1), ("conv3 up 16->32 1280->1280", -16, 1280, 1280, 9, 1),
                       ("conv3 up 8->16 1280->1280", -8, 1280, 1280, 9, 1)]
        for label, side, cin, cout, taps, cnt in shapes:
            up = 1 if side < 0 else 0
            side_in, side = abs(side), abs(side) << up
            M = B2 * side * side
            x, w, b = rnd(B2 * side_in * side_in, cin), rnd(cout, taps * cin), rnd(cout)
            out = torch.empty(M, cout, dtype=torch.float16, device=dev)
            kw = dict(taps=9, Hi=side_in, Wi=side_in, Ho=side, Wo=side, up=up) if taps == 9 else {}
            t = timeit(lambda: ops.gemm(x, w, out, M=M, N=cout, c1=cin, bias=b, ws=ws, **kw))
            fl = 2.0 * M * cout * cin * taps
            rows.append((label, f"M={M}", t * 1e6, fl / t / 1e12, "TF/s", cnt))
            alg_bytes[label] = 2.0 * (x.shape[0] * cin + cout * taps * cin + M * cout)
            if taps == 9 and side >= 32 and cin == cout and not up:      # the same conv emitting GroupNorm statistics from its epilogue
                t2 = timeit(lambda: ops.gemm(x, w, out, M=M, N=cout, c1=cin, bias=b, ws=ws, gn_hw=side * side, **kw))
                rows.append((label + " +gn-stats", f"M={M}", t2 * 1e6, fl / t2 / 1e12, "TF/s", 0))
        sdxl = a.family == "sdxl"
        for label, side, c in ((("geglu X1", 64, 640), ("geglu X2", 32, 1280)) if sdxl else
                               (("geglu L0", 64, 320), ("geglu L1", 32, 640), ("geglu L2", 16, 1280))):
            M = B2 * side * side
            x, w, b = rnd(M, c), rnd(8 * c, c), rnd(8 * c)
            out = torch.empty(M, 4 * c, dtype=torch.float16, device=dev)
            t = timeit(lambda: ops.gemm(x, w, out, M=M, N=8 * c, c1=c, bias=b, mode=1))
            rows.append((label, f"M={M}", t * 1e6, 2.0 * M * 8 * c * c / t / 1e12, "TF/s", 5))
            alg_bytes[label] = 2.0 * (M * c + 8 * c * c + M * 4 * c)
            lns, lnb = rnd(8 * c).float(), rnd(8 * c).float()
            t = timeit(lambda: ops.gemm(x, w, out, M=M, N=8 * c, c1=c, mode=1, ln=(lns, lnb, 1e-5)))
            rows.append((label + " +LN fold", f"M={M}", t * 1e6, 2.0 * M * 8 * c * c / t / 1e12, "TF/s", 0))
            alg_bytes[label + " +LN fold"] = alg_bytes[label]
        for label, side, c, heads in ((("qkv X1", 64, 640, 10), ("qkv X2", 32, 1280, 20)) if sdxl else
                                      (("qkv L0", 64, 320, 8), ("qkv L1", 32, 640, 8), ("qkv L2", 16, 1280, 8))):
            N = side * side
            M = B2 * N
            d = c // heads
            x, w = rnd(M, c), rnd(3 * c, c)
            qk = torch.empty(M, 2 * c, dtype=torch.float16, device=dev)
            vt = torch.empty(B2 * heads * ops.dvp_of(d) * N, dtype=torch.float16, device=dev)
            t = timeit(lambda: ops.gemm(x, w, qk, M=M, N=3 * c, c1=c, mode=2, vt=vt, n_vt0=2 * c, heads=heads,
                                        dhead=d, ntok=N))
            rows.append((label, f"M={M}", t * 1e6, 2.0 * M * 3 * c * c / t / 1e12, "TF/s", 5))
            alg_bytes[label] = 2.0 * (M * c + 3 * c * c + M * 3 * c)
            lns, lnb = rnd(3 * c).float(), rnd(3 * c).float()
            t = timeit(lambda: ops.gemm(x, w, qk, M=M, N=3 * c, c1=c, mode=2, vt=vt, n_vt0=2 * c, heads=heads,
                                        dhead=d, ntok=N, ln=(lns, lnb, 1e-5)))
            rows.append((label + " +LN fold", f"M={M}", t * 1e6, 2.0 * M * 3 * c * c / t / 1e12, "TF/s", 0))
            alg_bytes[label + " +LN fold"] = alg_bytes[label]

    if "attn" in only:
        for label, side, c, heads in (("self-attn L0 d40", 64, 320, 8), ("self-attn L1 d80", 32, 640, 8),
                                      ("self-attn L2 d160", 16, 1280, 8), ("self-attn mid d160", 8, 1280, 8)):
            N = side * side
            d = c // heads
            qk = rnd(B2 * N, 2 * c)
            vt = rnd(B2 * heads * ops.dvp_of(d) * N)
            out = torch.empty(B2 * N, c, dtype=torch.float16, device=dev)
            t = timeit(lambda: ops.self_attn(qk, qk[:, c:], vt, out, B=B2, N=N, heads=heads, d=d, ldq=2 * c,
                                             ldk=2 * c, ldo=c))
            rows.append((label, f"N={N}", t * 1e6, 4.0 * B2 * N * N * c / t / 1e12, "TF/s", 5))

    if "xattn" in only:
        for label, side, c, heads in (("id-xattn L0", 64, 320, 8), ("id-xattn L1", 32, 640, 8),
                                      ("id-xattn L2", 16, 1280, 8), ("id-xattn mid", 8, 1280, 8)):
            N = side * side
            L = 81
            x = rnd(B2, N, c)
            out = torch.empty_like(x)
            wq, wo, bo = rnd(c, c), rnd(c, c), rnd(c)
            ke, ve = ops.kv_pack_elems(c, heads)
            kp, vp = rnd(B2 * ke), rnd(B2 * ve)
            kvrow = torch.arange(B2, dtype=torch.int32, device=dev)
            lg, lb = rnd(c), rnd(c)
            t = timeit(lambda: ops.id_xattn(x, out, wq=wq, wo=wo, bo=bo, kp=kp, vp=vp, kvrow=kvrow, B=B2, N=N, C_=c,
                                            heads=heads, n_txt=77, n_ip=4, ip_scale=1.0, residual=x, ln_gamma=lg,
                                            ln_beta=lb))
            fl = B2 * (4.0 * N * c * c + 4.0 * N * L * c)
            rows.append((label, f"N={N} C={c}", t * 1e6, fl / t / 1e12, "TF/s", 5 if side != 8 else 1))

    if "xattn" in only or "xattn3" in only:
        # the fused kernel of the SD1.5 level-0 geometry (csrc/xattn3.hip)
        from consistentid_amd import xattn_pack
        side, c, heads, L = 64, 320, 8, 81
        N = side * side
        x = rnd(B2, N, c)
        out = torch.empty_like(x)
        wo, bo = rnd(c, c), rnd(c)
        wq_f, qs, qb = xattn_pack.fold_layernorm(rnd(c, c).float(), rnd(c).float() + 1, rnd(c).float())
        ke, ve = ops.kv_pack2_elems(c, heads)
        kp, vp = rnd(B2 * ke), rnd(B2 * ve)
        kvrow = torch.arange(B2, dtype=torch.int32, device=dev)
        fl = B2 * (4.0 * N * c * c + 4.0 * N * L * c)
        # third generation (64-token tiles, two workgroups per CU, packed A-operand streams)
        wq_p, wo_p = xattn_pack.pack_w3(wq_f), xattn_pack.pack_w3(wo)
        t = timeit(lambda: ops.id_xattn3(x, out, wq_p=wq_p, q_rowsum=qs, q_bias=qb, wo_p=wo_p, bo=bo, kp=kp, vp=vp,
                                         kvrow=kvrow, B=B2, N=N, C_=c, heads=heads, n_txt=77, n_ip=4, ip_scale=1.0,
                                         has_ln=True, add_residual=True), iters=50)
        rows.append(("id-xattn3 L0", f"N={N} C={c}", t * 1e6, fl / t / 1e12, "TF/s", 5))

    if "norm" in only:
        gws = torch.zeros(ops.groupnorm_ws_bytes(B2, 2560), dtype=torch.uint8, device=dev)
        for label, side, c1, c2 in (("groupnorm L0 320", 64, 320, 0), ("groupnorm L0 640+320", 64, 640, 320),
                                    ("groupnorm L1 640", 32, 640, 0), ("groupnorm L2 1280+1280", 16, 1280, 1280),
                                    ("groupnorm L3 1280", 8, 1280, 0)):
            HW = side * side
            x1 = rnd(B2 * HW, c1)
            x2 = rnd(B2 * HW, c2) if c2 else None
            gm, bt = rnd(c1 + c2), rnd(c1 + c2)
            out = torch.empty(B2 * HW, c1 + c2, dtype=torch.float16, device=dev)
            t = timeit(lambda: ops.groupnorm(x1, out, gm, bt, gws, B=B2, HW=HW, c1=c1, x2=x2, c2=c2))
            by = 2.0 * B2 * HW * (c1 + c2) * 2
            rows.append((label, f"HW={HW}", t * 1e6, by / t / 1e9, "GB/s", 1))
            if HW > ops.GN_SMALL_MAX_HW and ops._lib.load().cid_groupnorm_stats_ok(c1, c2, 32):
                # the same GroupNorm on statistics its producer(s) emitted (values irrelevant for timing)
                rows_ = 256
                x1._gn_stats = (torch.rand(B2 * HW // rows_, 32, 2, device=dev) + 1.0, rows_)
                if x2 is not None:
                    x2._gn_stats = (torch.rand(B2 * HW // rows_, 32, 2, device=dev) + 1.0, rows_)
                t = timeit(lambda: ops.groupnorm(x1, out, gm, bt, gws, B=B2, HW=HW, c1=c1, x2=x2, c2=c2))
                rows.append((label + " (epilogue stats)", f"HW={HW}", t * 1e6, by / t / 1e9, "GB/s", 0))
                del x1._gn_stats
        for label, side, c in (("layernorm L0", 64, 320), ("layernorm L1", 32, 640), ("layernorm L2", 16, 1280)):
            M = B2 * side * side
            x, gm, bt = rnd(M, c), rnd(c), rnd(c)
            out = torch.empty_like(x)
            t = timeit(lambda: ops.layernorm(x, out, gm, bt, M=M, C_=c))
            rows.append((label, f"M={M}", t * 1e6, 2.0 * M * c * 2 / t / 1e9, "GB/s", 10))

    tot = 0.0
    print(f"{'kernel':40s} {'shape':14s} {'us':>9s} {'rate':>9s} unit  {'alg GB/s':>9s}  x/fwd  ms/fwd")
    for label, shape, us, rate, unit, cnt in rows:
        tot += us * cnt
        gbs = f"{alg_bytes[label] / us / 1e3:9.0f}" if label in alg_bytes else " " * 9     # vs 6300 GB/s achievable HBM
        print(f"{label:40s} {shape:14s} {us:9.1f} {rate:9.1f} {unit:5s} {gbs}  {cnt:5d} {us * cnt / 1e3:7.3f}")
    print(f"sum over listed kernels x count: {tot / 1e3:.2f} ms per UNet forward (B2={B2})")


if __name__ == "__main__":
    main()
