"""Phase stamps of the 3x3 convolution kernel (experiment builds only).
  python -m consistentid_amd.build --variant ctr_stag CID_CONV_TRACE CID_HALO_STAGGER=1
  python -m consistentid_amd.build --variant ctr_lock CID_CONV_TRACE CID_HALO_STAGGER=0
  CID_LIBRARY=consistentid_amd/libcid_ctr_stag.so python tools/conv_trace.py
Prints, for waves 0 and NW/2 of workgroup 100 of the level-0 320->320 convolution, the median cycles between consecutive stamps
of an iteration (s_memtime ticks) and the median iteration length."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from consistentid_amd import _lib, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    fetch = lib.cid_debug_conv_trace
    fetch.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    B, H, W, C = 8, 64, 64, 320
    g = torch.Generator(device="cpu").manual_seed(0)
    x = (torch.randn(B * H * W, C, generator=g) * 0.5).half().to(dev)
    w = (torch.randn(C, 9 * C, generator=g) * 0.02).half().to(dev)
    b = torch.zeros(C).half().to(dev)
    M = B * H * W
    out = torch.empty(M, C, dtype=torch.float16, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    run = lambda: ops.gemm(x, w, out, M=M, N=C, c1=C, bias=b, ws=ws, taps=9, Hi=H, Wi=W, Ho=H, Wo=W)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    host = np.zeros(2 * 4096, dtype=np.uint64)
    rc = fetch(host.ctypes.data, host.size)
    assert rc == 0, rc
    for slot, name in ((0, "wave 0"), (1, "wave NW/2")):
        t = host[slot * 4096:(slot + 1) * 4096].astype(np.int64)
        n = int((t != 0).sum()) // 8
        t = t[:n * 8].reshape(n, 8)
        body = t[4:-2]
        d = np.diff(body, axis=1)
        it = body[1:, 0] - body[:-1, 0]
        print(f"{name}: {n} iterations; median iteration {np.median(it):.0f} ticks (p10 {np.percentile(it, 10):.0f}, p90 {np.percentile(it, 90):.0f}); total {t[-1, 7] - t[0, 0]} ticks")
        print("   median gap stamp k -> k+1:", " ".join(f"{k}>{k + 1}:{np.median(d[:, k]):.0f}" for k in range(7)),
              f" 7>next0:{np.median(body[1:, 0] - body[:-1, 7]):.0f}")
        # even / odd iterations separately (the staggered pipeline alternates k-steps 0 and 1)
        for par in (0, 1):
            dd = d[par::2]
            print(f"   iterations {par} mod 2:", " ".join(f"{np.median(dd[:, k]):.0f}" for k in range(7)),
                  f"| next {np.median((body[1:, 0] - body[:-1, 7])[par::2]):.0f}")
    # absolute timeline of a few iterations, both waves on one clock
    t0 = host[0:4096].astype(np.int64); t1 = host[4096:].astype(np.int64)
    base = t0[8 * 20]
    ev = []
    for it_ in range(20, 25):
        for k in range(8):
            ev.append((t0[8 * it_ + k] - base, f"w0  it{it_} s{k}"))
            ev.append((t1[8 * it_ + k] - base, f"w4  it{it_} s{k}"))
    for tt, nm in sorted(ev):
        print(f"   {tt:7d}  {nm}")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"launch (traced build): {e0.elapsed_time(e1) * 50:.1f} us")


if __name__ == "__main__":
    main()
