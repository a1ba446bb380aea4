#!/usr/bin/env python
"""Per-level table of the identity cross-attention block  x + attn2(LayerNorm(x), ehs)  (SURVEY 8 a1): for every
distinct (channels, tokens) level of the SD1.5 and SDXL UNets, the launch sequence the engine uses
(HipUNet.cross_attention) timed with HIP events, its algorithmic flop rate against the dense fp16 MFMA peak, and the
alternative path (one launch of the first-generation fused kernel where the engine runs LN + GEMM + core + GEMM, and the
other way round).  usage: python tools/xattn_levels.py [--b2-sd15 8] [--b2-sdxl 4]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from consistentid_amd import synth, unet_spec  # noqa: E402
from consistentid_amd.unet import HipUNet  # noqa: E402


def timeit(fn, iters=30, warm=3):
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
    ap.add_argument("--b2-sd15", type=int, default=8)
    ap.add_argument("--b2-sdxl", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    print(f"{'family':6s} {'C':>5s} {'heads':>5s} {'tokens':>6s} {'B2':>3s} {'layers':>6s}  {'engine path':62s} {'us':>7s} {'TF/s':>7s} {'frac':>6s}   "
          f"{'other path':34s} {'us':>7s}")
    for family, cfg, side, B2 in (("sd15", unet_spec.sd15_config(), 64, a.b2_sd15), ("sdxl", unet_spec.sdxl_config(), 128, a.b2_sdxl)):
        sd = synth.random_unet_state_dict(cfg, seed=0, device=dev)
        ad = synth.random_adapter_state_dict(cfg, sd, rank=128, seed=1, device=dev)
        unet = HipUNet(cfg, sd, ad, device=dev)
        del sd, ad
        g = torch.Generator(device=dev).manual_seed(5)
        ehs = (torch.randn(3, 81, cfg.cross_attention_dim, generator=g, device=dev) * 0.5).half()
        unet.set_context(ehs)
        kvrow = (torch.arange(B2, dtype=torch.int32, device=dev) % 3).contiguous()
        seen = {}
        for blk in unet.downs + [unet.mid] + unet.ups:
            for t in blk.attentions:
                seen.setdefault((t.channels, t.heads, t.name.split(".")[0] == "mid_block"), []).append(t)
        for (C, heads, is_mid), ts in sorted(seen.items(), key=lambda kv: (kv[0][0], kv[0][2])):
            t = ts[0]
            level = cfg.block_out_channels.index(C) if not is_mid else len(cfg.block_out_channels) - 1
            N = (side >> level) ** 2
            layer = f"{t.name}.transformer_blocks.0"
            layers = sum(q.n_layers for q in ts)
            x = (torch.randn(B2 * N, C, generator=g, device=dev) * 0.7).half()
            run = lambda: unet.cross_attention(layer, x, B2, N, C, heads, kvrow)
            path = unet.cross_attention_path(layer, C, B2, N)
            dt = timeit(run)
            fl = bench.xattn_flops(B2, N, C)
            other, dt2 = "-", float("nan")
            if not unet._ctx.v2.get(layer):          # swap split <-> first-generation fused (same K/V operands)
                keep, rule = unet._xattn_fused_max_c, unet._fused_gen1
                fused_now = rule(C, B2 * N)
                unet._fused_gen1 = lambda c, tokens: not fused_now
                try:
                    other = unet.cross_attention_path(layer, C, B2, N)
                    dt2 = timeit(run)
                except Exception as e:      # geometry the other kernel is not built for
                    other = f"not available ({type(e).__name__})"
                unet._fused_gen1 = rule
            third = ""
            if not unet._ctx.v2.get(layer) and unet._qattn:      # the same level without the attention epilogue (four launches)
                unet._qattn = False
                fused_now = unet._fused_gen1(C, B2 * N)
                rule = unet._fused_gen1
                unet._fused_gen1 = lambda c, tokens: False
                third = f"   four launches {timeit(run) * 1e6:6.1f}"
                unet._fused_gen1 = rule
                unet._qattn = True
            print(f"{family:6s} {C:5d} {heads:5d} {N:6d} {B2:3d} {layers:6d}  {path[:62]:62s} {dt * 1e6:7.1f} {fl / dt / 1e12:7.1f} "
                  f"{fl / dt / 1e12 / bench.MFMA_F16_PEAK_TFLOPS:6.3f}   {other[:34]:34s} {dt2 * 1e6:7.1f}{third}")
        del unet
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
