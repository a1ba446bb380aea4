#!/usr/bin/env python
"""Probe: does the denoise step run faster as K independent sample groups on K HIP streams (graph branches) than as one
in-order kernel sequence on the whole CFG batch?  Same kernels, same weights; each lane is the SD1.5 UNet forward + CFG +
DDIM update of B / K samples (CFG batch 2 B / K) with its own workspaces (a second engine object sharing the packed weights
and the K/V cache).  Prints ms per step for 1 / 2 / 4 lanes, forked inside ONE captured graph.

  python tools/lanes_probe.py [--batch 4] [--family sd15|sdxl] [--iters 30]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from consistentid_amd import ops, synth, unet_spec   # noqa: E402
from consistentid_amd.unet import HipUNet             # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--batch", type=int, default=4)
    p.add_argument("--family", default="sd15")
    p.add_argument("--iters", type=int, default=30)
    p.add_argument("--lanes", default="1,2,4")
    a = p.parse_args()
    dev = torch.device("cuda:0")
    sdxl = a.family == "sdxl"
    cfg = unet_spec.sdxl_config() if sdxl else unet_spec.sd15_config()
    px = 1024 if sdxl else 512
    sd = synth.random_unet_state_dict(cfg, seed=0, device=dev)
    ad = synth.random_adapter_state_dict(cfg, sd, rank=128, seed=1, device=dev)
    unet = HipUNet(cfg, sd, ad, device=dev)
    del sd, ad
    torch.cuda.empty_cache()
    B = a.batch
    inp = synth.random_inputs(cfg, B, px, px, device=dev)
    unet.set_context(torch.cat([inp["null"], inp["text"], inp["augmented"]], 0))
    lat0 = inp["latents"].to(dev).half().contiguous()
    t_buf = torch.full((1,), 981.0, dtype=torch.float32, device=dev)
    coef = torch.tensor([0.99, -0.05, 0.0, 0.0, 1.0], dtype=torch.float32, device=dev)
    temb = None
    added_all = None
    if not sdxl:
        temb = unet.time_embed_table(t_buf)[:1].contiguous()
    ar = torch.arange(B, dtype=torch.int32, device=dev)
    per_sample = lat0[0].numel()
    results = {}
    ref_out = None
    for K in [int(v) for v in a.lanes.split(",")]:
        if B % K:
            continue
        Bg = B // K
        engines = [unet] + [HipUNet(cfg, device=dev, packed=unet.packed) for _ in range(K - 1)]
        for e in engines[1:]:
            e._ctx = unet._ctx
        streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
        lat = lat0.clone()
        kv = [torch.cat([ar[g * Bg:(g + 1) * Bg], ar[g * Bg:(g + 1) * Bg] + B]).contiguous() for g in range(K)]
        added = [None] * K
        if sdxl:
            for g in range(K):
                sl = slice(g * Bg, (g + 1) * Bg)
                added[g] = {"text_embeds": torch.cat([inp["pooled_null"][sl], inp["pooled_text"][sl]]).to(dev).half().contiguous(),
                            "time_ids": inp["time_ids"][:2 * Bg].to(dev).float().contiguous()}

        def step():
            cur = torch.cuda.current_stream()
            if K == 1:
                eps = engines[0].forward_tokens(lat, t_buf, kv[0], 2 * B, added[0], temb=temb)
                ops.cfg_ddim_step(eps, lat, coef, 5.0, B=B, per_sample=per_sample)
                return
            for g in range(K):
                streams[g].wait_stream(cur)
                with torch.cuda.stream(streams[g]):
                    lg = lat[g * Bg:(g + 1) * Bg]
                    eps = engines[g].forward_tokens(lg, t_buf, kv[g], 2 * Bg, added[g], temb=temb)
                    ops.cfg_ddim_step(eps, lg, coef, 5.0, B=Bg, per_sample=per_sample)
            for g in range(K):
                cur.wait_stream(streams[g])

        step()          # eager warm-up
        torch.cuda.synchronize()
        out1 = lat.clone()
        if ref_out is None:
            ref_out = out1
        else:
            d = float((out1.float() - ref_out.float()).norm() / ref_out.float().norm())
            print(f"lanes={K}: one step vs lanes=1 rel_l2 = {d:.3e}")
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            step()
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            gr.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.iters * 1e3
        results[K] = ms
        print(f"{a.family} B={B} lanes={K}: {ms:.3f} ms per step  ({B / (ms * 1e-3 * (30 if sdxl else 50)):.3f} images/s at "
              f"{30 if sdxl else 50} steps)", flush=True)
        del gr, engines
        torch.cuda.empty_cache()
    return results


if __name__ == "__main__":
    main()
