#!/usr/bin/env python
"""The reference's default inference workload (infer.py:33,48-49,59,63-64: Euler scheduler, 50 steps, merge step 30,
512 x 768) on synthetic weights: checks that the non-square size (mid block on a padded token axis) runs at full width and
prints the throughput.  usage: python tools/run_infer_default.py [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistentid_amd import pipeline, scheduler, synth, unet_spec  # noqa: E402
from consistentid_amd.unet import HipUNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
cfg = unet_spec.sd15_config()
sd = synth.random_unet_state_dict(cfg, device=dev)
ad = synth.random_adapter_state_dict(cfg, sd, rank=128, device=dev)
unet = HipUNet(cfg, sd, ad, device=dev)
del sd, ad
pipe = pipeline.ConsistentIDStableDiffusionPipeline(unet, scheduler=scheduler.EulerDiscreteScheduler())
H, W = 768, 512          # infer.py: height 768, width 512
inp = synth.random_inputs(cfg, B, H, W, device=dev)
kw = dict(prompt_embeds=torch.cat([inp["null"], inp["augmented"], inp["text"]]), latents=inp["latents"],
          num_inference_steps=50, guidance_scale=5.0, start_merge_step=30, output_type="latent")
out = pipe(**kw).images
torch.cuda.synchronize()
t0 = time.perf_counter()
out = pipe(**kw).images
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert torch.isfinite(out.float()).all() and out.shape == (B, 4, H // 8, W // 8)
print(f"infer.py default (Euler, 50 steps, merge 30, {W}x{H}, batch {B}): {dt:.3f} s per generation, {B / dt:.2f} images/s, "
      f"latent std {out.float().std():.3f}")
