"""Parity on BASELINE.json's own configurations at FULL size (the cases the toy-UNet tests cannot stand in for):

  * config 2: SD1.5, 512x512, batch 4 (CFG batch 8), 50 DDIM steps, start_merge_step 30, hipGraph replay --
    the whole trajectory against the oracle loop, plus two single forwards at CFG batch 8 against the fp32 CPU oracle;
  * the reference's default script workload (infer.py:63-64): 512x768, one full-size forward (its 8x12 mid level is
    not a multiple of the attention tile: padded-token path at real width);
  * config 3's per-GPU shape (plain SD1.5, batch 8, 50 steps: what bench.py --gpus N > 1 times) as a trajectory;
  * the reference's SDXL default (infer_SDXL.py:61-62): 864x1152, padded-token path at SDXL's widths;
  * the implicit K/V cache of the diffusers-style call under address reuse.

Criterion everywhere: error against the fp32 oracle <= max(1e-3, 1.5 x the error of the oracle's own modules run in fp16
with stock PyTorch-ROCm kernels on the same GPU) -- conftest.check_vs_fp16_arm.  The oracle UNet restates
diffusers==0.23.0 (not vendored by the reference): PARITY UNPINNED below the processors, see DESIGN.md section 2."""
import copy

import pytest
import torch

from conftest import check_vs_fp16_arm, half_arm, rel_l2
from oracle_utils import build_oracle, make_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd15(dev):
    from consistentid_amd.unet import HipUNet
    # LoRA rank 128: the reference's own (lora_rank = 128, pipline_StableDiffusion_ConsistentID.py:47) and the benchmarked
    # configuration's -- the merge at the real widths runs on the rank the bench line uses
    cfg, sd, ad = make_weights("sd15", rank=128, device=dev)
    hip = HipUNet(cfg, sd, ad, device=dev)
    oracle = build_oracle("sd15", sd, ad, rank=128)         # fp32, CPU
    del sd, ad
    torch.cuda.empty_cache()
    return cfg, oracle, hip


@pytest.fixture(scope="module")
def sdxl(dev):
    from consistentid_amd.unet import HipUNet
    cfg, sd, ad = make_weights("sdxl", rank=16, device=dev)
    hip = HipUNet(cfg, sd, ad, device=dev)
    oracle = build_oracle("sdxl", sd, ad, rank=16)          # fp32, CPU (2.57 B parameters)
    del sd, ad
    torch.cuda.empty_cache()
    return cfg, oracle, hip


def test_sd15_unet_forward_full_size(dev, sd15):
    """Config (1) of BASELINE.json at UNet granularity: SD1.5, 512x512, B=1 (CFG batch 2), one forward
    of the real-size UNet against the fp32 CPU oracle."""
    from consistentid_amd import synth
    cfg, oracle, hip = sd15
    inp = synth.random_inputs(cfg, 1, 512, 512)
    ehs = torch.cat([inp["null"], inp["augmented"]])
    lat2 = torch.cat([inp["latents"]] * 2)
    with torch.no_grad():
        ref = oracle(lat2.float(), 981, ehs.float()).sample
    out = hip(lat2.to(dev), 981, encoder_hidden_states=ehs.to(dev)).sample
    torch.cuda.synchronize()
    with torch.no_grad():
        arm = half_arm(oracle, dev)(lat2.to(dev).half(), 981, ehs.to(dev).half()).sample
    check_vs_fp16_arm(out, ref, arm, "SD1.5 UNet forward 64x64 latents")


def test_sdxl_unet_forward_full_size(dev, sdxl):
    """SDXL (2.57 B parameters, 70 transformer layers), 1024x1024 (128x128 latents), B=1 (CFG batch 2), text_time
    conditioning -- one forward against the fp32 CPU oracle."""
    from consistentid_amd import synth
    cfg, oracle, hip = sdxl
    inp = synth.random_inputs(cfg, 1, 1024, 1024)
    ehs = torch.cat([inp["null"], inp["augmented"]])
    te = torch.cat([inp["pooled_null"], inp["pooled_augmented"]])
    lat2 = torch.cat([inp["latents"]] * 2)
    with torch.no_grad():
        ref = oracle(lat2.float(), 741, ehs.float(),
                     added_cond_kwargs={"text_embeds": te.float(), "time_ids": inp["time_ids"]}).sample
    out = hip(lat2.to(dev), 741, encoder_hidden_states=ehs.to(dev),
              added_cond_kwargs={"text_embeds": te.to(dev), "time_ids": inp["time_ids"].to(dev)}).sample
    torch.cuda.synchronize()
    with torch.no_grad():
        arm = half_arm(oracle, dev)(lat2.to(dev).half(), 741, ehs.to(dev).half(),
                                    added_cond_kwargs={"text_embeds": te.to(dev).half(),
                                                       "time_ids": inp["time_ids"].to(dev)}).sample
    check_vs_fp16_arm(out, ref, arm, "SDXL UNet forward 128x128 latents")


def test_config4_per_gpu_trajectory(dev, sdxl):
    """BASELINE config 4 at its per-GPU shape: SDXL 1024x1024, 2 images per GPU (CFG batch 4), 30 DDIM steps, embeds and
    pooled embeds switch after step 18, TWO unconditional sets (ref SDXL :586-590, :620-631), one hipGraph replayed 30
    times (second generation: replay only).  fp32 reference = the oracle loop in fp32 on the GPU."""
    from consistentid_amd import pipeline, synth
    from oracle import ddim, loop
    cfg, oracle, hip = sdxl
    B, steps, merge, g = 2, 30, 18, 7.5
    inp = synth.random_inputs(cfg, B, 1024, 1024)
    null_facial = (inp["null"].float() + 0.5 * torch.randn(inp["null"].shape, generator=torch.Generator().manual_seed(9))).half()

    def run(m, cast):
        c = lambda k: cast(inp[k].to(dev))
        return loop.denoise(m, ddim.DDIMScheduler(), c("latents"), c("null"), c("augmented"), c("text"),
                            num_inference_steps=steps, guidance_scale=g, start_merge_step=merge,
                            null_embeds_post=cast(null_facial.to(dev)), add_text_embeds_null=c("pooled_null"),
                            add_text_embeds_text=c("pooled_text"), add_text_embeds_aug=c("pooled_augmented"),
                            add_time_ids=inp["time_ids"].to(dev))
    o32 = copy.deepcopy(oracle).to(dev)
    ref = run(o32, lambda t: t.float())
    del o32
    torch.cuda.empty_cache()
    arm_m = half_arm(oracle, dev)
    arm = run(arm_m, lambda t: t.half())
    del arm_m
    torch.cuda.empty_cache()
    pipe = pipeline.ConsistentIDStableDiffusionXLPipeline(hip, use_graph=True)
    pe4 = torch.cat([inp["null"], inp["augmented"], inp["text"], null_facial]).to(dev)
    for _ in range(2):
        out = pipe(prompt_embeds=pe4, latents=inp["latents"].to(dev), num_inference_steps=steps, guidance_scale=g,
                   start_merge_step=merge, output_type="latent", pooled_prompt_embeds=inp["pooled_augmented"],
                   pooled_prompt_embeds_text_only=inp["pooled_text"], negative_pooled_prompt_embeds=inp["pooled_null"],
                   add_time_ids=inp["time_ids"]).images
        torch.cuda.synchronize()
        e, ea = check_vs_fp16_arm(out, ref, arm, "config 4 (per-GPU shape): SDXL 30-step trajectory, final latents")
    print(f"[drift] config 4 end of trajectory: ours {e:.3e}, stock-fp16 arm {ea:.3e} (rel L2 vs the fp32 oracle loop)")


def test_config5_controlnet_inpaint_trajectory(dev, sd15):
    """BASELINE config 5 at full size: SD1.5 + the 361 M-parameter ControlNet encoder INSIDE the captured step + inpaint
    mask blend, batch 8 (UNet CFG batch 16, ControlNet batch 8), 12 DDIM steps, embeds switch after step 5, conditioning
    scale 0.5, ControlNet switched off for the last two steps (control_guidance_end 0.85: the keep window, CN :364-371).
    fp32 reference = the oracle loop (UNet + ControlNet) in fp32 on the GPU."""
    from consistentid_amd import pipeline, synth
    from consistentid_amd.controlnet import HipControlNet
    from oracle import ddim, loop
    from oracle import unet as ounet
    from oracle.controlnet import ControlNetModel
    cfg, oracle, hip = sd15
    cn_sd = synth.random_controlnet_state_dict(cfg, seed=4, device=dev)
    h_cn = HipControlNet(cfg, cn_sd, device=dev)
    o_cn = ControlNetModel(ounet.sd15_config())
    o_cn.load_state_dict({k: v.detach().cpu().float() for k, v in cn_sd.items()}, strict=True)
    o_cn.eval()
    del cn_sd
    B, steps, merge, g = 8, 12, 5, 7.5
    inp = synth.random_inputs(cfg, B, 512, 512)
    gen = torch.Generator().manual_seed(23)
    img = torch.rand(B, 3, 512, 512, generator=gen).half()
    init = torch.randn(B, 4, 64, 64, generator=gen).half()
    noise = torch.randn(B, 4, 64, 64, generator=gen).half()
    mask = torch.zeros(B, 1, 64, 64)
    mask[:, :, 16:48, 16:48] = 1.0                      # centred 32 x 32 ones in 64 x 64
    mask = mask.half()

    def run(mu, mc, cast):
        c = lambda t: cast(t.to(dev))
        return loop.denoise(mu, ddim.DDIMScheduler(), c(inp["latents"]), c(inp["null"]), c(inp["augmented"]), c(inp["text"]),
                            num_inference_steps=steps, guidance_scale=g, start_merge_step=merge, inpaint_mask=c(mask),
                            inpaint_init=c(init), inpaint_noise=c(noise), controlnet=mc, control_image=c(img),
                            conditioning_scale=0.5, control_guidance_start=0.0, control_guidance_end=0.85)
    u32, c32 = copy.deepcopy(oracle).to(dev), copy.deepcopy(o_cn).to(dev)
    ref = run(u32, c32, lambda t: t.float())
    del u32, c32
    torch.cuda.empty_cache()
    arm = run(half_arm(oracle, dev), half_arm(o_cn, dev), lambda t: t.half())
    torch.cuda.empty_cache()
    pipe = pipeline.StableDiffusionControlNetInpaintConsistentIDPipeline(hip, controlnet=h_cn, use_graph=True)
    pe = torch.cat([inp["null"], inp["augmented"], inp["text"]]).to(dev)
    for _ in range(2):
        out = pipe(prompt_embeds=pe, latents=inp["latents"].to(dev), control_image=img.to(dev), num_inference_steps=steps,
                   guidance_scale=g, start_merge_step=merge, output_type="latent", image_latents=init.to(dev),
                   noise=noise.to(dev), mask_latents=mask.to(dev), controlnet_conditioning_scale=0.5,
                   control_guidance_end=0.85).images
        torch.cuda.synchronize()
        e, ea = check_vs_fp16_arm(out, ref, arm, "config 5: SD1.5 + native ControlNet + inpaint blend, batch 8, final latents")
    keep = (1 - mask.float()).to(dev).expand_as(out).bool()
    assert torch.equal(out[keep], init.to(dev)[keep]), "outside the mask the result is the initial image's latents"
    print(f"[drift] config 5 end of trajectory: ours {e:.3e}, stock-fp16 arm {ea:.3e} (rel L2 vs the fp32 oracle loop)")


def test_engines_rebuilt_from_the_broadcast_arena_are_bit_identical(dev, sd15):
    """What a rank != 0 does after distributed.broadcast_weights (bench.py): packed weights -> flat fp16 arena ->
    (shape, offset) views -> PackedUNet.from_tensors -> HipUNet / HipControlNet.  The rebuilt engines must compute the
    same bits as the source engines (fp32 fold vectors travel as fp16 bit patterns, meta carries the non-tensor state)."""
    from consistentid_amd import distributed, synth
    from consistentid_amd.controlnet import HipControlNet
    from consistentid_amd.unet import HipUNet
    from consistentid_amd.weights import PackedUNet
    cfg, _, hip = sd15

    def rebuild(engine, cls):
        keys = sorted(engine.W)
        flat, meta = distributed.flatten([engine.W[k] for k in keys])
        named = dict(zip(keys, distributed.unflatten(flat.clone(), meta)))      # a different allocation, like a receiver's
        m = engine.packed.meta()
        import json, pickle
        m = pickle.loads(pickle.dumps(m))                                       # the meta travels by broadcast_object_list
        return cls(cfg, device=dev, packed=PackedUNet.from_tensors(cfg, named, m, dev))
    twin = rebuild(hip, HipUNet)
    inp = synth.random_inputs(cfg, 2, 512, 512)
    ehs = torch.cat([inp["null"], inp["augmented"]]).to(dev)
    lat2 = torch.cat([inp["latents"]] * 2).to(dev)
    a = hip(lat2, 441, encoder_hidden_states=ehs).sample.clone()
    b = twin(lat2, 441, encoder_hidden_states=ehs).sample
    torch.cuda.synchronize()
    assert torch.equal(a, b), "UNet rebuilt from the arena differs from the source engine"
    del twin
    h_cn = HipControlNet(cfg, synth.random_controlnet_state_dict(cfg, seed=4, device=dev), device=dev)
    t_cn = rebuild(h_cn, HipControlNet)
    img = torch.rand(2, 3, 512, 512, generator=torch.Generator().manual_seed(3)).half().to(dev)
    kw = dict(encoder_hidden_states=inp["augmented"].to(dev), controlnet_cond=img, conditioning_scale=0.5, return_dict=False)
    d0, m0 = h_cn(inp["latents"].to(dev), 441, **kw)
    d1, m1 = t_cn(inp["latents"].to(dev), 441, **kw)
    torch.cuda.synchronize()
    assert torch.equal(m0, m1) and all(torch.equal(x, y) for x, y in zip(d0, d1)), "ControlNet rebuilt from the arena differs"


def test_bench_two_ranks_on_one_gpu(dev):
    """bench.py --gpus 2 end to end on this one GPU (CID_BENCH_SHARE_GPU folds the ranks onto device 0, gloo carries the
    weight broadcast instead of RCCL): self-spawn under torch.distributed.run, arena broadcast, rank 1 rebuilds its
    engine from the arena, image sharding by global index, barrier + max-over-ranks timing, one JSON line from rank 0."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, CID_BENCH_SHARE_GPU="1", CID_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--ddim-steps", "6",
                        "--batch-per-gpu", "4", "--no-cpu-baseline", "--no-torch-baseline", "--no-roofline"], capture_output=True, text=True, env=env,
                       timeout=900, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 8 and res["value"] > 0 and res["scaling"] == "weak"


def test_rccl_backend_executes_on_one_rank(dev):
    """backend="nccl" IS RCCL on ROCm.  Two RCCL ranks cannot share one GPU (duplicate-device communicators are refused), so
    on this one-GPU rig the RCCL code path runs as a ONE-rank group: communicator init, the bucketed arena broadcast
    (several buckets, one boundary inside the tensor list), barrier and the latent all-gather on device tensors.  What it
    proves: the product's collective calls execute on RCCL with HIP memory; what it cannot prove: xGMI transport."""
    import os
    import subprocess
    import sys
    import textwrap
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    code = textwrap.dedent("""
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        from consistentid_amd import distributed
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        assert dist.get_backend() == "nccl"
        g = torch.Generator(device="cuda").manual_seed(0)
        named = {f"w{i}": torch.randn(1 << 18, generator=g, device="cuda").half() for i in range(7)}
        want = {k: v.clone() for k, v in named.items()}
        got = distributed.broadcast_weights(named, torch.device("cuda:0"), src=0, bucket_bytes=1 << 20, single_rank_too=True)
        dist.barrier()
        lat = torch.randn(4, 4, 64, 64, generator=g, device="cuda").half()
        allv = distributed.all_gather_latents(lat, 4, single_rank_too=True)
        torch.cuda.synchronize()
        assert all(torch.equal(got[k], want[k]) for k in want) and torch.equal(allv, lat)
        dist.destroy_process_group()
        print("RCCL-OK")
    """ % str(root))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_config2_trajectory_and_forwards(dev, sd15):
    """BASELINE config 2 end to end: 4 images, 50 DDIM steps, embeds switch after step 30, one hipGraph replayed 50 times.
    The fp32 trajectory is the oracle loop run in fp32 on the GPU (the CPU would need ~40 min for 50 CFG-batch-8
    forwards); the CPU fp32 oracle checks two single forwards at CFG batch 8, one before and one after the merge step."""
    from consistentid_amd import pipeline, synth
    from oracle import ddim, loop
    cfg, oracle, hip = sd15
    B, steps, merge, g = 4, 50, 30, 5.0
    inp = synth.random_inputs(cfg, B, 512, 512)
    o32 = copy.deepcopy(oracle).to(dev)
    snap = {}

    def keep(i, t, lat):
        if i == merge:                # latents entering step merge + 1, the first step that sees the augmented embeds
            snap["lat"] = lat.detach().clone()

    f = lambda k: inp[k].to(dev).float()
    ref = loop.denoise(o32, ddim.DDIMScheduler(), f("latents"), f("null"), f("augmented"), f("text"),
                       num_inference_steps=steps, guidance_scale=g, start_merge_step=merge, on_step=keep)
    del o32
    h = lambda k: inp[k].to(dev).half()
    arm_m = half_arm(oracle, dev)
    arm = loop.denoise(arm_m, ddim.DDIMScheduler(), h("latents"), h("null"), h("augmented"), h("text"),
                       num_inference_steps=steps, guidance_scale=g, start_merge_step=merge)
    pipe = pipeline.ConsistentIDStableDiffusionPipeline(hip, use_graph=True)
    pe = torch.cat([inp["null"], inp["augmented"], inp["text"]]).to(dev)
    out = None
    for _ in range(2):                # the second generation replays the captured graph
        out = pipe(prompt_embeds=pe, latents=inp["latents"].to(dev), num_inference_steps=steps, guidance_scale=g,
                   start_merge_step=merge, output_type="latent").images
        torch.cuda.synchronize()
        e, ea = check_vs_fp16_arm(out, ref, arm, "config 2: 50-step trajectory, final latents")
    print(f"[drift] config 2 end of trajectory: ours {e:.3e}, stock-fp16 arm {ea:.3e} (rel L2 vs the fp32 oracle loop)")

    # ---- two single forwards at CFG batch 8 against the fp32 CPU oracle
    sch = ddim.DDIMScheduler()
    sch.set_timesteps(steps)
    cases = [("before the merge step", inp["latents"], int(sch.timesteps[0]), inp["text"]),
             ("after the merge step", snap["lat"].half().cpu(), int(sch.timesteps[merge + 1]), inp["augmented"])]
    for what, lat, t, cond in cases:
        ehs = torch.cat([inp["null"], cond])
        lat2 = torch.cat([lat] * 2)
        with torch.no_grad():
            r = oracle(lat2.float(), t, ehs.float()).sample
            a = arm_m(lat2.to(dev).half(), t, ehs.to(dev).half()).sample
        o = hip(lat2.to(dev), t, encoder_hidden_states=ehs.to(dev)).sample
        torch.cuda.synchronize()
        check_vs_fp16_arm(o, r, a, f"config 2: one forward at CFG batch 8, t={t}, {what}")


def test_config3_shard_trajectory(dev, sd15):
    """BASELINE config 3 at its per-GPU shape -- what `bench.py --gpus N > 1` times on every rank: plain SD1.5, 512x512,
    8 images per GPU (CFG batch 16), 50 DDIM steps, embeds switch after step 30, one hipGraph replayed 50 times (second
    generation: replay only).  fp32 reference = the oracle loop in fp32 on the GPU."""
    from consistentid_amd import pipeline, synth
    from oracle import ddim, loop
    cfg, oracle, hip = sd15
    B, steps, merge, g = 8, 50, 30, 5.0
    inp = synth.random_inputs(cfg, B, 512, 512, seed_latents=2032, seed_embeds=9)

    def run(m, cast):
        c = lambda k: cast(inp[k].to(dev))
        return loop.denoise(m, ddim.DDIMScheduler(), c("latents"), c("null"), c("augmented"), c("text"),
                            num_inference_steps=steps, guidance_scale=g, start_merge_step=merge)
    o32 = copy.deepcopy(oracle).to(dev)
    ref = run(o32, lambda t: t.float())
    del o32
    torch.cuda.empty_cache()
    arm_m = half_arm(oracle, dev)
    arm = run(arm_m, lambda t: t.half())
    del arm_m
    torch.cuda.empty_cache()
    pipe = pipeline.ConsistentIDStableDiffusionPipeline(hip, use_graph=True)
    pe = torch.cat([inp["null"], inp["augmented"], inp["text"]]).to(dev)
    for _ in range(2):
        out = pipe(prompt_embeds=pe, latents=inp["latents"].to(dev), num_inference_steps=steps, guidance_scale=g,
                   start_merge_step=merge, output_type="latent").images
        torch.cuda.synchronize()
        e, ea = check_vs_fp16_arm(out, ref, arm, "config 3 (per-GPU shape): SD1.5 batch 8, 50-step trajectory, final latents")
    print(f"[drift] config 3 shard end of trajectory: ours {e:.3e}, stock-fp16 arm {ea:.3e} (rel L2 vs the fp32 oracle loop)")


def test_sdxl_reference_default_resolution_forward(dev, sdxl):
    """infer_SDXL.py:61-62 runs width 864 x height 1152: latents 144 x 108, 72 x 54 = 3888 tokens at the 640-channel level and
    36 x 27 = 972 at the 1280-channel level -- neither a multiple of the attention tiles (zero-padded token axis with masked
    pad keys at SDXL's widths and head counts), image rows of 108 / 54 / 27 pixels (no 256-token row tiles: the convolutions
    leave the halo kernels for the gather path).  One full-size forward against the fp32 CPU oracle."""
    from consistentid_amd import synth
    cfg, oracle, hip = sdxl
    inp = synth.random_inputs(cfg, 1, 1152, 864)
    assert inp["latents"].shape[-2:] == (144, 108)
    ehs = torch.cat([inp["null"], inp["augmented"]])
    te = torch.cat([inp["pooled_null"], inp["pooled_augmented"]])
    lat2 = torch.cat([inp["latents"]] * 2)
    with torch.no_grad():
        ref = oracle(lat2.float(), 741, ehs.float(),
                     added_cond_kwargs={"text_embeds": te.float(), "time_ids": inp["time_ids"]}).sample
    out = hip(lat2.to(dev), 741, encoder_hidden_states=ehs.to(dev),
              added_cond_kwargs={"text_embeds": te.to(dev), "time_ids": inp["time_ids"].to(dev)}).sample
    torch.cuda.synchronize()
    with torch.no_grad():
        arm = half_arm(oracle, dev)(lat2.to(dev).half(), 741, ehs.to(dev).half(),
                                    added_cond_kwargs={"text_embeds": te.to(dev).half(),
                                                       "time_ids": inp["time_ids"].to(dev)}).sample
    check_vs_fp16_arm(out, ref, arm, "SDXL UNet forward at 864x1152 (144x108 latents)")


def test_reference_default_resolution_forward(dev, sd15):
    """infer.py:63-64 runs 512 x 768: latents 64 x 96, 6144 / 1536 / 384 / 96 tokens per level -- the mid block's 96
    tokens are not a multiple of the attention tile (padded-token path at 1280 channels)."""
    from consistentid_amd import synth
    cfg, oracle, hip = sd15
    inp = synth.random_inputs(cfg, 1, 512, 768)
    assert inp["latents"].shape[-2:] == (64, 96)
    ehs = torch.cat([inp["null"], inp["augmented"]])
    lat2 = torch.cat([inp["latents"]] * 2)
    with torch.no_grad():
        ref = oracle(lat2.float(), 661, ehs.float()).sample
        arm = half_arm(oracle, dev)(lat2.to(dev).half(), 661, ehs.to(dev).half()).sample
    out = hip(lat2.to(dev), 661, encoder_hidden_states=ehs.to(dev)).sample
    torch.cuda.synchronize()
    check_vs_fp16_arm(out, ref, arm, "SD1.5 UNet forward at 512x768 (64x96 latents)")


def test_context_cache_survives_address_reuse(dev):
    """The diffusers-style call caches the projected K/V per encoder_hidden_states tensor.  A caller that frees one
    prompt's embeddings and allocates the next prompt's (same shape) typically gets the SAME address back from the
    caching allocator, version counter 0: a cache keyed on (address, version, shape) alone would serve the previous
    prompt's K/V.  The cache keeps its keyed tensor alive, so the address cannot be recycled while the entry is valid."""
    from consistentid_amd import synth
    from consistentid_amd.unet import HipUNet
    cfg, sd, ad = make_weights("tiny", rank=8)
    hip = HipUNet(cfg, sd, ad, device=dev)
    fresh = HipUNet(cfg, sd, ad, device=dev)
    side = cfg.sample_size * 8
    inp = synth.random_inputs(cfg, 1, side, side)
    lat2 = torch.cat([inp["latents"]] * 2).to(dev)
    g = torch.Generator().manual_seed(1)
    e1 = torch.randn(2, 81, cfg.cross_attention_dim, generator=g).half().to(dev)
    e2_host = torch.randn(2, 81, cfg.cross_attention_dim, generator=g).half()
    o1 = hip(lat2, 501, encoder_hidden_states=e1).sample.clone()
    addr = e1.data_ptr()
    del e1                                   # the caller drops the first prompt ...
    e2 = e2_host.to(dev)                     # ... and allocates the second one (same shape)
    o2 = hip(lat2, 501, encoder_hidden_states=e2).sample
    want = fresh(lat2, 501, encoder_hidden_states=e2_host.to(dev)).sample
    torch.cuda.synchronize()
    print(f"[cache] second prompt at the first one's address: {e2.data_ptr() == addr}")
    assert torch.equal(o2, want), "stale K/V served for a new prompt"
    assert rel_l2(o2, o1) > 1e-2             # the two prompts do give different outputs
    # and a processor-level plugin call: temporaries passed straight as arguments (a fresh tensor every step)
    from consistentid_amd import attention as pattn
    from oracle.unet import Attention
    C, Dc, heads = 64, cfg.cross_attention_dim, 2
    attn = Attention(C, Dc, heads).to(dev).half()
    proc = pattn.Consistent_IPAttProcessor(hidden_size=C, cross_attention_dim=Dc, rank=4).to(dev).half()
    x = torch.randn(2, 64, C, generator=g).half().to(dev)
    ea, eb = torch.randn(2, 81, Dc, generator=g).half(), torch.randn(2, 81, Dc, generator=g).half()
    ya = proc(attn, x, encoder_hidden_states=torch.cat([ea[:1], ea[1:]]).to(dev)).clone()
    yb = proc(attn, x, encoder_hidden_states=torch.cat([eb[:1], eb[1:]]).to(dev)).clone()
    proc2 = pattn.Consistent_IPAttProcessor(hidden_size=C, cross_attention_dim=Dc, rank=4).to(dev).half()
    proc2.load_state_dict(proc.state_dict())
    yb_want = proc2(attn, x, encoder_hidden_states=eb.to(dev))
    torch.cuda.synchronize()
    assert torch.equal(yb, yb_want) and not torch.equal(ya, yb)
