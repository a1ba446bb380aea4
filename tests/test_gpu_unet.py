"""End-to-end parity on the GPU: processors vs the reference-generated golden vectors, UNet
forwards and denoise loops vs the CPU oracle on identical synthetic weights / seeds."""
import numpy as np
import pytest
import torch

from conftest import check_close, check_vs_fp16_arm, dev_half, half_arm
from oracle_utils import build_oracle, make_weights
from test_oracle_golden import GOLD, load_case

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------------------------- plugin boundary
@pytest.mark.parametrize("path", GOLD, ids=lambda p: p.stem)
def test_processors_match_reference_golden(dev, path):
    """consistentid_amd.attention.* driven through the diffusers processor protocol, compared with
    outputs of the REAL reference processors (tests/golden/make_golden.py)."""
    from consistentid_amd import attention as pattn
    c = load_case(path, torch.float32)
    B, N, C, heads, Dc, L, rank = c["meta"]
    p1 = pattn.Consistent_AttProcessor(hidden_size=C, cross_attention_dim=None, rank=rank)
    p2 = pattn.Consistent_IPAttProcessor(hidden_size=C, cross_attention_dim=Dc, rank=rank, scale=c["ip_scale"],
                                         num_tokens=4)
    p1.load_state_dict(c["p1"].state_dict(), strict=True)     # same keys as the reference's adapter_modules
    p2.load_state_dict(c["p2"].state_dict(), strict=True)
    p1, p2 = p1.to(dev).half(), p2.to(dev).half()
    import copy
    a1, a2 = copy.deepcopy(c["attn1"]).to(dev).half(), copy.deepcopy(c["attn2"]).to(dev).half()
    hid, ehs = c["hidden"].to(dev).half(), c["ehs"].to(dev).half()
    o1 = p1(a1, hid)
    o2 = p2(a2, hid, encoder_hidden_states=ehs)
    torch.cuda.synchronize()
    assert o1.shape == hid.shape and o1.dtype == torch.float16 and o1.device == hid.device
    # reference-precision arm: the oracle restatement of the same processors in fp16 on the GPU
    with torch.no_grad():
        arm1 = copy.deepcopy(c["p1"]).to(dev).half()(a1, hid)
        arm2 = copy.deepcopy(c["p2"]).to(dev).half()(a2, hid, encoder_hidden_states=ehs)
    check_vs_fp16_arm(o1, c["out_self"], arm1, f"Consistent_AttProcessor {path.stem}")
    check_vs_fp16_arm(o2, c["out_ip"], arm2, f"Consistent_IPAttProcessor {path.stem}")
    # runtime-mutable .scale (ref set_scale, pipline_StableDiffusion_ConsistentID.py:211-214)
    p2.scale = 0.0
    o3 = p2(a2, hid, encoder_hidden_states=ehs)
    c["p2"].scale = 0.0
    with torch.no_grad():
        ref3 = c["p2"](c["attn2"], c["hidden"], encoder_hidden_states=c["ehs"])
    check_close(o3, ref3, "Consistent_IPAttProcessor scale=0", tol_l2=2e-3, tol_max=8e-3)


# ----------------------------------------------------------------------------- UNet forward
def _unet_pair(name, dev, rank=8):
    from consistentid_amd.unet import HipUNet
    cfg, sd, ad = make_weights(name, rank=rank)
    oracle = build_oracle(name, sd, ad, rank=rank)
    hip = HipUNet(cfg, sd, ad, device=dev)
    return cfg, oracle, hip


@pytest.mark.parametrize("name", ["tiny", "tinyxl"])
def test_tiny_unet_forward(dev, name):
    from consistentid_amd import synth
    cfg, oracle, hip = _unet_pair(name, dev)
    B = 2
    side = cfg.sample_size * 8
    inp = synth.random_inputs(cfg, B, side, side)
    ehs = torch.cat([inp["null"], inp["text"]])
    kw_o, kw_h = {}, {}
    if name == "tinyxl":
        te = torch.cat([inp["pooled_null"], inp["pooled_text"]])
        kw_o = dict(added_cond_kwargs={"text_embeds": te.float(), "time_ids": inp["time_ids"]})
        kw_h = dict(added_cond_kwargs={"text_embeds": te.to(dev), "time_ids": inp["time_ids"].to(dev)})
    lat2 = torch.cat([inp["latents"]] * 2)
    with torch.no_grad():
        ref = oracle(lat2.float(), 501, ehs.float(), **kw_o).sample
    out = hip(lat2.to(dev), 501, encoder_hidden_states=ehs.to(dev), cross_attention_kwargs={}, **kw_h).sample
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    arm_m = half_arm(oracle, dev)
    with torch.no_grad():
        arm = arm_m(lat2.to(dev).half(), 501, ehs.to(dev).half(), **dev_half(kw_o, dev)).sample
    check_vs_fp16_arm(out, ref, arm, f"{name} UNet forward")
    # ControlNet-style extra residuals (CN :418-425), batch-B residuals broadcast over the CFG halves
    if name == "tiny":
        g = torch.Generator().manual_seed(5)
        shapes = [(64, 32), (64, 32), (64, 16), (128, 16), (128, 8), (128, 8)]
        dres = [(torch.randn(B, c, h, h, generator=g) * 0.1).half() for c, h in shapes]
        mres = (torch.randn(B, 128, 8, 8, generator=g) * 0.1).half()
        with torch.no_grad():
            ref2 = oracle(lat2.float(), 501, ehs.float(),
                          down_block_additional_residuals=[torch.cat([d, d]).float() for d in dres],
                          mid_block_additional_residual=torch.cat([mres, mres]).float()).sample
        out2 = hip(lat2.to(dev), 501, encoder_hidden_states=ehs.to(dev),
                   down_block_additional_residuals=[d.to(dev) for d in dres],
                   mid_block_additional_residual=mres.to(dev)).sample
        torch.cuda.synchronize()
        with torch.no_grad():
            arm2 = arm_m(lat2.to(dev).half(), 501, ehs.to(dev).half(),
                         down_block_additional_residuals=[torch.cat([d, d]).to(dev) for d in dres],
                         mid_block_additional_residual=torch.cat([mres, mres]).to(dev)).sample
        check_vs_fp16_arm(out2, ref2, arm2, "tiny UNet forward + ControlNet residuals")


@pytest.mark.parametrize("name,use_graph", [("tiny", False), ("tiny", True), ("tinyxl", True)])
def test_tiny_denoise_loop(dev, name, use_graph):
    """ConsistentIDStableDiffusion[XL]Pipeline.__call__ hot loop vs the oracle loop, 4 DDIM steps,
    embed switch after start_merge_step=1, user-supplied latents (seed-exact parity)."""
    from consistentid_amd import pipeline, synth
    from oracle import ddim, loop
    cfg, oracle, hip = _unet_pair(name, dev)
    B, steps, merge, g = 2, 4, 1, 5.0
    side = cfg.sample_size * 8
    inp = synth.random_inputs(cfg, B, side, side)
    f = lambda k: inp[k].float()
    kw = {}
    if name == "tinyxl":
        kw = dict(add_text_embeds_null=f("pooled_null"), add_text_embeds_text=f("pooled_text"),
                  add_text_embeds_aug=f("pooled_augmented"), add_time_ids=inp["time_ids"])
    ref = loop.denoise(oracle, ddim.DDIMScheduler(), f("latents"), f("null"), f("augmented"), f("text"),
                       num_inference_steps=steps, guidance_scale=g, start_merge_step=merge, **kw)
    pe = torch.cat([inp["null"], inp["augmented"], inp["text"]]).to(dev)
    if name == "tinyxl":
        pipe = pipeline.ConsistentIDStableDiffusionXLPipeline(hip, use_graph=use_graph)
        res = pipe(prompt_embeds=pe, latents=inp["latents"].to(dev), num_inference_steps=steps, guidance_scale=g,
                   start_merge_step=merge, output_type="latent", pooled_prompt_embeds=inp["pooled_augmented"],
                   pooled_prompt_embeds_text_only=inp["pooled_text"],
                   negative_pooled_prompt_embeds=inp["pooled_null"], add_time_ids=inp["time_ids"])
    else:
        pipe = pipeline.ConsistentIDStableDiffusionPipeline(hip, use_graph=use_graph)
        res = pipe(prompt_embeds=pe, latents=inp["latents"].to(dev), num_inference_steps=steps, guidance_scale=g,
                   start_merge_step=merge, output_type="latent")
    torch.cuda.synchronize()
    arm_m = half_arm(oracle, dev)
    h = lambda k: inp[k].to(dev).half()
    arm = loop.denoise(arm_m, ddim.DDIMScheduler(), h("latents"), h("null"), h("augmented"), h("text"),
                       num_inference_steps=steps, guidance_scale=g, start_merge_step=merge, **dev_half(kw, dev))
    check_vs_fp16_arm(res.images, ref, arm, f"{name} 4-step denoise graph={use_graph}")
    if use_graph and name == "tiny":   # second generation replays the cached graph with new inputs
        inp2 = synth.random_inputs(cfg, B, side, side, seed_latents=7, seed_embeds=8)
        f2 = lambda k: inp2[k].float()
        ref2 = loop.denoise(oracle, ddim.DDIMScheduler(), f2("latents"), f2("null"), f2("augmented"), f2("text"),
                            num_inference_steps=steps, guidance_scale=g, start_merge_step=merge)
        res2 = pipe(prompt_embeds=torch.cat([inp2["null"], inp2["augmented"], inp2["text"]]).to(dev),
                    latents=inp2["latents"].to(dev), num_inference_steps=steps, guidance_scale=g,
                    start_merge_step=merge, output_type="latent")
        torch.cuda.synchronize()
        h2 = lambda k: inp2[k].to(dev).half()
        arm2 = loop.denoise(arm_m, ddim.DDIMScheduler(), h2("latents"), h2("null"), h2("augmented"), h2("text"),
                            num_inference_steps=steps, guidance_scale=g, start_merge_step=merge)
        check_vs_fp16_arm(res2.images, ref2, arm2, "tiny 4-step denoise, graph replayed on new inputs")


def test_tiny_controlnet_inpaint_loop(dev):
    from consistentid_amd import pipeline, synth
    from oracle import ddim, loop
    cfg, oracle, hip = _unet_pair("tiny", dev)
    B, steps = 2, 3
    inp = synth.random_inputs(cfg, B, 256, 256)
    f = lambda k: inp[k].float()
    g = torch.Generator().manual_seed(11)
    shapes = [(64, 32), (64, 32), (64, 16), (128, 16), (128, 8), (128, 8)]
    dres = [(torch.randn(B, c, h, h, generator=g) * 0.1).half() for c, h in shapes]
    mres = (torch.randn(B, 128, 8, 8, generator=g) * 0.1).half()
    mask = torch.zeros(B, 1, 32, 32)
    mask[:, :, 8:24, 8:24] = 1.0
    init, noise = (torch.randn(B, 4, 32, 32, generator=g)).half(), torch.randn(B, 4, 32, 32, generator=g).half()
    ref = loop.denoise(oracle, ddim.DDIMScheduler(), f("latents"), f("null"), f("augmented"), f("text"),
                       num_inference_steps=steps, guidance_scale=7.5, start_merge_step=0,
                       down_residuals=[d.float() for d in dres], mid_residual=mres.float(),
                       inpaint_mask=mask, inpaint_init=init.float(), inpaint_noise=noise.float())
    tok = lambda r: r.permute(0, 2, 3, 1).reshape(r.shape[0], -1, r.shape[1]).contiguous()
    pipe = pipeline.StableDiffusionControlNetInpaintConsistentIDPipeline(hip, use_graph=True)
    res = pipe(prompt_embeds=torch.cat([inp["null"], inp["augmented"], inp["text"]]).to(dev),
               latents=inp["latents"].to(dev), num_inference_steps=steps, guidance_scale=7.5, start_merge_step=0,
               output_type="latent", image_latents=init, noise=noise, mask_latents=mask,
               down_block_res_samples=[tok(d) for d in dres], mid_block_res_sample=tok(mres))
    torch.cuda.synchronize()
    h = lambda k: inp[k].to(dev).half()
    arm = loop.denoise(half_arm(oracle, dev), ddim.DDIMScheduler(), h("latents"), h("null"), h("augmented"), h("text"),
                       num_inference_steps=steps, guidance_scale=7.5, start_merge_step=0,
                       down_residuals=[d.to(dev) for d in dres], mid_residual=mres.to(dev),
                       inpaint_mask=mask.to(dev).half(), inpaint_init=init.to(dev), inpaint_noise=noise.to(dev))
    check_vs_fp16_arm(res.images, ref, arm, "tiny ControlNet-inpaint 3-step loop")


def test_pipeline_rejects_out_of_scope_inputs(dev):
    from consistentid_amd import pipeline
    cfg, _, hip = _unet_pair("tiny", dev)
    pipe = pipeline.ConsistentIDStableDiffusionPipeline(hip)
    with pytest.raises(NotImplementedError):
        pipe(prompt="a photo of a man", input_id_images=[object()])
    with pytest.raises(ValueError):     # pixel outputs need the pipeline to be built with a VAE decoder
        pipe(prompt_embeds=torch.zeros(3, 81, cfg.cross_attention_dim), latents=torch.zeros(1, 4, 32, 32), output_type="pil")


def test_tiny_unet_odd_resolution(dev):
    """Latent sizes whose token counts are not multiples of the attention tiles (24x40 latents: 960 / 240 / 60 tokens per
    level): the transformer blocks run on a zero-padded token axis with masked pad keys; result vs the oracle."""
    from consistentid_amd import synth
    cfg, oracle, hip = _unet_pair("tiny", dev)
    B = 2
    inp = synth.random_inputs(cfg, B, 24 * 8, 40 * 8)
    assert inp["latents"].shape[-2:] == (24, 40)
    ehs = torch.cat([inp["null"], inp["augmented"]])
    lat2 = torch.cat([inp["latents"]] * 2)
    with torch.no_grad():
        ref = oracle(lat2.float(), 333, ehs.float()).sample
    out = hip(lat2.to(dev), 333, encoder_hidden_states=ehs.to(dev), cross_attention_kwargs={}).sample
    torch.cuda.synchronize()
    with torch.no_grad():
        arm = half_arm(oracle, dev)(lat2.to(dev).half(), 333, ehs.to(dev).half()).sample
    check_vs_fp16_arm(out, ref, arm, "tiny UNet forward, 24x40 latents")


@pytest.mark.parametrize("inpaint", [False, True])
def test_tiny_denoise_loop_euler(dev, inpaint):
    """The scheduler of the reference's canonical scripts (infer.py:33: EulerDiscreteScheduler.from_config): model-input
    scale inside conv_in, x += (sigma_next - sigma) * eps, initial latents scaled by init_noise_sigma, inpaint re-noising
    with sigma_next -- against the oracle loop driven by the oracle's Euler restatement."""
    from consistentid_amd import pipeline, scheduler, synth
    from oracle import ddim, loop
    cfg, oracle, hip = _unet_pair("tiny", dev)
    B, steps, merge, g = 2, 5, 2, 5.0
    side = cfg.sample_size * 8
    inp = synth.random_inputs(cfg, B, side, side)
    f = lambda k: inp[k].float()
    osch = ddim.EulerDiscreteScheduler()
    osch.set_timesteps(steps)          # the pipelines set the timesteps before scaling the initial noise (ref :510, :517)
    kw_o, kw_h = {}, {}
    if inpaint:
        gen = torch.Generator().manual_seed(3)
        init = torch.randn(B, 4, side // 8, side // 8, generator=gen).half()
        noise = torch.randn(B, 4, side // 8, side // 8, generator=gen).half()
        mask = (torch.rand(B, 1, side // 8, side // 8, generator=gen) > 0.5).half()
        kw_o = dict(inpaint_mask=mask.float(), inpaint_init=init.float(), inpaint_noise=noise.float())
        kw_h = dict(image_latents=init.to(dev), noise=noise.to(dev), mask_latents=mask.to(dev))
    ref = loop.denoise(oracle, osch, f("latents") * osch.init_noise_sigma, f("null"), f("augmented"), f("text"),
                       num_inference_steps=steps, guidance_scale=g, start_merge_step=merge, **kw_o)
    cls = pipeline.StableDiffusionInpaintConsistentIDPipeline if inpaint else pipeline.ConsistentIDStableDiffusionPipeline
    pipe = cls(hip, scheduler=scheduler.EulerDiscreteScheduler())
    pe = torch.cat([inp["null"], inp["augmented"], inp["text"]]).to(dev)
    h = lambda k: inp[k].to(dev).half()
    asch = ddim.EulerDiscreteScheduler()
    asch.set_timesteps(steps)
    arm = loop.denoise(half_arm(oracle, dev), asch, h("latents") * asch.init_noise_sigma, h("null"), h("augmented"), h("text"),
                       num_inference_steps=steps, guidance_scale=g, start_merge_step=merge, **dev_half(kw_o, dev))
    for _ in range(2):
        out = pipe(prompt_embeds=pe, latents=inp["latents"].to(dev), num_inference_steps=steps, guidance_scale=g,
                   start_merge_step=merge, output_type="latent", **kw_h).images
        torch.cuda.synchronize()
        check_vs_fp16_arm(out, ref, arm, f"tiny Euler denoise loop (inpaint={inpaint})")


@pytest.mark.parametrize("given_latents", [True, False])
def test_tiny_inpaint_strength_window(dev, given_latents):
    """``strength`` < 1 (get_timesteps + prepare_latents of the inpaint pipelines, inpaint ref :246-275): the loop runs the last
    int(S * strength) timesteps, the embed switch counts from the truncated list, and without user latents the start is
    add_noise(image_latents, noise, first timestep).  Oracle: loop.denoise(strength=...)."""
    from consistentid_amd import pipeline, synth
    from oracle import ddim, loop
    cfg, oracle, hip = _unet_pair("tiny", dev)
    B, steps, merge, g, strength = 2, 10, 2, 7.5, 0.6
    side = cfg.sample_size * 8
    inp = synth.random_inputs(cfg, B, side, side)
    gen = torch.Generator().manual_seed(31)
    init = torch.randn(B, 4, side // 8, side // 8, generator=gen).half()
    noise = torch.randn(B, 4, side // 8, side // 8, generator=gen).half()
    mask = (torch.rand(B, 1, side // 8, side // 8, generator=gen) > 0.5).half()
    osch = ddim.DDIMScheduler()
    osch.set_timesteps(steps)
    t0 = int(osch.timesteps[steps - int(steps * strength)])
    start = inp["latents"].float() if given_latents else osch.add_noise(init.float(), noise.float(), t0)
    f = lambda k: inp[k].float()
    kw = dict(num_inference_steps=steps, guidance_scale=g, start_merge_step=merge, strength=strength)
    ref = loop.denoise(oracle, ddim.DDIMScheduler(), start, f("null"), f("augmented"), f("text"),
                       inpaint_mask=mask.float(), inpaint_init=init.float(), inpaint_noise=noise.float(), **kw)
    h = lambda k: inp[k].to(dev).half()
    arm = loop.denoise(half_arm(oracle, dev), ddim.DDIMScheduler(), start.to(dev).half(), h("null"), h("augmented"), h("text"),
                       inpaint_mask=mask.to(dev), inpaint_init=init.to(dev), inpaint_noise=noise.to(dev), **kw)
    pipe = pipeline.StableDiffusionInpaintConsistentIDPipeline(hip)
    seen = []
    out = pipe(prompt_embeds=torch.cat([inp["null"], inp["augmented"], inp["text"]]).to(dev),
               latents=inp["latents"].to(dev) if given_latents else None, num_inference_steps=steps, guidance_scale=g,
               start_merge_step=merge, strength=strength, output_type="latent", image_latents=init.to(dev),
               noise=noise.to(dev), mask_latents=mask.to(dev), callback=lambda i, t, l: seen.append((i, t))).images
    torch.cuda.synchronize()
    assert [i for i, _ in seen] == list(range(int(steps * strength))) and seen[0][1] == t0
    check_vs_fp16_arm(out, ref, arm, f"tiny inpaint loop, strength {strength}, latents given={given_latents}")


def test_tinyxl_two_unconditional_sets(dev):
    """The SDXL loop switches BOTH halves at the merge step: cat([negative text embeds, uncond ID tokens]) before,
    cat([FacialEncoder(negative embeds), uncond ID tokens]) after (ref SDXL :586-590, :620-631).  Four embed sets in the
    K/V context, selected per phase."""
    from consistentid_amd import pipeline, synth
    from oracle import ddim, loop
    cfg, oracle, hip = _unet_pair("tinyxl", dev)
    B, steps, merge, g = 2, 5, 1, 7.5
    side = cfg.sample_size * 8
    inp = synth.random_inputs(cfg, B, side, side)
    null_facial = (inp["null"].float() + 0.5 * torch.randn(inp["null"].shape, generator=torch.Generator().manual_seed(9))).half()
    f = lambda k: inp[k].float()
    kw = dict(add_text_embeds_null=f("pooled_null"), add_text_embeds_text=f("pooled_text"),
              add_text_embeds_aug=f("pooled_augmented"), add_time_ids=inp["time_ids"])
    ref = loop.denoise(oracle, ddim.DDIMScheduler(), f("latents"), f("null"), f("augmented"), f("text"),
                       num_inference_steps=steps, guidance_scale=g, start_merge_step=merge,
                       null_embeds_post=null_facial.float(), **kw)
    same = loop.denoise(oracle, ddim.DDIMScheduler(), f("latents"), f("null"), f("augmented"), f("text"),
                        num_inference_steps=steps, guidance_scale=g, start_merge_step=merge, **kw)
    assert (ref - same).norm() / same.norm() > 1e-2           # the second null matters
    h = lambda k: inp[k].to(dev).half()
    arm = loop.denoise(half_arm(oracle, dev), ddim.DDIMScheduler(), h("latents"), h("null"), h("augmented"), h("text"),
                       num_inference_steps=steps, guidance_scale=g, start_merge_step=merge,
                       null_embeds_post=null_facial.to(dev), **dev_half(kw, dev))
    pipe = pipeline.ConsistentIDStableDiffusionXLPipeline(hip)
    pe4 = torch.cat([inp["null"], inp["augmented"], inp["text"], null_facial]).to(dev)
    for how in ("four sets in prompt_embeds", "negative_prompt_embeds_facial"):
        extra = {} if how.startswith("four") else dict(negative_prompt_embeds_facial=null_facial.to(dev))
        pe = pe4 if how.startswith("four") else pe4[:3 * B]
        out = pipe(prompt_embeds=pe, latents=inp["latents"].to(dev), num_inference_steps=steps, guidance_scale=g,
                   start_merge_step=merge, output_type="latent", pooled_prompt_embeds=inp["pooled_augmented"],
                   pooled_prompt_embeds_text_only=inp["pooled_text"], negative_pooled_prompt_embeds=inp["pooled_null"],
                   add_time_ids=inp["time_ids"], **extra).images
        torch.cuda.synchronize()
        check_vs_fp16_arm(out, ref, arm, f"tiny SDXL loop with two unconditional sets ({how})")


@pytest.mark.parametrize("pipe_kind,euler", [("inpaint", True), ("controlnet", False)])
def test_tiny_inpaint_nine_channel_unet(dev, pipe_kind, euler):
    """Row a11's 9-channel branch: ``torch.cat([latent_model_input, mask, masked_image_latents], dim=1)`` in front of a
    UNet with ``in_channels == 9`` (pipelines/StableDIffusionInpaint_ConsistentID.py:320-321,
    StableDIffusionControlNetInpaint_ConsistentID.py:415-416).  The engine's conv_in reads the latents and
    cat([mask, masked_image_latents]) from two tensors; with Euler the model-input scale must reach the latents only
    (the reference concatenates after scale_model_input).  ControlNet flavour: precomputed residuals.
    The reference does NOT blend here: ``if num_channels_unet == 4:`` guards the whole per-step mask blend (inpaint ref
    :340-353, CN :437-449) and ``return_image_latents = num_channels_unet == 4`` (:258, CN :319) -- so the oracle loop runs
    without ``inpaint_mask`` and the pipeline is called WITHOUT image_latents / noise; passing them must change nothing."""
    from consistentid_amd import pipeline, scheduler, synth
    from consistentid_amd.unet import HipUNet
    from oracle import ddim, loop
    cfg, sd, ad = make_weights("tiny", rank=8, in_channels=9)
    oracle = build_oracle("tiny", sd, ad, rank=8, in_channels=9)
    hip = HipUNet(cfg, sd, ad, device=dev)
    assert hip.in_channels == 9
    B, steps, merge, g = 2, 4, 1, 7.5
    side = cfg.sample_size * 8
    h8 = side // 8
    inp = synth.random_inputs(make_weights("tiny")[0], B, side, side)        # 4-channel latents
    gen = torch.Generator().manual_seed(41)
    init = torch.randn(B, 4, h8, h8, generator=gen).half()
    noise = torch.randn(B, 4, h8, h8, generator=gen).half()
    mask = (torch.rand(B, 1, h8, h8, generator=gen) > 0.5).half()
    masked = (init.float() * (1 - mask.float())).half()                      # VAE latents of image * (mask < 0.5)
    extra = torch.cat([mask, masked], dim=1)
    osch = (ddim.EulerDiscreteScheduler if euler else ddim.DDIMScheduler)
    s0 = osch(); s0.set_timesteps(steps)
    f = lambda k: inp[k].float()
    kw_o = dict(num_inference_steps=steps, guidance_scale=g, start_merge_step=merge)      # no blend: num_channels_unet == 9
    kw_h = {}
    if pipe_kind == "controlnet":
        from consistentid_amd.unet_spec import walk
        shapes, res = [], []
        c, hh = cfg.block_out_channels[0], h8
        shapes.append((c, hh))
        downs, _, _ = walk(cfg)
        for blk in downs:
            for r in blk.resnets:
                shapes.append((r.cout, hh))
            if blk.sampler:
                hh //= 2
                shapes.append((blk.resnets[-1].cout, hh))
        res = [(torch.randn(B, c_, h_, h_, generator=gen) * 0.1).half() for c_, h_ in shapes]
        mid = (torch.randn(B, shapes[-1][0], shapes[-1][1], shapes[-1][1], generator=gen) * 0.1).half()
        kw_o.update(down_residuals=[r.float() for r in res], mid_residual=mid.float())
        tok = lambda r: r.permute(0, 2, 3, 1).reshape(r.shape[0], -1, r.shape[1]).contiguous().to(dev)
        kw_h.update(down_block_res_samples=[tok(r) for r in res], mid_block_res_sample=tok(mid))
    ref = loop.denoise(oracle, osch(), f("latents") * s0.init_noise_sigma, f("null"), f("augmented"), f("text"),
                       unet_extra=extra.float(), **kw_o)
    no_extra_effect = loop.denoise(oracle, osch(), f("latents") * s0.init_noise_sigma, f("null"), f("augmented"), f("text"),
                                   unet_extra=torch.zeros_like(extra).float(), **kw_o)
    assert (ref - no_extra_effect).norm() / ref.norm() > 1e-3                 # the five extra channels matter
    h = lambda k: inp[k].to(dev).half()
    arm = loop.denoise(half_arm(oracle, dev), osch(), h("latents") * s0.init_noise_sigma, h("null"), h("augmented"), h("text"),
                       unet_extra=extra.to(dev), **dev_half(kw_o, dev))
    cls = (pipeline.StableDiffusionControlNetInpaintConsistentIDPipeline if pipe_kind == "controlnet"
           else pipeline.StableDiffusionInpaintConsistentIDPipeline)
    pipe = cls(hip, scheduler=(scheduler.EulerDiscreteScheduler() if euler else scheduler.DDIMScheduler()))
    pe = torch.cat([inp["null"], inp["augmented"], inp["text"]]).to(dev)
    with pytest.raises(ValueError):       # the oracle loop itself refuses the configuration the reference never executes
        loop.denoise(oracle, osch(), f("latents"), f("null"), f("augmented"), f("text"), unet_extra=extra.float(),
                     inpaint_mask=mask.float(), inpaint_init=init.float(), inpaint_noise=noise.float(), **kw_o)
    outs = []
    for given in (False, True):      # second generation: graph replay; image_latents / noise are ignored (no blend)
        opt = dict(image_latents=init.to(dev), noise=noise.to(dev)) if given else {}
        out = pipe(prompt_embeds=pe, latents=inp["latents"].to(dev), num_inference_steps=steps, guidance_scale=g,
                   start_merge_step=merge, output_type="latent", mask_latents=mask.to(dev),
                   masked_image_latents=masked.to(dev), **opt, **kw_h).images
        torch.cuda.synchronize()
        check_vs_fp16_arm(out, ref, arm, f"tiny 9-channel {pipe_kind} loop (euler={euler}, image_latents given={given})")
        outs.append(out.clone())
    assert torch.equal(outs[0], outs[1])
    keep = (1 - mask.float()).expand_as(ref).bool()
    assert not torch.allclose(ref[keep], init.float()[keep], atol=1e-2)   # a blend would pin the kept region to the init latents
    with pytest.raises(ValueError):
        pipe(prompt_embeds=pe, latents=inp["latents"].to(dev), num_inference_steps=steps, output_type="latent",
             mask_latents=mask.to(dev))                                                      # masked_image_latents missing


def test_tiny_inpaint_four_channel_unet_still_blends(dev):
    """The other side of ``if num_channels_unet == 4`` (inpaint ref :340-353): a 4-channel UNet given ``mask_latents``
    blends every step (kept region == image latents at the end, bit for bit), ignores ``masked_image_latents`` like the
    reference's 4-channel path does, and asks for image_latents / noise by name instead of dereferencing None."""
    from consistentid_amd import pipeline, synth
    from oracle import ddim, loop
    cfg, oracle, hip = _unet_pair("tiny", dev)
    B, steps, merge, g = 2, 4, 1, 7.5
    side = cfg.sample_size * 8
    h8 = side // 8
    inp = synth.random_inputs(cfg, B, side, side)
    gen = torch.Generator().manual_seed(43)
    init = torch.randn(B, 4, h8, h8, generator=gen).half()
    noise = torch.randn(B, 4, h8, h8, generator=gen).half()
    mask = (torch.rand(B, 1, h8, h8, generator=gen) > 0.5).half()
    f = lambda k: inp[k].float()
    kw_o = dict(num_inference_steps=steps, guidance_scale=g, start_merge_step=merge, inpaint_mask=mask.float(),
                inpaint_init=init.float(), inpaint_noise=noise.float())
    ref = loop.denoise(oracle, ddim.DDIMScheduler(), f("latents"), f("null"), f("augmented"), f("text"), **kw_o)
    h = lambda k: inp[k].to(dev).half()
    arm = loop.denoise(half_arm(oracle, dev), ddim.DDIMScheduler(), h("latents"), h("null"), h("augmented"), h("text"),
                       **dev_half(kw_o, dev))
    pipe = pipeline.StableDiffusionInpaintConsistentIDPipeline(hip)
    pe = torch.cat([inp["null"], inp["augmented"], inp["text"]]).to(dev)
    common = dict(prompt_embeds=pe, latents=inp["latents"].to(dev), num_inference_steps=steps, guidance_scale=g,
                  start_merge_step=merge, output_type="latent", mask_latents=mask.to(dev))
    out = pipe(image_latents=init.to(dev), noise=noise.to(dev), **common).images.clone()
    out2 = pipe(image_latents=init.to(dev), noise=noise.to(dev), masked_image_latents=init.to(dev), **common).images
    torch.cuda.synchronize()
    check_vs_fp16_arm(out, ref, arm, "tiny 4-channel inpaint loop (blend)")
    assert torch.equal(out, out2)                                              # masked_image_latents unused at 4 channels
    keep = (1 - mask).expand_as(out).bool().to(dev)
    assert torch.equal(out[keep], init.to(dev)[keep])
    with pytest.raises(ValueError, match="image_latents"):
        pipe(**common)


def test_tinyxl_raw_negative_prompt_embeds(dev):
    """ref SDXL :586-590: ``negative_prompt_embeds_text_only = cat([negative_prompt_embeds, uncond_prompt_tokens_faceid], 1)``
    is the unconditional set up to the merge step.  Raw [B, 77, Dc] negative embeds get the null set's trailing ID tokens
    appended by the pipeline; the result must equal the call with the assembled four sets, bit for bit."""
    from consistentid_amd import pipeline, synth
    cfg, oracle, hip = _unet_pair("tinyxl", dev)
    B, steps, merge, g = 2, 4, 1, 7.5
    side = cfg.sample_size * 8
    inp = synth.random_inputs(cfg, B, side, side)
    neg77 = torch.randn(B, 77, cfg.cross_attention_dim, generator=torch.Generator().manual_seed(5)).half()
    null_pre = torch.cat([neg77, inp["null"][:, -4:]], dim=1)        # what the reference assembles (:590)
    pipe = pipeline.ConsistentIDStableDiffusionXLPipeline(hip)
    common = dict(latents=inp["latents"].to(dev), num_inference_steps=steps, guidance_scale=g, start_merge_step=merge,
                  output_type="latent", pooled_prompt_embeds=inp["pooled_augmented"],
                  pooled_prompt_embeds_text_only=inp["pooled_text"], negative_pooled_prompt_embeds=inp["pooled_null"],
                  add_time_ids=inp["time_ids"])
    want = pipe(prompt_embeds=torch.cat([null_pre, inp["augmented"], inp["text"], inp["null"]]).to(dev), **common).images.clone()
    pe3 = torch.cat([inp["null"], inp["augmented"], inp["text"]]).to(dev)
    got = pipe(prompt_embeds=pe3, negative_prompt_embeds=neg77.to(dev), **common).images.clone()
    got81 = pipe(prompt_embeds=pe3, negative_prompt_embeds=null_pre.to(dev), **common).images.clone()
    plain = pipe(prompt_embeds=pe3, **common).images
    torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(got81, want)
    assert not torch.equal(plain, want)
    with pytest.raises(ValueError):
        pipe(prompt_embeds=pe3, negative_prompt_embeds=neg77[:, :50].to(dev), **common)
