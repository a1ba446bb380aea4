"""SURVEY.md section 8 row f-1 on the GPU: the ControlNet encoder (condition embedding, encoder + mid with default
attention, zero convs) against the CPU oracle, and the ControlNet -> UNet hand-over through the reference's call
protocol (pipelines/StableDIffusionControlNetInpaint_ConsistentID.py:405-425)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import check_close, check_vs_fp16_arm, half_arm
from oracle_utils import build_oracle, make_weights, product_cfg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cin,cout,stride,silu,side", [(3, 16, 1, True, 40), (16, 32, 2, True, 33), (96, 96, 1, True, 16),
                                                       (96, 256, 2, True, 16), (256, 64, 1, False, 8)])
def test_conv3x3_small(dev, cin, cout, stride, silu, side):
    from consistentid_amd import ops
    g = torch.Generator().manual_seed(cin * 7 + cout)
    B = 2
    x = (torch.randn(B, cin, side, side + 3, generator=g)).half()
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).half()
    b = (torch.randn(cout, generator=g) * 0.1).half()
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1)
    if silu:
        ref = F.silu(ref)
    Ho, Wo = ref.shape[-2:]
    xt = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().to(dev)
    wt = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(dev)
    out = torch.empty(B * Ho * Wo, cout, dtype=torch.float16, device=dev)
    ops.conv3x3_small(xt, out, wt, b.to(dev), B=B, Hi=side, Wi=side + 3, cin=cin, cout=cout, stride=stride, silu=silu)
    torch.cuda.synchronize()
    got = out.view(B, Ho, Wo, cout).permute(0, 3, 1, 2)
    check_close(got, ref, f"conv3x3_small {cin}->{cout} s{stride}")


def _controlnet_pair(dev, seed=3):
    from consistentid_amd import synth
    from consistentid_amd.controlnet import HipControlNet
    from oracle import unet as ounet
    from oracle.controlnet import ControlNetModel
    cfg = product_cfg("tiny")
    sd = synth.random_controlnet_state_dict(cfg, seed=seed)
    oracle = ControlNetModel(ounet.tiny_config("sd15"))
    oracle.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    return cfg, oracle.eval(), HipControlNet(cfg, sd, device=dev)


@pytest.mark.parametrize("scale", [1.0, 0.6])
def test_tiny_controlnet_forward(dev, scale):
    """13 residuals of the tiny ControlNet vs the oracle; all 81 context tokens are plain keys (default attention)."""
    from consistentid_amd import synth
    cfg, oracle, hip = _controlnet_pair(dev)
    B = 2
    side = cfg.sample_size * 8
    inp = synth.random_inputs(cfg, B, side, side)
    g = torch.Generator().manual_seed(11)
    img = torch.rand(B, 3, side, side, generator=g).half()
    with torch.no_grad():
        rd, rm = oracle(inp["latents"].float(), 481, inp["text"].float(), img.float(), conditioning_scale=scale)
    hd, hm = hip(inp["latents"].to(dev), 481, encoder_hidden_states=inp["text"].to(dev), controlnet_cond=img.to(dev),
                 conditioning_scale=scale, return_dict=False)
    torch.cuda.synchronize()
    assert len(hd) == len(rd) == 6
    with torch.no_grad():
        ad_, am_ = half_arm(oracle, dev)(inp["latents"].to(dev), 481, inp["text"].to(dev), img.to(dev), conditioning_scale=scale)
    for i, (h, r, a) in enumerate(zip(hd + [hm], rd + [rm], ad_ + [am_])):
        assert h.shape == r.shape, (i, h.shape, r.shape)
        check_vs_fp16_arm(h, r, a, f"tiny ControlNet residual {i} (scale {scale})")
    # the condition embedding is cached per control image: a second call must not change the answer
    hd2, hm2 = hip(inp["latents"].to(dev), 481, encoder_hidden_states=inp["text"].to(dev), controlnet_cond=img.to(dev),
                   conditioning_scale=scale, return_dict=False)
    torch.cuda.synchronize()
    assert torch.equal(hm, hm2) and all(torch.equal(a, b) for a, b in zip(hd, hd2))


def test_controlnet_into_unet(dev):
    """The reference's per-step sequence (CN :389-425): ControlNet on the B conditional latents with the conditional
    embeds, then the 2B-batch UNet with the B-row residuals broadcast over the CFG halves."""
    from consistentid_amd import synth
    from consistentid_amd.unet import HipUNet
    cfg, o_cn, h_cn = _controlnet_pair(dev)
    _, sd, ad = make_weights("tiny", rank=8)
    o_unet = build_oracle("tiny", sd, ad, rank=8)
    h_unet = HipUNet(cfg, sd, ad, device=dev)
    B = 2
    side = cfg.sample_size * 8
    inp = synth.random_inputs(cfg, B, side, side)
    img = torch.rand(B, 3, side, side, generator=torch.Generator().manual_seed(12)).half()
    ehs = torch.cat([inp["null"], inp["augmented"]])
    lat2 = torch.cat([inp["latents"]] * 2)
    with torch.no_grad():
        rd, rm = o_cn(inp["latents"].float(), 301, inp["augmented"].float(), img.float())
        ref = o_unet(lat2.float(), 301, ehs.float(), down_block_additional_residuals=[torch.cat([d, d]) for d in rd],
                     mid_block_additional_residual=torch.cat([rm, rm])).sample
    hd, hm = h_cn(inp["latents"].to(dev), 301, encoder_hidden_states=inp["augmented"].to(dev),
                  controlnet_cond=img.to(dev), conditioning_scale=1.0, return_dict=False)
    out = h_unet(lat2.to(dev), 301, encoder_hidden_states=ehs.to(dev), cross_attention_kwargs={},
                 down_block_additional_residuals=hd, mid_block_additional_residual=hm).sample
    torch.cuda.synchronize()
    with torch.no_grad():
        ad_, am_ = half_arm(o_cn, dev)(inp["latents"].to(dev), 301, inp["augmented"].to(dev), img.to(dev))
        arm = half_arm(o_unet, dev)(lat2.to(dev), 301, ehs.to(dev), down_block_additional_residuals=[torch.cat([d, d]) for d in ad_],
                                    mid_block_additional_residual=torch.cat([am_, am_])).sample
    check_vs_fp16_arm(out, ref, arm, "tiny ControlNet -> UNet")


@pytest.mark.parametrize("use_graph", [False, True])
def test_controlnet_inpaint_loop(dev, use_graph):
    """StableDiffusionControlNetInpaintConsistentIDPipeline.__call__ with a native ControlNet vs the oracle loop:
    4 DDIM steps, embed switch after step 1, inpaint blend, conditioning scale 0.5 (the reference default) and a
    guidance window that switches the ControlNet off for the last step (CN :364-371)."""
    from consistentid_amd import pipeline, synth
    from consistentid_amd.unet import HipUNet
    from oracle import ddim, loop
    cfg, o_cn, h_cn = _controlnet_pair(dev)
    _, sd, ad = make_weights("tiny", rank=8)
    o_unet = build_oracle("tiny", sd, ad, rank=8)
    h_unet = HipUNet(cfg, sd, ad, device=dev)
    B, steps, merge, g = 2, 4, 1, 5.0
    side = cfg.sample_size * 8
    inp = synth.random_inputs(cfg, B, side, side)
    gen = torch.Generator().manual_seed(21)
    img = torch.rand(B, 3, side, side, generator=gen).half()
    init = torch.randn(B, 4, side // 8, side // 8, generator=gen).half()
    noise = torch.randn(B, 4, side // 8, side // 8, generator=gen).half()
    mask = (torch.rand(B, 1, side // 8, side // 8, generator=gen) > 0.5).half()
    f = lambda k: inp[k].float()
    ref = loop.denoise(o_unet, ddim.DDIMScheduler(), f("latents"), f("null"), f("augmented"), f("text"),
                       num_inference_steps=steps, guidance_scale=g, start_merge_step=merge,
                       inpaint_mask=mask.float(), inpaint_init=init.float(), inpaint_noise=noise.float(),
                       controlnet=o_cn, control_image=img.float(), conditioning_scale=0.5,
                       control_guidance_start=0.0, control_guidance_end=0.75)
    h = lambda k: inp[k].to(dev)
    arm = loop.denoise(half_arm(o_unet, dev), ddim.DDIMScheduler(), h("latents"), h("null"), h("augmented"), h("text"),
                       num_inference_steps=steps, guidance_scale=g, start_merge_step=merge,
                       inpaint_mask=mask.to(dev), inpaint_init=init.to(dev), inpaint_noise=noise.to(dev),
                       controlnet=half_arm(o_cn, dev), control_image=img.to(dev), conditioning_scale=0.5,
                       control_guidance_start=0.0, control_guidance_end=0.75)
    pipe = pipeline.StableDiffusionControlNetInpaintConsistentIDPipeline(h_unet, controlnet=h_cn, use_graph=use_graph)
    pe = torch.cat([inp["null"], inp["augmented"], inp["text"]]).to(dev)
    for _ in range(2):     # second generation replays the captured graphs
        out = pipe(prompt_embeds=pe, latents=inp["latents"].to(dev), control_image=img.to(dev), num_inference_steps=steps,
                   guidance_scale=g, start_merge_step=merge, output_type="latent", image_latents=init.to(dev),
                   noise=noise.to(dev), mask_latents=mask.to(dev), control_guidance_end=0.75).images
        torch.cuda.synchronize()
        check_vs_fp16_arm(out, ref, arm, f"tiny ControlNet-inpaint loop (graph={use_graph})")


def test_sd15_controlnet_forward_full_size(dev):
    """Config (5) of BASELINE.json at ControlNet granularity: the 361 M-parameter SD1.5 ControlNet, 512x512 control
    image, B=1 -- all 13 residuals against the fp32 CPU oracle."""
    from consistentid_amd import synth
    from consistentid_amd.controlnet import HipControlNet
    from oracle import unet as ounet
    from oracle.controlnet import ControlNetModel
    cfg = product_cfg("sd15")
    sd = synth.random_controlnet_state_dict(cfg, seed=4, device=dev)
    hip = HipControlNet(cfg, sd, device=dev)
    oracle = ControlNetModel(ounet.sd15_config())
    oracle.load_state_dict({k: v.detach().cpu().float() for k, v in sd.items()}, strict=True)
    del sd
    inp = synth.random_inputs(cfg, 1, 512, 512)
    img = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(13)).half()
    with torch.no_grad():
        rd, rm = oracle.eval()(inp["latents"].float(), 621, inp["augmented"].float(), img.float(), conditioning_scale=1.0)
    hd, hm = hip(inp["latents"].to(dev), 621, encoder_hidden_states=inp["augmented"].to(dev), controlnet_cond=img.to(dev),
                 conditioning_scale=1.0, return_dict=False)
    torch.cuda.synchronize()
    assert len(hd) == 12
    with torch.no_grad():
        ad_, am_ = half_arm(oracle, dev)(inp["latents"].to(dev), 621, inp["augmented"].to(dev), img.to(dev), conditioning_scale=1.0)
    for i, (h, r, a) in enumerate(zip(hd + [hm], rd + [rm], ad_ + [am_])):
        check_vs_fp16_arm(h, r, a, f"SD1.5 ControlNet residual {i}")
