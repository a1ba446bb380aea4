"""SURVEY.md section 8 row f-2 on the GPU: the VAE decoder engine against the CPU oracle, and the pipelines'
``output_type`` other than "latent"."""
import copy

import numpy as np
import pytest
import torch

from conftest import check_close, check_vs_fp16_arm, half_arm

pytestmark = pytest.mark.gpu


def _pair(dev, name):
    from consistentid_amd import synth, vae_spec
    from consistentid_amd.vae import HipVAEDecoder
    from oracle import vae as ovae
    cfg = vae_spec.tiny_vae_config() if name == "tiny" else vae_spec.sd_vae_config()
    ocfg = ovae.tiny_vae_config() if name == "tiny" else ovae.sd_vae_config()
    sd = synth.random_vae_state_dict(cfg, seed=5, device="cpu" if name == "tiny" else dev)
    oracle = ovae.AutoencoderKL(ocfg)
    oracle.load_state_dict({k: v.detach().cpu().float() for k, v in sd.items()}, strict=True)
    return cfg, oracle.eval(), HipVAEDecoder(cfg, sd, device=dev)


def test_softmax_rows(dev):
    from consistentid_amd import ops
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(37, 264, generator=g) * 4).half()
    ref = torch.softmax(x.double() * np.log(2.0), dim=-1)
    y = x.to(dev).clone()
    ops.softmax_rows(y, rows=37, cols=256, ld=264)     # the 8 pad columns of every row stay untouched
    torch.cuda.synchronize()
    check_close(y[:, :256], torch.softmax(x[:, :256].double() * np.log(2.0), dim=-1), "softmax_rows")
    assert torch.equal(y[:, 256:].cpu(), x[:, 256:])
    del ref


@pytest.mark.parametrize("side", [16, 24])
def test_tiny_vae_decode(dev, side):
    from oracle.vae import decode_latents
    cfg, oracle, hip = _pair(dev, "tiny")
    lat = (torch.randn(2, 4, side, side, generator=torch.Generator().manual_seed(7)) * 0.18215 * 4).half()
    with torch.no_grad():
        ref_raw = oracle.decode(lat.float() / cfg.scaling_factor)
        ref_img = decode_latents(oracle, lat.float())
    out_raw = hip.decode_tokens(lat.to(dev))
    out_img = hip.decode_latents(lat.to(dev))
    (dec,) = hip.decode(lat.to(dev) / cfg.scaling_factor, return_dict=False)
    torch.cuda.synchronize()
    assert out_raw.shape == ref_raw.shape == (2, 3, side * 2, side * 2)
    arm_m = half_arm(oracle, dev)
    with torch.no_grad():
        arm_raw = arm_m.decode(lat.to(dev) / cfg.scaling_factor)
        arm_img = decode_latents(arm_m, lat.to(dev))
    check_vs_fp16_arm(out_raw, ref_raw, arm_raw, "tiny VAE decode")
    check_vs_fp16_arm(dec, ref_raw, arm_raw, "tiny VAE decode (diffusers protocol)")
    check_vs_fp16_arm(out_img, ref_img, arm_img, "tiny VAE decode_latents")


def test_sd_vae_decode_full_size(dev):
    """the 49 M-parameter SD decoder (of the 83,653,863-parameter AutoencoderKL) on one 64x64 latent -> 512x512 image"""
    cfg, oracle, hip = _pair(dev, "sd")
    lat = (torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(8)) * 0.18215 * 4).half()
    with torch.no_grad():
        ref = oracle.decode(lat.float() / cfg.scaling_factor)
    out = hip.decode_tokens(lat.to(dev))
    torch.cuda.synchronize()
    assert out.shape == (1, 3, 512, 512)
    with torch.no_grad():
        arm = half_arm(oracle, dev).decode(lat.to(dev) / cfg.scaling_factor)
    check_vs_fp16_arm(out, ref, arm, "SD VAE decode 512x512")


def test_pipeline_pixel_outputs(dev):
    """ConsistentIDStableDiffusionPipeline.__call__ with output_type "np" / "pil" / (inpaint) "pt": denoise loop + VAE
    decode + the reference's post-processing (SD :581-598), against oracle loop -> oracle decode_latents."""
    from consistentid_amd import pipeline, synth
    from consistentid_amd.unet import HipUNet
    from oracle import ddim, loop
    from oracle.vae import decode_latents
    from oracle_utils import build_oracle, make_weights
    vcfg, o_vae, h_vae = _pair(dev, "tiny")
    cfg, sd, ad = make_weights("tiny", rank=8)
    o_unet = build_oracle("tiny", sd, ad, rank=8)
    h_unet = HipUNet(cfg, sd, ad, device=dev)
    B, steps, merge, g = 2, 3, 1, 5.0
    inp = synth.random_inputs(cfg, B, cfg.sample_size * 8, cfg.sample_size * 8)
    side = cfg.sample_size * 2 ** (len(vcfg.block_out_channels) - 1)      # the toy VAE has two levels: x2, not x8
    f = lambda k: inp[k].float()
    lat = loop.denoise(o_unet, ddim.DDIMScheduler(), f("latents"), f("null"), f("augmented"), f("text"),
                       num_inference_steps=steps, guidance_scale=g, start_merge_step=merge)
    with torch.no_grad():
        ref = decode_latents(o_vae, lat)                                   # [B, 3, H, W] in [0, 1]
        h = lambda k: inp[k].to(dev)
        alat = loop.denoise(half_arm(o_unet, dev), ddim.DDIMScheduler(), h("latents"), h("null"), h("augmented"), h("text"),
                            num_inference_steps=steps, guidance_scale=g, start_merge_step=merge)
        arm = decode_latents(half_arm(o_vae, dev), alat)
    pe = torch.cat([inp["null"], inp["augmented"], inp["text"]]).to(dev)
    kw = dict(prompt_embeds=pe, latents=inp["latents"].to(dev), num_inference_steps=steps, guidance_scale=g,
              start_merge_step=merge)
    pipe = pipeline.ConsistentIDStableDiffusionPipeline(h_unet, vae=h_vae)
    arr = pipe(output_type="np", **kw).images
    assert isinstance(arr, np.ndarray) and arr.dtype == np.float32 and arr.shape == (B, side, side, 3)
    check_vs_fp16_arm(torch.from_numpy(arr).permute(0, 3, 1, 2), ref, arm, "pipeline output_type=np")
    pil = pipe(output_type="pil", **kw).images
    assert len(pil) == B and pil[0].size == (side, side)
    got = torch.from_numpy(np.stack([np.asarray(p) for p in pil])).permute(0, 3, 1, 2).float() / 255
    assert (got - ref).abs().max() < 2e-2 + 1 / 255
    pt = pipeline.StableDiffusionInpaintConsistentIDPipeline(h_unet, vae=h_vae)(output_type="pt", **kw).images
    assert torch.is_tensor(pt) and pt.shape == (B, 3, side, side)
    check_vs_fp16_arm(pt, ref, arm, "pipeline output_type=pt")


# ----------------------------------------------------------------------------- fp32 path (SDXL VAE: force_upcast)
TOL32 = dict(tol_l2=5e-5, tol_max=5e-4)      # fp32 kernels against the fp32 oracle: summation order and exp2 / exp ulps only


@pytest.mark.parametrize("B,H,cin,cout,taps,up,res", [
    (2, 16, 64, 96, 9, 0, True), (1, 12, 32, 64, 9, 1, False), (2, 8, 4, 100, 9, 0, False),    # cin % 4 != ... 4 -> vector path, N ragged
    (1, 10, 6, 40, 9, 0, True),                                                               # scalar gather path (c % 4 != 0)
    (3, 7, 128, 3, 9, 0, False),                                                              # conv_out: N = 3
    (1, 33, 160, 72, 1, 0, True),                                                             # linear / 1x1, ragged M
])
def test_gemm_f32(dev, B, H, cin, cout, taps, up, res):
    from consistentid_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(B, cin, H, H, generator=g)
    k = 3 if taps == 9 else 1
    w = torch.randn(cout, cin, k, k, generator=g) * (cin * taps) ** -0.5
    b = torch.randn(cout, generator=g)
    xin = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = torch.nn.functional.conv2d(xin.double(), w.double(), b.double(), padding=1 if taps == 9 else 0)
    Ho = H << up
    r = torch.randn(B, cout, Ho, Ho, generator=g) if res else None
    if res:
        ref = ref + r.double()
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()
    out = torch.empty(B * Ho * Ho, cout, dtype=torch.float32, device=dev)
    ops.gemm_f32(tok(x).to(dev), w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().to(dev), out, M=B * Ho * Ho, N=cout,
                 c=cin, bias=b.to(dev), res=tok(r).to(dev) if res else None, taps=taps, Hi=H, Wi=H, up=up)
    torch.cuda.synchronize()
    check_close(out, tok(ref), f"gemm_f32 {cin}->{cout} taps={taps} up={up}", **TOL32)


def test_groupnorm_and_softmax_f32(dev):
    from consistentid_amd import ops
    g = torch.Generator().manual_seed(4)
    B, HW, C = 2, 300, 128
    x = torch.randn(B, HW, C, generator=g) * 3 + 1
    gm, bt = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(x.double().transpose(1, 2), 32, gm.double(), bt.double(), 1e-6)).transpose(1, 2)
    out = torch.empty(B * HW, C, dtype=torch.float32, device=dev)
    ws = torch.empty(ops.groupnorm_f32_ws_bytes(B, HW, C), dtype=torch.uint8, device=dev)
    ops.groupnorm_f32(x.reshape(-1, C).to(dev), out, gm.to(dev), bt.to(dev), ws, B=B, HW=HW, C_=C)
    s = torch.randn(70, 1000, generator=g) * 4
    sd = s.to(dev)
    ops.softmax_rows_f32(sd, rows=70, cols=1000, ld=1000)
    torch.cuda.synchronize()
    check_close(out.view(B, HW, C), ref, "groupnorm_f32 + SiLU", **TOL32)
    check_close(sd, torch.softmax(s.double() * np.log(2.0), -1), "softmax_rows_f32", **TOL32)


def _pair32(dev, name):
    from consistentid_amd import synth, vae_spec
    from consistentid_amd.vae import HipVAEDecoderF32, make_vae_decoder
    from oracle import vae as ovae
    tiny = vae_spec.VAEConfig(block_out_channels=(64, 128), layers_per_block=1, scaling_factor=0.13025, force_upcast=True)
    cfg = tiny if name == "tiny" else vae_spec.sdxl_vae_config()
    ocfg = ovae.tiny_vae_config() if name == "tiny" else ovae.sd_vae_config()
    sd = synth.random_vae_state_dict(cfg, seed=6, device="cpu" if name == "tiny" else dev)
    oracle = ovae.AutoencoderKL(ocfg)
    oracle.load_state_dict({k: v.detach().cpu().float() for k, v in sd.items()}, strict=True)
    hip = make_vae_decoder(cfg, sd, device=dev)
    assert isinstance(hip, HipVAEDecoderF32)
    return cfg, oracle.eval(), hip


def test_tiny_vae_decode_fp32(dev):
    """force_upcast VAE (the SDXL convention): fp32 decode (SDXL :670-676) against the fp32 oracle at fp32 tolerance"""
    cfg, oracle, hip = _pair32(dev, "tiny")
    lat = torch.randn(2, 4, 20, 20, generator=torch.Generator().manual_seed(7)) * cfg.scaling_factor * 4
    with torch.no_grad():
        ref = oracle.decode(lat / cfg.scaling_factor)
    out = hip.decode_tokens(lat.to(dev))
    (dec,) = hip.decode(lat.to(dev) / cfg.scaling_factor, return_dict=False)
    torch.cuda.synchronize()
    assert out.dtype == torch.float32 and out.shape == ref.shape
    check_close(out, ref, "tiny fp32 VAE decode", **TOL32)
    check_close(dec, ref, "tiny fp32 VAE decode (diffusers protocol)", **TOL32)
    check_close(hip.decode_latents(lat.to(dev)), (ref / 2 + 0.5).clamp(0, 1), "tiny fp32 VAE decode_latents", **TOL32)


def test_sdxl_vae_decode_full_size(dev):
    """BASELINE config 4's pixel stage: the 49 M-parameter decoder of the SDXL VAE in fp32, one 128 x 128 latent ->
    1024 x 1024 image (its mid-block attention runs over 16384 positions), against the fp32 CPU oracle."""
    cfg, oracle, hip = _pair32(dev, "sdxl")
    lat = torch.randn(1, 4, 128, 128, generator=torch.Generator().manual_seed(8)) * cfg.scaling_factor * 4
    # the fp32 oracle runs on the GPU here (stock PyTorch fp32 kernels; the host CPU needs minutes for the 16384 x 16384
    # attention and the 1024^2 convolutions); the tiny fp32 VAE above is checked against the CPU oracle
    oracle = oracle.to(dev)
    with torch.no_grad():
        ref = oracle.decode(lat.to(dev) / cfg.scaling_factor)
    out = hip.decode_tokens(lat.to(dev))
    torch.cuda.synchronize()
    assert out.shape == (1, 3, 1024, 1024)
    check_close(out, ref, "SDXL VAE fp32 decode 1024x1024", **TOL32)


def test_sdxl_pipeline_pixel_outputs(dev):
    """ConsistentIDStableDiffusionXLPipeline with a force_upcast VAE: fp16 denoise loop, latents upcast, fp32 decode
    (ref SDXL :670-684) -> "pt" / "np" images against oracle loop -> oracle fp32 decode."""
    from consistentid_amd import pipeline, synth
    from consistentid_amd.unet import HipUNet
    from oracle import ddim, loop
    from oracle.vae import decode_latents
    from oracle_utils import build_oracle, make_weights
    vcfg, o_vae, h_vae = _pair32(dev, "tiny")
    cfg, sd, ad = make_weights("tinyxl", rank=8)
    o_unet = build_oracle("tinyxl", sd, ad, rank=8)
    h_unet = HipUNet(cfg, sd, ad, device=dev)
    B, steps, merge, g = 2, 3, 1, 7.5
    inp = synth.random_inputs(cfg, B, cfg.sample_size * 8, cfg.sample_size * 8)
    f = lambda k: inp[k].float()
    okw = dict(add_text_embeds_null=f("pooled_null"), add_text_embeds_text=f("pooled_text"),
               add_text_embeds_aug=f("pooled_augmented"), add_time_ids=inp["time_ids"])
    lat = loop.denoise(o_unet, ddim.DDIMScheduler(), f("latents"), f("null"), f("augmented"), f("text"),
                       num_inference_steps=steps, guidance_scale=g, start_merge_step=merge, **okw)
    with torch.no_grad():
        ref = (o_vae.decode(lat / vcfg.scaling_factor) / 2 + 0.5).clamp(0, 1)
        h = lambda k: inp[k].to(dev)
        alat = loop.denoise(half_arm(o_unet, dev), ddim.DDIMScheduler(), h("latents"), h("null"), h("augmented"), h("text"),
                            num_inference_steps=steps, guidance_scale=g, start_merge_step=merge,
                            add_text_embeds_null=h("pooled_null"), add_text_embeds_text=h("pooled_text"),
                            add_text_embeds_aug=h("pooled_augmented"), add_time_ids=inp["time_ids"].to(dev))
        # the reference-precision arm decodes in fp32 too (that is what upcast_vae is for)
        arm = (copy.deepcopy(o_vae).to(dev).decode(alat.float() / vcfg.scaling_factor) / 2 + 0.5).clamp(0, 1)
    pipe = pipeline.ConsistentIDStableDiffusionXLPipeline(h_unet, vae=h_vae)
    kw = dict(prompt_embeds=torch.cat([inp["null"], inp["augmented"], inp["text"]]).to(dev), latents=inp["latents"].to(dev),
              num_inference_steps=steps, guidance_scale=g, start_merge_step=merge, pooled_prompt_embeds=inp["pooled_augmented"],
              pooled_prompt_embeds_text_only=inp["pooled_text"], negative_pooled_prompt_embeds=inp["pooled_null"],
              add_time_ids=inp["time_ids"])
    pt = pipe(output_type="pt", **kw).images
    assert torch.is_tensor(pt) and pt.dtype == torch.float32
    check_vs_fp16_arm(pt, ref, arm, "SDXL pipeline output_type=pt (fp32 VAE)")
    arr = pipe(output_type="np", **kw).images
    check_vs_fp16_arm(torch.from_numpy(arr).permute(0, 3, 1, 2), ref, arm, "SDXL pipeline output_type=np (fp32 VAE)")
