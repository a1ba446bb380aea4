"""SURVEY.md section 8 row f-2 on the GPU: the VAE decoder engine against the CPU oracle, and the pipelines'
``output_type`` other than "latent"."""
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
