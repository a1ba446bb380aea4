"""Self-made pins for the parts of the oracle that restate diffusers (parity unpinned by the
reference itself): structural invariants and closed forms."""
import math

import numpy as np
import torch

from consistentid_amd import scheduler as psched
from consistentid_amd import unet_spec
from oracle import ddim, loop
from oracle import processors as oproc
from oracle import unet as ounet
from oracle_utils import build_oracle, make_weights


def _meta_unet(cfg):
    with torch.device("meta"):
        m = ounet.UNet2DConditionModel(cfg)
        oproc.set_ip_adapter(m, lora_rank=128)
    return m


def test_param_counts_match_published_unets():
    # SD1.5 UNet: 859,520,964 parameters; SDXL-base UNet: 2,567,463,684
    for ocfg, pcfg, want in ((ounet.sd15_config(), unet_spec.sd15_config(), 859_520_964),
                             (ounet.sdxl_config(), unet_spec.sdxl_config(), 2_567_463_684)):
        m = _meta_unet(ocfg)
        n = sum(p.numel() for k, p in m.state_dict().items() if ".processor." not in k)
        assert n == want
        assert unet_spec.count_params(unet_spec.unet_param_shapes(pcfg)) == want


def test_processor_enumeration_and_adapter_keys():
    for ocfg, pcfg, nproc in ((ounet.sd15_config(), unet_spec.sd15_config(), 32),
                              (ounet.sdxl_config(), unet_spec.sdxl_config(), 140)):
        m = _meta_unet(ocfg)
        names = list(m.attn_processors.keys())
        assert len(names) == nproc
        assert names == unet_spec.attn_processor_names(pcfg)
        # down -> up -> mid ; attn1/attn2 alternate (attn2 = odd index)
        assert names[0].startswith("down_blocks") and names[-1].startswith("mid_block")
        assert all(n.endswith("attn1.processor") for n in names[0::2])
        assert all(n.endswith("attn2.processor") for n in names[1::2])
        ad = {k: tuple(v.shape) for k, v in oproc.adapter_modules(m).state_dict().items()}
        assert ad == dict(unet_spec.adapter_param_shapes(pcfg, 128))
        sd = {k: tuple(v.shape) for k, v in m.state_dict().items() if ".processor." not in k}
        assert sd == dict(unet_spec.unet_param_shapes(pcfg))


def test_two_range_softmax_identity_fp64():
    """softmax(QKt^T)Vt + s*softmax(QKi^T)Vi == one score tile over concatenated keys with
    per-range normalisation (what cid_id_xattn_f16 computes)."""
    g = torch.Generator().manual_seed(0)
    q = torch.randn(5, 16, dtype=torch.float64, generator=g)
    kt, vt = torch.randn(77, 16, dtype=torch.float64, generator=g), torch.randn(77, 16, dtype=torch.float64, generator=g)
    ki, vi = torch.randn(4, 16, dtype=torch.float64, generator=g), torch.randn(4, 16, dtype=torch.float64, generator=g)
    s = 0.7
    ref = (q @ kt.T).softmax(-1) @ vt + s * ((q @ ki.T).softmax(-1) @ vi)
    S = q @ torch.cat([kt, ki]).T
    P = torch.cat([S[:, :77].softmax(-1), s * S[:, 77:].softmax(-1)], -1)
    assert torch.allclose(P @ torch.cat([vt, vi]), ref, atol=1e-13)


def test_lora_merge_identity_fp64():
    g = torch.Generator().manual_seed(1)
    W = torch.randn(32, 24, dtype=torch.float64, generator=g)
    up, down = torch.randn(32, 4, dtype=torch.float64, generator=g), torch.randn(4, 24, dtype=torch.float64, generator=g)
    x = torch.randn(7, 24, dtype=torch.float64, generator=g)
    assert torch.allclose(x @ W.T + 0.5 * (x @ down.T) @ up.T, x @ (W + 0.5 * up @ down).T, atol=1e-12)


def test_ddim_closed_form_and_product_scheduler():
    o = ddim.DDIMScheduler()
    o.set_timesteps(50)
    p = psched.DDIMScheduler()
    p.set_timesteps(50)
    assert o.timesteps.tolist() == p.timesteps.tolist() == list(range(981, 0, -20))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 8, 8, dtype=torch.float64, generator=g)
    e = torch.randn(2, 4, 8, 8, dtype=torch.float64, generator=g)
    for t in (981, 501, 1):
        sa, s1a, sp, s1p = o.coefficients(t)
        want = sp * (x - s1a * e) / sa + s1p * e
        assert torch.allclose(o.step(e, t, x), want)
        cx, ce = p.step_coefficients(t)
        assert torch.allclose(cx * x + ce * e, want, rtol=1e-5, atol=1e-5)
    # last step lands on alphas_cumprod[0] (set_alpha_to_one False)
    assert abs(o.coefficients(1)[2] ** 2 - float(o.alphas_cumprod[0])) < 1e-7
    tab = p.coefficient_table(inpaint=True)
    assert tab.shape == (50, 5) and tab[-1, 2] == 1.0 and tab[-1, 3] == 0.0 and (tab[:, 4] == 1.0).all()
    a = float(o.alphas_cumprod[961])
    assert abs(tab[0, 2] - a ** 0.5) < 1e-6 and abs(tab[0, 3] - (1 - a) ** 0.5) < 1e-6


def test_timestep_embedding_layout():
    e = ounet.timestep_embedding(torch.tensor([3.0]), 8)
    f = torch.exp(-math.log(10000.0) * torch.arange(4) / 4)
    assert torch.allclose(e[0, :4], torch.cos(3.0 * f)) and torch.allclose(e[0, 4:], torch.sin(3.0 * f))


def test_tiny_unet_and_loop_run_on_cpu():
    for name in ("tiny", "tinyxl"):
        cfg, sd, ad = make_weights(name)
        m = build_oracle(name, sd, ad)
        from consistentid_amd import synth
        inp = synth.random_inputs(cfg, 1, cfg.sample_size * 8, cfg.sample_size * 8)
        kw = {}
        if name == "tinyxl":
            kw = dict(add_text_embeds_null=inp["pooled_null"].float(), add_text_embeds_text=inp["pooled_text"].float(),
                      add_text_embeds_aug=inp["pooled_augmented"].float(), add_time_ids=inp["time_ids"])
        sch = ddim.DDIMScheduler()
        out = loop.denoise(m, sch, inp["latents"].float(), inp["null"].float(), inp["augmented"].float(),
                           inp["text"].float(), num_inference_steps=3, guidance_scale=5.0, start_merge_step=0, **kw)
        assert out.shape == inp["latents"].shape and torch.isfinite(out).all()


def test_controlnet_inventory():
    """diffusers ControlNetModel for the SD1.5 config.  Published size of the SD1.5 ControlNets
    (lllyasviel/sd-controlnet-*, control_v11p_sd15_*): 361,279,120 parameters; the product's shape inventory loads
    into the oracle module tree with strict=True (same names, same shapes)."""
    import torch
    from consistentid_amd import unet_spec
    from oracle import unet as ounet
    from oracle.controlnet import ControlNetModel
    shapes = unet_spec.controlnet_param_shapes(unet_spec.sd15_config())
    assert unet_spec.count_params(shapes) == 361_279_120
    assert len(unet_spec.controlnet_zero_conv_channels(unet_spec.sd15_config())) == 12
    with torch.device("meta"):
        m = ControlNetModel(ounet.sd15_config())
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == dict(shapes)


def test_vae_inventory():
    """diffusers AutoencoderKL for the SD configuration has the published 83,653,863 parameters
    (decoder 49,490,179); the product's shape inventory loads into the oracle module tree name for name."""
    import torch
    from consistentid_amd import unet_spec, vae_spec
    from oracle import vae as ovae
    shapes = vae_spec.vae_param_shapes(vae_spec.sd_vae_config())
    assert unet_spec.count_params(shapes) == 83_653_863
    with torch.device("meta"):
        m = ovae.AutoencoderKL(ovae.sd_vae_config())
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == dict(shapes)
    assert sum(p.numel() for p in m.decoder.parameters()) == 49_490_179
