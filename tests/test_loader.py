"""Base-model loader (consistentid_amd/loader.py): diffusers config.json -> engine configuration, component folders
-> state dicts.  File IO and host logic only; the engines themselves are exercised by the GPU tests."""
import json

import pytest
import torch

from consistentid_amd import loader, synth, unet_spec, vae_spec

SD15_JSON = {  # the published runwayml/stable-diffusion-v1-5 unet/config.json fields that matter
    "_class_name": "UNet2DConditionModel", "act_fn": "silu", "attention_head_dim": 8,
    "block_out_channels": [320, 640, 1280, 1280], "center_input_sample": False, "cross_attention_dim": 768,
    "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    "downsample_padding": 1, "flip_sin_to_cos": True, "freq_shift": 0, "in_channels": 4, "layers_per_block": 2,
    "mid_block_scale_factor": 1, "norm_eps": 1e-05, "norm_num_groups": 32, "out_channels": 4, "sample_size": 64,
    "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"]}
SDXL_JSON = {
    "_class_name": "UNet2DConditionModel", "act_fn": "silu", "addition_embed_type": "text_time",
    "addition_time_embed_dim": 256, "attention_head_dim": [5, 10, 20], "block_out_channels": [320, 640, 1280],
    "cross_attention_dim": 2048, "down_block_types": ["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
    "in_channels": 4, "layers_per_block": 2, "norm_eps": 1e-05, "norm_num_groups": 32, "out_channels": 4,
    "projection_class_embeddings_input_dim": 2816, "sample_size": 128, "transformer_layers_per_block": [1, 2, 10],
    "up_block_types": ["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"], "use_linear_projection": True,
    "upcast_attention": None}


def test_published_configs_map_to_the_builtin_topologies():
    assert loader.unet_config_from_diffusers(SD15_JSON) == unet_spec.sd15_config()
    assert loader.unet_config_from_diffusers(SDXL_JSON) == unet_spec.sdxl_config()
    cn = dict(SD15_JSON, _class_name="ControlNetModel")
    del cn["up_block_types"], cn["out_channels"]
    assert loader.unet_config_from_diffusers(cn).block_out_channels == (320, 640, 1280, 1280)
    with pytest.raises(NotImplementedError):
        loader.unet_config_from_diffusers(dict(SD15_JSON, class_embed_type="timestep"))
    with pytest.raises(NotImplementedError):
        loader.unet_config_from_diffusers(dict(SD15_JSON, down_block_types=["AttnDownBlock2D"] * 4))


def _write_component(folder, cfg_json, sd, fmt):
    folder.mkdir(parents=True)
    (folder / "config.json").write_text(json.dumps(cfg_json))
    if fmt == "safetensors":
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in sd.items()}, str(folder / "diffusion_pytorch_model.safetensors"))
    else:
        torch.save(sd, folder / "diffusion_pytorch_model.bin")


@pytest.mark.parametrize("fmt", ["safetensors", "bin"])
def test_read_component_round_trip(tmp_path, fmt):
    cfg = unet_spec.tiny_config("sd15")
    sd = synth.random_unet_state_dict(cfg, seed=0)
    tiny_json = {"block_out_channels": list(cfg.block_out_channels), "down_block_types": list(cfg.down_block_types),
                 "up_block_types": list(cfg.up_block_types), "layers_per_block": 1, "attention_head_dim": 2,
                 "cross_attention_dim": 128, "sample_size": 32}
    _write_component(tmp_path / "model" / "unet", tiny_json, sd, fmt)
    got_cfg, got_sd = loader.read_component(tmp_path / "model" / "unet")
    assert loader.unet_config_from_diffusers(got_cfg) == cfg
    assert got_sd.keys() == sd.keys() and all(torch.equal(got_sd[k], sd[k]) for k in sd)
    with pytest.raises(FileNotFoundError):
        loader.read_component(tmp_path / "model" / "vae")
    vcfg = vae_spec.tiny_vae_config()
    _write_component(tmp_path / "model" / "vae", {"block_out_channels": list(vcfg.block_out_channels), "layers_per_block": 1,
                                                 "scaling_factor": 0.18215}, synth.random_vae_state_dict(vcfg), fmt)
    assert loader.vae_config_from_diffusers(loader.read_component(tmp_path / "model" / "vae")[0]) == vcfg


@pytest.mark.gpu
def test_from_pretrained_then_checkpoint_matches_direct_construction(tmp_path, dev):
    """The reference's order -- from_pretrained(base model dir), then load_ConsistentID_model(checkpoint) -- gives the
    same engine as packing the same weights directly."""
    from consistentid_amd import pipeline
    from consistentid_amd.unet import HipUNet
    cfg = unet_spec.tiny_config("sd15")
    sd = synth.random_unet_state_dict(cfg, seed=0)
    ad = synth.random_adapter_state_dict(cfg, sd, rank=8, seed=1)
    tiny_json = {"block_out_channels": list(cfg.block_out_channels), "down_block_types": list(cfg.down_block_types),
                 "up_block_types": list(cfg.up_block_types), "layers_per_block": 1, "attention_head_dim": 2,
                 "cross_attention_dim": 128, "sample_size": 32}
    _write_component(tmp_path / "base" / "unet", tiny_json, sd, "safetensors")
    vcfg = vae_spec.tiny_vae_config()
    _write_component(tmp_path / "base" / "vae", {"block_out_channels": list(vcfg.block_out_channels), "layers_per_block": 1},
                     synth.random_vae_state_dict(vcfg), "safetensors")
    # the reference scripts' own keyword sets (infer.py:17-21; infer_SDXL.py; demo/controlnet_demo.py) and their .to(device)
    pipe = pipeline.ConsistentIDStableDiffusionPipeline.from_pretrained(
        str(tmp_path / "base"), torch_dtype=torch.float16, variant="fp16", safety_checker=None, use_safetensors=True,
        device=dev).to(dev)
    with pytest.raises(TypeError):
        pipeline.ConsistentIDStableDiffusionPipeline.from_pretrained(str(tmp_path / "base"), no_such_option=1, device=dev)
    with pytest.raises(ValueError):
        pipe.to("cpu")
    assert pipe.vae is not None and pipe.unet.config == cfg
    pipe.load_ConsistentID_model({"adapter_modules": ad}, lora_rank=8)
    direct = pipeline.ConsistentIDStableDiffusionPipeline(HipUNet(cfg, sd, ad, device=dev))
    inp = synth.random_inputs(cfg, 2, 256, 256)
    kw = dict(prompt_embeds=torch.cat([inp["null"], inp["augmented"], inp["text"]]).to(dev), latents=inp["latents"].to(dev),
              num_inference_steps=3, guidance_scale=5.0, start_merge_step=1, output_type="latent")
    a, b = pipe(**kw).images, direct(**kw).images
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    img = pipe(**dict(kw, output_type="np")).images       # the VAE decoder of the directory is wired in
    assert img.shape[0] == 2 and img.shape[-1] == 3 and bool((img >= 0).all()) and bool((img <= 1).all())


def test_read_scheduler_dispatch_and_fallback(tmp_path):
    """scheduler/scheduler_config.json: the base model's sampler class decides the engine scheduler (SDXL base ships
    EulerDiscrete, SD1.5 PNDM -> DDIM on its config with a warning); a config the tables do not implement (a saved
    DDIM default carries clip_sample=true) falls back to the defaults instead of failing from_pretrained."""
    import warnings
    from consistentid_amd import scheduler
    d = tmp_path / "m" / "scheduler"
    d.mkdir(parents=True)
    base = {"beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear", "num_train_timesteps": 1000,
            "steps_offset": 1, "timestep_spacing": "leading"}
    (d / "scheduler_config.json").write_text(json.dumps(dict(base, _class_name="EulerDiscreteScheduler")))
    assert isinstance(loader.read_scheduler(tmp_path / "m"), scheduler.EulerDiscreteScheduler)
    (d / "scheduler_config.json").write_text(json.dumps(dict(base, _class_name="DDIMScheduler", timestep_spacing="trailing")))
    sch = loader.read_scheduler(tmp_path / "m")
    assert isinstance(sch, scheduler.DDIMScheduler) and sch.timestep_spacing == "trailing"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        (d / "scheduler_config.json").write_text(json.dumps(dict(base, _class_name="PNDMScheduler", skip_prk_steps=True)))
        assert isinstance(loader.read_scheduler(tmp_path / "m"), scheduler.DDIMScheduler)
        # an unsupported sampling MODIFIER is dropped, the rest of the saved config survives (non-default betas / spacing)
        odd = dict(base, beta_start=0.001, beta_end=0.02, timestep_spacing="trailing")
        (d / "scheduler_config.json").write_text(json.dumps(dict(odd, _class_name="DDIMScheduler", clip_sample=True)))
        sch = loader.read_scheduler(tmp_path / "m")
        assert isinstance(sch, scheduler.DDIMScheduler) and sch.timestep_spacing == "trailing"
        want = scheduler.DDIMScheduler.from_config(odd)
        sch.set_timesteps(10); want.set_timesteps(10)
        assert (sch.coefficient_table(False) == want.coefficient_table(False)).all()
        dflt = scheduler.DDIMScheduler(); dflt.set_timesteps(10)
        assert not (sch.coefficient_table(False) == dflt.coefficient_table(False)).all()
        # a foreign SCHEDULE (v-prediction) cannot be kept: defaults, with a warning
        (d / "scheduler_config.json").write_text(json.dumps(dict(base, _class_name="DDIMScheduler", prediction_type="v_prediction")))
        assert isinstance(loader.read_scheduler(tmp_path / "m"), scheduler.DDIMScheduler)
    assert len(w) == 3
    # flags that change the noise SCHEDULE itself are an error, not a warning: the samples would silently differ
    for flag in (dict(use_karras_sigmas=True), dict(rescale_betas_zero_snr=True), dict(interpolation_type="log_linear")):
        (d / "scheduler_config.json").write_text(json.dumps(dict(base, _class_name="DDIMScheduler", **flag)))
        with pytest.raises(NotImplementedError, match="noise schedule"):
            loader.read_scheduler(tmp_path / "m")
    # ... while thresholding, like clip_sample, is dropped with a warning
    with warnings.catch_warnings(record=True) as w2:
        warnings.simplefilter("always")
        (d / "scheduler_config.json").write_text(json.dumps(dict(base, _class_name="DDIMScheduler", thresholding=True)))
        assert isinstance(loader.read_scheduler(tmp_path / "m"), scheduler.DDIMScheduler)
    assert len(w2) == 1
    assert loader.read_scheduler(tmp_path / "nowhere") is None


def test_strength_window_rejects_empty_loops():
    """int(S * strength) == 0 leaves no denoising step: diffusers raises, so does the engine (no zero-step generation)"""
    from consistentid_amd import pipeline

    class _U:      # the method reads nothing but the argument values
        config = type("C", (), {"in_channels": 4})()
    p = pipeline.StableDiffusionInpaintConsistentIDPipeline.__new__(pipeline.StableDiffusionInpaintConsistentIDPipeline)
    p.unet = _U()
    with pytest.raises(ValueError):
        p._strength_window(0.5, 1, torch.zeros(1, 4, 8, 8), None, None)
    with pytest.raises(ValueError):
        p._strength_window(0.0, 10, torch.zeros(1, 4, 8, 8), None, None)
    first, lat, scaled = p._strength_window(0.6, 10, torch.zeros(1, 4, 8, 8), None, None)
    assert first == 4 and scaled
