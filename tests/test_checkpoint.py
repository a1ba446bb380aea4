"""ConsistentID checkpoint layouts (evaluation/convert_weights.py:14-25 writer,
pipline_StableDiffusion_ConsistentID.py:111-144 reader).  CPU part: file formats; GPU part: adapters merged into a live
engine equal an engine built with them."""
import pytest
import torch

from consistentid_amd import checkpoint, synth, unet_spec


def _adapters():
    cfg = unet_spec.tiny_config()
    sd = synth.random_unet_state_dict(cfg)
    return cfg, sd, synth.random_adapter_state_dict(cfg, sd, rank=8)


def test_training_checkpoint_conversion_and_files(tmp_path):
    cfg, sd, ad = _adapters()
    flat = {f"unet.{k}": v for k, v in list(sd.items())[:3]}            # frozen UNet entries are dropped
    flat.update({f"adapter_modules.{k}": v for k, v in ad.items()})
    flat.update({"image_proj_model.proj.weight": torch.ones(2, 2), "FacialEncoder.mlp.fc.bias": torch.zeros(3)})
    conv = checkpoint.convert_training_checkpoint(flat)
    assert set(conv) == {"image_proj", "adapter_modules", "FacialEncoder"}
    assert set(conv["adapter_modules"]) == set(unet_spec.adapter_param_shapes(cfg, rank=8))
    assert list(conv["image_proj"]) == ["proj.weight"] and list(conv["FacialEncoder"]) == ["mlp.fc.bias"]
    # the released .bin: the converter names the projector "image_proj_model", the loader wants "image_proj"
    released = {"image_proj_model": conv["image_proj"], "adapter_modules": conv["adapter_modules"],
                "FacialEncoder": conv["FacialEncoder"]}
    torch.save(released, tmp_path / "ConsistentID-v1.bin")
    for src in (tmp_path / "ConsistentID-v1.bin", released, flat):
        got = checkpoint.load_checkpoint(src)
        assert set(got) == {"image_proj", "adapter_modules", "FacialEncoder"}
        assert all(torch.equal(got["adapter_modules"][k], ad[k]) for k in ad)
    got = checkpoint.load_checkpoint(tmp_path, weight_name="ConsistentID-v1.bin")      # directory + weight_name
    assert torch.equal(got["image_proj"]["proj.weight"], torch.ones(2, 2))
    # flat safetensors file
    from safetensors.torch import save_file
    save_file({f"{part}.{k}": v.contiguous() for part, d in conv.items() for k, v in d.items()},
              str(tmp_path / "ckpt.safetensors"))
    got = checkpoint.load_checkpoint(tmp_path / "ckpt.safetensors")
    assert all(torch.equal(got["adapter_modules"][k], ad[k]) for k in ad)
    with pytest.raises(KeyError):
        checkpoint.load_checkpoint({"image_proj": {}})
    with pytest.raises(FileNotFoundError):
        checkpoint.load_checkpoint(tmp_path / "missing.bin")


@pytest.mark.gpu
def test_adapters_loaded_into_live_engine(dev, tmp_path):
    from consistentid_amd import pipeline
    from consistentid_amd.unet import HipUNet
    cfg, sd, ad = _adapters()
    direct = HipUNet(cfg, sd, ad, device=dev)
    live = HipUNet(cfg, sd, None, device=dev, keep_base=True)
    inp = synth.random_inputs(cfg, 1, cfg.sample_size * 8, cfg.sample_size * 8)
    pe = torch.cat([inp["null"], inp["augmented"], inp["text"]]).to(dev)
    kw = dict(prompt_embeds=pe, latents=inp["latents"].to(dev), num_inference_steps=3, guidance_scale=5.0,
              start_merge_step=0, output_type="latent")
    pipe = pipeline.ConsistentIDStableDiffusionPipeline(live)
    before = pipe(**kw).images.clone()                      # no adapters: ID stream off, plain LoRA-free attention
    addr = {k: v.data_ptr() for k, v in live.W.items()}
    torch.save({"image_proj_model": {}, "adapter_modules": ad, "FacialEncoder": {}}, tmp_path / "ConsistentID-v1.bin")
    pipe.load_ConsistentID_model(str(tmp_path), weight_name="ConsistentID-v1.bin", lora_rank=8)
    assert addr == {k: v.data_ptr() for k, v in live.W.items()}, "weights must be updated in place"
    for k in direct.W:
        assert torch.equal(direct.W[k], live.W[k]), k
    after = pipe(**kw).images
    ref = pipeline.ConsistentIDStableDiffusionPipeline(direct)(**kw).images
    torch.cuda.synchronize()
    assert torch.equal(after, ref)
    assert not torch.equal(after, before)
    bad = dict(ad)
    bad.pop(next(iter(bad)))
    with pytest.raises(RuntimeError):
        live.load_adapter_modules(bad)
    with pytest.raises(RuntimeError):
        direct.load_adapter_modules(ad)                     # built without keep_base
