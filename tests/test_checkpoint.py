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
    for k in direct.W:      # (bit patterns: the fp32 fold vectors travel in fp16-typed tensors, whose halves may read as NaN)
        assert torch.equal(direct.W[k].view(torch.int16), live.W[k].view(torch.int16)), k
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


@pytest.mark.gpu
def test_checkpoint_to_latents_end_to_end(dev, tmp_path):
    """ConsistentID-v1.bin (all three parts) -> load_ConsistentID_model -> prepare_prompt_embeds from encoder outputs ->
    denoise, against the oracle chain (idstack.assemble_prompt_embeds -> loop.denoise)."""
    from consistentid_amd import pipeline
    from consistentid_amd.unet import HipUNet
    from oracle import ddim, idstack, loop
    from oracle_utils import build_oracle, idstack_weights
    from conftest import check_close
    cfg, sd, ad = _adapters()
    Dc = cfg.cross_attention_dim
    o_ip = idstack.ProjPlusModel(cross_attention_dim=Dc, id_embeddings_dim=64, clip_embeddings_dim=192, num_tokens=4)
    o_fe = idstack.FacialEncoder(embedding_dim=192, output_dim=Dc, embed_dim=Dc)
    sd_ip, sd_fe = idstack_weights(o_ip, 31), idstack_weights(o_fe, 32)
    o_ip.load_state_dict(sd_ip)
    o_fe.load_state_dict(sd_fe)
    torch.save({"image_proj_model": sd_ip, "adapter_modules": ad, "FacialEncoder": sd_fe}, tmp_path / "ConsistentID-v1.bin")
    g = torch.Generator().manual_seed(17)
    rnd = lambda *s: torch.randn(*s, generator=g).half()
    enc = dict(text_embeds=rnd(1, 77, Dc), negative_embeds=rnd(1, 77, Dc), text_only_embeds=rnd(1, 77, Dc),
               faceid_embeds=rnd(1, 64), clip_embeds=rnd(1, 33, 192), uncond_clip_embeds=rnd(1, 33, 192),
               facial_embeds=rnd(1, 5, 33, 192), uncond_facial_embeds=rnd(1, 5, 33, 192))
    fmask = torch.zeros(1, 77, dtype=torch.bool)
    fmask[0, [3, 8]] = True
    vmask = torch.tensor([[True, False, True, False, False]])
    lat = synth.random_inputs(cfg, 1, cfg.sample_size * 8, cfg.sample_size * 8)["latents"]
    # oracle chain
    pe_ref = idstack.assemble_prompt_embeds(o_ip.eval(), o_fe.eval(), **{k: v.float() for k, v in enc.items()},
                                            facial_token_mask=fmask, valid_facial_mask=vmask)
    null_e, aug_e, text_e = pe_ref.chunk(3)
    o_unet = build_oracle("tiny", sd, ad, rank=8)
    ref = loop.denoise(o_unet, ddim.DDIMScheduler(), lat.float(), null_e, aug_e, text_e, num_inference_steps=3,
                       guidance_scale=5.0, start_merge_step=1)
    # product chain
    pipe = pipeline.ConsistentIDStableDiffusionPipeline(HipUNet(cfg, sd, None, device=dev, keep_base=True))
    with pytest.raises(RuntimeError):
        pipe.prepare_prompt_embeds(**enc, facial_token_mask=fmask, valid_facial_mask=vmask)
    pipe.load_ConsistentID_model(str(tmp_path / "ConsistentID-v1.bin"), lora_rank=8)
    pe = pipe.prepare_prompt_embeds(**enc, facial_token_mask=fmask, valid_facial_mask=vmask)
    assert pe.shape == (3, 81, Dc)
    check_close(pe, pe_ref, "prompt_embeds from the checkpoint's ID modules", tol_l2=3e-3, tol_max=1.5e-2)
    out = pipe(prompt_embeds=pe, latents=lat.to(dev), num_inference_steps=3, guidance_scale=5.0, start_merge_step=1,
               output_type="latent").images
    torch.cuda.synchronize()
    check_close(out, ref, "checkpoint -> prompt embeds -> latents", tol_l2=6e-3, tol_max=2.5e-2)
