"""SURVEY.md section 8 row f-3 on the GPU: the identity-conditioning engine (ProjPlusModel, FacialEncoder, prompt assembly)
against (a) vectors produced by the REAL reference classes (tests/golden/make_golden_idstack.py) and (b) the pinned oracle at
the production widths.  Criterion: the one of every fp16 parity test (conftest.check_vs_fp16_arm) -- error against the
fp32 / fp64 reference output <= max(1e-3, 1.5 x the error of the same modules run in fp16 with stock PyTorch-ROCm kernels
on this GPU), for the relative L2 error and for the largest single error."""
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import check_close, check_vs_fp16_arm, dev_half, half_arm
from oracle_utils import idstack_weights

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def test_gelu_and_small_attention(dev):
    from consistentid_amd import ops
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(3, 40, generator=g) * 2).half()
    y = ops.gelu_(x.to(dev).clone())
    check_close(y, torch.nn.functional.gelu(x.double()), "gelu")
    B, Lq, n1, n2, H = 2, 4, 261 - 4, 4, 3
    q = torch.randn(B * Lq, H * 64, generator=g).half()
    kv1 = torch.randn(B * n1, 2 * H * 64, generator=g).half()
    kv2 = torch.randn(B * n2, 2 * H * 64, generator=g).half()
    out = torch.empty(B * Lq, H * 64, dtype=torch.float16, device=dev)
    ops.small_attn(q.to(dev), kv1.to(dev), kv2.to(dev), out, B=B, Lq=Lq, n1=n1, n2=n2, heads=H)
    torch.cuda.synchronize()
    qd = q.double().view(B, Lq, H, 64).transpose(1, 2)
    kv = torch.cat([kv1.double().view(B, n1, -1), kv2.double().view(B, n2, -1)], dim=1)
    k, v = kv.chunk(2, dim=-1)
    k, v = k.view(B, n1 + n2, H, 64).transpose(1, 2), v.view(B, n1 + n2, H, 64).transpose(1, 2)
    ref = (torch.softmax(qd @ k.transpose(-1, -2) / 8.0, dim=-1) @ v).transpose(1, 2).reshape(B * Lq, H * 64)
    check_close(out, ref, "small_attn")


@pytest.mark.parametrize("tag", ["small", "sd15"])
def test_projplus_matches_reference_golden(dev, tag):
    """HipProjPlusModel vs outputs of the reference's own ProjPlusModel (functions.py:490-522)"""
    from consistentid_amd.idstack import HipProjPlusModel
    from oracle import idstack
    z = np.load(GOLD / f"idstack_projplus_{tag}.npz")
    ca, idd, clipd, nt = [int(v) for v in z["kw"]]
    sd = idstack_weights(idstack.ProjPlusModel(cross_attention_dim=ca, id_embeddings_dim=idd, clip_embeddings_dim=clipd,
                                               num_tokens=nt), int(z["seed"]))
    hip = HipProjPlusModel(sd, device=dev)
    assert (hip.cross_attention_dim, hip.num_tokens) == (ca, nt)
    ide, clip = torch.from_numpy(z["id_embeds"]), torch.from_numpy(z["clip_embeds"])
    o0 = hip(ide, clip)
    o1 = hip(ide, clip, shortcut=True, scale=0.7)
    torch.cuda.synchronize()
    om = idstack.ProjPlusModel(cross_attention_dim=ca, id_embeddings_dim=idd, clip_embeddings_dim=clipd, num_tokens=nt)
    om.load_state_dict(sd)
    arm_m = half_arm(om, dev)
    with torch.no_grad():
        a0 = arm_m(ide.to(dev).half(), clip.to(dev).half())
        a1 = arm_m(ide.to(dev).half(), clip.to(dev).half(), shortcut=True, scale=0.7)
    check_vs_fp16_arm(o0, torch.from_numpy(z["out"]), a0, f"ProjPlusModel {tag}")
    check_vs_fp16_arm(o1, torch.from_numpy(z["out_shortcut"]), a1, f"ProjPlusModel {tag} shortcut")


def test_facial_encoder_matches_reference_golden(dev):
    """HipFacialEncoder vs outputs of the reference's own FacialEncoder (attention.py:72-88)"""
    from consistentid_amd.idstack import HipFacialEncoder
    from oracle import idstack
    z = np.load(GOLD / "idstack_facial_encoder.npz")
    om = idstack.FacialEncoder(embedding_dim=192, output_dim=128, embed_dim=128)
    sd = idstack_weights(om, int(z["seed"]))
    om.load_state_dict(sd)
    hip = HipFacialEncoder(sd, device=dev)
    args = (torch.from_numpy(z["prompt_embeds"]), torch.from_numpy(z["multi_image_embeds"]),
            torch.from_numpy(z["class_tokens_mask"]), torch.from_numpy(z["valid_id_mask"]))
    out = hip(*args)
    torch.cuda.synchronize()
    ref = torch.from_numpy(z["out"])
    with torch.no_grad():
        arm = half_arm(om, dev)(*dev_half(args, dev))
    check_vs_fp16_arm(out, ref, arm, "FacialEncoder")
    cm = torch.from_numpy(z["class_tokens_mask"])
    assert torch.equal(out.cpu()[~cm], torch.from_numpy(z["prompt_embeds"]).half()[~cm]), "other prompt rows untouched"


def test_prompt_assembly_production_widths(dev):
    """ProjPlusModel (768 / 512 / 1280, 257 CLIP tokens) + FacialEncoder (1280 -> 768, 5 crops) + the reference's cat order
    (:479-507), against the pinned oracle; the result feeds the pipeline as ``prompt_embeds``."""
    from consistentid_amd.idstack import HipIDConditioner
    from oracle import idstack
    o_ip, o_fe = idstack.ProjPlusModel(), idstack.FacialEncoder()
    sd_ip, sd_fe = idstack_weights(o_ip, 21), idstack_weights(o_fe, 22)
    o_ip.load_state_dict(sd_ip)
    o_fe.load_state_dict(sd_fe)
    g = torch.Generator().manual_seed(9)
    rnd = lambda *s: torch.randn(*s, generator=g).half()
    B = 1
    kw = dict(text_embeds=rnd(B, 77, 768), negative_embeds=rnd(B, 77, 768), text_only_embeds=rnd(B, 77, 768),
              faceid_embeds=rnd(B, 512), clip_embeds=rnd(B, 257, 1280), uncond_clip_embeds=rnd(B, 257, 1280),
              facial_embeds=rnd(B, 5, 257, 1280), uncond_facial_embeds=rnd(B, 5, 257, 1280))
    fmask = torch.zeros(B, 77, dtype=torch.bool)
    fmask[0, [4, 9, 15]] = True
    vmask = torch.tensor([[True, True, True, False, False]])
    ref = idstack.assemble_prompt_embeds(o_ip.eval(), o_fe.eval(), **{k: v.float() for k, v in kw.items()},
                                         facial_token_mask=fmask, valid_facial_mask=vmask)
    hip = HipIDConditioner(sd_ip, sd_fe, device=dev)
    out = hip(**kw, facial_token_mask=fmask, valid_facial_mask=vmask)
    torch.cuda.synchronize()
    assert out.shape == ref.shape == (3 * B, 81, 768)
    with torch.no_grad():
        arm = idstack.assemble_prompt_embeds(half_arm(o_ip, dev), half_arm(o_fe, dev), **dev_half(kw, dev),
                                             facial_token_mask=fmask.to(dev), valid_facial_mask=vmask.to(dev))
    check_vs_fp16_arm(out, ref, arm, "prompt_embeds assembly")


def _clip_pair(dev, cfg_kw, seed):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    from consistentid_amd.clip_vision import HipCLIPVision
    cfg = CLIPVisionConfig(hidden_act="gelu", **cfg_kw)
    torch.manual_seed(seed)
    ref = CLIPVisionModelWithProjection(cfg).eval()
    with torch.no_grad():                       # fp16-representable weights; LayerNorm gains / biases off their init
        for n, p in ref.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.05)
            p.copy_(p.half().float())
    hip = HipCLIPVision(ref.state_dict(), num_heads=cfg.num_attention_heads, patch_size=cfg.patch_size, device=dev)
    return cfg, ref, hip


@pytest.mark.parametrize("name", ["small", "vit_h"])
def test_clip_vision_hidden_states(dev, name):
    """HipCLIPVision.hidden_states(-2) vs transformers' CLIPVisionModelWithProjection (the class the reference loads,
    ref :54-56) on random weights: a 2-head toy tower and the real ViT-H/14 geometry (632 M parameters, 257 tokens)."""
    kw = dict(small=dict(hidden_size=128, intermediate_size=512, num_hidden_layers=4, num_attention_heads=2, image_size=70,
                         patch_size=14, projection_dim=64),
              vit_h=dict(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224,
                         patch_size=14, projection_dim=1024))[name]
    cfg, ref, hip = _clip_pair(dev, kw, seed=3)
    B = 2 if name == "small" else 1
    img = torch.randn(B, 3, cfg.image_size, cfg.image_size, generator=torch.Generator().manual_seed(4)).half()
    with torch.no_grad():
        hs = ref(img.float(), output_hidden_states=True).hidden_states
    out = hip.hidden_states(img.to(dev), -2)
    torch.cuda.synchronize()
    assert out.shape == hs[-2].shape
    arm_m = half_arm(ref, dev)       # transformers' own modules in fp16 on this GPU: what the reference's pipeline runs
    with torch.no_grad():
        ahs = arm_m(img.to(dev), output_hidden_states=True).hidden_states
    check_vs_fp16_arm(out, hs[-2], ahs[-2], f"CLIP vision {name} hidden_states[-2]")
    if name == "small":
        check_vs_fp16_arm(hip.hidden_states(img.to(dev), 0), hs[0], ahs[0], "CLIP vision embeddings + pre-LN")
        zero = hip.hidden_states(torch.zeros_like(img).to(dev), -2)      # the reference's "uncond" image (ref :183, :201)
        with torch.no_grad():
            zref = ref(torch.zeros_like(img).float(), output_hidden_states=True).hidden_states[-2]
            zarm = arm_m(torch.zeros_like(img).to(dev), output_hidden_states=True).hidden_states[-2]
        check_vs_fp16_arm(zero, zref, zarm, "CLIP vision of a zero image")
