"""Pin the oracle's processors against vectors produced by the REAL reference code
(tests/golden/make_golden.py ran /root/reference/attention.py:90-294 in fp64)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import processors as oproc
from oracle.unet import Attention

GOLD = sorted((Path(__file__).parent / "golden").glob("processors_*.npz"))


def load_case(path, dtype=torch.float64):
    z = np.load(path)
    B, N, C, heads, Dc, L, rank = [int(v) for v in z["meta"]]
    t = lambda k: torch.from_numpy(z[k]).to(dtype)
    attn1, attn2 = Attention(C, None, heads).to(dtype), Attention(C, Dc, heads).to(dtype)
    p1 = oproc.Consistent_AttProcessor(hidden_size=C, cross_attention_dim=None, rank=rank).to(dtype)
    p2 = oproc.Consistent_IPAttProcessor(hidden_size=C, cross_attention_dim=Dc, rank=rank,
                                         scale=float(z["ip_scale"]), num_tokens=4).to(dtype)
    if "seed" in z.files:       # real-width cases: weights from the shared seeded generator, not stored
        from oracle_utils import seeded_processor_weights
        seeded_processor_weights({"attn1": attn1, "attn2": attn2, "proc1": p1, "proc2": p2}, int(z["seed"]))
    else:
        for prefix, m in (("attn1", attn1), ("attn2", attn2), ("proc1", p1), ("proc2", p2)):
            sd = {k[len(prefix) + 1:]: t(k) for k in z.files if k.startswith(prefix + ".")}
            m.load_state_dict(sd, strict=True)
    return dict(attn1=attn1, attn2=attn2, p1=p1, p2=p2, hidden=t("hidden"), ehs=t("ehs"),
                out_self=t("out_self"), out_ip=t("out_ip"), meta=(B, N, C, heads, Dc, L, rank),
                ip_scale=float(z["ip_scale"]))


@pytest.mark.parametrize("path", GOLD, ids=lambda p: p.stem)
def test_oracle_processors_match_reference(path):
    c = load_case(path)
    with torch.no_grad():
        o1 = c["p1"](c["attn1"], c["hidden"])
        o2 = c["p2"](c["attn2"], c["hidden"], encoder_hidden_states=c["ehs"])
    # golden outputs were stored as fp32 of an fp64 computation
    assert torch.allclose(o1.float(), c["out_self"].float(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(o2.float(), c["out_ip"].float(), rtol=1e-5, atol=1e-6)


def test_golden_present():
    assert len(GOLD) >= 7       # two small cases + the five real SD1.5 / SDXL (width, heads) combinations


# ----------------------------------------------------------------------------- identity-conditioning stack (row f-3)
def _npz(name):
    return np.load(Path(__file__).parent / "golden" / name)


@pytest.mark.parametrize("tag", ["small", "sd15"])
def test_projplus_restatement_matches_reference(tag):
    """oracle.idstack.ProjPlusModel vs vectors produced by the REAL functions.py:490-522 (make_golden_idstack.py)"""
    from oracle import idstack
    from oracle_utils import idstack_weights
    z = _npz(f"idstack_projplus_{tag}.npz")
    ca, idd, clipd, nt = [int(v) for v in z["kw"]]
    m = idstack.ProjPlusModel(cross_attention_dim=ca, id_embeddings_dim=idd, clip_embeddings_dim=clipd, num_tokens=nt)
    m.load_state_dict(idstack_weights(m, int(z["seed"])), strict=True)
    m = m.double().eval()
    ide, clip = torch.from_numpy(z["id_embeds"]).double(), torch.from_numpy(z["clip_embeds"]).double()
    with torch.no_grad():
        o0, o1 = m(ide, clip), m(ide, clip, shortcut=True, scale=0.7)
    assert torch.allclose(o0.float(), torch.from_numpy(z["out"]), atol=1e-5, rtol=1e-5)
    assert torch.allclose(o1.float(), torch.from_numpy(z["out_shortcut"]), atol=1e-5, rtol=1e-5)


def test_facial_encoder_restatement_matches_reference():
    """oracle.idstack.FacialEncoder vs the REAL attention.py:72-88 (AttentionMLP functions.py:524-593, FuseModule :10-48)"""
    from oracle import idstack
    from oracle_utils import idstack_weights
    z = _npz("idstack_facial_encoder.npz")
    m = idstack.FacialEncoder(embedding_dim=192, output_dim=128, embed_dim=128)
    m.load_state_dict(idstack_weights(m, int(z["seed"])), strict=True)
    m = m.double().eval()
    with torch.no_grad():
        out = m(torch.from_numpy(z["prompt_embeds"]).double(), torch.from_numpy(z["multi_image_embeds"]).double(),
                torch.from_numpy(z["class_tokens_mask"]), torch.from_numpy(z["valid_id_mask"]))
    assert torch.allclose(out.float(), torch.from_numpy(z["out"]), atol=1e-5, rtol=1e-5)
