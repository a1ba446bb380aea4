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
    assert len(GOLD) >= 2
