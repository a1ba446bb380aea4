#!/usr/bin/env python
"""Golden vectors for the identity-conditioning modules, produced by the REAL reference classes
(/root/reference/functions.py: ProjPlusModel; /root/reference/attention.py: FacialEncoder) in fp64.

Shims: ``cv2`` (imported by functions.py, unused by these classes) and the diffusers symbols attention.py imports
(unused by FacialEncoder).  Weights are NOT stored: both this script and the tests draw them from
``tests/oracle_utils.idstack_weights(module, seed)`` (parameter-name order, seeded CPU generator), so a golden file holds only the
inputs and the reference's outputs.  FacialEncoder hard-codes its AttentionMLP at dim 1024 / depth 8 / 16 heads
(attention.py:75), so that part runs at the real width on a short token axis.

Run (only where /root/reference exists):  python tests/golden/make_golden_idstack.py
"""
import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


sys.path.insert(0, str(OUT.parent))           # tests/: oracle_utils.idstack_weights (weights from a seed, shared with the tests)
sys.path.insert(0, str(OUT.parent.parent))    # repo root
from oracle_utils import idstack_weights  # noqa: E402


def _load_reference():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    mod("cv2")
    mod("diffusers"); mod("diffusers.models"); mod("diffusers.utils")
    mod("diffusers.models.lora", LoRALinearLayer=type("LoRALinearLayer", (nn.Module,), {}))
    mod("diffusers.utils.import_utils", is_xformers_available=lambda: False)
    sys.path.insert(0, str(REF))
    out = {}
    for name in ("functions", "attention"):
        spec = importlib.util.spec_from_file_location(name, REF / f"{name}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        out[name] = m
    return out["functions"], out["attention"]


def main():
    functions, attention = _load_reference()
    g = torch.Generator().manual_seed(123)
    rnd = lambda *s: torch.randn(*s, generator=g).half().double()
    # ---- ProjPlusModel at reduced widths (constructor arguments), 2 samples, 19 CLIP tokens
    for tag, kw, ntok in (("small", dict(cross_attention_dim=128, id_embeddings_dim=64, clip_embeddings_dim=192, num_tokens=4), 19),
                          ("sd15", dict(cross_attention_dim=768, id_embeddings_dim=512, clip_embeddings_dim=1280, num_tokens=4), 9)):
        m = functions.ProjPlusModel(**kw)
        m.load_state_dict(idstack_weights(m, seed=11), strict=True)
        m = m.double().eval()
        ide, clip = rnd(2, kw["id_embeddings_dim"]), rnd(2, ntok, kw["clip_embeddings_dim"])
        with torch.no_grad():
            o0 = m(ide, clip, shortcut=False, scale=1.0)
            o1 = m(ide, clip, shortcut=True, scale=0.7)
        np.savez_compressed(OUT / f"idstack_projplus_{tag}.npz", seed=11, kw=np.array(list(kw.values())),
                            id_embeds=ide.float().numpy(), clip_embeds=clip.float().numpy(),
                            out=o0.float().numpy(), out_shortcut=o1.float().numpy())
    # ---- FacialEncoder (AttentionMLP fixed at dim 1024, depth 8): 2 prompts x 3 facial crops x 7 CLIP tokens
    fe = attention.FacialEncoder(embedding_dim=192, output_dim=128, embed_dim=128)
    fe.load_state_dict(idstack_weights(fe, seed=12), strict=True)
    fe = fe.double().eval()
    pe, mi = rnd(2, 13, 128), rnd(2, 3, 7, 192)
    cmask = torch.zeros(2, 13, dtype=torch.bool)
    cmask[0, [2, 5]] = True
    cmask[1, [7]] = True
    vmask = torch.tensor([[True, True, False], [False, True, False]])
    with torch.no_grad():
        out = fe(pe.clone(), mi, cmask, vmask)
    np.savez_compressed(OUT / "idstack_facial_encoder.npz", seed=12, prompt_embeds=pe.float().numpy(),
                        multi_image_embeds=mi.float().numpy(), class_tokens_mask=cmask.numpy(),
                        valid_id_mask=vmask.numpy(), out=out.float().numpy())
    print("wrote", sorted(p.name for p in OUT.glob("idstack_*.npz")))


if __name__ == "__main__":
    main()
