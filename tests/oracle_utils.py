"""Test-only helpers that build the CPU oracle with the same synthetic weights as the HIP path."""
import torch

from consistentid_amd import synth, unet_spec
from oracle import processors as oproc
from oracle import unet as ounet

_OCFG = {
    "tiny": lambda: ounet.tiny_config("sd15"), "tinyxl": lambda: ounet.tiny_config("sdxl"),
    "sd15": ounet.sd15_config, "sdxl": ounet.sdxl_config,
}
_PCFG = {
    "tiny": lambda: unet_spec.tiny_config("sd15"), "tinyxl": lambda: unet_spec.tiny_config("sdxl"),
    "sd15": unet_spec.sd15_config, "sdxl": unet_spec.sdxl_config,
}


def product_cfg(name):
    return _PCFG[name]()


def make_weights(name, rank=8, seed=0, device="cpu", **cfg_overrides):
    import dataclasses
    cfg = dataclasses.replace(product_cfg(name), **cfg_overrides)
    sd = synth.random_unet_state_dict(cfg, seed=seed, device=device)
    ad = synth.random_adapter_state_dict(cfg, sd, rank=rank, seed=seed + 1, device=device)
    return cfg, sd, ad


def build_oracle(name, sd, ad, rank=8, dtype=torch.float32, **cfg_overrides):
    """``cfg_overrides``: dataclass fields replaced in the oracle's config (e.g. in_channels=9 for an inpainting UNet)"""
    import dataclasses
    m = ounet.UNet2DConditionModel(dataclasses.replace(_OCFG[name](), **cfg_overrides))
    oproc.set_ip_adapter(m, lora_rank=rank)
    missing, unexpected = m.load_state_dict({k: v.detach().cpu().to(dtype) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(".processor." in k for k in missing), (missing[:3], unexpected[:3])
    oproc.adapter_modules(m).load_state_dict({k: v.detach().cpu().to(dtype) for k, v in ad.items()}, strict=True)
    return m.to(dtype).eval()


def idstack_weights(module, seed: int):
    """Deterministic state_dict for an identity-conditioning module (reference class or oracle restatement: same
    parameter names).  Shared by tests/golden/make_golden_idstack.py and the tests, so golden files store no weights.
    1-D ``*.weight`` are LayerNorm gains (1 + 0.1 N), other 1-D are biases (0.1 N), matrices N / sqrt(fan_in);
    everything fp16-representable so the fp16 engine sees identical values."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, t in module.state_dict().items():
        if t.ndim == 1:
            v = torch.randn(t.shape, generator=g) * 0.1 + (1.0 if name.endswith("weight") else 0.0)
        else:
            v = torch.randn(t.shape, generator=g) / t.shape[-1] ** 0.5
        sd[name] = v.half().float()
    return sd


def seeded_processor_weights(modules: dict, seed: int):
    """Deterministic weights for {prefix: module} (attention stand-ins + the two processors; reference classes or their
    oracle restatements: same parameter names).  Shared by tests/golden/make_golden.py and the tests, so the golden files
    of the real-width cases store no weights.  Every parameter draws from its own generator (seed + crc32 of its
    name), so the values do not depend on the order in which a class declares its parameters."""
    import zlib
    with torch.no_grad():
        for prefix, m in modules.items():
            for name, prm in m.named_parameters():
                g = torch.Generator().manual_seed(seed + zlib.crc32(f"{prefix}.{name}".encode()))
                fan_in = prm.shape[-1]
                s = 3.0 / fan_in ** 0.5 if name.startswith(("to_q", "to_k.")) else 1.0 / fan_in ** 0.5
                if "lora.up" in name:
                    s = 0.05
                prm.copy_((torch.randn(prm.shape, generator=g) * s).half().to(prm.dtype))
