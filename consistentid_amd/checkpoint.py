"""ConsistentID checkpoint files.

The released ``ConsistentID-v1.bin`` is a ``torch.save``d dict with three sub-dicts
(evaluation/convert_weights.py:14-25 writes it, pipline_StableDiffusion_ConsistentID.py:111-144 reads it):

  "adapter_modules"   ``{idx}.to_{q,k,v,out}_lora.{down,up}.weight`` and ``{idx}.to_{k,v}_ip.weight`` for every attention
                      processor, ``idx`` = position in ``unet.attn_processors`` -- the only part the denoising hot path
                      needs (``HipUNet.load_adapter_modules``)
  "image_proj"        ProjPlusModel weights      (the converter writes the key "image_proj_model", the loader reads
  "FacialEncoder"     FacialEncoder weights       "image_proj": both spellings are accepted here)

The last two belong to the once-per-image ID-conditioning stack (row f-3, consistentid_amd/idstack.py): they are
returned untouched and built into a HipIDConditioner by pipeline.load_ConsistentID_model.
"""
from __future__ import annotations

import os
from typing import Dict, Union

import torch

TRAINING_PREFIXES = (("image_proj_model.", "image_proj"), ("adapter_modules.", "adapter_modules"),
                     ("FacialEncoder.", "FacialEncoder"))


def convert_training_checkpoint(state_dict: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
    """evaluation/convert_weights.py: split a training ``pytorch_model.bin`` (flat keys ``unet.*`` frozen and dropped,
    ``image_proj_model.*``, ``adapter_modules.*``, ``FacialEncoder.*``) into the released three-part layout."""
    out = {"image_proj": {}, "adapter_modules": {}, "FacialEncoder": {}}
    for k, v in state_dict.items():
        for prefix, part in TRAINING_PREFIXES:
            if k.startswith(prefix):
                out[part][k[len(prefix):]] = v
                break
    return out


def load_checkpoint(path_or_dict: Union[str, os.PathLike, Dict], weight_name: str = "", subfolder: str = "") -> Dict:
    """The dict the reference obtains at pipline_StableDiffusion_ConsistentID.py:111-133 (local files only: there is
    no hub access here).  ``.safetensors`` files with flat ``<part>.<key>`` names are split on the first component."""
    if isinstance(path_or_dict, dict):
        sd = path_or_dict
    else:
        path = os.fspath(path_or_dict)
        if os.path.isdir(path):
            path = os.path.join(path, subfolder, weight_name) if subfolder else os.path.join(path, weight_name)
        if not os.path.isfile(path):
            raise FileNotFoundError(f"{path}: checkpoint files are read from the local file system (no hub download)")
        if path.endswith(".safetensors"):
            from safetensors import safe_open
            sd = {}
            with safe_open(path, framework="pt", device="cpu") as f:
                for key in f.keys():
                    part, _, rest = key.partition(".")
                    sd.setdefault(part, {})[rest] = f.get_tensor(key)
        else:
            sd = torch.load(path, map_location="cpu", weights_only=True)
    if "adapter_modules" not in sd and any(k.startswith("adapter_modules.") for k in sd):
        sd = convert_training_checkpoint(sd)          # a raw training checkpoint
    if "image_proj" not in sd and "image_proj_model" in sd:
        sd = dict(sd)
        sd["image_proj"] = sd.pop("image_proj_model")
    if "adapter_modules" not in sd:
        raise KeyError("checkpoint has no 'adapter_modules' entry")
    return sd
