"""Base-model loader: a diffusers model directory on disk -> the HIP engines.

The reference builds its pipelines with ``ConsistentIDPipeline.from_pretrained(base_model_path, torch_dtype=torch.float16)``
(/root/reference/infer.py:17-21, infer_SDXL.py:19-25) -- diffusers' loader reading ``model_index.json`` and one
sub-folder per component.  This module reads the components the hot path needs, straight from their files:

    <root>/unet/config.json + diffusion_pytorch_model.{safetensors,bin}     -> HipUNet      (required)
    <root>/vae/config.json  + diffusion_pytorch_model.{safetensors,bin}     -> HipVAEDecoder (optional)
    <controlnet dir>/config.json + diffusion_pytorch_model.{safetensors,bin} -> HipControlNet (separate call, like the
                                                                                reference: demo/controlnet_demo.py:44-47)

Local files only (there is no hub access in the engine), ``.fp16`` variants are picked up, sharded checkpoints are
not.  Tokenizer / text encoders / safety checker folders are left alone: prompt encoding is pre-loop.
``scheduler/scheduler_config.json`` is read for its betas / steps_offset / timestep spacing (the engine carries its own
DDIM / Euler coefficient tables, scheduler.py; the base model's sampler class itself is not built).

Host Python + file IO; the tensors go to the GPU inside the engines' weight packers.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional, Sequence, Tuple, Union

import torch

from .unet_spec import UNetConfig
from .vae_spec import VAEConfig

WEIGHT_NAMES = ("diffusion_pytorch_model.fp16.safetensors", "diffusion_pytorch_model.safetensors",
                "diffusion_pytorch_model.fp16.bin", "diffusion_pytorch_model.bin")


def _tup(v, n) -> Tuple[int, ...]:
    return tuple(v) if isinstance(v, (list, tuple)) else (int(v),) * n


def unet_config_from_diffusers(cfg: Dict) -> UNetConfig:
    """``unet/config.json`` (diffusers UNet2DConditionModel / ControlNetModel) -> UNetConfig.

    diffusers 0.23 quirk kept: ``num_attention_heads = num_attention_heads or attention_head_dim`` -- for the SD1.5 and
    SDXL checkpoints the field called attention_head_dim holds the head COUNT per level (8 / [5, 10, 20])."""
    boc = tuple(cfg["block_out_channels"])
    n = len(boc)
    heads = cfg.get("num_attention_heads") or cfg.get("attention_head_dim", 8)
    down = tuple(cfg.get("down_block_types", ("CrossAttnDownBlock2D",) * (n - 1) + ("DownBlock2D",)))
    up = cfg.get("up_block_types")
    if up is None:      # ControlNet configs have no decoder: mirror the encoder so that the shared topology walk works
        up = tuple("CrossAttnUpBlock2D" if t.startswith("CrossAttn") else "UpBlock2D" for t in reversed(down))
    for t in tuple(down) + tuple(up):
        if t not in ("CrossAttnDownBlock2D", "DownBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"):
            raise NotImplementedError(f"block type {t}: only the SD1.5 / SDXL UNet topologies are built")
    if cfg.get("mid_block_type", "UNetMidBlock2DCrossAttn") != "UNetMidBlock2DCrossAttn":
        raise NotImplementedError(f"mid block {cfg['mid_block_type']}")
    expected = {"class_embed_type": (None,), "encoder_hid_dim": (None,), "only_cross_attention": (False, None),
                "dual_cross_attention": (False, None), "upcast_attention": (False, None),
                "resnet_time_scale_shift": ("default", None), "time_embedding_type": ("positional", None),
                "act_fn": ("silu", None), "class_embeddings_concat": (False, None), "conv_in_kernel": (3, None),
                "conv_out_kernel": (3, None), "flip_sin_to_cos": (True, None), "freq_shift": (0, None)}
    for key, ok in expected.items():
        if cfg.get(key) not in ok:
            raise NotImplementedError(f"UNet config {key}={cfg[key]!r} is outside the SD1.5 / SDXL geometry the engine builds")
    add = cfg.get("addition_embed_type")
    if add not in (None, "text_time"):
        raise NotImplementedError(f"addition_embed_type {add!r}")
    return UNetConfig(
        sample_size=int(cfg.get("sample_size", 64)), in_channels=int(cfg.get("in_channels", 4)),
        out_channels=int(cfg.get("out_channels", 4)), block_out_channels=boc, down_block_types=down, up_block_types=tuple(up),
        layers_per_block=int(cfg.get("layers_per_block", 2)),
        transformer_layers_per_block=_tup(cfg.get("transformer_layers_per_block", 1), n),
        num_attention_heads=_tup(heads, n), cross_attention_dim=int(cfg.get("cross_attention_dim", 768)),
        norm_num_groups=int(cfg.get("norm_num_groups", 32)), norm_eps=float(cfg.get("norm_eps", 1e-5)),
        use_linear_projection=bool(cfg.get("use_linear_projection", False)), addition_embed_type=add,
        addition_time_embed_dim=cfg.get("addition_time_embed_dim"),
        projection_class_embeddings_input_dim=cfg.get("projection_class_embeddings_input_dim"),
        family="sdxl" if add == "text_time" else "sd15")


def vae_config_from_diffusers(cfg: Dict) -> VAEConfig:
    return VAEConfig(in_channels=int(cfg.get("in_channels", 3)), out_channels=int(cfg.get("out_channels", 3)),
                     latent_channels=int(cfg.get("latent_channels", 4)), block_out_channels=tuple(cfg["block_out_channels"]),
                     layers_per_block=int(cfg.get("layers_per_block", 2)), norm_num_groups=int(cfg.get("norm_num_groups", 32)),
                     scaling_factor=float(cfg.get("scaling_factor", 0.18215)), force_upcast=bool(cfg.get("force_upcast", False)))


def read_component(folder: Union[str, os.PathLike]) -> Tuple[Dict, Dict[str, torch.Tensor]]:
    """(config.json as dict, state_dict on the CPU) of one diffusers component folder"""
    folder = os.fspath(folder)
    cpath = os.path.join(folder, "config.json")
    if not os.path.isfile(cpath):
        raise FileNotFoundError(f"{cpath}: not a diffusers component folder (the engine reads local files only)")
    with open(cpath) as f:
        cfg = json.load(f)
    for name in WEIGHT_NAMES:
        wpath = os.path.join(folder, name)
        if os.path.isfile(wpath):
            break
    else:
        if os.path.isfile(os.path.join(folder, "diffusion_pytorch_model.safetensors.index.json")):
            raise NotImplementedError(f"{folder}: sharded checkpoints are not read; merge the shards first")
        raise FileNotFoundError(f"{folder}: none of {WEIGHT_NAMES}")
    if wpath.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(wpath, device="cpu")
    else:
        sd = torch.load(wpath, map_location="cpu", weights_only=True)
    return cfg, sd


def load_unet(root: Union[str, os.PathLike], device="cuda:0", subfolder: str = "unet", keep_base: bool = True, **kw):
    """-> HipUNet.  ``keep_base`` keeps the un-merged attention weights so that ``load_ConsistentID_model`` can merge a
    ConsistentID checkpoint afterwards (the reference's order: from_pretrained, then load_ConsistentID_model)."""
    from .unet import HipUNet
    cfg, sd = read_component(os.path.join(os.fspath(root), subfolder) if subfolder else root)
    return HipUNet(unet_config_from_diffusers(cfg), sd, None, device=device, keep_base=keep_base, **kw)


def load_controlnet(folder: Union[str, os.PathLike], device="cuda:0"):
    from .controlnet import HipControlNet
    cfg, sd = read_component(folder)
    return HipControlNet(unet_config_from_diffusers(cfg), sd, device=device)


def load_vae(root: Union[str, os.PathLike], device="cuda:0", subfolder: str = "vae"):
    from .vae import make_vae_decoder
    cfg, sd = read_component(os.path.join(os.fspath(root), subfolder) if subfolder else root)
    return make_vae_decoder(vae_config_from_diffusers(cfg), sd, device=device)      # fp32 engine for force_upcast (SDXL)


def read_scheduler(root: Union[str, os.PathLike]):
    """``<root>/scheduler/scheduler_config.json`` -> the engine's DDIMScheduler on that config (None without the file).
    The base model's own sampler class (PNDM for SD1.5) is not built; its CONFIG is kept, so that the reference scripts' next
    line -- ``pipe.scheduler = EulerDiscreteScheduler.from_config(pipe.scheduler.config)`` (infer.py:33) -- sees the base
    model's betas / steps_offset / timestep spacing.  Until then the engine's DDIM runs on that config."""
    spath = os.path.join(os.fspath(root), "scheduler", "scheduler_config.json")
    if not os.path.isfile(spath):
        return None
    import warnings
    from .scheduler import DDIMScheduler, EulerDiscreteScheduler
    with open(spath) as f:
        cfg = json.load(f)
    # the base model's sampler class: SDXL base ships EulerDiscrete (built), SD1.5 PNDM (not built: DDIM on its config)
    name = cfg.get("_class_name", "DDIMScheduler")
    cls = EulerDiscreteScheduler if name == "EulerDiscreteScheduler" else DDIMScheduler
    if name not in ("EulerDiscreteScheduler", "DDIMScheduler"):
        warnings.warn(f"{name} is not built: the engine samples with DDIM on its config until pipe.scheduler is replaced "
                      "(infer.py:33 / demo/controlnet_demo.py:67 do that on the next line)")
    try:
        return cls.from_config(cfg)
    except NotImplementedError as e:     # e.g. clip_sample=true of a saved DDIM default: must not block the load
        # drop ONLY the sampling modifiers the engine does not build and keep the rest of the saved config (betas,
        # steps_offset, timestep spacing); defaults only if the schedule itself (beta schedule / prediction type) is foreign
        # clip_sample / thresholding act on the predicted sample only (clip_sample clamps pred_x0 to [-1, 1], which DOES
        # change SD latents where it is on); the engine does not build them and ignores them with a warning.  use_karras_sigmas / rescale_betas_zero_snr / a non-linear interpolation_type change the sigma / alpha
        # SCHEDULE itself: a checkpoint saved with them would sample on a materially different schedule here, so that is an
        # error, not a warning.
        harmless = ("clip_sample", "thresholding")
        schedule = ("use_karras_sigmas", "interpolation_type", "rescale_betas_zero_snr")
        changed = sorted(k for k in schedule if cfg.get(k) not in (None, False, "linear"))
        if changed:
            raise NotImplementedError(f"{spath}: {changed} change the noise schedule and are not built; the engine would not "
                                      f"reproduce the reference's samples -- pass the scheduler to use explicitly, "
                                      f"`from_pretrained(..., scheduler=DDIMScheduler(...))` (the saved config is then not "
                                      f"read; infer.py:33 replaces the loaded scheduler in the same way)") from e
        kept = {k: v for k, v in cfg.items() if k not in harmless}
        dropped = sorted(k for k in harmless if cfg.get(k) not in (None, False))
        try:
            sch = cls.from_config(kept)
            warnings.warn(f"{spath}: {e}; the engine does not build {dropped} and IGNORES them (samples differ from a "
                          "scheduler that applies them); the rest of the saved scheduler config is kept")
            return sch
        except NotImplementedError as e2:
            warnings.warn(f"{spath}: {e2}; falling back to the engine's default {cls.__name__} configuration")
            return cls()


def from_pretrained(pipeline_cls, root: Union[str, os.PathLike], torch_dtype=torch.float16, device="cuda:0",
                    controlnet: Optional[Union[str, os.PathLike, object]] = None, use_graph: bool = True, **kw):
    """``Pipeline.from_pretrained(base_model_path, torch_dtype=torch.float16)`` of the reference scripts (infer.py:17-21;
    with ``controlnet=`` demo/controlnet_demo.py:44-47): UNet (required) and VAE decoder (when the folder exists) of a
    local diffusers model directory; ``controlnet`` = a HipControlNet or the folder of a ControlNetModel."""
    if torch_dtype not in (torch.float16, None):
        raise NotImplementedError("the engine computes in fp16 (the reference's own inference dtype, infer.py:19)")
    # diffusers loader keywords the reference scripts pass (infer.py:17-21 variant="fp16"; infer_SDXL.py safety_checker=None;
    # demo/controlnet_demo.py use_safetensors=True): they select files / components the engine does not read -- the
    # .fp16 weight files are preferred anyway, there is no safety checker and no hub access
    for name in ("variant", "use_safetensors", "safety_checker", "feature_extractor", "requires_safety_checker",
                 "local_files_only", "cache_dir", "revision", "low_cpu_mem_usage", "add_watermarker"):
        kw.pop(name, None)
    unknown = [k for k in kw if k not in ("scheduler", "num_tokens", "lora_rank")]
    if unknown:
        raise TypeError(f"from_pretrained: unexpected keyword(s) {unknown}")
    root = os.fspath(root)
    if not os.path.isdir(root):
        raise FileNotFoundError(f"{root}: local diffusers model directory expected (no hub access)")
    unet = load_unet(root, device=device)
    vae = load_vae(root, device=device) if os.path.isdir(os.path.join(root, "vae")) else None
    args = dict(use_graph=use_graph, vae=vae, **kw)
    if "scheduler" not in args:
        base = read_scheduler(root)
        if base is not None:
            args["scheduler"] = base
    if controlnet is not None:
        cn = load_controlnet(controlnet, device=device) if isinstance(controlnet, (str, os.PathLike)) else controlnet
        return pipeline_cls(unet, controlnet=cn, **args)
    return pipeline_cls(unet, **args)
