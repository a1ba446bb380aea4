"""Oracle (test infrastructure): the per-step glue of the reference pipelines'
``__call__`` loops, restated for B independent samples.

  SD1.5        pipline_StableDiffusion_ConsistentID.py:527-579
  SDXL         pipline_StableDiffusionXL_ConsistentID.py:611-667
  inpaint      pipelines/StableDIffusionInpaint_ConsistentID.py:305-359
  CN-inpaint   pipelines/StableDIffusionControlNetInpaint_ConsistentID.py:375-456

PARITY UNPINNED for the scheduler arithmetic (diffusers, see ddim.py); the control
flow (embed switch at ``i <= start_merge_step``, CFG combine, mask blend, the
ControlNet residual broadcast quirk) follows the cited reference lines.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

from .ddim import DDIMScheduler


@torch.no_grad()
def denoise(unet, scheduler: DDIMScheduler, latents: torch.Tensor,
            null_embeds: torch.Tensor, augmented_embeds: torch.Tensor, text_embeds: torch.Tensor,
            num_inference_steps: int, guidance_scale: float, start_merge_step: int,
            # SDXL extras (ref SDXL :620-631)
            add_text_embeds_null: Optional[torch.Tensor] = None,
            add_text_embeds_text: Optional[torch.Tensor] = None,
            add_text_embeds_aug: Optional[torch.Tensor] = None,
            add_time_ids: Optional[torch.Tensor] = None,
            # ControlNet-inpaint extras (ref CN :389-449)
            down_residuals: Optional[Sequence[torch.Tensor]] = None,
            mid_residual: Optional[torch.Tensor] = None,
            inpaint_mask: Optional[torch.Tensor] = None,
            inpaint_init: Optional[torch.Tensor] = None,
            inpaint_noise: Optional[torch.Tensor] = None,
            # native ControlNet (ref CN :364-412): residuals recomputed every step from the current latents
            controlnet=None, control_image: Optional[torch.Tensor] = None, conditioning_scale: float = 1.0,
            control_guidance_start: float = 0.0, control_guidance_end: float = 1.0,
            on_step: Optional[Callable[[int, torch.Tensor, torch.Tensor], None]] = None,
            # SDXL: the unconditional set used after the merge step (ref SDXL :586-590, :620-631); None = the same one
            null_embeds_post: Optional[torch.Tensor] = None,
            # inpaint pipelines: strength < 1 keeps the last int(S * strength) timesteps (get_timesteps, inpaint ref :246-252)
            strength: float = 1.0,
            # 9-channel inpainting UNets: cat([mask, masked_image_latents]) [B,5,h,w] (inpaint ref :320-321, CN :415-416)
            unet_extra: Optional[torch.Tensor] = None):
    """Returns the final latents [B,4,h,w].  ``*_embeds`` are [B,L,Dc] (L = 77 + 4)."""
    scheduler.set_timesteps(num_inference_steps)
    timesteps = scheduler.timesteps
    t_start = max(num_inference_steps - min(int(num_inference_steps * strength), num_inference_steps), 0)
    timesteps = timesteps[t_start:]
    for i, t in enumerate(timesteps):
        lat_in = scheduler.scale_model_input(torch.cat([latents] * 2), t)          # SD :537-540
        merged = i > start_merge_step                                              # SD :542-549
        cond = augmented_embeds if merged else text_embeds
        null = null_embeds_post if (merged and null_embeds_post is not None) else null_embeds
        ehs = torch.cat([null, cond], dim=0)
        kw = {}
        if add_time_ids is not None:                                               # SDXL :620-631
            pooled = add_text_embeds_aug if merged else add_text_embeds_text
            kw["added_cond_kwargs"] = {
                "text_embeds": torch.cat([add_text_embeds_null, pooled], dim=0),
                "time_ids": add_time_ids}
        if controlnet is not None:
            # CN :364-371 keep window; :389-396 the ControlNet sees the B conditional latents and the conditional
            # embeds only (all 81 tokens through its default attention); :405-412 the call
            n = len(timesteps)
            keep = 1.0 - float(i / n < control_guidance_start or (i + 1) / n > control_guidance_end)
            down_residuals, mid_residual = controlnet(scheduler.scale_model_input(latents, t), t, cond, control_image,
                                                      conditioning_scale=conditioning_scale * keep)
        if down_residuals is not None:
            # The reference hands batch-B residuals to a batch-2B UNet and relies on
            # broadcasting at B == 1 (CN :405-425); for B > 1 that is cat([d, d]).
            kw["down_block_additional_residuals"] = [torch.cat([d, d], dim=0) for d in down_residuals]
            kw["mid_block_additional_residual"] = torch.cat([mid_residual, mid_residual], dim=0)
        if unet_extra is not None:
            # inpaint ref :320-321: torch.cat([latent_model_input, mask, masked_image_latents], dim=1), AFTER
            # scale_model_input; mask / masked_image_latents were doubled for CFG in prepare_mask_latents
            lat_in = torch.cat([lat_in, torch.cat([unet_extra] * 2)], dim=1)
        eps = unet(lat_in, t, encoder_hidden_states=ehs, cross_attention_kwargs={}, **kw).sample
        eps_u, eps_c = eps.chunk(2)                                                # SD :561-564
        eps = eps_u + guidance_scale * (eps_c - eps_u)
        latents = scheduler.step(eps, t, latents)                                  # SD :569
        # inpaint ref :340, CN :437: ``if num_channels_unet == 4:`` -- a 9-channel UNet (unet_extra given) is NOT blended
        # and has no image latents (``return_image_latents = num_channels_unet == 4``, inpaint ref :258, CN :319)
        if inpaint_mask is not None and unet_extra is not None:
            raise ValueError("reference semantics: no mask blend for a 9-channel UNet (inpaint ref :340)")
        if inpaint_mask is not None:                                               # inpaint :340-353, CN :437-449
            init = inpaint_init
            if i < len(timesteps) - 1:
                init = scheduler.add_noise(inpaint_init, inpaint_noise, timesteps[i + 1])
            latents = (1 - inpaint_mask) * init + inpaint_mask * latents
        if on_step is not None:
            on_step(i, t, latents)
    return latents
