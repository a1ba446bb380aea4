"""Hot-path pipelines: the denoising loops of the reference's four pipeline classes with
the reference's ``__call__`` keyword surface, running on the HIP engine.

  ConsistentIDStableDiffusionPipeline                  pipline_StableDiffusion_ConsistentID.py:33, loop :535-579
  ConsistentIDStableDiffusionXLPipeline                pipline_StableDiffusionXL_ConsistentID.py:44, loop :611-667
  StableDiffusionInpaintConsistentIDPipeline           pipelines/StableDIffusionInpaint_ConsistentID.py:94, loop :305-359
  StableDiffusionControlNetInpaintConsistentIDPipeline pipelines/StableDIffusionControlNetInpaint_ConsistentID.py:94, :375-456

Scope: the per-step path, the ControlNet encoder (row f-1) and the VAE decode (row f-2).
The once-per-image pre-loop (FaceID, face parsing, CLIP text/vision encoders, FacialEncoder /
ProjPlusModel, VAE encode) is out of scope this round, so the pipelines take what that pre-loop produces:
``prompt_embeds`` = cat([null, augmented, text_only]) of shape [3B, 77+4, Dc] exactly as the
reference assembles it before ``.chunk(3)`` (ref :494-507, :527-531), and ``latents``.
String prompts / ID images raise NotImplementedError naming the missing component instead of silently doing
something else; ``output_type`` other than "latent" needs the pipeline to be built with ``vae=HipVAEDecoder(...)``.

B > 1 is this framework's extension (the reference is effectively B = 1 per call,
SURVEY.md Appendix B): B independent samples, each with its own CFG pair.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import ops
from .scheduler import DDIMScheduler, EulerDiscreteScheduler  # noqa: F401
from .unet import HipUNet


@dataclass
class StableDiffusionPipelineOutput:
    images: Any
    nsfw_content_detected: Optional[List[bool]] = None


@dataclass
class StableDiffusionXLPipelineOutput:
    images: Any


class _DenoiseEngine:
    """One generation = set_context (K/V of the 3 embed sets) + S x [UNet(2B) + CFG + DDIM].
    The step is captured once into a hipGraph and replayed; per-step scalars live on device."""

    def __init__(self, unet: HipUNet, scheduler: DDIMScheduler, use_graph: bool = True):
        self.unet = unet
        self.scheduler = scheduler
        self.use_graph = use_graph
        self._graph = None           # truthy while the static buffers the captured graphs point at are unchanged
        self._graph_key = None
        self._graphs: Dict[bool, Any] = {}      # captured step, without / with the ControlNet forward
        self._warm_keys = set()
        self._static: Dict[str, torch.Tensor] = {}

    def _static_tensor(self, name: str, like: torch.Tensor, dtype=None) -> torch.Tensor:
        """persistent device buffer (stable address across generations -> the captured graph stays valid)"""
        dtype = dtype or like.dtype
        cur = self._static.get(name)
        if cur is None or cur.shape != like.shape or cur.dtype != dtype:
            cur = torch.empty(like.shape, dtype=dtype, device=self.unet.device)
            self._static[name] = cur
            self._graph = None   # an address changed: re-capture
        cur.copy_(like.to(device=self.unet.device, dtype=dtype))
        return cur

    @torch.no_grad()
    def run(self, latents: torch.Tensor, null_embeds, augmented_embeds, text_embeds, *, num_inference_steps: int,
            guidance_scale: float, start_merge_step: int, null_embeds_post=None, first_step: int = 0,
            pooled: Optional[Sequence[torch.Tensor]] = None, time_ids: Optional[torch.Tensor] = None,
            down_residuals=None, mid_residual=None, inpaint_mask=None, inpaint_init=None, inpaint_noise=None,
            controlnet=None, control_image=None, conditioning_scale: float = 1.0,
            control_guidance_start: float = 0.0, control_guidance_end: float = 1.0,
            callback: Optional[Callable[[int, int, torch.Tensor], None]] = None, callback_steps: int = 1,
            scale_initial: bool = True, unet_extra: Optional[torch.Tensor] = None):
        """``first_step``: the loop runs schedule entries [first_step, S) -- the inpaint pipelines' ``strength`` < 1
        window (get_timesteps, inpaint ref :246-252); the embed switch and the ControlNet keep window count steps from
        there, exactly like the reference's ``for i, t in enumerate(timesteps)`` over the truncated list.
        ``unet_extra`` [B, 5, h, w]: cat([mask, masked_image_latents]) of a 9-channel inpainting UNet (inpaint ref
        :320-321, CN :415-416) -- conv_in reads it beside the (scaled) latents, the ControlNet does not see it."""
        unet, sch = self.unet, self.scheduler
        dev = unet.device
        B = latents.shape[0]
        S = self._static_tensor
        sch.set_timesteps(num_inference_steps)                      # ref :510, before prepare_latents (:517)
        # prepare_latents (diffusers; ref :517-526) scales the initial noise by the scheduler's init_noise_sigma
        lat = S("lat", latents.to(dev).float() * (float(sch.init_noise_sigma) if scale_initial else 1.0), torch.float16)
        per_sample = lat[0].numel()
        # rows [0,B) null, [B,2B) text-only, [2B,3B) augmented   (ref :527-531 + :542-549); the SDXL pipeline has a
        # second unconditional set for the steps after the merge (ref SDXL :586-590, :620-631): rows [3B,4B)
        ctx_before = unet.context_addresses()
        sets = [null_embeds.to(dev), text_embeds.to(dev), augmented_embeds.to(dev)]
        if null_embeds_post is not None:
            sets.append(null_embeds_post.to(dev))
        unet.set_context(torch.cat(sets, dim=0))
        if unet.context_addresses() != ctx_before:
            self._graph = None
        sch.set_timesteps(num_inference_steps)
        ts = sch.timesteps
        inpaint = inpaint_mask is not None
        coefs = torch.from_numpy(sch.coefficient_table(inpaint)).to(dev)
        tvals = torch.tensor(ts.astype(np.float32), device=dev)
        ar = torch.arange(B, dtype=torch.int32, device=dev)
        kv_pre = torch.cat([ar, ar + B]).contiguous()       # i <= start_merge_step: (null, text)
        kv_post = torch.cat([ar + (3 * B if null_embeds_post is not None else 0), ar + 2 * B]).contiguous()  # afterwards
        t_buf = S("t", torch.zeros(1), torch.float32)
        coef_buf = S("coef", torch.zeros(5), torch.float32)      # c_x, c_eps, c_init, c_noise, model-input scale
        in_scale = coef_buf[4:5]
        kvrow = S("kvrow", kv_pre, torch.int32)
        added = None
        pooled_post = None
        if time_ids is not None:
            p_null, p_text, p_aug = [p.to(device=dev, dtype=torch.float16) for p in pooled]
            pooled_buf = S("pooled", torch.cat([p_null, p_text], 0), torch.float16)
            pooled_post = torch.cat([p_null, p_aug], 0).contiguous()
            added = {"text_embeds": pooled_buf, "time_ids": S("time_ids", time_ids, torch.float32)}
        mask = init = noise = None
        if inpaint:
            if unet_extra is not None:
                raise ValueError("a 9-channel inpainting UNet is not blended: the reference guards the mask blend with "
                                 "`if num_channels_unet == 4` (inpaint ref :340, CN :437)")
            if inpaint_init is None or inpaint_noise is None:
                raise ValueError("the mask blend of a 4-channel UNet needs image_latents and noise (inpaint ref :340-353)")
            mask = S("mask", inpaint_mask.to(dev).expand_as(lat), torch.float16)
            init = S("init", inpaint_init, torch.float16)
            noise = S("noise", inpaint_noise, torch.float16)
        extra = S("unet_extra", unet_extra, torch.float16) if unet_extra is not None else None
        dres = mres = None
        if down_residuals is not None:
            dres = [S(f"dres{j}", r, torch.float16) for j, r in enumerate(down_residuals)]
            mres = S("mres", mid_residual, torch.float16)
        cn_cond = cn_kvrow = None
        cn_keep = [0.0] * len(ts)
        if controlnet is not None:
            # native ControlNet (CN :389-412): conditional latents + conditional embeds, residuals recomputed per step.
            # Its K/V cache holds rows [0,B) text-only and [B,2B) augmented, selected like the UNet's.
            assert down_residuals is None, "pass either a ControlNet or precomputed residuals"
            cn_before = controlnet.context_addresses()
            controlnet.set_context(torch.cat([text_embeds.to(dev), augmented_embeds.to(dev)], dim=0), num_tokens=0)
            if controlnet.context_addresses() != cn_before:
                self._graphs.clear()
                self._warm_keys.clear()
            cn_cond = S("cn_cond", controlnet.cond_embedding(control_image), torch.float16)
            cn_kvrow = S("cn_kvrow", ar, torch.int32)
            n = len(ts) - first_step
            cn_keep = [0.0] * first_step + [1.0 - float(i / n < control_guidance_start or (i + 1) / n > control_guidance_end)
                                            for i in range(n)]                      # CN :364-371
        # time path: one table per generation instead of three weight-streaming GEMVs per step (not with SDXL's
        # text_time conditioning, whose rows also depend on the sample)
        temb_tab = cn_temb_tab = temb_buf = cn_temb_buf = None
        if unet.config.addition_embed_type is None and not os.environ.get("CID_NO_TEMB_TABLE"):
            temb_tab = unet.time_embed_table(tvals)
            temb_buf = S("temb", temb_tab[:1], torch.float16)
            if controlnet is not None:
                cn_temb_tab = controlnet.time_embed_table(tvals)
                cn_temb_buf = S("cn_temb", cn_temb_tab[:1], torch.float16)
        # every per-step host value of the reference's `for i, t in enumerate(timesteps)` as one device table: row i holds t,
        # the scheduler coefficients, the embed-set rows (ref :542-549: text-only while i <= start_merge_step), the time-
        # embedding row and SDXL's pooled embeds (ref SDXL :620-631); cid_step_select, the first launch of the captured
        # step, copies row `counter` into the buffers the step's kernels read and increments the counter
        n_ts = len(ts)
        merged_at = torch.tensor([(i - first_step) > start_merge_step for i in range(n_ts)], device=dev)
        cols = [(t_buf, tvals.view(n_ts, 1)), (coef_buf, coefs.view(n_ts, 5).float()),
                (kvrow, torch.where(merged_at[:, None], kv_post[None], kv_pre[None]))]
        if cn_kvrow is not None:
            cols.append((cn_kvrow, torch.where(merged_at[:, None], (ar + B)[None], ar[None])))
        if temb_buf is not None:
            cols.append((temb_buf, temb_tab.view(n_ts, -1)))
            if cn_temb_buf is not None:
                cols.append((cn_temb_buf, cn_temb_tab.view(n_ts, -1)))
        if pooled_post is not None:
            pre = torch.cat([p_null, p_text], 0)
            cols.append((added["text_embeds"], torch.where(merged_at[:, None, None], pooled_post[None], pre[None])))
        table = ops.StepTable(cols, dev, alloc=S)     # table + counter are static buffers too
        table.reset(first_step)

        # every static buffer exists now: a new one (S() cleared _graph) or a new configuration invalidates the captured
        # graphs AND their eager warm-up (the first step after a shape change must run eagerly again)
        key = (B, tuple(lat.shape), float(guidance_scale), inpaint, time_ids is not None, dres is not None,
               controlnet is not None, float(conditioning_scale), extra is not None)
        if key != self._graph_key or self._graph is None:
            self._graphs.clear()
            self._warm_keys.clear()
            self._graph, self._graph_key = True, key     # (_graph: "static buffers valid" marker, cleared by S())

        def step(with_cn: bool):
            table.select()
            d, m = dres, mres
            if with_cn:
                d, m = controlnet.forward_tokens(lat, t_buf, cn_kvrow, B, cn_cond, conditioning_scale, temb=cn_temb_buf,
                                                 in_scale=in_scale)
            eps = unet.forward_tokens(lat, t_buf, kvrow, 2 * B, added, d, m, temb=temb_buf, in_scale=in_scale, extra=extra)
            ops.cfg_ddim_step(eps, lat, coef_buf, guidance_scale, B=B, per_sample=per_sample,
                              mask=mask, init=init, noise=noise)

        for i in range(first_step, len(ts)):
            with_cn = controlnet is not None and cn_keep[i] > 0.0     # keep = 0: the residuals are zero (CN :397-403)
            if not self.use_graph:
                step(with_cn)
            elif with_cn not in self._warm_keys:
                step(with_cn)   # eager warm-up: configures kernels, sizes the allocator pools
                self._warm_keys.add(with_cn)
            else:
                g = self._graphs.get(with_cn)
                if g is None:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        step(with_cn)
                    self._graphs[with_cn] = g
                g.replay()
            if callback is not None and (i - first_step) % callback_steps == 0:
                callback(i - first_step, int(ts[i]), lat)
        return lat.clone()


class _BasePipeline:
    default_guidance = 5.0
    vae_scale_factor = 8

    def __init__(self, unet: HipUNet, scheduler: Optional[DDIMScheduler] = None, use_graph: bool = True,
                 num_tokens: int = 4, lora_rank: int = 128, vae=None):
        """``vae``: a ``consistentid_amd.vae.HipVAEDecoder`` -- enables every ``output_type`` besides "latent"."""
        self.unet = unet
        self.vae = vae
        self.num_tokens = num_tokens
        self.lora_rank = lora_rank
        self.device = unet.device
        self._engine = _DenoiseEngine(unet, scheduler or DDIMScheduler(), use_graph)

    def to(self, device=None, *args, **kwargs):
        """``pipe.to(device)`` of the reference scripts (infer.py:21, demo/controlnet_demo.py:60): the engines are built on
        their device by ``from_pretrained(..., device=)``; this only checks that the request names that device."""
        if device is not None and not isinstance(device, torch.dtype):
            want = torch.device(device)
            if want.type != "cuda" or (want.index is not None and want.index != (self.device.index or 0)):
                raise ValueError(f"the engine lives on {self.device} (no CPU path, weights are packed per device): "
                                 f"build the pipeline with from_pretrained(..., device={str(want)!r})")
        return self

    @property
    def scheduler(self):
        """the reference scripts replace the scheduler AFTER construction (infer.py:33
        ``pipe.scheduler = EulerDiscreteScheduler.from_config(pipe.scheduler.config)``, demo/controlnet_demo.py:67 with DDIM):
        the attribute is the denoise engine's scheduler, so the assignment takes effect on the next call (the per-step
        coefficients live in device buffers the captured step reads, the graphs stay valid)"""
        return self._engine.scheduler

    @scheduler.setter
    def scheduler(self, sch):
        if not hasattr(sch, "coefficient_table"):
            raise TypeError(f"{type(sch).__name__}: the engine takes consistentid_amd.scheduler.DDIMScheduler / "
                            "EulerDiscreteScheduler (build one with .from_config(diffusers_scheduler.config))")
        self._engine.scheduler = sch

    # -- surface kept from the reference ------------------------------------------------------
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=torch.float16, device="cuda:0", **kwargs):
        """``ConsistentIDPipeline.from_pretrained(base_model_path, torch_dtype=torch.float16)`` (infer.py:17-21): UNet and VAE
        decoder of a LOCAL diffusers model directory -> engines (loader.py); ``controlnet=`` as in demo/controlnet_demo.py:44-47.
        Follow with ``load_ConsistentID_model`` exactly like the reference."""
        from . import loader
        return loader.from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=torch_dtype, device=device, **kwargs)

    def load_ConsistentID_model(self, pretrained_model_name_or_path_or_dict, weight_name: str = "", subfolder: str = "",
                                trigger_word_ID: str = "<|image|>", trigger_word_facial: str = "<|facial|>",
                                image_encoder_path: str = "", bise_net_cp: str = "", torch_dtype=torch.float16,
                                num_tokens: int = 4, lora_rank: int = 128, **kwargs):
        """Reference: pipline_StableDiffusion_ConsistentID.py:36-150.  The ``adapter_modules`` entry of the checkpoint
        (dict, ``.bin`` or ``.safetensors`` path; local files only) is merged into the engine in place -- the UNet must
        have been built with ``keep_base=True``.  FacialEncoder / image_proj weights build the ID-conditioning engine
        (``prepare_prompt_embeds``, row f-3); CLIP / FaceAnalysis / BiSeNet construction (ref :54-69) is not done here."""
        from .checkpoint import load_checkpoint
        state_dict = load_checkpoint(pretrained_model_name_or_path_or_dict, weight_name, subfolder)
        self.lora_rank, self.num_tokens, self.torch_dtype = lora_rank, num_tokens, torch_dtype
        self.trigger_word_ID, self.trigger_word_facial = trigger_word_ID, trigger_word_facial
        self.unet.num_tokens = num_tokens
        self.unet.load_adapter_modules(state_dict["adapter_modules"])                  # ref :143-144 (strict)
        # once-per-image ID-conditioning modules (ProjPlusModel / FacialEncoder, ref :93-100, :141-142)
        self.image_proj_state = state_dict.get("image_proj")
        self.facial_encoder_state = state_dict.get("FacialEncoder")
        self.id_conditioner = None
        if self.image_proj_state and self.facial_encoder_state:
            from .idstack import HipIDConditioner
            self.id_conditioner = HipIDConditioner(self.image_proj_state, self.facial_encoder_state, device=self.device)
        self._engine._graphs.clear()
        return self

    def prepare_prompt_embeds(self, **encoder_outputs) -> torch.Tensor:
        """``prompt_embeds`` = cat([null, augmented, text_only]) from the upstream encoders' outputs (what ref :479-507
        computes between the CLIP / FaceID / text encoders and the loop): see ``idstack.HipIDConditioner.__call__`` for
        the keywords.  Needs a checkpoint with ``image_proj`` and ``FacialEncoder`` loaded (``load_ConsistentID_model``)."""
        if getattr(self, "id_conditioner", None) is None:
            raise RuntimeError("no ID-conditioning weights: load_ConsistentID_model(checkpoint with image_proj + FacialEncoder)")
        return self.id_conditioner(**encoder_outputs)

    def _check_hot_path_inputs(self, prompt, input_id_images, prompt_embeds, latents, output_type):
        if prompt is not None or input_id_images is not None:
            raise NotImplementedError(
                "the pre-loop (FaceID / face parsing / CLIP encoders / FacialEncoder, ref :437-507) is outside "
                "this round's scope (SURVEY.md 8f-3): pass prompt_embeds=[3B,81,Dc] and latents")
        if prompt_embeds is None or latents is None:
            raise ValueError("prompt_embeds (cat([null, augmented, text_only])) and latents are required")
        if output_type != "latent" and self.vae is None:
            raise ValueError("output_type other than 'latent' needs a VAE decoder: build the pipeline with "
                             "vae=HipVAEDecoder(...)")

    def _postprocess(self, latents: torch.Tensor, output_type: str, legacy_numpy: bool = False):
        """SD1.5 (ref :581-598): decode_latents -> NHWC float32 numpy in [0, 1] (-> PIL for "pil"); any other
        non-latent output_type also yields numpy there (``legacy_numpy``).  The image_processor-based pipelines
        (SDXL :676-684, inpaint) additionally know "pt" (the [B, 3, H, W] tensor in [0, 1])."""
        if output_type == "latent":
            return latents
        img = self.vae.decode_latents(latents)
        if output_type == "pt" and not legacy_numpy:
            return img
        arr = img.float().permute(0, 2, 3, 1).cpu().numpy()
        if output_type == "pil":
            from PIL import Image
            return [Image.fromarray(a) for a in (arr * 255).round().astype("uint8")]
        return arr

    def _split(self, prompt_embeds):
        assert prompt_embeds.shape[0] % 3 == 0
        return prompt_embeds.chunk(3)   # null, augmented, text-only (ref :527-531)


class ConsistentIDStableDiffusionPipeline(_BasePipeline):
    def __call__(self, prompt=None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 5.0, negative_prompt=None,
                 num_images_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None,
                 latents: Optional[torch.Tensor] = None, prompt_embeds: Optional[torch.Tensor] = None,
                 negative_prompt_embeds=None, output_type: Optional[str] = "pil", return_dict: bool = True,
                 cross_attention_kwargs=None, original_size=None, target_size=None, callback=None,
                 callback_steps: int = 1, input_id_images=None, start_merge_step: int = 0,
                 class_tokens_mask=None, prompt_embeds_text_only=None):
        self._check_hot_path_inputs(prompt, input_id_images, prompt_embeds, latents, output_type)
        assert guidance_scale >= 1.0, "the reference asserts classifier-free guidance (ref :434,:441)"
        assert eta == 0.0, "DDIM eta = 0 only"
        null_e, aug_e, text_e = self._split(prompt_embeds)
        out = self._engine.run(latents, null_e, aug_e, text_e, num_inference_steps=num_inference_steps,
                               guidance_scale=guidance_scale, start_merge_step=start_merge_step,
                               callback=callback, callback_steps=callback_steps)
        out = self._postprocess(out, output_type, legacy_numpy=True)   # no safety checker: has_nsfw_concept = None
        if not return_dict:
            return (out, None)
        return StableDiffusionPipelineOutput(images=out, nsfw_content_detected=None)


class ConsistentIDStableDiffusionXLPipeline(_BasePipeline):
    default_guidance = 7.5

    def __call__(self, prompt=None, prompt_2=None, height=None, width=None, num_inference_steps: int = 50,
                 denoising_end=None, guidance_scale: float = 7.5, negative_prompt=None, negative_prompt_2=None,
                 num_images_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None, latents=None,
                 prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None,
                 negative_pooled_prompt_embeds=None, output_type: Optional[str] = "pil", return_dict: bool = True,
                 callback=None, callback_steps: int = 1, cross_attention_kwargs=None, guidance_rescale: float = 0.0,
                 original_size=None, crops_coords_top_left=(0, 0), target_size=None, input_id_images=None,
                 start_merge_step: int = 0, class_tokens_mask=None, prompt_embeds_text_only=None,
                 pooled_prompt_embeds_text_only=None, add_time_ids: Optional[torch.Tensor] = None,
                 negative_prompt_embeds_facial: Optional[torch.Tensor] = None):
        """pooled_prompt_embeds = pooled embeds used AFTER the merge step, pooled_prompt_embeds_text_only
        BEFORE it, negative_pooled_prompt_embeds for the unconditional half (ref SDXL :620-631);
        add_time_ids [2B, 6] (ref :531-539)."""
        self._check_hot_path_inputs(prompt, input_id_images, prompt_embeds, latents, output_type)
        assert guidance_scale >= 1.0 and eta == 0.0
        # The SDXL loop has TWO unconditional sets (ref SDXL :586-590, :620-631): cat([negative text embeds, uncond ID
        # tokens]) up to start_merge_step, cat([FacialEncoder(negative embeds), uncond ID tokens]) afterwards.
        # prompt_embeds = cat([null_text_only, augmented, text_only, null_facial]) (4B rows); with 3B rows the one null
        # serves both phases (the SD1.5 convention).  negative_prompt_embeds_facial overrides / supplies the second one.
        null_post = negative_prompt_embeds_facial
        if prompt_embeds.shape[0] % 4 == 0 and prompt_embeds.shape[0] // 4 == latents.shape[0]:
            null_e, aug_e, text_e, null_post4 = prompt_embeds.chunk(4)
            null_post = null_post if null_post is not None else null_post4
        else:
            null_e, aug_e, text_e = self._split(prompt_embeds)
        if negative_prompt_embeds is not None:
            # ref SDXL :586-590: negative_prompt_embeds_text_only = cat([negative_prompt_embeds, uncond_prompt_tokens_faceid],
            # dim=1) is the unconditional set up to start_merge_step.  Raw [B, 77, Dc] negative embeds get the unconditional
            # ID tokens appended here -- the trailing num_tokens rows of the null set in prompt_embeds ARE those tokens
            # (ref :582-583: every unconditional set ends with uncond_prompt_tokens_faceid); [B, 77 + 4, Dc] is taken as is.
            neg = negative_prompt_embeds.to(null_e.device, null_e.dtype)
            nt = self.num_tokens
            if neg.shape[1] == null_e.shape[1] - nt:
                neg = torch.cat([neg, null_e[:, -nt:]], dim=1)
            if neg.shape != null_e.shape:
                raise ValueError(f"negative_prompt_embeds {tuple(negative_prompt_embeds.shape)}: expected [B, {null_e.shape[1] - nt}"
                                 f" or {null_e.shape[1]}, {null_e.shape[2]}]")
            if null_post is None:
                null_post = null_e      # after the merge the reference keeps FacialEncoder(negative): the assembled null set
            null_e = neg
        if add_time_ids is None:
            H, W = latents.shape[-2] * 8, latents.shape[-1] * 8
            add_time_ids = torch.tensor([[H, W, 0, 0, H, W]], dtype=torch.float32).repeat(2 * latents.shape[0], 1)
        out = self._engine.run(latents, null_e, aug_e, text_e, num_inference_steps=num_inference_steps,
                               guidance_scale=guidance_scale, start_merge_step=start_merge_step, null_embeds_post=null_post,
                               pooled=(negative_pooled_prompt_embeds, pooled_prompt_embeds_text_only,
                                       pooled_prompt_embeds), time_ids=add_time_ids,
                               callback=callback, callback_steps=callback_steps)
        out = self._postprocess(out, output_type)
        if not return_dict:
            return (out,)
        return StableDiffusionXLPipelineOutput(images=out)


class StableDiffusionInpaintConsistentIDPipeline(_BasePipeline):
    default_guidance = 7.5

    def _strength_window(self, strength: float, num_inference_steps: int, latents, image_latents, noise):
        """get_timesteps + prepare_latents of the inpaint pipelines (inpaint ref :246-252, :258-275; diffusers 0.23):
        the loop runs the LAST int(S * strength) schedule entries; user-supplied ``latents`` are the initial noise
        (x init_noise_sigma) whatever the strength, without them the start is pure noise at strength 1 and
        add_noise(image_latents, noise, first timestep) below it.  Returns (first_step, initial latents, scale flag)."""
        if not 0.0 < strength <= 1.0:
            raise ValueError(f"strength must be in (0, 1], got {strength}")
        S = num_inference_steps
        first = max(S - min(int(S * strength), S), 0)
        if first >= S:      # diffusers: "After adjusting the num_inference_steps by strength parameter: ... < 1"
            raise ValueError(f"strength {strength} with {S} inference steps leaves no denoising step")
        if latents is not None:
            return first, latents, True
        if noise is None or image_latents is None:
            raise ValueError("without latents the inpaint pipelines need image_latents and noise")
        if strength == 1.0:
            return first, noise, True
        self.scheduler.set_timesteps(S)
        ca, cn_ = self.scheduler.add_noise_coefficients(self.scheduler.timesteps[first])
        return first, ca * image_latents.float() + cn_ * noise.float(), False


    def _unet_extra(self, latents, mask_latents, masked_image_latents):
        """9-channel inpainting UNets (``unet.config.in_channels == 9``): the per-step
        ``torch.cat([latent_model_input, mask, masked_image_latents], dim=1)`` (inpaint ref :320-321, CN :415-416).  Returns
        cat([mask, masked_image_latents]) [B, 5, h, w] for conv_in's second source, None for 4-channel UNets (which ignore
        masked_image_latents like the reference does)."""
        cin = getattr(self.unet.config, "in_channels", 4)
        if cin == 4:
            return None
        if cin != 9:
            raise ValueError(f"inpainting UNets have 4 or 9 input channels, this one has {cin}")
        if mask_latents is None or masked_image_latents is None:
            raise ValueError("a 9-channel inpainting UNet needs mask_latents [B,1,h,w] and masked_image_latents [B,4,h,w]")
        B = latents.shape[0]
        m = mask_latents.to(self.device, torch.float16).expand(B, 1, *latents.shape[-2:])
        mi = masked_image_latents.to(self.device, torch.float16).expand(B, -1, -1, -1)
        if m.shape[1] + mi.shape[1] + latents.shape[1] != cin:    # the reference's check (inpaint ref :285-293)
            raise ValueError(f"latents {latents.shape[1]} + mask {m.shape[1]} + masked image {mi.shape[1]} channels != {cin}")
        return torch.cat([m, mi], dim=1).contiguous()

    @staticmethod
    def _blend_inputs(extra, mask_latents, image_latents, noise):
        """The per-step ``latents = (1 - mask) * noised_init + mask * latents`` exists only for 4-channel UNets:
        ``if num_channels_unet == 4`` (inpaint ref :340-353, CN :437-449) and ``return_image_latents = num_channels_unet
        == 4`` (inpaint ref :258, CN :319) -- a 9-channel UNet sees the mask through its input channels and its loop
        neither blends nor has image latents.  Returns the engine's (inpaint_mask, inpaint_init, inpaint_noise)."""
        if extra is not None or mask_latents is None:
            return None, None, None
        if image_latents is None or noise is None:
            raise ValueError("inpainting with a 4-channel UNet blends every step (inpaint ref :340-353): image_latents and "
                             "noise are required beside mask_latents")
        return mask_latents, image_latents, noise

    def __call__(self, prompt=None, image=None, mask_image=None, masked_image_latents=None, height=None, width=None,
                 strength: float = 1.0, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 negative_prompt=None, num_images_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None,
                 latents=None, prompt_embeds=None, negative_prompt_embeds=None, output_type: Optional[str] = "pil",
                 return_dict: bool = True, callback=None, callback_steps: int = 1, cross_attention_kwargs=None,
                 input_id_images=None, start_merge_step: int = 0, class_tokens_mask=None,
                 prompt_embeds_text_only=None, image_latents: Optional[torch.Tensor] = None,
                 noise: Optional[torch.Tensor] = None, mask_latents: Optional[torch.Tensor] = None,
                 down_block_res_samples=None, mid_block_res_sample=None):
        """Hot-path inputs replace the image pre-processing / VAE encode of ref :255-352:
        ``image_latents`` (init latents), ``noise`` and ``mask_latents`` [B,1,h,w] (1 = repaint)."""
        first, latents, scaled = self._strength_window(strength, num_inference_steps, latents, image_latents, noise)
        self._check_hot_path_inputs(prompt, input_id_images, prompt_embeds, latents, output_type)
        extra = self._unet_extra(latents, mask_latents, masked_image_latents)
        null_e, aug_e, text_e = self._split(prompt_embeds)
        b_mask, b_init, b_noise = self._blend_inputs(extra, mask_latents, image_latents, noise)
        out = self._engine.run(latents, null_e, aug_e, text_e, num_inference_steps=num_inference_steps,
                               guidance_scale=guidance_scale, start_merge_step=start_merge_step,
                               down_residuals=down_block_res_samples, mid_residual=mid_block_res_sample,
                               inpaint_mask=b_mask, inpaint_init=b_init, inpaint_noise=b_noise,
                               callback=callback, callback_steps=callback_steps, first_step=first, scale_initial=scaled,
                               unet_extra=extra)
        out = self._postprocess(out, output_type)
        if not return_dict:
            return (out, None)
        return StableDiffusionPipelineOutput(images=out, nsfw_content_detected=None)


class StableDiffusionControlNetInpaintConsistentIDPipeline(StableDiffusionInpaintConsistentIDPipeline):
    """ControlNet-inpaint loop (pipelines/StableDIffusionControlNetInpaint_ConsistentID.py:375-456).

    With a ``controlnet`` (``consistentid_amd.controlnet.HipControlNet``) and a ``control_image`` [B, 3, 8h, 8w] the
    ControlNet encoder runs natively inside every captured step: on the B conditional latents, with the conditional
    embeds of the step (text-only up to ``start_merge_step``, augmented afterwards) seen through default attention
    (CN :389-396, :405-412), residuals scaled by ``controlnet_conditioning_scale`` x the keep window (CN :364-371).
    Precomputed residuals (``down_block_res_samples`` token-major [B, HW_i, C_i] x 12, ``mid_block_res_sample``) are
    still accepted instead.  Either way the reference adds batch-B residuals to the batch-2B UNet by broadcasting at
    B = 1 (CN :418-425) -- the SAME residual for the uncond and cond halves; cid_add_inplace_f16 reproduces that as
    y[i] += a[i mod len(a)]."""

    def __init__(self, unet: HipUNet, controlnet=None, scheduler: Optional[DDIMScheduler] = None, **kw):
        super().__init__(unet, scheduler, **kw)
        self.controlnet = controlnet

    def __call__(self, prompt=None, image=None, mask_image=None, control_image=None, height=None, width=None,
                 strength: float = 1.0, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 negative_prompt=None, num_images_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None,
                 latents=None, prompt_embeds=None, negative_prompt_embeds=None, output_type: Optional[str] = "pil",
                 return_dict: bool = True, cross_attention_kwargs=None, original_size=None, target_size=None,
                 callback=None, callback_steps: int = 1,
                 controlnet_conditioning_scale: Union[float, List[float]] = 0.5, guess_mode: bool = False,
                 control_guidance_start: Union[float, List[float]] = 0.0,
                 control_guidance_end: Union[float, List[float]] = 1.0,
                 input_id_images=None, start_merge_step: int = 0, class_tokens_mask=None, prompt_embeds_text_only=None,
                 image_latents: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
                 mask_latents: Optional[torch.Tensor] = None, down_block_res_samples=None, mid_block_res_sample=None,
                 masked_image_latents: Optional[torch.Tensor] = None):
        first_step, latents, scaled = self._strength_window(strength, num_inference_steps, latents, image_latents, noise)
        self._check_hot_path_inputs(prompt, input_id_images, prompt_embeds, latents, output_type)
        extra = self._unet_extra(latents, mask_latents, masked_image_latents)
        first = lambda v: v[0] if isinstance(v, (list, tuple)) else v        # single ControlNet (CN :352-358, :399-402)
        scale, g0, g1 = first(controlnet_conditioning_scale), first(control_guidance_start), first(control_guidance_end)
        cn = None
        if control_image is not None:
            if self.controlnet is None:
                raise ValueError("control_image given but the pipeline was built without a controlnet")
            if down_block_res_samples is not None:
                raise ValueError("pass either control_image (native ControlNet) or precomputed residuals")
            if not torch.is_tensor(control_image):
                raise NotImplementedError("PIL / numpy control images (prepare_control_image, CN :267-279, is image "
                                          "pre-processing): pass a float tensor [B, 3, 8h, 8w] in [0, 1]")
            cn = self.controlnet
        null_e, aug_e, text_e = self._split(prompt_embeds)
        b_mask, b_init, b_noise = self._blend_inputs(extra, mask_latents, image_latents, noise)
        out = self._engine.run(latents, null_e, aug_e, text_e, num_inference_steps=num_inference_steps,
                               guidance_scale=guidance_scale, start_merge_step=start_merge_step,
                               down_residuals=down_block_res_samples, mid_residual=mid_block_res_sample,
                               inpaint_mask=b_mask, inpaint_init=b_init, inpaint_noise=b_noise,
                               controlnet=cn, control_image=control_image, conditioning_scale=float(scale),
                               control_guidance_start=float(g0), control_guidance_end=float(g1),
                               callback=callback, callback_steps=callback_steps, first_step=first_step, scale_initial=scaled,
                               unet_extra=extra)
        out = self._postprocess(out, output_type)
        if not return_dict:
            return (out, None)
        return StableDiffusionPipelineOutput(images=out, nsfw_content_detected=None)
