"""HIP execution engine for the ControlNet encoder.

Drop-in for the object the reference's ControlNet-inpaint pipeline calls as
``self.controlnet(control_model_input, t, encoder_hidden_states=controlnet_prompt_embeds,
controlnet_cond=control_image, conditioning_scale=cond_scale, return_dict=False)``
(pipelines/StableDIffusionControlNetInpaint_ConsistentID.py:405-412; the model is diffusers'
``ControlNetModel`` loaded at demo/controlnet_demo.py:44-47).

It is the UNet's encoder half with diffusers' DEFAULT attention (the reference installs the
ConsistentID processors on the UNet only): all 81 context tokens are keys of ONE softmax, which the
fused cross-attention kernel runs as "n_txt = 81, n_ip = 0".  Everything reuses the UNet engine's
kernels; new here are the condition embedding (small-channel direct convs, computed once per control
image -- it does not depend on the latents or the timestep) and the 1x1 "zero convs" (plain GEMMs).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .unet import HipUNet
from .unet_spec import UNetConfig


class HipControlNet(HipUNet):
    def __init__(self, cfg: UNetConfig, controlnet_sd: Optional[Dict[str, torch.Tensor]] = None, device="cuda:0",
                 packed=None):
        super().__init__(cfg, controlnet_sd, None, device, num_tokens=0, packed=packed, encoder_only=True)
        self._cond_key = None
        self._cond_ref = None
        self._cond_emb: Optional[torch.Tensor] = None
        self._scaled: Dict[float, Dict[str, torch.Tensor]] = {}

    # ------------------------------------------------------------------ condition embedding (once per image)
    def cond_embedding(self, controlnet_cond: torch.Tensor) -> torch.Tensor:
        """[B, 3, 8h, 8w] image -> token-major [B * h * w, C0]; cached until the image tensor changes."""
        # (the keyed tensor is kept alive in _cond_ref so that its address cannot be recycled for another image)
        key = (controlnet_cond.data_ptr(), controlnet_cond._version, tuple(controlnet_cond.shape))
        if key == self._cond_key and self._cond_ref is controlnet_cond:
            return self._cond_emb
        W = self.W
        img = controlnet_cond.to(device=self.device, dtype=torch.float16)
        B, cin, H, Wd = img.shape
        x = img.permute(0, 2, 3, 1).reshape(B * H * Wd, cin).contiguous()     # token-major, once per generation
        for name, ci, co, stride, silu in self.packed.cond_convs:
            assert ci == cin, (name, ci, cin)
            Ho, Wo = (H - 1) // stride + 1, (Wd - 1) // stride + 1
            y = self._empty(B * Ho * Wo, co)
            ops.conv3x3_small(x, y, W[f"{name}.w"], W[f"{name}.b"], B=B, Hi=H, Wi=Wd, cin=ci, cout=co,
                              stride=stride, silu=silu)
            x, cin, H, Wd = y, co, Ho, Wo
        self._cond_key, self._cond_emb, self._cond_ref = key, x, controlnet_cond
        self._cond_hw = (H, Wd)
        return x

    def _zero_weights(self, scale: float) -> Dict[str, torch.Tensor]:
        """zero-conv weights with ``conditioning_scale`` folded in (the reference multiplies every residual by it,
        diffusers controlnet.py; one rounding instead of two)"""
        if scale == 1.0:
            return self.W
        if scale not in self._scaled:
            names = [f"controlnet_down_blocks.{i}" for i in range(self.packed.n_zero)] + ["controlnet_mid_block"]
            self._scaled[scale] = {f"{n}.{s}": (self.W[f"{n}.{s}"].float() * scale).half() for n in names for s in "wb"}
        return self._scaled[scale]

    # ------------------------------------------------------------------ forward
    def forward_tokens(self, sample: torch.Tensor, t_dev: torch.Tensor, kvrow: torch.Tensor, B: int,
                       cond_emb: torch.Tensor, conditioning_scale: float = 1.0, temb: Optional[torch.Tensor] = None,
                       in_scale: Optional[torch.Tensor] = None) -> Tuple[List[torch.Tensor], torch.Tensor]:
        """sample [B, 4, h, w] fp16 NCHW; ``cond_emb`` from :meth:`cond_embedding` (B or 1 images).
        Returns the 12 (+1) residuals token-major ``[B * HW_i, C_i]`` -- the layout
        ``HipUNet.forward_tokens(down_residuals=..., mid_residual=...)`` consumes."""
        cfg, W = self.config, self.W
        Bin, cin, H, Wd = sample.shape
        if temb is None:
            temb = self.time_embed(t_dev, B, None)
        trows = temb.shape[0]
        c0 = cfg.block_out_channels[0]
        x = self._empty(B * H * Wd, c0)
        ops.conv_in(sample, x, W["conv_in.w"], W["conv_in.b"], B=B, Bin=Bin, cin=cin, H=H, W=Wd, cout=c0, in_scale=in_scale)
        assert cond_emb.shape[1] == c0 and (B * H * Wd) % cond_emb.shape[0] == 0, "control image must be 8x the latent size"
        ops.add_inplace(x, cond_emb)                        # sample = conv_in(sample) + cond_embedding(cond)
        skips = [(x, c0, H, Wd)]
        c = c0
        for blk in self.downs:
            for j, r in enumerate(blk.resnets):
                x = self._resnet(r, x, None, c, 0, B, H, Wd, temb, trows)
                c = r.cout
                if blk.attentions:
                    x = self._transformer(blk.attentions[j], x, B, H, Wd, kvrow)
                skips.append((x, c, H, Wd))
            if blk.sampler:
                n = f"{blk.name}.{blk.sampler}.conv"
                Ho, Wo = H // 2, Wd // 2
                y = self._empty(B * Ho * Wo, c)
                ops.gemm(x, W[f"{n}.w"], y, M=B * Ho * Wo, N=c, c1=c, bias=W[f"{n}.b"], taps=9,
                         Hi=H, Wi=Wd, Ho=Ho, Wo=Wo, stride=2, ws=self._gemm_ws)
                x, H, Wd = y, Ho, Wo
                skips.append((x, c, H, Wd))
        x = self._resnet(self.mid.resnets[0], x, None, c, 0, B, H, Wd, temb, trows)
        x = self._transformer(self.mid.attentions[0], x, B, H, Wd, kvrow)
        x = self._resnet(self.mid.resnets[1], x, None, c, 0, B, H, Wd, temb, trows)
        Z = self._zero_weights(float(conditioning_scale))
        assert len(skips) == self.packed.n_zero
        down = []
        for i, (s, sc_, sh, sw) in enumerate(skips):
            o = self._empty(B * sh * sw, sc_)
            n = f"controlnet_down_blocks.{i}"
            ops.gemm(s, Z[f"{n}.w"], o, M=B * sh * sw, N=sc_, c1=sc_, bias=Z[f"{n}.b"], ws=self._gemm_ws)
            down.append(o)
        mid = self._empty(B * H * Wd, c)
        ops.gemm(x, Z["controlnet_mid_block.w"], mid, M=B * H * Wd, N=c, c1=c, bias=Z["controlnet_mid_block.b"],
                 ws=self._gemm_ws)
        self._last_shapes = [(sc_, sh, sw) for (_, sc_, sh, sw) in skips] + [(c, H, Wd)]
        return down, mid

    # ------------------------------------------------------------------ diffusers-style call
    @torch.no_grad()
    def __call__(self, sample, timestep, encoder_hidden_states=None, controlnet_cond=None, conditioning_scale=1.0,
                 guess_mode: bool = False, return_dict: bool = False):
        """Returns ``(down_block_res_samples, mid_block_res_sample)`` as [B, C, H, W]-shaped (channels-last) views,
        which ``HipUNet.__call__`` takes back without a copy."""
        if guess_mode:
            raise NotImplementedError("guess_mode residual scaling (the reference never forwards it, CN :405-412)")
        if return_dict:
            raise NotImplementedError("return_dict=True (the reference passes return_dict=False, CN :411)")
        if isinstance(conditioning_scale, (list, tuple)):
            raise NotImplementedError("MultiControlNet")
        sample = sample.to(device=self.device, dtype=torch.float16).contiguous()
        B = sample.shape[0]
        ehs = encoder_hidden_states
        key = (ehs.data_ptr(), ehs._version, tuple(ehs.shape))
        if self._ctx.key != key or self._ctx.key_ref is not ehs:
            self.set_context(ehs, num_tokens=0)
            self._ctx.key, self._ctx.key_ref = key, ehs
        kvrow = torch.arange(B, dtype=torch.int32, device=self.device)
        self._t_buf.fill_(float(timestep))
        cond = self.cond_embedding(controlnet_cond)
        down, mid = self.forward_tokens(sample, self._t_buf, kvrow, B, cond, conditioning_scale)

        def nchw(t, shp):
            c, h, w = shp
            return t.view(B, h, w, c).permute(0, 3, 1, 2)

        shapes = self._last_shapes
        return [nchw(t, s) for t, s in zip(down, shapes[:-1])], nchw(mid, shapes[-1])
