"""Oracle (test infrastructure): diffusers==0.23.0 ``DDIMScheduler`` as configured by
the Stable Diffusion checkpoints the reference loads (demo/controlnet_demo.py:67).
PARITY UNPINNED (third-party, not vendored).  Call sites:
pipline_StableDiffusion_ConsistentID.py:510 (set_timesteps), :540
(scale_model_input, identity for DDIM), :569 (step);
pipelines/StableDIffusionControlNetInpaint_ConsistentID.py:443-446 (add_noise).

Config: scaled_linear betas 0.00085 -> 0.012 over 1000 train steps, epsilon
prediction, clip_sample False, set_alpha_to_one False, steps_offset 1, "leading"
timestep spacing, eta 0.
"""
from __future__ import annotations

import numpy as np
import torch


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 steps_offset=1, set_alpha_to_one=False):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                               dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.timesteps = None
        self.num_inference_steps = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts + self.steps_offset)

    def scale_model_input(self, sample, t=None):
        return sample

    def coefficients(self, t: int):
        """(sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)) for eta = 0."""
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return (float(a_t.sqrt()), float((1 - a_t).sqrt()), float(a_p.sqrt()), float((1 - a_p).sqrt()))

    def step(self, eps, t, sample):
        sa, s1a, sp, s1p = self.coefficients(int(t))
        x0 = (sample - s1a * eps) / sa
        return sp * x0 + s1p * eps

    def add_noise(self, original, noise, t):
        a = self.alphas_cumprod[int(t)]
        return float(a.sqrt()) * original + float((1 - a).sqrt()) * noise
