"""Oracle (test infrastructure): diffusers==0.23.0 ``DDIMScheduler`` as configured by
the Stable Diffusion checkpoints the reference loads (demo/controlnet_demo.py:67).
PARITY UNPINNED (third-party, not vendored).  Call sites:
pipline_StableDiffusion_ConsistentID.py:510 (set_timesteps), :540
(scale_model_input, identity for DDIM), :569 (step);
pipelines/StableDIffusionControlNetInpaint_ConsistentID.py:443-446 (add_noise).

Config: scaled_linear betas 0.00085 -> 0.012 over 1000 train steps, epsilon
prediction, clip_sample False, set_alpha_to_one False, steps_offset 1, "leading"
timestep spacing (the Stable Diffusion configs; "linspace" / "trailing" restated too), eta 0.
"""
from __future__ import annotations

import numpy as np
import torch


def spaced_timesteps(T, n, spacing, offset):
    """``set_timesteps`` of diffusers 0.23 (both schedulers; the callers cast: DDIM to int64 after rounding, Euler to float32)"""
    if spacing == "leading":
        return (np.arange(0, n) * (T // n)).round()[::-1].copy().astype(np.float64) + offset
    if spacing == "linspace":
        return np.linspace(0, T - 1, n)[::-1].copy()
    if spacing == "trailing":
        return np.round(np.arange(T, 0, -T / n)) - 1
    raise ValueError(spacing)


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 steps_offset=1, set_alpha_to_one=False, timestep_spacing="leading"):
        self.timestep_spacing = timestep_spacing
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
        ts = spaced_timesteps(self.num_train_timesteps, num_inference_steps, self.timestep_spacing, self.steps_offset)
        self.timesteps = torch.from_numpy(ts.round().astype(np.int64))

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


class EulerDiscreteScheduler:
    """diffusers==0.23.0 ``EulerDiscreteScheduler`` as the reference's canonical scripts configure it
    (infer.py:33, infer_SDXL.py:37: ``EulerDiscreteScheduler.from_config(pipe.scheduler.config)`` -- the Stable Diffusion
    scheduler config carries scaled_linear betas, steps_offset 1 and "leading" spacing over).  PARITY UNPINNED.
    epsilon prediction, linear sigma interpolation, no Karras sigmas, s_churn = 0 (the pipelines pass no extra kwargs)."""
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 timestep_spacing="leading"):
        self.timestep_spacing = timestep_spacing
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_train_timesteps, self.steps_offset = num_train_timesteps, steps_offset
        ac = self.alphas_cumprod.numpy()
        self._train_sigmas = np.array(((1 - ac) / ac) ** 0.5)
        self.sigmas = torch.from_numpy(np.concatenate([self._train_sigmas[::-1], [0.0]]).astype(np.float32))
        self.timesteps = None

    @property
    def init_noise_sigma(self):
        if self.timestep_spacing in ("linspace", "trailing"):
            return float(self.sigmas.max())
        return float((self.sigmas.max() ** 2 + 1) ** 0.5)          # "leading" spacing branch

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ts = spaced_timesteps(self.num_train_timesteps, num_inference_steps, self.timestep_spacing,
                              self.steps_offset).astype(np.float32)
        sig = np.interp(ts, np.arange(0, len(self._train_sigmas)), self._train_sigmas)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)

    def _index(self, t):
        return int((self.timesteps == float(t)).nonzero()[0])

    def scale_model_input(self, sample, t):
        sigma = float(self.sigmas[self._index(t)])
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, eps, t, sample):
        i = self._index(t)
        sigma, nxt = float(self.sigmas[i]), float(self.sigmas[i + 1])
        x0 = sample - sigma * eps
        return sample + ((sample - x0) / sigma) * (nxt - sigma)

    def add_noise(self, original, noise, t):
        return original + noise * float(self.sigmas[self._index(t)])
