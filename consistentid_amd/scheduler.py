"""Host side of the DDIM update used by the reference demos (demo/controlnet_demo.py:67;
loop call sites pipline_StableDiffusion_ConsistentID.py:510,540,569).  Stable Diffusion
scheduler config: scaled_linear betas 0.00085..0.012, 1000 train steps, epsilon prediction,
steps_offset 1, set_alpha_to_one False, eta 0; timestep spacing "leading" (the base models' configs) by default,
"linspace" / "trailing" on request; ``from_config`` takes a diffusers scheduler config.

Only tiny tables are computed here (float64 on the host, once); the per-element update runs
in cid_cfg_ddim_step_f16 which reads the coefficients from device memory."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

SPACINGS = ("leading", "linspace", "trailing")


def _config_args(config, accepted) -> dict:
    """the arguments of a diffusers scheduler config (dict or FrozenDict-like) that this engine scheduler takes; refuses
    what the coefficient tables do not implement instead of silently computing something else"""
    cfg = dict(config)
    if cfg.get("beta_schedule", "scaled_linear") != "scaled_linear" or cfg.get("trained_betas") is not None:
        raise NotImplementedError(f"beta_schedule {cfg.get('beta_schedule')!r} / trained_betas: only scaled_linear is built")
    if cfg.get("prediction_type", "epsilon") != "epsilon":
        raise NotImplementedError(f"prediction_type {cfg.get('prediction_type')!r}: the step kernel implements epsilon prediction")
    if cfg.get("use_karras_sigmas") or cfg.get("interpolation_type", "linear") != "linear" or cfg.get("clip_sample") \
            or cfg.get("thresholding") or cfg.get("rescale_betas_zero_snr"):
        raise NotImplementedError("karras sigmas / log-linear interpolation / clip_sample / thresholding / zero-SNR betas are not built")
    return {k: cfg[k] for k in accepted if k in cfg and cfg[k] is not None}


def _spaced_timesteps(T: int, n: int, spacing: str, offset: int) -> np.ndarray:
    """diffusers 0.23 ``set_timesteps`` (float64, descending): "leading" = multiples of T // n plus steps_offset,
    "linspace" = T - 1 ... 0 evenly, "trailing" = T - 1 downwards in steps of T / n"""
    if spacing == "leading":
        return (np.arange(0, n) * (T // n)).round()[::-1].astype(np.float64) + offset
    if spacing == "linspace":
        return np.linspace(0, T - 1, n)[::-1].astype(np.float64)
    if spacing == "trailing":
        return np.round(np.arange(T, 0, -T / n)).astype(np.float64) - 1
    raise ValueError(f"timestep_spacing {spacing!r}: one of {SPACINGS}")


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 steps_offset: int = 1, set_alpha_to_one: bool = False, timestep_spacing: str = "leading"):
        if timestep_spacing not in SPACINGS:
            raise ValueError(f"timestep_spacing {timestep_spacing!r}: one of {SPACINGS}")
        betas = np.linspace(np.float32(beta_start) ** 0.5, np.float32(beta_end) ** 0.5, num_train_timesteps,
                            dtype=np.float32) ** 2
        self.alphas_cumprod = np.cumprod((1.0 - betas).astype(np.float32), dtype=np.float32)
        self.final_alpha_cumprod = np.float32(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset, self.timestep_spacing = steps_offset, timestep_spacing
        self.timesteps: np.ndarray = np.zeros(0, dtype=np.int64)
        self.num_inference_steps = 0
        # what ``OtherScheduler.from_config(pipe.scheduler.config)`` of the reference scripts reads
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule="scaled_linear", trained_betas=None, steps_offset=steps_offset,
                           set_alpha_to_one=set_alpha_to_one, timestep_spacing=timestep_spacing, prediction_type="epsilon",
                           clip_sample=False)

    @classmethod
    def from_config(cls, config) -> "DDIMScheduler":
        """``DDIMScheduler.from_config(pipe.scheduler.config)`` (demo/controlnet_demo.py:67): a diffusers scheduler config
        (the dict of ``scheduler/scheduler_config.json``, or ``scheduler.config``) -> the engine's coefficient tables"""
        return cls(**_config_args(config, ("num_train_timesteps", "beta_start", "beta_end", "steps_offset", "set_alpha_to_one",
                                           "timestep_spacing")))

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ts = _spaced_timesteps(self.num_train_timesteps, num_inference_steps, self.timestep_spacing, self.steps_offset)
        self.timesteps = ts.round().astype(np.int64)

    def scale_model_input(self, sample, t=None):
        return sample

    def alphas(self, t: int) -> Tuple[float, float]:
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_p

    def step_coefficients(self, t: int) -> Tuple[float, float]:
        """x_prev = c_x * x + c_eps * eps  (eta = 0)."""
        a_t, a_p = self.alphas(t)
        c_x = (a_p / a_t) ** 0.5
        c_eps = (1.0 - a_p) ** 0.5 - (a_p ** 0.5) * ((1.0 - a_t) ** 0.5) / (a_t ** 0.5)
        return c_x, c_eps

    def add_noise_coefficients(self, t) -> Tuple[float, float]:
        a = float(self.alphas_cumprod[int(t)])
        return a ** 0.5, (1.0 - a) ** 0.5

    def coefficient_table(self, inpaint: bool = False) -> np.ndarray:
        """[steps, 4] fp32: c_x, c_eps, c_init, c_noise (last two for the inpaint blend of the
        NEXT timestep, CN :437-449; identity on the final step)."""
        rows: List[List[float]] = []
        ts = self.timesteps
        for i, t in enumerate(ts):
            c_x, c_e = self.step_coefficients(int(t))
            ci, cn = 1.0, 0.0
            if inpaint and i < len(ts) - 1:
                ci, cn = self.add_noise_coefficients(int(ts[i + 1]))
            rows.append([c_x, c_e, ci, cn, 1.0])          # last column: model-input scale (identity for DDIM)
        return np.asarray(rows, dtype=np.float32)


class EulerDiscreteScheduler:
    """The scheduler of the reference's canonical scripts (infer.py:33, infer_SDXL.py:37:
    ``EulerDiscreteScheduler.from_config(pipe.scheduler.config)``; SD config: scaled_linear betas, steps_offset 1, "leading"
    spacing; epsilon prediction, linear sigma interpolation, s_churn 0).  In the engine's terms it is the same per-element
    update as DDIM with other coefficients -- x_prev = x + (sigma_next - sigma) * eps -- plus a model-input scale
    1 / sqrt(sigma^2 + 1) (applied inside conv_in) and an initial latent scale ``init_noise_sigma``."""
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 steps_offset: int = 1, timestep_spacing: str = "leading"):
        # "leading" is what ``EulerDiscreteScheduler.from_config(pipe.scheduler.config)`` inherits from the base models'
        # PNDM / DDIM / Euler scheduler configs (a loaded scheduler's config carries every init argument, defaults included);
        # diffusers' own class default would be "linspace" -- pass the base model's config to from_config to be sure
        if timestep_spacing not in SPACINGS:
            raise ValueError(f"timestep_spacing {timestep_spacing!r}: one of {SPACINGS}")
        betas = np.linspace(np.float32(beta_start) ** 0.5, np.float32(beta_end) ** 0.5, num_train_timesteps,
                            dtype=np.float32) ** 2
        ac = np.cumprod((1.0 - betas).astype(np.float32), dtype=np.float32)
        self._train_sigmas = np.array(((1 - ac) / ac) ** 0.5)
        self.num_train_timesteps, self.steps_offset, self.timestep_spacing = num_train_timesteps, steps_offset, timestep_spacing
        self.sigmas = np.concatenate([self._train_sigmas[::-1], [0.0]]).astype(np.float32)
        self.timesteps: np.ndarray = np.zeros(0, dtype=np.float32)
        self.num_inference_steps = 0
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule="scaled_linear", trained_betas=None, steps_offset=steps_offset,
                           timestep_spacing=timestep_spacing, prediction_type="epsilon", interpolation_type="linear",
                           use_karras_sigmas=False)

    @classmethod
    def from_config(cls, config) -> "EulerDiscreteScheduler":
        """``EulerDiscreteScheduler.from_config(pipe.scheduler.config)`` (infer.py:33, infer_SDXL.py:37).  A raw
        ``scheduler_config.json`` without a ``timestep_spacing`` key belongs to a scheduler class whose default is
        "leading" (PNDM, DDIM): that is what the loaded scheduler's config would hand over, so it is the fallback here."""
        return cls(**_config_args(config, ("num_train_timesteps", "beta_start", "beta_end", "steps_offset", "timestep_spacing")))

    @property
    def init_noise_sigma(self) -> float:
        """diffusers 0.23: max sigma of the CURRENT table for "linspace" / "trailing" spacing, sqrt(max sigma^2 + 1) for
        "leading".  The pipelines call set_timesteps before prepare_latents (ref :510 then :517), so this is the first
        inference sigma, not the training maximum"""
        m = float(self.sigmas.max())
        return m if self.timestep_spacing in ("linspace", "trailing") else float((m ** 2 + 1) ** 0.5)

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ts = _spaced_timesteps(self.num_train_timesteps, num_inference_steps, self.timestep_spacing,
                               self.steps_offset).astype(np.float32)
        sig = np.interp(ts, np.arange(0, len(self._train_sigmas)), self._train_sigmas)
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.timesteps = ts

    def add_noise_coefficients(self, t: float):
        """add_noise(original, noise, t) = original + sigma(t) * noise (t: one of the current inference timesteps)"""
        i = int(np.argmin(np.abs(self.timesteps - float(t))))
        return 1.0, float(self.sigmas[i])

    def coefficient_table(self, inpaint: bool = False) -> np.ndarray:
        """[steps, 5] fp32: c_x, c_eps, c_init, c_noise, c_in (see DDIMScheduler.coefficient_table)"""
        rows: List[List[float]] = []
        for i in range(len(self.timesteps)):
            s, nxt = float(self.sigmas[i]), float(self.sigmas[i + 1])
            ci, cn = 1.0, 0.0
            if inpaint and i < len(self.timesteps) - 1:
                ci, cn = 1.0, nxt                           # add_noise at the NEXT timestep: init + sigma_next * noise
            rows.append([1.0, nxt - s, ci, cn, 1.0 / (s * s + 1.0) ** 0.5])
        return np.asarray(rows, dtype=np.float32)
