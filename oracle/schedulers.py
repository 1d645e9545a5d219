"""DDIM / DDPM reverse-step arithmetic, restated from the published algorithms.
TEST INFRASTRUCTURE: see oracle/__init__.py.

The reference delegates this arithmetic to the third-party package `diffusers`
(`requirements.txt:4`, UNPINNED; call sites
grasp_ldm/models/diffusion/gaussian_diffusion.py:112,122,146-160,177,272).
`diffusers` is not installed in this image and not vendored in the reference,
so parity at this boundary is UNPINNED by the reference; what is restated here:

* schedule tables (f32): betas = linspace(beta_start, beta_end, T),
  alphas_cumprod = cumprod(1 - betas)                       ["linear"]
* DDIM (Song et al. 2021, eq. 12) with eta = 0, clip_sample = True (range 1),
  set_alpha_to_one = True, steps_offset = 0, "leading" spacing,
  use_clipped_model_output = False
* DDPM (Ho et al. 2020, eq. 7) posterior mean from the clipped x0 prediction,
  variance_type "fixed_large" = 1 - abar_t/abar_prev (current beta), noise iff
  t > 0, drawn with torch.randn on the sample's device from the global RNG

Scalar coefficients are computed with 0-dim f32 torch tensors exactly like the
library does (alphas_cumprod[t] ** 0.5 etc.), so that per-step coefficients are
bit-identical to an f32 evaluation on any device.

These classes also serve as the `diffusers` shim when the reference's Python is
imported in the build container (oracle/ref_import.py).
"""
from types import SimpleNamespace

import torch


def _betas(num_train_timesteps, beta_start, beta_end, beta_schedule):
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    raise NotImplementedError(beta_schedule)


class _SchedulerBase:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02,
                 beta_schedule="linear", prediction_type="epsilon", clip_sample=True,
                 clip_sample_range=1.0):
        if prediction_type != "epsilon":
            raise NotImplementedError("only epsilon prediction is on the hot path")
        self.num_train_timesteps = int(num_train_timesteps)
        self.betas = _betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.clip_sample = clip_sample
        self.clip_sample_range = clip_sample_range
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)

    def set_timesteps(self, num_inference_steps):
        self.num_inference_steps = int(num_inference_steps)
        ratio = self.num_train_timesteps // self.num_inference_steps
        self.timesteps = (torch.arange(0, self.num_inference_steps) * ratio).flip(0)

    def _prev(self, t):
        n = self.num_inference_steps if self.num_inference_steps else self.num_train_timesteps
        return t - self.num_train_timesteps // n

    def add_noise(self, x0, noise, t):
        ac = self.alphas_cumprod.to(x0.device)
        a = ac[t] ** 0.5
        s = (1 - ac[t]) ** 0.5
        while a.ndim < x0.ndim:
            a, s = a.unsqueeze(-1), s.unsqueeze(-1)
        return a * x0 + s * noise


class DDIMScheduler(_SchedulerBase):
    def coefficients(self, t):
        """(sqrt(1-abar_t), sqrt(abar_t), sqrt(abar_prev), sqrt(1-abar_prev-sigma^2)) as 0-dim f32 tensors."""
        prev_t = self._prev(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        variance = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
        std_dev_t = 0.0 * variance ** 0.5  # eta = 0
        return b_t ** 0.5, a_t ** 0.5, a_prev ** 0.5, (1 - a_prev - std_dev_t ** 2) ** 0.5

    def step(self, model_output, timestep, sample):
        t = int(timestep)
        sb, sa, sap, sdir = self.coefficients(t)
        x0 = (sample - sb * model_output) / sa
        if self.clip_sample:
            x0 = x0.clamp(-self.clip_sample_range, self.clip_sample_range)
        prev = sap * x0 + sdir * model_output
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)


class DDPMScheduler(_SchedulerBase):
    def __init__(self, variance_type="fixed_small", **kw):
        super().__init__(**kw)
        self.variance_type = variance_type

    def coefficients(self, t):
        """(sqrt(1-abar_t), sqrt(abar_t), coef_x0, coef_xt, sqrt(var)) as 0-dim f32 tensors."""
        prev_t = self._prev(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1 - cur_alpha
        c_x0 = (a_prev ** 0.5 * cur_beta) / b_t
        c_xt = cur_alpha ** 0.5 * b_prev / b_t
        var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_beta, min=1e-20)
        if self.variance_type == "fixed_large":
            var = cur_beta
        elif self.variance_type != "fixed_small":
            raise NotImplementedError(self.variance_type)
        return b_t ** 0.5, a_t ** 0.5, c_x0, c_xt, var ** 0.5

    def step(self, model_output, timestep, sample, noise=None):
        t = int(timestep)
        sb, sa, c_x0, c_xt, sd = self.coefficients(t)
        x0 = (sample - sb * model_output) / sa
        if self.clip_sample:
            x0 = x0.clamp(-self.clip_sample_range, self.clip_sample_range)
        prev = c_x0 * x0 + c_xt * sample
        if t > 0:
            if noise is None:
                noise = torch.randn(model_output.shape, dtype=model_output.dtype, device=model_output.device)
            prev = prev + sd * noise
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)
