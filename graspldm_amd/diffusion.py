"""Reverse-diffusion sampling for the grasp latent: schedule tables on the host,
the T-step loop fused in one HIP launch (csrc/resnet1d.hip: gldm_denoise).

Mirrors `GaussianDiffusion1D` (grasp_ldm/models/diffusion/gaussian_diffusion.py:
10-277; sampling half only).  The reference delegates the step arithmetic to
`diffusers` (unpinned third-party, absent here); DDIM (eta = 0) and DDPM
("fixed_large"/"fixed_small") are restated from the published algorithms with the
scheduler defaults the reference relies on: clip_sample range 1, leading
timestep spacing, steps_offset 0, set_alpha_to_one.  Coefficients are evaluated
with 0-dim f32 torch tensors like the library does and handed to the kernel as a
[steps, 8] table, so the per-element update is bit-identical given the same eps.
"""
import torch
from torch import nn

from .r1d_pack import SCHED_COEF_STRIDE, SCHED_DDIM, SCHED_DDPM, SCHED_NONE  # noqa: F401


def _alphas_cumprod(num_steps, beta_start, beta_end, beta_schedule):
    if beta_schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, num_steps, dtype=torch.float32)
    elif beta_schedule == "scaled_linear":
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_steps, dtype=torch.float32) ** 2
    else:
        raise NotImplementedError(f"beta_schedule {beta_schedule!r} is not on the hot path")
    return torch.cumprod(1.0 - betas, dim=0)


def inference_timesteps(num_train_steps, num_inference_steps):
    """The reference's loop: reversed(range(0, T, T // S))  (gaussian_diffusion.py:258-263)."""
    n = num_inference_steps if num_inference_steps else num_train_steps
    return list(reversed(range(0, num_train_steps, num_train_steps // n)))


def make_schedule_tables(kind, num_train_steps, beta_start, beta_end, beta_schedule="linear",
                         variance_type="fixed_large", num_inference_steps=None):
    """-> (timesteps int32 [S], coef f32 [S, 8]); layout documented in include/gldm.h."""
    ac = _alphas_cumprod(num_train_steps, beta_start, beta_end, beta_schedule)
    one = torch.tensor(1.0)
    n = num_inference_steps if num_inference_steps else num_train_steps
    stride = num_train_steps // n
    ts = inference_timesteps(num_train_steps, num_inference_steps)
    coef = torch.zeros(len(ts), SCHED_COEF_STRIDE, dtype=torch.float32)
    for i, t in enumerate(ts):
        prev_t = t - stride
        a_t = ac[t]
        a_prev = ac[prev_t] if prev_t >= 0 else one
        b_t = 1 - a_t
        coef[i, 0] = b_t ** 0.5
        coef[i, 1] = a_t ** 0.5
        if kind == "ddim":
            variance = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
            std = 0.0 * variance ** 0.5
            coef[i, 2] = a_prev ** 0.5
            coef[i, 3] = (1 - a_prev - std ** 2) ** 0.5
        elif kind == "ddpm":
            cur_alpha = a_t / a_prev
            cur_beta = 1 - cur_alpha
            coef[i, 4] = (a_prev ** 0.5 * cur_beta) / b_t
            coef[i, 5] = cur_alpha ** 0.5 * (1 - a_prev) / b_t
            var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_beta, min=1e-20)
            if variance_type == "fixed_large":
                var = cur_beta
            elif variance_type != "fixed_small":
                raise NotImplementedError(f"variance_type {variance_type!r}")
            coef[i, 6] = var ** 0.5
            coef[i, 7] = 1.0 if t > 0 else 0.0
        else:
            raise NotImplementedError(kind)
    return torch.tensor(ts, dtype=torch.int32), coef
