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


class GaussianDiffusion1D(nn.Module):
    """Sampling half of gaussian_diffusion.py:10-277 with the same constructor arguments
    and attributes (`model`, `n_dims`, `beta_start/end`, `num_steps`,
    `num_inference_steps`, `set_inference_timesteps`, `sample`)."""
    NOISE_SCHEDULERS = ["ddpm", "ddim"]
    BETA_SCHEDULES = ["linear", "scaled_linear", "squaredcos_cap_v2", "cosine"]
    VARIANCE_TYPES = ["fixed_small", "fixed_small_log", "fixed_large", "fixed_large_log", "learned", "learned_range"]

    def __init__(self, model, n_dims, noise_scheduler_type="ddpm", beta_schedule="linear",
                 variance_type="fixed_small", pred_type="epsilon", beta_start=0.0001, beta_end=0.02, num_steps=1000,
                 loss_type="l1", clip_sample=True):
        super().__init__()
        assert noise_scheduler_type in self.NOISE_SCHEDULERS, f"{noise_scheduler_type} Not supported"
        assert beta_schedule in self.BETA_SCHEDULES, f"{beta_schedule} not supported"
        assert variance_type in self.VARIANCE_TYPES, f"{variance_type} not supported"
        if pred_type != "epsilon":
            raise NotImplementedError("only epsilon prediction is on the generation hot path")
        if variance_type in ("learned", "learned_range"):
            raise NotImplementedError("learned variance is not on the generation hot path")
        assert model.out_channels == 1, (
            f"For pre-defined variance type {variance_type}, the score model should have only one output channel")
        self.num_train_timesteps = self.num_steps = num_steps
        self.beta_start, self.beta_end = beta_start, beta_end
        self.beta_schedule = beta_schedule if beta_schedule != "cosine" else "squaredcos_cap_v2"
        self.variance_type, self.pred_type, self.clip_sample = variance_type, pred_type, clip_sample
        self.model, self.n_dims, self.channels = model, n_dims, 1
        self.loss_type = loss_type
        self._noise_scheduler_type = noise_scheduler_type
        self._num_inference_steps = None
        self._tables = {}
        if hasattr(model, "max_timesteps"):
            model.max_timesteps = max(model.max_timesteps, num_steps)

    @property
    def num_inference_steps(self):
        return self._num_inference_steps if self._num_inference_steps is not None else self.num_steps

    def set_inference_timesteps(self, num_steps):
        self._num_inference_steps = int(num_steps)

    def _schedule(self, device):
        key = (str(device), self._noise_scheduler_type, self._num_inference_steps)
        if key not in self._tables:
            ts, coef = make_schedule_tables(self._noise_scheduler_type, self.num_steps, self.beta_start, self.beta_end,
                                            self.beta_schedule, self.variance_type, self._num_inference_steps)
            self._tables[key] = (ts.to(device), coef.to(device))
        return self._tables[key]

    @torch.no_grad()
    def sample(self, z_cond=None, batch_size=1, return_all=False, device=None, samples_per_cond=1, x_T=None,
               step_noise=None, noise_source="tensor", noise_seed=None, noise_base=0, **kwargs):
        """Reverse diffusion (gaussian_diffusion.py:232-277).  x_T is drawn on the CPU
        generator then moved, exactly like the reference (:253); DDPM per-step noise is
        drawn on the device ([steps, B, 1, D]) unless `step_noise` is given.  One HIP launch
        runs every step.  `z_cond` is [B/samples_per_cond, R, Dc] (the reference passes the
        repeat_interleaved tensor: samples_per_cond = 1).

        noise_source="kernel" (DDPM, return_all=False): the per-step normals are drawn INSIDE the launch from a counter-based
        generator (gldm_denoise_rng) instead of from a [steps, B, 1, D] tensor -- the same distribution, not torch's
        stream; `noise_seed` defaults to one draw of torch's CPU generator (so torch.manual_seed still fixes a run) and
        `noise_base` is the global index of this batch's first latent: ranks of a sharded job that share a seed MUST pass
        their own (distributed.shard_noise_base), or they add identical step noise to different latents."""
        device = torch.device(device if device is not None else z_cond.device)
        if device.type != "cuda":
            raise RuntimeError("sampling runs on the GPU only (graspldm_amd has no CPU path)")
        if x_T is None:
            x_T = torch.randn((batch_size, self.channels, self.n_dims))
        x_T = x_T.to(device)
        ts, coef = self._schedule(device)
        kind = SCHED_DDIM if self._noise_scheduler_type == "ddim" else SCHED_DDPM
        if noise_source not in ("tensor", "kernel"):
            raise ValueError(f"noise_source must be 'tensor' or 'kernel', not {noise_source!r}")
        in_kernel = noise_source == "kernel" and kind == SCHED_DDPM and step_noise is None and not return_all
        if noise_source == "kernel" and not in_kernel and kind == SCHED_DDPM:
            # (DDIM adds no step noise: nothing to ignore there)
            import warnings
            warnings.warn("noise_source='kernel' is not honoured with return_all=True or an explicit step_noise: the per-step "
                          "normals come from torch's generator / the given tensor, so this run does NOT reproduce the "
                          "in-kernel stream of the same seed", RuntimeWarning, stacklevel=2)
        if in_kernel and noise_seed is None:
            noise_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        if kind == SCHED_DDPM and step_noise is None and not in_kernel:
            step_noise = torch.randn((ts.numel(), batch_size, self.channels, self.n_dims), device=device)
        model = self.model
        model._cond_rows_of(z_cond)
        eng = model.engine(device)
        cemb = eng.cond_embed(z_cond.to(device))
        # class-conditioned denoiser: the label travels in kwargs (cls_cond or metas["mode_cls"]) to the model
        # call in the reference (gaussian_diffusion.py:271); here it becomes one [n, emb] operand of the launch
        semb = model.class_embedding(kwargs.get("cls_cond"), n=batch_size, metas=kwargs.get("metas")) \
            if hasattr(model, "class_embedding") else None
        if in_kernel:
            x = eng.denoise_rng(x_T, cemb, samples_per_cond, ts, coef, noise_seed, noise_base=noise_base,
                                clip_sample=self.clip_sample, sample_emb=semb)
            return x, []
        if not return_all:
            x = eng.denoise(x_T, cemb, samples_per_cond, timesteps=ts, sched_kind=kind, clip_sample=self.clip_sample,
                            coef=coef, step_noise=step_noise, sample_emb=semb)
            return x, []
        trace, x = [x_T], x_T
        for i in range(ts.numel()):
            x = eng.denoise(x, cemb, samples_per_cond, timesteps=ts[i:i + 1], sched_kind=kind,
                            clip_sample=self.clip_sample, coef=coef[i:i + 1],
                            step_noise=None if step_noise is None else step_noise[i:i + 1], sample_emb=semb)
            trace.append(x)
        return x, trace

    def forward(self, *args, **kwargs):
        raise NotImplementedError("training (denoising loss) is out of scope: graspldm_amd is the generation path")
