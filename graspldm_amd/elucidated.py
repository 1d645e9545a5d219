"""ElucidatedDiffusion (sampling half): mirror of `grasp_ldm/models/diffusion/elucidated_diffusion.py:36-313`
with the same constructor arguments and attributes.  The DPM-Solver++(2M) sampler (`sample_using_dpmpp`, :259-313)
runs as ONE fused HIP launch (gldm_denoise, GLDM_SCHED_DPMPP): the preconditioning c_in / c_skip / c_out, the
continuous time c_noise(sigma) = log(sigma) / 4 (one time-embedding row per step, built on the host like the
module builds it), the second-order blend with the previous denoised sample and the exponential-integrator update
are per-step f32 coefficients computed here with 0-dim f32 tensors exactly as the reference computes them.

`sample_normal` (the stochastic Heun sampler, :177-257: two network evaluations and fresh noise per step) drives
the same engine one network evaluation per launch (gldm_denoise with GLDM_SCHED_NONE and a one-row time-embedding
table); the few elementwise updates between the evaluations are torch ops on the [B,1,D] latent.  The reference's
inference entry point only ever asks for DPM++ (tools/inference.py:607-609), so this path is kept simple, not fused.
"""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn

from .r1d_pack import SCHED_COEF_STRIDE

SCHED_DPMPP = 3


def _log(t, eps=1e-20):
    return torch.log(t.clamp(min=eps))


class ElucidatedDiffusion(nn.Module):
    def __init__(self, net, *, seq_length, channels=1, num_sample_steps=32, sigma_min=0.002, sigma_max=80,
                 sigma_data=0.5, rho=7, P_mean=-1.2, P_std=1.2, S_churn=80, S_tmin=0.05, S_tmax=50, S_noise=1.003):
        super().__init__()
        assert net.random_or_learned_sinusoidal_cond
        self.self_condition = False
        self.net = net
        self.channels, self.seq_length = channels, seq_length
        self.sigma_min, self.sigma_max, self.sigma_data, self.rho = sigma_min, sigma_max, sigma_data, rho
        self.P_mean, self.P_std, self.num_sample_steps = P_mean, P_std, num_sample_steps
        self.S_churn, self.S_tmin, self.S_tmax, self.S_noise = S_churn, S_tmin, S_tmax, S_noise
        self._noise_scheduler_type = "elucidated"

    @property
    def device(self):
        return next(self.net.parameters()).device

    # derived preconditioning params - Table 1 of the paper (elucidated_diffusion.py:103-115)
    def c_skip(self, sigma):
        return (self.sigma_data ** 2) / (sigma ** 2 + self.sigma_data ** 2)

    def c_out(self, sigma):
        return sigma * self.sigma_data * (self.sigma_data ** 2 + sigma ** 2) ** -0.5

    def c_in(self, sigma):
        return 1 * (sigma ** 2 + self.sigma_data ** 2) ** -0.5

    def c_noise(self, sigma):
        return _log(sigma) * 0.25

    def sample_schedule(self, num_sample_steps=None):
        """elucidated_diffusion.py:149-162 (computed on the CPU in f32: the schedule is host data here)."""
        n = self.num_sample_steps if num_sample_steps is None else num_sample_steps
        inv_rho = 1 / self.rho
        steps = torch.arange(n, dtype=torch.float32)
        sigmas = (self.sigma_max ** inv_rho + steps / (n - 1) * (self.sigma_min ** inv_rho - self.sigma_max ** inv_rho)) ** self.rho
        return F.pad(sigmas, (0, 1), value=0.0)

    def dpmpp_tables(self, num_sample_steps):
        """-> (sigmas f32 [S+1], times f32 [S] = c_noise(sigma_i), coef f32 [S, 8]); layout in include/gldm.h."""
        sigmas = self.sample_schedule(num_sample_steps)
        S = sigmas.numel() - 1
        coef = torch.zeros(S, SCHED_COEF_STRIDE, dtype=torch.float32)
        times = torch.zeros(S, dtype=torch.float32)
        t_fn = lambda sg: sg.log().neg()
        sigma_fn = lambda t: t.neg().exp()
        for i in range(S):
            # preconditioned_network_forward gets sigma as a python float and rebuilds an f32 tensor (:120-123)
            sg = torch.full((1,), sigmas[i].item())
            coef[i, 0], coef[i, 1], coef[i, 2] = self.c_in(sg)[0], self.c_skip(sg)[0], self.c_out(sg)[0]
            times[i] = self.c_noise(sg)[0]
            t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
            h = t_next - t
            if i == 0 or sigmas[i + 1] == 0:
                coef[i, 3], coef[i, 4], coef[i, 7] = 1.0, 0.0, 0.0
            else:
                h_last = t - t_fn(sigmas[i - 1])
                r = h_last / h
                gamma = -1 / (2 * r)
                coef[i, 3], coef[i, 4], coef[i, 7] = 1 - gamma, gamma, 1.0
            coef[i, 5] = sigma_fn(t_next) / sigma_fn(t)
            coef[i, 6] = (-h).expm1()
        return sigmas, times, coef

    def _time_table(self, times):
        """time_mlp(times) on the host, like RandomOrLearnedSinusoidalPosEmb + the MLP do for a float time
        (resnets.py:44-56,517-522): one row per sampling step."""
        sd = {k: v.detach().float().cpu() for k, v in self.net.state_dict().items() if k.startswith("time_mlp.")}
        t = times.reshape(-1, 1)
        freqs = t * sd["time_mlp.0.weights"].reshape(1, -1) * 2 * math.pi
        four = torch.cat((t, freqs.sin(), freqs.cos()), dim=-1)
        h = F.linear(four, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
        return F.linear(F.gelu(h), sd["time_mlp.3.weight"], sd["time_mlp.3.bias"]).contiguous()

    def sample(self, **kwargs):
        if kwargs.pop("use_dpmpp"):  # KeyError when absent, like the reference (:165)
            return self.sample_using_dpmpp(**kwargs)
        return self.sample_normal(**kwargs)

    def _net_eval(self, eng, cemb, samples_per_cond, x_in, sigma, device):
        """preconditioned_network_forward (:117-139) through the engine: F(c_in x; c_noise(sigma))."""
        sg = torch.full((1,), float(sigma))
        temb = self._time_table(self.c_noise(sg)).to(device)
        zero = torch.zeros(1, dtype=torch.int32, device=device)
        c_in, c_skip, c_out = (f(sg)[0].to(device) for f in (self.c_in, self.c_skip, self.c_out))
        net = eng.denoise(c_in * x_in, cemb, samples_per_cond, timesteps=zero, temb=temb)
        return c_skip * x_in + c_out * net

    @torch.no_grad()
    def sample_normal(self, batch_size=16, z_cond=None, num_sample_steps=None, clamp=False, return_all=False, noise=None,
                      step_noise=None, samples_per_cond=1, device=None):
        """elucidated_diffusion.py:177-257: stochastic sampler with the second-order (Heun) correction.  `noise`
        [B,1,D] and `step_noise` [S,B,1,D] (unit normal) default to draws on the model's device, in the reference's
        order (x first, then one draw per step)."""
        from math import sqrt
        device = torch.device(self.device if device is None else device)
        if device.type != "cuda":
            raise RuntimeError("sampling runs on the GPU only (graspldm_amd has no CPU path)")
        n_steps = self.num_sample_steps if num_sample_steps is None else num_sample_steps
        shape = (batch_size, self.channels, self.seq_length)
        sigmas = self.sample_schedule(n_steps)
        gammas = torch.where((sigmas >= self.S_tmin) & (sigmas <= self.S_tmax),
                             min(self.S_churn / n_steps, sqrt(2) - 1), 0.0)
        if noise is None:
            noise = torch.randn(shape, device=device)
        x = sigmas[0].to(device) * noise.to(device)
        net = self.net
        net._cond_rows_of(z_cond)
        eng = net.engine(device)
        cemb = eng.cond_embed(z_cond.to(device))
        all_x = [x]
        for i in range(n_steps):
            sigma, sigma_next, gamma = sigmas[i].item(), sigmas[i + 1].item(), gammas[i].item()
            eps = self.S_noise * (torch.randn(shape, device=device) if step_noise is None else step_noise[i].to(device))
            sigma_hat = sigma + gamma * sigma
            x_hat = x + sqrt(sigma_hat ** 2 - sigma ** 2) * eps
            out = self._net_eval(eng, cemb, samples_per_cond, x_hat, sigma_hat, device)
            if clamp:
                out = out.clamp(-1.0, 1.0)
            d_over = (x_hat - out) / sigma_hat
            x_next = x_hat + (sigma_next - sigma_hat) * d_over
            if sigma_next != 0:
                out2 = self._net_eval(eng, cemb, samples_per_cond, x_next, sigma_next, device)
                if clamp:
                    out2 = out2.clamp(-1.0, 1.0)
                d_prime = (x_next - out2) / sigma_next
                x_next = x_hat + 0.5 * (sigma_next - sigma_hat) * (d_over + d_prime)
            x = x_next
            if return_all:
                all_x.append(x)
        return x, all_x

    @torch.no_grad()
    def sample_using_dpmpp(self, batch_size=16, z_cond=None, num_sample_steps=20, clamp=False, return_all=False,
                           noise=None, samples_per_cond=1, device=None):
        """elucidated_diffusion.py:259-313.  `noise` [B,1,D] (unit normal; default: drawn on the model's device like
        the reference, :281) is scaled by sigma_0.  `z_cond` is [B / samples_per_cond, R, Dc]."""
        device = torch.device(self.device if device is None else device)
        if device.type != "cuda":
            raise RuntimeError("sampling runs on the GPU only (graspldm_amd has no CPU path)")
        if batch_size != z_cond.shape[0] * samples_per_cond:
            warnings.warn(f"The batch size for sample generation {batch_size} is different from conditioning "
                          f"batch_size {z_cond.shape[0] * samples_per_cond}.")
        n_steps = self.num_sample_steps if num_sample_steps is None else num_sample_steps
        sigmas, times, coef = self.dpmpp_tables(n_steps)
        if noise is None:
            noise = torch.randn((batch_size, self.channels, self.seq_length), device=device)
        x = sigmas[0].to(device) * noise.to(device)
        net = self.net
        net._cond_rows_of(z_cond)
        eng = net.engine(device)
        cemb = eng.cond_embed(z_cond.to(device))
        temb = self._time_table(times).to(device)   # one row per step; the engine indexes it with `timesteps`
        ts = torch.arange(n_steps, dtype=torch.int32, device=device)
        if return_all:
            # the per-step form (elucidated_diffusion.py:287-311 keeps every x): one network launch per step, the
            # 2M update as elementwise torch ops in the fused launch's operation order (one rounding per operation)
            cf = coef.to(device)
            all_x, old = [x], None
            for i in range(n_steps):
                net_out = eng.denoise(cf[i, 0] * x, cemb, samples_per_cond, timesteps=ts[i:i + 1], temb=temb)
                den = cf[i, 1] * x + cf[i, 2] * net_out
                if clamp:
                    den = den.clamp(-1.0, 1.0)
                d = den if (coef[i, 7] == 0 or old is None) else cf[i, 3] * den + cf[i, 4] * old
                old = den
                x = cf[i, 5] * x - cf[i, 6] * d
                all_x.append(x)
            return x, all_x
        out = eng.denoise(x, cemb, samples_per_cond, timesteps=ts, sched_kind=SCHED_DPMPP, clip_sample=clamp,
                          coef=coef.to(device), temb=temb)
        return out, [x]

    def forward(self, *a, **k):
        raise NotImplementedError("training (denoising loss) is out of scope: graspldm_amd is the generation path")
