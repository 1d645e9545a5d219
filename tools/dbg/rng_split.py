"""Which rows of a cut batch differ from the whole batch (in-kernel noise and memory-fed noise side by side)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_schema
from graspldm_amd.synthetic import synthetic_state_dict
from graspldm_amd.r1d import R1dEngine, pack_resnet1d, SCHED_DDPM, step_noise_rng
from graspldm_amd.diffusion import make_schedule_tables

sd = synthetic_state_dict(load_schema("schema_fpc_ldm.json"), seed=0)
den = R1dEngine(pack_resnet1d(sd, "diffusion_model.model.", groups=4, seq_len=4, num_steps=1000), "cuda:0")
steps, n, base, seed = 12, 300, 1000, 0x1234567887654321
ts, coef = make_schedule_tables("ddpm", 1000, 5e-5, 1e-3, "linear", "fixed_large", steps)
ts, coef = ts.cuda(), coef.cuda()
g = torch.Generator().manual_seed(5)
x_T = torch.randn((n, 1, 4), generator=g).cuda()
cemb = den.cond_embed(torch.randn((n, 3, 64), generator=g).cuda())
noise = torch.stack([step_noise_rng(seed, base, s, n, 4, "cuda:0") for s in range(steps)]).reshape(steps, n, 1, 4)
want = den.denoise(x_T, cemb, 1, timesteps=ts, sched_kind=SCHED_DDPM, coef=coef, step_noise=noise)
for cut in (172, 160, 16, 64):
    a = den.denoise_rng(x_T[:cut], cemb[:cut], 1, ts, coef, seed, noise_base=base)
    b = den.denoise_rng(x_T[cut:], cemb[cut:], 1, ts, coef, seed, noise_base=base + cut)
    am = den.denoise(x_T[:cut], cemb[:cut], 1, timesteps=ts, sched_kind=SCHED_DDPM, coef=coef, step_noise=noise[:, :cut].contiguous())
    bm = den.denoise(x_T[cut:], cemb[cut:], 1, timesteps=ts, sched_kind=SCHED_DDPM, coef=coef, step_noise=noise[:, cut:].contiguous())
    bad = ((torch.cat([a, b]) - want).abs().amax((1, 2)) > 0).nonzero().flatten().tolist()
    badm = ((torch.cat([am, bm]) - want).abs().amax((1, 2)) > 0).nonzero().flatten().tolist()
    print("cut", cut, "rng rows differing:", bad[:8], len(bad), "| memory-fed:", badm[:8], len(badm))

want2 = den.denoise(x_T, cemb, 1, timesteps=ts, sched_kind=SCHED_DDPM, coef=coef, step_noise=noise)
print("n=300 twice equal:", torch.equal(want, want2))
for nn in (160, 144, 128, 16):
    p1 = den.denoise(x_T[:nn], cemb[:nn], 1, timesteps=ts, sched_kind=SCHED_DDPM, coef=coef, step_noise=noise[:, :nn].contiguous())
    p2 = den.denoise(x_T[:nn], cemb[:nn], 1, timesteps=ts, sched_kind=SCHED_DDPM, coef=coef, step_noise=noise[:, :nn].contiguous())
    d = (p1 - want[:nn]).abs().amax((1, 2))
    print("prefix", nn, "twice equal:", torch.equal(p1, p2), "rows differing from n=300:", (d > 0).nonzero().flatten().tolist()[:10], "max", d.max().item())
for k in range(1, 13):
    w = den.denoise(x_T, cemb, 1, timesteps=ts[:k].contiguous(), sched_kind=SCHED_DDPM, coef=coef[:k].contiguous(), step_noise=noise[:k].contiguous())
    p = den.denoise(x_T[:160], cemb[:160], 1, timesteps=ts[:k].contiguous(), sched_kind=SCHED_DDPM, coef=coef[:k].contiguous(), step_noise=noise[:k, :160].contiguous())
    d = (p - w[:160]).abs().amax((1, 2))
    print("steps", k, "rows differing:", (d > 0).nonzero().flatten().tolist()[:10], "max", d.max().item())
