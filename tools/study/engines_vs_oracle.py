"""Both denoiser engines against the CPU oracle on the same 1024 random latents, 100 DDIM steps (run on the GPU box):
the position-major engine (split-f16 GEMMs on the bf16 matrix pipe) and the sample-major engine (f32 matrix pipe only:
a descriptor without the split weight copies).   python tools/study/engines_vs_oracle.py
Measured (round 3): split path max 1.4e-5 / mean 3.3e-7 from the oracle, f32 path max 2.0e-5 / mean 3.3e-7."""
import sys, os, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from conftest import load_schema
from graspldm_amd.synthetic import synthetic_state_dict
from graspldm_amd.r1d import R1dEngine, SCHED_DDIM
from graspldm_amd.r1d_pack import pack_resnet1d
from graspldm_amd.diffusion import make_schedule_tables
from oracle import torch_ref as R
sd = synthetic_state_dict(load_schema("schema_fpc_ldm.json"), seed=0)
pre = "diffusion_model.model."
dev = torch.device("cuda:0")
packed = pack_resnet1d(sd, pre, groups=4, seq_len=4, num_steps=1000)
eng = R1dEngine(packed, dev)
import copy
p32 = pack_resnet1d(sd, pre, groups=4, seq_len=4, num_steps=1000)
for rb in p32["desc"].rb: rb.c1_w3 = rb.c2_w3 = 0
for lv in p32["desc"].lv: lv.qkvn_w3 = lv.out_w3 = lv.down_w3 = 0
eng32 = R1dEngine(p32, dev)
g = torch.Generator().manual_seed(5)
n = 1024
x = torch.randn(n, 1, 4, generator=g); z = torch.randn(n, 3, 64, generator=g)
ts, coef = make_schedule_tables("ddim", 1000, 5e-5, 1e-3, "linear", "fixed_large", 100)
a = eng.denoise(x.to(dev), eng.cond_embed(z.to(dev)), 1, timesteps=ts.to(dev), sched_kind=SCHED_DDIM, coef=coef.to(dev)).cpu()
b = eng32.denoise(x.to(dev), eng32.cond_embed(z.to(dev)), 1, timesteps=ts.to(dev), sched_kind=SCHED_DDIM, coef=coef.to(dev)).cpu()
sched = R.make_scheduler("ddim"); sched.set_timesteps(100)
t0 = time.time()
o, _ = R.sample_latents(sd, pre, z, sched, 4, x_T=x)
print("oracle s", time.time() - t0)
for name, v in (("split-f16 PM engine", a), ("f32 sample-major engine", b)):
    d = (v - o).abs().flatten()
    print(f"{name}: vs oracle max {d.max():.2e} mean {d.mean():.2e} p99.9 {torch.quantile(d, 0.999):.2e}")
d = (a - b).abs().flatten(); print(f"engines vs each other: max {d.max():.2e} mean {d.mean():.2e}")
