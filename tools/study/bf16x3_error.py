"""Numerical study (CPU): the denoiser's convs evaluated as split-bf16 products -- every f32 operand written as
hi + mid + lo (three bf16 numbers, exact), the six partial products with weight >= 2^-16 kept, accumulation in f32 --
against the plain f32 graph and against the golden vectors captured from the reference.  Emulates what
v_mfma_f32_16x16x32_bf16 would compute (up to the hardware's internal summation order).
    python tools/study/bf16x3_error.py [terms]      terms = 6 (default) or 9 or 3
"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import torch_ref as R
from conftest import load_golden, load_schema
from graspldm_amd.synthetic import synthetic_state_dict

TERMS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
PAIRS = {3: [(0, 0), (0, 1), (1, 0)], 6: [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)],
         5: [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1)],   # (weight plane, activation plane): weights kept to hi + mid only
         4: [(0, 0), (0, 1), (1, 0), (1, 1)],
         9: [(i, j) for i in range(3) for j in range(3)]}[TERMS]

def split3(x):
    hi = x.to(torch.bfloat16).float(); r = x - hi
    mid = r.to(torch.bfloat16).float(); r = r - mid
    lo = r.to(torch.bfloat16).float()
    return hi, mid, lo

real_conv1d = F.conv1d
def conv1d_split(x, w, b=None, stride=1, padding=0, **kw):
    if w.shape[1] < 16:   # init conv / 4-channel level stay on the VALU in f32
        return real_conv1d(x, w, b, stride=stride, padding=padding)
    xs, ws = split3(x), split3(w)
    out = None
    pairs = PAIRS
    if os.environ.get("WIDE_TWO_PLANES") and w.shape[0] == 256 and w.shape[2] == 3:
        pairs = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1)]   # weights hi + mid only on the 256-output k=3 convs
    for i, j in pairs:
        t = real_conv1d(xs[j], ws[i], None, padding=padding)
        out = t if out is None else out + t
    return out + b.view(1, -1, 1) if b is not None else out

sd = synthetic_state_dict(load_schema("schema_fpc_ldm.json"), seed=0)
g = load_golden("denoiser.npz")
def run_forward():
    outs = []
    for t in g["t"].tolist():
        tb = torch.full((8,), t, dtype=torch.long)
        outs.append(R.resnet1d_forward(sd, "diffusion_model.model.", g["x"], z_cond=g["z_cond"], time=tb))
    return torch.stack(outs)
ref32 = run_forward()
R.F.conv1d = conv1d_split
spl = run_forward()
R.F.conv1d = real_conv1d
print(f"terms={TERMS}: single forward: f32 oracle vs golden {(ref32 - g['eps']).abs().max():.2e}; "
      f"split vs golden {(spl - g['eps']).abs().max():.2e}; split vs f32 {(spl - ref32).abs().max():.2e}  (|eps| max {g['eps'].abs().max():.2f})")

# 100-step DDIM trajectory
gd = load_golden("ddim_traj.npz")
sched = R.make_scheduler("ddim"); sched.set_timesteps(100)
x32, _ = R.sample_latents(sd, "diffusion_model.model.", gd["z_cond"], sched, 4, x_T=gd["x_T"])
R.F.conv1d = conv1d_split
sched = R.make_scheduler("ddim"); sched.set_timesteps(100)
xs, _ = R.sample_latents(sd, "diffusion_model.model.", gd["z_cond"], sched, 4, x_T=gd["x_T"])
R.F.conv1d = real_conv1d
print(f"100 DDIM steps: f32 oracle vs golden {(x32 - gd['x0']).abs().max():.2e}; split vs golden {(xs - gd['x0']).abs().max():.2e}")
