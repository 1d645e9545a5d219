"""Numerical study (CPU): the denoiser's convs evaluated as split-f16 products -- every f32 operand written as
hi + lo (two f16 numbers: 11 + 11 significant bits), the three partial products hi hi, hi lo, lo hi kept (lo lo is
<= 2^-22 |a b|), accumulation in f32 -- against the plain f32 graph and the golden vectors captured from the reference.
Emulates what v_mfma_f32_16x16x32_f16 would compute (up to the hardware's internal summation order).
    python tools/study/f16x2_error.py [mode]
mode: keep   = f16 subnormals kept (what the matrix pipe does if it honours MODE.denorm = keep)
      flush  = f16 subnormal operands flushed to zero (worst case)
      scale  = lo stored as f16(r * 2^11) (never subnormal unless x is), its two products summed in a second accumulator
               that is folded in with one fma; subnormal hi flushed (worst case for that form)
"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import torch_ref as R
from conftest import load_golden, load_schema
from graspldm_amd.synthetic import synthetic_state_dict

MODE = sys.argv[1] if len(sys.argv) > 1 else "keep"
TINY = 2.0 ** -14

def to16(x, flush):
    h = x.to(torch.float16).float()
    if flush:
        h = torch.where(h.abs() < TINY, torch.zeros_like(h), h)
    return h

def split2(x):
    if MODE == "scale":
        hi = to16(x, True); lo = to16((x - hi) * 2048.0, True)
        return hi, lo
    hi = to16(x, MODE == "flush"); lo = to16(x - hi, MODE == "flush")
    return hi, lo

real_conv1d = F.conv1d
stats = {"max_abs": 0.0}
def conv1d_split(x, w, b=None, stride=1, padding=0, **kw):
    if w.shape[1] < 16:
        return real_conv1d(x, w, b, stride=stride, padding=padding)
    stats["max_abs"] = max(stats["max_abs"], float(x.abs().max()), float(w.abs().max()))
    xh, xl = split2(x); wh, wl = split2(w)
    main = real_conv1d(xh, wh, None, padding=padding)
    small = real_conv1d(xl, wh, None, padding=padding) + real_conv1d(xh, wl, None, padding=padding)
    out = main + (small * (1.0 / 2048.0) if MODE == "scale" else small)
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
print(f"mode={MODE}: single forward: f32 oracle vs golden {(ref32 - g['eps']).abs().max():.2e}; "
      f"split vs golden {(spl - g['eps']).abs().max():.2e}; split vs f32 {(spl - ref32).abs().max():.2e}  (|eps| max {g['eps'].abs().max():.2f}; "
      f"largest conv operand {stats['max_abs']:.1f})")

gd = load_golden("ddim_traj.npz")
sched = R.make_scheduler("ddim"); sched.set_timesteps(100)
x32, _ = R.sample_latents(sd, "diffusion_model.model.", gd["z_cond"], sched, 4, x_T=gd["x_T"])
R.F.conv1d = conv1d_split
sched = R.make_scheduler("ddim"); sched.set_timesteps(100)
xs, _ = R.sample_latents(sd, "diffusion_model.model.", gd["z_cond"], sched, 4, x_T=gd["x_T"])
R.F.conv1d = real_conv1d
print(f"100 DDIM steps: f32 oracle vs golden {(x32 - gd['x0']).abs().max():.2e}; split vs golden {(xs - gd['x0']).abs().max():.2e}; split vs f32 {(xs - x32).abs().max():.2e}")
