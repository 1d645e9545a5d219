# Needs a diagnostic build of the library: `make -C graspldm_amd/csrc clean all EXTRA=-DGLDM_DEBUG_KNOBS`
# (the shipped build reads no environment variable: GLDM_R1D_SKIP is compiled out).
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, json, torch
sys.path.insert(0, %r)
from graspldm_amd import _lib
if os.environ.get('GLDM_LIB'): _lib.LIB_PATH = os.environ['GLDM_LIB']
from graspldm_amd.pipeline import build_fpc_ldm
from graspldm_amd.r1d_pack import SCHED_DDIM
ldm = build_fpc_ldm(device="cuda:0"); ldm.set_inference_timesteps(100)
eng = ldm.diffusion_model.model.engine(torch.device("cuda:0"))
n = int(os.environ.get("NLAT", "4096"))
z = torch.randn(n // 16, 3, 64, device="cuda"); x = torch.randn(n, 1, 4, device="cuda")
cemb = eng.cond_embed(z); ts, coef = ldm.diffusion_model._schedule(torch.device("cuda:0"))
ts, coef = ts[:20].contiguous(), coef[:20].contiguous()
f = lambda: eng.denoise(x, cemb, 16, timesteps=ts, sched_kind=SCHED_DDIM, coef=coef)
f(); torch.cuda.synchronize(); t = time.time(); f(); torch.cuda.synchronize()
print("RESULT", (time.time() - t) * 1e3 / 20)
''' % ROOT
for name, mask in [("all", 0), ("noGN", 1), ("noLN", 2), ("noAttnMath", 4), ("noGEMM", 8), ("noSS", 16), ("onlyGEMM", 1 | 2 | 4 | 16),
                   ("nothing", 31), ("nothing-G", 31 | 32), ("nothing-G-init", 31 | 32 | 64), ("tape only", 31 | 32 | 64 | 128)]:
    env = dict(os.environ, GLDM_R1D_SKIP=str(mask))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    val = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print(f"{name:12s} mask={mask:2d}  ms/step = {val[0].split()[1] if val else r.stderr[-300:]}", flush=True)
