"""Debug: ppc denoiser (16 positions) on the 64-column engine vs the golden forwards, row by row."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, load_schema
from graspldm_amd.builder import build_model_from_cfg
from graspldm_amd.pipeline import fpc_model_config
from graspldm_amd.synthetic import synthetic_state_dict
cfg = fpc_model_config(scheduler="ddpm", latent=16, pc_latent=256)
ldm = build_model_from_cfg(cfg["ddm"]); ldm.set_vae_model(build_model_from_cfg(cfg["vae"]))
ldm.load_state_dict(synthetic_state_dict(load_schema("schema_ppc_ldm.json"), seed=0), strict=True)
ldm = ldm.cuda().eval()
g = load_golden("ppc_ldm_e2e.npz")
den = ldm.diffusion_model.model
t = g["den_t"].tolist()[0]
for n in (1, 2, 3, 6):
    tb = torch.full((n,), t, dtype=torch.long, device="cuda")
    eps = den(g["den_x"][:n].cuda(), time=tb, z_cond=g["den_zc"][:n].cuda())
    e = (eps.cpu() - g["den_eps"][0][:n]).abs().flatten(1).max(dim=1).values
    print("den n", n, [f"{v:.1e}" for v in e.tolist()])
for n in (1, 2, 6):
    tmrp, logit = ldm.vae_model.decoder(g["dec_zh"][:n].cuda(), g["den_zc"][:n].cuda())
    e = (tmrp.cpu() - g["dec_tmrp"][:n]).abs().max(dim=1).values
    print("dec n", n, [f"{v:.1e}" for v in e.tolist()])
