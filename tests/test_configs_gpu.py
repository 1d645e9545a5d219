"""GPU (MI355X): the BASELINE.json configurations that are not single small batches, each pinned to vectors captured
from the reference's own Python graph (oracle/make_golden.py):

  configs[2]  256 clouds x 20 grasps, 100 DDIM steps: the two golden clouds of ldm_e2e.npz sit at batch positions 0 and
              255 of a full 256-cloud batch with their golden x_T rows, so the 320-tile / step-segment-chain regime of
              the denoise launch is checked against the REFERENCE (not against another HIP launch);
  configs[4]  4096-point partial cloud, encoder n_points = 4096, 1000 DDPM steps (fixed_large), G = 200;
  ppc         the reference's second shipped experiment (partial_pc config: 16-dim grasp latent, 3 x 256 cloud latent,
              DDPM): denoiser / decoder forwards and the end-to-end poses.

Tolerance 1e-4 on tmrp / logits / H entries (north-star bound), 5e-5 on encoder latents, 2e-5 on single forwards.
The DDPM noise is regenerated from the recorded CPU seed in the reference's draw order (x_T, then one draw per step
with t > 0)."""
import pytest
import torch

from conftest import load_golden, load_schema
from test_modules_cpu import build_fpc

pytestmark = pytest.mark.gpu


def _err(a, b):
    return (a.detach().cpu() - b).abs().max().item()


def _ddpm_noise(seed, n, dim, steps=1000):
    """torch.manual_seed(seed); x_T; then the scheduler's draws for t = 999..1 (none at t = 0)."""
    torch.manual_seed(seed)
    x_T = torch.randn(n, 1, dim)
    noise = torch.stack([torch.randn(n, 1, dim) for _ in range(steps - 1)] + [torch.zeros(n, 1, dim)])
    return x_T, noise


def test_config3_full_batch_with_golden_clouds_embedded(fpc_state_dict):
    from graspldm_amd.r1d import pose_epilogue
    from graspldm_amd.synthetic import synthetic_batch
    g = load_golden("ldm_e2e.npz")
    ldm = build_fpc(scheduler="ddim")
    ldm.load_state_dict(fpc_state_dict, strict=True)
    ldm = ldm.cuda().eval()
    ldm.set_inference_timesteps(100)
    B, G = 256, 20
    pcs, metas = synthetic_batch(B, 1024, first_index=100)
    pcs[0], pcs[B - 1] = g["pc"][0], g["pc"][1]
    torch.manual_seed(int(g["seed"]))
    x_gold = torch.randn(2 * G, 1, 4)
    x_T = torch.randn(B * G, 1, 4, generator=torch.Generator().manual_seed(99))
    x_T[:G], x_T[(B - 1) * G:] = x_gold[:G], x_gold[G:]
    (tmrp, logit), _ = ldm.generate_grasps(pcs.cuda(), num_grasps=G, x_T=x_T)
    assert tmrp.shape == (B * G, 6)
    sel = torch.cat([torch.arange(G), torch.arange((B - 1) * G, B * G)])
    assert _err(tmrp[sel], g["tmrp"]) < 1e-4, _err(tmrp[sel], g["tmrp"])
    assert _err(logit[sel], g["logit"]) < 1e-4
    gm, gs = metas["grasp_mean"].clone(), metas["grasp_std"].clone()
    gm[0], gm[B - 1], gs[0], gs[B - 1] = g["grasp_mean"][0], g["grasp_mean"][1], g["grasp_std"][0], g["grasp_std"][1]
    H, _, conf = pose_epilogue(tmrp, logit, gm.cuda(), gs.cuda(), G)
    assert _err(H[sel].view(2, G, 4, 4), g["H"]) < 1e-4
    assert _err(conf[sel].view(2, G, 1), g["confidence"]) < 1e-4
    assert torch.isfinite(tmrp).all() and torch.isfinite(H).all()


def test_config5_reference_golden(fpc_state_dict):
    """BASELINE configs[4] (one object of it): 4096-point partial cloud, 1000 DDPM steps, 200 grasps."""
    from graspldm_amd.r1d import pose_epilogue
    from graspldm_amd.synthetic import synthetic_state_dict
    g = load_golden("c5_ldm_e2e.npz")
    G = int(g["num_grasps"])
    schema = dict(load_schema("schema_fpc_ldm.json"))
    schema["vae_model.encoder.pc_encoder.out_layer.1.weight"] = ((64, 4096), torch.float32)  # the only N-dependent entry
    ldm = build_fpc(n_points=4096, scheduler="ddpm")
    ldm.load_state_dict(synthetic_state_dict(schema, seed=0), strict=True)
    ldm = ldm.cuda().eval()
    assert ldm.diffusion_model.num_inference_steps == 1000
    z = ldm.vae_model.encode_pc(g["pc"].cuda())
    assert _err(z, g["z"]) < 5e-5, _err(z, g["z"])
    x_T, noise = _ddpm_noise(int(g["seed"]), G, 4)
    (tmrp, logit), _ = ldm.generate_grasps(g["pc"].cuda(), num_grasps=G, x_T=x_T, step_noise=noise.cuda())
    assert tmrp.shape == (G, 6)
    assert _err(tmrp, g["tmrp"]) < 1e-4, _err(tmrp, g["tmrp"])
    assert _err(logit, g["logit"]) < 1e-4
    H, _, conf = pose_epilogue(tmrp, logit, g["grasp_mean"].cuda(), g["grasp_std"].cuda(), G)
    assert _err(H.view(1, G, 4, 4), g["H"]) < 1e-4
    assert _err(conf.view(1, G, 1), g["confidence"]) < 1e-4


@pytest.fixture(scope="module")
def ppc():
    from graspldm_amd.builder import build_model_from_cfg
    from graspldm_amd.pipeline import fpc_model_config
    from graspldm_amd.synthetic import synthetic_state_dict
    cfg = fpc_model_config(scheduler="ddpm", latent=16, pc_latent=256)
    ldm = build_model_from_cfg(cfg["ddm"])
    ldm.set_vae_model(build_model_from_cfg(cfg["vae"]))
    ldm.load_state_dict(synthetic_state_dict(load_schema("schema_ppc_ldm.json"), seed=0), strict=True)
    return ldm.cuda().eval()


def test_ppc_encoder_denoiser_decoder_forwards(ppc):
    g = load_golden("ppc_ldm_e2e.npz")
    z = ppc.vae_model.encode_pc(g["pc"].cuda())
    assert z.shape == (2, 3, 256) and _err(z, g["z"]) < 5e-5, _err(z, g["z"])
    den = ppc.diffusion_model.model
    for i, t in enumerate(g["den_t"].tolist()):
        tb = torch.full((6,), t, dtype=torch.long, device="cuda")
        eps = den(g["den_x"].cuda(), time=tb, z_cond=g["den_zc"].cuda())
        assert _err(eps, g["den_eps"][i]) < 2e-5, (t, _err(eps, g["den_eps"][i]))
    tmrp, logit = ppc.vae_model.decoder(g["dec_zh"].cuda(), g["den_zc"].cuda())
    assert _err(tmrp, g["dec_tmrp"]) < 2e-5 and _err(logit, g["dec_logit"]) < 2e-5


def test_ppc_end_to_end_golden(ppc):
    from graspldm_amd.r1d import pose_epilogue
    g = load_golden("ppc_ldm_e2e.npz")
    G = int(g["num_grasps"])
    x_T, noise = _ddpm_noise(int(g["seed"]), 2 * G, 16)
    (tmrp, logit), _ = ppc.generate_grasps(g["pc"].cuda(), num_grasps=G, x_T=x_T, step_noise=noise.cuda())
    assert _err(tmrp, g["tmrp"]) < 1e-4, _err(tmrp, g["tmrp"])
    assert _err(logit, g["logit"]) < 1e-4
    H, _, conf = pose_epilogue(tmrp, logit, g["grasp_mean"].cuda(), g["grasp_std"].cuda(), G)
    assert _err(H.view(2, G, 4, 4), g["H"]) < 1e-4 and _err(conf.view(2, G, 1), g["confidence"]) < 1e-4
