"""CPU: the oracle (oracle/torch_ref.py + oracle/point_ops.c + oracle/schedulers.py)
against the golden vectors captured from the reference's Python
(oracle/make_golden.py).  Tolerances are for fp32 reassociation only."""
import torch

from conftest import load_golden, load_schema
from oracle import torch_ref as R
from graspldm_amd.synthetic import synthetic_state_dict

ATOL = 2e-6


def _close(a, b, atol=ATOL, rtol=1e-5):
    err = (a - b).abs().max().item()
    assert torch.allclose(a, b, atol=atol, rtol=rtol), f"max abs err {err:.3e}"


def test_g2_pvcnn_encoder(fpc_state_dict, fpc_spec):
    g = load_golden("pvcnn_encoder.npz")
    z = R.pvcnn_encoder_forward(fpc_state_dict, "vae_model.encoder.pc_encoder.", g["pc"], fpc_spec)
    _close(z, g["z"], atol=1e-5)


def test_g3_denoiser(fpc_state_dict):
    g = load_golden("denoiser.npz")
    for i, t in enumerate(g["t"].tolist()):
        tb = torch.full((g["x"].shape[0],), t, dtype=torch.long)
        eps = R.resnet1d_forward(fpc_state_dict, "diffusion_model.model.", g["x"], z_cond=g["z_cond"], time=tb)
        _close(eps, g["eps"][i])


def test_g4_decoder(fpc_state_dict):
    g = load_golden("decoder.npz")
    tmrp, logit = R.decoder_forward(fpc_state_dict, "vae_model.decoder.", g["z_h"], g["z_cond"])
    _close(tmrp, g["tmrp"])
    _close(logit, g["logit"])


def test_g5_ddim_trajectory(fpc_state_dict):
    g = load_golden("ddim_traj.npz")
    sched = R.make_scheduler("ddim")
    sched.set_timesteps(100)
    x0, trace = R.sample_latents(fpc_state_dict, "diffusion_model.model.", g["z_cond"], sched, 4,
                                 x_T=g["x_T"], return_all=True)
    for j, i in enumerate(g["probes"].tolist()):
        _close(trace[i], g["x"][j], atol=1e-5)
    _close(x0, g["x0"], atol=1e-5)


def test_g5_ddpm_trajectory(fpc_state_dict):
    g = load_golden("ddpm_traj.npz")
    sched = R.make_scheduler("ddpm")
    x0, trace = R.sample_latents(fpc_state_dict, "diffusion_model.model.", g["z_cond"], sched, 4,
                                 x_T=g["x_T"], step_noise=g["step_noise"], return_all=True)
    for j, i in enumerate(g["probes"].tolist()):
        _close(trace[i], g["x"][j], atol=2e-5)
    _close(x0, g["x0"], atol=2e-5)


def test_g6_tmrp_to_H():
    g = load_golden("tmrp_to_H.npz")
    _close(R.tmrp_to_H(g["tmrp"]), g["H"], atol=1e-6)


def test_g7_ldm_end_to_end(fpc_state_dict, fpc_spec):
    g = load_golden("ldm_e2e.npz")
    torch.manual_seed(int(g["seed"]))
    x_T = torch.randn(40, 1, 4)
    sched = R.make_scheduler("ddim")
    sched.set_timesteps(100)
    tmrp, logit = R.ldm_generate(fpc_state_dict, g["pc"], 20, sched, fpc_spec, x_T=x_T)
    _close(tmrp, g["tmrp"], atol=2e-5)
    _close(logit, g["logit"], atol=2e-5)
    out = R.pose_epilogue(tmrp, logit, dict(grasp_mean=g["grasp_mean"], grasp_std=g["grasp_std"]), 2, 20)
    _close(out["grasps"], g["H"], atol=2e-5)
    _close(out["confidence"], g["confidence"], atol=2e-5)


def test_g7_vae_end_to_end(fpc_state_dict, fpc_spec):
    g = load_golden("vae_e2e.npz")
    torch.manual_seed(int(g["seed"]))
    z_h = torch.randn(20, 4)
    tmrp, logit = R.vae_generate(fpc_state_dict, g["pc"], 20, fpc_spec, z_h=z_h, prefix="vae_model.")
    _close(tmrp, g["tmrp"], atol=1e-5)
    _close(logit, g["logit"], atol=1e-5)


def test_g7_vae_n64(fpc_spec):
    g = load_golden("vae_e2e_n64.npz")
    sd = synthetic_state_dict(load_schema("schema_fpc_ldm_n64.json"), seed=0)
    torch.manual_seed(int(g["seed"]))
    z_h = torch.randn(20, 4)
    tmrp, logit = R.vae_generate(sd, g["pc"], 20, fpc_spec, z_h=z_h, prefix="vae_model.")
    _close(tmrp, g["tmrp"], atol=1e-5)
    _close(logit, g["logit"], atol=1e-5)


def test_g8_sa_modules():
    g = load_golden("sa_module.npz")
    sd1 = synthetic_state_dict(load_schema("schema_sa1.json"), seed=1)
    sd2 = synthetic_state_dict(load_schema("schema_sa2.json"), seed=2)
    f1, c1 = R.sa_module(sd1, "", None, g["coords"], 512, [0.2], [64])
    assert torch.equal(c1, g["c1"])  # FPS indices -> gathered coords are exact
    _close(f1[:, :, ::4], g["f1"], atol=1e-5)
    f2, c2 = R.sa_module(sd2, "", f1, c1, 128, [0.4], [64])
    assert torch.equal(c2, g["c2"])
    _close(f2, g["f2"], atol=1e-5)


def test_g8_pointnet2_ssg():
    g = load_golden("pointnet2_ssg.npz")
    sd = synthetic_state_dict(load_schema("schema_pointnet2_ssg.json"), seed=3)
    out = R.pointnet2_ssg_forward(sd, "", g["coords"])
    _close(out[:, :, ::8], g["out"], atol=2e-5)


def _ddpm_noise(seed, n, dim, steps=1000):
    torch.manual_seed(seed)
    x_T = torch.randn(n, 1, dim)
    return x_T, torch.stack([torch.randn(n, 1, dim) for _ in range(steps - 1)])


def test_ppc_config_forwards_and_end_to_end(fpc_spec):
    """The reference's partial-cloud experiment (z16, pc256, DDPM 1000 steps): oracle vs the reference's vectors."""
    g = load_golden("ppc_ldm_e2e.npz")
    sd = synthetic_state_dict(load_schema("schema_ppc_ldm.json"), seed=0)
    z = R.pvcnn_encoder_forward(sd, "vae_model.encoder.pc_encoder.", g["pc"], fpc_spec)
    _close(z, g["z"], atol=1e-5)
    for i, t in enumerate(g["den_t"].tolist()):
        tb = torch.full((6,), t, dtype=torch.long)
        _close(R.resnet1d_forward(sd, "diffusion_model.model.", g["den_x"], z_cond=g["den_zc"], time=tb), g["den_eps"][i])
    tmrp, logit = R.decoder_forward(sd, "vae_model.decoder.", g["dec_zh"], g["den_zc"])
    _close(tmrp, g["dec_tmrp"])
    _close(logit, g["dec_logit"])
    G = int(g["num_grasps"])
    x_T, noise = _ddpm_noise(int(g["seed"]), 2 * G, 16)
    tmrp, logit = R.ldm_generate(sd, g["pc"], G, R.make_scheduler("ddpm"), fpc_spec, n_dims=16, x_T=x_T, step_noise=noise)
    _close(tmrp, g["tmrp"], atol=2e-5)
    _close(logit, g["logit"], atol=2e-5)


def test_config5_end_to_end(fpc_spec):
    """BASELINE configs[4]: 4096-point partial cloud, 1000 DDPM steps, G = 200 (oracle vs the reference's vectors)."""
    g = load_golden("c5_ldm_e2e.npz")
    schema = dict(load_schema("schema_fpc_ldm.json"))
    schema["vae_model.encoder.pc_encoder.out_layer.1.weight"] = ((64, 4096), torch.float32)
    sd = synthetic_state_dict(schema, seed=0)
    G = int(g["num_grasps"])
    x_T, noise = _ddpm_noise(int(g["seed"]), G, 4)
    tmrp, logit = R.ldm_generate(sd, g["pc"], G, R.make_scheduler("ddpm"), fpc_spec, x_T=x_T, step_noise=noise)
    _close(tmrp, g["tmrp"], atol=2e-5)
    _close(logit, g["logit"], atol=2e-5)
    out = R.pose_epilogue(tmrp, logit, dict(grasp_mean=g["grasp_mean"], grasp_std=g["grasp_std"]), 1, G)
    _close(out["grasps"], g["H"], atol=2e-5)
