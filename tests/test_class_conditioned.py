"""Class-conditioned denoiser (grasp_ldm/models/modules/class_conditioned_resnet.py):
golden captured from the reference module (`python -m oracle.make_golden class_cond`)."""
import pytest
import torch

from conftest import load_golden, load_schema

ARGS = dict(dim=4, channels=1, block_channels=(32, 64, 128, 256), input_conditioning_dims=64, resnet_block_groups=4,
            dropout=0.1, is_time_conditioned=True, learned_variance=False, learned_sinusoidal_cond=False,
            random_fourier_features=True)


def _weights():
    from graspldm_amd.synthetic import synthetic_state_dict
    return synthetic_state_dict(load_schema("schema_class_denoiser.json"), seed=5)


def test_oracle_matches_reference_vectors():
    from oracle import torch_ref as R
    g, sd = load_golden("class_denoiser.npz"), _weights()
    for i, t in enumerate(g["t"].tolist()):
        e = R.resnet1d_forward(sd, "", g["x"], z_cond=g["z_cond"], time=torch.full((8,), t, dtype=torch.long),
                               cls_cond=g["cls"])
        assert (e - g["eps"][i]).abs().max() < 1e-6


def test_state_dict_schema_and_registry():
    from graspldm_amd.builder import ALL_MODELS
    m = ALL_MODELS["ClassTimeConditionedResNet1D"](**ARGS)
    m.load_state_dict(_weights(), strict=True)
    assert m.cls_embed[0].weight.shape == (16, 1)


@pytest.mark.gpu
def test_module_forward_golden():
    from graspldm_amd.resnets import ClassTimeConditionedResNet1D
    g = load_golden("class_denoiser.npz")
    m = ClassTimeConditionedResNet1D(**ARGS)
    m.load_state_dict(_weights(), strict=True)
    m = m.cuda().eval()
    for i, t in enumerate(g["t"].tolist()):
        tb = torch.full((8,), t, dtype=torch.long, device="cuda")
        eps = m(g["x"].cuda(), time=tb, z_cond=g["z_cond"].cuda(), cls_cond=g["cls"].cuda())
        assert (eps.cpu() - g["eps"][i]).abs().max() < 2e-5
        via = m(g["x"].cuda(), time=tb, z_cond=g["z_cond"].cuda(), metas={"mode_cls": g["cls"].cuda()})
        assert torch.equal(via, eps)
    with pytest.raises(AssertionError, match="Class conditioning tensor is required"):
        m(g["x"].cuda(), time=tb, z_cond=g["z_cond"].cuda(), metas={})


@pytest.mark.gpu
def test_class_conditioned_sampling_vs_oracle(fpc_state_dict):
    """20 DDIM steps through GraspLatentDDM with the class-conditioned denoiser: the label changes the result,
    and the latents match the oracle loop (1e-4)."""
    from oracle import torch_ref as R
    from graspldm_amd.inference import Conditioning, InferenceLDM  # noqa: F401
    from graspldm_amd.pipeline import fpc_model_config
    from graspldm_amd.builder import build_model_from_cfg
    from graspldm_amd.synthetic import synthetic_batch
    cfg = fpc_model_config()
    cfg["ddm"]["model"]["args"]["model"]["type"] = "ClassTimeConditionedResNet1D"
    ldm = build_model_from_cfg(cfg["ddm"])
    ldm.set_vae_model(build_model_from_cfg(cfg["vae"]))
    sd = dict(fpc_state_dict)
    csd = _weights()
    for k, v in csd.items():
        sd["diffusion_model.model." + k] = v
    ldm.load_state_dict(sd, strict=True)
    inf = InferenceLDM(model=ldm, num_inference_steps=20, device="cuda:0")
    pcs, metas = synthetic_batch(1, 1024)
    x_T = torch.randn(6, 1, 4, generator=torch.Generator().manual_seed(8))
    a = inf.generate_class_conditioned_grasps(pcs, num_grasps=6, metas=metas, class_label=2, x_T=x_T)
    b = inf.generate_class_conditioned_grasps(pcs, num_grasps=6, metas=metas, class_label=0, x_T=x_T)
    assert (a["grasp_tmrp"] - b["grasp_tmrp"]).abs().max() > 1e-3
    # oracle: same encoder latent (taken from the HIP encoder), oracle denoise loop + decoder
    z = ldm.vae_model.encode_pc(pcs.cuda()).cpu()
    zc = z.repeat_interleave(6, dim=0)
    sched = R.make_scheduler("ddim")
    sched.set_timesteps(20)
    x = x_T.clone()
    cls = torch.full((6, 1), 2.0)
    for t in reversed(range(0, 1000, 50)):
        eps = R.resnet1d_forward(csd, "", x, z_cond=zc, time=torch.full((6,), t, dtype=torch.long), cls_cond=cls)
        x = sched.step(eps, t, x).prev_sample
    tm, lg = R.decoder_forward({k: v for k, v in sd.items()}, "vae_model.decoder.", x.squeeze(-2), zc)
    exp = R.pose_epilogue(tm, lg, {k: v for k, v in metas.items() if isinstance(v, torch.Tensor)}, 1, 6)
    assert (a["grasp_tmrp"].cpu() - exp["grasp_tmrp"]).abs().max() < 1e-4
    assert (a["grasps"].cpu() - exp["grasps"]).abs().max() < 1e-4
