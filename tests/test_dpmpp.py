"""ElucidatedDiffusion / DPM-Solver++(2M) sampler (elucidated_diffusion.py:259-313): golden
captured from the reference's own sampler + denoiser (`python -m oracle.make_golden dpmpp`).  fp32 tolerance on the
sampled latents: 1e-4 (20 network evaluations; the latents are O(1))."""
import pytest
import torch

from conftest import load_golden


def test_oracle_matches_reference_vectors(fpc_state_dict):
    from oracle import torch_ref as R
    g = load_golden("dpmpp.npz")
    for name, clamp in (("plain", False), ("clamp", True)):
        x = R.dpmpp_sample(fpc_state_dict, "diffusion_model.model.", g["z_cond"], g["noise"], 20, clamp)
        assert (x - g["x_" + name]).abs().max() < 1e-6


def test_oracle_heun_matches_reference_vectors(fpc_state_dict):
    from oracle import torch_ref as R
    g = load_golden("dpmpp.npz")
    x = R.heun_sample(fpc_state_dict, "diffusion_model.model.", g["z_cond"], g["heun_noise"], g["heun_step_noise"], 8)
    assert (x - g["x_heun"]).abs().max() < 1e-6


def test_tables_follow_the_reference_formulas():
    from graspldm_amd.elucidated import ElucidatedDiffusion

    class _Net(torch.nn.Module):
        random_or_learned_sinusoidal_cond = True
    ed = ElucidatedDiffusion(_Net(), seq_length=4)
    sig, times, coef = ed.dpmpp_tables(20)
    assert sig.shape == (21,) and sig[-1] == 0 and abs(sig[0].item() - 80.0) < 1e-3 and abs(sig[-2].item() - 0.002) < 1e-6
    assert coef[0, 7] == 0 and coef[-1, 7] == 0 and coef[1:-1, 7].eq(1).all()
    assert coef[-1, 5] == 0 and coef[-1, 6] == -1          # sigma_next = 0: x' = denoised
    assert torch.allclose(times, torch.log(sig[:-1]) * 0.25)
    with pytest.raises(KeyError):
        ed.sample(batch_size=1)                             # the reference pops use_dpmpp without a default


@pytest.mark.gpu
@pytest.mark.parametrize("name,clamp", [("plain", False), ("clamp", True)])
def test_fused_dpmpp_golden(fpc_state_dict, name, clamp):
    from graspldm_amd.elucidated import ElucidatedDiffusion
    from graspldm_amd.resnets import TimeConditionedResNet1D
    g = load_golden("dpmpp.npz")
    net = TimeConditionedResNet1D(dim=4, channels=1, block_channels=(32, 64, 128, 256), input_conditioning_dims=64,
                                  resnet_block_groups=4, dropout=0.1, is_time_conditioned=True, learned_variance=False,
                                  learned_sinusoidal_cond=False, random_fourier_features=True)
    pre = "diffusion_model.model."
    net.load_state_dict({k[len(pre):]: v for k, v in fpc_state_dict.items() if k.startswith(pre)}, strict=True)
    ed = ElucidatedDiffusion(net=net, seq_length=4).cuda().eval()
    x, first = ed.sample(use_dpmpp=True, batch_size=8, z_cond=g["z_cond"].cuda(), num_sample_steps=20, clamp=clamp,
                         noise=g["noise"])
    assert (x.cpu() - g["x_" + name]).abs().max() < 1e-4, (x.cpu() - g["x_" + name]).abs().max()
    assert torch.allclose(first[0].cpu(), 80.0 * g["noise"], rtol=1e-6)
    # return_all=True (elucidated_diffusion.py:306: every x is kept): the per-step form, same bits as the fused launch
    x2, trace = ed.sample(use_dpmpp=True, batch_size=8, z_cond=g["z_cond"].cuda(), num_sample_steps=20, clamp=clamp,
                          noise=g["noise"], return_all=True)
    assert len(trace) == 21 and torch.equal(trace[-1], x2)
    assert (x2 - x).abs().max() <= 1e-6, (x2 - x).abs().max()


@pytest.mark.gpu
def test_ldm_with_elucidated_diffusion_end_to_end(fpc_state_dict):
    """GraspLatentDDM(elucidated_diffusion=True).generate_grasps(use_dpmpp=True, num_sample_steps=...) (grasp_ldm.py:58-62,
    215-220) against the oracle loop + decoder; more tiles than workgroup slots (whole-tile distribution, no chains)."""
    from oracle import torch_ref as R
    from graspldm_amd.builder import build_model_from_cfg
    from graspldm_amd.pipeline import fpc_model_config
    from graspldm_amd.synthetic import synthetic_batch
    cfg = fpc_model_config()
    cfg["ddm"]["model"]["args"]["elucidated_diffusion"] = True
    ldm = build_model_from_cfg(cfg["ddm"])
    ldm.set_vae_model(build_model_from_cfg(cfg["vae"]))
    sd = {k.replace("diffusion_model.model.", "diffusion_model.net."): v for k, v in fpc_state_dict.items()}
    ldm.load_state_dict(sd, strict=True)
    ldm = ldm.cuda().eval()
    pcs, _ = synthetic_batch(2, 1024)
    noise = torch.randn(8, 1, 4, generator=torch.Generator().manual_seed(2))
    (tm, lg), _ = ldm.generate_grasps(pcs.cuda(), num_grasps=4, use_dpmpp=True, num_sample_steps=12, noise=noise,
                                      metas={"ignored": True})
    z = ldm.vae_model.encode_pc(pcs.cuda()).cpu()
    zc = z.repeat_interleave(4, dim=0)
    x = R.dpmpp_sample(fpc_state_dict, "diffusion_model.model.", zc, noise, 12, False)
    etm, elg = R.decoder_forward(fpc_state_dict, "vae_model.decoder.", x.squeeze(-2), zc)
    assert (tm.cpu() - etm).abs().max() < 1e-4 and (lg.cpu() - elg).abs().max() < 1e-4
    # a batch beyond one round of workgroups: same bits as the small batch for the shared samples
    n = 2 * torch.cuda.get_device_properties(0).multi_processor_count * 8 + 40
    ed = ldm.diffusion_model
    big_noise = torch.randn(n, 1, 4, generator=torch.Generator().manual_seed(3))
    zc_big = torch.randn(n, 3, 64, generator=torch.Generator().manual_seed(4)).cuda()
    xb, _ = ed.sample(use_dpmpp=True, batch_size=n, z_cond=zc_big, num_sample_steps=6, noise=big_noise)
    xs, _ = ed.sample(use_dpmpp=True, batch_size=48, z_cond=zc_big[-48:].contiguous(), num_sample_steps=6,
                      noise=big_noise[-48:])
    assert torch.equal(xb[-48:], xs)


@pytest.mark.gpu
def test_inference_ldm_elucidated_mode(fpc_state_dict):
    """InferenceLDM(use_elucidated=True): DPM++ with 32 steps by default (tools/inference.py:472-477,607-609)."""
    from graspldm_amd.builder import build_model_from_cfg
    from graspldm_amd.inference import InferenceLDM
    from graspldm_amd.pipeline import fpc_model_config
    from graspldm_amd.synthetic import synthetic_batch
    cfg = fpc_model_config()
    cfg["ddm"]["model"]["args"]["elucidated_diffusion"] = True
    ldm = build_model_from_cfg(cfg["ddm"])
    ldm.set_vae_model(build_model_from_cfg(cfg["vae"]))
    ldm.load_state_dict({k.replace("diffusion_model.model.", "diffusion_model.net."): v for k, v in fpc_state_dict.items()},
                        strict=True)
    inf = InferenceLDM(model=ldm, use_elucidated=True, device="cuda:0")
    assert inf.fast_sampler == "DPMPP" and inf.num_inference_steps == 32
    pcs, metas = synthetic_batch(1, 1024)
    noise = torch.randn(5, 1, 4, generator=torch.Generator().manual_seed(6))
    a = inf.generate_grasps(pcs, metas, num_grasps=5, noise=noise)
    b = inf.generate_grasps(pcs, metas, num_grasps=5, noise=noise)
    assert a["grasps"].shape == (1, 5, 4, 4) and torch.isfinite(a["grasps"]).all()
    assert (a["grasps"] - b["grasps"]).abs().max() < 1e-5
    # return_intermediate: the reference decodes 50 of the per-step latents (grasp_ldm.py:223-227)
    c = inf.generate_grasps(pcs, metas, num_grasps=5, noise=noise, return_intermediate=True)
    assert len(c["all_steps_grasps"]) == 50 and (c["grasps"] - a["grasps"]).abs().max() < 1e-5
    assert (c["all_steps_grasps"][-1].cuda() - c["grasps"]).abs().max() < 1e-5
    # use_fast_sampler=False: the stochastic Heun sampler (elucidated_diffusion.py:177-257)
    heun = InferenceLDM(model=ldm, use_elucidated=True, use_fast_sampler=False, num_inference_steps=6, device="cuda:0")
    assert heun.fast_sampler == "HEUN"
    torch.manual_seed(1)
    h = heun.generate_grasps(pcs, metas, num_grasps=5, noise=noise)
    assert h["grasps"].shape == (1, 5, 4, 4) and torch.isfinite(h["grasps"]).all()


@pytest.mark.gpu
def test_heun_sampler_golden(fpc_state_dict):
    """ElucidatedDiffusion.sample_normal (stochastic Heun, elucidated_diffusion.py:177-257), 8 steps = 15 network
    evaluations through the engine, noise draws from the golden: 1e-4 on latents of magnitude ~2."""
    from graspldm_amd.elucidated import ElucidatedDiffusion
    from graspldm_amd.resnets import TimeConditionedResNet1D
    g = load_golden("dpmpp.npz")
    net = TimeConditionedResNet1D(dim=4, channels=1, block_channels=(32, 64, 128, 256), input_conditioning_dims=64,
                                  resnet_block_groups=4, dropout=0.1, is_time_conditioned=True, learned_variance=False,
                                  learned_sinusoidal_cond=False, random_fourier_features=True)
    pre = "diffusion_model.model."
    net.load_state_dict({k[len(pre):]: v for k, v in fpc_state_dict.items() if k.startswith(pre)}, strict=True)
    ed = ElucidatedDiffusion(net=net, seq_length=4).cuda().eval()
    x, trace = ed.sample(use_dpmpp=False, batch_size=8, z_cond=g["z_cond"].cuda(), num_sample_steps=8,
                         noise=g["heun_noise"], step_noise=g["heun_step_noise"], return_all=True)
    assert len(trace) == 9
    assert (x.cpu() - g["x_heun"]).abs().max() < 1e-4, (x.cpu() - g["x_heun"]).abs().max()
