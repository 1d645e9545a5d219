"""Capture golden vectors from the reference's own Python.  CONTAINER-ONLY
(needs /root/reference); run from the repo root:

    python -m oracle.make_golden            # writes tests/golden/*.npz

The reference modules run as shipped (oracle/ref_import.py lists the five
stand-ins for absent third-party / CUDA-only pieces).  Weights come from the
deterministic recipe in graspldm_amd/synthetic.py (seed 0), inputs from the
same module, noise from torch.manual_seed(1234) in the reference's draw order.
Fixtures hold inputs and expected outputs only (data, no reference source).

  G2 pvcnn_encoder.npz     PVCNNEncoder (fpc config) on 2 clouds -> z[2,3,64]
  G3 denoiser.npz          TimeConditionedResNet1D.forward at 6 timesteps
  G4 decoder.npz           ConditionalGraspPoseDecoder.forward
  G5 ddim_traj.npz         GaussianDiffusion1D.sample, 100 DDIM steps (5 probes)
     ddpm_traj.npz         1000 DDPM steps, fixed_large, recorded noise
  G6 tmrp_to_H.npz         utils/rotations.tmrp_to_H on 64 poses (incl |m|~1)
  G7 ldm_e2e.npz           GraspLatentDDM.generate_grasps B=2 G=20 N=1024 + epilogue
     vae_e2e.npz           GraspCVAE.generate_grasps     B=1 G=20 (N=1024 and N=64)
  schema_*.json            state-dict key -> (shape, dtype) of the reference modules
  G8 sa_module.npz         PointNetSAModule on one cloud (SSG SA1/SA2 shapes)
     pointnet2_ssg.npz     PointNet2SSG forward on one cloud (every 8th point kept)
     c5_ldm_e2e.npz        BASELINE configs[4]: n_points=4096, one partial cloud, 1000 DDPM steps, G=200
                           (`python -m oracle.make_golden c5`)
     ppc_ldm_e2e.npz       the shipped partial-cloud experiment (z16, pc256, DDPM): denoiser / decoder forwards and
                           end to end on 2 partial clouds (`python -m oracle.make_golden ppc`)
     pvcnn2.npz            PVCNN2 forward on one cloud (every 16th point kept); `python -m oracle.make_golden pvcnn2`
                           writes only this one
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from graspldm_amd import synthetic  # noqa: E402
from oracle import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 1234


def _np(t):
    return t.detach().cpu().numpy()


def _save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (_np(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print(f"  {name:22s} {os.path.getsize(path) / 1024:8.1f} KiB")


def _schema(name, module):
    import json
    sd = module.state_dict()
    path = os.path.join(OUT, name)
    with open(path, "w") as f:
        json.dump({k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()}, f, indent=0)
    print(f"  {name:22s} {os.path.getsize(path) / 1024:8.1f} KiB  ({len(sd)} entries)")


def _cond(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 3, 64, generator=g)


@torch.no_grad()
def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref_import.install_shims()

    # ---- full LDM (DDIM flavour for G5/G7; weights are scheduler-independent)
    ldm = ref_import.build_reference_ldm(noise_scheduler_type="ddim")
    synthetic.load_synthetic_weights(ldm, seed=0)
    _schema("schema_fpc_ldm.json", ldm)
    den = ldm.diffusion_model.model
    vae = ldm.vae_model
    pcs, metas = synthetic.synthetic_batch(2, 1024)

    # G2
    z = vae.encode_pc(pcs)
    _save("pvcnn_encoder.npz", pc=pcs, z=z)

    # G3
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 1, 4, generator=g)
    zc = _cond(8, 8)
    ts = [0, 1, 10, 500, 990, 999]
    eps = torch.stack([den(x, time=torch.full((8,), t, dtype=torch.long), z_cond=zc) for t in ts])
    _save("denoiser.npz", x=x, z_cond=zc, t=np.array(ts), eps=eps)

    # G4
    g = torch.Generator().manual_seed(9)
    zh = torch.randn(8, 4, generator=g)
    tmrp, logit = vae.decoder(zh, zc)
    _save("decoder.npz", z_h=zh, z_cond=zc, tmrp=tmrp, logit=logit)

    # G5 DDIM 100 steps through the reference's own sample() loop
    ldm.set_inference_timesteps(100)
    torch.manual_seed(SEED)
    x_T = torch.randn(8, 1, 4)
    torch.manual_seed(SEED)
    x0, trace = ldm.diffusion_model.sample(z_cond=zc, batch_size=8, return_all=True, device="cpu")
    assert torch.equal(trace[0], x_T)
    probes = [1, 10, 50, 90, 100]  # trace[i] = x after i steps -> t = 990, 900, 500, 100, 0
    _save("ddim_traj.npz", x_T=x_T, z_cond=zc, probes=np.array(probes),
          x=torch.stack([trace[i] for i in probes]), x0=x0)

    # G7 LDM end to end (B=2, G=20) + the inference epilogue from rotations.py
    from grasp_ldm.utils.rotations import tmrp_to_H
    torch.manual_seed(SEED)
    (tm, lg), _ = ldm.generate_grasps(pcs, num_grasps=20, device="cpu")
    un = tm.view(2, 20, 6) * metas["grasp_std"].unsqueeze(-2) + metas["grasp_mean"].unsqueeze(-2)
    H = tmrp_to_H(un)
    _save("ldm_e2e.npz", pc=pcs, grasp_mean=metas["grasp_mean"], grasp_std=metas["grasp_std"],
          tmrp=tm, logit=lg, H=H, confidence=torch.sigmoid(lg.view(2, 20, 1)), seed=SEED)

    # G7b VAE mode, N=1024
    torch.manual_seed(SEED)
    tm_v, lg_v = vae.generate_grasps(pcs[:1], num_grasps=20)
    _save("vae_e2e.npz", pc=pcs[:1], tmrp=tm_v, logit=lg_v, seed=SEED)

    # G5b DDPM 1000 steps, fixed_large, recorded noise
    ldm_p = ref_import.build_reference_ldm(noise_scheduler_type="ddpm")
    synthetic.load_synthetic_weights(ldm_p, seed=0)
    torch.manual_seed(SEED)
    noise = [torch.randn(4, 1, 4)]
    for _ in range(999):
        noise.append(torch.randn(4, 1, 4))
    torch.manual_seed(SEED)
    x0p, trace_p = ldm_p.diffusion_model.sample(z_cond=zc[:4], batch_size=4, return_all=True, device="cpu")
    assert torch.equal(trace_p[0], noise[0])
    probes_p = [1, 100, 500, 900, 1000]
    _save("ddpm_traj.npz", x_T=noise[0], step_noise=torch.stack(noise[1:]), z_cond=zc[:4],
          probes=np.array(probes_p), x=torch.stack([trace_p[i] for i in probes_p]), x0=x0p)

    # G7c VAE mode with a 64-point encoder (literal BASELINE config 1)
    ldm64 = ref_import.build_reference_ldm(n_points=64)
    synthetic.load_synthetic_weights(ldm64, seed=0)
    _schema("schema_fpc_ldm_n64.json", ldm64)
    pc64, _ = synthetic.synthetic_batch(1, 64)
    torch.manual_seed(SEED)
    tm64, lg64 = ldm64.vae_model.generate_grasps(pc64, num_grasps=20)
    _save("vae_e2e_n64.npz", pc=pc64, tmrp=tm64, logit=lg64, seed=SEED)

    # G6
    g = torch.Generator().manual_seed(11)
    poses = torch.randn(64, 6, generator=g)
    poses[:8, 3:] = poses[:8, 3:] / poses[:8, 3:].norm(dim=-1, keepdim=True)  # |m| = 1
    poses[8:12, 3:] = 0
    _save("tmrp_to_H.npz", tmrp=poses, H=tmrp_to_H(poses))

    # G8 set abstraction
    from grasp_ldm.models.modules.ext.pvcnn.modules.pointnet import PointNetSAModule
    from grasp_ldm.models.modules.ext.pvcnn.pointnet2 import PointNet2SSG
    cloud = pcs[:1].transpose(1, 2).contiguous() * 0.05 / 0.12  # ~unit-scale coords so the radii bite
    sa1 = PointNetSAModule(num_centers=512, radius=0.2, num_neighbors=64, in_channels=0, out_channels=(64, 64, 128)).eval()
    synthetic.load_synthetic_weights(sa1, seed=1)
    _schema("schema_sa1.json", sa1)
    f1, c1 = sa1((None, cloud))
    sa2 = PointNetSAModule(num_centers=128, radius=0.4, num_neighbors=64, in_channels=128, out_channels=(128, 128, 256)).eval()
    synthetic.load_synthetic_weights(sa2, seed=2)
    _schema("schema_sa2.json", sa2)
    f2, c2 = sa2((f1, c1))
    _save("sa_module.npz", coords=cloud, f1=f1[:, :, ::4], c1=c1, f2=f2, c2=c2)
    ssg = PointNet2SSG(extra_feature_channels=0).eval()
    synthetic.load_synthetic_weights(ssg, seed=3)
    _schema("schema_pointnet2_ssg.json", ssg)
    _save("pointnet2_ssg.npz", coords=cloud, out=ssg(cloud)[:, :, ::8])
    pvcnn2_golden(cloud)
    print("golden fixtures written to", OUT)


@torch.no_grad()
def pvcnn2_golden(cloud=None):
    """G8: PVCNN2 (set abstraction + PVConv + feature propagation, pvcnn_base.py:147-279) on one cloud."""
    from grasp_ldm.models.modules.ext.pvcnn.pvcnn_base import PVCNN2
    if cloud is None:
        pcs, _ = synthetic.synthetic_batch(2, 1024)
        cloud = pcs[:1].transpose(1, 2).contiguous() * 0.05 / 0.12
    net = PVCNN2().eval()
    synthetic.load_synthetic_weights(net, seed=4)
    _schema("schema_pvcnn2.json", net)
    _save("pvcnn2.npz", coords=cloud, out=net(cloud)[:, :, ::16])


def front_end_golden():
    """the reference's own PointCloudHelpers (utils/pointcloud_helpers.py; pure numpy / torch, imported
    with the trimesh stand-in) on seeded clouds: greedy farthest points, both branches of regularize_pc_point_count
    (np.random seeded), regularize_pointcloud (torch seeded)."""
    ref_import.install_shims()
    from grasp_ldm.utils.pointcloud_helpers import PointCloudHelpers as P
    out = {}
    rng = np.random.RandomState(7)
    cases = [(300, 64), (1500, 1024), (4096, 1024), (2500, 2500)]
    for i, (n, m) in enumerate(cases):
        pc = (rng.standard_normal((n, 3)) * np.array([0.08, 0.05, 0.02]) + np.array([0.3, -0.1, 0.6])).astype(np.float32)
        if i == 0:
            pc[10:20] = pc[0:10]  # duplicates: zero distances and ties
        _, centers = P.farthest_points(pc, m, P.distance_by_translation_point, return_center_indexes=True)
        out[f"fps{i}_pc"], out[f"fps{i}_idx"] = pc, centers.astype(np.int32)
    pc = out["fps1_pc"]
    out["reg_fps"] = P.regularize_pc_point_count(pc, 1024, use_farthest_point=True)
    np.random.seed(11)
    out["reg_down"] = P.regularize_pc_point_count(pc, 1024, use_farthest_point=False)
    np.random.seed(12)
    out["reg_up"] = P.regularize_pc_point_count(out["fps0_pc"], 1024)
    torch.manual_seed(13)
    out["regt_up"] = P.regularize_pointcloud(torch.from_numpy(out["fps0_pc"]), 1024).numpy()
    torch.manual_seed(14)
    out["regt_down"] = P.regularize_pointcloud(torch.from_numpy(pc), 1024).numpy()
    _save("front_end.npz", **out)


def class_cond_golden():
    """the reference's ClassTimeConditionedResNet1D (class_conditioned_resnet.py) with the fpc
    denoiser arguments, recipe weights (seed 5), 8 samples, labels 0..3, at t = 0, 500, 999."""
    import json
    ref_import.install_shims()
    from grasp_ldm.models.modules.class_conditioned_resnet import ClassTimeConditionedResNet1D
    m = ClassTimeConditionedResNet1D(dim=4, channels=1, block_channels=(32, 64, 128, 256), input_conditioning_dims=64,
                                     resnet_block_groups=4, dropout=0.1, is_time_conditioned=True,
                                     learned_variance=False, learned_sinusoidal_cond=False, random_fourier_features=True)
    synthetic.load_synthetic_weights(m, seed=5)
    m.eval()
    _schema("schema_class_denoiser.json", m)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(8, 1, 4, generator=g)
    zc = torch.randn(8, 3, 64, generator=g)
    cls = torch.tensor([0, 1, 2, 3, 3, 2, 1, 0], dtype=torch.float32).reshape(8, 1)
    ts = [0, 500, 999]
    with torch.no_grad():
        eps = torch.stack([m(x, time=torch.full((8,), t, dtype=torch.long), z_cond=zc, cls_cond=cls) for t in ts])
        via_metas = m(x, time=torch.full((8,), 500, dtype=torch.long), z_cond=zc, metas={"mode_cls": cls})
    assert torch.equal(via_metas, eps[1])
    _save("class_denoiser.npz", x=x, z_cond=zc, cls=cls, t=np.array(ts), eps=eps)


def dpmpp_golden():
    """the reference's ElucidatedDiffusion.sample_using_dpmpp (pure reference code: elucidated_diffusion.py
    + resnets.py) around the fpc denoiser with recipe weights (seed 0), 8 latents, 20 steps, with and without clamp."""
    ref_import.install_shims()
    from grasp_ldm.models.diffusion.elucidated_diffusion import ElucidatedDiffusion
    from grasp_ldm.models.modules.resnets import TimeConditionedResNet1D
    net = TimeConditionedResNet1D(dim=4, channels=1, block_channels=(32, 64, 128, 256), input_conditioning_dims=64,
                                  resnet_block_groups=4, dropout=0.1, is_time_conditioned=True, learned_variance=False,
                                  learned_sinusoidal_cond=False, random_fourier_features=True)
    # the denoiser weights of the full fpc recipe state dict (seed 0), so tests can use their `fpc_state_dict` fixture
    import json
    with open(os.path.join(OUT, "schema_fpc_ldm.json")) as f:
        schema = {k: (tuple(sh), getattr(torch, dt)) for k, (sh, dt) in json.load(f).items()}
    full = synthetic.synthetic_state_dict(schema, seed=0)
    pre = "diffusion_model.model."
    net.load_state_dict({k[len(pre):]: v for k, v in full.items() if k.startswith(pre)}, strict=True)
    ed = ElucidatedDiffusion(net=net, seq_length=4).eval()
    zc = _cond(8, 41)
    out = {"z_cond": zc}
    for name, clamp in (("plain", False), ("clamp", True)):
        torch.manual_seed(SEED)
        noise = torch.randn(8, 1, 4)
        torch.manual_seed(SEED)
        x, all_x = ed.sample(use_dpmpp=True, batch_size=8, z_cond=zc, num_sample_steps=20, clamp=clamp, return_all=False)
        assert torch.equal(all_x[0], ed.sample_schedule(20)[0] * noise)
        out["noise"], out["x_" + name] = noise, x
    # the stochastic Heun sampler (sample_normal), 8 steps; the draws are recorded in the reference's order
    torch.manual_seed(SEED + 1)
    h_noise = torch.randn(8, 1, 4)
    h_steps = torch.stack([torch.randn(8, 1, 4) for _ in range(8)])
    torch.manual_seed(SEED + 1)
    xh, _ = ed.sample(use_dpmpp=False, batch_size=8, z_cond=zc, num_sample_steps=8, clamp=False, return_all=False)
    out["heun_noise"], out["heun_step_noise"], out["x_heun"] = h_noise, h_steps, xh
    _save("dpmpp.npz", **out)


@torch.no_grad()
def c5_golden():
    """BASELINE.json configs[4] on the reference's own graph: the fpc LDM rebuilt with a 4096-point encoder and the
    DDPM scheduler (fixed_large), ONE partial synthetic cloud (camera-facing side, resampled to 4096 points with
    duplicates like regularize_pc_point_count), G = 200 grasps, the full 1000-step loop of
    GaussianDiffusion1D.sample (gaussian_diffusion.py:232-277) + the decoder + the epilogue of rotations.py.
    The noise is NOT stored (3.2 MB): it is torch.manual_seed(SEED) followed by the reference's draw order
    (x_T [200,1,4], then one [200,1,4] draw per step with t > 0), which the test regenerates on the CPU."""
    ref_import.install_shims()
    import json
    ldm = ref_import.build_reference_ldm(noise_scheduler_type="ddpm", n_points=4096)
    synthetic.load_synthetic_weights(ldm, seed=0)
    _schema("schema_fpc_ldm_n4096.json", ldm)
    pcs, metas = synthetic.synthetic_batch(1, 4096, partial=True, first_index=50)
    from grasp_ldm.utils.rotations import tmrp_to_H
    G = 200
    torch.manual_seed(SEED)
    (tm, lg), _ = ldm.generate_grasps(pcs, num_grasps=G, device="cpu")
    un = tm.view(1, G, 6) * metas["grasp_std"].unsqueeze(-2) + metas["grasp_mean"].unsqueeze(-2)
    H = tmrp_to_H(un)
    z = ldm.vae_model.encode_pc(pcs)
    _save("c5_ldm_e2e.npz", pc=pcs, grasp_mean=metas["grasp_mean"], grasp_std=metas["grasp_std"], z=z,
          tmrp=tm, logit=lg, H=H, confidence=torch.sigmoid(lg.view(1, G, 1)), seed=SEED, num_grasps=G)


@torch.no_grad()
def ppc_golden():
    """The reference's second shipped experiment, the partial-cloud config
    (configs/generation/partial_pc/ppc_1a_..._latentc3_z16_pc256_180k.py): 16-dim grasp latent (the denoiser runs on
    16 positions), 256-dim cloud latent (3 x 256 conditioning), DDPM fixed_large, 1000 steps.  Two partial clouds,
    G = 10, full loop + decoder + epilogue; plus single denoiser forwards at 4 timesteps and the encoder latent.
    Noise: torch.manual_seed(SEED), the reference's draw order (regenerated by the test)."""
    ref_import.install_shims()
    rel = "configs/generation/partial_pc/ppc_1a_partial_63cat8k_filtered_latentc3_z16_pc256_180k.py"
    ldm = ref_import.build_reference_ldm(rel)
    synthetic.load_synthetic_weights(ldm, seed=0)
    _schema("schema_ppc_ldm.json", ldm)
    assert ldm.diffusion_model._noise_scheduler_type == "ddpm"
    pcs, metas = synthetic.synthetic_batch(2, 1024, partial=True, first_index=60)
    from grasp_ldm.utils.rotations import tmrp_to_H
    G = 10
    z = ldm.vae_model.encode_pc(pcs)
    den = ldm.diffusion_model.model
    g = torch.Generator().manual_seed(21)
    x = torch.randn(6, 1, 16, generator=g)
    zc = torch.randn(6, 3, 256, generator=g)
    ts = [0, 7, 500, 999]
    eps = torch.stack([den(x, time=torch.full((6,), t, dtype=torch.long), z_cond=zc) for t in ts])
    zh = torch.randn(6, 16, generator=g)
    d_tm, d_lg = ldm.vae_model.decoder(zh, zc)
    torch.manual_seed(SEED)
    (tm, lg), _ = ldm.generate_grasps(pcs, num_grasps=G, device="cpu")
    un = tm.view(2, G, 6) * metas["grasp_std"].unsqueeze(-2) + metas["grasp_mean"].unsqueeze(-2)
    _save("ppc_ldm_e2e.npz", pc=pcs, grasp_mean=metas["grasp_mean"], grasp_std=metas["grasp_std"], z=z,
          den_x=x, den_zc=zc, den_t=np.array(ts), den_eps=eps, dec_zh=zh, dec_tmrp=d_tm, dec_logit=d_lg,
          tmrp=tm, logit=lg, H=tmrp_to_H(un), confidence=torch.sigmoid(lg.view(2, G, 1)), seed=SEED, num_grasps=G)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dpmpp":
        os.makedirs(OUT, exist_ok=True)
        dpmpp_golden()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "class_cond":
        os.makedirs(OUT, exist_ok=True)
        class_cond_golden()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in ("c5", "ppc"):
        os.makedirs(OUT, exist_ok=True)
        torch.set_num_threads(8)
        {"c5": c5_golden, "ppc": ppc_golden}[sys.argv[1]]()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "front_end":
        os.makedirs(OUT, exist_ok=True)
        front_end_golden()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pvcnn2":  # add this one fixture without rewriting the others
        os.makedirs(OUT, exist_ok=True)
        torch.set_num_threads(8)
        ref_import.install_shims()
        pvcnn2_golden()
    else:
        main()
