"""CLI surface: the reference's flags parse (CPU); an end-to-end synthetic run on the GPU."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cli():
    spec = importlib.util.spec_from_file_location("generate_grasps_cli", os.path.join(ROOT, "tools", "generate_grasps.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reference_flags_parse():
    a = _cli().parse_args(["--exp_path", "output/exp", "--data_root", "d", "--mode", "LDM", "--split", "test",
                           "--num_grasps", "7", "--no_ema", "--num_samples", "3", "--conditioning", "unconditional",
                           "--inference_steps", "50", "--visualize"])
    assert a.mode == "LDM" and a.num_grasps == 7 and a.use_ema_model is False and a.inference_steps == 50
    d = _cli().parse_args(["--exp_path", "x"])
    assert d.mode == "VAE" and d.num_grasps == 20 and d.num_samples == 11 and d.inference_steps == 100


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["LDM", "VAE"])
def test_cli_synthetic_run(mode, tmp_path):
    import numpy as np
    out = str(tmp_path / "g.npz")
    res = _cli().main(["--synthetic", "1024", "--mode", mode, "--num_samples", "2", "--num_grasps", "5",
                       "--inference_steps", "10", "--seed", "3", "--out", out])
    assert len(res) == 2 and res[0]["grasps"].shape == (1, 5, 4, 4)
    z = np.load(out)
    H = z["grasps"]
    assert H.shape == (2, 5, 4, 4) and np.isfinite(H).all()
    R = H[..., :3, :3]
    assert np.allclose(R @ np.swapaxes(R, -1, -2), np.eye(3), atol=1e-4)  # proper rotations
    assert ((z["confidence"] > 0) & (z["confidence"] < 1)).all()


def test_pc_file_flags_parse():
    a = _cli().parse_args(["--exp_path", "x", "--mode", "LDM", "--pc_file", "a.npy", "--pc_file", "b.ply", "--num_points", "1024"])
    assert a.pc_file == ["a.npy", "b.ply"] and a.num_points == 1024 and not a.random_resample


def test_cloud_file_readers(tmp_path):
    """.npy / .npz / ascii and binary .ply / .xyz all give the same float32 [N,3]; non-finite rows are dropped."""
    import struct
    import numpy as np
    from graspldm_amd.pointcloud import read_cloud_file
    rng = np.random.RandomState(0)
    pts = rng.standard_normal((50, 3)).astype(np.float32)
    np.save(tmp_path / "c.npy", np.concatenate([pts, rng.rand(50, 3).astype(np.float32)], axis=1))  # xyz + rgb columns
    np.savez(tmp_path / "c.npz", points=pts)
    np.savetxt(tmp_path / "c.xyz", pts, fmt="%.9g")
    with open(tmp_path / "a.ply", "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment test\nelement vertex 50\nproperty float x\nproperty float y\n"
                "property float z\nproperty uchar red\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n")
        for p in pts:
            f.write("%.9g %.9g %.9g 7\n" % tuple(p))
    with open(tmp_path / "b.ply", "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 50\nproperty double x\nproperty double y\n"
                b"property double z\nproperty uchar intensity\nend_header\n")
        for p in pts:
            f.write(struct.pack("<dddB", float(p[0]), float(p[1]), float(p[2]), 9))
    for name in ("c.npy", "c.npz", "c.xyz", "a.ply", "b.ply"):
        got = read_cloud_file(str(tmp_path / name))
        assert got.dtype == np.float32 and got.shape == (50, 3) and np.array_equal(got, pts), name
    bad = pts.copy()
    bad[3, 1] = np.nan
    np.save(tmp_path / "n.npy", bad)
    assert read_cloud_file(str(tmp_path / "n.npy")).shape == (49, 3)
    with pytest.raises(ValueError):
        np.save(tmp_path / "w.npy", pts[:, :2])
        read_cloud_file(str(tmp_path / "w.npy"))


@pytest.mark.gpu
def test_cli_on_experiment_dir_with_pc_file(tmp_path, fpc_state_dict):
    """The CLI on a real experiment layout (Lightning-style directory written by tests/test_checkpoint.py) and a raw
    sensor cloud file: == InferenceLDM.generate_on_pointcloud on the same cloud and seed, and == the CPU oracle run on
    the regularised, normalised cloud (<= 1e-4 on H entries)."""
    import numpy as np
    import torch
    from test_checkpoint import _write_experiment
    from graspldm_amd.inference import InferenceLDM
    from graspldm_amd.synthetic import synthetic_cloud
    from oracle import front_end as F
    from oracle import torch_ref as R
    ema = {k: (v + 0.01 if v.is_floating_point() else v) for k, v in fpc_state_dict.items()}
    _write_experiment(str(tmp_path), "exp_cli", fpc_state_dict, ema)
    raw = synthetic_cloud(77, 1500).numpy() + np.array([0.4, -0.2, 0.7], dtype=np.float32)   # 1500 points, off-centre
    np.save(tmp_path / "cloud.npy", raw)
    argv = ["--exp_path", str(tmp_path / "exp_cli"), "--mode", "LDM", "--num_grasps", "6", "--inference_steps", "20",
            "--pc_file", str(tmp_path / "cloud.npy"), "--seed", "5", "--out", str(tmp_path / "o.npz")]
    res = _cli().main(argv)
    assert len(res) == 1 and res[0]["grasps"].shape == (1, 6, 4, 4)
    # the same through the Python API
    inf = InferenceLDM(exp_name="exp_cli", exp_out_root=str(tmp_path), num_inference_steps=20, use_fast_sampler=True)
    torch.manual_seed(5)
    np.random.seed(5)
    api = inf.generate_on_pointcloud(torch.from_numpy(raw), num_grasps=6, num_points=1024)
    assert torch.equal(api["grasps"], res[0]["grasps"]) and torch.equal(api["confidence"], res[0]["confidence"])
    # ... and against the CPU oracle: farthest-point regularisation, normalisation, LDM, epilogue
    reg = F.regularize_pc_point_count(raw, 1024, use_farthest_point=True)
    pcn, metas = F.normalize_input(torch.from_numpy(reg).unsqueeze(0))
    torch.manual_seed(5)
    x_T = torch.randn(6, 1, 4)
    sched = R.make_scheduler("ddim")
    sched.set_timesteps(20)
    tm, lg = R.ldm_generate(ema, pcn, 6, sched, R.pvcnn_block_spec(0.75, 0.75), x_T=x_T)
    exp = R.pose_epilogue(tm, lg, metas, 1, 6)
    assert (res[0]["grasps"].cpu() - exp["grasps"]).abs().max().item() < 1e-4
    assert (res[0]["confidence"].cpu() - exp["confidence"]).abs().max().item() < 1e-4
    z = np.load(tmp_path / "o.npz")
    assert z["grasps"].shape == (1, 6, 4, 4)
