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
