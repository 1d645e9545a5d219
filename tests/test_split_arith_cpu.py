"""CPU: the split-bf16 arithmetic of the HIP GEMMs, restated in torch (every f32 operand = hi + mid + lo, three bf16
numbers; the six partial products of weight >= 2^-16; f32 accumulation), reproduces the vectors captured from the
reference to the accuracy of an f32 fma chain -- and no package module imports the oracle."""
import os
import re

import torch
import torch.nn.functional as F

from conftest import load_golden
from oracle import torch_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]   # (weight plane, activation plane)


def _split3(x):
    hi = x.to(torch.bfloat16).float()
    r = x - hi
    mid = r.to(torch.bfloat16).float()
    lo = (r - mid).to(torch.bfloat16).float()
    return hi, mid, lo


def test_split_is_exact_and_planes_shrink():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, generator=g) * torch.logspace(-20, 20, 4096)
    hi, mid, lo = _split3(x)
    assert torch.equal(hi + mid + lo, x)                       # 3 x 8 significant bits cover the 24 of an f32
    assert (mid.abs() <= hi.abs() * 2.0 ** -8 * 1.01).all() and (lo.abs() <= hi.abs() * 2.0 ** -16 * 1.01).all()


def test_six_partial_products_reproduce_the_reference_denoiser(fpc_state_dict, monkeypatch):
    real = F.conv1d

    def conv1d_split(x, w, b=None, stride=1, padding=0, **kw):
        if w.shape[1] < 16:      # init conv and the 4-channel level stay on the VALU in f32 in the engine too
            return real(x, w, b, stride=stride, padding=padding)
        xs, ws = _split3(x), _split3(w)
        out = sum(real(xs[j], ws[i], None, padding=padding) for i, j in PAIRS)
        return out + b.view(1, -1, 1) if b is not None else out

    g = load_golden("denoiser.npz")
    monkeypatch.setattr(R.F, "conv1d", conv1d_split)
    worst = 0.0
    for i, t in enumerate(g["t"].tolist()):
        tb = torch.full((8,), t, dtype=torch.long)
        eps = R.resnet1d_forward(fpc_state_dict, "diffusion_model.model.", g["x"], z_cond=g["z_cond"], time=tb)
        worst = max(worst, (eps - g["eps"][i]).abs().max().item())
    assert worst < 2e-6, worst                                  # measured 5.4e-7; the f32 oracle itself is bit-identical


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "graspldm_amd")
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            text = open(os.path.join(pkg, name)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), name
            assert "from oracle" not in text and "import oracle" not in text, name
