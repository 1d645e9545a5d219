"""CPU: the split-f16 arithmetic of the HIP GEMMs, restated in torch (every f32 operand = hi + lo, two f16 numbers:
11 + 11 significant bits; the three partial products hi hi, hi lo, lo hi; f32 accumulation), reproduces the vectors
captured from the reference to the accuracy of an f32 fma chain -- and no package module imports the oracle."""
import os
import re

import torch
import torch.nn.functional as F

from conftest import load_golden
from oracle import torch_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = [(0, 1), (1, 0), (0, 0)]   # (weight plane, activation plane), small terms first as the kernels issue them


def _split2(x):
    hi = x.to(torch.float16).float()      # f16 subnormals kept: the matrix pipe keeps them (tools/micro/mfma_f16_split)
    lo = (x - hi).to(torch.float16).float()
    return hi, lo


def test_split_is_tight_and_lo_shrinks():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, generator=g) * torch.logspace(-8, 4, 4096)
    hi, lo = _split2(x)
    # 22 significant bits where lo is a normal f16, an absolute 2^-25 where it is subnormal
    assert ((hi + lo - x).abs() <= torch.maximum(x.abs() * 2.0 ** -22, torch.full_like(x, 2.0 ** -25))).all()
    assert (lo.abs() <= hi.abs() * 2.0 ** -11 * 1.01 + 2.0 ** -25).all()


def test_three_partial_products_reproduce_the_reference_denoiser(fpc_state_dict, monkeypatch):
    real = F.conv1d

    def conv1d_split(x, w, b=None, stride=1, padding=0, **kw):
        if w.shape[1] < 16:      # init conv and the 4-channel level stay on the VALU in f32 in the engine too
            return real(x, w, b, stride=stride, padding=padding)
        assert float(x.abs().max()) < 65504 and float(w.abs().max()) < 65504   # the f16 range
        xs, ws = _split2(x), _split2(w)
        out = sum(real(xs[j], ws[i], None, padding=padding) for i, j in PAIRS)
        return out + b.view(1, -1, 1) if b is not None else out

    g = load_golden("denoiser.npz")
    monkeypatch.setattr(R.F, "conv1d", conv1d_split)
    worst = 0.0
    for i, t in enumerate(g["t"].tolist()):
        tb = torch.full((8,), t, dtype=torch.long)
        eps = R.resnet1d_forward(fpc_state_dict, "diffusion_model.model.", g["x"], z_cond=g["z_cond"], time=tb)
        worst = max(worst, (eps - g["eps"][i]).abs().max().item())
    assert worst < 5e-6, worst                                  # measured 1.7e-6; the f32 oracle itself is bit-identical


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "graspldm_amd")
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            text = open(os.path.join(pkg, name)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), name
            assert "from oracle" not in text and "import oracle" not in text, name
