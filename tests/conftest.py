import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU skips the `gpu` tests instead of failing them (the HIP path has no
    CPU fallback, so they cannot run there)."""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no HIP device visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: torch.from_numpy(z[k]) if z[k].dtype != object else z[k] for k in z.files}


def load_schema(name):
    with open(os.path.join(GOLDEN, name)) as f:
        raw = json.load(f)
    return {k: (tuple(shape), getattr(torch, dt)) for k, (shape, dt) in raw.items()}


@pytest.fixture(scope="session")
def fpc_state_dict():
    """Synthetic-recipe weights (seed 0) for the full fpc GraspLatentDDM schema."""
    from graspldm_amd.synthetic import synthetic_state_dict
    return synthetic_state_dict(load_schema("schema_fpc_ldm.json"), seed=0)


@pytest.fixture(scope="session")
def fpc_spec():
    from oracle.torch_ref import pvcnn_block_spec
    return pvcnn_block_spec(0.75, 0.75)
