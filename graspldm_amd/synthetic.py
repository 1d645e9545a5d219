"""Deterministic synthetic inputs: weights for any state-dict schema, object
point clouds and their normalisation `metas`.

There is no network for checkpoints or ACRONYM data, so benches, tests and the
golden fixtures all use this recipe.  Values depend only on
(key name, shape, seed) -- never on module construction order -- so the
reference model (container), the CPU oracle and the HIP path get bit-identical
parameters from the same call.
"""
import math
import zlib

import torch


def _gen(key, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synthetic_tensor(key, shape, dtype=torch.float32, seed=0, tame=True):
    """One parameter/buffer tensor.  Kinds are recognised from the key suffix:
    conv/linear weights ~ N(0, 1/fan_in); biases ~ 0.1 N(0,1); norm gains
    ~ 1 + 0.1 N(0,1); BatchNorm running_mean ~ 0.1 N, running_var ~ U(0.75,1.25);
    the frozen random-Fourier `weights` ~ N(0,1) (resnets.py:44-50); the raw-timestep column
    of `time_mlp.1.weight` is scaled by 1e-3 and `final_conv.weight` by 0.1 unless tame=False (O(1) gains: the sampler
    is then expansive and only short horizons can be compared between two f32 implementations)."""
    shape = tuple(shape)
    g = _gen(key, seed)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_var":
        t = 0.75 + 0.5 * torch.rand(shape, generator=g)
    elif leaf == "running_mean":
        t = 0.1 * torch.randn(shape, generator=g)
    elif leaf == "weights":
        t = torch.randn(shape, generator=g)
    elif leaf == "g":
        t = 1.0 + 0.1 * torch.randn(shape, generator=g)
    elif leaf == "bias":
        t = 0.1 * torch.randn(shape, generator=g)
    elif leaf == "weight" and len(shape) == 1:
        t = 1.0 + 0.1 * torch.randn(shape, generator=g)
    elif leaf == "weight":
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        t = torch.randn(shape, generator=g) / math.sqrt(max(fan_in, 1))
        if tame and key.endswith("time_mlp.1.weight"):
            # input 0 of this layer is the RAW timestep 0..999 (resnets.py:52-56); a trained
            # network keeps its gain ~1/T, random O(1) gain makes the sampler chaotic
            t[:, 0] = t[:, 0] * 1e-3
        if tame and key.endswith("final_conv.weight"):
            # diffusion nets initialise / keep the output projection small; with O(1) random
            # gain the 100-step sampler amplifies a 1e-5 input change 100x (measured) and no
            # two fp32 implementations can agree to 1e-4.  x0.1 makes the map contractive.
            t = t * 0.1
    else:
        t = 0.1 * torch.randn(shape, generator=g)
    return t.to(dtype)


def synthetic_state_dict(schema, seed=0, tame=True):
    """schema: mapping key -> tensor (shape/dtype donor) or (shape, dtype)."""
    out = {}
    for k in sorted(schema):
        v = schema[k]
        shape, dtype = (v.shape, v.dtype) if hasattr(v, "shape") else v
        out[k] = synthetic_tensor(k, shape, torch.float32 if dtype.is_floating_point else dtype, seed, tame)
    return out


def load_synthetic_weights(module, seed=0):
    """Fill `module` in place (strict) and return it."""
    module.load_state_dict(synthetic_state_dict(module.state_dict(), seed), strict=True)
    return module


def _random_rotation(g):
    q = torch.randn(4, generator=g, dtype=torch.float64)
    q = q / q.norm()
    w, x, y, z = q.tolist()
    return torch.tensor([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ], dtype=torch.float64)


def synthetic_cloud(index, n_points=1024, partial=False):
    """One object-scale cloud [N,3] in metres: N points uniform on the surface of
    an axis-aligned box (half-extents U(0.03,0.12) m), randomly rotated and
    offset; generator seed 1000+index.  `partial=True` keeps the camera-facing
    side and resamples to N with replacement, like regularize_pc_point_count
    (utils/pointcloud_helpers.py:124-160)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1000 + int(index))
    half = 0.03 + 0.09 * torch.rand(3, generator=g, dtype=torch.float64)
    areas = torch.stack([half[1] * half[2], half[0] * half[2], half[0] * half[1]])
    m = n_points * (3 if partial else 1)
    face = torch.multinomial(areas.repeat_interleave(2), m, replacement=True, generator=g)
    uv = 2 * torch.rand(m, 2, generator=g, dtype=torch.float64) - 1
    axis = face // 2
    sign = (face % 2).to(torch.float64) * 2 - 1
    pts = torch.zeros(m, 3, dtype=torch.float64)
    nrm = torch.zeros(m, 3, dtype=torch.float64)
    for a in range(3):
        sel = axis == a
        o = [i for i in range(3) if i != a]
        pts[sel, a] = sign[sel] * half[a]
        pts[sel, o[0]] = uv[sel, 0] * half[o[0]]
        pts[sel, o[1]] = uv[sel, 1] * half[o[1]]
        nrm[sel, a] = sign[sel]
    R = _random_rotation(g)
    offset = 0.2 * (torch.rand(3, generator=g, dtype=torch.float64) - 0.5)
    pts = pts @ R.T + offset
    if partial:
        view = torch.randn(3, generator=g, dtype=torch.float64)
        view = view / view.norm()
        keep = ((nrm @ R.T) @ view) > 0
        vis = pts[keep]
        if vis.shape[0] == 0:
            vis = pts
        pick = torch.randint(0, vis.shape[0], (n_points,), generator=g)
        pts = vis[pick]
    return pts.to(torch.float32)


PC_STD = 0.05       # acronym_pointclouds.py:15-16,368-376  (pc / 0.05)
MRP_STD = 0.5       # grasp_std = [0.05 x3, 0.5 x3]


def normalize_cloud(pc):
    """Dataset item contract (acronym_pointclouds.py:204-245,247-288): centre on
    the mean, divide by 0.05; returns (pc_norm [N,3], metas)."""
    mean = pc.mean(dim=0)
    pc_n = (pc - mean) / PC_STD
    metas = dict(
        pc_mean=mean.clone(),
        pc_std=torch.full((3,), PC_STD),
        grasp_mean=torch.cat([mean, torch.zeros(3)]),
        grasp_std=torch.tensor([PC_STD] * 3 + [MRP_STD] * 3),
        dataset_normalized=True,
    )
    return pc_n, metas


def synthetic_batch(num_clouds, n_points=1024, partial=False, first_index=0):
    """Batched clouds [B,N,3] (normalised) + batched metas ([B,3] / [B,6])."""
    pcs, metas = [], []
    for b in range(num_clouds):
        p, m = normalize_cloud(synthetic_cloud(first_index + b, n_points, partial))
        pcs.append(p)
        metas.append(m)
    out = {k: torch.stack([m[k] for m in metas]) for k in ("pc_mean", "pc_std", "grasp_mean", "grasp_std")}
    out["dataset_normalized"] = True
    return torch.stack(pcs), out
