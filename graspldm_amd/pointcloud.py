"""Raw-cloud front end on the GPU: what a sensor cloud goes through before
`generate_grasps` -- point-count regularisation and dataset-statistics normalisation.

Mirrors `PointCloudHelpers` (grasp_ldm/utils/pointcloud_helpers.py: regularize_pointcloud :40-71,
regularize_pc_point_count :124-158, farthest_points :160-217) and `normalize_input`
(tools/inference.py:570-591).  The point arithmetic (greedy farthest-point selection, row gathers,
centring/scaling) runs in libgldm_hip.so; the RANDOM index draws stay on the host and use exactly
the generator calls the reference makes (np.random.choice / torch.randperm), so a seeded run picks
the same points.  CPU tensors are rejected like everywhere else in this package.
"""
import numpy as np
import torch

from . import _lib as L

PC_SHIFT, PC_SCALE, MRP_SCALE = 0.0, 0.05, 0.5


def _need_cuda(t, name="pc"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor (graspldm_amd has no CPU path)")


def _as_batch(pc):
    _need_cuda(pc)
    if pc.ndim not in (2, 3) or pc.shape[-1] != 3:
        raise ValueError(f"Expected point cloud to have shape (N, 3) or (B, N, 3), got {tuple(pc.shape)}.")
    return (pc.unsqueeze(0) if pc.ndim == 2 else pc).contiguous().float()


def gather_points(pc, idx):
    """pc [B,N,3], idx int [B,M] -> [B,M,3]."""
    pcb = _as_batch(pc)
    idx = idx.to(device=pcb.device, dtype=torch.int32).reshape(pcb.shape[0], -1).contiguous()
    b, n, _ = pcb.shape
    m = idx.shape[1]
    out = torch.empty((b, m, 3), dtype=torch.float32, device=pcb.device)
    if m:
        with torch.cuda.device(pcb.device):
            L.call("gldm_gather_points", L.ptr(pcb), L.ptr(idx), b, n, m, L.ptr(out), L.current_stream(pcb.device))
    return out


def farthest_point_indices(pc, nclusters):
    """Centre indices of PointCloudHelpers.farthest_points(pc, nclusters, distance_by_translation_point,
    return_center_indexes=True) (pointcloud_helpers.py:160-223) for every cloud of pc [B,N,3] / [N,3]:
    int32 [B, min(nclusters, N)]; `nclusters >= N` returns arange(N) like the reference (:185-191)."""
    pcb = _as_batch(pc)
    b, n, _ = pcb.shape
    if nclusters >= n:
        return torch.arange(n, dtype=torch.int32, device=pcb.device).unsqueeze(0).repeat(b, 1)
    idx = torch.empty((b, int(nclusters)), dtype=torch.int32, device=pcb.device)
    with torch.cuda.device(pcb.device):
        L.call("gldm_farthest_points_euclid", L.ptr(pcb), b, n, int(nclusters), L.ptr(idx), L.current_stream(pcb.device))
    return idx


class PointCloudHelpers:
    """The point-count helpers of the reference class of the same name, on CUDA tensors."""

    @staticmethod
    def farthest_points(data, nclusters, dist_func=None, return_center_indexes=True, **_):
        if dist_func is not None and getattr(dist_func, "__name__", "") != "distance_by_translation_point":
            raise NotImplementedError("only the Euclidean point distance (distance_by_translation_point) is built")
        if not return_center_indexes:
            raise NotImplementedError("cluster labels are not on the generation path; ask for the centre indexes")
        idx = farthest_point_indices(data, nclusters)
        return idx[0] if data.ndim == 2 else idx

    @staticmethod
    def regularize_pc_point_count(pc, npoints, use_farthest_point=False, rng=None):
        """pointcloud_helpers.py:124-158 on one cloud [N,3] -> [npoints,3].  `rng`: a np.random.RandomState /
        Generator-like object with `.choice`; default = the global np.random the reference draws from."""
        _need_cuda(pc)
        assert pc.ndim == 2
        rng = np.random if rng is None else rng
        n = pc.shape[0]
        if n > npoints:
            if use_farthest_point:
                idx = farthest_point_indices(pc, npoints)
            else:
                idx = torch.from_numpy(np.asarray(rng.choice(range(n), size=npoints, replace=False))).unsqueeze(0)
            return gather_points(pc, idx)[0]
        required = npoints - n
        if required > 0:
            extra = torch.from_numpy(np.asarray(rng.choice(range(n), size=required))).unsqueeze(0)
            idx = torch.cat([torch.arange(n).unsqueeze(0), extra.long()], dim=1)
            return gather_points(pc, idx)[0]
        return pc.contiguous().float()

    @staticmethod
    def regularize_pointcloud(pc, num_points):
        """pointcloud_helpers.py:40-71 on one cloud [N,3] -> [1,num_points,3]; draws torch.randperm on the
        CPU generator exactly where the reference does."""
        _need_cuda(pc)
        assert pc.ndim == 2
        n = pc.shape[0]
        if n < num_points:
            mult = max(num_points // n, 1)
            base = torch.arange(n).repeat(mult)              # pc.repeat(multiplier, 1)
            extra = num_points - base.shape[0]
            idx = torch.cat([base, base[torch.randperm(base.shape[0])[:extra]]])
        elif n > num_points:
            idx = torch.randperm(n)[:num_points]
        else:
            return pc.contiguous().float().unsqueeze(0)
        return gather_points(pc, idx.unsqueeze(0))


def _vec(v, n, device):
    t = torch.as_tensor(v, dtype=torch.float32).reshape(-1)
    return (t.expand(n) if t.numel() == 1 else t).to(device).contiguous()


def normalize_input(pc, pc_shift=PC_SHIFT, pc_scale=PC_SCALE, mrp_scale=MRP_SCALE, grasp_shift=None):
    """Centre every cloud on its mean, normalise with the dataset statistics, build the metas that
    `unnormalize_grasps` / `unnormalize_pc` need -- `normalize_input` of grasp_ldm/inference/inference_base.py:
    181-212 (statistics from set_normalization_params :103-130) and of tools/inference.py:570-591 (which reads
    class constants PC_MEAN / PC_STD / GRASP_MEAN / GRASP_STD that that file never defines; the values are the
    dataset's: shift 0, translation scale 0.05, rotation scale 0.5, acronym_pointclouds.py:15-16).  One launch.
    pc [N,3] or [B,N,3] (not modified; the reference subtracts in place) -> (pc_norm of the same rank, metas)
    with grasp_mean [B,6] and grasp_std [1,6] as in tools/inference.py:581-588.  A single [N,3] cloud counts as
    B = 1 (the reference repeats grasp_mean N times there, :581, which its unnormalize_grasps then broadcasts
    into N copies of every grasp)."""
    _need_cuda(pc)
    assert pc.ndim in (2, 3)
    pcb = _as_batch(pc)
    b, n, _ = pcb.shape
    dev = pcb.device
    sh, sc = _vec(pc_shift, 3, "cpu"), _vec(pc_scale, 3, "cpu")
    out = torch.empty_like(pcb)
    mean = torch.empty((b, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        L.call("gldm_normalize_cloud", L.ptr(pcb), b, n, float(sh[0]), float(sh[1]), float(sh[2]), float(sc[0]),
               float(sc[1]), float(sc[2]), L.ptr(out), L.ptr(mean), L.current_stream(dev))
    gm = (torch.zeros(6) if grasp_shift is None else _vec(grasp_shift, 6, "cpu")).to(dev).unsqueeze(0).repeat(b, 1)
    gm[:, :3] += mean
    gstd = torch.cat([sc, _vec(mrp_scale, 3, "cpu")]).to(dev).unsqueeze(0)
    pc_mean = sh.to(dev) + mean
    metas = dict(pc_mean=pc_mean if pc.ndim == 3 else pc_mean[0], pc_std=sc.to(dev).unsqueeze(0),
                 grasp_mean=gm, grasp_std=gstd, dataset_normalized=True, use_dataset_statistics=False)
    return (out if pc.ndim == 3 else out[0]), metas


def read_cloud_file(path):
    """A sensor cloud from disk -> float32 numpy [N, 3] (metres, sensor / world frame).  `.npy` ([N,3] or [N,>=3]),
    `.npz` (first of the keys pc / points / xyz / arr_0), `.ply` (ascii or binary_little_endian, vertex x y z) and
    whitespace-separated text (`.xyz`, `.txt`, `.pts`).  The reference's CLI only iterates dataset items
    (tools/generate_grasps.py:109-131); `generate_on_pointcloud` (inference_base.py:161-212) is its entry point for
    such clouds, this is the file front of it."""
    ext = path.rsplit(".", 1)[-1].lower() if "." in path else ""
    if ext == "npy":
        a = np.load(path)
    elif ext == "npz":
        with np.load(path) as z:
            key = next((k for k in ("pc", "points", "xyz", "arr_0") if k in z.files), None)
            if key is None:
                raise ValueError(f"{path}: none of the arrays pc / points / xyz / arr_0 found (has {z.files})")
            a = z[key]
    elif ext == "ply":
        a = _read_ply_vertices(path)
    elif ext in ("xyz", "txt", "pts", "csv"):
        a = np.loadtxt(path, delimiter="," if ext == "csv" else None, ndmin=2)
    else:
        raise ValueError(f"{path}: unknown cloud format (.npy .npz .ply .xyz .txt .pts .csv)")
    a = np.asarray(a)
    if a.ndim != 2 or a.shape[1] < 3 or a.shape[0] == 0:
        raise ValueError(f"{path}: expected an [N, 3] array of points, got {a.shape}")
    a = a[:, :3].astype(np.float32)
    if not np.isfinite(a).all():
        a = a[np.isfinite(a).all(axis=1)]   # depth cameras mark invalid pixels with NaN / inf
    return np.ascontiguousarray(a)


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def _read_ply_vertices(path):
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n_vert, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: PLY header has no end_header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n_vert = int(tok[2])
                elif not props:
                    raise ValueError(f"{path}: an element precedes `vertex`; not supported")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list property on vertices is not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        if not all(k in names for k in ("x", "y", "z")):
            raise ValueError(f"{path}: vertex element has no x / y / z")
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=n_vert, ndmin=2)
            return np.stack([rows[:, names.index(k)] for k in ("x", "y", "z")], axis=1)
        if fmt != "binary_little_endian":
            raise ValueError(f"{path}: PLY format {fmt!r} is not supported (ascii, binary_little_endian)")
        dt = np.dtype([(n, "<" + t) for n, t in props])
        v = np.frombuffer(f.read(dt.itemsize * n_vert), dtype=dt, count=n_vert)
        return np.stack([v["x"], v["y"], v["z"]], axis=1)
