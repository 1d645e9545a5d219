"""`_backend`: the twelve Python-visible functions of the reference's pybind
module `_pvcnn_backend` (functional/src/bindings.cpp:10-37), served by
libgldm_hip.so.

Same tensor signatures, same argument checks (utils.hpp:7-18 -> RuntimeError),
same ownership: outputs are allocated here with torch on the input's device,
inputs are borrowed.  Every launch goes to torch's current HIP stream (the
reference launches half of its kernels on the default stream: SURVEY.md §5).
Backward entry points exist by name and raise (inference-only scope).
"""
import torch

from . import _lib as L


def _f32(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float tensor")


def _i32(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int tensor")


def _stream(t):
    return L.current_stream(t.device)


class _HipBackend:
    # ---- ball_query/ball_query.cpp:6-30
    @staticmethod
    def ball_query(centers_coords, points_coords, radius, num_neighbors):
        _f32(centers_coords, "centers_coords")
        _f32(points_coords, "points_coords")
        b, _, m = centers_coords.shape
        n = points_coords.shape[2]
        out = torch.zeros((b, m, int(num_neighbors)), dtype=torch.int32, device=centers_coords.device)
        with torch.cuda.device(out.device):
            L.call("gldm_ball_query", L.ptr(centers_coords), L.ptr(points_coords), b, n, m, float(radius),
                   int(num_neighbors), L.ptr(out), _stream(out))
        return out

    # ---- grouping/grouping.cpp:6-24
    @staticmethod
    def grouping_forward(features, indices):
        _f32(features, "features")
        _i32(indices, "indices")
        b, c, n = features.shape
        _, m, u = indices.shape
        out = torch.zeros((b, c, m, u), dtype=torch.float32, device=features.device)
        if out.numel():
            with torch.cuda.device(out.device):
                L.call("gldm_grouping_forward", L.ptr(features), L.ptr(indices), b, c, n, m, u, L.ptr(out), _stream(out))
        return out

    # ---- sampling/sampling.cpp:6-22
    @staticmethod
    def gather_features_forward(features, indices):
        _f32(features, "features")
        _i32(indices, "indices")
        b, c, n = features.shape
        m = indices.shape[1]
        out = torch.zeros((b, c, m), dtype=torch.float32, device=features.device)
        if out.numel():
            with torch.cuda.device(out.device):
                L.call("gldm_gather_features_forward", L.ptr(features), L.ptr(indices), b, c, n, m, L.ptr(out), _stream(out))
        return out

    # ---- sampling/sampling.cpp:43-58
    @staticmethod
    def furthest_point_sampling(coords, num_samples):
        _f32(coords, "coords")
        b, _, n = coords.shape
        out = torch.zeros((b, int(num_samples)), dtype=torch.int32, device=coords.device)
        if out.numel():
            with torch.cuda.device(out.device):
                L.call("gldm_furthest_point_sampling", L.ptr(coords), b, n, int(num_samples), L.ptr(out), _stream(out))
        return out

    # ---- interpolate/neighbor_interpolate.cpp:6-40
    @staticmethod
    def three_nearest_neighbors_interpolate_forward(points_coords, centers_coords, centers_features):
        _f32(points_coords, "points_coords")
        _f32(centers_coords, "centers_coords")
        _f32(centers_features, "centers_features")
        b, c, m = centers_features.shape
        n = points_coords.shape[2]
        dev = points_coords.device
        idx = torch.zeros((b, 3, n), dtype=torch.int32, device=dev)
        wgt = torch.zeros((b, 3, n), dtype=torch.float32, device=dev)
        out = torch.zeros((b, c, n), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.call("gldm_three_nn_interpolate_forward", L.ptr(points_coords), L.ptr(centers_coords),
                   L.ptr(centers_features), b, c, m, n, L.ptr(out), L.ptr(idx), L.ptr(wgt), _stream(out))
        return [out, idx, wgt]

    # ---- voxelization/vox.cpp:17-43
    @staticmethod
    def avg_voxelize_forward(features, coords, resolution):
        _f32(features, "features")
        _i32(coords, "coords")
        b, c, n = features.shape
        r = int(resolution)
        dev = features.device
        ind = torch.empty((b, n), dtype=torch.int32, device=dev)   # every point's entry is written (a fill launch before)
        out = torch.empty((b, c, r ** 3), dtype=torch.float32, device=dev)  # every voxel is written by the C entry point
        cnt = torch.empty((b, r ** 3), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            L.call("gldm_avg_voxelize_forward", L.ptr(features), L.ptr(coords), b, c, n, r, L.ptr(out), L.ptr(ind),
                   L.ptr(cnt), _stream(out))
        return [out, ind, cnt]

    # ---- interpolate/trilinear_devox.cpp:18-55
    @staticmethod
    def trilinear_devoxelize_forward(r, is_training, coords, features):
        _f32(features, "features")
        _f32(coords, "coords")
        b, c, _ = features.shape
        n = coords.shape[2]
        dev = features.device
        outs = torch.zeros((b, c, n), dtype=torch.float32, device=dev)
        if is_training:
            inds = torch.zeros((b, 8, n), dtype=torch.int32, device=dev)
            wgts = torch.zeros((b, 8, n), dtype=torch.float32, device=dev)
        else:
            inds = torch.zeros((1,), dtype=torch.int32, device=dev)
            wgts = torch.zeros((1,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.call("gldm_trilinear_devoxelize_forward", L.ptr(coords), L.ptr(features), b, c, n, int(r),
                   1 if is_training else 0, L.ptr(outs), L.ptr(inds) if is_training else None,
                   L.ptr(wgts) if is_training else None, _stream(outs))
        return [outs, inds, wgts]

    # ---- backward halves (bindings.cpp:13,20,25,31,35): names kept, inference-only
    @staticmethod
    def _inference_only(*args, **kwargs):
        raise NotImplementedError(
            "graspldm_amd implements the grasp-generation (inference) path; backward kernels are out of scope")

    gather_features_backward = _inference_only
    grouping_backward = _inference_only
    three_nearest_neighbors_interpolate_backward = _inference_only
    trilinear_devoxelize_backward = _inference_only
    avg_voxelize_backward = _inference_only


_backend = _HipBackend()
__all__ = ["_backend"]
