"""ctypes front end of oracle/point_ops.c exposing the twelve function names of
the reference's pybind module (functional/src/bindings.cpp:10-37) on CPU
tensors.  TEST INFRASTRUCTURE: see oracle/__init__.py.

Output allocation mirrors the reference's C++ wrappers (zero-initialised
outputs: ball_query.cpp:20-22, vox.cpp:31-36, sampling.cpp:51-54, ...).
Backward entry points exist by name and raise (inference-only scope).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgldm_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "point_ops.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libgldm_oracle.so"])
    return _LIB_PATH


def _load():
    if not os.path.exists(_LIB_PATH):
        build()
    return ctypes.CDLL(_LIB_PATH)


_lib = _load()
_f = ctypes.c_void_p
_i = ctypes.c_int
_lib.oracle_ball_query.argtypes = [_i, _i, _i, ctypes.c_float, _i, _f, _f, _f]
_lib.oracle_grouping.argtypes = [_i, _i, _i, _i, _i, _f, _f, _f]
_lib.oracle_gather.argtypes = [_i, _i, _i, _i, _f, _f, _f]
_lib.oracle_fps.argtypes = [_i, _i, _i, _f, _f, _f]
_lib.oracle_three_nn_interpolate.argtypes = [_i, _i, _i, _i, _f, _f, _f, _f, _f, _f]
_lib.oracle_avg_voxelize.argtypes = [_i, _i, _i, _i, _f, _f, _f, _f, _f]
_lib.oracle_trilinear_devoxelize.argtypes = [_i, _i, _i, _i, _f, _f, _f, _f, _f]
for _fn in ("oracle_ball_query", "oracle_grouping", "oracle_gather", "oracle_fps",
            "oracle_three_nn_interpolate", "oracle_avg_voxelize", "oracle_trilinear_devoxelize"):
    getattr(_lib, _fn).restype = None


def _chk(t, dtype, name):
    if t.device.type != "cpu":
        raise RuntimeError(f"{name} must be a CPU tensor for the oracle backend")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be a {'float' if dtype == torch.float32 else 'int'} tensor")


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class _CpuBackend:
    """Namespace object standing in for the pybind module `_pvcnn_backend`."""

    @staticmethod
    def ball_query(centers_coords, points_coords, radius, num_neighbors):
        _chk(centers_coords, torch.float32, "centers_coords")
        _chk(points_coords, torch.float32, "points_coords")
        b, _, m = centers_coords.shape
        n = points_coords.shape[2]
        out = torch.zeros((b, m, num_neighbors), dtype=torch.int32)
        r = np.float32(radius)  # `const float radius`
        r2 = ctypes.c_float(float(np.float32(r * r)))  # radius * radius in f32 (ball_query.cpp:24)
        _lib.oracle_ball_query(b, n, m, r2, num_neighbors, _p(centers_coords), _p(points_coords), _p(out))
        return out

    @staticmethod
    def grouping_forward(features, indices):
        _chk(features, torch.float32, "features")
        _chk(indices, torch.int32, "indices")
        b, c, n = features.shape
        _, m, u = indices.shape
        out = torch.zeros((b, c, m, u), dtype=torch.float32)
        _lib.oracle_grouping(b, c, n, m, u, _p(features), _p(indices), _p(out))
        return out

    @staticmethod
    def gather_features_forward(features, indices):
        _chk(features, torch.float32, "features")
        _chk(indices, torch.int32, "indices")
        b, c, n = features.shape
        m = indices.shape[1]
        out = torch.zeros((b, c, m), dtype=torch.float32)
        _lib.oracle_gather(b, c, n, m, _p(features), _p(indices), _p(out))
        return out

    @staticmethod
    def furthest_point_sampling(coords, num_samples):
        _chk(coords, torch.float32, "coords")
        b, _, n = coords.shape
        out = torch.zeros((b, num_samples), dtype=torch.int32)
        dist = torch.empty((b, n), dtype=torch.float32)
        _lib.oracle_fps(b, n, num_samples, _p(coords), _p(dist), _p(out))
        return out

    @staticmethod
    def three_nearest_neighbors_interpolate_forward(points_coords, centers_coords, centers_features):
        _chk(points_coords, torch.float32, "points_coords")
        _chk(centers_coords, torch.float32, "centers_coords")
        _chk(centers_features, torch.float32, "centers_features")
        b, c, m = centers_features.shape
        n = points_coords.shape[2]
        idx = torch.zeros((b, 3, n), dtype=torch.int32)
        wgt = torch.zeros((b, 3, n), dtype=torch.float32)
        out = torch.zeros((b, c, n), dtype=torch.float32)
        _lib.oracle_three_nn_interpolate(b, c, m, n, _p(points_coords), _p(centers_coords),
                                         _p(centers_features), _p(idx), _p(wgt), _p(out))
        return [out, idx, wgt]

    @staticmethod
    def avg_voxelize_forward(features, coords, resolution):
        _chk(features, torch.float32, "features")
        _chk(coords, torch.int32, "coords")
        b, c, n = features.shape
        r3 = resolution ** 3
        ind = torch.zeros((b, n), dtype=torch.int32)
        out = torch.zeros((b, c, r3), dtype=torch.float32)
        cnt = torch.zeros((b, r3), dtype=torch.int32)
        _lib.oracle_avg_voxelize(b, c, n, resolution, _p(features), _p(coords), _p(ind), _p(cnt), _p(out))
        return [out, ind, cnt]

    @staticmethod
    def trilinear_devoxelize_forward(r, is_training, coords, features):
        _chk(features, torch.float32, "features")
        _chk(coords, torch.float32, "coords")
        b, c, _ = features.shape
        n = coords.shape[2]
        outs = torch.zeros((b, c, n), dtype=torch.float32)
        if is_training:
            inds = torch.zeros((b, 8, n), dtype=torch.int32)
            wgts = torch.zeros((b, 8, n), dtype=torch.float32)
            _lib.oracle_trilinear_devoxelize(b, c, n, r, _p(coords), _p(features), _p(inds), _p(wgts), _p(outs))
        else:
            inds = torch.zeros((1,), dtype=torch.int32)
            wgts = torch.zeros((1,), dtype=torch.float32)
            _lib.oracle_trilinear_devoxelize(b, c, n, r, _p(coords), _p(features), None, None, _p(outs))
        return [outs, inds, wgts]

    # backward halves: present by name (bindings.cpp:13,20,25,31,35), out of scope
    @staticmethod
    def _no_backward(*a, **k):
        raise NotImplementedError("inference-only oracle: backward kernels are out of scope")

    gather_features_backward = _no_backward
    grouping_backward = _no_backward
    three_nearest_neighbors_interpolate_backward = _no_backward
    trilinear_devoxelize_backward = _no_backward
    avg_voxelize_backward = _no_backward


_backend = _CpuBackend()
__all__ = ["_backend", "build"]
