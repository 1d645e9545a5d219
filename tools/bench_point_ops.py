"""Time the point operators on one MI355X at the BASELINE shapes and print the
achieved algorithmic bandwidth (SURVEY.md section 8d byte counts)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graspldm_amd import _lib as L  # noqa: E402
from graspldm_amd.backend import _backend as hip  # noqa: E402
from graspldm_amd.synthetic import synthetic_batch  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=256)
    args = ap.parse_args()
    B = args.clouds
    pcs, _ = synthetic_batch(min(B, 64), 1024)
    pts = (pcs.transpose(1, 2) * (0.05 / 0.12)).contiguous()
    pts = pts.repeat((B + pts.shape[0] - 1) // pts.shape[0], 1, 1)[:B].contiguous().cuda()
    res = {}

    def rec(name, sec, nbytes=None, extra=None):
        d = {"ms": sec * 1e3}
        if nbytes:
            d["GB/s"] = nbytes / sec / 1e9
            d["frac_of_8TBps"] = nbytes / sec / 8e12
        if extra:
            d.update(extra)
        res[name] = d
        print(name, json.dumps(d))

    # SSG SA1: N=1024 -> M=512, SA2: N=512 -> M=128
    rec("fps_1024_to_512", timeit(lambda: hip.furthest_point_sampling(pts, 512)), extra={"us_per_cloud_amortised": None})
    idx1 = hip.furthest_point_sampling(pts, 512)
    c1 = hip.gather_features_forward(pts, idx1)
    rec("fps_512_to_128", timeit(lambda: hip.furthest_point_sampling(c1, 128)))
    c2 = hip.gather_features_forward(c1, hip.furthest_point_sampling(c1, 128))
    f1 = torch.randn(B, 128, 512, device="cuda")
    N, M, U, C = 512, 128, 64, 128
    nb = hip.ball_query(c2, c1, 0.4, U)
    rec("ball_query_sa2", timeit(lambda: hip.ball_query(c2, c1, 0.4, U)))
    rec("grouping_sa2", timeit(lambda: hip.grouping_forward(f1, nb)), B * (4 * C * N + 4 * M * U + 4 * C * M * U))
    out = torch.empty(B, 3 + C, M, U, device="cuda")
    st = L.current_stream()
    by = B * (12 * N + 4 * C * N + 12 * M + 4 * (C + 3) * M * U)
    rec("sa_group_sa2", timeit(lambda: L.call("gldm_sa_group", L.ptr(c1), L.ptr(c2), L.ptr(f1), B, C, N, M, 0.4, U,
                                              L.ptr(out), None, st)), by)
    out1 = torch.empty(B, 3, 512, 64, device="cuda")
    by1 = B * (12 * 1024 + 12 * 512 + 4 * 3 * 512 * 64)
    rec("sa_group_sa1", timeit(lambda: L.call("gldm_sa_group", L.ptr(pts), L.ptr(c1), None, B, 0, 1024, 512, 0.2, 64,
                                              L.ptr(out1), None, st)), by1)
    # PVCNN shapes
    for (c, r) in [(3, 24), (48, 12)]:
        feat = torch.randn(B, c, 1024, device="cuda")
        nc = torch.empty(B, 3, 1024, device="cuda")
        vc = torch.empty(B, 3, 1024, dtype=torch.int32, device="cuda")
        rec(f"voxel_coords_r{r}", timeit(lambda: L.call("gldm_voxel_coords", L.ptr(pts), B, 1024, r, 0, 0.0, L.ptr(nc), L.ptr(vc), st)))
        rec(f"avg_voxelize_c{c}_r{r}", timeit(lambda: hip.avg_voxelize_forward(feat, vc, r)),
            B * (4 * c * 1024 + 12 * 1024 + 4 * c * r ** 3))
        co = c * 16 if c == 3 else 96
        grid = torch.randn(B, co, r ** 3, device="cuda")
        rec(f"devoxelize_c{co}_r{r}", timeit(lambda: hip.trilinear_devoxelize_forward(r, False, nc, grid)),
            B * (4 * co * r ** 3 + 12 * 1024 + 4 * co * 1024))
    cf = torch.randn(B, 256, 128, device="cuda")
    rec("three_nn_128_to_512", timeit(lambda: hip.three_nearest_neighbors_interpolate_forward(c1, c2, cf)))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/point_ops_bench.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
