#!/usr/bin/env python3
"""Encoder shoot-out on one MI355X: the counterpart of the reference's only performance
harness, `grasp_ldm/models/modules/ext/pvcnn/benchmark.py` (main :491-560: PVCNN vs PVCNN2 vs PointNet2SSG,
batch sizes 1/4/16/64/256, 1024 points, 3 input channels, scale 0.5/0.5; metrics :31-41,62-145: average /
p95 / p99 latency over timed iterations after warm-up, samples per second, peak memory, parameter size).

Same models, constructor arguments, input distribution (randn [B,3,N]) and metric definitions, on the HIP
path; no plots (matplotlib / seaborn are not in this image): a JSON file and a markdown table.  `--shipped`
adds the encoder the shipped configs really use (PVCNNEncoder, scale 0.75/0.75).

Added to the reference's columns: achieved TFLOP/s on the reference graph's FLOP count (tests/golden/encoder_flops.json:
torch's flop counter over the reference's own modules, oracle/count_flops.py) and its fraction of the f32 MFMA peak
(157.3 TFLOP/s); `--full-pvcnn2` adds PVCNN2 at width / resolution 1 (42.4 GFLOP per cloud).  Kernel breakdowns:
`bash tools/prof_kernels.sh <tag> tools/bench_encoders.py --only PVCNN2 --batch-sizes 256 --iterations 5`.

    python tools/bench_encoders.py --shipped --full-pvcnn2 --out profiles/r03_encoder_shootout.json
"""
import argparse
import gc
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graspldm_amd.pvcnn import PVCNN, PVCNN2, PointNet2SSG  # noqa: E402
from graspldm_amd.synthetic import load_synthetic_weights  # noqa: E402


def measure_inference_time(model, x, num_iterations, warmup_iterations):
    times = []
    with torch.inference_mode():
        for _ in range(warmup_iterations):
            model(x)
            torch.cuda.synchronize()
        for _ in range(num_iterations):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model(x)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
    return np.array(times)


def benchmark_model(model, batch_sizes, num_points, in_channels, num_iterations, warmup):
    out = {}
    param_mb = sum(p.numel() * p.element_size() for p in model.parameters()) / 2 ** 20
    for b in batch_sizes:
        torch.cuda.empty_cache()
        gc.collect()
        torch.cuda.reset_peak_memory_stats()
        x = torch.randn(b, in_channels, num_points, device="cuda", dtype=torch.float32)
        with torch.inference_mode():
            model(x)
            torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() / 2 ** 20
        t = measure_inference_time(model, x, num_iterations, warmup)
        out[b] = dict(avg_latency_ms=float(t.mean()), p95_latency_ms=float(np.percentile(t, 95)),
                      p99_latency_ms=float(np.percentile(t, 99)), throughput_samples_per_sec=float(b * 1000 / t.mean()),
                      peak_memory_mb=float(peak), model_parameters_mb=float(param_mb))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch-sizes", type=int, nargs="+", default=[1, 4, 16, 64, 256])
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--iterations", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shipped", action="store_true", help="also time PVCNNEncoder of the shipped fpc config")
    ap.add_argument("--full-pvcnn2", action="store_true", help="also time PVCNN2 at width / resolution multiplier 1")
    ap.add_argument("--only", type=str, default=None, help="run only this model (kernel traces)")
    ap.add_argument("--out", type=str, default=None)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs the MI355X"
    torch.manual_seed(0)
    models = {
        "PVCNN": PVCNN(in_channels=3, extra_feature_channels=0, scale_channels=0.5, scale_voxel_resolution=0.5),
        "PVCNN2": PVCNN2(in_channels=3, extra_feature_channels=0, width_multiplier=0.5, voxel_resolution_multiplier=0.5),
        "PointNet2": PointNet2SSG(extra_feature_channels=0, width_multiplier=1, voxel_resolution_multiplier=1),
    }
    if args.shipped:
        from graspldm_amd.pc_encoders import PVCNNEncoder

        class _Tr(torch.nn.Module):  # the encoder takes [B,N,3]
            def __init__(self, m):
                super().__init__()
                self.m = m

            def forward(self, x):
                return self.m(x.transpose(1, 2))
        models["PVCNNEncoder(fpc)"] = _Tr(PVCNNEncoder(in_features=3, out_features=64, n_points=args.points,
                                                       scale_channels=0.75, scale_voxel_resolution=0.75,
                                                       num_blocks=(1, 1, 1, 1), out_channels=3))
    if args.full_pvcnn2:
        models["PVCNN2(full)"] = PVCNN2(in_channels=3, extra_feature_channels=0)
    if args.only:
        models = {k: v for k, v in models.items() if k == args.only}
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "encoder_flops.json")) as f:
        flops = {k: v["flop_per_cloud"] for k, v in json.load(f)["models"].items()}
    PEAK = 157.3                      # f32 MFMA, dense
    PEAK_SPLIT = 2500.0 / 3           # f32 products as three f16 partial products on the f16 matrix pipe
    # Executed FLOP per cloud by the pipe they run on, where part of a model runs as split-f16 products: the shipped
    # PVCNNEncoder (768 -> 1536 layer + its 96 -> 768 front layer + the 48 / 96-channel voxel convs; head convs folded into
    # one 1536 -> 3 GEMM: 2.4 GFLOP of the reference graph never executed).
    # The set-abstraction MLPs of the two PointNet++-style backbones run on the split kernel too (sa_mlp3_kernel), as do the
    # feature-propagation layers whose shapes the fused split launch takes (cin % 128 == 0, cout % 256 == 0).
    def sa(m, u, chans):   # FLOP of a grouped MLP over m centres x u neighbours
        return 2 * m * u * sum(a * b for a, b in zip(chans[:-1], chans[1:]))
    ssg_split = sa(512, 64, (3, 64, 64, 128)) + sa(128, 64, (131, 128, 128, 256)) \
        + 2 * 128 * (256 * 512 + 512 * 1024 + 256 * 256) + 2 * 512 * 384 * 256
    pv2_split = sa(1024, 32, (19, 16, 32)) + sa(256, 32, (35, 32, 64)) + sa(64, 32, (67, 64, 128))
    split_exec = {"PVCNNEncoder(fpc)": (2 * 1024 * (96 * 768 + 768 * 1536) + 2 * 27 * ((3 * 48 + 48 * 48) * 24 ** 3 + 48 * 96 * 12 ** 3 + 96 * 96 * 12 ** 3),
                                        8.115050112e9 - 2 * 768 * 1536 * 1024 - 2 * 3 * 768 * 1024 + 2 * 3 * 1536 * 1024),
                  "PointNet2": (ssg_split, flops.get("PointNet2", 0)), "PVCNN2": (pv2_split, flops.get("PVCNN2", 0))}
    # Models without a hand-written entry above: the split-f16 share of one forward is COUNTED from the launches themselves
    # (a shim around _lib.call adds up 2 cin cout n / 2 * 27 cin cout r^3 / ... of every `*_f16x2*` entry point of ONE cloud).
    def split_flop_of_forward(model, n_points):
        from graspldm_amd import _lib as L
        tot = [0]
        real = L.call
        def counting(name, *a):
            if name == "gldm_pointwise_mlp_f16x2":
                tot[0] += 2 * a[3] * a[4] * a[5] * a[6]
            elif name == "gldm_pointwise_mlp_f16x2_add":
                tot[0] += 2 * a[7] * a[8] * a[9] * a[10]
            elif name == "gldm_pointwise_mlp2_f16x2":
                tot[0] += 2 * a[6] * a[9] * (a[3] * a[7] + a[7] * a[8])
            elif name in ("gldm_conv3d_k3_f16x2", "gldm_conv3d_k3_f16x2_gn"):
                o = 1 if name.endswith("_gn") else 0
                tot[0] += 2 * 27 * a[3 + o] * a[4 + o] * a[5 + o] * a[6 + o] ** 3
            return real(name, *a)
        L.call = counting
        try:
            with torch.no_grad():
                model(torch.randn(1, 3, n_points, device="cuda"))
        finally:
            L.call = real
        return tot[0]
    results = {}
    for name, m in models.items():
        load_synthetic_weights(m, seed=0)
        m = m.cuda().eval()
        if name in ("PVCNN", "PVCNN2", "PVCNN2(full)") and name in flops:   # + the set-abstraction share entered by hand above
            base = split_exec.get(name, (0, flops[name]))
            split_exec[name] = (base[0] + split_flop_of_forward(m, args.points), base[1])
        results[name] = benchmark_model(m, args.batch_sizes, args.points, 3, args.iterations, args.warmup)
        for b, r in results[name].items():
            fl = flops.get(name) if args.points == 1024 else None
            r["reference_gflop_per_cloud"] = fl / 1e9 if fl else None
            r["achieved_tflops"] = fl * b / (r["avg_latency_ms"] * 1e-3) / 1e12 if fl else None
            # fraction of the peak of the pipe(s) the model runs on: never above 1 (a split-f16 kernel rated against the f32
            # peak read 1.6 in round 3).  Mixed models: executed FLOP / the time every GEMM would take at its own pipe's peak.
            if fl and name in split_exec and args.points == 1024:
                f_split, f_exec = split_exec[name]
                t_floor = b * (f_split / (PEAK_SPLIT * 1e12) + (f_exec - f_split) / (PEAK * 1e12))
                r["executed_gflop_per_cloud"] = f_exec / 1e9
                r["frac_of_pipe_peak"] = t_floor / (r["avg_latency_ms"] * 1e-3)
                r["pipes"] = "split-f16 (2500 / 3 TFLOP/s) + f32 MFMA (157.3), time-weighted"
            else:
                r["frac_of_pipe_peak"] = r["achieved_tflops"] / PEAK if fl else None
                r["pipes"] = "f32 MFMA (157.3 TFLOP/s)"
            print(f"{name:18s} B={b:4d}  avg {r['avg_latency_ms']:8.3f} ms  p95 {r['p95_latency_ms']:8.3f}  "
                  f"{r['throughput_samples_per_sec']:10.1f} clouds/s  peak {r['peak_memory_mb']:8.1f} MB", flush=True)
    ref = "PVCNN" if "PVCNN" in results else next(iter(results))
    lines = ["| Batch Size | Model | Avg Latency (ms) | P95 (ms) | P99 (ms) | Throughput (clouds/s) | Peak Memory (MB) | "
             "Relative Speedup | Parameters (MB) | GFLOP / cloud (reference graph) | TFLOP/s (reference FLOP) | of the peak of the pipe(s) it runs on |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for b in args.batch_sizes:
        for name, res in results.items():
            r = res[b]
            lines.append(f"| {b} | {name} | {r['avg_latency_ms']:.3f} | {r['p95_latency_ms']:.3f} | {r['p99_latency_ms']:.3f} | "
                         f"{r['throughput_samples_per_sec']:.1f} | {r['peak_memory_mb']:.1f} | "
                         f"{results[ref][b]['avg_latency_ms'] / r['avg_latency_ms']:.2f}x | {r['model_parameters_mb']:.1f} | "
                         + (f"{r['reference_gflop_per_cloud']:.3f} | {r['achieved_tflops']:.2f} | {100 * r['frac_of_pipe_peak']:.1f} % |"
                            if r.get("achieved_tflops") else "- | - | - |"))
    table = "\n".join(lines)
    print(table)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(dict(gpu=torch.cuda.get_device_name(), torch=torch.__version__, points=args.points,
                           iterations=args.iterations, results=results), f, indent=1)
        with open(os.path.splitext(args.out)[0] + ".md", "w") as f:
            f.write(f"# Encoder shoot-out ({torch.cuda.get_device_name()}, {args.points} points, f32, "
                    f"{args.iterations} timed iterations)\n\n`python tools/bench_encoders.py"
                    f"{' --shipped' if args.shipped else ''}`; models and metrics as in the reference's "
                    "ext/pvcnn/benchmark.py.\n\n" + table + "\n")


if __name__ == "__main__":
    main()
