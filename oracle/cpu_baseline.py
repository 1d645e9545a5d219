"""Time the CPU oracle on the host cores for bench.py's `cpu_baseline` leg.
TEST INFRASTRUCTURE (see oracle/__init__.py): a reported baseline, never the product.
Runs as its own process (CPU only, never touches the GPU):

    python -m oracle.cpu_baseline --clouds 1 --grasps 20 --points 1024 --ddim-steps 100 --threads 16

Prints one JSON line {"seconds": .., "grasps": .., "threads": ..}.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")
os.environ.setdefault("HIP_VISIBLE_DEVICES", "")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=1)
    ap.add_argument("--grasps", type=int, default=20)
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--ddim-steps", type=int, default=100)
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from graspldm_amd.pipeline import build_fpc_ldm
    from graspldm_amd.synthetic import synthetic_batch
    from oracle import torch_ref as R
    sd = {k: v.detach() for k, v in build_fpc_ldm(n_points=a.points).state_dict().items()}
    pcs, _ = synthetic_batch(a.clouds, a.points)
    x_T = torch.randn(a.clouds * a.grasps, 1, 4, generator=torch.Generator().manual_seed(1234))
    sched = R.make_scheduler("ddim")
    sched.set_timesteps(a.ddim_steps)
    spec = R.pvcnn_block_spec(0.75, 0.75)
    with torch.no_grad():
        R.pvcnn_encoder_forward(sd, "vae_model.encoder.pc_encoder.", pcs[:1], spec)  # warm the thread pool
        t0 = time.perf_counter()
        R.ldm_generate(sd, pcs, a.grasps, sched, spec, x_T=x_T)
        dt = time.perf_counter() - t0
    print(json.dumps(dict(seconds=dt, grasps=a.clouds * a.grasps, threads=a.threads)))


if __name__ == "__main__":
    main()
