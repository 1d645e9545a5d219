"""The fused set-abstraction core (gather + grouped MLP + max) on the PointNet2SSG SA2 shape, timed alone with HIP
events: B clouds, N = 512 points, M = 128 centres, U = 64 neighbours, C = 128, MLP 131-128-128-256."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graspldm_amd.pvcnn import PointNetSAModule, ball_query
from graspldm_amd.sa_pack import SaMlpPlan
from graspldm_amd.synthetic import load_synthetic_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, M, U, C = 512, 128, 64, 128
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
pts = (torch.rand(B, 3, N, generator=g) * 2 - 1).to(dev)
ctr = pts[:, :, :M].contiguous()
feat = torch.randn(B, C, N, generator=g).to(dev)
sa = load_synthetic_weights(PointNetSAModule(num_centers=M, radius=0.4, num_neighbors=U, in_channels=C,
                                             out_channels=(128, 128, 256)).eval(), seed=2).to(dev)
plan = SaMlpPlan(sa.mlps[0], dev)
idx = ball_query(ctr, pts, 0.4, U)
for _ in range(3):
    plan.run(pts, ctr, feat, idx)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    plan.run(pts, ctr, feat, idx)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
flop = B * 2 * M * U * (131 * 128 + 128 * 128 + 128 * 256)
print(f"B={B}: {t * 1e3:.3f} ms  {flop / t / 1e12:.1f} TFLOP/s")
