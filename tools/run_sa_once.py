"""The set-abstraction kernels of the north star on the PointNet2SSG SA2 shape, a few launches each, for counter
collection (tools/pmc_kernels.sh): sa_group_kernel (ball query + grouping + centre subtraction + concat, HBM bound) and
sa_mlp2_kernel (fused gather + grouped MLP 131-128-128-256 + max).  B clouds, N = 512, M = 128, U = 64, C = 128."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graspldm_amd import _lib as L
from graspldm_amd.pvcnn import PointNetSAModule, ball_query, furthest_point_sample
from graspldm_amd.sa_pack import SaMlpPlan
from graspldm_amd.synthetic import load_synthetic_weights, synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, M, U, C = 512, 128, 64, 128
dev = torch.device("cuda:0")
pcs, _ = synthetic_batch(min(B, 32), 1024)
pts = (pcs.transpose(1, 2) * (0.05 / 0.12)).contiguous()
pts = pts.repeat((B + pts.shape[0] - 1) // pts.shape[0], 1, 1)[:B].contiguous().to(dev)
c1 = furthest_point_sample(pts, N)
c2 = furthest_point_sample(c1, M)
f1 = torch.randn(B, C, N, device=dev)
grouped = torch.empty(B, 3 + C, M, U, device=dev)
st = L.current_stream(dev)
sa = load_synthetic_weights(PointNetSAModule(num_centers=M, radius=0.4, num_neighbors=U, in_channels=C,
                                             out_channels=(128, 128, 256)).eval(), seed=2).to(dev)
plan = SaMlpPlan(sa.mlps[0], dev)
idx = ball_query(c2, c1, 0.4, U)
for _ in range(5):
    L.call("gldm_sa_group", L.ptr(c1), L.ptr(c2), L.ptr(f1), B, C, N, M, 0.4, U, L.ptr(grouped), None, st)
    plan.run(c1, c2, f1, idx)
torch.cuda.synchronize()
by = B * (12 * N + 4 * C * N + 12 * M + 4 * (C + 3) * M * U)
print(f"B={B}: sa_group algorithmic bytes per launch {by}; sa_mlp2 algorithmic FLOP per launch {B * 2 * M * U * (131 * 128 + 128 * 128 + 128 * 256)}")
