import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graspldm_amd.backend import _backend as B
from graspldm_amd.synthetic import synthetic_batch
pcs, _ = synthetic_batch(32, 1024)
x = pcs.repeat(8, 1, 1).transpose(1, 2).contiguous().cuda()   # [256, 3, 1024]
def vox_of(r):
    m = x.mean(2, keepdim=True); c = x - m
    d = c.norm(dim=1).max(dim=1).values.view(-1, 1, 1) * 2
    return ((c / d + 0.5) * r).clamp(0, r - 1).round().int().contiguous()
for (c, r, kind) in [(3, 24, "cloud"), (8, 12, "cloud"), (48, 12, "cloud"), (48, 12, "rand"), (48, 12, "one"), (64, 32, "cloud")]:
    feat = torch.randn(256, c, 1024, device="cuda")
    if kind == "cloud": vc = vox_of(r)
    elif kind == "rand": vc = torch.randint(0, r, (256, 3, 1024), device="cuda", dtype=torch.int32)
    else: vc = torch.zeros((256, 3, 1024), device="cuda", dtype=torch.int32)
    for _ in range(3): B.avg_voxelize_forward(feat, vc, r)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): B.avg_voxelize_forward(feat, vc, r)
    torch.cuda.synchronize()
    print(f"c={c} r={r} {kind}: {(time.time() - t) * 100:.3f} ms", flush=True)
