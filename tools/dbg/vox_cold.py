import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graspldm_amd.backend import _backend as B
from graspldm_amd.synthetic import synthetic_batch
pcs, _ = synthetic_batch(32, 1024)
x = pcs.repeat(8, 1, 1).transpose(1, 2).contiguous().cuda()
def vox_of(r):
    m = x.mean(2, keepdim=True); c = x - m
    d = c.norm(dim=1).max(dim=1).values.view(-1, 1, 1) * 2
    return ((c / d + 0.5) * r).clamp(0, r - 1).round().int().contiguous()
junk = torch.empty(1 << 28, device="cuda")   # 1 GiB
for (c, r) in [(3, 24), (48, 12)]:
    feat = torch.randn(256, c, 1024, device="cuda"); vc = vox_of(r)
    for flush in (False, True):
        ts = []
        for _ in range(6):
            if flush: junk.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); B.avg_voxelize_forward(feat, vc, r); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"c={c} r={r} flush={flush}: " + " ".join(f"{t:.0f}" for t in ts) + " us", flush=True)
