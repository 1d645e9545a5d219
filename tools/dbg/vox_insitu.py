import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graspldm_amd.pipeline import build_fpc_ldm
from graspldm_amd.synthetic import synthetic_batch
from graspldm_amd import pvcnn
ldm = build_fpc_ldm(device="cuda:0")
pcs, _ = synthetic_batch(32, 1024)
x = pcs.repeat(8, 1, 1).contiguous().cuda()
saved = []
orig = pvcnn.Voxelization.forward
def fwd(self, features, coords):
    torch.cuda.synchronize(); t = time.time()
    out = orig(self, features, coords)
    torch.cuda.synchronize()
    saved.append((self.r, features.shape, (time.time() - t) * 1e3, features.detach().clone(), coords.detach().clone()))
    return out
pvcnn.Voxelization.forward = fwd
for it in range(3):
    saved.clear()
    ldm.vae_model.encode_pc(x)
for r, shp, ms, f, c in saved:
    print("in situ r", r, tuple(shp), f"{ms:.3f} ms", "feat contiguous", f.is_contiguous(), "finite", bool(torch.isfinite(f).all()))
    pvcnn.Voxelization.forward = orig
    m = pvcnn.Voxelization(r, normalize=True, eps=0)
    for _ in range(3): m(f, c)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): vox, nc = m(f, c)
    torch.cuda.synchronize()
    print("   replay", f"{(time.time() - t) * 100:.3f} ms")
    from graspldm_amd.backend import _backend as B
    vc = torch.round(nc).int().contiguous()
    key = vc[:, 0] * r * r + vc[:, 1] * r + vc[:, 2]
    mx = max(int(torch.bincount(key[b].long()).max()) for b in range(key.shape[0]))
    print("   max points per voxel", mx)
