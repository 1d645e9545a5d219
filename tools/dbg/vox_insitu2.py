import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graspldm_amd.pipeline import build_fpc_ldm
from graspldm_amd.synthetic import synthetic_batch
from graspldm_amd import pvcnn
ldm = build_fpc_ldm(device="cuda:0")
pcs, _ = synthetic_batch(32, 1024)
x = pcs.repeat(8, 1, 1).contiguous().cuda()
orig = pvcnn.avg_voxelize
def av(features, coords, r):
    key = (coords[:, 0] * r * r + coords[:, 1] * r + coords[:, 2]).long()
    mx = [int(torch.bincount(key[b]).max()) for b in range(key.shape[0])]
    occ = [int((torch.bincount(key[b]) > 0).sum()) for b in range(key.shape[0])]
    torch.cuda.synchronize(); t = time.time()
    out = orig(features, coords, r)
    torch.cuda.synchronize()
    print("r", r, tuple(features.shape), f"{(time.time() - t) * 1e3:.3f} ms  max pts/voxel {max(mx)}  mean occupied {sum(occ) / len(occ):.0f}", flush=True)
    return out
pvcnn.avg_voxelize = av
for m in ldm.modules():
    if isinstance(m, pvcnn.Voxelization): print("Voxelization r", m.r, "normalize", m.normalize, "eps", m.eps)
ldm.vae_model.encode_pc(x); ldm.vae_model.encode_pc(x)
