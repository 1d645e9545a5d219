import os, sys, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graspldm_amd import _lib as L
from graspldm_amd.pvcnn import PVCNN2, PointNet2SSG
which = sys.argv[1] if len(sys.argv) > 1 else "PVCNN2"
m = (PVCNN2 if which == "PVCNN2" else PointNet2SSG)(extra_feature_channels=0, width_multiplier=1, voxel_resolution_multiplier=1).cuda().eval()
x = torch.randn(8, 3, 1024, device="cuda") * 0.3
orig = L.call
log = []
def call(name, *a):
    ints = [v for v in a if isinstance(v, int) and not isinstance(v, bool) and abs(v) < 100000]
    log.append((name, tuple(ints[:8])))
    return orig(name, *a)
L.call = call
import graspldm_amd.sa_pack, graspldm_amd.dense, graspldm_amd.voxel, graspldm_amd.pvcnn
with torch.no_grad(): m(x)
for n, i in log: print(n, i)
