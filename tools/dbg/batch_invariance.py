import sys, torch
sys.path.insert(0, "/root/repo")
from graspldm_amd.pvcnn import PointNet2SSG, PVCNN2
from graspldm_amd.synthetic import load_synthetic_weights
for cls in (PointNet2SSG, PVCNN2):
    m = cls(extra_feature_channels=0); load_synthetic_weights(m, seed=0); m = m.cuda().eval()
    x = torch.randn(3, 3, 1024, generator=torch.Generator().manual_seed(0)).cuda() * 0.3
    with torch.no_grad():
        y3 = m(x); y1 = m(x[:1].contiguous())
    print(cls.__name__, tuple(y3.shape), float((y3[:1] - y1).abs().max()), bool(torch.isfinite(y3).all()))
