"""Debug: 16-position denoisers of growing depth on the 64-column engine vs the torch oracle."""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import torch_ref as R
from graspldm_amd import _lib as L
from graspldm_amd.resnets import TimeConditionedResNet1D
from graspldm_amd.synthetic import load_synthetic_weights
for bc in [(32,), (32, 64), (32, 64, 128), (32, 64, 128, 256)]:
    net = TimeConditionedResNet1D(dim=16, channels=1, block_channels=bc, input_conditioning_dims=64, resnet_block_groups=4,
                                  dropout=0.1, is_time_conditioned=True, learned_variance=False, learned_sinusoidal_cond=False,
                                  random_fourier_features=True)
    load_synthetic_weights(net, seed=3)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda().eval()
    g = torch.Generator().manual_seed(5)
    n = 7
    x, z = torch.randn(n, 1, 16, generator=g), torch.randn(n, 3, 64, generator=g)
    t = torch.randint(0, 1000, (n,), generator=g)
    eng = net.engine(torch.device("cuda:0"))
    cols = L.lib().gldm_r1d_tile_columns(eng._desc_ptr())
    exp = R.resnet1d_forward(sd, "", x, z_cond=z, time=t)
    for rep in range(2):
        eps = net(x.cuda(), time=t.cuda(), z_cond=z.cuda())
        e = (eps.cpu() - exp).abs().flatten(1).max(dim=1).values
        print(bc, "cols", cols, [f"{v:.1e}" for v in e.tolist()])
