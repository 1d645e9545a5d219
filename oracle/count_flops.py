"""Reference-graph FLOP counts of the point-cloud encoders (CONTAINER-ONLY: imports the reference's Python).

    python -m oracle.count_flops        # writes tests/golden/encoder_flops.json

torch.utils.flop_counter.FlopCounterMode over one forward of the reference's own modules (conv / linear / matmul FLOPs,
2 per multiply-add) on a [1, 3, 1024] cloud: the models of the reference's encoder benchmark
(ext/pvcnn/benchmark.py:491-542: PVCNN and PVCNN2 at scale 0.5 / 0.5, PointNet2SSG) and the shipped PVCNNEncoder
(fpc config: 0.75 / 0.75).  tools/bench_encoders.py divides these by its measured latencies."""
import json
import os
import sys

import torch
from torch.utils.flop_counter import FlopCounterMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402


@torch.no_grad()
def main():
    ref_import.install_shims()
    from grasp_ldm.models.modules.ext.pvcnn.pvcnn_base import PVCNN, PVCNN2
    from grasp_ldm.models.modules.ext.pvcnn.pointnet2 import PointNet2SSG
    from grasp_ldm.models.modules.pc_encoders import PVCNNEncoder
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 3, 1024, generator=g)
    models = {
        "PVCNN": (PVCNN(in_channels=3, extra_feature_channels=0, scale_channels=0.5, scale_voxel_resolution=0.5), x),
        "PVCNN2": (PVCNN2(in_channels=3, extra_feature_channels=0, width_multiplier=0.5, voxel_resolution_multiplier=0.5), x),
        "PVCNN2(full)": (PVCNN2(in_channels=3, extra_feature_channels=0), x),
        "PointNet2": (PointNet2SSG(extra_feature_channels=0, width_multiplier=1, voxel_resolution_multiplier=1), x),
        "PVCNNEncoder(fpc)": (PVCNNEncoder(in_features=3, out_features=64, n_points=1024, scale_channels=0.75,
                                           scale_voxel_resolution=0.75, num_blocks=(1, 1, 1, 1), out_channels=3,
                                           use_global_attention=False), x.transpose(1, 2).contiguous()),
    }
    out = {}
    for name, (m, inp) in models.items():
        m.eval()
        with FlopCounterMode(display=False) as fc:
            m(inp)
        out[name] = dict(flop_per_cloud=int(fc.get_total_flops()), params=sum(p.numel() for p in m.parameters()))
        print(f"{name:20s} {out[name]['flop_per_cloud'] / 1e9:8.3f} GFLOP per 1024-point cloud, {out[name]['params']:,} parameters")
    with open(os.path.join(ROOT, "tests", "golden", "encoder_flops.json"), "w") as f:
        json.dump(dict(points=1024, note="torch FlopCounterMode over the reference's modules (oracle/count_flops.py)",
                       models=out), f, indent=1)
    # ---- the 1-D ResNets of both shipped experiments: one denoiser forward per latent, one decoder forward per grasp
    from grasp_ldm.models.modules.resnets import ResNet1D, TimeConditionedResNet1D
    r1d = {}
    for tag, latent, pc_latent in (("fpc", 4, 64), ("ppc", 16, 256)):
        rn = dict(block_channels=(32, 64, 128, 256), input_conditioning_dims=pc_latent, resnet_block_groups=4, dropout=0.1)
        den = TimeConditionedResNet1D(dim=latent, channels=1, is_time_conditioned=True, learned_variance=False,
                                      learned_sinusoidal_cond=False, random_fourier_features=True, **rn).eval()
        xb = torch.randn(1, 1, latent, generator=g)
        zc = torch.randn(1, 3, pc_latent, generator=g)
        with FlopCounterMode(display=False) as fc:
            den(xb, time=torch.tensor([500]), z_cond=zc)
        r1d[f"denoiser({tag})"] = dict(flop_per_latent_step=int(fc.get_total_flops()), seq_len=latent, cond_dims=pc_latent)
        dec = ResNet1D(dim=16, channels=1, **rn).eval()   # the pose decoder's trunk at feature resolution 16 (grasp_vae.py:401-436)
        with FlopCounterMode(display=False) as fc:
            dec(torch.randn(1, 1, 16, generator=g), z_cond=zc)
        r1d[f"decoder_trunk({tag})"] = dict(flop_per_grasp=int(fc.get_total_flops()), seq_len=16, cond_dims=pc_latent)
    for k, v in r1d.items():
        print(k, v)
    with open(os.path.join(ROOT, "tests", "golden", "r1d_flops.json"), "w") as f:
        json.dump(dict(note="torch FlopCounterMode over the reference's TimeConditionedResNet1D / ResNet1D (oracle/count_flops.py); "
                            "includes the conditioning Linear + time MLP of every call, which the HIP path hoists out of the step loop",
                       models=r1d), f, indent=1)


if __name__ == "__main__":
    main()
