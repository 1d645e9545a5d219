"""Fixed-seed denoise + encoder + decode outputs of the library named by GLDM_LIB, saved for a bitwise comparison of two builds:
    GLDM_LIB=a.so python tools/dbg/dump_denoise.py out_a.pt; GLDM_LIB=b.so python tools/dbg/dump_denoise.py out_b.pt
    python tools/dbg/dump_denoise.py --compare out_a.pt out_b.pt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if sys.argv[1] == "--compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        d = (a[k].double() - b[k].double()).abs().max().item()
        print(f"{k:10s} bitwise equal: {torch.equal(a[k], b[k])}  max abs diff {d:.3e}")
    sys.exit(0)
from graspldm_amd.pipeline import build_fpc_ldm
from graspldm_amd.synthetic import synthetic_batch
torch.manual_seed(0)
ldm = build_fpc_ldm(device="cuda:0")
pcs, _ = synthetic_batch(8, 1024)
x_T = torch.randn(8 * 20, 1, 4)
z = ldm.vae_model.encode_pc(pcs.cuda())
(tm, lg), _ = ldm.generate_grasps(pcs.cuda(), num_grasps=20, x_T=x_T)
torch.save({"z_pc": z.cpu(), "tmrp": tm.cpu(), "logit": lg.cpu()}, sys.argv[1])
print("saved", sys.argv[1])
