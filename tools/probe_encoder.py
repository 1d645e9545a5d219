import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graspldm_amd.pipeline import build_fpc_ldm
from graspldm_amd.synthetic import synthetic_batch
print("cpus", os.cpu_count(), flush=True)
ldm = build_fpc_ldm(device="cuda:0")
pcs, _ = synthetic_batch(8, 1024)
for B in (1, 8, 64, 256):
    x = pcs.repeat((B + 7) // 8, 1, 1)[:B].contiguous().cuda()
    for it in range(3):
        torch.cuda.synchronize(); t = time.time()
        z = ldm.vae_model.encode_pc(x)
        torch.cuda.synchronize()
        print(f"B={B} iter{it} encode {1e3*(time.time()-t):.1f} ms", flush=True)
