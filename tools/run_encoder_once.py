"""The shipped PVCNN encoder on a 256-cloud batch, a few forwards: the workload of tools/pmc_kernels.sh / prof_kernels.sh
when the encoder's kernels are profiled on their own.  `python tools/run_encoder_once.py [clouds] [iterations]`."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graspldm_amd.pipeline import build_fpc_ldm
from graspldm_amd.synthetic import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ldm = build_fpc_ldm(device="cuda:0")
pcs, _ = synthetic_batch(32, 1024)
x = pcs.repeat((B + 31) // 32, 1, 1)[:B].contiguous().cuda()
for it in range(iters):
    torch.cuda.synchronize(); t = time.time()
    z = ldm.vae_model.encode_pc(x)
    torch.cuda.synchronize()
    print(f"B={B} iter{it} encode {1e3 * (time.time() - t):.2f} ms", flush=True)
