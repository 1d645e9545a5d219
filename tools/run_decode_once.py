"""One fused decode launch (n grasps) for counter / stamp collection; prints the launch time.  GLDM_LIB selects a
diagnostic build (GLDM_R1D_STAMP=1 then prints per-op cycle stamps of workgroup 0)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graspldm_amd import _lib
if os.environ.get("GLDM_LIB"):
    _lib.LIB_PATH = os.environ["GLDM_LIB"]
from graspldm_amd.pipeline import build_fpc_ldm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
G = 20 if n % 20 == 0 else 16
dev = torch.device("cuda:0")
ldm = build_fpc_ldm(device=dev)
dec = ldm.vae_model.decoder
z = torch.randn(n // G, 3, 64, device=dev); zh = torch.randn(n, 4, device=dev)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    dec(zh, z, samples_per_cond=G)
    torch.cuda.synchronize(); dt = time.time() - t0
print(f"n={n}: {dt * 1e3:.3f} ms  {n * 30.7e6 / dt / 1e12:.1f} TFLOP/s")
