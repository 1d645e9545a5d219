"""One fused denoise launch (n latents, s steps) for counter collection; prints the launch time."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graspldm_amd import _lib
if os.environ.get("GLDM_LIB"):  # e.g. a diagnostic build: make -C graspldm_amd/csrc EXTRA=-DGLDM_DEBUG_KNOBS BUILD=... OUT=...
    _lib.LIB_PATH = os.environ["GLDM_LIB"]
from graspldm_amd.pipeline import build_fpc_ldm
from graspldm_amd.r1d_pack import SCHED_DDIM
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
G = int(sys.argv[3]) if len(sys.argv) > 3 else (20 if n % 20 == 0 else 16)  # latents per cloud
dev = torch.device("cuda:0")
ldm = build_fpc_ldm(device=dev); ldm.set_inference_timesteps(100)
eng = ldm.diffusion_model.model.engine(dev)
if os.environ.get("GLDM_BLOCKS"):   # e.g. GLDM_BLOCKS=32,64,128: another width sequence (experiments on the weight footprint)
    from graspldm_amd.resnets import TimeConditionedResNet1D
    from graspldm_amd.synthetic import load_synthetic_weights
    net = TimeConditionedResNet1D(dim=4, channels=1, block_channels=tuple(int(v) for v in os.environ["GLDM_BLOCKS"].split(",")),
                                  input_conditioning_dims=64, resnet_block_groups=4, dropout=0.1, is_time_conditioned=True,
                                  learned_variance=False, learned_sinusoidal_cond=False, random_fourier_features=True)
    eng = load_synthetic_weights(net, seed=1).to(dev).eval().engine(dev)
z = torch.randn(n // G, 3, 64, device=dev); x = torch.randn(n, 1, 4, device=dev)
cemb = eng.cond_embed(z); ts, coef = ldm.diffusion_model._schedule(dev)
ts, coef = ts[:steps].contiguous(), coef[:steps].contiguous()
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    eng.denoise(x, cemb, G, timesteps=ts, sched_kind=SCHED_DDIM, coef=coef)
    torch.cuda.synchronize(); dt = time.time() - t0
print(f"n={n} steps={steps}: {dt * 1e3:.2f} ms  {n * steps * 7589120 / dt / 1e12:.1f} TFLOP/s")
