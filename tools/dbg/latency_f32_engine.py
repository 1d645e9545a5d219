"""One-object latency probe: the fused denoise launch of few samples on the position-major split engine and on the
sample-major f32 engine (32-column tiles of 8 samples; a tile of <= 4 samples runs its ops on one 16-column n-tile)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graspldm_amd.pipeline import build_fpc_ldm
from graspldm_amd.synthetic import synthetic_batch
from graspldm_amd.r1d import R1dEngine
from graspldm_amd.r1d_pack import pack_resnet1d
dev = torch.device("cuda:0")
ldm = build_fpc_ldm(device=dev)
ldm.set_inference_timesteps(100)
den = ldm.diffusion_model.model
sd32 = {k: v.detach().float().cpu() for k, v in den.state_dict().items()}
packed = pack_resnet1d(sd32, "", groups=den.groups, seq_len=den.in_features, num_steps=den.max_timesteps, cond_rows=getattr(den, "cond_rows", 3))
for rb in packed["desc"].rb: rb.c1_w3 = rb.c2_w3 = 0
for lv in packed["desc"].lv: lv.qkvn_w3 = lv.out_w3 = lv.down_w3 = 0
eng32 = R1dEngine(packed, dev)
from graspldm_amd.r1d_pack import SCHED_DDIM
eng = den.engine(dev)
pcs, _ = synthetic_batch(1, 1024)
z = ldm.vae_model.encode_pc(pcs.to(dev))
ts, coef = ldm.diffusion_model._schedule(dev)


def timed(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, e in (("split position-major", eng), ("f32 sample-major", eng32)):
    if e is None: continue
    cemb = e.cond_embed(z)
    for g in (20, 16, 8, 4):
        x_T = torch.randn(g, 1, 4, device=dev)
        t = timed(lambda: e.denoise(x_T, cemb, g, timesteps=ts, sched_kind=SCHED_DDIM, coef=coef))
        print(f"{name}: {g:3d} samples, 100 steps: {t:.2f} ms", flush=True)
