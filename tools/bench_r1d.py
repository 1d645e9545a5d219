"""Quick throughput probe of the fused denoise loop and decoder."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graspldm_amd.synthetic import synthetic_state_dict
from graspldm_amd.r1d import R1dEngine, pack_resnet1d, SCHED_DDIM
from graspldm_amd.diffusion import make_schedule_tables

schema = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "schema_fpc_ldm.json")))
sd = synthetic_state_dict({k: (tuple(s), getattr(torch, d)) for k, (s, d) in schema.items()}, 0)
den = R1dEngine(pack_resnet1d(sd, "diffusion_model.model.", groups=4, seq_len=4, num_steps=1000), "cuda:0")
p = "vae_model.decoder."
dec = R1dEngine(pack_resnet1d(sd, p + "net.", groups=4, seq_len=16, decoder=dict(
    in_w=sd[p + "in_layer.weight"], in_b=sd[p + "in_layer.bias"], tmrp_w=sd[p + "tmrp.weight"],
    tmrp_b=sd[p + "tmrp.bias"], cls_w=sd[p + "class_logits.weight"], cls_b=sd[p + "class_logits.bias"])), "cuda:0")
ts, coef = make_schedule_tables("ddim", 1000, 5e-5, 1e-3, "linear", "fixed_large", 100)
ts, coef = ts.cuda(), coef.cuda()
for B in (1, 13, 205, 256, 512):
    G = 20
    z = torch.randn(B, 3, 64, device="cuda")
    x = torch.randn(B * G, 1, 4, device="cuda")
    cemb = den.cond_embed(z)
    f = lambda: den.denoise(x, cemb, G, timesteps=ts, sched_kind=SCHED_DDIM, coef=coef)
    f(); torch.cuda.synchronize()
    t0 = time.time(); f(); torch.cuda.synchronize(); dt = time.time() - t0
    fl = B * G * 100 * 7589120
    cd = dec.cond_embed(z)
    zh = torch.randn(B * G, 4, device="cuda")
    g = lambda: dec.decode(zh, cd, G)
    g(); torch.cuda.synchronize()
    t0 = time.time(); g(); torch.cuda.synchronize(); dt2 = time.time() - t0
    print(f"B={B} latents={B*G}: denoise100 {dt*1e3:.2f} ms  {fl/dt/1e12:.1f} TFLOP/s  {B*G/dt:.0f} latents/s | decode {dt2*1e3:.2f} ms {B*G*30.7e6/dt2/1e12:.1f} TF")
