"""Debug: pose decoder on the 16-position 64-column engine vs the golden, row by row."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, load_schema
from graspldm_amd.r1d import R1dEngine, pack_resnet1d
from graspldm_amd.synthetic import synthetic_state_dict
sd = synthetic_state_dict(load_schema("schema_fpc_ldm.json"), seed=0)
p = "vae_model.decoder."
dec = R1dEngine(pack_resnet1d(sd, p + "net.", groups=4, seq_len=16, decoder=dict(
    in_w=sd[p + "in_layer.weight"], in_b=sd[p + "in_layer.bias"], tmrp_w=sd[p + "tmrp.weight"],
    tmrp_b=sd[p + "tmrp.bias"], cls_w=sd[p + "class_logits.weight"], cls_b=sd[p + "class_logits.bias"])), "cuda:0")
g = load_golden("decoder.npz")
print("z_h", tuple(g["z_h"].shape), "z_cond", tuple(g["z_cond"].shape))
tmrp, logit = dec.decode(g["z_h"].cuda(), dec.cond_embed(g["z_cond"].cuda()), 1)
err = (tmrp.cpu() - g["tmrp"]).abs().max(dim=1).values
print("row errors:", [f"{e:.2e}" for e in err.tolist()])
for n in (1, 2, 3, 4, 5, 8):
    t2, _ = dec.decode(g["z_h"][:n].cuda(), dec.cond_embed(g["z_cond"][:n].cuda()), 1)
    e2 = (t2.cpu() - g["tmrp"][:n]).abs().max(dim=1).values
    print(n, [f"{e:.1e}" for e in e2.tolist()])
