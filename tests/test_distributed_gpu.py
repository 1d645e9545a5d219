"""GPU, 2 ranks over RCCL (skipped on a 1-GPU box): generate_sharded on two MI355X returns, on every rank, exactly
the rows the single-GPU run produces (clouds are independent end to end; x_T is drawn globally and sliced)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RANK_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert dist.is_available()
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", device_id=dev)
from graspldm_amd.distributed import generate_sharded
from graspldm_amd.pipeline import build_fpc_ldm
from graspldm_amd.synthetic import synthetic_batch
ldm = build_fpc_ldm(device=dev)
ldm.set_inference_timesteps(20)
B, G = int(sys.argv[2]), 4
pcs, _ = synthetic_batch(B, 1024)
x_T = torch.randn(B * G, 1, 4, generator=torch.Generator().manual_seed(5))
gen = lambda pc, xt: ldm.generate_grasps(pc, num_grasps=G, x_T=xt)[0]
tm, lg = generate_sharded(gen, pcs.to(dev), G, x_T)
dist.barrier()
dist.destroy_process_group()
tm1, lg1 = gen(pcs.to(dev), x_T)   # the whole batch on this GPU alone
assert torch.equal(tm, tm1) and torch.equal(lg, lg1), (rank, (tm - tm1).abs().max().item())
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_generation_equals_single_gpu(tmp_path, world):
    """Every world size the box can host (2, 4, 8 ranks; skipped beyond the visible device count): equal shards (world
    divides B) and, at 4 ranks, a ragged one (B = 6: the last rank owns no cloud and still joins the all-gather)."""
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    B = {2: 6, 4: 6, 8: 16}[world]
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(B)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
