"""CPU, world_size 2 over gloo: shard / gather bookkeeping of the multi-GPU path
(graspldm_amd/distributed.py).  The per-shard generator is a stand-in pure function of
(cloud, noise) -- the HIP path itself needs a GPU -- so this checks that results are
identical to the single-process run and independent of the world size."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from graspldm_amd.distributed import gather_results, generate_sharded, shard_bounds


def _fake_generate(pc, x_T):
    g = x_T.shape[0] // pc.shape[0]
    feat = pc.mean(dim=1).repeat_interleave(g, 0)                     # [n, 3]
    tm = torch.cat([feat + x_T[:, 0, :3], feat * x_T[:, 0, 1:4]], dim=1)
    return tm, x_T[:, 0, :1] - feat[:, :1]


def _fake_ddpm_generate(pc, x_T, step_noise):
    """Stand-in for a DDPM run: x_{t-1} = 0.9 x_t + 0.1 f(cloud) + 0.05 noise_t, per latent."""
    g = x_T.shape[0] // pc.shape[0]
    feat = pc.mean(dim=1).repeat_interleave(g, 0)
    x = x_T[:, 0, :3]
    for t in range(step_noise.shape[0]):
        x = 0.9 * x + 0.1 * feat + 0.05 * step_noise[t, :, 0, :3]
    return torch.cat([x, x * feat], dim=1), x[:, :1]


def _worker_ddpm(rank, world, port, B, G, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    pcs = torch.randn(B, 32, 3, generator=g)
    x_T = torch.randn(B * G, 1, 4, generator=g)
    noise = torch.randn(T, B * G, 1, 4, generator=g)
    checked = []
    tm, lg = generate_sharded(_fake_ddpm_generate, pcs, G, x_T, step_noise=noise, check=lambda: checked.append(1))
    q.put((rank, tm, lg, len(checked)))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, B, G, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    pcs = torch.randn(B, 32, 3, generator=g)
    x_T = torch.randn(B * G, 1, 4, generator=g)
    tm, lg = generate_sharded(_fake_generate, pcs, G, x_T)
    q.put((rank, tm, lg))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 4, 1])
def test_two_ranks_match_single_process(B):
    G = 3
    ctx = mp.get_context("spawn")
    outs = None
    for attempt in range(3):  # the probed port can be taken by someone else before the ranks bind it: new port, again
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, B, G, q)) for r in range(2)]
        [p.start() for p in procs]
        outs, waited = [], 0
        while len(outs) < 2 and waited < 240:
            try:
                outs.append(q.get(timeout=5))
            except Exception:
                waited += 5
                if not any(p.is_alive() for p in procs) and q.empty():
                    break   # a rank died without a result
        outs = outs if len(outs) == 2 else None
        [p.join(timeout=60) for p in procs]
        for p in procs:
            if p.is_alive():
                p.kill()   # the exact processes this test started
        if outs is not None:
            break
    assert outs is not None, "two-rank gloo run failed three times"
    g = torch.Generator().manual_seed(0)
    pcs = torch.randn(B, 32, 3, generator=g)
    x_T = torch.randn(B * G, 1, 4, generator=g)
    tm_ref, lg_ref = _fake_generate(pcs, x_T)
    for _, tm, lg in outs:
        assert torch.equal(tm, tm_ref) and torch.equal(lg, lg_ref)


def _run_two(target, args):
    ctx = mp.get_context("spawn")
    for attempt in range(3):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        q = ctx.Queue()
        procs = [ctx.Process(target=target, args=(r, 2, port) + args + (q,)) for r in range(2)]
        [p.start() for p in procs]
        outs, waited = [], 0
        while len(outs) < 2 and waited < 240:
            try:
                outs.append(q.get(timeout=5))
            except Exception:
                waited += 5
                if not any(p.is_alive() for p in procs) and q.empty():
                    break
        [p.join(timeout=60) for p in procs]
        for p in procs:
            if p.is_alive():
                p.kill()
        if len(outs) == 2:
            return outs
    return None


@pytest.mark.parametrize("B", [5, 2])
def test_ddpm_step_noise_is_world_size_independent(B):
    """DDPM draws fresh noise at every step (gaussian_diffusion.py:258-272).  Drawn globally [steps, B*G, 1, D] and
    sliced per rank (shard_step_noise), a two-rank run equals the single-process run bit for bit; the engine check hook
    runs once per rank that owns clouds."""
    from graspldm_amd.distributed import shard_step_noise
    G, T = 3, 7
    outs = _run_two(_worker_ddpm, (B, G, T))
    assert outs is not None, "two-rank gloo run failed three times"
    g = torch.Generator().manual_seed(0)
    pcs = torch.randn(B, 32, 3, generator=g)
    x_T = torch.randn(B * G, 1, 4, generator=g)
    noise = torch.randn(T, B * G, 1, 4, generator=g)
    tm_ref, lg_ref = _fake_ddpm_generate(pcs, x_T, noise)
    for _, tm, lg, nchk in outs:
        assert torch.equal(tm, tm_ref) and torch.equal(lg, lg_ref) and nchk == 1
    sl = shard_step_noise(noise, G, 1, 2)
    assert sl.is_contiguous() and torch.equal(sl, noise[:, G:2 * G])


def test_shard_bounds_cover_everything():
    for B in (1, 7, 8, 2048):
        for W in (1, 2, 4, 8):
            spans = [shard_bounds(B, W, r) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(hi - lo <= per for lo, hi, per in spans)


def test_gather_single_process_is_identity():
    rows = torch.arange(21.).view(3, 7)
    assert torch.equal(gather_results(rows, 4, 3), rows)


def test_bench_spawns_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` with no outside launcher starts two ranks itself (gloo here via --dry-run) and
    prints exactly one JSON line whose n_gpus is the world size the process group saw."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2",
                        "--clouds-per-gpu", "3", "--grasps", "5"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] and rec["gather_ok"] and rec["rows_gathered"] == 2 * 3 * 5


def test_bench_eight_ranks_dry_run():
    """The shape of the driver's 8-GPU run (BASELINE.json configs[3]: 8 ranks, one shard each) on gloo: eight ranks
    rendezvous, every one contributes its rows, one JSON line comes out.  No node with 8 GPUs has been available to any
    round: this is what makes sure the first real run cannot fail on rank bookkeeping."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "2",
                        "--clouds-per-gpu", "2", "--grasps", "3"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["dry_run"] and rec["gather_ok"] and rec["rows_gathered"] == 8 * 2 * 3


def test_noise_base_of_the_last_rank_of_eight():
    """BASELINE.json configs[4] on 8 GPUs: 64 clouds x 200 grasps = 12,800 latents per rank.  The in-kernel step noise of
    latent i is a function of (seed, noise_base + local index, step): rank 7's base is its first GLOBAL latent, the bases
    tile [0, 102400) without gap or overlap, and they match the rows shard_noise hands the same rank."""
    from graspldm_amd.distributed import shard_noise, shard_noise_base
    B, G, W = 8 * 64, 200, 8
    spans = [shard_bounds(B, W, r) for r in range(W)]
    bases = [shard_noise_base(G, lo) for lo, _, _ in spans]
    assert bases[7] == 7 * 12800 and bases[0] == 0
    assert all(b + (hi - lo) * G == nb for b, (lo, hi, _), nb in zip(bases, spans, bases[1:] + [B * G]))
    x_T = torch.arange(B * G, dtype=torch.float32).view(-1, 1, 1)
    for (lo, hi, _), base in zip(spans, bases):
        rows = shard_noise(x_T, G, lo, hi)
        assert rows.shape[0] == 12800 and int(rows[0, 0, 0]) == base
    # ragged: 13 clouds over 8 ranks, the last ranks own one or no cloud; bases still tile the latents
    spans = [shard_bounds(13, 8, r) for r in range(8)]
    assert [shard_noise_base(5, lo) for lo, _, _ in spans] == [0, 10, 20, 30, 40, 50, 60, 65]
    assert isinstance(shard_noise_base(200, 2 ** 40), int) and shard_noise_base(200, 2 ** 40) == 200 * 2 ** 40


def test_bench_workload_label_follows_arguments():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert "configs[2]" in b.workload_label(256, 20, 1024, 100, "ddim")
    assert "configs[4]" in b.workload_label(8, 200, 4096, 1000, "ddpm")
    assert "configs[2]" not in b.workload_label(8, 200, 4096, 1000, "ddpm")
    assert "custom" in b.workload_label(256, 20, 1024, 50, "ddim")
    assert "ppc_1a" in b.workload_label(256, 20, 1024, 1000, "ddpm", "ppc") and "not a BASELINE.json configuration" in b.workload_label(256, 20, 1024, 1000, "ddpm", "ppc")
    assert b.denoiser_executed_mfma_flop_fpc() == 6193152 and 0.7 < b.denoiser_executed_mfma_flop_fpc() / b.DENOISER_FLOP_PER_LATENT_STEP < 0.9
    assert 0.9 < b.denoiser_executed_mfma_flop_l16() / b.DENOISER_FLOP_PER_LATENT_STEP_PPC < 1.3
    # the CPU leg sizes itself by the CPUs the job may use (affinity mask, cgroup quota), never more than the host reports
    assert 1 <= b.usable_cores() <= (os.cpu_count() or 1)
