#!/usr/bin/env python3
"""bench.py -- grasps/sec of the GraspLDM generation hot path on MI355X.

One "step" = one full pass of the hot path over one batch of synthetic clouds that are
already resident in HBM: PVCNN encoder -> 100 DDIM steps of the latent denoiser (one fused
launch) -> pose decoder -> pose epilogue (-> all-gather of the result rows when N > 1).
Workload per GPU: 256 synthetic 1024-point clouds x 20 grasps (BASELINE.json configs[2];
configs[3] = the same 256 clouds per GPU on 8 GPUs => weak scaling).

Prints ONE JSON line (rank 0) with the driver's keys plus
  roofline      dominant kernel (the fused denoise loop, f32 MFMA bound), timed with HIP
                events on the launch stream inside this run
  cpu_baseline  the CPU oracle (a torch-CPU port of the reference graph) timed on this
                box's host cores on a bounded sample (rank 0, N = 1 only)
  kernels       extra per-kernel roofline records (set-abstraction gather: HBM bound, ...)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix), spec
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: Peak BF16/FP16 MFMA, dense (the headline 5 PF is with 2:1 sparsity)
# The GEMMs run on the f16 matrix pipe with every f32 operand split into two f16 numbers (hi + lo: 22 significant bits)
# and THREE partial products per f32 product (hi hi, hi lo, lo hi; f32 accumulation): the ceiling for algorithmic f32
# FLOP through that scheme is the f16 dense peak / 3.  (Rounds 3-4 split into three bf16 numbers and issued six.)
SPLIT_PRODUCTS = 3
PEAK_SPLIT_TFLOPS = PEAK_F16_MFMA_TFLOPS / SPLIT_PRODUCTS
PEAK_HBM_GBS = 8000.0          # HBM3E peak, spec
DENOISER_FLOP_PER_LATENT_STEP = 7_589_120   # SURVEY.md Appendix B (matches torch flop counter, tests/golden/r1d_flops.json)
DENOISER_FLOP_PER_LATENT_STEP_PPC = 30_783_104   # the ppc experiment's 16-position denoiser (tests/golden/r1d_flops.json)
DECODER_FLOP_PER_GRASP = 30.7e6


def denoiser_executed_mfma_flop_l16():
    """The same count for a 16-position net on the 64-column engine (ppc denoiser; the pose decoder's trunk is the same
    graph): every (tap, position) product of a k = 3 conv is issued (48 per output position block of 16; the two that fall on
    the zero entries beside a sample's ends included), a 16-channel level as one zero-padded 32-channel block, the attention
    products (16 x 16 per head and sample) on the f32 matrix pipe."""
    dims = (16, 32, 64, 128, 256)
    pad = lambda c: max(c, 32)
    k3 = sum(4 * pad(c) * c for c in dims[:4]) + 2 * 256 * 256 + sum(pad(a) * b for a, b in zip(dims[:4], dims[1:]))
    flop = 2 * 16 * 3 * k3                                        # k = 3 convs: 16 positions x 3 taps
    flop += 2 * 16 * sum(384 * pad(c) + 128 * max(c, 16) for c in dims[:4])   # qkv and to_out 1x1 convs
    flop += 4 * 4 * 2 * (16 * 16 * 32 + 32 * 16 * 16)             # attention: 4 levels x 4 heads x (K^T Q + V A)
    flop += 2 * sum(2 * c * 64 for c in (16, 16, 32, 32, 64, 64, 128, 128, 256))   # scale/shift rows (E = 64), per sample
    return flop


def denoiser_executed_mfma_flop_fpc():
    """FLOP the position-major engine really issues on the matrix pipes per latent and step (f32-equivalent: one per
    algorithmic multiply-add pair, before the x3 of the split): the k = 3 convs multiply 10 of their 12 (tap, position)
    pairs (the zero-padding taps are never issued), the 1x1 convs all of theirs, the to_out conv of the 4-channel level a
    padded 16-row m-tile; the scale/shift Linears (f32 MFMA, per sample); the conditioning Linear and the time MLP are
    hoisted out of the step loop and the attention cores / 4-channel ResnetBlocks run on the VALU."""
    dims = (32, 64, 128, 256)
    k3 = 4 * sum(c * c for c in dims[:3]) + 2 * 256 * 256 + sum(a * b for a, b in zip(dims[:3], dims[1:]))
    flop = 2 * 10 * k3                                   # k = 3 convs of the 32..256-channel levels
    flop += 2 * 4 * sum(384 * c + 128 * c for c in dims[:3])   # qkv and to_out 1x1 convs, 4 positions
    flop += 2 * 4 * (384 * 4 + 128 * 16)                 # the 4-channel level's qkv (K = 4) and padded to_out
    flop += 2 * sum(2 * c * 16 for c in (32, 32, 64, 64, 128, 128, 256))   # scale/shift rows, per sample
    return flop
CPU_BASELINE_CLOUDS = 128                   # bounded sample of the same workload for the CPU leg
ENCODER_FLOP_PER_CLOUD = 8.115e9            # the reference graph (shipped fpc config, N = 1024)
# executed: conv_downscale (1536 -> 768) and out_layer[0] (768 -> 3) are folded into one 1536 -> 3 GEMM
ENCODER_FLOP_EXECUTED_PER_CLOUD = 8.115e9 - 2 * 768 * 1536 * 1024 - 2 * 3 * 768 * 1024 + 2 * 3 * 1536 * 1024


def usable_cores():
    """CPUs this process may actually use: the smaller of the affinity mask and the cgroup's CPU-time quota (cpu.max /
    cfs_quota), not os.cpu_count() -- a container on a 256-core host is usually granted far fewer."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p_))
        except (OSError, ValueError):
            pass
    return max(1, n)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--clouds-per-gpu", type=int, default=256)
    ap.add_argument("--grasps", type=int, default=20)
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--experiment", choices=["fpc", "ppc"], default="fpc",
                    help="shipped experiment: fpc (4-dim grasp latent, 3 x 64 cloud latent; BASELINE.json's workload) or ppc "
                         "(configs/generation/partial_pc/ppc_1a_..._z16_pc256_180k: 16-dim latent, 3 x 256 cloud latent, "
                         "DDPM; defaults to --scheduler ddpm --ddim-steps 1000 on partial clouds)")
    ap.add_argument("--ddim-steps", type=int, default=None, help="inference steps (default 100; ppc: 1000)")
    ap.add_argument("--scheduler", choices=["ddim", "ddpm"], default=None,
                    help="default ddim (ppc: ddpm); ddpm with --ddim-steps 1000 --points 4096 --grasps 200 is BASELINE.json configs[4]")
    ap.add_argument("--noise", choices=["tensor", "kernel", "device"], default="tensor",
                    help="DDPM step noise: a [steps, B G, 1, D] torch.randn tensor per batch (the reference's draws, default) or "
                         "drawn inside the fused launch (gldm_denoise_rng: Philox4x32-10 keyed on the global latent index)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--minimal", action="store_true",
                    help="only the timed steps + the roofline launches of the dominant kernel (no stage split, no "
                         "set-abstraction / one-object records): what the rocprofv3 --stats run uses, so that the "
                         "kernel's average duration in the trace is over bench-workload launches only")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / collective plumbing only, on CPU over gloo (no HIP, no numbers): used by the tests")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the steps alternate over (2: batch k+1's encoder overlaps batch k's denoise tail)")
    a = ap.parse_args()
    if a.scheduler is None:
        a.scheduler = "ddpm" if a.experiment == "ppc" else "ddim"
    if a.ddim_steps is None:
        a.ddim_steps = 1000 if a.experiment == "ppc" else 100
    return a


def event_time(fn, iters, lead=1):
    """HIP-event time per call.  `lead` untimed calls go first, without a sync: the GPU is still busy with them while
    the host queues the timed ones, so a multi-launch stage is timed at its GPU cost, not at the host's launch pace."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(lead):
        fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def count_gpus_without_hip():
    """GPU nodes of the KFD topology (`simd_count` > 0), read from sysfs: no HIP / torch.cuda call, so the launcher parent
    really never touches the GPU.  None when the topology is not readable (the ranks then find out for themselves)."""
    import glob
    n, seen = 0, False
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(line.split(None, 1) for line in open(path).read().splitlines() if " " in line)
        except OSError:
            continue
        seen = True
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    return n if seen else None


def spawn_ranks(args):
    """`python bench.py --gpus N` without an outside launcher: N child processes of this script, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), started BEFORE anything here touches the GPU
    (this parent never does -- devices are counted from sysfs -- and nothing is re-exec'd).  Rank 0's JSON line is
    relayed.  All children are polled: the first one to fail ends the run at once (the others are terminated; a rank
    that died before the rendezvous would otherwise leave rank 0 waiting for the process-group timeout)."""
    import socket
    import subprocess
    import threading
    if not args.dry_run:
        have = count_gpus_without_hip()
        if have is not None and have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) in the KFD topology")
    # The rendezvous port: picked by binding port 0, and HELD (SO_REUSEADDR, listening) until the ranks have been started,
    # so that no other process of this host can be handed the same number in between; rank 0's TCPStore binds it with
    # SO_REUSEADDR as well.  The user's HSA_ENABLE_IPC_MODE_LEGACY, if set, is kept.
    sk = socket.socket()
    sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    sk.close()
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for i, p in enumerate(procs):   # the exact children started above
                if codes[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    # ONE JSON line on stdout: anything else rank 0's libraries wrote there (gloo / RCCL banners) goes to stderr
    for line in out0:
        line = line.rstrip("\n")
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr)
    sys.stdout.flush()
    if any(codes):
        raise SystemExit(f"bench.py: rank exit codes {codes}")


def dry_run(args, world, rank):
    """The N > 1 plumbing on CPU (gloo): barriers, the result all-gather, MAX-over-ranks timing, one JSON line."""
    import torch.distributed as dist
    from graspldm_amd.distributed import gather_results
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=__import__("datetime").timedelta(seconds=120))
    B, G = args.clouds_per_gpu, args.grasps
    rows = torch.full((B * G, 7), float(rank))
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = gather_results(rows, B * G, world * B * G)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ok = all(bool((out[r * B * G:(r + 1) * B * G] == r).all()) for r in range(world))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        n_seen = dist.get_world_size()
        dist.destroy_process_group()
    else:
        n_seen = 1
    if rank == 0:
        print(json.dumps(dict(metric="dry run (no GPU work)", value=None, unit="grasps/s", n_gpus=n_seen, steps=args.steps,
                              warmup=args.warmup, ms_per_step=dt / max(args.steps, 1) * 1e3, dry_run=True, gather_ok=ok,
                              rows_gathered=int(out.shape[0]))))


def workload_label(B, G, N, S, sched, experiment="fpc"):
    """Which BASELINE.json configuration (if any) the arguments are."""
    if experiment == "ppc":
        return ("the reference's second shipped experiment, configs/generation/partial_pc/ppc_1a_partial_63cat8k_filtered_"
                "latentc3_z16_pc256_180k.py (16-dim grasp latent, 3 x 256 cloud latent, DDPM): not a BASELINE.json configuration")
    if (N, S, sched, G) == (1024, 100, "ddim", 20):
        if B == 256:
            return "BASELINE.json configs[2] per GPU (configs[3] = this per-GPU workload on 8 GPUs)"
        if B == 1:
            return "BASELINE.json configs[1] (one object)"
    if (N, S, sched, G) == (4096, 1000, "ddpm", 200):
        return (f"BASELINE.json configs[4] per GPU: {B} clouds per GPU (4096-pt partial clouds, 1000 DDPM steps, 200 grasps "
                "per cloud; the configuration is quoted on 8 GPUs)")
    return "custom arguments (not a BASELINE.json configuration)"


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}",
              file=sys.stderr)
    # Host threads: the ranks of a node share the CPUs this job may use (16 on the GPU boxes, whatever the host reports);
    # torch's default of one intra-op thread per reported core (256) turns every host-side tensor op of the set-up into
    # an oversubscribed parallel region -- per rank
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    torch.set_num_threads(max(1, min(16, usable_cores() // max(1, local_world))))
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev, timeout=__import__("datetime").timedelta(seconds=300))
        world = dist.get_world_size()   # n_gpus in the line = the ranks RCCL saw

    from graspldm_amd import _lib
    _lib.lib()  # fail loudly here if the HIP library is missing
    from graspldm_amd.distributed import gather_results
    from graspldm_amd.pipeline import build_fpc_ldm
    from graspldm_amd.r1d import pose_epilogue
    from graspldm_amd.synthetic import synthetic_batch

    B, G, N, S = args.clouds_per_gpu, args.grasps, args.points, args.ddim_steps
    ppc = args.experiment == "ppc"
    D = 16 if ppc else 4   # grasp latent size = positions of the denoiser
    ldm = build_fpc_ldm(n_points=N, scheduler=args.scheduler, device=dev, **(dict(latent=16, pc_latent=256) if ppc else {}))
    ldm.set_inference_timesteps(S)
    uniq = min(B, 32)   # 32 distinct synthetic objects per rank, tiled to B (resident in HBM)
    partial = ppc or (N, S, args.scheduler) == (4096, 1000, "ddpm")   # ppc / configs[4]: camera-facing side, resampled with duplicates
    pcs_u, metas_u = synthetic_batch(uniq, N, partial=partial, first_index=rank * uniq)
    reps = (B + uniq - 1) // uniq
    pcs = pcs_u.repeat(reps, 1, 1)[:B].contiguous().to(dev)
    gmean = metas_u["grasp_mean"].repeat(reps, 1)[:B].contiguous().to(dev)
    gstd = metas_u["grasp_std"].repeat(reps, 1)[:B].contiguous().to(dev)
    xgen = torch.Generator().manual_seed(1234 + rank)
    x_T = torch.randn(B * G, 1, D, generator=xgen).to(dev)   # the stage records below; the timed batches draw their own
    den = ldm.diffusion_model.model
    eng = den.engine(dev)
    from graspldm_amd import _lib as L
    pm_engine = L.lib().gldm_r1d_tile_columns(eng._desc_ptr()) == 64   # position-major split-f16 engine (else: sample-major, f32 pipe)

    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else None
    counter = [0]

    # in-kernel noise: the seed of the job, the rank's first latent as the counter base (results do not depend on the world size)
    noise_kw = dict(noise_source="kernel", noise_seed=1234, noise_base=rank * B * G) \
        if (args.noise in ("kernel", "device") and args.scheduler == "ddpm") else {}   # "device" = "kernel"

    def step():
        if streams is None:
            return one_batch()
        st = streams[counter[0] % len(streams)]
        counter[0] += 1
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            return one_batch()

    def one_batch():
        # x_T is drawn per batch on the CPU generator and uploaded, as sample() does (gaussian_diffusion.py:253: part of the
        # step since round 6); DDPM: the per-step noise of every latent is drawn inside sample(), on the device, every batch
        xb = torch.randn(B * G, 1, D, generator=xgen).pin_memory().to(dev, non_blocking=True)
        (tm, lg), _ = ldm.generate_grasps(pcs, num_grasps=G, x_T=xb, **noise_kw)
        # the epilogue runs on the rank's OWN rows; what crosses xGMI is one all-gather of the result rows [B G, 7]
        # (the latent-level outputs the north star names: 28 bytes per grasp), nothing is repeated per rank
        H, un, conf = pose_epilogue(tm, lg, gmean, gstd, G)
        if world > 1:
            gather_results(torch.cat([tm, lg], dim=1), B * G, world * B * G)
        return H

    if streams is not None:
        # setup, not a step: each stream's first batch allocates its memory pool and its engine scratch
        # (hipMalloc, ~15 ms per stream); done here so it cannot land in the timed region for small --warmup
        for st in streams:
            st.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(st):
                one_batch()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    H_last = None
    for _ in range(args.steps):
        H_last = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the timed batches must be real results: no lost step-segment hand-off (the engine's error word), finite poses
    eng.check()
    if H_last is not None and not bool(torch.isfinite(H_last).all()):
        raise SystemExit("bench.py: non-finite poses in the timed region")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    grasps_per_s = world * B * G * args.steps / dt

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel: the fused denoise loop
        z = ldm.vae_model.encode_pc(pcs)
        cemb = eng.cond_embed(z)
        ts, coef = ldm.diffusion_model._schedule(dev)
        from graspldm_amd.r1d_pack import SCHED_DDIM, SCHED_DDPM
        kind = SCHED_DDIM if args.scheduler == "ddim" else SCHED_DDPM
        noise = torch.randn((ts.numel(), B * G, 1, D), device=dev) if kind == SCHED_DDPM else None
        dn = lambda: eng.denoise(x_T, cemb, G, timesteps=ts, sched_kind=kind, coef=coef, step_noise=noise)
        dn()
        t_den = event_time(dn, 3, lead=0)
        flop_ls = DENOISER_FLOP_PER_LATENT_STEP_PPC if ppc else DENOISER_FLOP_PER_LATENT_STEP
        flop = B * G * S * flop_ls
        # memory-side bytes of one launch from the committed PMC passes (tools/pmc_denoise.sh: FETCH_SIZE, doubled
        # per the gfx950 16-B/lane rule, + WRITE_SIZE); only quoted for the workload they were collected on
        traffic = None
        traffic_source = None
        import glob
        for pmc_path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_denoise_pmc.json")), reverse=True):
            pmc = json.load(open(pmc_path))
            if pm_engine and pmc.get("n_latents") == B * G and pmc.get("steps") == S and args.scheduler == "ddim":
                traffic = pmc["fetch_bytes_corrected"] + pmc["write_bytes"]
                traffic_source = ("static: " + os.path.relpath(pmc_path, ROOT) + " (rocprofv3 --pmc passes of this workload, "
                                  "tools/pmc_denoise.sh; not re-measured in this run)")
                break
        if pm_engine:
            exec_flop = B * G * S * (denoiser_executed_mfma_flop_fpc() if D == 4 else denoiser_executed_mfma_flop_l16())
            roof = dict(kernel="r1d_kernel<64, %d> (gldm_denoise: %d %s steps fused, 64-column tiles = %s, split-f16 GEMMs)"
                               % (D, S, args.scheduler.upper(), "16 samples x 4 positions" if D == 4 else "4 samples x 16 positions"),
                        bound="mfma",
                        achieved=flop / t_den / 1e12, peak=PEAK_SPLIT_TFLOPS, unit="TFLOP/s",
                        frac=flop / t_den / 1e12 / PEAK_SPLIT_TFLOPS, traffic=traffic, traffic_source=traffic_source,
                        algorithmic_flop_per_launch=flop, avg_launch_ms=t_den * 1e3,
                        executed=dict(f32_equivalent_flop_per_launch=exec_flop, f16_flop_per_launch=SPLIT_PRODUCTS * exec_flop,
                                      achieved_f16_tflops=SPLIT_PRODUCTS * exec_flop / t_den / 1e12, peak_f16_tflops=PEAK_F16_MFMA_TFLOPS,
                                      frac=SPLIT_PRODUCTS * exec_flop / t_den / 1e12 / PEAK_F16_MFMA_TFLOPS,
                                      note="what the matrix pipe really executes (zero-padding taps of the k = 3 convs never "
                                           "issued, conditioning Linear / time MLP hoisted out of the loop, attention cores on "
                                           "the VALU) x 3 f16 products, against the f16 dense peak"),
                        peak_note="peak = f16 dense MFMA peak (2500 TFLOP/s) / 3: the kernel computes every f32 product as three "
                                  "f16 partial products (operands split into hi + lo, 22 significant bits) with f32 accumulation, "
                                  "so 3 executed f16 FLOP per algorithmic FLOP (rounds 3-4: six bf16 products, peak 2500 / 6)",
                        timing="HIP events around 3 launches on their own (no other stream active); in the pipelined "
                               "steps the other stream's encoder kernels share the GPU with the launch",
                        flop_note=f"algorithmic FLOP = the reference graph's count ({flop_ls:,} per latent and step, = torch's "
                                  "flop counter over the reference module, which counts a k=3 conv's zero-padding taps: "
                                  "tests/golden/r1d_flops.json).  "
                                  "Arithmetic: f32 in, f32 out, f32 accumulation; products formed from f16 pieces whose dropped "
                                  "cross term is <= 2^-22 relative (parity bars unchanged: 2e-5 single forward, 1e-4 poses)")
        else:
            roof = dict(kernel="r1d_kernel<32, %d> (gldm_denoise: %d %s steps fused, sample-major 32-column tiles, f32 MFMA)" % (D, S, args.scheduler.upper()),
                        bound="mfma", achieved=flop / t_den / 1e12, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                        frac=flop / t_den / 1e12 / PEAK_F32_MFMA_TFLOPS, traffic=None, traffic_source=None,
                        algorithmic_flop_per_launch=flop, avg_launch_ms=t_den * 1e3,
                        peak_note="the 16-position latent denoiser runs on the sample-major engine: exact f32 products on "
                                  "v_mfma_f32_16x16x4_f32 (157.3 TFLOP/s dense); the split-f16 position-major engine is built "
                                  "for 4-position latents only",
                        flop_note="algorithmic FLOP = torch's flop counter over the reference's TimeConditionedResNet1D "
                                  "(dim 16, cond 256): tests/golden/r1d_flops.json")
        kernels = []
        if not args.minimal:
          # ---- stage split and the set-abstraction gather (north-star HBM kernel), same run
          t_enc = event_time(lambda: ldm.vae_model.encode_pc(pcs), 6)
          dec = ldm.vae_model.decoder
          dec_cols = L.lib().gldm_r1d_tile_columns(dec._get_engine(dev, z.shape[1])._desc_ptr())
          lat = dn().squeeze(-2)
          t_dec = event_time(lambda: dec(lat, z, samples_per_cond=G), 10, lead=2)
          if N == 1024:
              # executed FLOP per cloud by the pipe they run on (head convs folded: 1536 -> 3 instead of 1536 -> 768 -> 3)
              # (the first voxel conv, 3 -> 48 @ 24^3, is counted at its 81 real products per output, not the 96 of its
              # three k-blocks)
              f_split = 2 * 1024 * (96 * 768 + 768 * 1536) + 2 * 27 * ((3 * 48 + 48 * 48) * 24 ** 3 + 48 * 96 * 12 ** 3 + 96 * 96 * 12 ** 3)
              f_f32 = ENCODER_FLOP_EXECUTED_PER_CLOUD - f_split
              t_floor = B * (f_split / (PEAK_SPLIT_TFLOPS * 1e12) + f_f32 / (PEAK_F32_MFMA_TFLOPS * 1e12))
              enc_rec = dict(achieved=B * ENCODER_FLOP_EXECUTED_PER_CLOUD / t_enc / 1e12,
                             peak=B * ENCODER_FLOP_EXECUTED_PER_CLOUD / t_floor / 1e12, frac=t_floor / t_enc,
                             executed_flop_per_cloud=dict(split_f16_pipe=f_split, f32_pipe=f_f32))
          else:
              enc_rec = dict(achieved=None, peak=None, frac=None)
          kernels = [dict(kernel="PVCNNEncoder.forward (all launches)", bound="mfma (mixed pipes)", avg_ms=t_enc * 1e3, unit="TFLOP/s",
                          note="executed FLOP (head convs folded; the reference graph has %.3f GFLOP per cloud); peak = executed "
                               "FLOP / the time the launches would take with each GEMM at the peak of the pipe it runs on (768 -> "
                               "1536 layer, 96 -> 768 layer and the four voxel convs: split-f16, 2500 / 3 TFLOP/s; the "
                               "rest: f32 MFMA, 157.3): frac = that time / measured, memory passes counted as zero"
                               % (ENCODER_FLOP_PER_CLOUD / 1e9), **enc_rec),
                     dict(kernel="r1d_kernel<64, 16> (gldm_decode: 64-column tiles = 4 samples x 16 positions, split-f16 GEMMs)"
                                 if dec_cols == 64 else "r1d_kernel<32, 16> (gldm_decode, f32 matrix pipe)", bound="mfma",
                          avg_ms=t_dec * 1e3, achieved=B * G * DECODER_FLOP_PER_GRASP / t_dec / 1e12,
                          peak=PEAK_SPLIT_TFLOPS if dec_cols == 64 else PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                          frac=B * G * DECODER_FLOP_PER_GRASP / t_dec / 1e12 / (PEAK_SPLIT_TFLOPS if dec_cols == 64 else PEAK_F32_MFMA_TFLOPS),
                          note="algorithmic FLOP = torch's flop counter over the reference decoder trunk (tests/golden/r1d_flops.json)")]
          # ---- the same launch on the f32 matrix pipe only: a descriptor without the split-f16 weight copies (what an
          # ABI-4 packer produces) runs the sample-major engine, exact f32 fma chains -- for comparison with the split
          # arithmetic of the headline path
          if args.scheduler == "ddim" and pm_engine:
              from graspldm_amd.r1d import R1dEngine
              from graspldm_amd.r1d_pack import pack_resnet1d
              sd32 = {k: v.detach().float().cpu() for k, v in den.state_dict().items()}
              packed = pack_resnet1d(sd32, "", groups=den.groups, seq_len=den.in_features, num_steps=den.max_timesteps,
                                     cond_rows=getattr(den, "cond_rows", 3))
              for i in range(len(packed["desc"].rb)):
                  packed["desc"].rb[i].c1_w3 = packed["desc"].rb[i].c2_w3 = 0
              for i in range(len(packed["desc"].lv)):
                  packed["desc"].lv[i].qkvn_w3 = packed["desc"].lv[i].out_w3 = packed["desc"].lv[i].down_w3 = 0
              eng32 = R1dEngine(packed, dev)
              dn32 = lambda: eng32.denoise(x_T, cemb, G, timesteps=ts, sched_kind=kind, coef=coef)
              x_split, x_f32 = dn(), dn32()
              t32 = event_time(dn32, 2, lead=0)
              kernels.append(dict(kernel="gldm_denoise on the f32 matrix pipe only (r1d_kernel<32, 4>, sample-major engine)", bound="mfma",
                                  avg_ms=t32 * 1e3, achieved=flop / t32 / 1e12, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                                  frac=flop / t32 / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                  max_abs_diff_vs_split_path=float((x_split - x_f32).abs().max()),
                                  mean_abs_diff_vs_split_path=float((x_split - x_f32).abs().mean()),
                                  p999_abs_diff_vs_split_path=float(torch.quantile((x_split - x_f32).abs().flatten().float(), 0.999)),
                                  note="same weights, inputs and schedule as the headline launch; the *_diff fields compare the "
                                       "two engines' latents after all steps (different summation orders on both sides; the "
                                       "sampler's clip of x0 to +-1 makes single elements discontinuous)"))
          from graspldm_amd.pvcnn import furthest_point_sample
          pts = (pcs.transpose(1, 2) * (0.05 / 0.12)).contiguous()
          c1 = furthest_point_sample(pts, 512)
          c2 = furthest_point_sample(c1, 128)
          f1 = torch.randn(B, 128, 512, device=dev)
          Ns, Ms, Us, Cs = 512, 128, 64, 128
          grouped = torch.empty(B, 3 + Cs, Ms, Us, device=dev)
          st = L.current_stream(dev)
          sa = lambda: L.call("gldm_sa_group", L.ptr(c1), L.ptr(c2), L.ptr(f1), B, Cs, Ns, Ms, 0.4, Us, L.ptr(grouped),
                              None, st)
          sa()
          t_sa = event_time(sa, 10)
          by = B * (12 * Ns + 4 * Cs * Ns + 12 * Ms + 4 * (Cs + 3) * Ms * Us)
          sa_traffic, sa_src = None, None
          for pmc_path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_point_ops_pmc.json")), reverse=True):
              rec = json.load(open(pmc_path)).get("sa_group_kernel")
              if rec and rec.get("clouds") == B:
                  sa_traffic = rec["fetch_bytes_raw"] + rec["write_bytes"]
                  sa_src = ("static: " + os.path.relpath(pmc_path, ROOT) + " (rocprofv3 --pmc passes of tools/run_sa_once.py, tools/pmc_kernels.sh; "
                            "FETCH_SIZE as reported: the reads are 4-byte gathers, for which the gfx950 doubling is not calibrated -- "
                            f"doubled, the total is {rec['fetch_bytes_corrected'] + rec['write_bytes']} bytes)")
                  break
          kernels.append(dict(kernel="sa_group_kernel (PointNet2SSG SA2 gather: N=512 M=128 U=64 C=128)", bound="hbm",
                              avg_ms=t_sa * 1e3, achieved=by / t_sa / 1e9, peak=PEAK_HBM_GBS, unit="GB/s",
                              frac=by / t_sa / 1e9 / PEAK_HBM_GBS, algorithmic_bytes_per_launch=by,
                              traffic=sa_traffic, traffic_source=sa_src))
          # farthest point sampling (us per cloud, distance evaluations per second; B = 1 and B = batch):
          # the first sampling of PointNet2SSG / PVCNN2, 1024 -> 512 (sampling.cu:86-174)
          from graspldm_amd.backend import _backend
          crd = pcs[:, :1024].transpose(1, 2).contiguous()          # [B, 3, 1024]
          for fb in (1, B):
              cb = crd[:fb].contiguous()
              fps = lambda: _backend.furthest_point_sampling(cb, 512)
              fps()
              t_fps = event_time(fps, 10)
              kernels.append(dict(kernel=f"fps_wave_kernel<16, 0> (furthest_point_sampling 1024 -> 512, B = {fb}: one wave per cloud)",
                                  bound="latency", avg_ms=t_fps * 1e3, us_per_cloud=t_fps * 1e6 / fb, us_per_round=t_fps * 1e6 / 511,
                                  distance_evals_per_s=fb * 511 * 1024 / t_fps,
                                  note="511 dependent rounds per cloud; a round = 1024 distance updates + a wave arg-max (round 4: "
                                       "16 waves + LDS exchange + barrier per round, 0.62 ms = 1.2 us per round)"))
          # fused SA module core (gather + grouped MLP + max; the grouped tensor never reaches HBM)
          from graspldm_amd.pvcnn import PointNetSAModule, ball_query
          from graspldm_amd.sa_pack import SaMlpPlan
          from graspldm_amd.synthetic import load_synthetic_weights
          sa2 = load_synthetic_weights(PointNetSAModule(num_centers=Ms, radius=0.4, num_neighbors=Us, in_channels=Cs,
                                                        out_channels=(128, 128, 256)).eval(), seed=2).to(dev)
          plan = SaMlpPlan(sa2.mlps[0], dev)
          idx = ball_query(c2, c1, 0.4, Us)
          fsa = lambda: plan.run(c1, c2, f1, idx)
          fsa()
          t_fsa = event_time(fsa, 10)
          sa_flop = B * 2 * Ms * Us * (131 * 128 + 128 * 128 + 128 * 256)
          # executed on the matrix pipe since round 6: the first layer's feature part once per POINT (N x 128 x 128), layers
          # 2 and 3 per (centre, neighbour) pair; its three coordinate products run on the VALU of the gather threads
          sa_exec = B * 2 * (Ns * 128 * 128 + Ms * Us * (128 * 128 + 128 * 256))
          kernels.append(dict(kernel="SSG SA2 set-abstraction MLP 131-128-128-256 + max: pointwise_mlp_sp_kernel<false, 2, 1> (first layer per "
                                     "point, point-major) + sa_mlp3_kernel<1, 4, true> (gather + layers 2-3 + max, 64-column tiles on split-f16 planes)",
                              bound="mfma", avg_ms=t_fsa * 1e3, achieved=sa_flop / t_fsa / 1e12, peak=PEAK_SPLIT_TFLOPS,
                              unit="TFLOP/s", frac=sa_flop / t_fsa / 1e12 / PEAK_SPLIT_TFLOPS,
                              executed_frac=SPLIT_PRODUCTS * sa_exec / t_fsa / 1e12 / PEAK_F16_MFMA_TFLOPS,
                              hbm_bytes_avoided=B * 4 * (Cs + 3) * Ms * Us,
                              note="both launches timed together; algorithmic FLOP = the module as the reference evaluates it (every "
                                   "layer per (centre, neighbour) pair).  gldm_pointwise_mlp_f16x2_pm + gldm_sa_mlp_forward_f16x2_pre: "
                                   "W1 [x - c; f] = W1a (x - c) + W1b f, the second term once per point (round 5, all three layers per "
                                   "pair in one launch: 1.49-1.56 ms = 0.22)"))
          # ---- BASELINE.json configs[1]: ONE object (B = 1, G grasps), latency per stage and end to end
          if args.scheduler == "ddim":
              pc1, x1 = pcs[:1].contiguous(), x_T[:G].contiguous()
              z1 = ldm.vae_model.encode_pc(pc1)
              ce1 = eng.cond_embed(z1)
              dn1 = lambda: eng.denoise(x1, ce1, G, timesteps=ts, sched_kind=kind, coef=coef)
              lat1 = dn1().squeeze(-2)
              def obj1():
                  (tm1, lg1), _ = ldm.generate_grasps(pc1, num_grasps=G, x_T=x1)
                  return pose_epilogue(tm1, lg1, gmean[:1], gstd[:1], G)
              obj1()
              kernels.append(dict(kernel=f"one object end to end (BASELINE configs[1]: B=1, G={G}, {S} DDIM steps)", bound="latency",
                                  avg_ms=event_time(obj1, 5) * 1e3,
                                  stages_ms=dict(encode=event_time(lambda: ldm.vae_model.encode_pc(pc1), 5) * 1e3,
                                                 denoise=event_time(dn1, 5) * 1e3,
                                                 decode=event_time(lambda: dec(lat1, z1, samples_per_cond=G), 5) * 1e3),
                                  note="latency of a single cloud: the denoise launch is 2 tiles on 2 of 256 CUs, "
                                       "i.e. the per-step critical path of one workgroup"))
          if args.scheduler == "ddim" and N == 1024 and not ppc:
              # ---- the whole step with EXACT f32 products (numerics.f32_only(): denoiser / decoder on the sample-major f32-MFMA
              # engine, encoder GEMMs and voxel convs on their f32-pipe kernels): the reference's own arithmetic, end to end,
              # next to the headline's split-f16 products -- same weights, clouds and x_T
              from graspldm_amd import numerics

              def whole(model):
                  (tm_, lg_), _ = model.generate_grasps(pcs, num_grasps=G, x_T=x_T)
                  return pose_epilogue(tm_, lg_, gmean, gstd, G)[0]
              H_split = whole(ldm)
              with numerics.f32_only():
                  ldm32 = build_fpc_ldm(n_points=N, scheduler=args.scheduler, device=dev)
                  ldm32.set_inference_timesteps(S)
                  H_f32 = whole(ldm32)
                  t_f32 = event_time(lambda: whole(ldm32), 2, lead=0)
                  ldm32.check_engines()
              dH = (H_split - H_f32).abs()
              kernels.append(dict(kernel="whole step with exact f32 products (graspldm_amd.numerics.f32_only(): every GEMM on the f32 "
                                         "matrix pipe, k-ordered fma chains)", bound="mfma", avg_ms=t_f32 * 1e3, unit="grasps/s",
                                  grasps_per_s=B * G / t_f32, headline_over_this=(B * G / t_f32) and grasps_per_s / world / (B * G / t_f32),
                                  max_abs_pose_diff_vs_headline=float(dH.max()), mean_abs_pose_diff_vs_headline=float(dH.mean()),
                                  note="one batch at a time on one stream (no pipelining); the pose difference is between two "
                                       "summation orders, both within the 1e-4 bar of the reference's vectors (tests/test_models_gpu.py)"))
              del ldm32
              # ---- the same experiment conditioned by the set-abstraction encoder family (north star: FPS / ball query /
              # grouped SA MLPs conditioning the VAE): PVCNN2Encoder in its repaired form at half width / half resolution
              # (the reference's encoder benchmark setting); parity: tests/test_models_gpu.py::
              # test_ldm_end_to_end_with_the_set_abstraction_encoder
              ldm_sa = build_fpc_ldm(n_points=N, scheduler=args.scheduler, device=dev, encoder="PVCNN2Encoder", encoder_scale=(0.5, 0.5))
              ldm_sa.set_inference_timesteps(S)
              whole(ldm_sa)
              t_sa_all = event_time(lambda: whole(ldm_sa), 3)
              t_sa_enc = event_time(lambda: ldm_sa.vae_model.encode_pc(pcs), 3)
              ldm_sa.check_engines()
              kernels.append(dict(kernel="whole step conditioned by PVCNN2Encoder (set abstraction + PVConv + feature propagation, "
                                         "scale_channels 0.5, scale_voxel_resolution 0.5)", bound="mfma", avg_ms=t_sa_all * 1e3,
                                  unit="grasps/s", grasps_per_s=B * G / t_sa_all, stages_ms=dict(encode=t_sa_enc * 1e3),
                                  note="one batch at a time on one stream; denoise and decode launches are the headline's"))
              del ldm_sa
        # ---- CPU baseline: the torch-CPU oracle on this box's host cores, bounded sample
        cpu = None
        if world == 1 and not args.no_cpu_baseline and args.scheduler == "ddim" and not ppc:
            # separate CPU-only processes on a bounded sample (full S steps): once ONE process with 16 threads on 128 clouds
            # (the oracle's ~110 small ops per step do not scale past a few cores: the figure of the earlier rounds) and, where
            # this process may use more CPUs than that, once with ALL of them as cores / 16 such processes side by side, 32
            # clouds each (clouds sharded over processes, the way a CPU deployment would use them); `value` is the better.
            # "May use": the GPU boxes report 256 host cores, but their cgroup grants this job 16 CPUs of time (cpu.max
            # 1600000 100000) -- hundreds of threads on that quota is why the all-threads runs of earlier builds never finished
            import subprocess
            ncores = usable_cores()
            runs = []

            def oracle_cmd(clouds, threads):
                return [sys.executable, "-m", "oracle.cpu_baseline", "--clouds", str(clouds), "--grasps", str(G), "--points", str(N),
                        "--ddim-steps", str(S), "--threads", str(threads)]
            t16 = min(ncores, 16)
            try:
                r = subprocess.run(oracle_cmd(CPU_BASELINE_CLOUDS, t16), cwd=ROOT, capture_output=True, text=True, timeout=120)
                rec = json.loads(r.stdout.strip().splitlines()[-1])
                runs.append(dict(threads=rec["threads"], processes=1, clouds=CPU_BASELINE_CLOUDS, value=rec["grasps"] / rec["seconds"],
                                 seconds=rec["seconds"]))
            except Exception as e:  # noqa: BLE001
                runs.append(dict(threads=t16, processes=1, value=None, error=f"{e!r}"[:160]))
            nproc = ncores // 16
            if nproc >= 2:
                per = max(CPU_BASELINE_CLOUDS // 4, 1)
                procs = [subprocess.Popen(oracle_cmd(per, 16), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                         for _ in range(nproc)]
                recs, deadline = [], time.time() + 120
                for pr in procs:
                    try:
                        out_s, _ = pr.communicate(timeout=max(1.0, deadline - time.time()))
                        recs.append(json.loads(out_s.strip().splitlines()[-1]))
                    except Exception:  # noqa: BLE001
                        pr.kill()
                        pr.wait()
                if len(recs) == nproc:
                    secs = max(r_["seconds"] for r_ in recs)
                    runs.append(dict(threads=16 * nproc, processes=nproc, clouds=per * nproc,
                                     value=sum(r_["grasps"] for r_ in recs) / secs, seconds=secs))
                else:
                    runs.append(dict(threads=16 * nproc, processes=nproc, value=None, error=f"{nproc - len(recs)} of {nproc} processes without a result within 120 s"))
            good = [r for r in runs if r.get("value")]
            best = max(good, key=lambda r: r["value"]) if good else None
            cpu = dict(value=best["value"] if best else None, unit="grasps/s", cores=best["threads"] if best else runs[0]["threads"], kind="port",
                       sample=f"{CPU_BASELINE_CLOUDS} clouds x {G} grasps, N={N}, {S} DDIM steps, torch-CPU oracle "
                              f"(oracle/torch_ref.py + oracle/point_ops.c); this job may use {ncores} of the host's {os.cpu_count()} cores "
                              f"(affinity mask and cgroup CPU quota); runs: "
                              + "; ".join(f"{r['processes']} x {r['threads'] // r['processes']} threads: "
                                          + (f"{r['value']:.1f} grasps/s ({r['clouds']} clouds in {r['seconds']:.1f} s)" if r.get("value") else "no result within 120 s")
                                          for r in runs),
                       runs=runs)
        out = dict(metric="grasps/sec whole-node (%d-pt cloud, %d %s steps)" % (N, S, args.scheduler.upper()), value=grasps_per_s,
                   unit="grasps/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step,
                   higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype=("f32 (GEMMs as split-f16 x3 partial products on the f16 matrix pipe, f32 accumulation; GroupNorm / "
                          "LayerNorm / attention / scheduler in f32)") if pm_engine else
                         "f32 (denoiser / decoder GEMMs on the f32 matrix pipe; the encoder's wide GEMMs as split-f16 x3 partial products)",
                   data="synthetic",
                   config=dict(workload=f"LDM mode, {B} synthetic {'partial ' if partial else ''}{N}-pt clouds per GPU x {G} grasps, "
                                        f"{S} {args.scheduler.upper()} steps: " + workload_label(B, G, N, S, args.scheduler, args.experiment),
                               experiment=args.experiment, clouds_per_gpu=B, grasps_per_cloud=G, points=N, ddim_steps=S,
                               encoder="PVCNNEncoder (shipped %s config)" % args.experiment, parallelism=f"cloud-sharded x{world}",
                               weights="synthetic recipe seed 0", streams=args.streams,
                               inputs=f"{uniq} distinct synthetic clouds per rank tiled to {B} (resident in HBM); x_T drawn per batch on the "
                                      "CPU generator and uploaded inside the timed step, as the reference's sample() does "
                                      f"(gaussian_diffusion.py:253: {B * G * D * 4} bytes per batch)"
                                      + (("; DDPM per-step noise drawn inside the fused launch (counter-based generator)" if noise_kw else
                                          "; DDPM per-step noise drawn on the device inside every timed batch") if args.scheduler == "ddpm" else "")),
                   roofline=roof, cpu_baseline=cpu, kernels=kernels)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
