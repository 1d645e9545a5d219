"""Multi-GPU generation: one process per GPU; clouds are independent end to end
(eval-mode BatchNorm, per-sample Group/LayerNorm), so the batch is split
contiguously over ranks with a full weight replica each and ONE collective at the
end: all_gather_into_tensor of the packed result rows [B_r*G, 7] (tmrp + logit,
28 B per grasp) over RCCL/xGMI (gloo on CPU in the tests).  The reference has no
multi-GPU inference path."""
import torch
import torch.distributed as dist


def shard_bounds(num_clouds, world_size, rank):
    """Contiguous split; every rank gets ceil(B/W) slots, the tail ranks may own fewer
    (possibly zero) real clouds."""
    per = (num_clouds + world_size - 1) // world_size
    lo = min(rank * per, num_clouds)
    hi = min(lo + per, num_clouds)
    return lo, hi, per


def shard_noise(x_T, num_grasps, lo, hi):
    """Slice the globally drawn x_T [B*G, 1, D] so results do not depend on world size."""
    return x_T[lo * num_grasps:hi * num_grasps]


def shard_step_noise(step_noise, num_grasps, lo, hi):
    """Slice the globally drawn DDPM per-step noise [steps, B*G, 1, D] (the draws of gaussian_diffusion.py:258-272, one
    per step and latent) along the latent axis: rank r sees exactly the rows its latents would see in a single-process
    run, so a DDPM run is as independent of the world size as a DDIM one.  Contiguous copy: the fused launch indexes it
    as [step][local latent]."""
    return step_noise[:, lo * num_grasps:hi * num_grasps].contiguous()


def shard_noise_base(num_grasps, lo):
    """`noise_base` of a rank whose first cloud is `lo` (shard_bounds): the global index of its first latent.  With
    noise_source="kernel" the step noise of latent i is a function of (seed, noise_base + local index, step) only, so ranks
    that pass this and share a seed draw exactly the single-process run's noise (int64 on the device side: 12,800 latents
    per rank x 8 ranks x any batch count stays far below 2^63)."""
    return int(lo) * int(num_grasps)


def gather_results(local_rows, per_rank_rows, total_rows, group=None):
    """local_rows [n_r, 7] (n_r <= per_rank_rows) -> [total_rows, 7] on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_rows[:total_rows]
    if local_rows.shape[0] == per_rank_rows:   # the usual case (equal shards): nothing to pad, nothing to allocate
        pad = local_rows.contiguous()
    else:
        pad = torch.zeros((per_rank_rows, local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
        pad[: local_rows.shape[0]] = local_rows
    out = torch.empty((world * per_rank_rows, local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return out[:total_rows]


@torch.no_grad()
def generate_sharded(generate_fn, pcs, num_grasps, x_T=None, group=None, step_noise=None, check=None):
    """generate_fn(pc_shard, x_T_shard[, step_noise_shard]) -> (tmrp [n,6], logit [n,1]).  pcs [B,N,3] is the
    GLOBAL batch (same on every rank); returns the global (tmrp, logit).  `step_noise` [steps, B*G, 1, D] (DDPM) is
    global too and sliced like x_T; it is passed as a third argument only when given.  `check` (e.g. the denoise
    engine's `check`: raises if a step-segment hand-off of the fused launch was lost) runs before the result rows
    leave the rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = pcs.shape[0]
    lo, hi, per = shard_bounds(B, world, rank)
    if hi > lo:
        xs = None if x_T is None else shard_noise(x_T, num_grasps, lo, hi)
        if step_noise is None:
            tm, lg = generate_fn(pcs[lo:hi], xs)
        else:
            tm, lg = generate_fn(pcs[lo:hi], xs, shard_step_noise(step_noise, num_grasps, lo, hi))
        if check is not None:
            check()
        rows = torch.cat([tm, lg], dim=1)
    else:
        rows = torch.zeros((0, 7), dtype=torch.float32, device=pcs.device)
    out = gather_results(rows, per * num_grasps, B * num_grasps, group)
    return out[:, :6].contiguous(), out[:, 6:7].contiguous()
