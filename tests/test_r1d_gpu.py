"""GPU (MI355X): the fused 1-D ResNet engine (gldm_denoise / gldm_decode /
gldm_pose_epilogue through the C ABI) against the golden vectors captured from
the reference and against the torch-CPU oracle on fresh seeded inputs.
fp32 tolerances: single forward 2e-5 abs (values are O(1)); 100-step DDIM and
1000-step DDPM trajectories 1e-4 (the north-star pose bound)."""
import pytest
import torch

from conftest import load_golden, load_schema

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines(fpc_state_dict):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from graspldm_amd.r1d import R1dEngine, pack_resnet1d
    sd = fpc_state_dict
    den = R1dEngine(pack_resnet1d(sd, "diffusion_model.model.", groups=4, seq_len=4, num_steps=1000), "cuda:0")
    p = "vae_model.decoder."
    dec = R1dEngine(pack_resnet1d(sd, p + "net.", groups=4, seq_len=16, decoder=dict(
        in_w=sd[p + "in_layer.weight"], in_b=sd[p + "in_layer.bias"], tmrp_w=sd[p + "tmrp.weight"],
        tmrp_b=sd[p + "tmrp.bias"], cls_w=sd[p + "class_logits.weight"], cls_b=sd[p + "class_logits.bias"])), "cuda:0")
    return den, dec


def _err(a, b):
    return (a.cpu() - b).abs().max().item()


def test_cond_embed_matches_oracle(engines, fpc_state_dict):
    import torch.nn.functional as F
    den, _ = engines
    z = torch.randn(5, 3, 64, generator=torch.Generator().manual_seed(1))
    exp = F.silu(F.linear(z, fpc_state_dict["diffusion_model.model.input_emb_layers.0.weight"],
                          fpc_state_dict["diffusion_model.model.input_emb_layers.0.bias"]))
    assert _err(den.cond_embed(z.cuda()), exp) < 2e-6


def test_g3_denoiser_forward_golden(engines):
    den, _ = engines
    g = load_golden("denoiser.npz")
    cemb = den.cond_embed(g["z_cond"].cuda())
    for i, t in enumerate(g["t"].tolist()):
        ts = torch.tensor([t], dtype=torch.int32, device="cuda")
        eps = den.denoise(g["x"].cuda(), cemb, 1, timesteps=ts)
        assert _err(eps, g["eps"][i]) < 2e-5, (t, _err(eps, g["eps"][i]))


def test_denoiser_per_sample_times_and_ragged_batch(engines, fpc_state_dict):
    from oracle import torch_ref as R
    den, _ = engines
    g = torch.Generator().manual_seed(3)
    n = 37  # not a multiple of the 16-sample tile
    x = torch.randn(n, 1, 4, generator=g)
    z = torch.randn(n, 3, 64, generator=g)
    t = torch.randint(0, 1000, (n,), generator=g)
    exp = R.resnet1d_forward(fpc_state_dict, "diffusion_model.model.", x, z_cond=z, time=t)
    eps = den.denoise(x.cuda(), den.cond_embed(z.cuda()), 1, sample_t=t.int().cuda())
    assert _err(eps, exp) < 2e-5


def test_g4_decoder_golden(engines):
    _, dec = engines
    g = load_golden("decoder.npz")
    tmrp, logit = dec.decode(g["z_h"].cuda(), dec.cond_embed(g["z_cond"].cuda()), 1)
    assert _err(tmrp, g["tmrp"]) < 2e-5 and _err(logit, g["logit"]) < 2e-5


def _ddim_tables(n_inf=100):
    from graspldm_amd.diffusion import make_schedule_tables
    return make_schedule_tables("ddim", 1000, 5e-5, 1e-3, "linear", "fixed_large", n_inf)


def test_g5_ddim_trajectory_golden(engines):
    from graspldm_amd.r1d import SCHED_DDIM
    den, _ = engines
    g = load_golden("ddim_traj.npz")
    ts, coef = _ddim_tables(100)
    cemb = den.cond_embed(g["z_cond"].cuda())
    x_T = g["x_T"].cuda()
    for j, i in enumerate(g["probes"].tolist()):
        x = den.denoise(x_T, cemb, 1, timesteps=ts[:i].cuda(), sched_kind=SCHED_DDIM, coef=coef[:i].cuda())
        assert _err(x, g["x"][j]) < 1e-4, (i, _err(x, g["x"][j]))
    x0 = den.denoise(x_T, cemb, 1, timesteps=ts.cuda(), sched_kind=SCHED_DDIM, coef=coef.cuda())
    assert _err(x0, g["x0"]) < 1e-4


def test_g5_ddpm_trajectory_golden(engines):
    from graspldm_amd.diffusion import make_schedule_tables
    from graspldm_amd.r1d import SCHED_DDPM
    den, _ = engines
    g = load_golden("ddpm_traj.npz")
    ts, coef = make_schedule_tables("ddpm", 1000, 5e-5, 1e-3, "linear", "fixed_large", None)
    noise = torch.cat([g["step_noise"], torch.zeros(1, 4, 1, 4)]).cuda()  # t = 999..1 then unused t = 0
    cemb = den.cond_embed(g["z_cond"].cuda())
    x0 = den.denoise(g["x_T"].cuda(), cemb, 1, timesteps=ts.cuda(), sched_kind=SCHED_DDPM, coef=coef.cuda(),
                     step_noise=noise)
    assert _err(x0, g["x0"]) < 1e-4, _err(x0, g["x0"])


def test_g6_pose_epilogue_golden():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from graspldm_amd.r1d import pose_epilogue
    g = load_golden("tmrp_to_H.npz")
    mean, std = torch.zeros(1, 6).cuda(), torch.ones(1, 6).cuda()
    H, un, conf = pose_epilogue(g["tmrp"].cuda(), torch.zeros(64, 1).cuda(), mean, std, 64)
    assert _err(H, g["H"]) < 1e-6 and _err(un, g["tmrp"]) == 0 and _err(conf, torch.full((64, 1), 0.5)) == 0


def test_shared_conditioning_index(engines, fpc_state_dict):
    """samples_per_cond = G: sample i uses cloud i // G (repeat_interleave in grasp_ldm.py:204)."""
    from oracle import torch_ref as R
    den, _ = engines
    g = torch.Generator().manual_seed(5)
    z = torch.randn(3, 3, 64, generator=g)
    x = torch.randn(60, 1, 4, generator=g)
    t = torch.full((60,), 500)
    exp = R.resnet1d_forward(fpc_state_dict, "diffusion_model.model.", x, z_cond=z.repeat_interleave(20, 0), time=t)
    eps = den.denoise(x.cuda(), den.cond_embed(z.cuda()), 20, timesteps=torch.tensor([500], dtype=torch.int32).cuda())
    assert _err(eps, exp) < 2e-5


@pytest.mark.parametrize("n", [4096 + 16 * 64, 4096 + 52, 2 * 4096 + 4 * 256 + 3])
def test_tail_tiling_matches_single_tile_results(engines, n):
    """More than one wave of workgroups: the tail of the batch runs as small (one n-tile)
    workgroups.  Every sample must equal what it gets when run alone in a full tile."""
    from graspldm_amd.r1d import SCHED_DDIM
    den, _ = engines
    g = torch.Generator().manual_seed(n)
    z = torch.randn(7, 3, 64, generator=g).cuda()
    cemb = den.cond_embed(z)
    x = torch.randn(n, 1, 4, generator=g).cuda()
    ts, coef = _ddim_tables(100)
    ts, coef = ts[:3].cuda(), coef[:3].cuda()
    spc = (n + 6) // 7
    full = den.denoise(x, cemb, spc, timesteps=ts, sched_kind=SCHED_DDIM, coef=coef)
    probe = torch.tensor([0, 15, 4095, 4096, 4097, n - 53, n - 2, n - 1])
    for i in probe.tolist():
        one = den.denoise(x[i:i + 1], cemb[i // spc:i // spc + 1], 1, timesteps=ts, sched_kind=SCHED_DDIM, coef=coef)
        assert torch.equal(one[0], full[i]), i


# ---- batches that do not fill whole rounds of workgroups: the left-over tiles are split along the step axis
# over chains of workgroups (resnet1d.hip: make_plan).  Samples are independent, so whatever the work
# distribution, every sample's result must be BITWISE what the same sample gives in a small batch.
def _slots():
    return 2 * torch.cuda.get_device_properties(0).multi_processor_count  # two 8-latent tiles per CU


@pytest.mark.parametrize("extra_tiles,steps,kind", [(3, 10, "ddim"), (1, 7, "ddpm"), (130, 24, "ddim")])
def test_step_split_chain_is_bitwise_batch_invariant(engines, extra_tiles, steps, kind):
    from graspldm_amd.diffusion import make_schedule_tables
    from graspldm_amd.r1d import SCHED_DDIM, SCHED_DDPM
    den, _ = engines
    full = _slots() * 8
    n = full + extra_tiles * 8 - 3        # ragged: the last tile holds 5 samples
    spc = 4                               # samples per conditioning row
    g = torch.Generator().manual_seed(100 + extra_tiles)
    x = torch.randn(n, 1, 4, generator=g).cuda()
    z = torch.randn((n + spc - 1) // spc, 3, 64, generator=g).cuda()
    ts, coef = make_schedule_tables(kind, 1000, 5e-5, 1e-3, "linear", "fixed_large", 100)
    ts, coef = ts[-steps:].contiguous().cuda(), coef[-steps:].contiguous().cuda()   # the last `steps` steps (t -> 0)
    noise = torch.randn(steps, n, 1, 4, generator=g).cuda() if kind == "ddpm" else None
    sk = SCHED_DDIM if kind == "ddim" else SCHED_DDPM
    cemb = den.cond_embed(z)
    big = den.denoise(x, cemb, spc, timesteps=ts, sched_kind=sk, coef=coef, step_noise=noise)
    # reference distribution: chunks of at most one round (no chain), chunk starts aligned to conditioning rows
    parts, chunk = [], 2048
    for i0 in range(0, n, chunk):
        i1 = min(n, i0 + chunk)
        parts.append(den.denoise(x[i0:i1], cemb[i0 // spc:], spc, timesteps=ts, sched_kind=sk, coef=coef,
                                 step_noise=None if noise is None else noise[:, i0:i1].contiguous()))
    assert torch.equal(big, torch.cat(parts))
    assert torch.isfinite(big).all()
    # the workspace re-arms itself: a second launch on it gives the same bits, and no bounded wait expired
    again = den.denoise(x, cemb, spc, timesteps=ts, sched_kind=sk, coef=coef, step_noise=noise)
    assert torch.equal(big, again)
    assert den.workspace_errors() == 0


def test_lost_handoff_raises_instead_of_returning_garbage(engines):
    """A step-segment hand-off that never arrives (here: the workspace's arrival-ticket counter is pre-set to 1, so the
    slot that owns the left-over tile's first segment does not exist) must surface as GldmError, not as rc 0 with
    stale latents: the bounded wait sets the error word (and NaN-poisons the tile), R1dEngine.check() -- called by the
    inference harness before results leave the device -- raises, and the next launch on the workspace is clean."""
    from graspldm_amd._lib import GldmError
    from graspldm_amd.r1d import SCHED_DDIM
    den, _ = engines
    n = _slots() * 8 + 16                 # one left-over 16-latent tile (two 8-latent ones on the 32-column engine)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, 1, 4, generator=g).cuda()
    z = torch.randn(n // 8, 3, 64, generator=g).cuda()
    ts, coef = _ddim_tables(100)
    ts, coef = ts[-16:].contiguous().cuda(), coef[-16:].contiguous().cuda()
    cemb = den.cond_embed(z)
    run = lambda: den.denoise(x, cemb, 8, timesteps=ts, sched_kind=SCHED_DDIM, coef=coef)
    good = run()
    den.check()
    ws = den._ws[torch.cuda.current_stream().cuda_stream]
    ws[0:4].view(torch.int32).fill_(1)
    run()
    with pytest.raises(GldmError, match="hand-off"):
        den.check()
    again = run()                          # header re-armed by the kernel, error word cleared by check()
    den.check()
    assert torch.equal(again, good) and den.workspace_errors() == 0
    # the lazy path: without an explicit check(), the NEXT launch on that workspace raises
    ws[0:4].view(torch.int32).fill_(1)
    run()
    torch.cuda.synchronize()
    with pytest.raises(GldmError):
        run()
    assert torch.equal(run(), good)


def test_step_split_chain_against_oracle(engines, fpc_state_dict):
    """The spliced tile itself against the CPU oracle (not only against another HIP launch)."""
    from oracle import torch_ref as R
    from graspldm_amd.r1d import SCHED_DDIM
    den, _ = engines
    n = _slots() * 8 + 8
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n, 1, 4, generator=g)
    z = torch.randn(n // 8, 3, 64, generator=g)
    ts, coef = _ddim_tables(100)
    ts, coef = ts[-16:].contiguous(), coef[-16:].contiguous()
    out = den.denoise(x.cuda(), den.cond_embed(z.cuda()), 8, timesteps=ts.cuda(), sched_kind=SCHED_DDIM, coef=coef.cuda())
    sched = R.make_scheduler("ddim")
    sched.set_timesteps(100)
    sel = slice(n - 8, n)   # the left-over tile (8 hand-offs of 2 steps) ...
    own = slice(8, 16)      # ... and a tile whose run is interrupted by a spliced segment
    for sl in (sel, own):
        xs = x[sl].clone()
        zc = z[sl.start // 8].unsqueeze(0).expand(8, 3, 64)
        for t in ts.tolist():
            eps = R.resnet1d_forward(fpc_state_dict, "diffusion_model.model.", xs, z_cond=zc,
                                     time=torch.full((8,), t, dtype=torch.long))
            xs = sched.step(eps, t, xs).prev_sample
        assert _err(out[sl], xs) < 1e-4, _err(out[sl], xs)


def test_decoder_more_tiles_than_slots(engines):
    _, dec = engines
    n = _slots() * 2 * 2 + 7              # two rounds of 2-sample tiles + 3.5 tiles
    g = torch.Generator().manual_seed(21)
    zh = torch.randn(n, 4, generator=g).cuda()
    zc = torch.randn(n, 3, 64, generator=g).cuda()
    cemb = dec.cond_embed(zc)
    tm, lg = dec.decode(zh, cemb, 1)
    tm2, lg2 = dec.decode(zh[-40:].contiguous(), cemb[-40:].contiguous(), 1)
    assert torch.equal(tm[-40:], tm2) and torch.equal(lg[-40:], lg2)
    tm3, lg3 = dec.decode(zh[:64].contiguous(), cemb[:64].contiguous(), 1)
    assert torch.equal(tm[:64], tm3) and torch.equal(lg[:64], lg3)


def test_repeated_launches_are_bitwise_equal(engines):
    """The same launch three times: every bit equal.  The wave-local chains read accumulators inside hand-written DPP blocks;
    one of them (quad_narrow.h: pos_max8, until round 6) took them straight out of the matrix pipe, closer than the hazard
    allows -- a stale maximum now and then, which a softmax turns into last-bit noise that no tolerance test sees.  Both
    engines, tiles in several rounds per workgroup and a partly filled last tile."""
    from graspldm_amd.r1d import SCHED_DDIM
    den, dec = engines
    g = torch.Generator().manual_seed(77)
    n = _slots() * 16 + 16 * 40 + 5       # whole rounds, left-over tiles cut along the step axis, a short last tile
    x = torch.randn(n, 1, 4, generator=g).cuda()
    cemb = den.cond_embed(torch.randn(n, 3, 64, generator=g).cuda())
    ts, coef = _ddim_tables(100)
    ts, coef = ts[-12:].contiguous().cuda(), coef[-12:].contiguous().cuda()
    outs = [den.denoise(x, cemb, 1, timesteps=ts, sched_kind=SCHED_DDIM, coef=coef).clone() for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    m = _slots() * 4 * 2 + 7
    zh = torch.randn(m, 4, generator=g).cuda()
    dcemb = dec.cond_embed(torch.randn(m, 3, 64, generator=g).cuda())
    runs = [tuple(t.clone() for t in dec.decode(zh, dcemb, 1)) for _ in range(3)]
    for r in runs[1:]:
        assert torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1])


def test_decoder_shared_conditioning_rows(engines):
    """The decoder reads its ResnetBlocks' scale/shift rows from a per-cloud table (ss_table_kernel).  With 3 grasps
    per cloud every other 2-sample tile straddles two clouds: must equal, bit for bit, the same batch with the
    conditioning repeated per grasp (one table row set per sample)."""
    _, dec = engines
    g = torch.Generator().manual_seed(33)
    n_cond, spc = 11, 3
    zh = torch.randn(n_cond * spc, 4, generator=g).cuda()
    zc = torch.randn(n_cond, 3, 64, generator=g).cuda()
    tm, lg = dec.decode(zh, dec.cond_embed(zc), spc)
    tm2, lg2 = dec.decode(zh, dec.cond_embed(zc.repeat_interleave(spc, dim=0)), 1)
    assert torch.equal(tm, tm2) and torch.equal(lg, lg2)


def test_sample_major_fallback_engine_for_other_denoiser_shapes():
    """A 4-position denoiser whose widths are outside the position-major engine's set (16 / 32 channels here) runs on
    the sample-major 32-column engine (r1d_kernel<32, 4>), chains included; checked against the oracle."""
    from oracle import torch_ref as R
    from graspldm_amd.resnets import TimeConditionedResNet1D
    from graspldm_amd.r1d import SCHED_DDIM
    from graspldm_amd.synthetic import load_synthetic_weights
    net = TimeConditionedResNet1D(dim=4, channels=1, block_channels=(16, 32), input_conditioning_dims=64,
                                  resnet_block_groups=4, dropout=0.1, is_time_conditioned=True, learned_variance=False,
                                  learned_sinusoidal_cond=False, random_fourier_features=True)
    load_synthetic_weights(net, seed=3)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda().eval()
    g = torch.Generator().manual_seed(17)
    x = torch.randn(13, 1, 4, generator=g)
    z = torch.randn(13, 3, 64, generator=g)
    t = torch.randint(0, 1000, (13,), generator=g)
    eps = net(x.cuda(), time=t.cuda(), z_cond=z.cuda())
    exp = R.resnet1d_forward(sd, "", x, z_cond=z, time=t)
    assert _err(eps, exp) < 2e-5
    # more tiles than slots: the step-segment chains of the 32-column engine, bitwise vs a small batch
    eng = net.engine(torch.device("cuda:0"))
    n = _slots() * 8 + 19
    xb = torch.randn(n, 1, 4, generator=g).cuda()
    zb = torch.randn(n, 3, 64, generator=g).cuda()
    ts, coef = _ddim_tables(100)
    ts, coef = ts[-6:].contiguous().cuda(), coef[-6:].contiguous().cuda()
    cemb = eng.cond_embed(zb)
    big = eng.denoise(xb, cemb, 1, timesteps=ts, sched_kind=SCHED_DDIM, coef=coef)
    small = eng.denoise(xb[-40:].contiguous(), cemb[-40:].contiguous(), 1, timesteps=ts, sched_kind=SCHED_DDIM, coef=coef)
    assert torch.equal(big[-40:], small) and eng.workspace_errors() == 0


def test_folded_layernorm_with_offset_columns_and_fallback_without_the_folded_block(engines, fpc_state_dict):
    """The position-major engine takes the PreNorm LayerNorm inside the qkv conv (rstd (W' x - mean s), statistics
    from a two-pass sweep of the column).  A latent far from the origin (|x| up to 30: residual streams with a large
    common offset per column) must still agree with the oracle, and a descriptor without the folded block (qkvn_w = 0,
    what an ABI 3 packer leaves) must run on the sample-major engine with the same result."""
    from oracle import torch_ref as R
    from graspldm_amd.r1d import R1dEngine
    from graspldm_amd.r1d_pack import pack_resnet1d
    pre = "diffusion_model.model."
    g = torch.Generator().manual_seed(23)
    n = 37
    x = torch.randn(n, 1, 4, generator=g) * 10.0 + 5.0
    z = torch.randn(n, 3, 64, generator=g)
    t = torch.randint(0, 1000, (n,), generator=g)
    exp = R.resnet1d_forward(fpc_state_dict, pre, x, z_cond=z, time=t)
    dev = torch.device("cuda:0")
    outs = []
    for drop in (False, True):
        packed = pack_resnet1d(fpc_state_dict, pre, groups=4, seq_len=4, num_steps=1000)
        if drop:
            for lv in range(4):
                packed["desc"].lv[lv].qkvn_w = 0
        eng = R1dEngine(packed, dev)
        outs.append(eng.denoise(x.cuda(), eng.cond_embed(z.cuda()), 1, sample_t=t.int().cuda()))
    scale = max(1.0, exp.abs().max().item())
    assert _err(outs[0], exp) < 2e-5 * scale, _err(outs[0], exp)
    assert _err(outs[1], exp) < 2e-5 * scale, _err(outs[1], exp)


@pytest.mark.parametrize("block_channels", [(64, 128), (32, 256), (128,), (32, 64, 64)])
def test_position_major_engine_other_widths(block_channels):
    """The 64-column engine on other width sequences of its supported set (first level 4 channels, then 32..256):
    wider first down-conv (4 -> 64 / 128), a 32 -> 256 down-conv, a repeated width; against the oracle, one forward
    (2e-5) and a 12-step DDIM run (1e-4)."""
    from oracle import torch_ref as R
    from graspldm_amd.resnets import TimeConditionedResNet1D
    from graspldm_amd.r1d import SCHED_DDIM
    from graspldm_amd.synthetic import load_synthetic_weights
    net = TimeConditionedResNet1D(dim=4, channels=1, block_channels=block_channels, input_conditioning_dims=64,
                                  resnet_block_groups=4, dropout=0.1, is_time_conditioned=True, learned_variance=False,
                                  learned_sinusoidal_cond=False, random_fourier_features=True)
    load_synthetic_weights(net, seed=len(block_channels) + block_channels[0])
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda().eval()
    g = torch.Generator().manual_seed(5)
    n = 21   # one full 16-sample tile + a 5-sample one
    x = torch.randn(n, 1, 4, generator=g)
    z = torch.randn(n, 3, 64, generator=g)
    t = torch.randint(0, 1000, (n,), generator=g)
    eps = net(x.cuda(), time=t.cuda(), z_cond=z.cuda())
    exp = R.resnet1d_forward(sd, "", x, z_cond=z, time=t)
    assert _err(eps, exp) < 2e-5, _err(eps, exp)
    eng = net.engine(torch.device("cuda:0"))
    ts, coef = _ddim_tables(100)
    ts, coef = ts[-12:].contiguous(), coef[-12:].contiguous()
    out = eng.denoise(x.cuda(), eng.cond_embed(z.cuda()), 1, timesteps=ts.cuda(), sched_kind=SCHED_DDIM, coef=coef.cuda())
    sched = R.make_scheduler("ddim")
    sched.set_timesteps(100)
    xs = x.clone()
    for tt in ts.tolist():
        e = R.resnet1d_forward(sd, "", xs, z_cond=z, time=torch.full((n,), tt, dtype=torch.long))
        xs = sched.step(e, tt, xs).prev_sample
    assert _err(out, xs) < 1e-4, _err(out, xs)


def test_untamed_gains_short_horizon(engines):
    """The recipe weights tame two gains (raw-timestep column x 1e-3, final_conv x 0.1) so that 100 steps are
    contractive.  With O(1) gains instead, a single forward and a SHORT DDIM run (8 steps) of the HIP engine still follow
    the oracle: the error budget of the split-f16 / fused arithmetic does not rely on a benign network."""
    from oracle import torch_ref as R
    from graspldm_amd.r1d import R1dEngine, SCHED_DDIM
    from graspldm_amd.r1d_pack import pack_resnet1d
    from graspldm_amd.synthetic import synthetic_state_dict
    sd = synthetic_state_dict(load_schema("schema_fpc_ldm.json"), seed=0, tame=False)
    pre = "diffusion_model.model."
    eng = R1dEngine(pack_resnet1d(sd, pre, groups=4, seq_len=4, num_steps=1000), torch.device("cuda:0"))
    g = torch.Generator().manual_seed(12)
    n = 24
    x, z = torch.randn(n, 1, 4, generator=g), torch.randn(n, 3, 64, generator=g)
    t = torch.randint(0, 1000, (n,), generator=g)
    exp = R.resnet1d_forward(sd, pre, x, z_cond=z, time=t)
    eps = eng.denoise(x.cuda(), eng.cond_embed(z.cuda()), 1, sample_t=t.int().cuda())
    scale = max(1.0, exp.abs().max().item())
    assert _err(eps, exp) < 2e-5 * scale, (_err(eps, exp), scale)
    ts, coef = _ddim_tables(100)
    ts, coef = ts[-8:].contiguous(), coef[-8:].contiguous()
    out = eng.denoise(x.cuda(), eng.cond_embed(z.cuda()), 1, timesteps=ts.cuda(), sched_kind=SCHED_DDIM, coef=coef.cuda())
    sched = R.make_scheduler("ddim")
    sched.set_timesteps(100)
    xs = x.clone()
    for tt in ts.tolist():
        e = R.resnet1d_forward(sd, pre, xs, z_cond=z, time=torch.full((n,), tt, dtype=torch.long))
        xs = sched.step(e, tt, xs).prev_sample
    assert _err(out, xs) < 1e-4, _err(out, xs)


@pytest.mark.parametrize("block_channels", [(32,), (32, 64), (32, 64, 128, 256), (64, 128)])
def test_sixteen_position_engine_other_depths(block_channels):
    """The 16-position 64-column engine (r1d_kernel<64, 16>) on other depth / width sequences of its supported set (first
    level 16 channels, then 32..256), with per-sample timesteps and a batch that does not fill its last 4-sample tile:
    against the oracle, one forward (2e-5) and an 8-step DDIM run (1e-4); the same descriptor without the padded 16-channel
    split copies runs the sample-major f32 engine with the same result."""
    import ctypes
    from oracle import torch_ref as R
    from graspldm_amd import _lib as L
    from graspldm_amd.resnets import TimeConditionedResNet1D
    from graspldm_amd.r1d import SCHED_DDIM
    from graspldm_amd.synthetic import load_synthetic_weights
    net = TimeConditionedResNet1D(dim=16, channels=1, block_channels=block_channels, input_conditioning_dims=64,
                                  resnet_block_groups=4, dropout=0.1, is_time_conditioned=True, learned_variance=False,
                                  learned_sinusoidal_cond=False, random_fourier_features=True)
    load_synthetic_weights(net, seed=3 + len(block_channels))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda().eval()
    g = torch.Generator().manual_seed(5)
    n = 23   # five full tiles of 4 samples + one of 3
    x = torch.randn(n, 1, 16, generator=g)
    z = torch.randn(n, 3, 64, generator=g)
    t = torch.randint(0, 1000, (n,), generator=g)
    eng = net.engine(torch.device("cuda:0"))
    assert L.lib().gldm_r1d_tile_columns(eng._desc_ptr()) == 64
    exp = R.resnet1d_forward(sd, "", x, z_cond=z, time=t)
    eps = net(x.cuda(), time=t.cuda(), z_cond=z.cuda())
    assert _err(eps, exp) < 2e-5, _err(eps, exp)
    ts, coef = _ddim_tables(100)
    ts, coef = ts[-8:].contiguous(), coef[-8:].contiguous()
    out = eng.denoise(x.cuda(), eng.cond_embed(z.cuda()), 1, timesteps=ts.cuda(), sched_kind=SCHED_DDIM, coef=coef.cuda())
    sched = R.make_scheduler("ddim")
    sched.set_timesteps(100)
    xs = x.clone()
    for tt in ts.tolist():
        e = R.resnet1d_forward(sd, "", xs, z_cond=z, time=torch.full((n,), tt, dtype=torch.long))
        xs = sched.step(e, tt, xs).prev_sample
    assert _err(out, xs) < 1e-4, _err(out, xs)
    keep = eng.desc.rb[0].c1_w3
    eng.desc.rb[0].c1_w3 = 0   # no padded split copy of the 16-channel level -> the f32 engine
    assert L.lib().gldm_r1d_tile_columns(eng._desc_ptr()) == 32
    eps32 = eng.denoise(x.cuda(), eng.cond_embed(z.cuda()), 1, sample_t=t.int().cuda())
    eng.desc.rb[0].c1_w3 = keep
    assert _err(eps32, exp) < 2e-5, _err(eps32, exp)


@pytest.mark.parametrize("extra_tiles,steps,kind", [(3, 6, "ddim"), (40, 9, "ddpm")])
def test_step_split_chain_on_the_sixteen_position_engine(extra_tiles, steps, kind):
    """The same invariance on r1d_kernel<64, 16> (tiles of 4 samples, one workgroup per CU): a batch of one full round of
    tiles plus a few is cut along the step axis over chains of workgroups; every sample's result is BITWISE what it is
    in a launch without chains, the workspace (hand-off granules + park scratch) re-arms itself."""
    from graspldm_amd.diffusion import make_schedule_tables
    from graspldm_amd.r1d import SCHED_DDIM, SCHED_DDPM
    from graspldm_amd.resnets import TimeConditionedResNet1D
    from graspldm_amd.synthetic import load_synthetic_weights
    from graspldm_amd import _lib as L
    net = TimeConditionedResNet1D(dim=16, channels=1, block_channels=(32, 64, 128, 256), input_conditioning_dims=64,
                                  resnet_block_groups=4, dropout=0.1, is_time_conditioned=True, learned_variance=False,
                                  learned_sinusoidal_cond=False, random_fourier_features=True)
    load_synthetic_weights(net, seed=21)
    den = net.cuda().eval().engine(torch.device("cuda:0"))
    assert L.lib().gldm_r1d_tile_columns(den._desc_ptr()) == 64
    full = torch.cuda.get_device_properties(0).multi_processor_count * 4   # one 4-sample tile per CU
    n = full + extra_tiles * 4 - 1        # ragged: the last tile holds 3 samples
    spc = 4
    g = torch.Generator().manual_seed(200 + extra_tiles)
    x = torch.randn(n, 1, 16, generator=g).cuda()
    z = torch.randn((n + spc - 1) // spc, 3, 64, generator=g).cuda()
    ts, coef = make_schedule_tables(kind, 1000, 5e-5, 1e-3, "linear", "fixed_large", 100)
    ts, coef = ts[-steps:].contiguous().cuda(), coef[-steps:].contiguous().cuda()
    noise = torch.randn(steps, n, 1, 16, generator=g).cuda() if kind == "ddpm" else None
    sk = SCHED_DDIM if kind == "ddim" else SCHED_DDPM
    cemb = den.cond_embed(z)
    big = den.denoise(x, cemb, spc, timesteps=ts, sched_kind=sk, coef=coef, step_noise=noise)
    parts, chunk = [], 512     # half a round per launch: no chain
    for i0 in range(0, n, chunk):
        i1 = min(n, i0 + chunk)
        parts.append(den.denoise(x[i0:i1], cemb[i0 // spc:], spc, timesteps=ts, sched_kind=sk, coef=coef,
                                 step_noise=None if noise is None else noise[:, i0:i1].contiguous()))
    assert torch.equal(big, torch.cat(parts))
    assert torch.isfinite(big).all()
    again = den.denoise(x, cemb, spc, timesteps=ts, sched_kind=sk, coef=coef, step_noise=noise)
    assert torch.equal(big, again)
    assert den.workspace_errors() == 0


# ---- DDPM noise drawn inside the launch (gldm_denoise_rng)
def test_in_kernel_noise_generator_matches_the_restated_philox():
    import numpy as np
    from oracle.philox import step_noise
    from graspldm_amd.r1d import step_noise_rng
    for seed, base, step, n, L in [(0, 0, 0, 257, 4), (0xDEADBEEFCAFEF00D, (1 << 32) - 100, 999, 300, 16), (42, 12800 * 7, 17, 1000, 4)]:
        got = step_noise_rng(seed, base, step, n, L, "cuda:0").cpu().numpy()
        want = step_noise(seed, base, step, n, L)
        # same words, same Box-Muller; the device's fast log / sin / cos differ from numpy's in the last bits
        assert np.abs(got - want).max() < 2e-5, np.abs(got - want).max()
    z = step_noise_rng(7, 0, 5, 1 << 20, 4, "cuda:0").double()
    assert abs(z.mean().item()) < 3e-3 and abs(z.var().item() - 1) < 5e-3 and abs((z ** 4).mean().item() - 3) < 0.03


def test_in_kernel_noise_equals_the_same_normals_fed_from_memory(engines):
    """gldm_denoise_rng is gldm_denoise with step_noise[s] = gldm_step_noise_rng(seed, base, s): bit for bit, and
    independent of how the batch is cut when every part passes its first latent's global index."""
    from graspldm_amd.diffusion import make_schedule_tables
    from graspldm_amd.r1d import SCHED_DDPM, step_noise_rng
    den, _ = engines
    steps, n, base, seed = 12, 300, 1000, 0x1234567887654321
    ts, coef = make_schedule_tables("ddpm", 1000, 5e-5, 1e-3, "linear", "fixed_large", steps)
    ts, coef = ts.cuda(), coef.cuda()
    g = torch.Generator().manual_seed(5)
    x_T = torch.randn((n, 1, 4), generator=g).cuda()
    cemb = den.cond_embed(torch.randn((n, 3, 64), generator=g).cuda())
    with pytest.raises(RuntimeError):
        den.cond_embed(torch.randn((n, 3, 4)).cuda())   # not this network's conditioning width
    noise = torch.stack([step_noise_rng(seed, base, s, n, 4, "cuda:0") for s in range(steps)]).reshape(steps, n, 1, 4)
    want = den.denoise(x_T, cemb, 1, timesteps=ts, sched_kind=SCHED_DDPM, coef=coef, step_noise=noise)
    got = den.denoise_rng(x_T, cemb, 1, ts, coef, seed, noise_base=base)
    assert torch.equal(got, want)
    cut = 172
    a = den.denoise_rng(x_T[:cut], cemb[:cut], 1, ts, coef, seed, noise_base=base)
    b = den.denoise_rng(x_T[cut:], cemb[cut:], 1, ts, coef, seed, noise_base=base + cut)
    assert torch.equal(torch.cat([a, b]), want)
    other = den.denoise_rng(x_T, cemb, 1, ts, coef, seed + 1, noise_base=base)
    assert (other - want).abs().max().item() > 1e-3



@pytest.mark.parametrize("engine_columns", [64, 32])
def test_in_kernel_noise_on_the_sixteen_position_engines(engine_columns):
    """gldm_denoise_rng on a 16-position denoiser (the `ppc` experiment's shape class): four position blocks per latent, on
    the 64-column engine and on the sample-major f32 engine -- each bit for bit what the memory-fed launch gives with
    gldm_step_noise_rng's normals."""
    from graspldm_amd import _lib as L
    from graspldm_amd.diffusion import make_schedule_tables
    from graspldm_amd.resnets import TimeConditionedResNet1D
    from graspldm_amd.r1d import SCHED_DDPM, step_noise_rng
    from graspldm_amd.synthetic import load_synthetic_weights
    net = TimeConditionedResNet1D(dim=16, channels=1, block_channels=(32, 64), input_conditioning_dims=64,
                                  resnet_block_groups=4, dropout=0.1, is_time_conditioned=True, learned_variance=False,
                                  learned_sinusoidal_cond=False, random_fourier_features=True)
    load_synthetic_weights(net, seed=11)
    net = net.cuda().eval()
    eng = net.engine(torch.device("cuda:0"))
    if engine_columns == 32:
        eng.desc.rb[0].c1_w3 = 0   # no padded split copy of the 16-channel level -> the f32 engine
    assert L.lib().gldm_r1d_tile_columns(eng._desc_ptr()) == engine_columns
    steps, n, base, seed = 9, 37, (1 << 33) + 5, 99
    ts, coef = make_schedule_tables("ddpm", 1000, 5e-5, 1e-3, "linear", "fixed_large", steps)
    ts, coef = ts.cuda(), coef.cuda()
    g = torch.Generator().manual_seed(8)
    x_T = torch.randn((n, 1, 16), generator=g).cuda()
    cemb = eng.cond_embed(torch.randn((n, 3, 64), generator=g).cuda())
    noise = torch.stack([step_noise_rng(seed, base, s, n, 16, "cuda:0") for s in range(steps)]).reshape(steps, n, 1, 16)
    want = eng.denoise(x_T, cemb, 1, timesteps=ts, sched_kind=SCHED_DDPM, coef=coef, step_noise=noise)
    got = eng.denoise_rng(x_T, cemb, 1, ts, coef, seed, noise_base=base)
    assert torch.isfinite(got).all() and torch.equal(got, want)
