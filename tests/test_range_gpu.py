"""GPU (MI355X): range of the split-f16 operands.  Every GEMM of the hot path multiplies two f16 pieces per f32 operand
(DESIGN.md §2); f16 has 5 exponent bits, so wherever the DATA sets an operand's magnitude the kernels split x / s with a
power-of-two s per tile (measured, or bounded from the layer's gain) and fold s back on the accumulators.  One test per
split entry point, with activations of 1e5 (hi = inf without the scale) and 1e-6 (hi an f16 subnormal with 4 significant
bits without it), against f64 / the oracle at the usual bars taken RELATIVE to the output's magnitude; ordinary magnitudes
are covered bit for bit by the existing tests (s = 1 there)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SCALES = [1e5, 1e-6, 3e7]


def _rel(got, ref):
    ref = ref.double()
    return ((got.detach().cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-300)).item()


@pytest.fixture(autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.mark.parametrize("scale", SCALES)
@pytest.mark.parametrize("cin,cout,n,hout", [(128, 256, 96, 0), (256, 512, 1024, 3), (128, 64, 48, 0)])
def test_pointwise_one_layer_input_range(scale, cin, cout, n, hout):
    """gldm_pointwise_mlp_f16x2: the input tile's scale is measured per tile; zero bias so that the 1e-6 case has nothing
    ordinary to hide behind (the output is then 1e-6-sized and the bar is relative to it)."""
    from graspldm_amd import dense
    from graspldm_amd.r1d_pack import mfma_a_fragments_f16x2
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(3, cin, n, generator=g) * scale
    x[1] *= 1e-3   # clouds of different magnitude in one launch: the scale is per tile
    w, b = torch.randn(cout, cin, generator=g) / cin ** 0.5, torch.zeros(cout)
    ref = torch.einsum("oc,bcn->bon", w.double(), x.double()).relu()
    head = None
    if hout:
        wh, bh = torch.randn(hout, cout, generator=g) / cout ** 0.5, torch.zeros(hout)
        zref = torch.einsum("oc,bcn->bon", wh.double(), ref)
        head = (dense.pack_head(wh).cuda(), bh.cuda(), hout)
    y, z = dense.pointwise_mlp(x.cuda(), mfma_a_fragments_f16x2(w).cuda(), b.cuda(), cout, True, head=head, split=True)
    assert torch.isfinite(y).all()
    for i in range(3):   # per cloud: each is held to its own magnitude
        assert _rel(y[i], ref[i]) < 2e-5, (i, _rel(y[i], ref[i]))
        if hout:
            assert _rel(z[i], zref[i]) < 2e-5, (i, _rel(z[i], zref[i]))


@pytest.mark.parametrize("scale", SCALES)
def test_pointwise_two_layers_range(scale):
    """gldm_pointwise_mlp2_f16x2 (the shipped encoder's 96 -> 768 -> 1536 -> head launch): input tile measured, the 768-row
    hidden tile scaled from the front layer's gain bound."""
    from graspldm_amd import dense
    from graspldm_amd.r1d_pack import mfma_a_fragments_f16x2
    g = torch.Generator().manual_seed(5)
    cin0, cin, cout, hout, n = 96, 768, 1536, 3, 1024
    x = torch.randn(2, cin0, n, generator=g) * scale
    w0, b0 = torch.randn(cin, cin0, generator=g) / cin0 ** 0.5, torch.randn(cin, generator=g) * 0.1
    w1, b1 = torch.randn(cout, cin, generator=g) / cin ** 0.5, torch.randn(cout, generator=g) * 0.1
    wh, bh = torch.randn(hout, cout, generator=g) / cout ** 0.5, torch.randn(hout, generator=g) * 0.1
    h = (torch.einsum("oc,bcn->bon", w0.double(), x.double()) + b0.double().view(1, -1, 1)).relu().float().double()
    yref = (torch.einsum("oc,bcn->bon", w1.double(), h) + b1.double().view(1, -1, 1)).relu()
    zref = torch.einsum("oc,bcn->bon", wh.double(), yref) + bh.double().view(1, -1, 1)
    front = (mfma_a_fragments_f16x2(w0).cuda(), b0.cuda(), cin, dense.range_gain(w0, b0))
    y, z = dense.pointwise_mlp(x.cuda(), mfma_a_fragments_f16x2(w1).cuda(), b1.cuda(), cout, True,
                               head=(dense.pack_head(wh).cuda(), bh.cuda(), hout), keep_y=True, front=front, split=True)
    assert torch.isfinite(y).all() and torch.isfinite(z).all()
    assert _rel(y, yref) < 2e-5 and _rel(z, zref) < 2e-5, (_rel(y, yref), _rel(z, zref))


@pytest.mark.parametrize("scale", SCALES)
@pytest.mark.parametrize("b,c,n,m,u,chans", [(2, 128, 512, 128, 64, (128, 128, 256)), (3, 0, 1024, 512, 64, (64, 64, 128)),
                                             (2, 32, 1024, 1024, 32, (32, 64)), (1, 5, 300, 37, 16, (16, 32, 32, 64))])
def test_sa_mlp_range(scale, b, c, n, m, u, chans):
    """gldm_sa_mlp_forward_f16x2 (single-tile and multi-tile forms): coordinates AND features scaled (user data when
    extra_feature_channels > 0), the ball radius with the coordinates so that the neighbourhoods are the same sets."""
    from graspldm_amd import sa_pack
    from graspldm_amd.pvcnn import PointNetSAModule
    from graspldm_amd.synthetic import load_synthetic_weights
    from oracle import torch_ref as R
    assert sa_pack.split_plan_ok([c + 3] + list(chans[:-1]), list(chans), u)
    rad = 0.35 * scale
    mod = PointNetSAModule(num_centers=m, radius=rad, num_neighbors=u, in_channels=c, out_channels=chans).eval()
    load_synthetic_weights(mod, seed=11)
    g = torch.Generator().manual_seed(2)
    coords = ((torch.rand(b, 3, n, generator=g) * 2 - 1) * scale).contiguous()
    feats = torch.randn(b, c, n, generator=g) * scale if c else None
    sd = {k: v.detach() for k, v in mod.state_dict().items()}
    exp, ectr = R.sa_module(sd, "", feats, coords, m, [rad], [u])
    mod = mod.cuda()
    with torch.no_grad():
        got, ctr = mod((feats.cuda() if c else None, coords.cuda()))
    assert torch.equal(ctr.cpu(), ectr)
    assert torch.isfinite(got).all()
    assert _rel(got, exp) < 2e-5, _rel(got, exp)


@pytest.mark.parametrize("scale", SCALES)
@pytest.mark.parametrize("cin,cout,r", [(3, 48, 24), (48, 48, 24), (48, 96, 12), (96, 96, 12), (64, 64, 32), (128, 128, 4)])
def test_conv3d_split_range(scale, cin, cout, r):
    """gldm_conv3d_k3_f16x2 on a grid of the data's magnitude (voxel averages of raw features): per-brick scale on the
    3-channel kernel, running per-block scale (accumulators rescaled when it grows) on the plane-staging one -- the second
    half of the channels is another 1e3 larger so that the scale does grow inside a brick."""
    import torch.nn.functional as F
    from graspldm_amd import _lib as L
    from graspldm_amd.voxel import pack_conv3d_f16x2, pack_conv3d_fewch_f16x2
    g = torch.Generator().manual_seed(cin * 100 + cout + 1)
    b = 2
    x = torch.randn(b, cin, r, r, r, generator=g) * scale
    x[:, :, ::3] = 0
    if cin >= 32:
        x[:, cin // 2:] *= 1e3
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    ref = F.conv3d(x.double(), w.double(), bias.double(), padding=1)
    y = torch.empty(b, cout, r, r, r, device="cuda")
    part = torch.empty(int(L.lib().gldm_conv3d_partial_floats(b, cout, r)), device="cuda")
    dw = (pack_conv3d_fewch_f16x2(w) if cin < 16 else pack_conv3d_f16x2(w)).cuda()
    dx, db = x.cuda(), bias.cuda()
    L.call("gldm_conv3d_k3_f16x2", L.ptr(dx), L.ptr(dw), L.ptr(db), b, cin, cout, r, L.ptr(y), L.ptr(part), L.current_stream())
    assert torch.isfinite(y).all()
    assert _rel(y, ref) < 2e-5, _rel(y, ref)
    # the GroupNorm partials of the same launch: (sum, sum of squares) per brick -> the normalised tensor
    gamma, beta = torch.ones(cout), torch.zeros(cout)
    gn = F.group_norm(ref.float(), 8, gamma, beta, 1e-5)
    ref2 = gn * torch.sigmoid(gn)
    if scale < 1e6:   # sums of squares of 3e7-sized values times 1e3 leave f32: the reference's own GroupNorm does too
        dg, dbt = gamma.cuda(), beta.cuda()   # locals: the pointers must outlive the call
        L.call("gldm_groupnorm_swish", L.ptr(y), L.ptr(part), L.ptr(dg), L.ptr(dbt), b, cout, r, 8, 1e-5, None, L.current_stream())
        assert (y.cpu() - ref2).abs().max() < 5e-5


@pytest.mark.parametrize("scale", [1e3, 1e-3])
def test_encoder_on_a_cloud_in_other_units(scale, fpc_state_dict):
    """PVCNNEncoder.forward on a cloud 1000 x larger (millimetres for metres) / smaller than the golden one: the reference
    stays finite there (f32 throughout), and so must the split path -- against the torch-CPU oracle on the same cloud."""
    from graspldm_amd.synthetic import synthetic_batch
    from oracle import torch_ref as R
    from test_modules_cpu import build_fpc
    m = build_fpc(scheduler="ddim")
    m.load_state_dict(fpc_state_dict, strict=True)
    pcs, _ = synthetic_batch(2, 1024)
    pcs = pcs * scale
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    exp = R.pvcnn_encoder_forward(sd, "vae_model.encoder.pc_encoder.", pcs, R.pvcnn_block_spec(0.75, 0.75))
    got = m.cuda().vae_model.encode_pc(pcs.cuda())
    assert torch.isfinite(got).all() and torch.isfinite(exp).all()
    assert _rel(got, exp) < 5e-5, _rel(got, exp)


def test_non_finite_poses_raise(fpc_state_dict):
    """_InferenceBase._results refuses to hand out poses that are not numbers (a NaN cloud here)."""
    from graspldm_amd._lib import GldmError
    from graspldm_amd.inference import InferenceLDM
    from graspldm_amd.synthetic import synthetic_batch
    from test_modules_cpu import build_fpc
    m = build_fpc(scheduler="ddim")
    m.load_state_dict(fpc_state_dict, strict=True)
    inf = InferenceLDM(model=m.cuda().eval(), num_inference_steps=3, device="cuda:0")
    pcs, _ = synthetic_batch(1, 1024)
    pcn, metas = inf.normalize_input(pcs[0].cuda() * 0.05)
    ok = inf.generate_grasps(pcn, metas, num_grasps=2)
    assert torch.isfinite(ok["grasps"]).all()
    bad = pcn.clone()
    bad[5, 1] = float("nan")
    with pytest.raises(GldmError, match="non-finite coordinates"):
        inf.generate_grasps(bad, metas, num_grasps=2)
    bad_metas = dict(metas)
    bad_metas["grasp_mean"] = metas["grasp_mean"] * float("inf")
    with pytest.raises(GldmError, match="not finite"):
        inf.generate_grasps(pcn, bad_metas, num_grasps=2)


@pytest.mark.parametrize("scale", [1e3, 3e4])
def test_resnet_engines_under_wild_conditioning(scale, fpc_state_dict):
    """The ResnetBlocks' H = act((scale + 1) GN(conv1) + shift) is the one operand of the fused ResNet1D engines whose size the
    DATA sets (through the conditioning embedding): with conditioning rows 1e3 .. 3e4 times their usual size the scale / shift
    rows reach 1e5 and H leaves the f16 range, so block1 writes H / hs with a power-of-two hs per sample and block2 folds it
    back (conv_pm3_wave, quad_narrow.h, quad16_narrow.h).  All three 64-column paths -- the shipped 4-position denoiser, the
    pose decoder (scale / shift rows from the per-cloud table) and a 16-position time-conditioned net (the `ppc` denoiser's
    shape: rows computed in front of the wave-local chain) -- against the f64 oracle: as close as torch's own f32 (5e-7; bar 5e-6)."""
    from oracle import torch_ref as R
    from graspldm_amd.r1d import R1dEngine, pack_resnet1d
    from graspldm_amd.resnets import TimeConditionedResNet1D
    from graspldm_amd.synthetic import load_synthetic_weights
    sd = fpc_state_dict
    g = torch.Generator().manual_seed(5)
    n = 11
    z = torch.randn(n, 3, 64, generator=g) * scale

    def err(a, b):
        return (a.cpu().double() - b).abs().max().item()

    # the shipped 4-position denoiser
    p = "diffusion_model.model."
    den = R1dEngine(pack_resnet1d(sd, p, groups=4, seq_len=4, num_steps=1000), "cuda:0")
    x4 = torch.randn(n, 1, 4, generator=g)
    t4 = torch.randint(0, 1000, (n,), generator=g)
    sub64 = {k[len(p):]: v.double() for k, v in sd.items() if k.startswith(p)}
    exp = R.resnet1d_forward(sub64, "", x4.double(), z_cond=z.double(), time=t4)
    got = den.denoise(x4.cuda(), den.cond_embed(z.cuda()), 1, sample_t=t4.int().cuda())
    assert torch.isfinite(got).all() and err(got, exp) < 5e-6, err(got, exp)
    # the pose decoder
    pd = "vae_model.decoder."
    dec = R1dEngine(pack_resnet1d(sd, pd + "net.", groups=4, seq_len=16, decoder=dict(
        in_w=sd[pd + "in_layer.weight"], in_b=sd[pd + "in_layer.bias"], tmrp_w=sd[pd + "tmrp.weight"],
        tmrp_b=sd[pd + "tmrp.bias"], cls_w=sd[pd + "class_logits.weight"], cls_b=sd[pd + "class_logits.bias"])), "cuda:0")
    zh = torch.randn(n, 4, generator=g)
    tm64, lg64 = R.decoder_forward({k: v.double() for k, v in sd.items()}, pd, zh.double(), z.double())
    tm, lg = dec.decode(zh.cuda(), dec.cond_embed(z.cuda()), 1)
    assert torch.isfinite(tm).all() and err(tm, tm64) < 5e-6 and err(lg, lg64.reshape(lg.shape)) < 5e-6, (err(tm, tm64), err(lg, lg64.reshape(lg.shape)))
    # a 16-position time-conditioned net
    net = TimeConditionedResNet1D(dim=16, channels=1, block_channels=(32, 64, 128, 256), input_conditioning_dims=64,
                                  resnet_block_groups=4, dropout=0.1, is_time_conditioned=True, learned_variance=False,
                                  learned_sinusoidal_cond=False, random_fourier_features=True)
    load_synthetic_weights(net, seed=7)
    sd16 = {k: v.detach().clone().double() for k, v in net.state_dict().items()}
    net = net.cuda().eval()
    x = torch.randn(n, 1, 16, generator=g)
    t = torch.randint(0, 1000, (n,), generator=g)
    exp = R.resnet1d_forward(sd16, "", x.double(), z_cond=z.double(), time=t)
    got = net(x.cuda(), time=t.cuda(), z_cond=z.cuda())
    assert torch.isfinite(got).all() and err(got, exp) < 5e-6, err(got, exp)
