"""GPU (MI355X): every point operator of libgldm_hip.so, called through the
C ABI via the `_backend` shim, against the scalar C oracle on the same seeded
inputs.  Integer outputs must be bit-exact; float outputs are bit-exact too
(same f32 operation order, no FMA contraction) unless a test states a tolerance."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from graspldm_amd.backend import _backend
    return _backend


@pytest.fixture(scope="module")
def cpu():
    from oracle.cpu_backend import _backend
    return _backend


def _cloud(b, n, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(b, 3, n, generator=g) * 2 - 1) * scale).contiguous()


def _surface_cloud(b, n):
    from graspldm_amd.synthetic import synthetic_batch
    pcs, _ = synthetic_batch(b, n)
    return pcs.transpose(1, 2).contiguous()


@pytest.mark.parametrize("b,n,m", [(2, 1024, 512), (3, 512, 128), (1, 100, 100), (2, 64, 16), (1, 4096, 1024),
                                   (2, 777, 300), (1, 2048, 64), (1, 8192, 32)])
def test_fps_indices_exact(hip, cpu, b, n, m):
    pts = _surface_cloud(b, n) if n in (1024, 4096) else _cloud(b, n, n)
    got = hip.furthest_point_sampling(pts.cuda(), m).cpu()
    assert torch.equal(got, cpu.furthest_point_sampling(pts, m))


def test_fps_ties_duplicated_points(hip, cpu):
    base = torch.tensor([[0., 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [.5, .5, 1]]).T
    pts = base.repeat(1, 300)[:, :1400].unsqueeze(0).contiguous()
    got = hip.furthest_point_sampling(pts.cuda(), 12).cpu()
    assert torch.equal(got, cpu.furthest_point_sampling(pts, 12))


def _tie_cloud(kind, n):
    """Clouds whose FPS rounds end in exact maximal-distance ties (same f32 bit pattern at several indices).
    'dup': five corner points repeated round-robin -- every round's maximum is shared by n/5 copies, some of them 512
    apart (the reference kernel's slot rule, sampling.cu:137-160, prefers the lowest k mod 512, then the lowest k);
    'sym': the 8 vertices of a cube and the 6 of an octahedron around the origin, repeated -- distinct points at the
    same distance; 'grid': integer lattice points (exactly representable squared distances, many equal)."""
    if kind == "dup":
        base = torch.tensor([[0., 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [.5, .5, 1]])
    elif kind == "sym":
        cube = torch.tensor([[x, y, z] for x in (-1., 1.) for y in (-1., 1.) for z in (-1., 1.)])
        octa = torch.tensor([[2., 0, 0], [-2, 0, 0], [0, 2, 0], [0, -2, 0], [0, 0, 2], [0, 0, -2]])
        base = torch.cat([cube, octa])
    else:
        base = torch.tensor([[x, y, z] for x in range(5) for y in range(5) for z in range(5)], dtype=torch.float32)
    rep = -(-n // base.shape[0])
    return base.repeat(rep, 1)[:n].T.unsqueeze(0).contiguous()   # [1, 3, n]


@pytest.mark.parametrize("kind", ["dup", "sym", "grid"])
@pytest.mark.parametrize("n", [64, 100, 512, 1000, 1024])
def test_fps_wave_kernel_tie_rule(hip, cpu, kind, n):
    """Every shipped shape (n <= 1024) runs fps_wave_kernel, whose key is reduced as two 32-bit halves (distance bits,
    then the tie word among the lanes that hold the maximal distance): exact ties must pick what sampling.cu picks."""
    pts = _tie_cloud(kind, n)
    for m in sorted({min(n, 12), n // 2, n}):
        got = hip.furthest_point_sampling(pts.cuda(), m).cpu()
        assert torch.equal(got, cpu.furthest_point_sampling(pts, m)), (kind, n, m)
    two = torch.cat([pts, pts.flip(2)]).contiguous()   # batched: the second cloud ties at other indices
    assert torch.equal(hip.furthest_point_sampling(two.cuda(), min(n, 40)).cpu(), cpu.furthest_point_sampling(two, min(n, 40)))


@pytest.mark.parametrize("b,n,m,u,r", [(2, 1024, 512, 64, 0.2), (2, 512, 128, 64, 0.4), (1, 1024, 1024, 32, 0.1),
                                       (3, 100, 10, 8, 0.5), (1, 257, 33, 16, 0.3), (1, 6000, 40, 32, 0.15),
                                       (2, 64, 64, 4, 0.05), (1, 1024, 16, 128, 2.0)])
def test_ball_query_exact(hip, cpu, b, n, m, u, r):
    pts = _cloud(b, n, 10 + n)
    ctr = pts[:, :, torch.randperm(n, generator=torch.Generator().manual_seed(1))[:m]].contiguous()
    got = hip.ball_query(ctr.cuda(), pts.cuda(), r, u).cpu()
    assert torch.equal(got, cpu.ball_query(ctr, pts, r, u))


def test_ball_query_empty_and_graspldm_scale(hip, cpu):
    # GraspLDM clouds are scaled by 1/0.05 (~+-3): with r=0.1 most balls hold only the centre
    pts = _surface_cloud(2, 1024)
    ctr = torch.cat([pts[:, :, :60], torch.full((2, 3, 4), 50.0)], dim=2).contiguous()
    got = hip.ball_query(ctr.cuda(), pts.cuda(), 0.1, 32).cpu()
    exp = cpu.ball_query(ctr, pts, 0.1, 32)
    assert torch.equal(got, exp) and got[:, 60:].abs().sum() == 0


@pytest.mark.parametrize("b,c,n,m,u", [(2, 128, 512, 128, 64), (1, 3, 1024, 512, 64), (2, 7, 100, 13, 5),
                                       (1, 35, 1024, 1024, 32)])
def test_grouping_exact(hip, cpu, b, c, n, m, u):
    g = torch.Generator().manual_seed(3)
    f = torch.randn(b, c, n, generator=g)
    idx = torch.randint(0, n, (b, m, u), generator=g, dtype=torch.int32)
    assert torch.equal(hip.grouping_forward(f.cuda(), idx.cuda()).cpu(), cpu.grouping_forward(f, idx))
    i1 = idx[:, :, 0].contiguous()
    assert torch.equal(hip.gather_features_forward(f.cuda(), i1.cuda()).cpu(), cpu.gather_features_forward(f, i1))


@pytest.mark.parametrize("b,c,m,n", [(2, 256, 128, 512), (1, 1024, 1, 128), (2, 6, 17, 90), (1, 128, 512, 1024)])
def test_three_nn_interpolate_exact(hip, cpu, b, c, m, n):
    pts, ctr = _cloud(b, n, 4), _cloud(b, m, 5)
    feat = torch.randn(b, c, m, generator=torch.Generator().manual_seed(6))
    o, i, w = hip.three_nearest_neighbors_interpolate_forward(pts.cuda(), ctr.cuda(), feat.cuda())
    eo, ei, ew = cpu.three_nearest_neighbors_interpolate_forward(pts, ctr, feat)
    assert torch.equal(i.cpu(), ei)
    assert torch.equal(w.cpu(), ew)
    assert torch.equal(o.cpu(), eo)


@pytest.mark.parametrize("b,c,n,r", [(2, 3, 1024, 24), (2, 48, 1024, 12), (1, 5, 200, 4), (1, 16, 4096, 32),
                                     (2, 4, 64, 2), (1, 3, 100, 5), (1, 2, 300, 40), (3, 11, 1000, 16), (1, 7, 333, 8),
                                     (2, 20, 2048, 16)])
def test_avg_voxelize_exact_and_deterministic(hip, cpu, b, c, n, r):
    g = torch.Generator().manual_seed(8)
    feat = torch.randn(b, c, n, generator=g)
    vc = torch.randint(0, r, (b, 3, n), generator=g, dtype=torch.int32)
    if r == 2:
        vc[:] = 1  # every point in one voxel: longest serial segment
    o, i, k = hip.avg_voxelize_forward(feat.cuda(), vc.cuda(), r)
    eo, ei, ek = cpu.avg_voxelize_forward(feat, vc, r)
    assert torch.equal(i.cpu(), ei) and torch.equal(k.cpu(), ek)
    assert torch.equal(o.cpu(), eo)  # same ascending-index summation order -> bitwise
    # (grids whose rows fit LDS are assembled on chip and written dense -- n = 333: scalar feature staging, n = 2048: the
    # keys sorted through LDS; r = 5 (r^3 % 4 != 0) and r = 40 take the memset + scattered-store form)
    o2, _, _ = hip.avg_voxelize_forward(feat.cuda(), vc.cuda(), r)
    assert torch.equal(o2, o)


@pytest.mark.parametrize("b,c,n,r,train", [(2, 48, 1024, 24, False), (2, 96, 1024, 12, False), (1, 3, 77, 6, True)])
def test_trilinear_devoxelize_exact(hip, cpu, b, c, n, r, train):
    g = torch.Generator().manual_seed(9)
    grid = torch.randn(b, c, r ** 3, generator=g)
    coords = torch.rand(b, 3, n, generator=g) * (r - 1)
    coords[:, :, 0] = r - 1
    coords[:, :, 1] = 0
    o, i, w = hip.trilinear_devoxelize_forward(r, train, coords.cuda(), grid.cuda())
    eo, ei, ew = cpu.trilinear_devoxelize_forward(r, train, coords, grid)
    assert torch.equal(o.cpu(), eo)
    assert torch.equal(i.cpu(), ei) and torch.equal(w.cpu(), ew)


@pytest.mark.parametrize("normalize", [0, 1])
def test_voxel_coords_matches_torch_front_end(hip, normalize):
    """Voxelization.forward front end (modules/voxelization.py:16-35).  The mean is
    an f64 tree here vs torch's f32 cascade: norm_coords agree to 2 ulp at r scale
    (tolerance 1e-5), voxel indices may differ only where a coordinate sits within
    that distance of a .5 rounding boundary."""
    from graspldm_amd import _lib as L
    pts = _surface_cloud(3, 1024)
    r = 24
    d = pts.cuda()
    nc = torch.empty_like(d)
    vc = torch.empty(d.shape, dtype=torch.int32, device="cuda")
    L.call("gldm_voxel_coords", L.ptr(d), 3, 1024, r, normalize, 0.0, L.ptr(nc), L.ptr(vc), L.current_stream())
    ref = pts - pts.mean(2, keepdim=True)
    if normalize:
        ref = ref / (ref.norm(dim=1, keepdim=True).max(dim=2, keepdim=True).values * 2.0) + 0.5
    else:
        ref = (ref + 1) / 2.0
    ref = torch.clamp(ref * r, 0, r - 1)
    assert torch.allclose(nc.cpu(), ref, atol=1e-5)
    rv = torch.round(ref).to(torch.int32)
    diff = vc.cpu() != rv
    near = ((ref - ref.floor() - 0.5).abs() < 1e-4)
    assert not (diff & ~near).any()
    assert torch.equal(vc.cpu(), torch.round(nc.cpu()).to(torch.int32))


@pytest.mark.parametrize("b,c,n,m,u,r", [(2, 128, 512, 128, 64, 0.4), (2, 0, 1024, 512, 64, 0.2),
                                         (1, 32, 1024, 1024, 32, 0.1), (1, 5, 300, 37, 7, 0.5)])
def test_sa_group_equals_ballquery_module(hip, cpu, b, c, n, m, u, r):
    """gldm_sa_group == BallQuery.forward (modules/ball_query.py:16-34) built from the oracle ops."""
    from graspldm_amd import _lib as L
    pts = _cloud(b, n, 20)
    ctr = pts[:, :, :m].contiguous()
    feat = torch.randn(b, c, n, generator=torch.Generator().manual_seed(21)) if c else None
    idx = cpu.ball_query(ctr, pts, r, u)
    exp = cpu.grouping_forward(pts, idx) - ctr.unsqueeze(-1)
    if c:
        exp = torch.cat([exp, cpu.grouping_forward(feat, idx)], dim=1)
    out = torch.empty(b, 3 + c, m, u, device="cuda")
    io = torch.empty(b, m, u, dtype=torch.int32, device="cuda")
    dp, dc, df = pts.cuda(), ctr.cuda(), (feat.cuda() if c else None)
    L.call("gldm_sa_group", L.ptr(dp), L.ptr(dc), L.ptr(df), b, c, n, m, r, u, L.ptr(out), L.ptr(io), L.current_stream())
    assert torch.equal(io.cpu(), idx)
    assert torch.equal(out.cpu(), exp)


def test_backend_argument_checks_on_gpu(hip):
    x = torch.zeros(1, 3, 8, device="cuda")
    with pytest.raises(RuntimeError, match="contiguous"):
        hip.ball_query(x.transpose(1, 2).transpose(1, 2)[:, :, ::2], x, 0.1, 2)
    with pytest.raises(RuntimeError, match="float tensor"):
        hip.ball_query(x.double(), x, 0.1, 2)
    with pytest.raises(RuntimeError, match="int tensor"):
        hip.grouping_forward(x, torch.zeros(1, 2, 2, dtype=torch.int64, device="cuda"))
