"""CPU: the scalar C oracle of the point ops against an independent torch
formulation (self-pin: the reference has no CPU implementation of these)."""
import pytest
import torch

from oracle.cpu_backend import _backend as B


def _cloud(b, n, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(b, 3, n, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize("n,m,u,r", [(100, 10, 8, 0.5), (257, 33, 16, 0.3), (64, 64, 4, 0.05)])
def test_ball_query_semantics(n, m, u, r):
    pts = _cloud(2, n, 1)
    ctr = pts[:, :, :m].contiguous()
    idx = B.ball_query(ctr, pts, r, u)
    d2 = ((ctr.unsqueeze(-1) - pts.unsqueeze(-2)) ** 2)
    d2 = d2[:, 0] + d2[:, 1] + d2[:, 2]
    r2 = torch.tensor(r, dtype=torch.float32) ** 2
    for b in range(2):
        for j in range(m):
            hits = torch.nonzero(d2[b, j] < r2).flatten()[:u].tolist()
            exp = (hits + [hits[0]] * (u - len(hits))) if hits else [0] * u
            assert idx[b, j].tolist() == exp


def test_ball_query_empty_ball_is_all_zero():
    pts = _cloud(1, 50, 2)
    ctr = torch.full((1, 3, 4), 10.0)
    assert B.ball_query(ctr, pts, 0.1, 8).abs().sum() == 0


def test_fps_matches_naive_when_no_ties():
    pts = _cloud(2, 300, 3)
    idx = B.furthest_point_sampling(pts, 40)
    for b in range(2):
        p = pts[b].T
        dist = torch.full((300,), 1e38)
        cur, exp = 0, [0]
        for _ in range(39):
            d = ((p - p[cur]) ** 2)
            d = d[:, 0] + d[:, 1] + d[:, 2]
            dist = torch.minimum(dist, d)
            cur = int(torch.argmax(dist))
            exp.append(cur)
        assert idx[b].tolist() == exp


def test_fps_tie_rule_lower_slot_then_lower_index():
    # 4 corner points + duplicates: after picking index 0 the remaining corners tie
    base = torch.tensor([[0., 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]]).T
    pts = base.repeat(1, 200)[:, :600].unsqueeze(0).contiguous()  # n=600 > 512: slots wrap
    idx = B.furthest_point_sampling(pts, 4).tolist()[0]
    assert idx[0] == 0 and idx[1] == 3  # farthest corner, lowest slot/index among its copies
    assert len(set(i % 4 for i in idx)) == 4


def test_grouping_and_gather():
    f = torch.arange(2 * 5 * 20, dtype=torch.float32).view(2, 5, 20)
    idx = torch.randint(0, 20, (2, 7, 3), dtype=torch.int32)
    out = B.grouping_forward(f, idx)
    exp = torch.gather(f.unsqueeze(2).expand(2, 5, 7, 20), 3, idx.long().unsqueeze(1).expand(2, 5, 7, 3))
    assert torch.equal(out, exp)
    g = B.gather_features_forward(f, idx[:, :, 0].contiguous())
    assert torch.equal(g, exp[..., 0])


def test_three_nn_weights_and_indices():
    pts, ctr = _cloud(2, 90, 4), _cloud(2, 17, 5)
    feat = torch.randn(2, 6, 17)
    out, idx, w = B.three_nearest_neighbors_interpolate_forward(pts, ctr, feat)
    d = ((pts.unsqueeze(-1) - ctr.unsqueeze(-2)) ** 2)
    d = d[:, 0] + d[:, 1] + d[:, 2]
    top = torch.topk(d, 3, dim=-1, largest=False)
    assert torch.equal(idx.long().permute(0, 2, 1), top.indices)
    inv = 1.0 / top.values.clamp(1e-10, 1e10).double()
    wexp = (inv / inv.sum(-1, keepdim=True)).float().permute(0, 2, 1)
    assert torch.allclose(w, wexp, atol=1e-6)
    oexp = (torch.gather(feat.unsqueeze(2).expand(2, 6, 90, 17), 3, idx.long().permute(0, 2, 1).unsqueeze(1).expand(2, 6, 90, 3))
            * w.permute(0, 2, 1).unsqueeze(1)).sum(-1)
    assert torch.allclose(out, oexp, atol=1e-6)


def test_three_nn_fewer_than_three_centres():
    pts, ctr = _cloud(1, 10, 6), _cloud(1, 1, 7)
    out, idx, w = B.three_nearest_neighbors_interpolate_forward(pts, ctr, torch.ones(1, 2, 1))
    assert idx.abs().sum() == 0 and torch.allclose(w.sum(1), torch.ones(1, 10), atol=1e-6)


@pytest.mark.parametrize("r", [4, 12])
def test_avg_voxelize_is_scatter_mean(r):
    g = torch.Generator().manual_seed(8)
    n, c = 200, 5
    feat = torch.randn(2, c, n, generator=g)
    vc = torch.randint(0, r, (2, 3, n), generator=g, dtype=torch.int32)
    out, ind, cnt = B.avg_voxelize_forward(feat, vc, r)
    flat = (vc[:, 0] * r * r + vc[:, 1] * r + vc[:, 2]).long()
    assert torch.equal(ind.long(), flat)
    exp = torch.zeros(2, c, r ** 3, dtype=torch.float64)
    exp.scatter_add_(2, flat.unsqueeze(1).expand(2, c, n), feat.double())
    cexp = torch.zeros(2, r ** 3, dtype=torch.int64).scatter_add_(1, flat, torch.ones_like(flat))
    assert torch.equal(cnt.long(), cexp)
    exp = exp / cexp.clamp(min=1).unsqueeze(1)
    assert torch.allclose(out.double(), exp, atol=1e-6)


def test_trilinear_devoxelize_reproduces_a_linear_field():
    r, n = 6, 64
    g = torch.Generator().manual_seed(9)
    xs = torch.arange(r, dtype=torch.float32)
    grid = (2 * xs.view(r, 1, 1) + 3 * xs.view(1, r, 1) - xs.view(1, 1, r) + 1).reshape(1, 1, -1)
    coords = torch.rand(1, 3, n, generator=g) * (r - 1)
    coords[0, :, 0] = r - 1  # the clamp boundary: fractional part 0, no out-of-range corner
    out, _, _ = B.trilinear_devoxelize_forward(r, False, coords, grid.contiguous())
    exp = 2 * coords[:, 0] + 3 * coords[:, 1] - coords[:, 2] + 1
    assert torch.allclose(out[:, 0], exp, atol=1e-4)
    o2, inds, wgts = B.trilinear_devoxelize_forward(r, True, coords, grid.contiguous())
    assert torch.equal(o2, out) and torch.allclose(wgts.sum(1), torch.ones(1, n), atol=1e-6)
    assert int(inds.max()) < r ** 3 and int(inds.min()) >= 0
