"""Point-cloud backbones of the GraspLDM encoder on MI355X: host-side mirror of
`grasp_ldm/models/modules/ext/pvcnn/**` (functional wrappers, nn.Modules and the
three backbones PVCNN / PVCNN2 / PointNet2SSG) with the reference's class names,
constructor arguments and state_dict keys, so reference checkpoints load with
`strict=True`.

Inference only.  Every point operator runs in libgldm_hip.so through the C ABI
(`backend._backend` / `_lib.call`); there is no CPU or autograd path: calling a
module on a CPU tensor raises like the reference's CHECK_CUDA does.
"""
import functools

import torch
import torch.nn as nn

from . import _lib as L
from . import dense
from .backend import _backend

# ------------------------------------------------------------------ functional
# (functional/{ball_query,grouping,sampling,interpolatation,voxelization,devoxelization}.py)


def ball_query(centers_coords, points_coords, radius, num_neighbors):
    return _backend.ball_query(centers_coords.contiguous(), points_coords.contiguous(), radius, num_neighbors)


def grouping(features, indices):
    return _backend.grouping_forward(features.contiguous(), indices.contiguous())


def gather(features, indices):
    return _backend.gather_features_forward(features.contiguous(), indices.int().contiguous())


def furthest_point_sample(coords, num_samples):
    coords = coords.contiguous()
    return gather(coords, _backend.furthest_point_sampling(coords, num_samples))


def nearest_neighbor_interpolate(points_coords, centers_coords, centers_features):
    return _backend.three_nearest_neighbors_interpolate_forward(
        points_coords.contiguous(), centers_coords.contiguous(), centers_features.contiguous())[0]


def avg_voxelize(features, coords, resolution):
    b, c, _ = features.shape
    out, _, _ = _backend.avg_voxelize_forward(features.contiguous(), coords.int().contiguous(), resolution)
    return out.view(b, c, resolution, resolution, resolution)


def trilinear_devoxelize(features, coords, resolution, is_training=False):
    b, c = features.shape[:2]
    return _backend.trilinear_devoxelize_forward(resolution, bool(is_training), coords.contiguous(),
                                                 features.contiguous().view(b, c, -1))[0]


def sa_group(points_coords, centers_coords, points_features, radius, num_neighbors):
    """BallQuery.forward with include_coordinates=True as one launch (gldm_sa_group)."""
    pts, ctr = points_coords.contiguous(), centers_coords.contiguous()
    feat = points_features.contiguous() if points_features is not None and points_features.shape[1] > 0 else None
    for t, nm in ((pts, "points_coords"), (ctr, "centers_coords")) + (((feat, "points_features"),) if feat is not None else ()):
        if not t.is_cuda:
            raise RuntimeError(f"{nm} must be a CUDA tensor")
        if t.dtype != torch.float32:
            raise RuntimeError(f"{nm} must be a float tensor")
    b, _, n = pts.shape
    m = ctr.shape[2]
    c = 0 if feat is None else feat.shape[1]
    out = torch.empty((b, 3 + c, m, int(num_neighbors)), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        L.call("gldm_sa_group", L.ptr(pts), L.ptr(ctr), L.ptr(feat), b, c, n, m, float(radius), int(num_neighbors),
               L.ptr(out), None, L.current_stream(pts.device))
    return out


# --------------------------------------------------------------------- modules


class Swish(nn.Module):
    def forward(self, x):
        return dense.swish(x)


class SharedMLP(nn.Module):
    """(Conv k=1 + BatchNorm + ReLU)*  -- shared_mlp.py:6-35.  Eval mode: BatchNorm is
    folded into the GEMM epilogue (dense.pointwise_conv_bn_relu)."""

    def __init__(self, in_channels, out_channels, dim=1):
        super().__init__()
        if dim == 1:
            conv, bn = nn.Conv1d, nn.BatchNorm1d
        elif dim == 2:
            conv, bn = nn.Conv2d, nn.BatchNorm2d
        else:
            raise ValueError
        if not isinstance(out_channels, (list, tuple)):
            out_channels = [out_channels]
        layers = []
        for oc in out_channels:
            layers.extend([conv(in_channels, oc, 1), bn(oc), nn.ReLU(True)])
            in_channels = oc
        self.layers = nn.Sequential(*layers)

    def _run(self, x):
        for i in range(0, len(self.layers), 3):
            x = dense.pointwise_conv_bn_relu(x, self.layers[i], self.layers[i + 1])
        return x

    def forward(self, inputs):
        if isinstance(inputs, (list, tuple)):
            return (self._run(inputs[0]), *inputs[1:])
        return self._run(inputs)


def _first_layer_of_concat(shared_mlp, parts, training):
    """First layer of a SharedMLP(dim=1) over cat(parts, dim=1) without the concatenation (dense.concat_conv_bn_relu), or
    None when the shapes do not fit its two forms."""
    if training or len(shared_mlp.layers) < 3 or not isinstance(shared_mlp.layers[0], nn.Conv1d):
        return None
    return dense.concat_conv_bn_relu(parts[0].float(), parts[1].float(), shared_mlp.layers[0], shared_mlp.layers[1])


def _rest_of(shared_mlp, x):
    for i in range(3, len(shared_mlp.layers), 3):
        x = dense.pointwise_conv_bn_relu(x, shared_mlp.layers[i], shared_mlp.layers[i + 1])
    return x


class BallQuery(nn.Module):
    """ball_query.py:9-34"""

    def __init__(self, radius, num_neighbors, include_coordinates=True):
        super().__init__()
        self.radius, self.num_neighbors, self.include_coordinates = radius, num_neighbors, include_coordinates

    def forward(self, points_coords, centers_coords, points_features=None):
        if self.include_coordinates:
            return sa_group(points_coords, centers_coords, points_features, self.radius, self.num_neighbors)
        assert points_features is not None, "No Features For Grouping"
        idx = ball_query(centers_coords, points_coords, self.radius, self.num_neighbors)
        return grouping(points_features, idx)

    def extra_repr(self):
        return "radius={}, num_neighbors={}{}".format(
            self.radius, self.num_neighbors, ", include coordinates" if self.include_coordinates else "")


def _as_nested(out_channels, n):
    if not isinstance(out_channels, (list, tuple)):
        return [[out_channels]] * n
    if not isinstance(out_channels[0], (list, tuple)):
        return [out_channels] * n
    return out_channels


class PointNetAModule(nn.Module):
    """pointnet.py:11-46"""

    def __init__(self, in_channels, out_channels, include_coordinates=True):
        super().__init__()
        out_channels = _as_nested(out_channels, 1)
        mlps, total = [], 0
        for oc in out_channels:
            mlps.append(SharedMLP(in_channels + (3 if include_coordinates else 0), oc, dim=1))
            total += oc[-1]
        self.include_coordinates = include_coordinates
        self.out_channels = total
        self.mlps = nn.ModuleList(mlps)

    def forward(self, inputs):
        features, coords = inputs
        parts = (features, coords) if self.include_coordinates else None
        if self.include_coordinates:
            features = None   # concatenated below only by the branches that need the tensor
        new_coords = torch.zeros((coords.size(0), 3, 1), device=coords.device)
        outs = []
        for mlp in self.mlps:
            x = _first_layer_of_concat(mlp, parts, self.training) if parts is not None else None
            if x is None:
                if features is None:
                    features = torch.cat(parts, dim=1)
                x = mlp(features)
            else:
                x = _rest_of(mlp, x)
            outs.append(dense.row_max(x))
        return (torch.cat(outs, dim=1) if len(outs) > 1 else outs[0]), new_coords


class PointNetSAModule(nn.Module):
    """pointnet.py:49-114: FPS -> ball query -> group -> SharedMLP2d -> max over neighbours."""

    def __init__(self, num_centers, radius, num_neighbors, in_channels, out_channels, include_coordinates=True):
        super().__init__()
        if not isinstance(radius, (list, tuple)):
            radius = [radius]
        if not isinstance(num_neighbors, (list, tuple)):
            num_neighbors = [num_neighbors] * len(radius)
        assert len(radius) == len(num_neighbors)
        out_channels = _as_nested(out_channels, len(radius))
        assert len(radius) == len(out_channels)
        groupers, mlps, total = [], [], 0
        for r, oc, k in zip(radius, out_channels, num_neighbors):
            groupers.append(BallQuery(radius=r, num_neighbors=k, include_coordinates=include_coordinates))
            mlps.append(SharedMLP(in_channels + (3 if include_coordinates else 0), oc, dim=2))
            total += oc[-1]
        self.num_centers = num_centers
        self.out_channels = total
        self.groupers = nn.ModuleList(groupers)
        self.mlps = nn.ModuleList(mlps)

    def _fused(self, g, grouper, mlp, coords, centers, features):
        """ball query -> ONE fused launch (gather + grouped MLP on MFMA + max): the grouped
        [B, 3+C, M, U] tensor is never materialised."""
        from .sa_pack import SaMlpPlan
        plans = self.__dict__.setdefault("_plans", {})
        from ._cache import params_key, publish
        key = params_key(mlp.state_dict(keep_vars=True).values(), coords.device)
        plan = plans.get(g)
        if plan is None or plan.key_dev != key:
            plan = SaMlpPlan(mlp, coords.device)
            plan.key_dev = key
            plans[g] = plan
            publish(coords.device)
        feats = features.contiguous() if features is not None and features.shape[1] > 0 else None
        idx = ball_query(centers, coords, grouper.radius, grouper.num_neighbors)
        return plan.run(coords.contiguous(), centers.contiguous(), feats, idx)

    def forward(self, inputs):
        from .sa_pack import fusable
        features, coords = inputs
        centers = furthest_point_sample(coords, self.num_centers)
        outs = []
        for g, (grouper, mlp) in enumerate(zip(self.groupers, self.mlps)):
            if grouper.include_coordinates and not self.training and fusable(mlp, grouper.num_neighbors):
                outs.append(self._fused(g, grouper, mlp, coords, centers, features))
            else:  # wide (> 256 channel) or U > 64 stages: gather kernel + GEMMs + max
                outs.append(mlp(grouper(coords, centers, features)).max(dim=-1).values)
        return (torch.cat(outs, dim=1) if len(outs) > 1 else outs[0]), centers

    def extra_repr(self):
        return f"num_centers={self.num_centers}, out_channels={self.out_channels}"


class PointNetFPModule(nn.Module):
    """pointnet.py:117-135"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.mlp = SharedMLP(in_channels=in_channels, out_channels=out_channels, dim=1)

    def forward(self, inputs):
        if len(inputs) == 3:
            points_coords, centers_coords, centers_features = inputs
            points_features = None
        else:
            points_coords, centers_coords, centers_features, points_features = inputs
        if points_features is not None and centers_coords.shape[2] == 1:
            # ONE centre: its feature vector is what every point would interpolate to -- no interpolation, no concatenation:
            # the first layer takes it as a per-cloud bias (dense.concat_conv_bn_relu)
            x = _first_layer_of_concat(self.mlp, (centers_features, points_features), self.training)
            if x is not None:
                return _rest_of(self.mlp, x), points_coords
        interp = nearest_neighbor_interpolate(points_coords, centers_coords, centers_features)
        if points_features is not None:
            x = _first_layer_of_concat(self.mlp, (interp, points_features), self.training)
            if x is not None:
                return _rest_of(self.mlp, x), points_coords
            interp = torch.cat([interp, points_features], dim=1)
        return self.mlp(interp), points_coords


class Voxelization(nn.Module):
    """voxelization.py:9-35; the coordinate front end is one kernel (gldm_voxel_coords)."""

    def __init__(self, resolution, normalize=True, eps=0):
        super().__init__()
        self.r, self.normalize, self.eps = int(resolution), normalize, eps

    def forward(self, features, coords):
        coords = coords.detach().contiguous()
        if not coords.is_cuda:
            raise RuntimeError("coords must be a CUDA tensor")
        b, _, n = coords.shape
        norm_coords = torch.empty_like(coords)
        vox = torch.empty(coords.shape, dtype=torch.int32, device=coords.device)
        with torch.cuda.device(coords.device):
            L.call("gldm_voxel_coords", L.ptr(coords), b, n, self.r, 1 if self.normalize else 0, float(self.eps),
                   L.ptr(norm_coords), L.ptr(vox), L.current_stream(coords.device))
        return avg_voxelize(features, vox, self.r), norm_coords

    def extra_repr(self):
        return "resolution={}{}".format(self.r, ", normalized eps = {}".format(self.eps) if self.normalize else "")


class SE3d(nn.Module):
    """se.py:12-25"""

    def __init__(self, channel, reduction=8, use_relu=False):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction, bias=False),
                                nn.ReLU(True) if use_relu else Swish(),
                                nn.Linear(channel // reduction, channel, bias=False), nn.Sigmoid())
        self.use_relu = use_relu

    def gate(self, chan_sum, r):
        """sigmoid(W2 act(W1 mean)) from the per-channel SUMS over the r^3 voxels [B, C]: one launch (gldm_se_gate)."""
        if not chan_sum.is_cuda:
            raise RuntimeError("SE3d runs on the GPU only (graspldm_amd has no CPU path)")
        w1, w2 = self.fc[0].weight, self.fc[2].weight
        b, c = chan_sum.shape
        gate = torch.empty((b, c), dtype=torch.float32, device=chan_sum.device)
        cs = chan_sum.contiguous().float()   # referenced by a local until the call returns, like every launch operand
        with torch.cuda.device(chan_sum.device):
            L.call("gldm_se_gate", L.ptr(cs), L.ptr(w1), L.ptr(w2), b, c, w1.shape[0], int(r),
                   1 if self.use_relu else 0, L.ptr(gate), L.current_stream(chan_sum.device))
        return gate

    def forward(self, inputs):
        """se.py:12-25 on [B, C, r, r, r] (inside PVConv the gate is folded into the devoxelize pass instead)."""
        g = self.gate(inputs.sum(dim=(2, 3, 4)), inputs.shape[-1])
        return inputs * g.view(inputs.shape[0], inputs.shape[1], 1, 1, 1)


class PVConv(nn.Module):
    """pvconv.py:13-84 (inference; `use_attention` voxel attention is off in every shipped config)."""

    def __init__(self, in_channels, out_channels, kernel_size, resolution, use_attention=False, dropout=0.1,
                 with_se=False, with_se_relu=False, normalize=True, eps=0):
        super().__init__()
        if use_attention:
            raise NotImplementedError("voxel attention is not on the generation hot path")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.resolution = kernel_size, resolution
        self.voxelization = Voxelization(resolution, normalize=normalize, eps=eps)
        layers = [nn.Conv3d(in_channels, out_channels, kernel_size, stride=1, padding=kernel_size // 2),
                  nn.GroupNorm(num_groups=8, num_channels=out_channels), Swish()]
        layers += [nn.Dropout(dropout)] if dropout is not None else []
        layers += [nn.Conv3d(out_channels, out_channels, kernel_size, stride=1, padding=kernel_size // 2),
                   nn.GroupNorm(num_groups=8, num_channels=out_channels), Swish()]
        if with_se:
            layers.append(SE3d(out_channels, use_relu=with_se_relu))
        self.voxel_layers = nn.Sequential(*layers)
        self.point_features = SharedMLP(in_channels, out_channels)

    def forward(self, inputs):
        features, coords = inputs
        vox, norm_coords = self.voxelization(features, coords)
        mods = list(self.voxel_layers)
        convs = [m for m in mods if isinstance(m, nn.Conv3d)]
        norms = [m for m in mods if isinstance(m, nn.GroupNorm)]
        se = mods[-1] if isinstance(mods[-1], SE3d) else None
        from . import voxel
        pf = self.point_features(features)
        # hand-written path: implicit-GEMM conv3d on MFMA (shapes with an instantiation) or the direct any-shape kernel
        # (every other width / resolution, partial edge bricks included), GN+Swish, SE gate folded into the devoxelize
        # pass together with the point-branch add
        plan = self.__dict__.get("_voxel_plan")
        from ._cache import params_key, publish
        key = params_key([c.weight for c in convs], vox.device)
        if plan is None or plan.key != key:
            plan = voxel.VoxelBranchPlan(convs, vox.device, self.resolution)
            plan.key = key
            self.__dict__["_voxel_plan"] = plan
            publish(vox.device)
        return voxel.run(plan, convs, norms, se, vox, norm_coords, pf, self.resolution), coords


# --------------------------------------------------------------------- builders
# (ext/pvcnn/utils.py:65-247, restated)


def create_pointnet_components(blocks, in_channels, with_se=False, normalize=True, eps=0, width_multiplier=1,
                               voxel_resolution_multiplier=1):
    r, vr = width_multiplier, voxel_resolution_multiplier
    layers, concat = [], 0
    for out_channels, num_blocks, voxel_resolution in blocks:
        out_channels = int(r * out_channels)
        if voxel_resolution is None:
            make = SharedMLP
        else:
            make = functools.partial(PVConv, kernel_size=3, resolution=int(vr * voxel_resolution), with_se=with_se,
                                     normalize=normalize, eps=eps)
        for _ in range(num_blocks):
            layers.append(make(in_channels, out_channels))
            in_channels = out_channels
            concat += out_channels
    return layers, in_channels, concat


# The two builders below are table driven: the block tables of a backbone (PVCNN2.sa_blocks / fp_blocks, PointNet2SSG's)
# are first turned into a PLAN -- plain data: per stage the (cin, cout, voxel resolution) of every conv block that exists
# and the arguments of its pooling / propagation module -- and the modules are then made from the plan.  The plan decides
# the nesting of the state_dict keys (a stage with one block is that block, with several an nn.Sequential), so it must
# agree with the reference's construction (ext/pvcnn/utils.py:97-247) -- including that, after the first stage, only the
# FIRST conv block of a multi-block stage exists (utils.py:142) -- and tests/golden/schema_pvcnn2.json /
# schema_pointnet2_ssg.json (captured from the reference's modules) check exactly that through strict loads.
def _scale_widths(widths, r):
    return [_scale_widths(w, r) if isinstance(w, (list, tuple)) else int(r * w) for w in widths]


def _pool_width(widths):
    """Output channels of a pooling module: the last width of its one MLP, or of every branch of a multi-scale one."""
    return sum(w[-1] for w in widths) if isinstance(widths[0], list) else widths[-1]


def sa_plan(sa_blocks, extra_feature_channels, embed_dim=0, use_attention=False, width_multiplier=1,
            voxel_resolution_multiplier=1):
    """-> (stages, sa_in_channels, out_channels, num_centers).  stages[i] = dict(convs=[dict(cin, cout, resolution | None,
    attention)], pool=dict(num_centers | None, radius, num_neighbors, in_channels, out_channels))."""
    width, stages, stage_inputs = extra_feature_channels + 3, [], []
    features = extra_feature_channels
    centers = None
    for i, (conv_cfg, (centers, radius, neighbors, pool_widths)) in enumerate(sa_blocks):
        stage_inputs.append(width)
        convs = []
        if conv_cfg is not None:
            cout, count, vres = conv_cfg
            cout = int(width_multiplier * cout)
            existing = count if i == 0 else min(count, 1)          # utils.py:139-143
            for j in range(existing):
                convs.append(dict(cin=(width if j == 0 else cout) + (embed_dim if i > 0 else 0), cout=cout,
                                  resolution=None if vres is None else int(voxel_resolution_multiplier * vres),
                                  attention=bool(use_attention and i % 2 == 1 and j == 0)))
            if count > 0:
                width = features = cout
        pool = dict(num_centers=centers, radius=radius, num_neighbors=neighbors,
                    in_channels=features + (embed_dim if (conv_cfg is None or conv_cfg[1] == 0) else 0),
                    out_channels=_scale_widths(pool_widths, width_multiplier))
        stages.append(dict(convs=convs, pool=pool))
        width = features = _pool_width(pool["out_channels"])
    return stages, stage_inputs, width, 1 if centers is None else centers


def _conv_block(spec, **pvconv_kw):
    if spec["resolution"] is None:
        return SharedMLP(spec["cin"], spec["cout"])
    return PVConv(spec["cin"], spec["cout"], kernel_size=3, resolution=spec["resolution"],
                  use_attention=spec.get("attention", False), **pvconv_kw)


def _stage_module(blocks):
    return blocks[0] if len(blocks) == 1 else nn.Sequential(*blocks)


def create_pointnet2_sa_components(sa_blocks, extra_feature_channels, embed_dim=0, use_attention=False, dropout=0.1,
                                   with_se=False, voxelization_normalize=True, eps=0, width_multiplier=1,
                                   voxel_resolution_multiplier=1):
    """utils.py:97-182 -> (sa_layers, sa_in_channels, channels_sa_features, num_centers), via sa_plan."""
    stages, stage_inputs, _, centers = sa_plan(sa_blocks, extra_feature_channels, embed_dim, use_attention,
                                               width_multiplier, voxel_resolution_multiplier)
    layers, width = [], None
    for st in stages:
        blocks = [_conv_block(c, dropout=dropout, with_se=with_se, with_se_relu=True, normalize=voxelization_normalize, eps=eps)
                  for c in st["convs"]]
        pool = st["pool"]
        if pool["num_centers"] is None:
            blocks.append(PointNetAModule(in_channels=pool["in_channels"], out_channels=pool["out_channels"],
                                          include_coordinates=True))
        else:
            blocks.append(PointNetSAModule(num_centers=pool["num_centers"], radius=pool["radius"],
                                           num_neighbors=pool["num_neighbors"], in_channels=pool["in_channels"],
                                           out_channels=pool["out_channels"], include_coordinates=True))
        width = blocks[-1].out_channels          # the module knows its own output width (sum over its MLP branches)
        layers.append(_stage_module(blocks))
    return layers, stage_inputs, width, centers


def fp_plan(fp_blocks, in_channels, sa_in_channels, embed_dim=0, width_multiplier=1, voxel_resolution_multiplier=1):
    """-> (stages, out_channels).  stages[i] = dict(fp=dict(in_channels, out_channels), convs=[dict(cin, cout, resolution)])."""
    stages, width = [], in_channels
    skips = list(reversed(sa_in_channels))                     # stage i propagates onto the input of SA stage -1 - i
    for (fp_widths, conv_cfg), skip in zip(fp_blocks, skips):
        outs = tuple(_scale_widths(fp_widths, width_multiplier))
        stage = dict(fp=dict(in_channels=width + skip + embed_dim, out_channels=outs), convs=[])
        width = outs[-1]
        if conv_cfg is not None:
            cout, count, vres = conv_cfg
            cout = int(width_multiplier * cout)
            for _ in range(count):
                stage["convs"].append(dict(cin=width, cout=cout,
                                           resolution=None if vres is None else int(voxel_resolution_multiplier * vres)))
                width = cout
        stages.append(stage)
    return stages, width


def create_pointnet2_fp_modules(fp_blocks, in_channels, sa_in_channels, embed_dim=0, use_attention=False, dropout=0.1,
                                with_se=False, normalize=True, eps=0, width_multiplier=1,
                                voxel_resolution_multiplier=1):
    """utils.py:185-247 -> (fp_layers, out_channels), via fp_plan (attention inside the propagation stages is off in every
    configuration this package builds; PVConv rejects use_attention=True itself)."""
    stages, width = fp_plan(fp_blocks, in_channels, sa_in_channels, embed_dim, width_multiplier, voxel_resolution_multiplier)
    layers = []
    for st in stages:
        blocks = [PointNetFPModule(in_channels=st["fp"]["in_channels"], out_channels=st["fp"]["out_channels"])]
        blocks += [_conv_block(c, dropout=dropout, with_se=with_se, with_se_relu=True, normalize=normalize, eps=eps)
                   for c in st["convs"]]
        layers.append(_stage_module(blocks))
    return layers, width


# -------------------------------------------------------------------- backbones


class PVCNN(nn.Module):
    """pvcnn_base.py:15-177 (the shipped encoder backbone; FiLM conditioning is off in
    every shipped config and not implemented)."""

    def __init__(self, in_channels=3, extra_feature_channels=0, scale_channels=0.25, scale_voxel_resolution=0.75,
                 num_blocks=(1, 2, 1, 1), is_conditioned=False, cond_dims=None, extra_block_channels=None):
        super().__init__()
        assert extra_feature_channels >= 0
        if not isinstance(num_blocks, (list, tuple)):
            raise TypeError("num_blocks must be of type List or Tuple")
        if len(num_blocks) != 4:
            raise ValueError("PVCNN is configured with 4 PVConv modules. The num_blocks sequence must of length 4.")
        if is_conditioned:
            raise NotImplementedError("conditioned PVCNN (FiLM) is not on the generation hot path")
        self.in_channels = in_channels + extra_feature_channels
        self.block_spec = self.get_blocks_spec(scale_channels, scale_voxel_resolution, num_blocks, extra_block_channels)
        self.out_channels = self.block_spec[-1][0]
        layers, _, _ = create_pointnet_components(blocks=self.block_spec, in_channels=self.in_channels, with_se=True,
                                                  normalize=False, width_multiplier=1, voxel_resolution_multiplier=1)
        self.point_features = nn.ModuleList(layers)
        self.is_conditioned = False

    @staticmethod
    def get_blocks_spec(c_mul, r_mul, num_blocks, extra_block_channels=None):
        c = [int(64 * c_mul), int(128 * c_mul), int(1024 * c_mul), int(2048 * c_mul)]
        r = [int(32 * r_mul), int(16 * r_mul), None, None]
        assert all(v % 2 == 0 for v in c) and r[0] % 2 == 0 and r[1] % 2 == 0
        blocks = tuple((c[i], num_blocks[i], r[i]) for i in range(4))
        if extra_block_channels is not None:
            blocks = blocks + tuple((ch, 1, None) for ch in extra_block_channels)
        return blocks

    def forward(self, inputs, *, cond=None):
        features = inputs[:, : self.in_channels, :]
        coords = features[:, :3, :].contiguous()
        for layer in self.point_features:
            features, _ = layer((features, coords))
        return features


class PVCNN2(nn.Module):
    """pvcnn_base.py:180-279"""
    sa_blocks = [((32, 1, 32), (1024, 0.1, 32, (32, 64))), ((64, 2, 16), (256, 0.2, 32, (64, 128))),
                 ((128, 1, 8), (64, 0.4, 32, (128, 256))), (None, (16, 0.8, 32, (256, 256, 512)))]
    fp_blocks = [((256, 256), (256, 1, 8)), ((256, 256), (256, 1, 8)), ((256, 128), (128, 2, 16)),
                 ((128, 128, 64), (64, 1, 32))]

    def __init__(self, in_channels=3, extra_feature_channels=0, width_multiplier=1, voxel_resolution_multiplier=1,
                 use_attention=False, dropout=0.1):
        super().__init__()
        self.in_channels = in_channels + extra_feature_channels
        sa_layers, sa_in, ch_sa, _ = create_pointnet2_sa_components(
            sa_blocks=self.sa_blocks, embed_dim=0, extra_feature_channels=extra_feature_channels, with_se=True,
            voxelization_normalize=True, use_attention=use_attention, dropout=dropout,
            width_multiplier=width_multiplier, voxel_resolution_multiplier=voxel_resolution_multiplier)
        self.sa_layers = nn.ModuleList(sa_layers)
        sa_in[0] = extra_feature_channels
        fp_layers, _ = create_pointnet2_fp_modules(
            fp_blocks=self.fp_blocks, in_channels=ch_sa, sa_in_channels=sa_in, with_se=True,
            width_multiplier=width_multiplier, voxel_resolution_multiplier=voxel_resolution_multiplier)
        self.fp_layers = nn.ModuleList(fp_layers)
        self.out_channels = self.fp_layers[-1][-1].out_channels

    def forward(self, inputs, cond=None):
        if isinstance(inputs, dict):
            inputs = inputs["features"]
        coords, features = inputs[:, :3, :].contiguous(), inputs
        coords_list, feats_list = [], []
        for sa in self.sa_layers:
            feats_list.append(features)
            coords_list.append(coords)
            features, coords = sa((features, coords))
        feats_list[0] = inputs[:, 3:, :].contiguous()
        for i, fp in enumerate(self.fp_layers):
            features, coords = fp((coords_list[-1 - i], coords, features, feats_list[-1 - i]))
        return features


class PointNet2(nn.Module):
    """pointnet2.py:13-95"""

    def __init__(self, sa_blocks, fp_blocks, with_one_hot_shape_id=False, num_shapes=0, extra_feature_channels=3,
                 width_multiplier=1, voxel_resolution_multiplier=1):
        super().__init__()
        assert extra_feature_channels >= 0
        self.in_channels = extra_feature_channels + 3
        self.num_shapes, self.with_one_hot_shape_id = num_shapes, with_one_hot_shape_id
        sa_layers, sa_in, ch_sa, _ = create_pointnet2_sa_components(
            sa_blocks=sa_blocks, extra_feature_channels=extra_feature_channels, width_multiplier=width_multiplier)
        self.sa_layers = nn.ModuleList(sa_layers)
        sa_in[0] += num_shapes if with_one_hot_shape_id else 0
        fp_layers, _ = create_pointnet2_fp_modules(
            fp_blocks=fp_blocks, in_channels=ch_sa, sa_in_channels=sa_in, width_multiplier=width_multiplier,
            voxel_resolution_multiplier=voxel_resolution_multiplier)
        self.fp_layers = nn.ModuleList(fp_layers)

    def forward(self, inputs):
        features = inputs[:, : self.in_channels, :]
        skip0 = inputs if self.with_one_hot_shape_id else features
        coords, features = features[:, :3, :].contiguous(), features[:, 3:, :].contiguous()
        coords_list, feats_list = [], []
        for sa in self.sa_layers:
            feats_list.append(features)
            coords_list.append(coords)
            features, coords = sa((features, coords))
        feats_list[0] = skip0.contiguous()
        for i, fp in enumerate(self.fp_layers):
            features, coords = fp((coords_list[-1 - i], coords, features, feats_list[-1 - i]))
        return features


class PointNet2SSG(PointNet2):
    """pointnet2.py:98-123"""
    sa_blocks = [(None, (512, 0.2, 64, (64, 64, 128))), (None, (128, 0.4, 64, (128, 128, 256))),
                 (None, (None, None, None, (256, 512, 1024)))]
    fp_blocks = [((256, 256), None), ((256, 128), None), ((128, 128, 128), None)]

    def __init__(self, num_shapes=0, extra_feature_channels=3, width_multiplier=1, voxel_resolution_multiplier=1):
        super().__init__(num_shapes=num_shapes, sa_blocks=self.sa_blocks, fp_blocks=self.fp_blocks,
                         with_one_hot_shape_id=False, extra_feature_channels=extra_feature_channels,
                         width_multiplier=width_multiplier, voxel_resolution_multiplier=voxel_resolution_multiplier)
