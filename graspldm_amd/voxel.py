"""Voxel branch of PVConv on the HIP path: Conv3d(k3) -> GroupNorm(8) -> Swish -> Conv3d(k3) ->
GroupNorm(8) -> Swish -> SE -> trilinear devoxelize (+ point branch), pvconv.py:47-84, as
gldm_conv3d_k3 / gldm_groupnorm_swish / gldm_se_gate / gldm_devoxelize_fused launches."""
import torch

from . import _lib as L
from .r1d_pack import mfma_a_fragments

SUPPORTED = {(3, 6), (6, 3), (2, 8), (4, 4), (8, 2), (4, 8), (2, 4), (4, 2), (1, 4), (2, 2), (4, 1), (8, 1),
             (16, 2), (8, 4)}  # (cout/16, r/4) instantiated in voxel_conv.hip (the last two as two half-width launches)


# (cout, r) of the plane-staging split-f16 kernel (gldm_conv3d_k3_f16x2; cin % 16 == 0): the shipped encoder's two shapes
# and, since round 5, PVCNN2's power-of-two ones
SPLIT_SHAPES = {(48, 24), (96, 12), (32, 32), (64, 32), (32, 16), (64, 16), (128, 16), (64, 8), (128, 8), (256, 8), (128, 4)}


def conv_supported(cout, r):
    return cout % 16 == 0 and r % 4 == 0 and (cout // 16, r // 4) in SUPPORTED


def pack_conv3d(weight):
    """[cout, cin, 3, 3, 3] -> MFMA A fragments of [cout, 27 * cin_pad], k = tap * cin_pad + ci."""
    cout, cin = weight.shape[:2]
    cpad = (cin + 15) // 16 * 16
    w = torch.zeros(cout, 27, cpad, dtype=torch.float32)
    w[:, :, :cin] = weight.detach().float().cpu().reshape(cout, cin, 27).permute(0, 2, 1)
    return mfma_a_fragments(w.reshape(cout, 27 * cpad))


def split_conv_supported(cin, cout, r):
    """Shapes gldm_conv3d_k3_f16x2 is built for (SPLIT_SHAPES with cin % 16 == 0, and the shipped encoder's first conv
    3 -> 48 @ 24^3 with K = 81 packed into three 32-deep blocks)."""
    if (cin, cout, r) == (3, 48, 24):
        from .numerics import split_enabled
        return split_enabled()   # under f32_only() its f32 form runs (K padded to 27 x 16: slower, exact f32 products)
    if cin % 16 or (cout, r) not in SPLIT_SHAPES:
        return False
    from .numerics import split_enabled
    return split_enabled() or not conv_supported(cout, r)   # a shape without an f32 instantiation keeps its only MFMA kernel


def pack_conv3d_fewch_f16x2(weight):
    """[cout, cin <= 4, 3, 3, 3] -> split-f16 A fragments of [cout, roundup(27 cin, 32)], k = tap * cin + ci (tap-major,
    no padding between taps), zero beyond 27 cin."""
    from .r1d_pack import mfma_a_fragments_f16x2
    cout, cin = weight.shape[:2]
    k = 27 * cin
    w = torch.zeros(cout, (k + 31) // 32 * 32, dtype=torch.float32)
    w[:, :k] = weight.detach().float().cpu().reshape(cout, cin, 27).permute(0, 2, 1).reshape(cout, k)
    return mfma_a_fragments_f16x2(w)


def pack_conv3d_f16x2(weight):
    """[cout, cin, 3, 3, 3] (cin % 16 == 0) -> split-f16 A fragments of [cout, cblocks * 14 * 32]: K is walked as
    (16-channel block, pair of taps), k = ((cb * 14 + p) * 32 + 16 (tap - 2 p) + ci; tap 27 (second half of the last
    pair) is zero."""
    from .r1d_pack import mfma_a_fragments_f16x2
    cout, cin = weight.shape[:2]
    cb = cin // 16
    w = torch.zeros(cout, cb, 14, 2, 16, dtype=torch.float32)
    wt = weight.detach().float().cpu().reshape(cout, cb, 16, 27)            # [cout, cb, ci, tap]
    taps = torch.zeros(cout, cb, 16, 28)
    taps[..., :27] = wt
    w[:] = taps.reshape(cout, cb, 16, 14, 2).permute(0, 1, 3, 4, 2)          # [cout, cb, pair, half, ci]
    return mfma_a_fragments_f16x2(w.reshape(cout, cb * 14 * 32))


class VoxelBranchPlan:
    """Packed conv weights of one PVConv on the device (f32 fragments, or split-f16 ones where the conv has that kernel)."""

    def __init__(self, convs, device, r=None):
        from .r1d_pack import SplitRangeError
        self.split = [r is not None and split_conv_supported(c.in_channels, c.out_channels, r) for c in convs]
        # shapes without an MFMA instantiation run the direct kernel on the raw nn.Conv3d weight
        self.generic = [r is not None and not conv_supported(c.out_channels, r) for c in convs]
        self.w = []
        for i, c in enumerate(convs):
            if self.split[i]:
                try:
                    self.w.append((pack_conv3d_fewch_f16x2(c.weight) if c.in_channels < 16 else pack_conv3d_f16x2(c.weight)).to(device))
                    continue
                except SplitRangeError:   # a weight beyond the f16 range: this conv keeps an f32 kernel
                    self.split[i] = False
            self.w.append((c.weight.detach().float().contiguous() if self.generic[i] else pack_conv3d(c.weight)).to(device))
        self.key = None  # set by the owner (PVConv.forward) from _cache.params_key


def run(plan, convs, norms, se, vox, norm_coords, point_feat, r):
    """vox [B, Cin, r, r, r] -> fused features [B, Cout, N] = gate * devox(voxel stack) + point_feat.

    GroupNorm + Swish never run as passes of their own where the consumer can apply them: every conv leaves its raw output
    and its per-brick statistics; gldm_groupnorm_coef folds those with the norm's affine into (a, s) per cloud and channel;
    a following split-f16 conv applies swish(a x + s) while it stages its bricks (gldm_conv3d_k3_f16x2_gn), the SE squeeze
    is a read-only pass (gldm_gn_swish_chan_sum) and the devoxelize pass applies it to the eight corners it reads
    (gldm_devoxelize_gn_fused).  The last conv of the stack, when it is one of the plane-staging kernels, writes its output
    channel-last so that both of those readers take a voxel's channels as one run (gldm_gn_swish_chan_sum_cl,
    gldm_devoxelize_gn_cl_fused: a point's eight corners are eight 16-byte-per-lane runs instead of 8 C scattered dwords).
    A conv without that staging (f32 / generic kernels) still gets gldm_groupnorm_swish in
    front of it."""
    dev = vox.device
    b = vox.shape[0]
    st = L.current_stream(dev)
    x = vox.contiguous()
    coef = None          # (a, s) of the GroupNorm + Swish still to be applied to x, or None
    chan_sum = None
    x_cl = False         # x is channel-last, [B, r^3, C]
    with torch.cuda.device(dev):
        for i, (conv, gn) in enumerate(zip(convs, norms)):
            cin, cout = conv.in_channels, conv.out_channels
            y = torch.empty((b, cout, r, r, r), dtype=torch.float32, device=dev)
            nf = L.lib().gldm_conv3d_partial_floats(b, cout, r)
            partial = torch.empty(int(nf), dtype=torch.float32, device=dev)
            last = i == len(convs) - 1
            staged = plan.split[i] and cin % 16 == 0          # the plane-staging kernels: folded input, channel-last output
            mfma32 = not plan.split[i] and not plan.generic[i]   # the f32-MFMA kernels: channel-last output
            # the last conv's readers (squeeze, devoxelize) take a voxel's channels as one run
            cl = last and (staged or mfma32) and cout % 4 == 0 and cout <= 256 and cout // gn.num_groups <= 64
            if staged and (coef is not None or cl):
                L.call("gldm_conv3d_k3_f16x2_gn", L.ptr(x), L.ptr(coef), L.ptr(plan.w[i]), L.ptr(conv.bias), b, cin, cout, r,
                       L.ptr(y), L.ptr(partial), 1 if cl else 0, st)
            elif cl:
                assert coef is None
                L.call("gldm_conv3d_k3_cl", L.ptr(x), L.ptr(plan.w[i]), L.ptr(conv.bias), b, cin, cout, r, L.ptr(y), L.ptr(partial), st)
            else:
                assert coef is None
                entry = "gldm_conv3d_k3_generic" if plan.generic[i] else ("gldm_conv3d_k3_f16x2" if plan.split[i] else "gldm_conv3d_k3")
                L.call(entry, L.ptr(x), L.ptr(plan.w[i]), L.ptr(conv.bias), b, cin, cout, r, L.ptr(y), L.ptr(partial), st)
            x_cl = cl
            # the consumer of this conv's GroupNorm + Swish: the next conv if it stages planes (cin % 16 == 0 split kernel),
            # else the SE pass + devoxelize (last conv), else a pass of its own
            # (measured per 256 clouds: at 24^3 the pass costs 0.26 ms and the staged form 0.1 ms; at 12^3 the pass is
            # 0.06 ms and the staged form 0.25 ms -- six channel blocks of exp / rcp in front of short tap loops -- so small
            # grids keep the pass between their convs)
            foldable = (last or (plan.split[i + 1] and convs[i + 1].in_channels % 16 == 0 and r >= 16)) \
                and cout // gn.num_groups <= 64
            if foldable:
                coef = torch.empty((b, cout, 2), dtype=torch.float32, device=dev)
                L.call("gldm_groupnorm_coef", L.ptr(partial), L.ptr(gn.weight), L.ptr(gn.bias), b, cout, r, gn.num_groups,
                       float(gn.eps), L.ptr(coef), st)
            else:
                coef = None
                if last and se is not None:
                    chan_sum = torch.empty((b, cout), dtype=torch.float32, device=dev)
                L.call("gldm_groupnorm_swish", L.ptr(y), L.ptr(partial), L.ptr(gn.weight), L.ptr(gn.bias), b, cout, r,
                       gn.num_groups, float(gn.eps), L.ptr(chan_sum) if (last and se is not None) else None, st)
            x = y
        c = x.shape[1]
        gate = None
        if se is not None:
            gate = torch.empty((b, c), dtype=torch.float32, device=dev)
            w1, w2 = se.fc[0].weight, se.fc[2].weight
            if x_cl:
                parts = int(L.lib().gldm_squeeze_parts())
                chan_sum = torch.empty((b, parts, c), dtype=torch.float32, device=dev)
                L.call("gldm_gn_swish_chan_sum_cl", L.ptr(x), L.ptr(coef), b, c, r, L.ptr(chan_sum), st)
                L.call("gldm_se_gate_parts", L.ptr(chan_sum), parts, L.ptr(w1), L.ptr(w2), b, c, w1.shape[0], r,
                       1 if se.use_relu else 0, L.ptr(gate), st)
            else:
                if coef is not None:
                    chan_sum = torch.empty((b, c), dtype=torch.float32, device=dev)
                    L.call("gldm_gn_swish_chan_sum", L.ptr(x), L.ptr(coef), b, c, r, L.ptr(chan_sum), st)
                L.call("gldm_se_gate", L.ptr(chan_sum), L.ptr(w1), L.ptr(w2), b, c, w1.shape[0], r,
                       1 if se.use_relu else 0, L.ptr(gate), st)
        n = norm_coords.shape[2]
        out = torch.empty((b, c, n), dtype=torch.float32, device=dev)
        pf = point_feat.contiguous() if point_feat is not None else None
        if x_cl:
            L.call("gldm_devoxelize_gn_cl_fused", L.ptr(norm_coords), L.ptr(x), L.ptr(coef), L.ptr(gate), L.ptr(pf), b, c, n, r,
                   L.ptr(out), st)
        elif coef is not None:
            L.call("gldm_devoxelize_gn_fused", L.ptr(norm_coords), L.ptr(x), L.ptr(coef), L.ptr(gate), L.ptr(pf), b, c, n, r,
                   L.ptr(out), st)
        else:
            L.call("gldm_devoxelize_fused", L.ptr(norm_coords), L.ptr(x), L.ptr(gate), L.ptr(pf), b, c, n, r, L.ptr(out), st)
    return out
