"""Dense k = 1 layers of the point-cloud encoders: every one of them a hand-written launch, on every encoder.  Shipped
PVCNN encoder: the wide SharedMLP layers + head (gldm_pointwise_mlp*: MFMA GEMMs, the 768 -> 1536 one on split-f16
operands), the narrow point-branch convs (gldm_pointwise_small) and the Linear over the point axis (gldm_linear_rows).
Layer shapes outside those kernels' sets (PVCNN2 / PointNet++ widths such as 384 -> 256 over 128 centres) run in the
any-shape f32-MFMA kernel (gldm_pointwise_any); since round 4 nothing here reaches rocBLAS / MIOpen, and the package no
longer sets MIOPEN_FIND_MODE.  Voxel convs have no library path either (csrc/voxel_conv.hip, with a direct VALU kernel
for shapes without an MFMA instantiation).  Never a CPU path: CPU tensors are rejected like everywhere else in this package.
"""
import ctypes

import torch


def range_gain(w2d, bias):
    """(largest row sum of |W|, largest |bias|) of a folded layer as a host float[2]: |W x + b| <= gain[0] max|x| + gain[1].
    The split-f16 launches scale a hidden layer's planes into the f16 range from this bound (include/gldm.h, ABI 10)."""
    w = w2d.detach().double().cpu()
    r = float(w.abs().sum(dim=1).max()) if w.numel() else 0.0
    b = float(bias.detach().double().abs().max()) if bias is not None and bias.numel() else 0.0
    return (ctypes.c_float * 2)(r, b)


def _need_cuda(x, name="input"):
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor (graspldm_amd has no CPU path)")


def swish(x):
    _need_cuda(x)
    return x * torch.sigmoid(x)


SMALL_CIN = (3, 6, 16, 24, 32, 48, 64)   # widths gldm_pointwise_small is instantiated for


def _gemm_bias_act(x, w2d, bias, relu):
    """act(W x + b) over [B, Cin, ...].  Narrow inputs (cin in SMALL_CIN: the PVConv point branches of the shipped
    encoder) run in the hand-written lane-per-point kernel (gldm_pointwise_small); every other shape the fused MFMA
    launches are not built for (PVCNN2 / PointNet++ widths outside the shipped encoder) in the any-shape f32-MFMA kernel
    (gldm_pointwise_any: GEMM + bias + activation in one launch).  No library GEMM."""
    from . import _lib as L
    shape = x.shape
    x3 = x.reshape(shape[0], shape[1], -1).contiguous()
    n = x3.shape[-1]
    y = torch.empty((shape[0], w2d.shape[0], n), dtype=torch.float32, device=x.device)
    wc = w2d.contiguous()
    bc = bias.contiguous() if bias is not None else None
    entry = "gldm_pointwise_small" if shape[1] in SMALL_CIN else "gldm_pointwise_any"
    with torch.cuda.device(x.device):
        L.call(entry, L.ptr(x3), L.ptr(wc), L.ptr(bc), int(shape[0]), int(shape[1]), int(w2d.shape[0]), n, int(relu),
               L.ptr(y), L.current_stream(x.device))
    return y.reshape(shape[0], w2d.shape[0], *shape[2:])


def pack_head(wh):
    """[hout <= 16, cout] head weights for gldm_pointwise_mlp: A-fragment order with the k index of every 16-block
    permuted so that k-step r holds rows {4 kq + r} of a 16-row block of y, the order in which a lane's accumulator
    registers hold them (k' = 4 (k % 4) + k // 4)."""
    from .r1d_pack import mfma_a_fragments
    hout, cout = wh.shape
    w = torch.zeros(16, cout, dtype=torch.float32)
    w[:hout] = wh.detach().float().cpu()
    w = w.view(16, cout // 16, 4, 4).permute(0, 1, 3, 2).reshape(16, cout)   # [.., kq, r] -> [.., r, kq]
    return mfma_a_fragments(w)


SPLIT_PAD_MIN_CIN = 32   # narrowest input the split launch takes with K padded to 128 (below: the lane-per-point kernel)


def split_supported(cin, cin0=0):
    """The split-f16 form of the fused launch (gldm_pointwise_mlp*_f16x2): A ring of four 32-deep blocks, the tile
    as planes (48 floats per channel) + the front layer's f32 tile (split once per wave into registers: cin0 <= 96).
    Without a front layer any multiple of 8 input rows from 32 up: K is zero-padded to whole trips of the ring
    (split_fragments)."""
    from .numerics import split_enabled
    kpad = (cin + 127) // 128 * 128
    ok_k = cin % 128 == 0 or (cin0 == 0 and cin % 8 == 0 and cin >= SPLIT_PAD_MIN_CIN)
    return split_enabled() and ok_k and cin0 % 32 == 0 and cin0 <= 96 and 4 * (48 * kpad + 32 * cin0) + 16 <= 160 * 1024


def split_fragments(w2d):
    """Split-f16 A fragments of [M, K] with K zero-padded to a multiple of 128 (the K gldm_pointwise_mlp_f16x2 walks)."""
    from .r1d_pack import mfma_a_fragments_f16x2
    w = w2d.detach().float().cpu()
    k = w.shape[1]
    kp = (k + 127) // 128 * 128
    if kp != k:
        w = torch.cat([w, torch.zeros(w.shape[0], kp - k)], dim=1)
    return mfma_a_fragments_f16x2(w)


def fused_mlp_supported(x, cin, cout):
    return (x.ndim == 3 and x.is_contiguous() and cin % 32 == 0 and cout % 256 == 0 and x.shape[-1] % 32 == 0
            and 4 * (32 * cin + 4096) <= 160 * 1024)


def split_mlp_supported(x, cin, cout):
    """The split-f16 launch on its own (gldm_pointwise_mlp_f16x2): output rows in units of 32 (fewer than 256 leave waves
    idle, still several times the any-shape kernel's rate: the 128-row feature-propagation layers of PointNet++ / PVCNN2)."""
    rows_ok = cout >= 64 and (cout % 32 == 0 or (cout < 256 and cout % 16 == 0))
    return x.ndim == 3 and x.is_contiguous() and rows_ok and x.shape[-1] % 32 == 0 and split_supported(cin)


def fused_mlp2_supported(x, cin0, cin, cout):
    """Two layers in one launch (gldm_pointwise_mlp2): cin0 -> cin -> cout."""
    return (x.ndim == 3 and x.is_contiguous() and cin0 % 32 == 0 and cin % 256 == 0 and cout % 256 == 0
            and x.shape[-1] % 32 == 0 and 4 * (32 * (cin + cin0) + 4096) <= 160 * 1024)


def pointwise_mlp(x, w_packed, bias, cout, relu, head=None, keep_y=True, front=None, split=False):
    """One fused launch: y = act(W x + b) over [B, Cin, N] (hand-written MFMA GEMM, csrc/resnet1d.hip:
    pointwise_mlp_kernel / pointwise_mlp_sp_kernel) and optionally z = Wh y + bh on the accumulators.
    head = (packed Wh, bh, hout).  front = (packed W0, b0, cin[, gain]): a ReLU layer x -> relu(W0 x + b0) in front, its
    output kept in LDS only; gain = range_gain(W0, b0) lets the split launch scale that layer's planes into the f16 range.
    split=True: `w_packed` (and the front layer's W0) hold split-f16 fragments (the GEMMs run on the f16 matrix pipe with
    three partial products per f32 product).

    Numerics depend on the path a shape takes: the split launch (hi + lo f16 operands, three products: ~2e-7 of sum|a b|
    per output, measured 1.7e-6 on a denoiser forward) and the f32-MFMA launch (exact f32 fma chain in k order) differ in
    the last bits, and so do two shapes of one layer that land on different launches (e.g. cout 64 vs 48).  Both are
    within the parity bars (2e-5 per forward); bitwise equality holds between runs of the SAME path only."""
    from . import _lib as L
    b, cin, n = x.shape
    y = torch.empty((b, cout, n), dtype=torch.float32, device=x.device) if keep_y or head is None else None
    z = torch.empty((b, head[2], n), dtype=torch.float32, device=x.device) if head is not None else None
    hp = (L.ptr(head[0]) if head else None, L.ptr(head[1]) if head else None, head[2] if head else 0)
    with torch.cuda.device(x.device):
        if front is not None:
            if not relu:
                raise ValueError("the two-layer launch applies ReLU after both layers")
            if split:
                gain = front[3] if len(front) > 3 else None
                L.call("gldm_pointwise_mlp2_f16x2", L.ptr(x), L.ptr(front[0]), L.ptr(front[1]), cin, L.ptr(w_packed),
                       L.ptr(bias), b, front[2], cout, n, *hp, ctypes.cast(gain, ctypes.c_void_p) if gain is not None else None,
                       L.ptr(y), L.ptr(z), L.current_stream(x.device))
            else:
                L.call("gldm_pointwise_mlp2", L.ptr(x), L.ptr(front[0]), L.ptr(front[1]), cin, L.ptr(w_packed), L.ptr(bias),
                       b, front[2], cout, n, *hp, L.ptr(y), L.ptr(z), L.current_stream(x.device))
        else:
            L.call("gldm_pointwise_mlp_f16x2" if split else "gldm_pointwise_mlp", L.ptr(x), L.ptr(w_packed), L.ptr(bias),
                   b, cin, cout, n, int(relu), *hp, L.ptr(y), L.ptr(z), L.current_stream(x.device))
    return y, z


def pointwise_gemm(x, w2d, bias):
    """W x + b over [B, Cin, ...] with an explicit (e.g. folded) weight matrix."""
    _need_cuda(x)
    return _gemm_bias_act(x.float(), w2d, bias, False)


def pointwise_conv(x, conv):
    """Conv1d/Conv2d with kernel 1 (+bias), no activation."""
    _need_cuda(x)
    return _gemm_bias_act(x.float(), conv.weight.reshape(conv.weight.shape[0], -1), conv.bias, False)


def folded_conv_bn(conv, bn, device):
    """BatchNorm(eval) folded into the k = 1 conv: (W', b', W' packed for gldm_pointwise_mlp or None, W' as split-f16
    fragments or None), on `device`, computed once per (weights, statistics) version and kept on the conv module."""
    from ._cache import params_key, publish
    src = [conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var] + ([conv.bias] if conv.bias is not None else [])
    key = params_key(src, device)
    hit = conv.__dict__.get("_gldm_folded")  # lives and dies with the module
    if hit is None or hit[0] != key:
        from .r1d_pack import SplitRangeError, mfma_a_fragments, mfma_a_fragments_f16x2
        s = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        w = (conv.weight.reshape(conv.weight.shape[0], -1) * s.view(-1, 1)).contiguous()
        cb = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
        b = ((cb - bn.running_mean) * s + bn.bias).contiguous()
        wp = ws = None
        try:   # a folded weight beyond the f16 range (|w| >= 65504: a huge BatchNorm gain) keeps the f32-pipe kernels
            if w.shape[1] % 32 == 0 and w.shape[0] % 256 == 0:
                wp = mfma_a_fragments(w.detach().float().cpu()).to(device)
                # split fragments: main layers of the split launch (cin % 128 == 0) and its narrow front layers (cin <= 128)
                if split_supported(w.shape[1]) or w.shape[1] <= 128:
                    ws = mfma_a_fragments_f16x2(w.detach().float().cpu()).to(device)
            elif w.shape[0] % 16 == 0 and w.shape[0] >= 64 and split_supported(w.shape[1]):
                ws = split_fragments(w).to(device)   # 64 .. 240 output rows / K padded to the ring: split launch only
        except SplitRangeError:
            ws = None
        hit = (key, w, b, wp, ws, range_gain(w, b))
        conv.__dict__["_gldm_folded"] = hit
        publish(device)
    return hit[1], hit[2], hit[3], hit[4]


def folded_range_gain(conv):
    """range_gain of the layer folded_conv_bn packed last (same weight version)."""
    return conv.__dict__["_gldm_folded"][5]


def pointwise_conv_bn_relu(x, conv, bn):
    """relu(BN_eval(conv(x))) with BN folded into the weights: y = relu(W' x + b').  Wide layers (cin % 32 == 0,
    cout % 256 == 0) run as ONE hand-written MFMA launch (GEMM + bias + ReLU in the native layout); the others in the
    narrow lane-per-point kernel or the any-shape f32-MFMA kernel (_gemm_bias_act)."""
    _need_cuda(x)
    w, b, wp, ws = folded_conv_bn(conv, bn, x.device)
    x = x.float()
    if wp is not None and fused_mlp_supported(x, w.shape[1], w.shape[0]):
        if ws is not None and w.shape[1] % 128 == 0 and split_supported(w.shape[1]):   # (narrow layers keep UNPADDED fragments, as front layers)
            return pointwise_mlp(x, ws, b, w.shape[0], True, split=True)[0]
        return pointwise_mlp(x, wp, b, w.shape[0], True)[0]
    if wp is None and ws is not None and split_mlp_supported(x, w.shape[1], w.shape[0]):
        return pointwise_mlp(x, ws, b, w.shape[0], True, split=True)[0]
    return _gemm_bias_act(x, w, b, True)


def concat_conv_bn_relu(xa, xb, conv, bn):
    """relu(BN_eval(conv(cat([xa, xb], dim=1)))) of a k = 1 conv WITHOUT building the concatenation:
    W [xa; xb] = Wa xa + Wb xb.  The wide part runs as the split-f16 launch with the other part as its addend
    (gldm_pointwise_mlp_f16x2_add).  Three shapes of the PointNet++-style backbones (pointnet.py:11-46, 117-135):
      * xa [B, Ca, N] wide (Ca % 128 == 0), xb [B, Cb, N] a few rows (Cb in SMALL_CIN: coordinates / raw features):
        addend = Wb xb + b from the lane-per-point kernel, a [B, Cout, N] tensor;
      * xa [B, Ca, 1] ONE centre's features that nearest-neighbour interpolation would broadcast to every point,
        xb [B, Cb, N] wide (Cb % 128 == 0): addend = Wa xa, a per-cloud bias [B, Cout];
      * both wide (Ca % 128 == 0 and Cb % 128 == 0): addend = Wb xb + b from a split launch of its own.
    Returns None when none applies (the caller concatenates and takes the plain path)."""
    from . import _lib as L
    if not (xa.is_cuda and xb.is_cuda and xa.ndim == 3 and xb.ndim == 3 and xa.dtype == torch.float32 and xb.dtype == torch.float32):
        return None
    bsz, ca, na = xa.shape
    cb, n = xb.shape[1], xb.shape[2]
    cout = conv.weight.shape[0]
    if conv.weight.shape[1] != ca + cb or cout % 32 or cout < 64 or n % 32:
        return None
    broadcast = na == 1 and n > 1
    if not broadcast and na != n:
        return None
    wide_c = cb if broadcast else ca
    # third form: BOTH parts wide (a feature-propagation layer joining two 128 / 256-channel tensors, PointNet2SSG's
    # 256 + 128 -> 256): Wb xb as a split launch of its own, then Wa xa with it as the addend
    both_wide = not broadcast and cb >= 128 and ca >= 128 and split_supported(ca) and split_supported(cb)
    if not split_supported(wide_c) or (not broadcast and cb not in SMALL_CIN and not both_wide):
        return None
    from ._cache import params_key, publish
    src = [conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var] + ([conv.bias] if conv.bias is not None else [])
    key = (params_key(src, xa.device), ca, broadcast, both_wide)
    hit = conv.__dict__.get("_gldm_concat")
    if hit is None or hit[0] != key:
        from .r1d_pack import SplitRangeError, mfma_a_fragments_f16x2
        w, b = folded_conv_bn(conv, bn, xa.device)[:2]
        wa, wb = w[:, :ca].contiguous(), w[:, ca:].contiguous()
        wide = wb if broadcast else wa
        try:
            w3 = split_fragments(wide).to(xa.device)
            w_other = wa if broadcast else (split_fragments(wb).to(xa.device) if both_wide else wb)
            # broadcast form: Wa xa over the clouds is itself a wide GEMM ([1, Ca, B] columns = clouds) where Ca allows
            wa3 = split_fragments(wa).to(xa.device) if broadcast and split_supported(ca) else None
        except SplitRangeError:
            return None   # a weight beyond the f16 range: the caller concatenates and takes the plain (f32) path
        hit = (key, w3, w_other, b, torch.zeros_like(b), wa3)
        conv.__dict__["_gldm_concat"] = hit
        publish(xa.device)
    _, w3, w_other, b, zero_b, wa3 = hit
    y = torch.empty((bsz, cout, n), dtype=torch.float32, device=xa.device)
    if broadcast:
        # Wa xa for every cloud at once: [1, Ca, B] columns = clouds (the split launch where the cloud count is a multiple of
        # 32 and Ca of 128, else the any-shape kernel)
        x1 = xa[:, :, 0].t().contiguous().unsqueeze(0)
        if wa3 is not None and split_mlp_supported(x1, ca, cout):
            g = pointwise_mlp(x1, wa3, zero_b, cout, False, split=True)[0][0].t().contiguous()                      # [B, Cout]
        else:
            g = _gemm_bias_act(x1, w_other, None, False)[0].t().contiguous()
        xw, add, strides, bias = xb.contiguous(), g, (cout, 1, 0), b
    elif both_wide:
        add = pointwise_mlp(xb.contiguous(), w_other, b, cout, False, split=True)[0]                             # [B, Cout, N] = Wb xb + b
        xw, strides, bias = xa.contiguous(), (cout * n, n, 1), zero_b
    else:
        add = _gemm_bias_act(xb.contiguous(), w_other, b, False)                                                   # [B, Cout, N]
        xw, strides, bias = xa.contiguous(), (cout * n, n, 1), zero_b
    with torch.cuda.device(xa.device):
        L.call("gldm_pointwise_mlp_f16x2_add", L.ptr(xw), L.ptr(w3), L.ptr(bias), L.ptr(add), *strides, bsz, wide_c, cout, n, 1,
               L.ptr(y), L.current_stream(xa.device))
    return y


def row_max(x):
    """[B, C, N] -> [B, C, 1] = x.max(dim=-1, keepdim=True).values (PointNetAModule's global pooling), one launch."""
    from . import _lib as L
    _need_cuda(x)
    xf = x.contiguous().float()
    b, c, n = xf.shape
    out = torch.empty((b, c, 1), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.call("gldm_row_max", L.ptr(xf), b * c, n, L.ptr(out), L.current_stream(x.device))
    return out


def linear(x, lin):
    """nn.Linear over the last axis of [..., n] (the encoder's out_layer[1] over the POINT axis): hand-written row
    kernel (gldm_linear_rows); n % 4 == 0 and n <= 16384, else the library GEMM."""
    _need_cuda(x)
    n, nout = lin.weight.shape[1], lin.weight.shape[0]
    if n % 4 == 0 and n <= 16384:
        from . import _lib as L
        xf = x.contiguous().float()
        rows = xf.numel() // n
        y = torch.empty((*xf.shape[:-1], nout), dtype=torch.float32, device=x.device)
        w = lin.weight.contiguous()
        with torch.cuda.device(x.device):
            L.call("gldm_linear_rows", L.ptr(xf), L.ptr(w), L.ptr(lin.bias), rows, n, nout, L.ptr(y),
                   L.current_stream(x.device))
        return y
    # other row lengths: the same product as a k = 1 conv over the transposed rows (gldm_pointwise_any), rare
    xt = x.reshape(-1, n).float().t().contiguous().unsqueeze(0)                     # [1, n, rows]
    yt = _gemm_bias_act(xt, lin.weight.float(), lin.bias, False)                      # [1, nout, rows]
    return yt[0].t().reshape(*x.shape[:-1], nout).contiguous()
