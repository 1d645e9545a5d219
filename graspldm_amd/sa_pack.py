"""Host-side preparation for the fused set-abstraction kernel (gldm_sa_mlp_forward):
fold eval-mode BatchNorm into each 1x1 conv of a SharedMLP (shared_mlp.py:6-35) and lay the
weights out in v_mfma_f32_16x16x4_f32 A-fragment order (K padded to a multiple of 16)."""
import ctypes

import torch

from . import _lib as L
from .r1d_pack import SplitRangeError, _Buf, mfma_a_fragments, mfma_a_fragments_f16x2

_OK_MTILES = (1, 2, 4, 8, 12, 16)
# modules without features (first layer = three coordinate products + bias): on the VALU of the gather threads (the hoisted
# form with a broadcast row) instead of a K = 32 block of the matrix pipe -- set by measurement, see DESIGN.md
# SSG-SA1 (3 -> 64 -> 64 -> 128, 512 centres x 64 at 256 clouds): hoisted in the single-tile kernel 1.81 ms, first layer on the
# matrix pipe in the multi-tile kernel 1.47 ms, hoisted in the multi-tile kernel (two tiles per pass) 1.22 ms
PRE_WITHOUT_FEATURES = True


def fold_conv_bn(conv, bn):
    w = conv.weight.detach().float().reshape(conv.weight.shape[0], -1)
    s = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + bn.eps)
    return w * s[:, None], (conv.bias.detach().float() - bn.running_mean.detach().float()) * s + bn.bias.detach().float()


def fusable(shared_mlp, num_neighbors):
    layers = shared_mlp.layers
    n = len(layers) // 3
    if n < 1 or n > 4 or 64 % int(num_neighbors) != 0:
        return False
    for i in range(n):
        cout = layers[3 * i].weight.shape[0]
        if cout % 16 or cout > 256 or (cout // 16) not in _OK_MTILES:
            return False
    return (layers[0].weight.shape[1] + 31) // 32 * 32 <= 256


def split_plan_ok(cins, couts, num_neighbors):
    """Shapes gldm_sa_mlp_forward_f16x2 takes (csrc/resnet1d.hip: sa_mlp3_kernel): 64-column tiles on split-f16 planes."""
    from .numerics import split_enabled
    if not split_enabled() or int(num_neighbors) not in (16, 32, 64) or not 1 <= len(couts) <= 4:
        return False
    # hidden widths are packed padded to the 32-row plane blocks (zero weight rows, zero bias: ReLU leaves zeros, and the
    # next layer's weights over those rows are zero): a 16-wide hidden layer (half-width PVCNN2) runs as a 32-wide one
    couts = [(c + 31) // 32 * 32 for c in couts[:-1]] + [couts[-1]]
    kpad = [(cins[0] + 31) // 32 * 32] + list(couts[:-1])
    if any(k % 32 or not (k // 32 <= 6 or k // 32 in (8, 9)) for k in kpad) or any(c % 16 for c in couts):
        return False
    if any(c not in (32, 64, 128, 256) for c in couts[:-1]):
        return False
    blocks_a = max([kpad[0] // 32] + [couts[l] // 32 for l in range(1, len(couts) - 1, 2)])
    blocks_b = max([0] + [couts[l] // 32 for l in range(0, len(couts) - 1, 2)])
    return (blocks_a + blocks_b) * 3072 * 4 <= 160 * 1024


class SaMlpPlan:
    """Packed weights of one SharedMLP(dim=2) on the device + the layer tables.  Where the layer plan fits the split-f16
    kernel (split_plan_ok: the PointNet++ / PVCNN2 set-abstraction shapes), `run` takes that one; the f32-MFMA plan
    is packed either way (other neighbour counts / widths)."""

    def __init__(self, shared_mlp, device):
        layers = shared_mlp.layers
        n = len(layers) // 3
        self._split = None
        self._pre = None
        self._pre_ws = None
        self._layers = shared_mlp
        buf = _Buf()
        cin_pad, cout, w_off, b_off = [], [], [], []
        for i in range(n):
            w, b = fold_conv_bn(layers[3 * i], layers[3 * i + 1])
            # K in pairs of 16-deep blocks: the kernels' weight-fragment pipeline runs two blocks per trip (an odd count
            # would fall back to load-wait-compute per block); the padding rows are zero in the weights and the tile
            kpad = (w.shape[1] + 31) // 32 * 32 if i == 0 else (w.shape[1] + 15) // 16 * 16  # later layers: cin = cout of the previous one
            wp = torch.zeros(w.shape[0], kpad)
            wp[:, : w.shape[1]] = w.cpu()
            cin_pad.append(kpad)
            cout.append(w.shape[0])
            w_off.append(buf.add(mfma_a_fragments(wp)))
            b_off.append(buf.add(b.cpu()))
        self.n_layers = n
        self.weights = buf.tensor().to(device)
        arr = ctypes.c_int32 * n
        self.cin_pad, self.cout, self.w_off, self.b_off = arr(*cin_pad), arr(*cout), arr(*w_off), arr(*b_off)
        self.cout_last = cout[-1]
        self._device = device

    def _split_plan(self):
        """Split-f16 fragments [cout x K padded to 32] per layer + biases, packed on first use."""
        if self._split is None:
            layers = self._layers.layers
            n = len(layers) // 3
            buf = _Buf()
            cin_pad, cout, w_off, b_off, gain = [], [], [], [], []
            for i in range(n):
                w, b = fold_conv_bn(layers[3 * i], layers[3 * i + 1])
                # |layer output| <= gain_r max|input| + gain_b: the kernel scales the hidden layers' planes from this bound
                gain += [float(w.double().abs().sum(dim=1).max()), float(b.double().abs().max())]
                kpad = (w.shape[1] + 31) // 32 * 32
                rows = w.shape[0] if i == n - 1 else (w.shape[0] + 31) // 32 * 32   # hidden widths: whole plane blocks
                wp = torch.zeros(rows, kpad)
                wp[: w.shape[0], : w.shape[1]] = w.cpu()
                bp = torch.zeros(rows)
                bp[: w.shape[0]] = b.cpu()
                cin_pad.append(kpad)
                cout.append(rows)
                w_off.append(buf.add(mfma_a_fragments_f16x2(wp)))
                b_off.append(buf.add(bp))
            arr = ctypes.c_int32 * n
            self._split = (buf.tensor().to(self._device), arr(*cin_pad), arr(*cout), arr(*w_off), arr(*b_off),
                           (ctypes.c_float * (2 * n))(*gain))
        return self._split

    def _pre_plan(self):
        """The split plan with the module's FIRST layer hoisted out of the (centre, neighbour) pairs
        (gldm_sa_mlp_forward_f16x2_pre): W1 [x - c; f] + b1 = W1a (x - c) + (W1b f + b1).  Returns (weights with W1a
        [c1p][4] behind the layer tables, tables of layers 2.., gains, wa_off, W1b [c1p, C] and b1 [c1p] on the device,
        c1p) -- c1p = the first layer's width padded to whole 32-row plane blocks (zero rows)."""
        if self._pre is None:
            layers = self._layers.layers
            n = len(layers) // 3
            buf = _Buf()
            cin_pad, cout, w_off, b_off, gain = [], [], [], [], []
            w1, b1 = fold_conv_bn(layers[0], layers[1])
            c1 = w1.shape[0]
            c1p = (c1 + 31) // 32 * 32
            for i in range(1, n):
                w, b = fold_conv_bn(layers[3 * i], layers[3 * i + 1])
                gain += [float(w.double().abs().sum(dim=1).max()), float(b.double().abs().max())]
                kpad = (w.shape[1] + 31) // 32 * 32
                rows = w.shape[0] if i == n - 1 else (w.shape[0] + 31) // 32 * 32
                wp = torch.zeros(rows, kpad)
                wp[: w.shape[0], : w.shape[1]] = w.cpu()
                bp = torch.zeros(rows)
                bp[: w.shape[0]] = b.cpu()
                cin_pad.append(kpad)
                cout.append(rows)
                w_off.append(buf.add(mfma_a_fragments_f16x2(wp)))
                b_off.append(buf.add(bp))
            wa = torch.zeros(c1p, 4)
            wa[:c1, :3] = w1[:, :3].cpu()
            wa_off = buf.add(wa.reshape(-1))
            w1b = torch.zeros(c1p, w1.shape[1] - 3)
            w1b[:c1] = w1[:, 3:].cpu()
            b1p = torch.zeros(c1p)
            b1p[:c1] = b1.cpu()
            arr = ctypes.c_int32 * (n - 1)
            self._pre = (buf.tensor().to(self._device), arr(*cin_pad), arr(*cout), arr(*w_off), arr(*b_off),
                         (ctypes.c_float * (2 * (n - 1)))(*gain), int(wa_off), w1b.to(self._device), b1p.to(self._device), c1p)
        return self._pre

    def _first_layer_per_point(self, features, w1b, b1p):
        """pre [B, N, c1p] (POINT-major) = W1b f + b1: the split-f16 pointwise launch writing that layout where its shape
        set allows, else the any-shape kernel and a transposing copy."""
        from . import dense
        x = features.contiguous().float()
        c1p, cin = w1b.shape
        if dense.split_mlp_supported(x, cin, c1p):
            if self._pre_ws is None:
                try:
                    self._pre_ws = dense.split_fragments(w1b).to(x.device)
                except SplitRangeError:
                    self._pre_ws = False
            if self._pre_ws is not False:
                b, _, n = x.shape
                y = torch.empty((b, n, c1p), dtype=torch.float32, device=x.device)
                with torch.cuda.device(x.device):
                    L.call("gldm_pointwise_mlp_f16x2_pm", L.ptr(x), L.ptr(self._pre_ws), L.ptr(b1p), b, cin, c1p, n, 0, L.ptr(y),
                           L.current_stream(x.device))
                return y
        return dense._gemm_bias_act(x, w1b, b1p, False).transpose(1, 2).contiguous()

    def run(self, points, centers, features, idx):
        b, _, n = points.shape
        m, u = idx.shape[1], idx.shape[2]
        c = 0 if features is None else features.shape[1]
        out = torch.empty((b, self.cout_last, m), dtype=torch.float32, device=points.device)
        lay = self._layers.layers
        cins = [lay[3 * i].weight.shape[1] for i in range(self.n_layers)]
        couts = [lay[3 * i].weight.shape[0] for i in range(self.n_layers)]
        # first layer per POINT instead of per (centre, neighbour) pair: whenever there are features to hoist and a layer
        # behind it (every point sits in m u / n balls on average: worth it from 2 upwards)
        hoist = (c > 0 and m * u >= 2 * n) or (c == 0 and PRE_WITHOUT_FEATURES)
        if hoist and self.n_layers >= 2 and couts[0] <= 256 and split_plan_ok(couts[:1] + couts[1:-1], couts[1:], u):
            try:
                w3, cin_pad, cout, w_off, b_off, gain, wa_off, w1b, b1p, c1p = self._pre_plan()
            except SplitRangeError:
                w3 = None
            if w3 is not None:
                pre = self._first_layer_per_point(features, w1b, b1p) if c > 0 else b1p   # no features: the row b1 for every point
                with torch.cuda.device(points.device):
                    L.call("gldm_sa_mlp_forward_f16x2_pre", L.ptr(points), L.ptr(centers), L.ptr(pre), 0 if c > 0 else 1, L.ptr(idx), L.ptr(w3),
                           wa_off, b, n, m, u, self.n_layers - 1, ctypes.cast(cin_pad, ctypes.c_void_p),
                           ctypes.cast(cout, ctypes.c_void_p), ctypes.cast(w_off, ctypes.c_void_p),
                           ctypes.cast(b_off, ctypes.c_void_p), ctypes.cast(gain, ctypes.c_void_p), L.ptr(out),
                           L.current_stream(points.device))
                return out
        split = None
        if split_plan_ok(cins, couts, u):
            try:
                split = self._split_plan()
            except SplitRangeError:   # a folded weight beyond the f16 range: the f32-MFMA plan below
                split = None
        if split is not None:
            w3, cin_pad, cout, w_off, b_off, gain = split
            with torch.cuda.device(points.device):
                L.call("gldm_sa_mlp_forward_f16x2", L.ptr(points), L.ptr(centers), L.ptr(features), L.ptr(idx), L.ptr(w3),
                       b, c, n, m, u, self.n_layers, ctypes.cast(cin_pad, ctypes.c_void_p), ctypes.cast(cout, ctypes.c_void_p),
                       ctypes.cast(w_off, ctypes.c_void_p), ctypes.cast(b_off, ctypes.c_void_p),
                       ctypes.cast(gain, ctypes.c_void_p), L.ptr(out), L.current_stream(points.device))
            return out
        with torch.cuda.device(points.device):
            L.call("gldm_sa_mlp_forward", L.ptr(points), L.ptr(centers), L.ptr(features), L.ptr(idx), L.ptr(self.weights),
                   b, c, n, m, u, self.n_layers, ctypes.cast(self.cin_pad, ctypes.c_void_p),
                   ctypes.cast(self.cout, ctypes.c_void_p), ctypes.cast(self.w_off, ctypes.c_void_p),
                   ctypes.cast(self.b_off, ctypes.c_void_p), L.ptr(out), L.current_stream(points.device))
        return out
