"""Host-side preparation for the fused set-abstraction kernel (gldm_sa_mlp_forward):
fold eval-mode BatchNorm into each 1x1 conv of a SharedMLP (shared_mlp.py:6-35) and lay the
weights out in v_mfma_f32_16x16x4_f32 A-fragment order (K padded to a multiple of 16)."""
import ctypes

import torch

from . import _lib as L
from .r1d_pack import _Buf, mfma_a_fragments

_OK_MTILES = (1, 2, 4, 8, 12, 16)


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


class SaMlpPlan:
    """Packed weights of one SharedMLP(dim=2) on the device + the layer tables."""

    def __init__(self, shared_mlp, device):
        layers = shared_mlp.layers
        n = len(layers) // 3
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

    def run(self, points, centers, features, idx):
        b, _, n = points.shape
        m, u = idx.shape[1], idx.shape[2]
        c = 0 if features is None else features.shape[1]
        out = torch.empty((b, self.cout_last, m), dtype=torch.float32, device=points.device)
        with torch.cuda.device(points.device):
            L.call("gldm_sa_mlp_forward", L.ptr(points), L.ptr(centers), L.ptr(features), L.ptr(idx), L.ptr(self.weights),
                   b, c, n, m, u, self.n_layers, ctypes.cast(self.cin_pad, ctypes.c_void_p),
                   ctypes.cast(self.cout, ctypes.c_void_p), ctypes.cast(self.w_off, ctypes.c_void_p),
                   ctypes.cast(self.b_off, ctypes.c_void_p), L.ptr(out), L.current_stream(points.device))
        return out
