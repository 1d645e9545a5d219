"""Dense layers of the PVCNN encoder (k=1 convs, 3x3x3 voxel convs, GroupNorm,
Swish) on the GPU.

Round-1 state: these run as PyTorch-ROCm device ops (rocBLAS / MIOpen) on CUDA
tensors; the f32-MFMA GEMM / implicit-GEMM kernels that replace them live in
csrc/ as they land (see DESIGN.md "Kernels").  Never a CPU path: CPU tensors are
rejected like everywhere else in this package.
"""
import torch
import torch.nn.functional as F


def _need_cuda(x, name="input"):
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor (graspldm_amd has no CPU path)")


def swish(x):
    _need_cuda(x)
    return x * torch.sigmoid(x)


def pointwise_conv(x, conv):
    """Conv1d/Conv2d with kernel 1 (+bias), no activation."""
    _need_cuda(x)
    fn = F.conv1d if conv.weight.ndim == 3 else F.conv2d
    return fn(x, conv.weight, conv.bias)


def pointwise_conv_bn_relu(x, conv, bn):
    """relu(BN_eval(conv(x))) with BN folded into the weights: y = relu(W' x + b')."""
    _need_cuda(x)
    s = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
    w = conv.weight * s.view(-1, *([1] * (conv.weight.ndim - 1)))
    b = (conv.bias - bn.running_mean) * s + bn.bias
    fn = F.conv1d if conv.weight.ndim == 3 else F.conv2d
    return torch.relu_(fn(x, w, b))


def conv3d_gn_swish(x, conv, gn):
    """Swish(GroupNorm(Conv3d_k3(x)))  (pvconv.py:48-66)."""
    _need_cuda(x)
    h = F.conv3d(x, conv.weight, conv.bias, padding=conv.padding)
    h = F.group_norm(h, gn.num_groups, gn.weight, gn.bias, gn.eps)
    return h * torch.sigmoid(h)


def linear(x, lin):
    _need_cuda(x)
    return F.linear(x, lin.weight, lin.bias)
