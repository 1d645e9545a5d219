"""Derived-weight caches (BatchNorm-folded GEMM weights, the folded encoder head, MFMA-packed
set-abstraction / voxel / ResNet1D weight buffers) share two rules:

  * key: an entry is rebuilt when any source tensor's (data_ptr, _version) changes, when the
    device changes, when the arithmetic mode changes (numerics.f32_only), or after `invalidate()`.  `_version` is bumped by in-place autograd-visible
    writes (`load_state_dict`, `p.copy_()`, optimiser steps) but NOT by writes through `p.data`;
    code that edits weights that way calls `graspldm_amd.invalidate_caches()`.
  * publication: an entry is produced by copies / kernels on whichever HIP stream is current at
    first use, and later consumed by launches on other streams (bench.py rotates three).  `publish()`
    blocks the host once, when the entry is built, until that stream has finished; every launch
    issued afterwards, on any stream, sees completed data.  Entries live on their module
    (no global dict keyed by id()).
"""
import torch

_EPOCH = [0]


def invalidate():
    """Drop every derived-weight cache (they rebuild on next use)."""
    _EPOCH[0] += 1


def params_key(tensors, device, *extra):
    # the arithmetic mode is part of every key: a plan packed under numerics.f32_only() (no split copies, f32 launches) is
    # not the plan of the default mode, and entering / leaving the switch after a module's first forward repacks instead of
    # silently mixing the two arithmetics
    from .numerics import split_enabled
    return (_EPOCH[0], str(device), split_enabled()) + tuple(extra) + tuple((t.data_ptr(), t._version) for t in tensors)


def publish(device):
    device = torch.device(device)
    if device.type == "cuda":
        torch.cuda.current_stream(device).synchronize()
