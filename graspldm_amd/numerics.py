"""Which arithmetic the GEMMs run on.

Default: every GEMM whose shape has a split kernel multiplies as three f16 partial products per f32 product on the f16
matrix pipe (hi + lo operands, f32 accumulation, ~1e-7 of sum|a b| per output -- the accuracy of an f32 fma
chain, but not its bits).  Shapes without a split kernel run exact f32 products on the f32 matrix pipe (k-ordered fma chain).
Results therefore depend, in the last bits, on the PATH a shape takes (n % 16, channel divisibility, ABI fields present).

`f32_only()` forces the f32-pipe kernels everywhere they exist, for reproducibility comparisons against the exact-f32 path
(3-5 x slower on the hot kernels):

    with graspldm_amd.numerics.f32_only():
        model = build_fpc_ldm(...)      # engines pack their descriptors without the split copies
        out = model.generate_grasps(...)

The switch is read where weights are packed / a launch is chosen; it is part of every derived-weight cache key
(_cache.params_key), so entering or leaving it repacks the plans instead of mixing the two arithmetics.  Voxel convs
without an f32 instantiation keep their only kernel; the shipped encoder's first 3 -> 48 conv runs its f32 form (K padded to
27 x 16) under the switch since round 6."""
import contextlib

_F32_ONLY = False


def split_enabled():
    return not _F32_ONLY


@contextlib.contextmanager
def f32_only(on=True):
    global _F32_ONLY
    old = _F32_ONLY
    _F32_ONLY = bool(on)
    try:
        yield
    finally:
        _F32_ONLY = old
