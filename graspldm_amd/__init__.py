"""MI355X (gfx950) implementation of GraspLDM's grasp-generation hot path behind the reference's
module / CLI interface.  The compute lives in libgldm_hip.so (C ABI: include/gldm.h)."""


def invalidate_caches():
    """Rebuild every derived-weight cache on next use (needed only after writes through `p.data`,
    which do not bump the tensor version the caches are keyed on: see graspldm_amd/_cache.py)."""
    from ._cache import invalidate
    invalidate()
