"""`tmrp_to_H` (grasp_ldm/utils/rotations.py:298-302: MRP -> quaternion -> SciPy-convention
rotation matrix -> 4x4) on the GPU via gldm_pose_epilogue."""
import torch

from .r1d import pose_epilogue


def tmrp_to_H(tmrp):
    if not tmrp.is_cuda:
        raise RuntimeError("tmrp must be a CUDA tensor (graspldm_amd has no CPU path)")
    shape = tmrp.shape[:-1]
    flat = tmrp.reshape(-1, tmrp.shape[-1])[:, :6].contiguous().float()
    n = flat.shape[0]
    zeros = torch.zeros(1, 6, device=tmrp.device)
    ones = torch.ones(1, 6, device=tmrp.device)
    H, _, _ = pose_epilogue(flat, None, zeros, ones, max(n, 1))
    return H.view(*shape, 4, 4)
