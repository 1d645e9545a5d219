"""TEST INFRASTRUCTURE, NOT PRODUCT CODE: numpy / torch-CPU restatement of the raw-cloud front end
(SURVEY.md 8f-1), each function citing the reference lines it follows.  Pinned by
tests/golden/front_end.npz, captured from the reference's own PointCloudHelpers
(`python -m oracle.make_golden front_end`); `normalize_input` is SELF-PINNED (tools/inference.py
cannot be imported here: pytorch_lightning / h5py / trimesh are absent)."""
import numpy as np
import torch


def distance_by_translation_point(p1, p2):
    """pointcloud_helpers.py:219-223"""
    return np.sqrt(np.sum(np.square(p1 - p2), axis=-1))


def farthest_points(data, nclusters):
    """pointcloud_helpers.py:160-217 with return_center_indexes=True: -> centre indexes (int32)."""
    if nclusters >= data.shape[0]:
        return np.arange(data.shape[0], dtype=np.int32)
    distances = np.ones((data.shape[0],), dtype=np.float32) * 1e7
    centers = []
    for _ in range(nclusters):
        index = np.argmax(distances)
        centers.append(index)
        new_distances = distance_by_translation_point(np.expand_dims(data[index], 0), data)
        distances = np.minimum(distances, new_distances)
    return np.asarray(centers, dtype=np.int32)


def regularize_pc_point_count(pc, npoints, use_farthest_point=False, rng=np.random):
    """pointcloud_helpers.py:124-158"""
    if pc.shape[0] > npoints:
        if use_farthest_point:
            center_indexes = farthest_points(pc, npoints)
        else:
            center_indexes = rng.choice(range(pc.shape[0]), size=npoints, replace=False)
        pc = pc[center_indexes, :]
    else:
        required = npoints - pc.shape[0]
        if required > 0:
            index = rng.choice(range(pc.shape[0]), size=required)
            pc = np.concatenate((pc, pc[index, :]), axis=0)
    return pc


def regularize_pointcloud(pc, num_points):
    """pointcloud_helpers.py:40-71 (torch; draws from the global CPU generator)."""
    if pc.shape[0] < num_points:
        multiplier = max(num_points // pc.shape[0], 1)
        pc = pc.repeat(multiplier, 1)
        num_extra_points = num_points - pc.shape[0]
        extra_points = pc[torch.randperm(pc.shape[0])[:num_extra_points]]
        pc = torch.cat((pc, extra_points), dim=0)
    elif pc.shape[0] > num_points:
        pc = pc[torch.randperm(pc.shape[0])[:num_points]]
    return pc.unsqueeze(0)


def normalize_input(pc, pc_shift=0.0, pc_scale=0.05, mrp_scale=0.5):
    """tools/inference.py:570-591 for a batch [B,N,3] (class constants :403-414 of the same file:
    PC_MEAN 0, PC_STD 0.05, GRASP_MEAN 0, GRASP_STD [0.05 x3, 0.5 x3])."""
    pc = pc.clone()
    pc_mean = torch.mean(pc, dim=-2)
    pc -= pc_mean.unsqueeze(1)
    pc = (pc - pc_shift) / pc_scale
    grasp_mean = torch.zeros(6).unsqueeze(0).repeat(pc.shape[0], 1)
    grasp_mean[..., :3] += pc_mean
    std = torch.tensor([pc_scale] * 3 + [mrp_scale] * 3)
    metas = dict(pc_mean=pc_shift + pc_mean, pc_std=torch.full((1, 3), pc_scale), grasp_mean=grasp_mean,
                 grasp_std=std.unsqueeze(0), dataset_normalized=True)
    return pc, metas
