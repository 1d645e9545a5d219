"""Convenience constructors for the shipped `fpc` experiment with synthetic weights
(no checkpoints exist offline)."""
import torch

from .builder import build_model_from_cfg
from .synthetic import load_synthetic_weights


def fpc_model_config(n_points=1024, scheduler="ddim", latent=4, pc_latent=64, pc_channels=3, encoder="PVCNNEncoder",
                     encoder_scale=None):
    """configs/generation/fpc/fpc_1a_latentc3_z4_pc64_180k.py:25-153 as data.  encoder="PVCNN2Encoder": the same
    experiment conditioned by the SET-ABSTRACTION encoder family of the registry (pc_encoders.py:139-197 in its repaired
    form: PointNet++ set abstraction + PVConv + feature propagation, then the same head); encoder_scale =
    (scale_channels, scale_voxel_resolution), default the shipped (0.75, 0.75) / PVCNN2's own width (1, 1)."""
    rn = dict(block_channels=(32, 64, 128, 256), input_conditioning_dims=pc_latent, resnet_block_groups=4, dropout=0.1)
    if encoder == "PVCNNEncoder":
        sc, sv = encoder_scale or (0.75, 0.75)
        enc = dict(type="PVCNNEncoder", args=dict(
            in_features=3, n_points=n_points, scale_channels=sc, scale_voxel_resolution=sv,
            num_blocks=(1, 1, 1, 1), out_channels=pc_channels, use_global_attention=False))
    elif encoder == "PVCNN2Encoder":
        sc, sv = encoder_scale or (1, 1)
        enc = dict(type="PVCNN2Encoder", args=dict(
            in_features=3, n_points=n_points, scale_channels=sc, scale_voxel_resolution=sv, out_channels=pc_channels))
    else:
        raise ValueError(f"encoder must be PVCNNEncoder or PVCNN2Encoder, not {encoder!r}")
    vae = dict(model=dict(type="GraspCVAE", args=dict(
        grasp_latent_size=latent, pc_latent_size=pc_latent,
        pc_encoder_config=enc,
        grasp_encoder_config=dict(type="ResNet1D", args=dict(in_features=7, **rn)),
        decoder_config=dict(type="ResNet1D", args=dict(**rn)),
        loss_config=dict(reconstruction_loss=dict(type="GraspReconstructionLoss"), latent_loss=dict(type="VAELatentLoss")),
        num_output_qualities=0, intermediate_feature_resolution=16)))
    ddm = dict(model=dict(type="GraspLatentDDM", args=dict(
        model=dict(type="TimeConditionedResNet1D", args=dict(
            dim=latent, channels=1, is_time_conditioned=True, learned_variance=False, learned_sinusoidal_cond=False,
            random_fourier_features=True, **rn)),
        latent_in_features=latent, diffusion_timesteps=1000, noise_scheduler_type=scheduler, diffusion_loss="l2",
        beta_schedule="linear", is_conditioned=True, joint_training=False, denoising_loss_weight=1,
        variance_type="fixed_large", elucidated_diffusion=False, beta_start=0.00005, beta_end=0.001)))
    return dict(vae=vae, ddm=ddm)


def build_fpc_ldm(n_points=1024, scheduler="ddim", seed=0, device=None, **kw):
    cfg = fpc_model_config(n_points, scheduler, **kw)
    ldm = build_model_from_cfg(cfg["ddm"])
    ldm.set_vae_model(build_model_from_cfg(cfg["vae"]))
    load_synthetic_weights(ldm, seed=seed)
    ldm.eval()
    return ldm.to(device) if device is not None else ldm

