"""Convenience constructors for the shipped `fpc` experiment with synthetic weights
(no checkpoints exist offline)."""
import torch

from .builder import build_model_from_cfg
from .synthetic import load_synthetic_weights


def fpc_model_config(n_points=1024, scheduler="ddim", latent=4, pc_latent=64, pc_channels=3):
    """configs/generation/fpc/fpc_1a_latentc3_z4_pc64_180k.py:25-153 as data."""
    rn = dict(block_channels=(32, 64, 128, 256), input_conditioning_dims=pc_latent, resnet_block_groups=4, dropout=0.1)
    vae = dict(model=dict(type="GraspCVAE", args=dict(
        grasp_latent_size=latent, pc_latent_size=pc_latent,
        pc_encoder_config=dict(type="PVCNNEncoder", args=dict(
            in_features=3, n_points=n_points, scale_channels=0.75, scale_voxel_resolution=0.75,
            num_blocks=(1, 1, 1, 1), out_channels=pc_channels, use_global_attention=False)),
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

