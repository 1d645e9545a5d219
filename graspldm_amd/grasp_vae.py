"""GraspCVAE (generation half): mirror of `grasp_ldm/models/grasp_vae.py` with the same
constructor arguments, sub-module names and state_dict keys (encoder.pc_encoder.*,
encoder.grasp_encoder.*, bottleneck.*, decoder.*).  The training-only parts
(grasp encoder, bottleneck, losses) are parameter containers so checkpoints load strictly."""
from typing import Union

import torch
from torch import nn

from .pc_encoders import PVCNN2Encoder, PVCNNEncoder
from .resnets import ResNet1D


def _get(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


class ConditionalGraspPoseDecoder(nn.Module):
    """grasp_vae.py:353-436: Linear(D->R) -> ResNet1D -> tmrp(6) / class_logits(1), one HIP launch."""
    MODELS = {"ResNet1D": ResNet1D}

    def __init__(self, config, in_features, feature_resolution, num_output_qualities=None):
        super().__init__()
        if _get(config, "type") not in self.MODELS:
            raise NotImplementedError(f"Base network arch of type=`{_get(config, 'type')}` is not implemented. "
                                      f"Available base network types are: {list(self.MODELS)}")
        self.in_features, self.feature_resolution = in_features, feature_resolution
        self.in_layer = nn.Linear(in_features, feature_resolution)
        self.net = self.MODELS[_get(config, "type")](dim=feature_resolution, **dict(_get(config, "args")))
        self.tmrp = nn.Linear(self.net.out_features, 6)
        self.class_logits = nn.Linear(self.net.out_features, 1)
        self._use_qualities = bool(num_output_qualities is not None and num_output_qualities > 0)
        if self._use_qualities:
            raise NotImplementedError("quality heads (num_output_qualities > 0) are not on the shipped hot path")
        self.num_qualities = None
        self.out_features = (6, 1)
        self._engine, self._key = None, None

    def _get_engine(self, device, rows):
        from ._cache import params_key, publish
        key = params_key(self.parameters(), device, rows)
        if self._engine is None or self._key != key:
            from .r1d import R1dEngine, pack_resnet1d
            sd = {k: v.detach().float().cpu() for k, v in self.state_dict().items()}
            packed = pack_resnet1d(sd, "net.", groups=self.net.groups, seq_len=self.feature_resolution,
                                   cond_rows=rows, decoder=dict(
                                       in_w=sd["in_layer.weight"], in_b=sd["in_layer.bias"],
                                       tmrp_w=sd["tmrp.weight"], tmrp_b=sd["tmrp.bias"],
                                       cls_w=sd["class_logits.weight"], cls_b=sd["class_logits.bias"]))
            self._engine, self._key = R1dEngine(packed, device), key
            publish(device)
        return self._engine

    @torch.no_grad()
    def forward(self, z_h, cond=None, samples_per_cond=1):
        if not z_h.is_cuda:
            raise RuntimeError("z_h must be a CUDA tensor (graspldm_amd has no CPU path)")
        eng = self._get_engine(z_h.device, 1 if cond.ndim == 2 else cond.shape[1])
        return eng.decode(z_h, eng.cond_embed(cond), samples_per_cond)


class ConditionalGraspPoseEncoder(nn.Module):
    """grasp_vae.py:439-536 (training-time: parameter container only)."""

    def __init__(self, config, latent_size, feature_resolution=16):
        super().__init__()
        args = dict(_get(config, "args"))
        self.in_features = args.pop("in_features")
        self.out_features = latent_size
        self.feature_resolution = feature_resolution
        self.in_layer = nn.Linear(self.in_features, feature_resolution)
        self.net = ResNet1D(dim=feature_resolution, **args)
        self.out_layer = nn.Linear(self.net.out_features, self.out_features)

    def forward(self, x, cond):
        raise NotImplementedError("grasp encoding is a training-time path (out of scope)")


class VAEBottleneck(nn.Module):
    def __init__(self, in_features, latent_size):
        super().__init__()
        self.mu = nn.Linear(in_features, latent_size)
        self.logvar = nn.Linear(in_features, latent_size)


class PcConditionedGraspEncoder(nn.Module):
    """grasp_vae.py:258-350"""
    PC_ENCODERS = {"PVCNNEncoder": PVCNNEncoder, "PVCNN2Encoder": PVCNN2Encoder}

    def __init__(self, pc_encoder_config, grasp_encoder_config, pc_latent_size=64, grasp_latent_size=4):
        super().__init__()
        t = _get(pc_encoder_config, "type")
        if t not in self.PC_ENCODERS:
            raise NotImplementedError(f"Pointcloud encoder network arch of type=`{t}` is not implemented. "
                                      f"Available base network types are: {list(self.PC_ENCODERS)}")
        self.pc_encoder = self.PC_ENCODERS[t](out_features=pc_latent_size, **dict(_get(pc_encoder_config, "args")))
        self.grasp_encoder = ConditionalGraspPoseEncoder(config=grasp_encoder_config, latent_size=grasp_latent_size)
        self.out_features = grasp_latent_size

    def encode_pc(self, xyz):
        return self.pc_encoder(xyz)

    def get_conditioning_latent(self, xyz):
        return self.encode_pc(xyz)


class GraspCVAE(nn.Module):
    """grasp_vae.py:17-255"""

    def __init__(self, grasp_latent_size: int, pc_latent_size: int, grasp_encoder_config: dict,
                 pc_encoder_config: dict, decoder_config: dict, loss_config: dict = None,
                 intermediate_feature_resolution: int = 16, num_output_qualities: Union[int, None] = None) -> None:
        super().__init__()
        self.grasp_latent_size, self.pc_latent_size = grasp_latent_size, pc_latent_size
        self.loss_config = loss_config  # accepted and ignored: losses are training-only (grasp_vae.py:55-69)
        self.encoder = PcConditionedGraspEncoder(pc_encoder_config=pc_encoder_config,
                                                 grasp_encoder_config=grasp_encoder_config,
                                                 pc_latent_size=pc_latent_size, grasp_latent_size=grasp_latent_size)
        self.bottleneck = VAEBottleneck(in_features=self.encoder.out_features, latent_size=grasp_latent_size)
        self.num_output_qualities = num_output_qualities
        self.decoder = ConditionalGraspPoseDecoder(in_features=grasp_latent_size, config=decoder_config,
                                                   num_output_qualities=num_output_qualities,
                                                   feature_resolution=intermediate_feature_resolution)
        self.out_features = self.decoder.out_features

    @property
    def use_grasp_qualities(self) -> bool:
        return bool(self.decoder._use_qualities)

    def encode_pc(self, xyz):
        return self.encoder.encode_pc(xyz)

    def sample_grasp_latent(self, batch_size, device):
        return torch.randn(batch_size, self.grasp_latent_size).to(device)

    @torch.no_grad()
    def generate_grasps(self, xyz, num_grasps=10, z_h=None):
        """grasp_vae.py:226-255: encode cloud -> N(0,I) latents (CPU generator, then moved) -> decode.
        The cloud latent is shared by index (sample i -> cloud i // num_grasps) instead of
        materialising repeat_interleave."""
        assert xyz.ndim == 3, (f"Input pointcloud should be  3-dim tensor of shape [B, N, 3]. "
                               f"Found a {xyz.ndim} dimensional tensor.")
        z_pc = self.encode_pc(xyz)
        if z_h is None:
            z_h = torch.randn(xyz.shape[0] * num_grasps, self.grasp_latent_size)
        return self.decoder(z_h.to(xyz.device), z_pc, samples_per_cond=num_grasps)

    def forward(self, *a, **k):
        raise NotImplementedError("VAE training forward is out of scope: graspldm_amd is the generation path")
