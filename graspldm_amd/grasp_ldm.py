"""GraspLatentDDM (generation half): mirror of `grasp_ldm/models/grasp_ldm.py:10-246`."""
import torch
from torch import nn

from .diffusion import GaussianDiffusion1D


class GraspLatentDDM(nn.Module):
    def __init__(self, model, latent_in_features, diffusion_timesteps, diffusion_loss, beta_schedule="linear",
                 noise_scheduler_type: str = "ddpm", is_conditioned=True, joint_training=False,
                 denoising_loss_weight=1, variance_type="fixed_small", elucidated_diffusion=False,
                 beta_start=5e-5, beta_end=5e-2) -> None:
        super().__init__()
        self.vae_model = None
        self.is_elucidated_diffusion = elucidated_diffusion
        if elucidated_diffusion:  # grasp_ldm.py:58-62
            from .elucidated import ElucidatedDiffusion
            self.diffusion_model = ElucidatedDiffusion(net=model, seq_length=latent_in_features)
        else:
            self.diffusion_model = GaussianDiffusion1D(
                model=model, n_dims=latent_in_features, num_steps=diffusion_timesteps, loss_type=diffusion_loss,
                beta_schedule=beta_schedule, beta_start=beta_start, beta_end=beta_end,
                noise_scheduler_type=noise_scheduler_type, variance_type=variance_type)
        self.is_conditioned, self.joint_training, self.loss_weight = is_conditioned, joint_training, denoising_loss_weight
        self.is_vae_frozen = False

    @property
    def use_grasp_qualities(self):
        return self.vae_model.use_grasp_qualities

    @property
    def scheduler_type(self):
        return self.diffusion_model._noise_scheduler_type

    def set_vae_model(self, vae_model):
        self.vae_model = vae_model

    def load_vae_weights(self, state_dict):
        self.vae_model.load_state_dict(state_dict, strict=True)

    def set_inference_timesteps(self, num_inference_steps):
        if self.is_elucidated_diffusion:
            self.diffusion_model.num_sample_steps = int(num_inference_steps)
        else:
            self.diffusion_model.set_inference_timesteps(num_inference_steps)

    @torch.no_grad()
    def generate_grasps(self, xyz, num_grasps=10, return_intermediate=False, **kwargs):
        """grasp_ldm.py:189-233: encode -> (repeat per grasp) -> reverse diffusion -> decode.
        Returns ((tmrp [B*G,6], cls_logit [B*G,1]), intermediates)."""
        z_pc = self.vae_model.encode_pc(xyz)
        n = z_pc.shape[0] * num_grasps
        # `metas` travels through **kwargs down to the denoiser call in the reference; TimeConditionedResNet1D
        # ignores it (resnets.py:565), the class-conditioned one reads metas["mode_cls"] (class_conditioned_resnet.py:73)
        denoiser = self.diffusion_model.net if self.is_elucidated_diffusion else self.diffusion_model.model
        if self.is_elucidated_diffusion or not hasattr(denoiser, "class_embedding"):
            kwargs.pop("metas", None)  # (the reference forwards it into sample_using_dpmpp, which raises TypeError)
        kwargs.setdefault("device", xyz.device)
        if self.is_elucidated_diffusion:
            kwargs.pop("x_T", None)
        out, all_outs = self.diffusion_model.sample(z_cond=z_pc, batch_size=n, return_all=return_intermediate,
                                                    samples_per_cond=num_grasps, **kwargs)
        final = self.vae_model.decoder(out.squeeze(-2), z_pc, samples_per_cond=num_grasps)
        if not return_intermediate:
            return final, []
        steps = []
        for idx in torch.linspace(0, len(all_outs) - 1, steps=50, dtype=torch.int):
            o = self.vae_model.decoder(all_outs[int(idx)].squeeze(-2), z_pc, samples_per_cond=num_grasps)
            steps.append([t.detach().cpu() for t in o])
        return final, steps

    def check_engines(self):
        """Raise GldmError if any fused denoise launch so far lost a step-segment hand-off between workgroups (its latents
        are NaN then): the synchronising form of the check every launch makes without a host sync.  The inference harness
        calls the same check before results leave the device; direct callers of generate_grasps call this."""
        denoiser = self.diffusion_model.net if self.is_elucidated_diffusion else self.diffusion_model.model
        eng = getattr(denoiser, "_engine", None)
        if eng is not None:
            eng.check()

    def forward(self, *a, **k):
        raise NotImplementedError("LDM training forward is out of scope: graspldm_amd is the generation path")
