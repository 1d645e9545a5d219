"""Inference harness: mirror of `tools/inference.py` (Experiment :97-158, InferenceLDM
:401-666, InferenceVAE :669-815, unnormalize_* :31-94) for the generation path.

Differences, all additive or documented:
  * a model can be injected directly (`model=`) so the path runs without an experiment
    directory (no ACRONYM data / checkpoints exist offline); datasets are out of scope,
    inputs follow the dataset item contract produced by graspldm_amd.synthetic;
  * `num_inference_steps` is honoured (the reference's CLI ignores `--inference_steps`
    because it passes use_fast_sampler=False: tools/generate_grasps.py:69-79);
  * unnormalise + tmrp_to_H + sigmoid are one HIP launch (gldm_pose_epilogue).
"""
import glob
import os
import warnings
from enum import Enum

import torch

from .builder import build_model_from_cfg
from .checkpoint import Experiment, fix_state_dict_prefix, load_weights as _load_weights, model_section  # noqa: F401
from .r1d import pose_epilogue

PC_STD, MRP_STD = 0.05, 0.5


class Conditioning(Enum):
    UNCONDITIONAL = "NORMAL"
    CLASS_CONDITIONED = "CLASS_CONDITIONED"
    REGION_CONDITIONED = "REGION_CONDITIONED"


class ModelType(Enum):
    LDM = "LDM"
    VAE = "VAE"


def unnormalize_pc(pc, metas):
    if pc.ndim == 2:
        return pc * metas["pc_std"].to(pc.device) + metas["pc_mean"].to(pc.device)
    return pc * metas["pc_std"].unsqueeze(-2).to(pc.device) + metas["pc_mean"].unsqueeze(-2).to(pc.device)


class _InferenceBase:
    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("graspldm_amd runs on the GPU only (no CPU path)")
        self.model = None
        self.dataset = None

    def _results(self, pc, metas, tmrp, cls_logit, num_pcs, num_grasps, all_steps=()):
        # before anything leaves the device: a lost hand-off inside the fused sampling launch raises here
        # (status, not silent garbage: the reference exits the process on a CUDA error, cuda_utils.cuh:28-37)
        for m in self.model.modules():
            eng = getattr(m, "_engine", None)
            if eng is not None:
                eng.check()
        if not bool(torch.isfinite(pc).all()):
            # (ReLU as max(x, 0) drops a NaN where torch.relu keeps it: a NaN point would come out as plausible-looking grasps)
            from ._lib import GldmError
            raise GldmError("the input point cloud holds non-finite coordinates")
        metas = {k: (v.to(self.device) if isinstance(v, torch.Tensor) else v) for k, v in metas.items()}
        # mean [B,6] and std [1,6] (normalize_input and the dataset both build them so) broadcast independently
        mean, std = metas["grasp_mean"], metas["grasp_std"]
        H, un, conf = pose_epilogue(tmrp, cls_logit, mean, std, num_grasps)
        # ... and so does a pose that is not a number: the GEMMs multiply f16 pieces -- operands whose size the
        # data sets carry range scales, and anything that still leaves the range must not reach the caller as a grasp
        if not bool(torch.isfinite(H).all()) or (conf is not None and not bool(torch.isfinite(conf).all())):
            from ._lib import GldmError
            bad = int((~torch.isfinite(H.view(-1, 16)).all(dim=1)).sum())
            raise GldmError(f"{bad} of {H.numel() // 16} generated poses are not finite (input cloud / metas out of range, "
                            "or weights whose activations leave the f16 range: rerun under graspldm_amd.numerics.f32_only())")
        steps_H = []
        if all_steps:
            if num_pcs > 1:  # tools/inference.py:631-634
                raise NotImplementedError("Batched grasps for all diffusion steps are not implemented")
            for step in all_steps:  # [tmrp [G,6], logit [G,1]] on the CPU (grasp_ldm.py:223-227)
                Hs, _, _ = pose_epilogue(step[0].to(self.device), None, mean, std, num_grasps)
                steps_H.append(Hs.view(1, num_grasps, 4, 4).cpu())
        return dict(grasps=H.view(num_pcs, num_grasps, 4, 4), grasp_tmrp=un.view(num_pcs, num_grasps, 6),
                    confidence=conf.view(num_pcs, num_grasps, 1), qualities=None, pc=unnormalize_pc(pc, metas),
                    all_steps_grasps=steps_H)

    def set_normalization_params(self, norm_config):
        """grasp_ldm/inference/inference_base.py:103-130: pc_shift, grasp_shift, translation_scale, rotation_scale."""
        get = (lambda k: norm_config[k]) if isinstance(norm_config, dict) else (lambda k: getattr(norm_config, k))
        for k in ("pc_shift", "grasp_shift", "translation_scale", "rotation_scale"):
            try:
                get(k)
            except (KeyError, AttributeError):
                raise AssertionError(f"norm_config should have `{k}`")
        self._norm = dict(pc_shift=get("pc_shift"), grasp_shift=get("grasp_shift"),
                          pc_scale=get("translation_scale"), mrp_scale=get("rotation_scale"))

    def normalize_input(self, pc):
        """tools/inference.py:570-591 / inference_base.py:181-212: centre on the mean, divide by the translation
        scale (0.05 unless set_normalization_params says otherwise), build metas.  One HIP launch."""
        from .pointcloud import normalize_input
        kw = getattr(self, "_norm", None) or dict(pc_shift=0.0, pc_scale=PC_STD, mrp_scale=MRP_STD, grasp_shift=None)
        return normalize_input(pc.to(self.device), **kw)

    def infer_on_pointcloud(self, pc, num_grasps=10, return_intermediate=False, num_points=None,
                            use_farthest_point=True):
        """tools/inference.py:658-666 (= generate_on_pointcloud, grasp_ldm/inference/inference_base.py:161-179).
        `num_points` (additive): first bring every cloud to the encoder's point count
        (PointCloudHelpers.regularize_pc_point_count; farthest-point selection by default)."""
        pc = pc.to(self.device)
        if num_points is not None and pc.shape[-2] != num_points:
            from .pointcloud import PointCloudHelpers
            clouds = pc.unsqueeze(0) if pc.ndim == 2 else pc
            reg = torch.stack([PointCloudHelpers.regularize_pc_point_count(c, num_points, use_farthest_point)
                               for c in clouds])
            pc = reg[0] if pc.ndim == 2 else reg
        pcn, metas = self.normalize_input(pc)
        return self.generate_grasps(pcn, metas, num_grasps=num_grasps, return_intermediate=return_intermediate)

    generate_on_pointcloud = infer_on_pointcloud

    def generate_class_conditioned_grasps(self, pc, num_grasps=10, metas=None, data_idx=None, class_label=0, **kwargs):
        """tools/inference.py:330-364: the label, repeated per grasp, travels in metas["mode_cls"] to the
        class-conditioned denoiser (ClassTimeConditionedResNet1D).  One cloud, like the reference."""
        metas = dict(metas)
        metas["mode_cls"] = torch.LongTensor([class_label]).unsqueeze(0).repeat((num_grasps, 1)).to(
            self.device, dtype=torch.float32)
        return self.generate_grasps(pc, metas=metas, num_grasps=num_grasps, **kwargs)

    def infer(self, data_idx=None, num_grasps=10, visualize=False, condition_type=Conditioning.UNCONDITIONAL,
              conditioning=None, **kwargs):
        if self.dataset is None:
            raise RuntimeError("no dataset attached: ACRONYM loading is out of scope; call generate_grasps(pc, metas) "
                               "with a dataset-contract item (graspldm_amd.synthetic.normalize_cloud)")
        if condition_type == Conditioning.REGION_CONDITIONED:
            raise NotImplementedError("region conditioned denoisers are not shipped (out of scope)")
        item = self.dataset[data_idx if data_idx is not None else 0]
        if condition_type == Conditioning.CLASS_CONDITIONED:
            res = self.generate_class_conditioned_grasps(item["pc"], num_grasps=num_grasps, metas=item["metas"],
                                                         class_label=conditioning, **kwargs)
        else:
            res = self.generate_grasps(item["pc"], item["metas"], num_grasps=num_grasps, **kwargs)
        res["inputs"] = dict(item)
        return res


class InferenceLDM(_InferenceBase):
    def __init__(self, exp_name=None, exp_out_root=None, data_root=None, data_split="test", use_ema_model=True,
                 ddm_ckpt_path=None, vae_ckpt_path=None, elucidated_ckpt_path=None, use_elucidated=False,
                 use_fast_sampler=True, num_inference_steps=None, augment_pc=False, load_dataset=False,
                 device="cuda:0", model=None):
        super().__init__(device)
        self.use_ema_model = use_ema_model
        self.ddm_mode = "ddm" if not use_elucidated else "elucidated_ddm"
        # _setup_ldm_sampler, tools/inference.py:463-490
        if use_fast_sampler:
            self.fast_sampler = "DDIM" if not use_elucidated else "DPMPP"
            default_steps = 100 if not use_elucidated else 32
            self.num_inference_steps = default_steps if num_inference_steps is None else num_inference_steps
        else:
            # elucidated + no fast sampler: ElucidatedDiffusion.sample(use_dpmpp=False) = the stochastic Heun sampler
            # (elucidated_diffusion.py:177-257), one network launch per evaluation
            self.fast_sampler, self.num_inference_steps = ("HEUN" if use_elucidated else None), num_inference_steps
        if model is not None:
            self.model = model.to(self.device).eval()
        else:
            from .checkpoint import load_ldm_from_experiment
            m, self.config, self.experiment = load_ldm_from_experiment(
                exp_name, exp_out_root, use_ema_model, ddm_ckpt_path if not use_elucidated else elucidated_ckpt_path,
                use_fast_sampler and not use_elucidated, mode=self.ddm_mode)
            self.model = m.to(self.device).eval()
        if load_dataset:
            warnings.warn("ACRONYM dataset loading is out of scope; use generate_grasps(pc, metas)")

    @torch.no_grad()
    def generate_grasps(self, pc, metas, num_grasps=10, return_intermediate=False, x_T=None, **kwargs):
        batch = (pc.unsqueeze(0) if pc.ndim == 2 else pc).to(self.device)
        extra_sampler = {}
        if self.fast_sampler == "DPMPP":  # tools/inference.py:607-609
            extra_sampler = dict(use_dpmpp=True, num_sample_steps=self.num_inference_steps)
            if "noise" in kwargs:
                extra_sampler["noise"] = kwargs["noise"]
        elif self.fast_sampler == "HEUN":
            extra_sampler = dict(use_dpmpp=False)
            if self.num_inference_steps is not None:
                extra_sampler["num_sample_steps"] = self.num_inference_steps
            extra_sampler.update({k: kwargs[k] for k in ("noise",) if k in kwargs})
        elif self.num_inference_steps is not None:
            self.model.set_inference_timesteps(self.num_inference_steps)
        if return_intermediate and batch.shape[0] > 1:  # the reference raises after sampling; fail before the work
            raise NotImplementedError("Batched grasps for all diffusion steps are not implemented")
        # noise_source="kernel" (+ noise_seed / noise_base): DDPM step noise drawn inside the fused launch (diffusion.py)
        extra = {k: kwargs[k] for k in ("step_noise", "cls_cond", "noise_source", "noise_seed", "noise_base") if k in kwargs}
        extra.update(extra_sampler)
        denoiser = getattr(self.model.diffusion_model, "model", None) or getattr(self.model.diffusion_model, "net")
        if hasattr(denoiser, "class_embedding"):
            extra["metas"] = {k: (v.to(self.device) if isinstance(v, torch.Tensor) else v) for k, v in metas.items()}
        (tmrp, logit), steps = self.model.generate_grasps(batch, num_grasps=num_grasps,
                                                          return_intermediate=return_intermediate, x_T=x_T, **extra)
        return self._results(batch, metas, tmrp, logit, batch.shape[0], num_grasps, all_steps=steps)


class InferenceVAE(_InferenceBase):
    def __init__(self, exp_name=None, exp_out_root=None, use_ema_model=True, data_root=None, data_split="test",
                 ddm_ckpt_path=None, vae_ckpt_path=None, augment_pc=False, load_dataset=False, device="cuda:0",
                 model=None):
        super().__init__(device)
        self.use_ema_model = use_ema_model
        if model is not None:
            self.model = model.to(self.device).eval()
        else:
            from .checkpoint import load_vae_from_experiment
            m, self.config, self.experiment = load_vae_from_experiment(exp_name, exp_out_root, use_ema_model,
                                                                        vae_ckpt_path)
            self.model = m.to(self.device).eval()

    @torch.no_grad()
    def generate_grasps(self, pc, metas, num_grasps=10, z_h=None, **kwargs):
        batch = (pc.unsqueeze(0) if pc.ndim == 2 else pc).to(self.device)
        tmrp, logit = self.model.generate_grasps(batch, num_grasps, z_h=z_h)
        out = self._results(batch, metas, tmrp, logit, batch.shape[0], num_grasps)
        out.pop("all_steps_grasps")
        return out
