"""String -> class registry and recursive config builder: mirror of
`grasp_ldm/models/builder.py:28-116` for the models on the generation path."""
from torch import nn

from .config import ConfigDict, _wrap
from .diffusion import GaussianDiffusion1D
from .grasp_ldm import GraspLatentDDM
from .grasp_vae import GraspCVAE
from .resnets import ClassTimeConditionedResNet1D, ResNet1D, TimeConditionedResNet1D

DIFFUSION_MODELS = {"GaussianDiffusion1D": GaussianDiffusion1D, "TimeConditionedResNet1D": TimeConditionedResNet1D,
                    "ClassTimeConditionedResNet1D": ClassTimeConditionedResNet1D}
STANDARD_MODULES = {"ResNet1D": ResNet1D}
ALL_MODELS = {"GraspCVAE": GraspCVAE, "GraspLatentDDM": GraspLatentDDM, **STANDARD_MODULES, **DIFFUSION_MODELS}


def build_model(model_cfg) -> nn.Module:
    if model_cfg["type"] not in ALL_MODELS:
        raise KeyError(f"`{model_cfg['type']}` in the model_registry. \n Supported models are: {list(ALL_MODELS)}")
    return ALL_MODELS[model_cfg["type"]](**model_cfg["args"])


def _build_recursive(cfg):
    """Every value under a `model` key that is a {type, args} dict becomes the built module
    (builder.py:57-93); works on a copy, so a loaded config can be built repeatedly."""
    if not isinstance(cfg, dict):
        return cfg
    out = ConfigDict()
    for k, v in cfg.items():
        if k == "args" and isinstance(v, dict):
            out[k] = _build_recursive(v)
        elif k == "model" and isinstance(v, dict):
            out[k] = build_model(_build_recursive(v))
        else:
            out[k] = v
    return out


def build_model_from_cfg(model_cfg) -> nn.Module:
    built = _build_recursive(_wrap(model_cfg))
    return built["model"] if "model" in built else built
