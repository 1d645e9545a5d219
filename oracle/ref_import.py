"""Import the reference's Python (``/root/reference``) inside the BUILD CONTAINER
so that golden vectors can be captured from it.  TEST INFRASTRUCTURE, and
container-only: ``/root/reference`` does not exist on the GPU box, nothing that
runs there may call into this module.

Five third-party / native pieces the reference imports are absent from this
image; each gets the minimal stand-in that the generation path exercises
(SURVEY.md Appendix D):

  addict                 attr-dict ``Dict`` (utils/config.py:19, grasp_vae.py:6)
  yapf                   ``FormatCode`` no-op (utils/config.py:20, pretty_text only)
  trimesh                empty module (utils/gripper.py:3 via grasp_classifier.py:7)
  diffusers              oracle/schedulers.py  (PARITY UNPINNED third-party boundary)
  ...functional.backend  oracle/cpu_backend.py (the reference has no CPU kernels;
                         its own backend.py would JIT-compile CUDA sources)

Nothing else of the reference is replaced: models, modules, config loader,
builder and rotations run as shipped.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("GLDM_REFERENCE_ROOT", "/root/reference")
_BACKEND_MOD = "grasp_ldm.models.modules.ext.pvcnn.modules.functional.backend"


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "grasp_ldm"))


class _AttrDict(dict):
    """Minimal ``addict.Dict``: recursive dict->Dict conversion, attribute access,
    auto-vivifying ``__missing__`` (the reference's ConfigDict overrides it)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for a in args:
            if a is None:
                continue
            items = a.items() if isinstance(a, dict) else a
            for k, v in items:
                self[k] = v
        for k, v in kwargs.items():
            self[k] = v

    @classmethod
    def _hook(cls, v):
        if isinstance(v, dict) and not isinstance(v, _AttrDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._hook(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._hook(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        try:
            return self[k]
        except KeyError:
            return self.__missing__(k)

    def __missing__(self, k):
        v = type(self)()
        super().__setitem__(k, v)
        return v

    def __delattr__(self, k):
        del self[k]

    def copy(self):
        return type(self)(self)

    def to_dict(self):
        out = {}
        for k, v in self.items():
            out[k] = v.to_dict() if isinstance(v, _AttrDict) else v
        return out

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v


def install_shims():
    """Register the five stand-ins in ``sys.modules`` and put the reference on
    ``sys.path``.  Idempotent."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT} (container-only helper)")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    if "addict" not in sys.modules:
        m = types.ModuleType("addict")
        _AttrDict.__name__ = "Dict"
        m.Dict = _AttrDict
        sys.modules["addict"] = m

    if "yapf" not in sys.modules:
        y = types.ModuleType("yapf")
        yl = types.ModuleType("yapf.yapflib")
        ya = types.ModuleType("yapf.yapflib.yapf_api")
        ya.FormatCode = lambda text, **kw: (text, False)
        y.yapflib, yl.yapf_api = yl, ya
        sys.modules.update({"yapf": y, "yapf.yapflib": yl, "yapf.yapflib.yapf_api": ya})

    if "trimesh" not in sys.modules:
        t = types.ModuleType("trimesh")
        t.Trimesh = type("Trimesh", (), {})
        sys.modules["trimesh"] = t

    if "diffusers" not in sys.modules:
        from . import schedulers
        d = types.ModuleType("diffusers")
        d.DDIMScheduler = schedulers.DDIMScheduler
        d.DDPMScheduler = schedulers.DDPMScheduler
        sys.modules["diffusers"] = d

    if _BACKEND_MOD not in sys.modules:
        from . import cpu_backend
        b = types.ModuleType(_BACKEND_MOD)
        b._backend = cpu_backend._backend
        b.__all__ = ["_backend"]
        sys.modules[_BACKEND_MOD] = b


def load_reference_config(rel_path="configs/generation/fpc/fpc_1a_latentc3_z4_pc64_180k.py"):
    install_shims()
    from grasp_ldm.utils.config import Config
    return Config.fromfile(os.path.join(REFERENCE_ROOT, rel_path))


def build_reference_ldm(rel_path="configs/generation/fpc/fpc_1a_latentc3_z4_pc64_180k.py",
                        noise_scheduler_type=None, n_points=None):
    """Build the reference's GraspLatentDDM (+ GraspCVAE) from a shipped config.
    Configs are single-use objects (models/builder.py:57-93) -> reload per build."""
    install_shims()
    from grasp_ldm.models.builder import build_model_from_cfg
    cfg = load_reference_config(rel_path)
    if noise_scheduler_type is not None:
        cfg.model.ddm.model.args.noise_scheduler_type = noise_scheduler_type
    if n_points is not None:
        cfg.model.vae.model.args.pc_encoder_config.args.n_points = n_points
    ddm = build_model_from_cfg(cfg.model.ddm)
    vae = build_model_from_cfg(cfg.model.vae)
    ddm.set_vae_model(vae)
    return ddm.eval()


def load_reference_leaf(rel_path, name):
    """Load one reference source file by path (no package import, no shims)."""
    spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE_ROOT, rel_path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
