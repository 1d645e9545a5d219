"""Experiment / checkpoint reader for the generation path.

Mirrors what `tools/inference.py` does to get weights into a model (Experiment :97-158,
InferenceLDM.load_model :514-566, InferenceVAE.load_model :715-747, fix_state_dict_prefix
grasp_ldm/utils/torch_utils.py:4-37) and hardens it for files written elsewhere:

  * Lightning `.ckpt` files pickle more than tensors (hyper-parameters holding the reference's
    `Config`, callbacks, optimiser states): none of those classes exist here.  `load_checkpoint` unpickles
    with an EXACT (module, name) allow-list (torch's own weights-only set: tensor / storage rebuild helpers, dtypes,
    OrderedDict; numpy's array reconstruction helpers; plain builtin containers) that never resolves a dotted name
    (protocol-4 `("torch", "os.system")` walks attributes); every other global -- importable or not -- becomes an inert
    placeholder, so the tensors are still read and the loader resolves no foreign callable; a bare state dict (no
    "state_dict" wrapper) is accepted too.
  * weights live under `model.` and, when an EMA copy was kept, `ema_model.online_model.` (the string the
    reference loads for use_ema_model=True, :521; kept verbatim).  Every other key (`ema_model.ema_model.*`,
    `ema_model.initted`, `ema_model.step`, loss buffers) is ignored, exactly like ignore_all_others=True.
  * old experiment configs keep the model section under `models`, new ones under `model` (:717).
"""
import glob
import os
import pickle
import warnings

import torch


class _Placeholder:
    """Stands in for any pickled class that is not importable here (Lightning / reference objects)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__["_state"] = state

    def __call__(self, *a, **k):
        return _Placeholder()

    # pickle rebuilds dict / list / set subclasses through these
    def __setitem__(self, k, v):
        pass

    def append(self, v):
        pass

    def extend(self, v):
        pass

    def add(self, v):
        pass

    def update(self, *a, **k):
        pass

    def __reduce_ex__(self, protocol):
        return (_Placeholder, ())


# Globals a Lightning checkpoint legitimately needs to rebuild its TENSORS and plain containers: an EXACT
# (module, name) allow-list -- torch's own weights-only set (storages, dtypes, the _rebuild_* helpers, OrderedDict,
# _codecs.encode) plus numpy's array reconstruction helpers and the plain builtin containers.  No module is allowed
# wholesale and no dotted name is ever resolved: with pickle protocol >= 4 `find_class("torch", "os.system")` walks
# attributes, so a prefix test on the module string would hand out `os.system` through any module that imports `os`.
# Everything else (hyper-parameter objects, callbacks, anything a crafted file might name) becomes an inert placeholder,
# whether or not it is importable here: the fallback loader never executes foreign code.
_SAFE_BUILTINS = ("dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray",
                  "complex", "slice", "range", "object")
_NUMPY_MODULES = ("numpy", "numpy.core.multiarray", "numpy._core.multiarray", "numpy.core.numeric", "numpy._core.numeric",
                  "numpy.dtypes")
_NUMPY_NAMES = ("ndarray", "dtype", "_reconstruct", "scalar", "float32", "float64", "int64", "int32", "bool_", "uint8",
                "int8", "int16", "float16", "Float32DType", "Float64DType", "Int64DType", "Int32DType")


def _static_torch_globals():
    """What a tensor-only state dict needs from torch, by exact name (the fallback for _get_allowed_globals)."""
    import collections
    import _codecs
    import torch._utils as tu
    out = {"collections.OrderedDict": collections.OrderedDict, "_codecs.encode": _codecs.encode,
           "torch.Size": torch.Size, "torch.device": torch.device, "torch.Tensor": torch.Tensor,
           "torch.nn.parameter.Parameter": torch.nn.Parameter, "torch.storage.UntypedStorage": torch.UntypedStorage,
           "torch.storage.TypedStorage": torch.storage.TypedStorage}
    for n in ("_rebuild_tensor_v2", "_rebuild_tensor", "_rebuild_parameter", "_rebuild_parameter_with_state",
              "_rebuild_qtensor", "_rebuild_device_tensor_from_numpy"):
        if hasattr(tu, n):
            out[f"torch._utils.{n}"] = getattr(tu, n)
    for n in ("FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage",
              "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage"):
        if hasattr(torch, n):
            out[f"torch.{n}"] = getattr(torch, n)
    for n in ("float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool"):
        out[f"torch.{n}"] = getattr(torch, n)
    return out


def _allowed_globals():
    """{"module.name": object} -- built once; the objects themselves, so nothing is looked up by attribute walk."""
    import importlib
    try:   # torch's own weights-only set; a private name, so a pinned static list stands in if a release moves it
        from torch._weights_only_unpickler import _get_allowed_globals
        allowed = dict(_get_allowed_globals())
    except Exception:  # noqa: BLE001
        allowed = _static_torch_globals()
    import builtins
    for n in _SAFE_BUILTINS:
        allowed[f"builtins.{n}"] = getattr(builtins, n)
    for m in _NUMPY_MODULES:
        try:
            mod = importlib.import_module(m)
        except Exception:
            continue
        for n in _NUMPY_NAMES:
            obj = mod.__dict__.get(n)
            if obj is not None:
                allowed[f"{m}.{n}"] = obj
    return allowed


_ALLOWED = None


def _is_safe_global(module, name):
    global _ALLOWED
    if "." in name:  # STACK_GLOBAL with a dotted name = an attribute walk: never
        return False
    if _ALLOWED is None:
        _ALLOWED = _allowed_globals()
    return f"{module}.{name}" in _ALLOWED


class _TolerantUnpickler(pickle._Unpickler):
    """The pure-Python unpickler (its opcode handlers can be overridden; the non-tensor part of a checkpoint is small).
    BUILD with slot state on a CLASS object sets class attributes process-wide -- on allow-listed torch classes that
    corrupts them for the rest of the process -- so BUILD is only ever applied to instances."""

    def find_class(self, module, name):
        if _is_safe_global(module, name):
            return _ALLOWED[f"{module}.{name}"]
        return type(name.rpartition(".")[2] or "Foreign", (_Placeholder,), {"__module__": module})

    def load_build(self):
        target = self.stack[-2]
        if isinstance(target, type) or callable(target) and not isinstance(target, _Placeholder):
            raise pickle.UnpicklingError(f"BUILD on a class or function object ({getattr(target, '__name__', target)!r}): refused")
        pickle._Unpickler.load_build(self)

    dispatch = dict(pickle._Unpickler.dispatch)
    dispatch[pickle.BUILD[0]] = load_build


class _TolerantPickle:
    """`pickle_module` for torch.load: stock pickle, except unknown classes become placeholders."""
    __name__ = "graspldm_amd_tolerant_pickle"
    Unpickler = _TolerantUnpickler
    load = staticmethod(lambda f, **kw: _TolerantUnpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    Pickler = pickle.Pickler
    PickleError = pickle.PickleError
    UnpicklingError = pickle.UnpicklingError
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    DEFAULT_PROTOCOL = pickle.DEFAULT_PROTOCOL


def load_checkpoint(path):
    """-> flat {key: tensor} of everything tensor-valued under the checkpoint's state dict (CPU)."""
    if not os.path.isfile(path):
        raise FileNotFoundError(f"Could not find any checkpoint in ckpt path: {path}")
    try:
        raw = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:  # Lightning checkpoints carry non-tensor objects (hyper-parameters, callbacks)
        warnings.warn(f"{os.path.basename(path)}: not loadable with weights_only=True ({type(e).__name__}); reading it with "
                      "the allow-listed unpickler (tensors and plain containers only, every other object becomes a "
                      "placeholder)")
        raw = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_TolerantPickle)
    sd = raw["state_dict"] if isinstance(raw, dict) and "state_dict" in raw else raw
    if not isinstance(sd, dict) or not any(isinstance(v, torch.Tensor) for v in sd.values()):
        raise RuntimeError(f"{path}: no state dict found (keys: {list(raw)[:8] if isinstance(raw, dict) else type(raw)})")
    return {k: v for k, v in sd.items() if isinstance(v, torch.Tensor)}


def fix_state_dict_prefix(state_dict, prefix="model", ignore_all_others=False):
    """grasp_ldm/utils/torch_utils.py:4-37: strip `<prefix>.`; optionally drop every other key."""
    items = [(k, v) for k, v in state_dict.items() if not ignore_all_others or k.startswith(prefix)]
    return type(state_dict)((k.partition(f"{prefix}.")[2], v) for k, v in items)


def select_weights(state_dict, use_ema_model):
    """The sub-dict the reference hands to load_state_dict (tools/inference.py:520-524).  A state dict that
    already has bare module keys (saved with model.state_dict()) passes through."""
    prefix = "model" if not use_ema_model else "ema_model.online_model"
    out = fix_state_dict_prefix(state_dict, prefix, ignore_all_others=True)
    if not out and not any(k.startswith(("model.", "ema_model.")) for k in state_dict):
        return dict(state_dict)
    return out


def load_weights(model, ckpt_path, use_ema_model):
    """strict load with the reference's error text (tools/inference.py:544-564)."""
    sd = select_weights(load_checkpoint(ckpt_path), use_ema_model)
    try:
        missing, unexpected = model.load_state_dict(sd, strict=True)
        if missing:
            warnings.warn(f"Missing keys while loading state dict: {missing}")
        if unexpected:
            warnings.warn(f"Found unexpected keys while loading state dict: {unexpected}")
    except Exception as e:
        msg = "Error while loading state dict: You might be using an incompatible state dict. \n"
        if use_ema_model:
            msg += ("EMA model is requested but may not be available. Check and set the `use_ema_model` flag "
                    "appropriately. \n")
        raise RuntimeError(msg + f"Error: {e}")
    return model


def model_section(config):
    """`model` (new configs) or `models` (old ones): tools/inference.py:717."""
    for key in ("model", "models"):
        if key in config:
            return key
    raise KeyError("experiment config has neither a `model` nor a `models` section")


class Experiment:
    """Experiment directory: {root}/{name}/{mode}/<config>.py + {mode}/checkpoints/last.ckpt
    (tools/inference.py:97-158).  A manually given checkpoint path wins when the file exists."""

    MODES = ("vae", "ddm", "elucidated_ddm")

    def __init__(self, exp_name, exp_out_root="output", modes=("vae", "ddm", "elucidated_ddm"), vae_ckpt_path=None,
                 ddm_ckpt_path=None, elucidated_ckpt_path=None):
        self.exp_name = exp_name
        self.exp_dir = os.path.join(exp_out_root, exp_name)
        self._modes = list(modes)
        if not os.path.isdir(self.exp_dir):
            raise FileNotFoundError(f"No experiment directory `{exp_name}` found in `{exp_out_root}/`")
        self._config_paths = {m: sorted(glob.glob(f"{self.exp_dir}/{m}/*.py")) for m in self._modes}
        manual = dict(vae=vae_ckpt_path, ddm=ddm_ckpt_path, elucidated_ddm=elucidated_ckpt_path)
        self._ckpt_paths = {}
        for m in self._modes:
            enforce = manual.get(m)
            path = enforce if enforce is not None and os.path.isfile(enforce) else f"{self.exp_dir}/{m}/checkpoints/last.ckpt"
            if not os.path.isfile(path):
                raise FileNotFoundError(f"For given mode ({m}) in `modes`:Could not find any checkpoint in ckpt path: {path}")
            self._ckpt_paths[m] = path

    def get_config(self, mode):
        from .config import Config
        assert mode in self._modes, f"Could not find mode ({mode}) in experiment modes "
        if not self._config_paths[mode]:
            raise FileNotFoundError(f"no config (*.py) under {self.exp_dir}/{mode}/")
        return Config.fromfile(self._config_paths[mode][0])

    def get_ckpt_path(self, mode):
        return self._ckpt_paths[mode]


def load_ldm_from_experiment(exp_name, exp_out_root, use_ema_model=True, ddm_ckpt_path=None, use_fast_sampler=True,
                             mode="ddm"):
    """InferenceLDM.__init__ + load_model of the reference (tools/inference.py:401-566) up to the weights:
    -> (GraspLatentDDM with its VAE attached, eval mode, on the CPU; config; Experiment).  use_fast_sampler
    switches the scheduler to DDIM before the model is built (:463-471; the reference patches `config.models`,
    which the shipped `model` configs do not have -- here the section that exists is patched)."""
    from .builder import build_model_from_cfg
    exp = Experiment(exp_name, exp_out_root, modes=[mode], **{("ddm_ckpt_path" if mode == "ddm" else "elucidated_ckpt_path"): ddm_ckpt_path})
    config = exp.get_config(mode)
    key = model_section(config)
    if use_fast_sampler:
        config[key]["ddm"]["model"]["args"]["noise_scheduler_type"] = "ddim"
    model = build_model_from_cfg(config[key]["ddm"])
    model.set_vae_model(build_model_from_cfg(config[key]["vae"]))
    return load_weights(model, exp.get_ckpt_path(mode), use_ema_model).eval(), config, exp


def load_vae_from_experiment(exp_name, exp_out_root, use_ema_model=True, vae_ckpt_path=None):
    """InferenceVAE.__init__ + load_model (tools/inference.py:669-747) up to the weights (CPU, eval)."""
    from .builder import build_model_from_cfg
    exp = Experiment(exp_name, exp_out_root, modes=["vae"], vae_ckpt_path=vae_ckpt_path)
    config = exp.get_config("vae")
    model = build_model_from_cfg(config[model_section(config)]["vae"])
    return load_weights(model, exp.get_ckpt_path("vae"), use_ema_model).eval(), config, exp
