"""Minimal reader for the reference's mmcv-style `.py` experiment configs
(grasp_ldm/utils/config.py:179-260: `Config.fromfile`): execute the file, keep the
public names, expose nested dicts with attribute access.  Configs are single-use in
the reference (the builder mutates them); here they are plain data and reusable."""
import os


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = _wrap(v)

    def copy(self):
        return _wrap({k: v for k, v in self.items()})


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, ConfigDict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        if not os.path.isfile(path):
            raise FileNotFoundError(path)
        ns = {"__file__": os.path.abspath(path)}
        with open(path) as f:
            exec(compile(f.read(), path, "exec"), ns)
        import types
        data = {k: v for k, v in ns.items() if not k.startswith("_") and not isinstance(v, types.ModuleType)
                and not callable(v)}
        cfg = Config(_wrap(data))
        dict.__setitem__(cfg, "filename", os.path.abspath(path))
        return cfg

    @staticmethod
    def fromdict(d):
        return Config(_wrap(d))
