"""mmcv-free registries + config loader, enough for the shipped projects/configs/uni3detr/*.py to load unchanged
(ref: extra_tools/train.py:97-127; SURVEY.md §5 "Config / flag system", Appendix A9)."""
import copy
import os


class ConfigDict(dict):
    """dict with attribute access (configs are read as `conv_cfg.type`, ref: models/backbones/second_3d.py:46)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_config(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: to_config(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_config(v) for v in obj)
    return obj


class Registry:
    def __init__(self, name):
        self.name = name
        self._map = {}

    def register_module(self, name=None, module=None, force=False):
        def deco(cls):
            key = name or cls.__name__
            if key in self._map and not force and self._map[key] is not cls:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._map[key] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        if key not in self._map:
            raise KeyError(f"'{key}' is not registered in the {self.name} registry (known: {sorted(self._map)})")
        return self._map[key]

    def build(self, cfg, **default_args):
        if cfg is None:
            return None
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise TypeError(f"{self.name}: cfg must be a dict with a 'type' key, got {cfg!r}")
        args = to_config(copy.deepcopy(dict(cfg)))
        t = args.pop("type")
        for k, v in default_args.items():
            args.setdefault(k, v)
        cls = self.get(t) if isinstance(t, str) else t
        return cls(**args)

    def __contains__(self, key):
        return key in self._map


DETECTORS = Registry("detector")
MIDDLE_ENCODERS = Registry("middle_encoder")
VOXEL_ENCODERS = Registry("voxel_encoder")
BACKBONES = Registry("backbone")
NECKS = Registry("neck")
HEADS = Registry("head")
LOSSES = Registry("loss")
TRANSFORMER = Registry("transformer")
TRANSFORMER_LAYER_SEQUENCE = Registry("transformer_layer_sequence")
TRANSFORMER_LAYER = Registry("transformer_layer")
ATTENTION = Registry("attention")
FEEDFORWARD_NETWORK = Registry("ffn")
POSITIONAL_ENCODING = Registry("positional_encoding")
BBOX_ASSIGNERS = Registry("bbox_assigner")
BBOX_CODERS = Registry("bbox_coder")
MATCH_COST = Registry("match_cost")


def build_model(cfg, train_cfg=None, test_cfg=None):
    return DETECTORS.build(cfg, train_cfg=train_cfg, test_cfg=test_cfg)


# --------------------------------------------------------------------------------------------------
# Config.fromfile with `_base_` inheritance; base files missing from the tree (they live in mmdetection3d,
# SURVEY.md App. A9) are served from uni3detr_amd/configs/_base_/.
# --------------------------------------------------------------------------------------------------
_FALLBACK_BASE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", "_base_")


def _merge(base, new):
    out = copy.deepcopy(base)
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            v = copy.deepcopy(v)
            if isinstance(v, dict):
                v.pop("_delete_", None)
            out[k] = v
    return out


def _exec_file(path):
    g = {"__file__": path}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), g)
    return {k: v for k, v in g.items() if not k.startswith("__") and not callable(v) and not isinstance(v, type(os))}


def _resolve_base(cfg_path, rel):
    p = os.path.normpath(os.path.join(os.path.dirname(cfg_path), rel))
    if os.path.exists(p):
        return p
    marker = "_base_" + os.sep
    if marker in p:
        q = os.path.join(_FALLBACK_BASE, p.split(marker, 1)[1])
        if os.path.exists(q):
            return q
    raise FileNotFoundError(f"_base_ config {rel!r} of {cfg_path} not found (also not under {_FALLBACK_BASE})")


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        path = os.path.abspath(path)
        d = _exec_file(path)
        bases = d.pop("_base_", [])
        if isinstance(bases, str):
            bases = [bases]
        merged = {}
        for b in bases:
            merged = _merge(merged, dict(Config.fromfile(_resolve_base(path, b))))
        merged = _merge(merged, d)
        c = Config(to_config(merged))
        dict.__setitem__(c, "filename", path)
        return c

    def merge_from_dict(self, options):
        """`--cfg-options a.b=v` (ref: extra_tools/train.py:98-99)."""
        for key, v in options.items():
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, ConfigDict())
            node[parts[-1]] = to_config(v)
