"""Registry / builder / config surface of the drop-in boundary.

The reference builds the hot path through mmcv's registry: ``BACKBONES.build(dict(type='ViT_CLIP',
...))`` from ``mmaction/models/builder.py:8-14,37-52`` with configs read by ``mmcv.Config.fromfile``
(``tools/train.py:81-83``).  mmcv is not installed in this image, so a minimal compatible
``Registry`` / ``build_from_cfg`` / ``Config`` lives here; when mmcv / mmaction ARE importable,
``register_into_mmaction()`` additionally registers the classes into the real registries
(``force=True``) so the unmodified ``configs/recognition/vit/*.py`` pick up this implementation.
"""
import ast
import copy
import os
from typing import Any, Dict, Optional


class Registry:
    """Name -> class map with mmcv's ``register_module`` / ``build`` / ``get`` semantics."""

    def __init__(self, name: str):
        self._name = name
        self._module_dict: Dict[str, type] = {}

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return f"Registry(name={self._name}, items={sorted(self._module_dict)})"

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key: str) -> Optional[type]:
        return self._module_dict.get(key)

    def _register(self, cls, name=None, force=False):
        if not isinstance(cls, type):
            raise TypeError(f"module must be a class, but got {type(cls)}")
        for n in ([cls.__name__] if name is None else ([name] if isinstance(name, str) else list(name))):
            if not force and n in self._module_dict:
                raise KeyError(f"{n} is already registered in {self._name}")
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError(f"force must be a boolean, but got {type(force)}")
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry: Registry, default_args: Optional[dict] = None):
    """mmcv semantics: ``cfg['type']`` names a registered class (or is a class); the remaining keys
    are constructor keyword arguments; ``default_args`` fill missing keys."""
    if not isinstance(cfg, dict):
        raise TypeError(f"cfg must be a dict, but got {type(cfg)}")
    if "type" not in cfg and not (default_args and "type" in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        cls = registry.get(obj_type)
        if cls is None:
            raise KeyError(f"{obj_type} is not in the {registry.name} registry")
    elif isinstance(obj_type, type):
        cls = obj_type
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(obj_type)}")
    try:
        return cls(**args)
    except Exception as e:
        raise type(e)(f"{cls.__name__}: {e}")


# one registry aliased four ways, like mmaction/models/builder.py:8-14
MODELS = Registry("models")
BACKBONES = HEADS = RECOGNIZERS = LOSSES = MODELS


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_recognizer(cfg, train_cfg=None, test_cfg=None):
    """``builder.build_recognizer`` (mmaction/models/builder.py:37-52)."""
    if train_cfg is not None or test_cfg is not None:
        import warnings
        warnings.warn("train_cfg and test_cfg is deprecated, please specify them in model. Details see this "
                      "PR: https://github.com/open-mmlab/mmaction2/pull/629", UserWarning)
    assert cfg.get("train_cfg") is None or train_cfg is None, "train_cfg specified in both outer field and model field"
    assert cfg.get("test_cfg") is None or test_cfg is None, "test_cfg specified in both outer field and model field"
    return RECOGNIZERS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_model(cfg, train_cfg=None, test_cfg=None):
    """``builder.build_model`` (mmaction/models/builder.py:65-80) for the recognizer family."""
    args = dict(cfg)
    obj_type = args.get("type")
    if obj_type in RECOGNIZERS:
        return build_recognizer(cfg, train_cfg, test_cfg)
    raise ValueError(f"{obj_type} is not registered in RECOGNIZERS (only the AIM ViT-CLIP path is implemented)")


def register_into_mmaction() -> bool:
    """Register this implementation into a real mmaction/mmcv install, if there is one."""
    try:
        from mmaction.models.builder import BACKBONES as MB   # type: ignore
    except Exception:
        return False
    from .aim_variant import AIM
    from .backbone import ViT_CLIP
    MB.register_module(name="ViT_CLIP", force=True, module=ViT_CLIP)
    MB.register_module(name="AIM", force=True, module=AIM)       # stock AIM (vitclip_aim.py:353), wind_attn=False
    return True


# ----------------------------------------------------------------------------------------------
# config files (python files with `_base_` inheritance and dotted overrides)
# ----------------------------------------------------------------------------------------------
class ConfigDict(dict):
    """dict with attribute access (enough of mmcv's ConfigDict for these configs)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def _to_cfgdict(o):
    if isinstance(o, dict):
        return ConfigDict({k: _to_cfgdict(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return type(o)(_to_cfgdict(v) for v in o)
    return o


def _merge(child: dict, base: dict) -> dict:
    """mmcv ``Config._merge_a_into_b``: child keys override; dicts merge unless ``_delete_=True``."""
    out = copy.deepcopy(base)
    for k, v in child.items():
        if isinstance(v, dict) and k in out and isinstance(out[k], dict) and not v.get("_delete_", False):
            out[k] = _merge(v, out[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != "_delete_"}
            out[k] = copy.deepcopy(v)
    return out


def _load_py(path: str) -> dict:
    src = open(path).read()
    ast.parse(src)   # syntax check, like mmcv
    ns: Dict[str, Any] = {"__file__": path}
    exec(compile(src, path, "exec"), ns)   # configs are python files by design (mmcv does the same)
    return {k: v for k, v in ns.items()
            if not k.startswith("__") and not isinstance(v, type(os)) and not callable(v)}


class Config:
    """``Config.fromfile(path)`` with ``_base_`` inheritance and ``merge_from_dict`` dotted overrides."""

    def __init__(self, cfg_dict: Optional[dict] = None, filename: Optional[str] = None):
        object.__setattr__(self, "_cfg_dict", _to_cfgdict(cfg_dict or {}))
        object.__setattr__(self, "filename", filename)

    @staticmethod
    def _file2dict(path: str) -> dict:
        path = os.path.abspath(os.path.expanduser(path))
        if not os.path.isfile(path):
            raise FileNotFoundError(path)
        cfg = _load_py(path)
        bases = cfg.pop("_base_", None)
        if bases is None:
            return cfg
        if isinstance(bases, str):
            bases = [bases]
        merged: dict = {}
        for b in bases:
            bd = Config._file2dict(os.path.join(os.path.dirname(path), b))
            dup = set(merged) & set(bd)
            if dup:
                raise KeyError(f"Duplicate key is not allowed among bases: {sorted(dup)}")
            merged.update(bd)
        return _merge(cfg, merged)

    @staticmethod
    def fromfile(path: str) -> "Config":
        return Config(Config._file2dict(path), filename=path)

    def merge_from_dict(self, options: dict):
        """``--cfg-options model.backbone.num_frames=8`` style overrides (tools/train.py:81-83)."""
        nested: dict = {}
        for full_key, v in options.items():
            d = nested
            keys = full_key.split(".")
            for k in keys[:-1]:
                d = d.setdefault(k, {})
            d[keys[-1]] = v
        object.__setattr__(self, "_cfg_dict", _to_cfgdict(_merge(nested, self._cfg_dict)))

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg_dict))
