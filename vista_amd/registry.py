"""Name -> class registries and config-dict construction for the model-side plugin API of this path.

The reference builds its reader / backbone / pipeline stage from `dict(type="Name", **kwargs)`
through `build_from_cfg(cfg, registry)` (det3d/utils/registry.py:49-78) with classes announced by
`@BACKBONES.register_module` / `@READERS.register_module` / `@PIPELINES.register_module`
(det3d/models/registry.py:3-4, det3d/datasets/registry.py). The same three registries exist here
under the same names, holding this repo's GPU implementations, so a reference config section such as
`backbone=dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8)` constructs unchanged.
"""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._classes = {}

    name = property(lambda self: self._name)
    module_dict = property(lambda self: self._classes)

    def __repr__(self):
        return f"Registry(name={self._name}, items={sorted(self._classes)})"

    def __contains__(self, key):
        return key in self._classes

    def get(self, key):
        return self._classes.get(key)

    def register_module(self, cls):
        """Class decorator; a second class of the same name is an error, as in the reference."""
        if not inspect.isclass(cls):
            raise TypeError(f"module must be a class, but got {type(cls)}")
        if cls.__name__ in self._classes:
            raise KeyError(f"{cls.__name__} is already registered in {self._name}")
        self._classes[cls.__name__] = cls
        return cls


def build_from_cfg(cfg, registry, default_args=None):
    """cfg["type"] is a registered name or a class; the remaining keys (then default_args for keys
    the cfg does not set) become constructor kwargs."""
    if not isinstance(cfg, dict) or "type" not in cfg:
        raise AssertionError('cfg must be a dict with the key "type"')
    if default_args is not None and not isinstance(default_args, dict):
        raise AssertionError("default_args must be a dict or None")
    kwargs = {k: v for k, v in cfg.items() if k != "type"}
    kind = cfg["type"]
    if isinstance(kind, str):
        cls = registry.get(kind)
        if cls is None:
            raise KeyError(f"{kind} is not in the {registry.name} registry")
    elif inspect.isclass(kind):
        cls = kind
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(kind)}")
    for k, v in (default_args or {}).items():
        kwargs.setdefault(k, v)
    return cls(**kwargs)


READERS = Registry("reader")
BACKBONES = Registry("backbone")
PIPELINES = Registry("pipeline")


def build_reader(cfg):
    return build_from_cfg(cfg, READERS)


def build_backbone(cfg):
    return build_from_cfg(cfg, BACKBONES)
