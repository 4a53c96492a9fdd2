"""Config / plugin helpers with the reference's names and behaviour (reference: sgm/util.py:170-194,
sgm/util.py `default`, `exists`, `append_dims`, `append_zero`, `disabled_train`)."""
from __future__ import annotations

import importlib
from inspect import isfunction

import torch


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if isfunction(d) else d


def get_obj_from_str(string: str, reload: bool = False, invalidate_cache: bool = True):
    """`pkg.mod.Class` -> the class object (sgm/util.py:179-187)."""
    module, cls = string.rsplit(".", 1)
    if invalidate_cache:
        importlib.invalidate_caches()
    if reload:
        importlib.reload(importlib.import_module(module))
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config):
    """{"target": dotted path, "params": {...}} -> object (sgm/util.py:170-176). Same sentinel strings, same errors."""
    if "target" not in config:
        if config == "__is_first_stage__":
            return None
        if config == "__is_unconditional__":
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def append_zero(x: torch.Tensor) -> torch.Tensor:
    return torch.cat([x, x.new_zeros([1])])


def append_dims(x: torch.Tensor, target_dims: int) -> torch.Tensor:
    """Append trailing singleton dims until `x` has `target_dims` dims (sgm/util.py:194-201)."""
    extra = target_dims - x.ndim
    if extra < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * extra]


def disabled_train(self, mode: bool = True):
    """Overwrite model.train with this function to make sure train/eval mode does not change anymore."""
    return self


def count_params(model, verbose: bool = False) -> int:
    total = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {total * 1.e-6:.2f} M params.")
    return total


def remap_targets(config, src_prefix: str = "sgm.", dst_prefix: str = "v3d_amd.sgm."):
    """Rewrite every `target:` string of a (nested) reference config onto this package — the whole drop-in switch."""
    if isinstance(config, dict):
        out = {}
        for k, v in config.items():
            if k == "target" and isinstance(v, str) and v.startswith(src_prefix):
                out[k] = dst_prefix + v[len(src_prefix):]
            else:
                out[k] = remap_targets(v, src_prefix, dst_prefix)
        return out
    if isinstance(config, (list, tuple)):
        return type(config)(remap_targets(v, src_prefix, dst_prefix) for v in config)
    return config
