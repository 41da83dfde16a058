"""The reference's "operator registry": ``target:`` dotted path -> class(**params)  (ldm/util.py:78-93).

Kept byte-compatible in behaviour: plain importlib lookup, ``KeyError("Expected key `target` to
instantiate.")`` when the key is missing, the two sentinel strings return None.
"""
import importlib


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    if reload:
        importlib.reload(importlib.import_module(module))
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config == "__is_first_stage__":
            return None
        elif config == "__is_unconditional__":
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def cfg_get(cfg, path, default=None):
    """Attribute/key access into OmegaConf nodes or plain dicts: cfg_get(c, 'other_params.ID_weight')."""
    cur = cfg
    for part in path.split("."):
        if cur is None:
            return default
        if isinstance(cur, dict):
            if part not in cur:
                return default
            cur = cur[part]
        else:
            try:
                if hasattr(cur, "__contains__") and not isinstance(cur, str) and part not in cur:
                    return default
            except TypeError:
                pass
            if not hasattr(cur, part):
                try:
                    cur = cur[part]
                    continue
                except Exception:
                    return default
            cur = getattr(cur, part)
    return cur
