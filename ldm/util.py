"""ldm/util.py surface used by the inference path (reference ldm/util.py:64-93)."""
from inspect import isfunction

from reface_amd.registry import get_obj_from_str, instantiate_from_config  # noqa: F401


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if isfunction(d) else d


def count_params(model, verbose=False):
    total_params = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {total_params * 1.e-6:.2f} M params.")
    return total_params
