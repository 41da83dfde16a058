"""YAML config loading for the CLI: OmegaConf when installed (as the reference uses), else PyYAML wrapped in an
attribute-access dict with the subset of OmegaConf behaviour the path relies on (attribute + key access,
``in``, ``.get``, ``hasattr``)."""
import yaml


class Node(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def wrap(o):
    if isinstance(o, dict):
        return Node({k: wrap(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return [wrap(v) for v in o]
    return o


def load(path):
    try:
        from omegaconf import OmegaConf
        return OmegaConf.load(path)
    except ImportError:
        with open(path) as f:
            return wrap(yaml.safe_load(f))
