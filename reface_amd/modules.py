"""Parameter containers with the reference's ``state_dict`` key layout.

The reference modules are ``nn.Module`` trees whose only role on the inference path, apart from
their ATen forward, is to own named parameters that a Lightning checkpoint is loaded into
(scripts/inference_test_bench.py:98-112: ``load_state_dict(sd, strict=False)``, ``.cuda()``,
``.eval()``).  ``ParamTree`` rebuilds exactly that naming from a flat spec (reface_amd/params.py)
so checkpoints load unchanged; the arithmetic lives in the HIP engines.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

_BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked")


class ParamTree(nn.Module):
    """Nested module whose leaves are parameters (or BatchNorm-style buffers) named by dotted keys."""

    def __init__(self, specs=None):
        super().__init__()
        if specs:
            for key, shape in specs.items():
                self._add(key, shape)

    def _add(self, key, shape):
        parts = key.split(".")
        node = self
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, ParamTree())
            node = node._modules[p]
        leaf = parts[-1]
        if leaf in _BUFFER_LEAVES:
            dt = torch.int64 if leaf == "num_batches_tracked" else torch.float32
            node.register_buffer(leaf, torch.zeros(tuple(shape), dtype=dt))
        else:
            node.register_parameter(leaf, nn.Parameter(torch.zeros(tuple(shape), dtype=torch.float32), requires_grad=False))

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("ParamTree holds parameters only; the compute path is the HIP engine")


def flat_state(module: nn.Module, prefix: str = "") -> "OrderedDict[str, torch.Tensor]":
    """state_dict of ``module`` restricted to keys under ``prefix`` with the prefix stripped."""
    out = OrderedDict()
    for k, v in module.state_dict().items():
        if k.startswith(prefix):
            out[k[len(prefix):]] = v
    return out


def weights_version(module: nn.Module) -> tuple:
    """Cheap change detector for packed-weight caches: (device, sum of tensor versions)."""
    dev = None
    ver = 0
    for t in list(module.parameters()) + list(module.buffers()):
        dev = t.device
        ver += t._version
    return (str(dev), ver)
