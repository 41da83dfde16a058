"""Stub modules that make the *reference* (/root/reference) importable in the build container.

Used ONLY by tools/gen_golden.py (fixture generation, build container only; never on the GPU
box, never by the product).  Recipe: SURVEY.md Appendix C.  The stubs replace packages the
container lacks (pytorch_lightning, torchvision, omegaconf, dlib, wandb, clip, kornia, taming,
LPIPS); none of them carries arithmetic of the hot path except torchvision's tensor
``resize`` / ``normalize``, which are restated with their documented semantics (bilinear,
align_corners=False, no antialias for tensor inputs in torchvision <= 0.14).
"""
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


def _mod(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


class AttrDict(dict):
    """dict with attribute access + hasattr semantics, standing in for OmegaConf nodes."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class ListConfig(list):
    pass


def to_attr(o):
    if isinstance(o, dict):
        return AttrDict({k: to_attr(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return ListConfig(to_attr(v) for v in o)
    return o


def install():
    import transformers  # noqa: F401  (must precede the fake torchvision)
    from transformers import CLIPModel, CLIPTokenizer  # noqa: F401

    if "pytorch_lightning" in sys.modules and getattr(sys.modules["pytorch_lightning"], "_reface_stub", False):
        return

    pl = _mod("pytorch_lightning")
    pl._reface_stub = True

    class LightningModule(nn.Module):
        @property
        def device(self):
            for p in self.parameters():
                return p.device
            return torch.device("cpu")

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.seed_everything = lambda s: torch.manual_seed(s)
    _mod("pytorch_lightning.utilities")
    d = _mod("pytorch_lightning.utilities.distributed")
    d.rank_zero_only = lambda f: f

    tv = _mod("torchvision")
    tvu = _mod("torchvision.utils")

    def make_grid(tensor, nrow=8, padding=2, normalize=False, value_range=None, scale_each=False, pad_value=0.0, **kw):
        """torchvision.utils.make_grid restated from its documented behaviour for a 4-D float tensor, normalize=False: images
        left to right, `nrow` per row, each preceded by `padding` pixels of `pad_value`, plus a closing border."""
        assert tensor.dim() == 4 and not normalize and not scale_each
        if tensor.size(0) == 1:
            return tensor.squeeze(0)
        import math
        nmaps = tensor.size(0)
        xmaps = min(nrow, nmaps)
        ymaps = int(math.ceil(float(nmaps) / xmaps))
        height, width = int(tensor.size(2) + padding), int(tensor.size(3) + padding)
        grid = tensor.new_full((tensor.size(1), height * ymaps + padding, width * xmaps + padding), pad_value)
        k = 0
        for y in range(ymaps):
            for x in range(xmaps):
                if k >= nmaps:
                    break
                grid.narrow(1, y * height + padding, height - padding).narrow(2, x * width + padding, width - padding).copy_(tensor[k])
                k += 1
        return grid

    tvu.make_grid = make_grid
    tvt = _mod("torchvision.transforms")
    tvf = _mod("torchvision.transforms.functional")
    tvm = _mod("torchvision.models")
    tv.utils, tv.transforms, tv.models = tvu, tvt, tvm
    tvt.functional = tvf

    def resize(img, size, *a, **k):
        if isinstance(size, int):
            size = (size, size)
        return F.interpolate(img, size=tuple(size), mode="bilinear", align_corners=False, antialias=False)

    def normalize(t, mean, std, inplace=False):
        mean = torch.as_tensor(mean, dtype=t.dtype, device=t.device).view(-1, 1, 1)
        std = torch.as_tensor(std, dtype=t.dtype, device=t.device).view(-1, 1, 1)
        return (t - mean) / std

    class Resize:
        def __init__(self, size, *a, **k):
            self.size = size

        def __call__(self, img):
            return resize(img, self.size)

    tvf.resize, tvf.normalize = resize, normalize
    tvt.Resize = Resize

    oc = _mod("omegaconf")
    ocl = _mod("omegaconf.listconfig")
    ocl.ListConfig = ListConfig
    oc.listconfig = ocl
    oc.ListConfig = ListConfig

    dl = _mod("dlib")
    dl.get_frontal_face_detector = lambda: (lambda img, up=1: [])
    dl.shape_predictor = lambda path: None

    wb = _mod("wandb")
    wb.log = lambda *a, **k: None
    _mod("clip")
    _mod("kornia")
    _mod("taming")
    _mod("taming.modules")
    _mod("taming.modules.vqvae")
    q = _mod("taming.modules.vqvae.quantize")
    q.VectorQuantizer2 = object

    _mod("eval_tool")
    _mod("eval_tool.lpips")
    lp = _mod("eval_tool.lpips.lpips")

    class LPIPS(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    lp.LPIPS = LPIPS

    sys.path.insert(0, "/root/reference")
    # The reference's `ldm` has no __init__.py (namespace package); this repository ships a regular `ldm` package (the drop-in
    # import-path shim), which would win the lookup as soon as the repo root is on sys.path.  Pin `ldm` to the reference tree.
    if "ldm" in sys.modules and not getattr(sys.modules["ldm"], "_reference_pinned", False):
        raise RuntimeError("`ldm` was imported before tools/ref_shims.install(): it would not be the reference's")
    import types
    ldm = types.ModuleType("ldm")
    ldm.__path__ = ["/root/reference/ldm"]
    ldm._reference_pinned = True
    sys.modules["ldm"] = ldm
