"""``ldm.models.diffusion.ddpm`` -> reface_amd.ddpm (configs/train.yaml:3)."""
from reface_amd.ddpm import DiffusionWrapper, LatentDiffusion  # noqa: F401
