"""``ldm.modules.diffusionmodules.openaimodel`` -> reface_amd.unet (configs/train.yaml:32)."""
from reface_amd.unet import UNetModel  # noqa: F401
