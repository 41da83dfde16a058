"""``ldm.models.diffusion.ddim`` -> reface_amd.ddim (scripts/inference_test_bench.py:20)."""
from reface_amd.ddim import DDIMSampler  # noqa: F401
