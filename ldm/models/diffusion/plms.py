"""``ldm.models.diffusion.plms`` -> reface_amd.plms (scripts/inference_test_bench.py --plms)."""
from reface_amd.plms import PLMSSampler  # noqa: F401
