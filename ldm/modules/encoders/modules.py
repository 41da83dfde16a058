"""``ldm.modules.encoders.modules`` -> reface_amd.encoders (configs/train.yaml:73)."""
from reface_amd.encoders import FrozenCLIPEmbedder  # noqa: F401
