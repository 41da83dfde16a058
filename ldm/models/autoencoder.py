"""``ldm.models.autoencoder`` -> reface_amd.vae (configs/train.yaml:50)."""
from reface_amd.vae import AutoencoderKL  # noqa: F401
