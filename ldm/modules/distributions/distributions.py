"""``ldm.modules.distributions.distributions`` -> reface_amd.vae."""
from reface_amd.vae import DiagonalGaussianDistribution  # noqa: F401
