"""reface_amd -- MI355X-native (gfx950) REFace inference hot path.

DDIM sampling over the 9-channel SD-inpainting UNet, KL-VAE decode and the CLIP / ArcFace
conditioning encoders, as hand-written HIP kernels behind a C-ABI (include/reface_hip.h),
driven from Python with the reference's own class / method surface.
"""
__version__ = "0.1.0"
