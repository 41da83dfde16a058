"""Schedule helpers of ldm/modules/diffusionmodules/util.py:21-74 -> reface_amd.schedule."""
from reface_amd.schedule import make_beta_schedule, make_ddim_sampling_parameters, make_ddim_timesteps  # noqa: F401
