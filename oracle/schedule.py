"""Oracle: diffusion / DDIM schedules (test infrastructure -- see oracle/__init__.py).

Follows ldm/modules/diffusionmodules/util.py:21-74 and ldm/models/diffusion/ddpm.py:255-307,
ldm/models/diffusion/ddim.py:110-139.
"""
import numpy as np
import torch


def make_beta_schedule(n_timestep=1000, linear_start=0.00085, linear_end=0.0120):
    """util.py:21-26 ("linear" schedule: linspace of sqrt(beta) in float64, squared)."""
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    return betas.numpy()


def alphas_cumprod(n_timestep=1000, linear_start=0.00085, linear_end=0.0120):
    """ddpm.py:262-272: cumprod in float64, stored as float32 buffers."""
    betas = make_beta_schedule(n_timestep, linear_start, linear_end)
    ac = np.cumprod(1.0 - betas, axis=0)
    return torch.tensor(ac, dtype=torch.float32)


def ddim_timesteps(num_ddim, num_ddpm=1000):
    """util.py:46-60 ("uniform"): arange(0, T, T // S) + 1."""
    c = num_ddpm // num_ddim
    return np.asarray(list(range(0, num_ddpm, c))) + 1


def ddim_parameters(ac: torch.Tensor, ts: np.ndarray, eta: float):
    """util.py:63-74 + ddim.py:130-135.

    ``alphas`` stays a float32 tensor; ``alphas_prev`` / ``sigmas`` become float64 numpy arrays
    (built from python floats of float32 values); ``sqrt_one_minus_alphas`` is float32.
    Returns dict of per-index python floats exactly as ``torch.full(..., value)`` would see them.
    """
    alphas = ac[ts]                                   # float32 tensor
    alphas_prev = np.asarray([ac[0]] + ac[ts[:-1]].tolist())   # float64 array (ac[0] -> float32 item)
    # util.py:69 mixes a float64 ndarray with a float32 tensor: ndarray / tensor dispatches to
    # Tensor.__rtruediv__ = reciprocal(self) [float32] * other [float64]; the rest is float64.
    a64 = alphas.double().numpy()
    recip = (1 - alphas).reciprocal().double().numpy()
    sigmas = eta * np.sqrt(recip * (1 - alphas_prev) * (1 - a64 / alphas_prev))
    sqrt_one_minus = torch.sqrt(1.0 - alphas)         # np.sqrt(tensor) == float32 tensor sqrt
    return {
        "alphas": alphas,
        "alphas_prev": alphas_prev,
        "sigmas": np.asarray(sigmas, dtype=np.float64),
        "sqrt_one_minus_alphas": sqrt_one_minus,
    }
