"""Host-side diffusion schedules (tiny fp64/fp32 tables; same arithmetic and dtypes as the reference).

make_beta_schedule / make_ddim_timesteps / make_ddim_sampling_parameters follow
ldm/modules/diffusionmodules/util.py:21-74; register_schedule buffers follow
ldm/models/diffusion/ddpm.py:255-307.
"""
import numpy as np
import torch


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    if schedule != "linear":
        raise NotImplementedError(f"beta schedule '{schedule}' is not on the REFace path")
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    return betas.numpy()


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    if ddim_discr_method != "uniform":
        raise NotImplementedError(f'ddim discretization "{ddim_discr_method}" is not on the REFace path')
    c = num_ddpm_timesteps // num_ddim_timesteps
    steps_out = np.asarray(list(range(0, num_ddpm_timesteps, c))) + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """alphacums: fp32 CPU tensor.  Returns (sigmas fp64 ndarray, alphas fp32 tensor, alphas_prev fp64 ndarray)."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([float(alphacums[0])] + alphacums[ddim_timesteps[:-1]].tolist())
    # the reference evaluates this with a float64 ndarray and a float32 tensor mixed: the division
    # becomes reciprocal(float32) * float64, everything else float64 (util.py:69)
    a64 = alphas.double().numpy()
    recip = (1 - alphas).reciprocal().double().numpy()
    sigmas = eta * np.sqrt(recip * (1 - alphas_prev) * (1 - a64 / alphas_prev))
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
        print(f"For the chosen value of eta, which is {eta}, this results in the following sigma_t schedule for ddim sampler {sigmas}")
    return sigmas, alphas, alphas_prev


def ddpm_buffers(timesteps=1000, linear_start=1e-4, linear_end=2e-2):
    """fp32 buffers of DDPM.register_schedule (ddpm.py:262-307) that the inference path reads."""
    betas = make_beta_schedule("linear", timesteps, linear_start=linear_start, linear_end=linear_end)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    f = lambda a: torch.tensor(a, dtype=torch.float32)
    return {
        "betas": f(betas),
        "alphas_cumprod": f(ac),
        "alphas_cumprod_prev": f(ac_prev),
        "sqrt_alphas_cumprod": f(np.sqrt(ac)),
        "sqrt_one_minus_alphas_cumprod": f(np.sqrt(1.0 - ac)),
        "log_one_minus_alphas_cumprod": f(np.log(1.0 - ac)),
        "sqrt_recip_alphas_cumprod": f(np.sqrt(1.0 / ac)),
        "sqrt_recipm1_alphas_cumprod": f(np.sqrt(1.0 / ac - 1)),
    }


def ddim_step_coefficients(alphas, alphas_prev, sigmas):
    """Per-index fp32 coefficients of p_sample_ddim (ddim.py:357-374), computed with fp32 tensor ops
    as ``torch.full`` + tensor arithmetic do in the reference:
    [sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1 - a_prev - sigma^2), sigma]  -> [S, 5] fp32."""
    a_t = alphas.to(torch.float32)
    a_prev = torch.tensor(np.asarray(alphas_prev), dtype=torch.float64).to(torch.float32)
    sig = torch.tensor(np.asarray(sigmas), dtype=torch.float64).to(torch.float32)
    sqrt_1m = torch.sqrt(1.0 - a_t)                       # ddim_sqrt_one_minus_alphas (ddim.py:135)
    return torch.stack([a_t.sqrt(), sqrt_1m, a_prev.sqrt(), (1.0 - a_prev - sig ** 2).sqrt(), sig], dim=1).contiguous()
