"""Oracle: DDIM sampling loop with classifier-free guidance.  TEST INFRASTRUCTURE.

Follows ldm/models/diffusion/ddim.py:141-251 (sample / ddim_sampling) and :323-375
(p_sample_ddim).  ``eps_fn(x9, t, ctx)`` is the UNet (oracle.unet.unet_forward bound to a
state dict); the coefficient arithmetic is done with fp32 tensors exactly as ``torch.full``
+ tensor ops do in the reference.
"""
import numpy as np
import torch

from . import schedule


def p_sample_ddim(eps_fn, x, c, t, index, params, z_inpaint, mask, scale, uc, noise=None):
    """ddim.py:323-375.  Returns (x_prev, pred_x0)."""
    b = x.shape[0]
    xin = torch.cat([x, z_inpaint, mask], dim=1)                      # :330
    if uc is None or scale == 1.0:
        e_t = eps_fn(xin, t, c)                                       # :336
    else:
        x_in = torch.cat([xin] * 2)                                   # :338
        t_in = torch.cat([t] * 2)
        c_in = torch.cat([uc, c])                                     # :344 (uncond first)
        e_u, e_c = eps_fn(x_in, t_in, c_in).chunk(2)
        e_t = e_u + scale * (e_c - e_u)                               # :346
    a_t = torch.full((b, 1, 1, 1), float(params["alphas"][index]))                    # :357
    a_prev = torch.full((b, 1, 1, 1), float(params["alphas_prev"][index]))            # :358
    sigma_t = torch.full((b, 1, 1, 1), float(params["sigmas"][index]))                # :359
    sqrt_1m = torch.full((b, 1, 1, 1), float(params["sqrt_one_minus_alphas"][index]))  # :360
    pred_x0 = (xin[:, :4] - sqrt_1m * e_t) / a_t.sqrt()               # :364
    dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t               # :370
    if noise is None:
        noise = torch.zeros_like(dir_xt)
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt + sigma_t * noise       # :371-374 (temperature 1)
    return x_prev, pred_x0


def sample(eps_fn, S, x_T, cond, uc, z_inpaint, mask, scale, eta=0.0, num_ddpm=1000,
           linear_start=0.00085, linear_end=0.0120, noises=None, log_every_t=100):
    """ddim.py:141-251.  ``noises[i]`` (optional) is the N(0,1) draw of iteration i (only used
    when eta > 0; at eta = 0 the reference multiplies its draw by sigma = 0)."""
    ac = schedule.alphas_cumprod(num_ddpm, linear_start, linear_end)
    ts = schedule.ddim_timesteps(S, num_ddpm)
    params = schedule.ddim_parameters(ac, ts, eta)
    img = x_T
    b = img.shape[0]
    inter = {"x_inter": [img], "pred_x0": [img]}
    time_range = np.flip(ts)                                          # :222
    total = ts.shape[0]
    for i, step in enumerate(time_range):
        index = total - i - 1                                         # :230
        t = torch.full((b,), int(step), dtype=torch.long)             # :231
        nz = None if noises is None else noises[i]
        img, pred_x0 = p_sample_ddim(eps_fn, img, cond, t, index, params, z_inpaint, mask, scale, uc, nz)
        if index % log_every_t == 0 or index == total - 1:            # :247-249
            inter["x_inter"].append(img)
            inter["pred_x0"].append(pred_x0)
    return img, inter


def q_sample(x_start, t, noise, num_ddpm=1000, linear_start=0.00085, linear_end=0.0120):
    """ddpm.py:412-415: sqrt(acp[t]) * x_start + sqrt(1 - acp[t]) * noise with the fp32 register_schedule buffers (ddpm.py:281-282)."""
    ac = schedule.alphas_cumprod(num_ddpm, linear_start, linear_end)
    sa = torch.tensor(np.sqrt(ac), dtype=torch.float32)
    s1 = torch.tensor(np.sqrt(1.0 - ac), dtype=torch.float32)
    shape = (x_start.shape[0],) + (1,) * (x_start.dim() - 1)
    return sa[t].reshape(shape) * x_start + s1[t].reshape(shape) * noise


def plms_sample(eps_fn, S, x_T, cond, uc, z_inpaint, mask, scale, num_ddpm=1000, linear_start=0.00085, linear_end=0.0120,
                log_every_t=100):
    """plms.py:116-237 (plms_sampling / p_sample_plms), eta = 0: pseudo improved Euler on the first step (one extra UNet call),
    then 2nd / 3rd / 4th-order Adams-Bashforth on the guided eps."""
    ac = schedule.alphas_cumprod(num_ddpm, linear_start, linear_end)
    ts = schedule.ddim_timesteps(S, num_ddpm)
    params = schedule.ddim_parameters(ac, ts, 0.0)
    b = x_T.shape[0]

    def model_output(x, t):                                            # plms.py:190-204
        xin = torch.cat([x, z_inpaint, mask], dim=1)                   # :225
        if uc is None or scale == 1.0:
            return eps_fn(xin, t, cond)
        e_u, e_c = eps_fn(torch.cat([xin] * 2), torch.cat([t] * 2), torch.cat([uc, cond])).chunk(2)
        return e_u + scale * (e_c - e_u)

    def x_prev_and_pred_x0(x, e_t, index):                             # plms.py:206-223
        a_t = torch.full((b, 1, 1, 1), float(params["alphas"][index]))
        a_prev = torch.full((b, 1, 1, 1), float(params["alphas_prev"][index]))
        sigma_t = torch.full((b, 1, 1, 1), float(params["sigmas"][index]))
        sqrt_1m = torch.full((b, 1, 1, 1), float(params["sqrt_one_minus_alphas"][index]))
        pred_x0 = (x - sqrt_1m * e_t) / a_t.sqrt()
        dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
        return a_prev.sqrt() * pred_x0 + dir_xt, pred_x0

    img = x_T
    inter = {"x_inter": [img], "pred_x0": [img]}
    time_range = np.flip(ts)
    total = ts.shape[0]
    old_eps = []
    for i, step in enumerate(time_range):
        index = total - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        t_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), dtype=torch.long)
        e_t = model_output(img, t)
        if len(old_eps) == 0:                                          # :227-231
            x_prev, _ = x_prev_and_pred_x0(img, e_t, index)
            e_t_prime = (e_t + model_output(x_prev, t_next)) / 2
        elif len(old_eps) == 1:
            e_t_prime = (3 * e_t - old_eps[-1]) / 2
        elif len(old_eps) == 2:
            e_t_prime = (23 * e_t - 16 * old_eps[-1] + 5 * old_eps[-2]) / 12
        else:
            e_t_prime = (55 * e_t - 59 * old_eps[-1] + 37 * old_eps[-2] - 9 * old_eps[-3]) / 24
        img, pred_x0 = x_prev_and_pred_x0(img, e_t_prime, index)
        old_eps.append(e_t)
        if len(old_eps) >= 4:
            old_eps.pop(0)
        if index % log_every_t == 0 or index == total - 1:
            inter["x_inter"].append(img)
            inter["pred_x0"].append(pred_x0)
    return img, inter
