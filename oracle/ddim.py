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
