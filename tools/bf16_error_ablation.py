#!/usr/bin/env python3
"""Where the throughput mode's distance to the exact-fp32 mode comes from.  Full-width UNet + VAE, seeded weights,
B = 2, 64x64 latents, CFG 3.5, S = 50, decoded [0, 1] images against the exact-fp32 engine on the same seeds:

  weights_bf16   exact-fp32 engine (fp32 storage, fp32 MFMA) fed weights ROUNDED to bf16      -> the weight-rounding share
  bf16           the throughput mode (bf16 storage of weights AND activations, fp32 accumulate)  -> + activation / residual-stream rounding
  one-evaluation numbers (eps of a single UNet call at t = 481) for the same two, and the latent distance after k DDIM steps (error growth).

  python tools/bf16_error_ablation.py > profiles/rNN_bf16_error_ablation.json
"""
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from reface_amd import ops, params as P  # noqa: E402
from reface_amd.ddim import DDIMSampler  # noqa: E402

dev = "cuda:0"
torch.cuda.set_device(0)
unet, vae, ldm, cpu_sd = bench.build_models(torch.float32, dev, 0, 1, True)
usd = cpu_sd[0]
B, h, S, scale = 2, 64, 50, 3.5
x_T, z_inp, mask, c, uc = bench.synthetic_inputs(B, h, 4242, dev)


def run(dtype, sd=None):
    if sd is not None:
        unet.load_state_dict(sd, strict=True)
        unet.to(dev)
    unet.set_compute_dtype(dtype)
    sampler = DDIMSampler(ldm)
    samples, inter = sampler.sample(S=S, conditioning=c, batch_size=B, shape=[4, h, h], verbose=False, unconditional_guidance_scale=scale,
                                    unconditional_conditioning=uc, eta=0.0, x_T=x_T, log_every_t=5,
                                    test_model_kwargs={"inpaint_image": z_inp, "inpaint_mask": mask})
    x = vae.decode(samples, inv_scale=1.0 / 0.18215)
    img = torch.empty_like(x)
    ops.to_image(x, img)()
    # one evaluation at t = 481 on a fixed input
    eng = unet.engine(2 * B, h, h, uniform_t=True, cfg_pair=True)
    xin = torch.cat([x_T, z_inp, mask], 1)
    ops.nchw_to_nhwc(torch.cat([xin, xin]).contiguous(), eng.x_in)()
    eng.set_context(torch.cat([uc, c]))
    eng.set_timesteps(torch.full((1,), 481.0, device=dev))
    eng.run()
    eps = eng.eps.clone()
    torch.cuda.synchronize()
    return img.double().cpu(), samples.double().cpu(), [t.double().cpu() for t in inter["x_inter"]], eps.double().cpu()


STEPS_DONE = sorted({1} | {i + 1 for i in range(S) if (S - i - 1) % 5 == 0})          # the steps after which the sampler logs x (ddim.py:247)


def dist(a, r):
    mse = ((a[0] - r[0]) ** 2).mean().item()
    return {"image_max_abs": (a[0] - r[0]).abs().max().item(), "image_mean_abs": (a[0] - r[0]).abs().mean().item(),
            "image_psnr_db": 10.0 * math.log10(1.0 / mse) if mse > 0 else float("inf"),
            "latent_rel_l2": ((a[1] - r[1]).norm() / r[1].norm()).item(),
            "one_evaluation_eps_rel_l2": ((a[3] - r[3]).norm() / r[3].norm()).item(),
            "latent_rel_l2_after_steps": {str(n): ((x - y).norm() / y.norm()).item() for n, x, y in zip(STEPS_DONE, a[2][1:], r[2][1:])}}


ref = run(torch.float32)
out = {"setup": f"full-width UNet / VAE, seeded weights, B = {B}, {8 * h}x{8 * h}, S = {S}, CFG {scale}; reference = exact-fp32 engine"}
rounded = {k: (v.to(torch.bfloat16).float() if v.dtype.is_floating_point and v.dim() >= 2 else v) for k, v in usd.items()}
out["weights_bf16 (fp32 engine, weights rounded to bf16)"] = dist(run(torch.float32, rounded), ref)
out["bf16 (throughput mode)"] = dist(run(torch.bfloat16, usd), ref)
print(json.dumps(out, indent=1))
