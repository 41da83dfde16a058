#!/usr/bin/env python3
"""Experiment: overlap the fp32 VAE decode of batch i (side stream) with the DDIM loop of batch i+1 (main stream)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from reface_amd import ops  # noqa: E402
from reface_amd.ddim import DDIMSampler  # noqa: E402

dev = "cuda:0"
torch.cuda.set_device(0)
unet, vae, ldm, _ = bench.build_models(torch.bfloat16, dev, 0, 1, False)
sampler = DDIMSampler(ldm)
B, h, S = 8, 64, 50
x_T, z_inp, mask, c, uc = bench.synthetic_inputs(B, h, 42, dev)
img_out = torch.empty((B, 3, 8 * h, 8 * h), dtype=torch.float32, device=dev)
side = torch.cuda.Stream()


def ddim():
    s, _ = sampler.sample(S=S, conditioning=c, batch_size=B, shape=[4, h, h], verbose=False, unconditional_guidance_scale=3.5,
                          unconditional_conditioning=uc, eta=0.0, x_T=x_T, test_model_kwargs={"inpaint_image": z_inp, "inpaint_mask": mask})
    return s


def decode(s):
    x = vae.decode(s, inv_scale=1.0 / 0.18215)
    ops.to_image(x, img_out)()


def serial(n):
    for _ in range(n):
        decode(ddim())


def overlapped(n):
    main = torch.cuda.current_stream()
    pending = None
    for _ in range(n):
        s = ddim().clone()
        if pending is not None:
            main.wait_stream(side)          # one decode in flight at a time (the VAE engine's buffers are single)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            decode(s)
        pending = s
    main.wait_stream(side)


for name, fn in (("serial", serial), ("overlapped", overlapped), ("serial", serial), ("overlapped", overlapped)):
    fn(1)
    torch.cuda.synchronize()
    n = int(os.environ.get("NB", "3"))
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:11s} {dt / n * 1e3:8.1f} ms/batch  {B * n / dt:.3f} img/s")
