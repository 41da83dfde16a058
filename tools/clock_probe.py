#!/usr/bin/env python3
"""Diagnostics: run the bench workload and sample the actual shader clock after every DDIM step (tools/clock_probe.hip).
Prints ms per batch and the mean / min relative clock (dependent-FMA iterations per 100 MHz tick)."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from reface_amd import ops  # noqa: E402
from reface_amd.ddim import DDIMSampler  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libclockprobe.so"))
lib.clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = "cuda:0"
torch.cuda.set_device(0)
unet, vae, ldm, _ = bench.build_models(torch.bfloat16, dev, 0, 1, False)
sampler = DDIMSampler(ldm)
B, h, S = 8, 64, 50
x_T, z_inp, mask, c, uc = bench.synthetic_inputs(B, h, 42, dev)
ITERS = 20000
probes = torch.zeros((S + 1, 2), dtype=torch.int64, device=dev)


def cb(px0, i):
    lib.clock_probe(probes[i].data_ptr(), ITERS, torch.cuda.current_stream().cuda_stream)


def run(callback):
    samples, _ = sampler.sample(S=S, conditioning=c, batch_size=B, shape=[4, h, h], verbose=False, unconditional_guidance_scale=3.5,
                                unconditional_conditioning=uc, eta=0.0, x_T=x_T, test_model_kwargs={"inpaint_image": z_inp, "inpaint_mask": mask},
                                img_callback=callback)
    return vae.decode(samples, inv_scale=1.0 / 0.18215)


run(None)
torch.cuda.synchronize()
# idle clock reference
time.sleep(0.5)
lib.clock_probe(probes[S].data_ptr(), ITERS, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
idle = ITERS / float(probes[S, 0])
for rep in range(2):
    t0 = time.perf_counter()
    run(cb)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    r = ITERS / probes[:S, 0].double().cpu()
    print(f"batch {ms:8.1f} ms   FMA iters / 10ns tick: idle-after-sleep {idle:.4f}  in-loop mean {r.mean():.4f} min {r.min():.4f} max {r.max():.4f}  first5 {[round(float(v),4) for v in r[:5]]}")
