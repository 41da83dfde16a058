"""Where does the wall time of one batch go?  (bench-sized workload, per-phase host + device timing)"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from reface_amd import ops
from reface_amd.ddim import DDIMSampler
dev = "cuda:0"
unet, vae, ldm, _ = bench.build_models(torch.bfloat16, dev, 0, 1, False)
sampler = DDIMSampler(ldm)
B, h, S = 8, 64, 50
x_T, z_inp, mask, c, uc = bench.synthetic_inputs(B, h, 42, dev)
def run():
    t0 = time.perf_counter()
    samples, _ = sampler.sample(S=S, conditioning=c, batch_size=B, shape=[4, h, h], verbose=False, unconditional_guidance_scale=3.5,
                                unconditional_conditioning=uc, eta=0.0, x_T=x_T, test_model_kwargs={"inpaint_image": z_inp, "inpaint_mask": mask})
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    x = vae.decode(samples, inv_scale=1 / 0.18215)
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    return t1 - t0, t2 - t0, t3 - t2, t4 - t2
run()
for _ in range(2):
    a, b_, c_, d = run()
    print(f"sample: host returns after {a*1e3:.1f} ms, device done after {b_*1e3:.1f} ms | decode: host {c_*1e3:.1f} ms, device {d*1e3:.1f} ms")
# per-step replay cost
plan = list(sampler._plans.values())[0]
g = list(plan["graphs"].values())[0]
nst = g["tab"].shape[0]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): g["graph"].replay()
torch.cuda.synchronize(); print(f"graph replay ({nst} steps per graph): {(time.perf_counter()-t0)/3/nst*1e3:.3f} ms per step")
eng = plan["eng"]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): ops.run(plan["step"])
torch.cuda.synchronize(); print(f"eager launch list: {(time.perf_counter()-t0)/5*1e3:.2f} ms per step ({len(plan['step'])} launches)")
