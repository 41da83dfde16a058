#!/usr/bin/env python3
"""Largest |activation| of every launch of one UNet evaluation -- is a set of weights inside fp16's range (65504)?

The fp16 throughput mode (UNetModel.set_compute_dtype(torch.float16)) stores the residual stream, the GEMM operands and the attention
operands in IEEE binary16; accumulation, GroupNorm / LayerNorm statistics, softmax state and the GELU argument stay fp32.  Anything above
65504 in a STORED tensor becomes inf.  This tool runs the bf16 engine (fp32's exponent range: nothing can overflow there) of the benchmark's
shape launch by launch and reports, per launch, the largest magnitude among the 16-bit tensors it reads or writes, then the same for the fp16
engine (count of non-finite values).  `--ckpt` takes a REFace checkpoint; without it the benchmark's seeded random-init weights are used.

  python tools/act_range.py [--ckpt model.ckpt] [--t 981] > profiles/rNN_act_range.txt
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from reface_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--t", type=float, nargs="*", default=[981.0, 481.0, 1.0])
    ap.add_argument("--batch", type=int, default=2)
    args = ap.parse_args()
    dev = "cuda:0"
    torch.cuda.set_device(0)
    unet, vae, ldm, _ = bench.build_models(torch.bfloat16, dev, 0, 1, False)
    if args.ckpt:
        sd = torch.load(args.ckpt, map_location="cpu")
        sd = sd.get("state_dict", sd)
        pre = "model.diffusion_model."
        unet.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
        unet.to(dev)
    B, h = args.batch, 64
    x_T, z_inp, mask, c, uc = bench.synthetic_inputs(B, h, 42, dev)
    xin = torch.cat([x_T, z_inp, mask], 1)
    sp = torch.cuda.current_stream().cuda_stream
    print(f"# weights: {'checkpoint ' + args.ckpt if args.ckpt else 'seeded random init (bench.py)'}; CFG batch {2 * B} at {h}x{h}; fp16 max = 65504")
    for dt in (torch.bfloat16, torch.float16):
        unet.set_compute_dtype(dt)
        eng = unet.engine(2 * B, h, h, uniform_t=True, cfg_pair=True)
        for t in args.t:
            ops.nchw_to_nhwc(torch.cat([xin, xin]).contiguous(), eng.x_in)()
            eng.set_context(torch.cat([uc, c]))
            eng.set_timesteps(torch.full((1,), t, device=dev))
            worst, bad, rows = 0.0, 0, []
            for l in eng.main:
                l(sp)
                m = 0.0
                for k in l.keep:
                    if isinstance(k, torch.Tensor) and k.dtype == dt and k.numel() >= 4096:
                        kf = k.float()
                        bad += int((~torch.isfinite(kf)).sum().item())
                        m = max(m, torch.nan_to_num(kf, nan=0.0, posinf=0.0, neginf=0.0).abs().max().item())
                rows.append((m, l.name))
                worst = max(worst, m)
            torch.cuda.synchronize()
            fin = bool(torch.isfinite(eng.eps).all().item())
            print(f"## {str(dt).split('.')[-1]} engine, t = {t:g}: largest |value| in any 16-bit tensor = {worst:.1f} ({worst / 65504 * 100:.2f} % of fp16's range), "
                  f"non-finite stored values = {bad}, eps finite = {fin}, |eps| max = {eng.eps.abs().max().item():.3f}")
            if dt == torch.bfloat16 and t == args.t[0]:
                for m, name in sorted(rows, reverse=True)[:12]:
                    print(f"   {m:10.1f}  {name}")
        unet._engines.clear()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
