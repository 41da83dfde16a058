#!/usr/bin/env python3
"""Why is the fp16 mode ~3 % slower than bf16 on identical kernels?  rf_conv_gemm on one long-K shape in both 16-bit types with (a) N(0,1)-class operands,
(b) all-zero operands, (c) operands whose low mantissa bits are zero in BOTH types (values with 4 significant bits): if (b) / (c) close the gap, the
difference is switching power in the multiplier array (the chip is power-managed: MI355X_MICROARCH.md DVFS note), not the instruction.
  python tools/f16_rate_probe.py > profiles/rNN_f16_rate_probe.txt"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reface_amd import ops  # noqa: E402

dev = "cuda"
M, N, K = 65536, 320, 2880
print(f"# rf_conv_gemm {M}x{N}x{K} (256x320 tile), sustained over ~1.5 s per case, us per launch; same kernel template, element type bf16_t / f16_t")
for kind in ("normal", "zeros", "4-bit mantissas"):
    row = []
    for dt in (torch.bfloat16, torch.float16):
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(M, K, device=dev, generator=g) * 0.5
        w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
        if kind == "zeros":
            x.zero_(); w.zero_()
        elif kind == "4-bit mantissas":          # keep sign, exponent and the top 3 explicit mantissa bits: exactly representable in both formats
            x = (x.view(torch.int32) & ~0xFFFFF).view(torch.float32)
            w = (w.view(torch.int32) & ~0xFFFFF).view(torch.float32)
        x, w = x.to(dt), w.to(dt)
        out = torch.empty(M, N, device=dev, dtype=dt)
        l = ops.linear(x, w, out, None)
        for _ in range(20):
            l()
        torch.cuda.synchronize()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 1.5:
            for _ in range(50):
                l()
            torch.cuda.synchronize()
            n += 50
        us = (time.perf_counter() - t0) / n * 1e6
        row.append(us)
    print(f"{kind:18s} bf16 {row[0]:7.1f} us ({2.0 * M * N * K / row[0] / 1e6:5.0f} TF)   fp16 {row[1]:7.1f} us ({2.0 * M * N * K / row[1] / 1e6:5.0f} TF)   fp16 / bf16 = {row[1] / row[0]:.3f}")
