#!/usr/bin/env python3
"""Time rf_groupnorm_fold_linear alone (the 10 launches of a configs[1] step: C = 320 at 64x64, C = 640 at 32x32, 16 samples).  Diagnostic only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reface_amd import ops
DEV = "cuda:0"
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
for C, HW, B, nch in ((320, 4096, 16, 16), (320, 4096, 16, 32), (320, 4096, 8, 32), (640, 1024, 16, 4), (640, 1024, 16, 8), (640, 1024, 16, 64)):
    W = torch.randn(C, C, device=DEV); g = torch.rand(C, device=DEV) + 0.5; b = torch.randn(C, device=DEV); bias = torch.randn(C, device=DEV)
    part = torch.rand(B, nch, 32, 2, dtype=torch.float64, device=DEV) * 100 + 1000
    part[..., 1] = part[..., 0] ** 2 / (HW * C / 32 / nch) * 1.5
    l, wout, rv = ops.groupnorm_fold_linear(W, g, b, bias, part, nch, B=B, HW=HW, eps=1e-6, dtype=torch.bfloat16)
    empty = torch.empty(1, device=DEV)
    print(f"C {C} HW {HW} B {B} nchunks {nch}: {timeit(l):6.2f} us   (W' {B * C * C * 2 / 1e6:.1f} MB)", flush=True)
