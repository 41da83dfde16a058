#!/usr/bin/env python3
"""Time rf_conv3x3_stem alone (configs[1]: 8 unique samples of 64x64, C = 320, stored twice) under the library REFACE_HIP_LIB names; beside it the implicit
GEMM it replaces (16 samples) + the statistics pass.  Diagnostic only."""
import math, os, sys, torch
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
def main():
    B, hw, c = int(os.environ.get("B", 8)), int(os.environ.get("HW", 64)), 320
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(1)
    x = torch.randn((2 * B, hw, hw, 16), generator=g).to(dt).to(DEV); x[..., 9:] = 0; x[B:] = x[:B]
    w = torch.randn((c, 9, 3, 3), generator=g) / 9.0
    wp = ops.pack_conv_weight(w, dt, cin_pad=16).to(DEV)
    bias = torch.randn((c,), generator=g).to(DEV)
    cat = torch.zeros((2 * B, hw, hw, 2 * c), dtype=dt, device=DEV)
    y = cat[..., c:]
    l = ops.conv3x3_stem(x[:B], wp, bias, y[:B], dup=y[B:])
    f1 = ops.fuse_groupnorm_stats(y[:B], [(l, 0, B * hw * hw, 0, c)])
    l0 = ops.conv3x3_stem(x[:B], wp, bias, y[:B], dup=y[B:])
    lg = ops.conv2d(x, wp, y, bias)
    part = torch.empty(2 * B * ops.GN_MAX_CHUNKS * 64, dtype=torch.float64, device=DEV)
    ls, _ = ops.groupnorm_stats(y[:B], part)
    lib = os.path.basename(os.environ.get("REFACE_HIP_LIB", "in-tree"))
    print(f"{lib:14s} B {B} {hw}x{hw}: stem+stats {timeit(l):6.1f} us | stem, no stats {timeit(l0):6.1f} us | implicit GEMM (2B samples) {timeit(lg):6.1f} us + statistics pass {timeit(ls):6.1f} us", flush=True)
main()
