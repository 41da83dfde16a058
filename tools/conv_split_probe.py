#!/usr/bin/env python3
"""configs[3] (768x768: 96x96 latent, 8 UNet samples): the top level's 3x3 convolutions have M = 73728 rows = 288 tiles of 256 x 320 (1.125 rounds of 256 CUs).
Times the launch on B = 8 samples against the same work split by samples (7 + 1, 6 + 2).  Diagnostic only."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reface_amd import ops
DEV = "cuda:0"
def timeit(fns, n=20):
    for _ in range(3):
        for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(4):
        e0.record()
        for _ in range(n):
            for f in fns: f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
def main():
    hw, co, dt = int(os.environ.get("HW", 96)), 320, torch.bfloat16
    ws = ops.new_workspace(DEV)
    with ops.workspace_scope(ws):
        for cin in (320, 640, 960):
            g = torch.Generator().manual_seed(cin)
            x = torch.randn((8, hw, hw, cin), generator=g).to(dt).to(DEV)
            w = torch.randn((co, cin, 3, 3), generator=g) / math.sqrt(9 * cin)
            wp = ops.pack_conv_weight(w, dt).to(DEV)
            bias = torch.randn((co,), generator=g).to(DEV)
            y = torch.empty((8, hw, hw, co), dtype=dt, device=DEV)
            res = {}
            for name, parts in (("8", [(0, 8)]), ("7+1", [(0, 7), (7, 8)]), ("6+2", [(0, 6), (6, 8)]), ("4+4", [(0, 4), (4, 8)]), ("7", [(0, 7)]), ("1", [(7, 8)]), ("2", [(6, 8)])):
                ls = [ops.conv2d(x[a:b], wp, y[a:b], bias) for a, b in parts]
                plans = [ops.gemm_plan2(l) for l in ls]
                res[name] = (timeit(ls), [(p["bm"], p["bn"], p["splitk"]) for p in plans])
            fl = 2.0 * 8 * hw * hw * co * 9 * cin
            print(f"{hw}x{hw} {cin} -> {co}: " + " | ".join(f"{k}: {v[0]:6.1f} us {v[1]}" for k, v in res.items()) + f" | whole at {fl / res['8'][0] / 1e6:.0f} TF, 7+1 at {fl / res['7+1'][0] / 1e6:.0f} TF", flush=True)
main()
