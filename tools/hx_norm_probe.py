#!/usr/bin/env python3
"""EXPERIMENT (profiles/r05ac): the row-extended-A 3x3 convolutions with a GroupNorm (+ SiLU) pass over the staged A tile IN LDS (variant library built from
profiles/r05ac_gn_in_lds.patch; RF_GEMM_DBG 1024 = scale / shift, 3072 = + SiLU; identity parameters, bit-identical results) -- times the launch alone."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reface_amd import ops
DEV = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(4):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
ws = ops.new_workspace(DEV)
tag = f"{os.path.basename(os.environ.get('REFACE_HIP_LIB', 'in-tree')):10s} RF_GEMM_DBG={os.environ.get('RF_GEMM_DBG', '0'):5s}"
with ops.workspace_scope(ws):
    for B, hw, cin, co in ((16, 64, 320, 320), (16, 64, 640, 320), (16, 32, 640, 640), (16, 32, 1280, 640)):
        dt = torch.bfloat16
        g = torch.Generator().manual_seed(cin + hw)
        x = torch.randn((B, hw, hw, cin), generator=g).to(dt).to(DEV)
        w = torch.randn((co, cin, 3, 3), generator=g) / math.sqrt(9 * cin)
        bias = torch.randn((co,), generator=g).to(DEV)
        y = torch.empty((B, hw, hw, co), dtype=dt, device=DEV)
        l = ops.conv2d(x, ops.pack_conv_weight(w, dt, korder=2).to(DEV), y, bias, korder=2)
        pl = ops.gemm_plan2(l)
        l(); torch.cuda.synchronize()
        chk = float(y.float().abs().sum().item())
        print(f"{tag} {B} x {hw}x{hw} {cin} -> {co} ({pl['bm']}x{pl['bn']}): {timeit(l):6.1f} us   checksum {chk:.6e}", flush=True)
