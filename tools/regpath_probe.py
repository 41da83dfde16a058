#!/usr/bin/env python3
"""What giving up the LDS-DMA costs a 3x3 convolution: the same convolution through rf_conv_gemm's direct-to-LDS main loop (row-extended A tiles where the
plan allows) and through its register-staged loop (global -> VGPR -> LDS: the path on which an operand could be normalised in registers), forced here by
presenting the input as a 2-source channel concat of its two halves.  Beside them the rf_groupnorm_apply pass such a fusion would remove.  Diagnostic only."""
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
with ops.workspace_scope(ws):
    for B, hw, cin, co in ((16, 64, 320, 320), (16, 64, 640, 320), (16, 32, 640, 640), (16, 32, 1280, 640)):
        dt = torch.bfloat16
        g = torch.Generator().manual_seed(cin + hw)
        x = torch.randn((B, hw, hw, cin), generator=g).to(dt).to(DEV)
        w = torch.randn((co, cin, 3, 3), generator=g) / math.sqrt(9 * cin)
        bias = torch.randn((co,), generator=g).to(DEV)
        y = torch.empty((B, hw, hw, co), dtype=dt, device=DEV)
        y2 = torch.empty_like(y)
        wp2 = ops.pack_conv_weight(w, dt, korder=2).to(DEV)
        wp0 = ops.pack_conv_weight(w, dt).to(DEV)
        try:
            l_hx = ops.conv2d(x, wp2, y, bias, korder=2); ops.gemm_plan2(l_hx)
        except Exception:
            l_hx = None
        l_dma = ops.conv2d(x, wp0, y, bias)
        h = cin // 2
        l_reg = ops.conv2d(x[..., :h], wp0, y2, bias, x2=x[..., h:])
        part = torch.empty(B * ops.GN_MAX_CHUNKS * 64, dtype=torch.float64, device=DEV)
        ls, n = ops.groupnorm_stats(x, part)
        ls()
        xn = torch.empty_like(x)
        la = ops.groupnorm_apply(x, torch.ones(cin, device=DEV), torch.zeros(cin, device=DEV), xn, part, n, eps=1e-5, silu=True)
        l_dma(); l_reg(); torch.cuda.synchronize()
        d = (y.float() - y2.float()).abs().max().item()
        t_hx = timeit(l_hx) if l_hx is not None else float("nan")
        t_dma, t_reg, t_app = timeit(l_dma), timeit(l_reg), timeit(la)
        fl = 2.0 * B * hw * hw * co * 9 * cin
        print(f"{B} x {hw}x{hw} {cin} -> {co}: LDS-DMA + row-extended A {t_hx:6.1f} us | LDS-DMA {t_dma:6.1f} us ({fl / t_dma / 1e6:.0f} TF) | register-staged {t_reg:6.1f} us ({fl / t_reg / 1e6:.0f} TF, max |d| {d:.1e}) | "
              f"the normalisation pass of its input {t_app:5.1f} us", flush=True)
