#!/usr/bin/env python3
"""Experiment: GEGLU -> ff.net.2 at the 64x64 level, whole (M = 65536) vs in row chunks whose hidden tensor stays cache-resident."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reface_amd import ops
dev, dt = "cuda", torch.bfloat16
M, C = 65536, 320
x = (torch.randn(M, C, device=dev)).to(dt)
x1 = (torch.randn(M, C, device=dev)).to(dt)
w1 = (torch.randn(8 * C, C, device=dev) / math.sqrt(C)).to(dt)
b1 = torch.randn(8 * C, device=dev)
w2 = (torch.randn(C, 4 * C, device=dev) / math.sqrt(4 * C)).to(dt)
b2 = torch.randn(C, device=dev)
hid = torch.empty(M, 4 * C, device=dev, dtype=dt)
out = torch.empty(M, C, device=dev, dtype=dt)


def build(nchunk):
    ls = []
    r = M // nchunk
    for i in range(nchunk):
        s = slice(i * r, (i + 1) * r)
        ls.append(ops.linear(x[s], w1, hid[s], b1, act=ops.ACT_GEGLU))
        ls.append(ops.linear(hid[s], w2, out[s], b2, residual=x1[s]))
    return ls


def build_reuse(nchunk):
    """chunks write the SAME small hidden buffer (stays in L2 / MALL, never needs to reach HBM)"""
    ls = []
    r = M // nchunk
    for i in range(nchunk):
        s = slice(i * r, (i + 1) * r)
        ls.append(ops.linear(x[s], w1, hid[:r], b1, act=ops.ACT_GEGLU))
        ls.append(ops.linear(hid[:r], w2, out[s], b2, residual=x1[s]))
    return ls


for name, ls in (("whole", build(1)), ("2 chunks", build(2)), ("4 chunks", build(4)), ("8 chunks", build(8)), ("4 chunks, one buffer", build_reuse(4)),
                 ("8 chunks, one buffer", build_reuse(8)), ("16 chunks, one buffer", build_reuse(16))):
    g = torch.cuda.CUDAGraph()
    ops.run(ls)
    s_ = torch.cuda.Stream()
    with torch.cuda.stream(s_):
        ops.run(ls)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        ops.run(ls)
    for _ in range(3):
        g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20):
        g.replay()
    b.record(); torch.cuda.synchronize()
    print(f"{name:24s} {a.elapsed_time(b) / 20 * 1e3:8.1f} us")
