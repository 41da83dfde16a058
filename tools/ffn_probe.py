#!/usr/bin/env python3
"""Time the fused feed-forward kernel alone (C = 320, M = 65536 tokens = the 64x64 level of configs[1]) under the library REFACE_HIP_LIB names:
plain (rf_ffn_geglu shape) and the token-resident tail (rf_ffn_block with proj_out).  Prints us / launch and TFLOP/s.  Diagnostic only."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reface_amd import ops
DEV = "cuda:0"
def rnd(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g)
def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    dt, C = torch.bfloat16, 320
    x = (rnd((M, C), 1) * 1.3).to(dt).to(DEV)
    xin = rnd((M, C), 2).to(dt).to(DEV)
    w1 = rnd((8 * C, C), 3) / math.sqrt(C); b1 = rnd((8 * C,), 4) * 0.5
    w2 = rnd((C, 4 * C), 5) / math.sqrt(4 * C); b2 = rnd((C,), 6)
    wpo = (rnd((C, C), 7) / math.sqrt(C)).to(dt).to(DEV); bpo = rnd((C,), 8).to(DEV)
    w1f, b1f = ops.fold_layernorm_geglu(w1, b1, torch.ones(C), torch.zeros(C))
    w1p, b1p = ops.pack_geglu(w1f, b1f, dt)
    w1p, b1p, w2q, b2 = w1p.to(DEV), b1p.to(DEV), ops.pack_ffn_w2(w2.to(DEV), dt), b2.to(DEV)
    y = torch.empty((M, C), dtype=dt, device=DEV)
    plain = ops.ffn_geglu(x, w1p, b1p, w2q, b2, y, residual=x, ln_eps=1e-5)
    tail = ops.ffn_block(x, w1p, b1p, w2q, b2, y, residual=x, wpo=wpo, bpo=bpo, res2=xin, res2_rows=0, ln_eps=1e-5)
    for name, fn, fl in (("plain", plain, 2.0 * M * C * 12 * C), ("tail", tail, 2.0 * M * C * 13 * C)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(5):
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        print(f"{os.path.basename(os.environ.get('REFACE_HIP_LIB', 'in-tree')):14s} {name:6s} M {M}  {best:8.1f} us  {fl / best / 1e6:7.1f} TFLOP/s", flush=True)
main()
