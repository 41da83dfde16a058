#!/usr/bin/env python3
"""rf_conv_gemm against the vendor GEMM (hipBLASLt through torch.nn.functional.linear) on the long-K shapes that make up most of a
DDIM step, under rocprofv3: the same plain bf16 GEMM  out[M, N] = x[M, K] W[N, K]^T + b  through both, `--reps` launches each, warm operands.

  rocprofv3 --kernel-trace --stats ...      -> durations, grid / workgroup / LDS / register footprint of both kernels
  rocprofv3 --pmc <set> ...                 -> instruction and wait counters per launch (one pass per counter set: tools/archive/run_r06b.sh)
  python tools/vendor_pmc.py --join DIR...  -> one table: counters per launch and per 1024 MFMA-FLOP-equivalents, ours vs the vendor's

Run with --which ours|vendor|both so that a PMC pass can be attributed by kernel name (hipBLASLt's kernels are named Cijk_*)."""
import argparse
import collections
import csv
import glob
import os
import sys

SHAPES = [(16384, 640, 5760), (4096, 1280, 11520), (65536, 320, 2880)]


def run(args):
    import torch
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    from reface_amd import ops
    F = torch.nn.functional
    dev = "cuda"
    for M, N, K in SHAPES:
        if args.shape and f"{M}x{N}x{K}" != args.shape:
            continue
        x = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        l = ops.linear(x, w, out, b)
        pl = ops.gemm_plan2(l)
        bb = b.bfloat16()
        for which in (("ours", "vendor") if args.which == "both" else (args.which,)):
            fn = (lambda: l()) if which == "ours" else (lambda: F.linear(x, w, bb))
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.reps):
                fn()
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) / args.reps * 1e3
            print(f"{M}x{N}x{K} {which:6s}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TFLOP/s" + (f"   plan {pl}" if which == "ours" else ""), flush=True)


def join(dirs):
    tot = collections.defaultdict(collections.Counter)
    n = collections.defaultdict(collections.Counter)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                key = "ours: " + k.split("<")[0][-40:] + ("+reduce" if "splitk_reduce" in k else "") if "rf::" in k else ("vendor: " + k[:60] if k.startswith("Cijk") else None)
                if key is None:
                    continue
                tot[key][r["Counter_Name"]] += float(r["Counter_Value"])
                n[key][r["Counter_Name"]] += 1
    names = sorted({c for v in tot.values() for c in v})
    keys = sorted(tot)
    print(f"{'counter (mean per launch)':34s}" + "".join(f"{k[:44]:>46s}" for k in keys))
    for c in names:
        print(f"{c:34s}" + "".join(f"{(tot[k][c] / n[k][c]) if n[k][c] else float('nan'):46.0f}" for k in keys))
    print("launches counted: " + ", ".join(f"{k[:30]}={max(n[k].values())}" for k in keys))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="both", choices=["ours", "vendor", "both"])
    ap.add_argument("--shape", default="")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--join", nargs="*", default=None)
    a = ap.parse_args()
    if a.join is not None:
        join(a.join)
    else:
        run(a)
