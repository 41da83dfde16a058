#!/usr/bin/env python3
"""Micro-benchmark of rf_conv_gemm / rf_attention / norms on the UNet's representative shapes (CFG batch 16, 512x512 images).
Usage: python tools/bench_gemm.py [--dtype bf16|f32] [--only substr] [--reps N] [--json out.json]"""
import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reface_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--only", default="")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--bc", type=int, default=16)
ap.add_argument("--json", default=None)
ap.add_argument("--sustain", type=float, default=0.0, help="also report the rate sustained over this many seconds (power-managed clocks)")
ap.add_argument("--korder", type=int, default=0)
ap.add_argument("--cold", type=int, default=0, help="time every launch alone behind a pass over a 1 GB buffer (operands come from HBM, as inside the step)")
ap.add_argument("--gn", type=int, default=0, help="conv cases also emit the fused GroupNorm statistics of their output")
ap.add_argument("--vendor", type=int, default=0, help="also time the SAME operation through PyTorch-ROCm's vendor libraries (hipBLASLt linear, MIOpen "
                "channels-last conv2d, the SDPA flash kernel, native group_norm) on the same tensors: what the reference's own modules would launch on this GPU")
args = ap.parse_args()
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
if args.vendor >= 2:
    torch.backends.cudnn.benchmark = True          # MIOpen searches its solvers per shape (the reference's scripts leave this off)
dev = "cuda"
Bc = args.bc
KORDER = args.korder


def r(*shape, scale=1.0, dtype=None):
    return (torch.randn(*shape, device=dev) * scale).to(dtype or dt)


cases = []
vendor = {}
F = torch.nn.functional


def conv(name, cin, cout, hw, *, c1=0, stride=1, ups=0):
    hin = hw // 2 if ups else hw
    ho = hw // stride
    x = r(Bc, hin, hin, cin + c1)       # the UNet's skip concat is ONE buffer (zero-copy), so a single source
    x2, cin, c1 = None, cin + c1, 0
    w = r(cout, 9 * (cin + c1), scale=1 / math.sqrt(9 * (cin + c1)))
    out = torch.empty(Bc, ho, ho, cout, device=dev, dtype=dt)
    res = r(Bc, ho, ho, cout)
    b = r(cout, dtype=torch.float32)
    l = ops.conv2d(x, w, out, b, stride=stride, ups=ups, x2=x2, residual=res, korder=KORDER, name=name)
    if args.gn:
        assert ops.fuse_groupnorm_stats(out, [(l, 0, Bc * ho * ho, 0, cout)]) is not None, name
    cases.append((name, l, 2.0 * Bc * ho * ho * cout * 9 * (cin + c1)))
    if args.vendor:
        xv = x.permute(0, 3, 1, 2)                                                     # NCHW view of the NHWC buffer = channels_last
        wv = w.view(cout, 3, 3, cin).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        bv, rv = b.to(dt), res.permute(0, 3, 1, 2)

        def run(xv=xv, wv=wv, bv=bv, rv=rv, stride=stride, ups=ups):
            xi = F.interpolate(xv, scale_factor=2.0, mode="nearest") if ups else xv
            return F.conv2d(xi, wv, bv, stride=stride, padding=1) + rv
        vendor[name] = run


def lin(name, M, N, K, act=ops.ACT_NONE, res=False):
    x = r(M, K)
    w = r(N, K, scale=1 / math.sqrt(K))
    b = r(N, dtype=torch.float32)
    out = torch.empty(M, N // 2 if act == ops.ACT_GEGLU else N, device=dev, dtype=dt)
    rs = r(M, N) if res else None
    cases.append((name, ops.linear(x, w, out, b, act=act, residual=rs, name=name), 2.0 * M * N * K))
    if args.vendor:
        bv = b.to(dt)

        def run(x=x, w=w, bv=bv, rs=rs, act=act):
            y = F.linear(x, w, bv)
            if act == ops.ACT_GEGLU:
                a, g = y.chunk(2, dim=-1)
                return a * F.gelu(g)
            return y + rs if rs is not None else y
        vendor[name] = run


def attn(name, heads, d, N):
    c = heads * d
    qkv = r(Bc, N, 3 * c)
    out = torch.empty(Bc, N, c, device=dev, dtype=dt)
    cases.append((name, ops.attention(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], out, heads=heads, scale=d ** -0.5, name=name),
                  4.0 * Bc * heads * N * N * d))
    if args.vendor:
        q, k, v = (qkv[..., i * c:(i + 1) * c].reshape(Bc, N, heads, d).transpose(1, 2) for i in range(3))
        vendor[name] = lambda q=q, k=k, v=v: F.scaled_dot_product_attention(q, k, v)


def gn(name, c, hw):
    x = r(Bc, hw, hw, c)
    out = torch.empty_like(x)
    part = torch.empty(Bc * ops.GN_MAX_CHUNKS * 64, dtype=torch.float64, device=dev)
    g, b = r(c, dtype=torch.float32), r(c, dtype=torch.float32)
    a, bb = ops.groupnorm(x, g, b, out, part, eps=1e-5, silu=True, name=name)
    nbytes = x.numel() * x.element_size()
    cases.append((name + ".stats", a, -float(nbytes)))
    cases.append((name + ".apply", bb, -2.0 * nbytes))
    if args.vendor:
        xv, gv, bv = x.permute(0, 3, 1, 2), g.to(dt), b.to(dt)
        vendor[name + ".apply"] = lambda xv=xv, gv=gv, bv=bv: F.silu(F.group_norm(xv, 32, gv, bv, 1e-5))        # (statistics + apply + SiLU together)


def ln(name, M, c):
    x = r(M, c)
    out = torch.empty_like(x)
    g, b = r(c, dtype=torch.float32), r(c, dtype=torch.float32)
    cases.append((name, ops.layernorm(x, g, b, out, name=name), -2.0 * x.numel() * x.element_size()))


def ffn(name, M, C):
    x = r(M, C)
    w1 = r(8 * C, C, scale=1 / math.sqrt(C))
    b1 = r(8 * C, dtype=torch.float32)
    w2 = r(C, 4 * C, scale=1 / math.sqrt(4 * C))
    b2 = r(C, dtype=torch.float32)
    res = r(M, C)
    out = torch.empty(M, C, device=dev, dtype=dt)
    w1p, b1p = ops.pack_geglu(w1.float(), b1, dt)
    cases.append((name, ops.ffn_geglu(x, w1p, b1p, ops.pack_ffn_w2(w2, dt), b2, out, residual=res, name=name), 2.0 * M * C * 8 * C + 2.0 * M * 4 * C * C))


if dt == torch.bfloat16:
    ffn("ffn fused 320 @64", Bc * 4096, 320)
conv("conv3x3 320->320 @64", 320, 320, 64)
conv("conv3x3 640->640 @32", 640, 640, 32)
conv("conv3x3 1280->1280 @16", 1280, 1280, 16)
conv("conv3x3 1280->1280 @8", 1280, 1280, 8)
conv("conv3x3 cat2560->1280 @8", 1280, 1280, 8, c1=1280)
conv("conv3x3 cat1920->640 @32", 1280, 640, 32, c1=640)
conv("conv3x3 cat960->320 @64", 640, 320, 64, c1=320)
conv("conv3x3 cat640->320 @64", 320, 320, 64, c1=320)
conv("down 320 @64->32", 320, 320, 64, stride=2)
conv("up 640 @32->64", 640, 640, 64, ups=1)
lin("geglu 320->2560 @64", Bc * 4096, 2560, 320, ops.ACT_GEGLU)
lin("ff2 1280->320 @64", Bc * 4096, 320, 1280)
lin("qkv 320->960 @64", Bc * 4096, 960, 320)
lin("proj 320->320 @64", Bc * 4096, 320, 320)
lin("proj+res 320->320 @64", Bc * 4096, 320, 320, res=True)
lin("proj+res 640->640 @32", Bc * 1024, 640, 640, res=True)
lin("proj+res 1280->1280 @16", Bc * 256, 1280, 1280, res=True)
lin("ff2+res 1280->320 @64", Bc * 4096, 320, 1280, res=True)
lin("geglu 640->5120 @32", Bc * 1024, 5120, 640, ops.ACT_GEGLU)
lin("ff2 2560->640 @32", Bc * 1024, 640, 2560)
lin("geglu 1280->10240 @16", Bc * 256, 10240, 1280, ops.ACT_GEGLU)
lin("ff2 5120->1280 @16", Bc * 256, 1280, 5120)
lin("skip1x1 2560->1280 @8", Bc * 64, 1280, 2560)
lin("proj 1280->1280 @8", Bc * 64, 1280, 1280)
lin("qkv 1280->3840 @8", Bc * 64, 3840, 1280)
lin("ff2 5120->1280 @8", Bc * 64, 1280, 5120)
lin("lin big 5760->320 @64", Bc * 4096, 320, 5760)
lin("lin big 5760->640 @32", Bc * 1024, 640, 5760)
lin("proj 640->640 @32", Bc * 1024, 640, 640)
lin("proj 1280->1280 @16", Bc * 256, 1280, 1280)
attn("attn d40 N4096", 8, 40, 4096)
attn("attn d80 N1024", 8, 80, 1024)
attn("attn d160 N256", 8, 160, 256)
attn("attn d160 N64", 8, 160, 64)
gn("gn 320 @64", 320, 64)
gn("gn 640 @64", 640, 64)
gn("gn 1280 @16", 1280, 16)
ln("ln 320 @64", Bc * 4096, 320)

results = []
stream = torch.cuda.current_stream()
for name, l, work in cases:
    if args.only and args.only not in name:
        continue
    for _ in range(3):
        l()
    if args.cold:
        flush = globals().setdefault("_flush", torch.zeros(256 << 20, dtype=torch.float32, device=dev))
        ts, gaps = [], []
        for _ in range(args.reps):
            flush.add_(1.0)
            s, e, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            s.record(stream)
            l()
            e.record(stream)
            e2.record(stream)
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
            gaps.append(e.elapsed_time(e2) * 1e3)
        us = sorted(ts)[len(ts) // 2] - sorted(gaps)[len(gaps) // 2]
        rate, unit = (work / us / 1e6, "TFLOP/s") if work > 0 else (-work / us / 1e3, "GB/s")
        results.append(dict(name=name, us=us, rate=rate, unit=unit, cold=True))
        print(f"{name:32s} {us:10.1f} us  {rate:9.1f} {unit}   (cold)", flush=True)
        continue
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record(stream)
    for _ in range(args.reps):
        l()
    e.record(stream)
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / args.reps * 1e3
    if work > 0:
        rate, unit = work / us / 1e6, "TFLOP/s"
    else:
        rate, unit = -work / us / 1e3, "GB/s"
    sus = ""
    if args.sustain > 0:
        n = max(10, int(args.sustain * 1e6 / us))
        for _ in range(n // 2):
            l()
        s.record(stream)
        for _ in range(n // 2):
            l()
        e.record(stream)
        torch.cuda.synchronize()
        us2 = s.elapsed_time(e) / (n // 2) * 1e3
        sus = f"   sustained {us2:10.1f} us  {abs(work) / us2 / (1e6 if work > 0 else 1e3):9.1f}"
    ven = ""
    if name in vendor:
        fn = vendor[name]
        try:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            s.record(stream)
            for _ in range(args.reps):
                fn()
            e.record(stream)
            torch.cuda.synchronize()
            vus = s.elapsed_time(e) / args.reps * 1e3
            ven = f"   | vendor {vus:10.1f} us  {abs(work) / vus / (1e6 if work > 0 else 1e3):9.1f}  (ours / vendor time {us / vus:5.2f})"
            results.append(dict(name=name + " [vendor]", us=vus, rate=abs(work) / vus / (1e6 if work > 0 else 1e3), unit=unit))
        except Exception as ex:          # a shape the vendor kernel does not take
            ven = f"   | vendor: {type(ex).__name__}: {str(ex)[:80]}"
    results.append(dict(name=name, us=us, rate=rate, unit=unit))
    print(f"{name:32s} {us:10.1f} us  {rate:9.1f} {unit}{sus}{ven}", flush=True)
if args.json:
    json.dump(results, open(args.json, "w"), indent=1)
