#!/usr/bin/env python3
"""Which hipBLASLt kernels (macro tile, split) PyTorch-ROCm picks for the UNet's GEMM shapes: run under rocprofv3 --kernel-trace --stats."""
import torch
F = torch.nn.functional
dev = "cuda"
shapes = [(16384, 640, 5760), (16384, 640, 2560), (4096, 1280, 11520), (4096, 1280, 5120), (1024, 1280, 11520), (1024, 3840, 1280), (1024, 1280, 5120),
          (65536, 320, 2880), (65536, 960, 320), (65536, 320, 320), (16384, 1920, 640), (4096, 10240, 1280)]
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev).bfloat16()
    for _ in range(3):
        F.linear(x, w, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10 + (M * N * K) % 7):          # a distinct call count per shape: identifies the shape in the stats table
        F.linear(x, w, b)
    e.record()
    torch.cuda.synchronize()
    n = 10 + (M * N * K) % 7
    us = s.elapsed_time(e) / n * 1e3
    print(f"{M}x{N}x{K}: {n + 3} calls, {us:.1f} us, {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s", flush=True)
