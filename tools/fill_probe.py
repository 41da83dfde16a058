#!/usr/bin/env python3
"""Diagnostics: per-CU operand staging rate by path (tools/fill_probe.hip).  Prints B/clk/CU (at the nominal 2.4 GHz) and TB/s chip-wide for
LDS-DMA pieces, plain register loads and register-staged tiles, by row pitch, depth in flight and how the source region is shared."""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libfillprobe.so"))
lib.fill_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                           ctypes.c_int, ctypes.c_void_p]
dev = "cuda:0"
torch.cuda.set_device(0)
sink = torch.zeros(16, dtype=torch.int32, device=dev)
buf = torch.randint(0, 2 ** 31 - 1, (256 * 1024 * 1024 // 4,), dtype=torch.int32, device=dev)          # 256 MB
NAMES = {0: "lds-dma", 1: "regs   ", 2: "regs+ds"}


def run(mode, pieces, depth, block_stride, region_rows, ld, blocks, tag):
    rows_per_iter = 64 * pieces
    iters = max(8, (64 << 20) // (rows_per_iter * 128 * 1))          # ~64 MB per block
    iters = min(iters, 4096)
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.fill_probe(mode, pieces, depth, buf.data_ptr(), block_stride, region_rows, ld, iters, sink.data_ptr(), blocks, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[len(ts) // 2] * 1e-3
    nbytes = blocks * iters * rows_per_iter * 128
    cus = min(blocks, 256)
    print(f"{tag:44s} {NAMES[mode]} pieces {pieces} depth {depth} ld {ld:5d} blocks {blocks:4d}: {nbytes / t / 1e12:6.2f} TB/s  {nbytes / t / cus / 2.4e9:6.1f} B/clk/CU", flush=True)


for ld in (128, 640, 2560):
    for (bs, rr, tag) in ((0, (2 << 20) // ld, "all blocks share one 2 MB region (L2 hits)"),
                          (64 << 10, (64 << 10) // ld, "own 64 KB region per block (L2 resident)"),
                          (512 << 10, (512 << 10) // ld, "own 512 KB region per block (MALL)")):
        for (m, p, d) in ((0, 4, 1), (0, 4, 2), (0, 4, 4), (0, 9, 1), (0, 9, 2), (0, 2, 8), (1, 4, 1), (1, 9, 2), (2, 4, 1)):
            if rr < 64 * p:
                continue
            run(m, p, d, bs, rr, ld, 256, tag)
    print()
# two blocks per CU (LDS permitting) and half the chip
run(0, 4, 2, 0, (2 << 20) // 640, 640, 512, "2 blocks per CU, shared 2 MB")
run(0, 4, 4, 0, (2 << 20) // 640, 640, 512, "2 blocks per CU, shared 2 MB")
run(0, 4, 2, 0, (2 << 20) // 640, 640, 128, "half the CUs, shared 2 MB")
run(0, 4, 2, 0, (2 << 20) // 640, 640, 32, "32 blocks, shared 2 MB")
