#!/usr/bin/env python3
"""Diagnostics: cost of the epilogue's global access PATTERN (tools/store_probe.hip): row-per-lane 16-byte pieces (what the direct epilogue of
gemm.hip issues) against pair- / quad- / line-coalesced forms of the same bytes.  Prints us and TB/s per mode for stores and loads."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libstoreprobe.so"))
lib.store_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda:0"
torch.cuda.set_device(0)
sink = torch.zeros(16, dtype=torch.int32, device=dev)
for (M, N, RB, tag) in ((65536, 320, 640, "65536 x 320 bf16, wave block 32 x 320 cols"), (65536, 320, 320, "65536 x 320 bf16, wave block 32 x 160 cols (two passes)"),
                        (65536, 1280, 640, "65536 x 1280 bf16 (GEGLU out), 32 x 320 per wave"), (16384, 640, 320, "16384 x 640 bf16")):
    buf = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
    ld = N * 2
    nbytes = M * RB * (N * 2 // RB if False else 1)
    for op, name in ((0, "store"), (1, "load ")):
        for mode in (0, 1, 2, 3):
            ts = []
            for rep in range(7):
                # column passes: cover the whole row width in RB-byte strips (each launch = one strip; time them together)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for strip in range(ld // RB):
                    rc = lib.store_probe(mode, op, buf.data_ptr() + strip * RB, ld, M, RB, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
                    assert rc == 0, rc
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            t = sorted(ts)[len(ts) // 2]
            print(f"{tag:58s} {name} mode {mode}: {t * 1e3:8.1f} us  {M * ld / t / 1e9:7.2f} TB/s", flush=True)
