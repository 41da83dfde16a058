#!/usr/bin/env python3
"""Cycles per {1 MFMA + 4 exponentials} group by how the exponentials are evaluated (tools/exp_probe.hip): v_exp_f32 against a range-reduced
degree-3 polynomial on the plain / packed VALU, with one and two waves per SIMD, beside the MFMAs and alone."""
import ctypes
import os
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libexpprobe.so"))
iters = 2000
buf = (ctypes.c_longlong * 20)()
assert lib.exp_probe(iters, buf) == 0
names = ["4 x v_exp_f32", "4 x polynomial (7 plain VALU each)", "2 x v_exp + 2 x polynomial", "4 values on packed polynomial (v_pk_fma_f32)", "2 x v_exp + 2 packed-polynomial values"]
k = 0
for waves, mf in ((1, 1), (1, 0), (2, 1), (2, 0)):
    print(f"--- {waves} wave(s) per SIMD, {'beside 1 v_mfma_f32_32x32x16_bf16 per group' if mf else 'VALU work alone'} (cycles per group per wave; two waves share the SIMD)")
    for n in names:
        print(f"   {n:48s} {buf[k] / (iters * 4):7.1f}")
        k += 1
