"""Cycles per {1 MFMA + N VALU} group for one wave per SIMD (tools/mfma_valu_probe.hip): is VALU work free in the MFMA's shadow?"""
import ctypes, os
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libmfmaprobe.so"))
iters = 2000
names = {0: "v_fma_f32", 1: "v_exp_f32", 2: "v_max3_f32", 3: "v_cvt_pk_bf16_f32"}
for acc in (0, 1):
    for kind in (0, 1, 2, 3):
        buf = (ctypes.c_longlong * 13)()
        rc = lib.mfma_valu_probe(kind, acc, iters, buf)
        per = [buf[n] / (iters * 4) for n in range(13)]
        print(f"acc in {'AGPR' if acc else 'VGPR'}  filler {names[kind]:18s} cycles per (MFMA + N fillers), N = 0..12 (s_memtime ticks): " + " ".join(f"{p:6.1f}" for p in per))

buf = (ctypes.c_longlong * 6)()
lib.mfma_chain_probe(iters, buf)
per = [buf[n] / (iters * 4) for n in range(6)]
print("dependent chains (accumulators in VGPRs): cycles per MFMA with 1 / 2 / 4 accumulators round-robin, no fillers: %.1f %.1f %.1f ; with 4 v_fma fillers: %.1f %.1f %.1f" % tuple(per))
