"""Test infrastructure: OCP e4m3fn (bias 7, no infinities, S.1111.111 = NaN) written from the format's definition in numpy -- the independent
reference of rf_quantize_fp8_rows / rf_quantize_fp8_act (tests/test_ops_gpu.py), itself checked on the CPU (tests/test_host_cpu.py)."""
import numpy as np


def e4m3fn_decode_np(code):
    """OCP e4m3fn byte -> float64, from the format's definition (bias 7, no infinities, S.1111.111 = NaN): independent of torch's float8 type."""
    code = np.asarray(code, dtype=np.int64)
    s, e, m = (code >> 7) & 1, (code >> 3) & 15, code & 7
    v = np.where(e == 0, m * 2.0 ** -9, (8 + m) * 2.0 ** (e.astype(np.float64) - 10))
    return np.where(s == 1, -v, v)


def e4m3fn_encode_rne_sat_np(x):
    """float -> OCP e4m3fn byte: round to nearest, ties to the even mantissa, saturating at +-448 (what v_cvt_pk_fp8_f32 does for finite input).
    Integer / exact-power-of-two arithmetic in float64 only (every step is exact for float32 input)."""
    x = np.asarray(x, dtype=np.float64)
    a = np.abs(x)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -30)))
    e = np.clip(e, -6, 8)                                     # subnormals share the exponent of the smallest normal (quantum 2^-9)
    quantum = 2.0 ** (e - 3)
    qn = np.rint(a / quantum)                                 # np.rint: ties to even -- a / quantum is exact (power-of-two divisor)
    val = np.minimum(qn * quantum, 448.0)                     # (qn = 16 carries into the next binade by itself; beyond 448: saturate)
    ee = np.floor(np.log2(np.maximum(val, 2.0 ** -30)))
    sub = val < 2.0 ** -6
    expf = np.where(sub, 0, ee + 7).astype(np.int64)
    man = np.where(sub, val / 2.0 ** -9, val / 2.0 ** (ee - 3) - 8).astype(np.int64)
    return ((np.signbit(x).astype(np.int64) << 7) | (expf << 3) | man).astype(np.uint8)
