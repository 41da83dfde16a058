"""Thin Python launchers over the C-ABI (reface_amd/_lib.py).

PyTorch is used for device memory and the current HIP stream only; every arithmetic op of the
hot path is a HIP kernel from libreface_hip.so.  Activations are channels-last torch tensors
([B, H, W, C] or [M, C]) in fp32, bf16 or fp16 (H16: the two 16-bit storage types run the same kernels); biases / norm affine parameters stay fp32.
"""
import ctypes as C
import threading

import torch

from . import _lib
from ._lib import (ACT_GEGLU, ACT_GELU, ACT_NONE, ACT_PRELU, ACT_QUICK_GELU, ACT_RELU, ACT_SIGMOID, ACT_SILU, RF_BF16, RF_BF16X3, RF_F16, RF_F32,
                   RF_FP8_E4M3, AttnInDesc, ConvGemmDesc, FfnDesc, StemDesc)

# Attention scores in the exp2 domain: the UNet folds d^-0.5 * log2(e) into the to_q weights and calls rf_attention with scale = ln 2
# (the kernels then multiply by exactly 1: no second rounding of q * scale to bf16 in the pipelined d = 40 kernel).
LOG2E = 1.4426950408889634
LN2 = 0.6931471805599453


def code(dt):
    if dt == torch.float32:
        return RF_F32
    if dt == torch.bfloat16:
        return RF_BF16
    if dt == torch.float16:
        return RF_F16
    raise TypeError(f"unsupported dtype {dt}")


H16 = (torch.bfloat16, torch.float16)


def vec(dt):
    return 4 if dt == torch.float32 else 8


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _require_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.RefaceHipError("reface_amd ops need device tensors (no CPU fallback)")


# ------------------------------------------------------------------------------------------------
# weight packing (done once at load time)
# ------------------------------------------------------------------------------------------------
def conv_korder(cin, dtype, ksize=3):
    """1 when the channel-chunk-major K order applies (input channels a multiple of the K tile)."""
    return 1 if (ksize > 1 and cin % (8 * vec(dtype)) == 0) else 0


def pack_conv_weight(w, dtype, cin_pad=None, korder=0):
    """[Cout, Cin, KH, KW] -> [Cout, KH*KW*Cin_pad].
    korder 0: k = (ky*KW + kx)*Cin_pad + c.   korder 1: k = ((c // BK)*KH*KW + tap)*BK + c % BK (BK = K-tile elements).
    korder 2: k = ((ky*(Cin_pad // BK) + c // BK)*KW + kx)*BK + c % BK -- filter row, channel chunk, filter column: the three horizontal taps of a
    (row, chunk) are consecutive K tiles and share one row-extended A tile in rf_conv_gemm (3x3, stride 1)."""
    co, ci, kh, kw = w.shape
    cp = ci if cin_pad is None else cin_pad
    wp = torch.zeros((co, kh, kw, cp), dtype=torch.float32, device=w.device)
    wp[..., :ci] = w.float().permute(0, 2, 3, 1)
    if korder == 2:
        bk = 8 * vec(dtype)
        assert cp % bk == 0
        wp = wp.reshape(co, kh, kw, cp // bk, bk).permute(0, 1, 3, 2, 4)
    elif korder:
        bk = 8 * vec(dtype)
        assert cp % bk == 0
        wp = wp.reshape(co, kh * kw, cp // bk, bk).permute(0, 2, 1, 3)
    return wp.reshape(co, kh * kw * cp).to(dtype).contiguous()


def pack_x3(w2d):
    """fp32 GEMM weight [N, K] (K a multiple of 64, already in rf_conv_gemm's K order) -> the split-bf16 operand of the RF_BF16X3 mode:
    [N, 3K] bf16, per 64-element K tile [64 hi | 64 lo | 64 hi] with hi = bf16(w), lo = bf16(w - hi)."""
    n, k = w2d.shape
    assert k % 64 == 0, k
    w = w2d.float()
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    h3, l3 = hi.reshape(n, k // 64, 64), lo.reshape(n, k // 64, 64)
    return torch.stack([h3, l3, h3], dim=2).reshape(n, 3 * k).contiguous()


def x3_eligible(K, cin=None):
    """The split-bf16 mode runs on the direct-to-LDS main loop: K (and the channel count of a convolution) multiples of 64."""
    return K % 64 == 0 and (cin is None or cin % 64 == 0)


def split_bf16(x, out, name="split_bf16"):
    """x fp32 [..., C] (uniform pixel pitch) -> out bf16 [..., 2C] = [hi | lo] per pixel (rf_split_bf16)."""
    lib = _lib.load()
    _require_gpu(x, out)
    Cc = x.shape[-1]
    M = x.numel() // Cc
    assert x.dtype == torch.float32 and out.dtype == torch.bfloat16 and out.shape[-1] == 2 * Cc and x.stride(-1) == 1 and out.stride(-1) == 1
    return Launch(lib.rf_split_bf16, (_p(x), M, Cc, x.stride(-2), _p(out), out.stride(-2)), (x, out), name)


def pack_geglu(w, b, dtype):
    """GEGLU projection [2F, C] (value rows 0..F-1, gate rows F..2F-1) -> rows interleaved in blocks
    of 32 (value block, gate block) so one MFMA wave tile holds matching value/gate columns."""
    f2, c = w.shape
    f = f2 // 2
    assert f % 32 == 0
    wv, wg = w[:f].reshape(f // 32, 32, c), w[f:].reshape(f // 32, 32, c)
    wp = torch.stack([wv, wg], dim=1).reshape(f2, c)
    bv, bg = b[:f].reshape(f // 32, 32), b[f:].reshape(f // 32, 32)
    bp = torch.stack([bv, bg], dim=1).reshape(f2)
    return wp.to(dtype).contiguous(), bp.float().contiguous()


FFN_W2_PERM = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)


def pack_ffn_w2(w2, dtype=torch.bfloat16):
    """ff.net.2 weight [C, 4C] -> the hidden-column order rf_ffn_geglu multiplies in (every 16-group: 0-3, 8-11, 4-7, 12-15: the K order a
    register-fed MFMA B operand implies)."""
    c, f = w2.shape
    idx = (torch.arange(f // 16)[:, None] * 16 + torch.tensor(FFN_W2_PERM)[None]).reshape(-1).to(w2.device)
    return w2[:, idx].to(dtype).contiguous()


def fold_layernorm_geglu(w1, b1, gamma, beta):
    """LayerNorm's affine folded into the GEGLU projection behind it:  (xhat * gamma + beta) W1^T + b1 = xhat (W1 diag(gamma))^T + (b1 + W1 beta).
    w1 [2F, C], b1 [2F], gamma / beta [C] (reference layout, fp32) -> (w1', b1') for pack_geglu; the kernel then only normalises (ln_eps)."""
    w1f = w1.float()
    return w1f * gamma.float()[None, :], b1.float() + w1f @ beta.float()


def ffn_geglu(x, w1p, b1p, w2q, b2, out, *, residual=None, ln_eps=0.0, name="ffn_geglu"):
    """out[M, C] = (GEGLU(x W1^T + b1)) W2^T + b2 (+ residual), one kernel, hidden tensor on chip (C = 320, bf16).  ln_eps > 0: the rows
    of x are LayerNorm-ed in registers first (no affine -- fold gamma / beta with fold_layernorm_geglu)."""
    lib = _lib.load()
    _require_gpu(x, w1p, b1p, w2q, b2, out, residual)
    M, Cc = x.shape
    assert x.dtype == w1p.dtype == w2q.dtype == out.dtype and x.dtype in H16 and b1p.dtype == b2.dtype == torch.float32
    assert w1p.shape == (8 * Cc, Cc) and w2q.shape == (Cc, 4 * Cc) and w1p.is_contiguous() and w2q.is_contiguous()
    assert x.stride(1) == 1 and out.stride(1) == 1 and (residual is None or (residual.stride(1) == 1 and residual.dtype == out.dtype))
    if x.dtype == torch.float16:          # the positional entry point is bf16; fp16 goes through the descriptor form with no projection behind it
        return ffn_block(x, w1p, b1p, w2q, b2, out, residual=residual, wpo=None, bpo=None, res2=None, ln_eps=ln_eps, name=name)
    return Launch(lib.rf_ffn_geglu, (_p(x), x.stride(0), _p(w1p), _p(b1p), _p(w2q), _p(b2), _p(residual),
                                     residual.stride(0) if residual is not None else 0, _p(out), out.stride(0), M, Cc, float(ln_eps)),
                  (x, w1p, b1p, w2q, b2, out, residual), name)


def ffn_block(x, w1p, b1p, w2q, b2, out, *, residual, wpo, bpo, res2, res2_rows=0, ln_eps=0.0, front=None, name="ffn_block"):
    """The token-resident tail of a SpatialTransformer block at C = 320 (rf_ffn_block): out[M, C] = ((GEGLU(LN?(x) W1^T + b1)) W2^T + b2 + residual)
    Wpo^T + bpo + res2[row % res2_rows] in one kernel.  x / residual / res2 / out: bf16 row-strided 2-D views; wpo [C, C] bf16 contiguous (plain rows).
    The launch can emit the GroupNorm statistics of `out` (fuse_groupnorm_stats accepts it as a producer: 128-row blocks, all C columns)."""
    lib = _lib.load()
    _require_gpu(x, w1p, b1p, w2q, b2, out, residual, wpo, bpo, res2)
    M, Cc = out.shape          # (with `front` and CFG sharing, x -- the attention's output -- has front_rows rows)
    assert x.shape == ((int(front.get("front_rows", 0)) or M) if front is not None else M, Cc)
    assert x.dtype == w1p.dtype == w2q.dtype == out.dtype and x.dtype in H16 and b1p.dtype == b2.dtype == torch.float32
    assert wpo is None or (wpo.dtype == x.dtype and bpo.dtype == torch.float32 and wpo.shape == (Cc, Cc) and wpo.is_contiguous())
    assert w1p.shape == (8 * Cc, Cc) and w2q.shape == (Cc, 4 * Cc) and w1p.is_contiguous() and w2q.is_contiguous()
    assert x.stride(1) == 1 and out.stride(1) == 1 and out.shape == (M, Cc)
    assert residual is None or (residual.stride(1) == 1 and residual.dtype == out.dtype and residual.shape == (M, Cc))
    assert res2 is None or (res2.stride(1) == 1 and res2.dtype == out.dtype and res2.shape == ((res2_rows or M), Cc))
    d = FfnDesc()
    d.x, d.ldx, d.w1p, d.b1p, d.w2q, d.b2 = _p(x), x.stride(0), _p(w1p), _p(b1p), _p(w2q), _p(b2)
    d.residual, d.ldr = _p(residual), (residual.stride(0) if residual is not None else 0)
    d.out, d.ldo, d.M, d.C, d.ln_eps = _p(out), out.stride(0), M, Cc, float(ln_eps)
    d.wpo, d.bpo = _p(wpo), _p(bpo)
    d.res2, d.ldr2, d.res2_rows = _p(res2), (res2.stride(0) if res2 is not None else 0), int(res2_rows)
    d.dtype = code(x.dtype)
    keep = (d, x, w1p, b1p, w2q, b2, out, residual, wpo, bpo, res2)
    if front is not None:
        # attn1.to_out in front (rf_ffn_desc.wo ...): x is the attention's output; front = dict(wo [C, C], bo [C], ctx [samples, C] fp32 or None, rows_per_sample,
        # res0 = tok [front_rows or M, C], front_rows (0: M), x1 = the buffer `residual` points at)
        wo, bo, ctx, res0, x1 = front["wo"], front["bo"], front.get("ctx"), front.get("res0"), front["x1"]
        _require_gpu(wo, bo, ctx, res0, x1)
        assert wpo is not None and wo.dtype == x.dtype and wo.shape == (Cc, Cc) and wo.is_contiguous() and bo.dtype == torch.float32 and bo.numel() == Cc
        assert x1.data_ptr() == residual.data_ptr() and x1.shape == (M, Cc) and x1.stride(1) == 1 and x1.dtype == x.dtype
        fr = int(front.get("front_rows", 0))
        assert res0 is None or (res0.dtype == x.dtype and res0.stride(1) == 1 and res0.shape == ((fr or M), Cc))
        assert ctx is None or (ctx.dtype == torch.float32 and ctx.stride(1) == 1 and ctx.shape[1] == Cc and ctx.shape[0] * front["rows_per_sample"] == M)
        d.wo, d.bo, d.ctx, d.ldc, d.rows_per_sample0 = _p(wo), _p(bo), _p(ctx), (ctx.stride(0) if ctx is not None else 0), int(front["rows_per_sample"])
        d.res0, d.ldr0, d.front_rows, d.x1, d.ldx1 = _p(res0), (res0.stride(0) if res0 is not None else 0), fr, _p(x1), x1.stride(0)
        keep = keep + (wo, bo, ctx, res0, x1)
    return Launch(lib.rf_ffn_block, (C.byref(d),), keep, name)


def attn_in(x, wps, rv, tok, wqkv, bqkv, qkv, *, rows_per_sample, ln_eps=1e-5, name="attn_in"):
    """The token-resident front of a SpatialTransformer block at C = 320 (rf_attn_in): tok[M, C] = x W'_s^T + r_s (proj_in with the GroupNorm folded in per sample:
    wps [S, C, C] / rv [S, C] from groupnorm_fold_linear), qkv[M, 3C] = LayerNorm(tok) wqkv^T + bqkv (norm1 without affine in registers: fold gamma / beta with
    fold_layernorm_geglu's scheme).  x / tok / qkv: row-strided 2-D views in one 16-bit type."""
    lib = _lib.load()
    _require_gpu(x, wps, rv, tok, wqkv, bqkv, qkv)
    M, Cc = x.shape
    S = wps.shape[0]
    assert x.dtype in H16 and x.dtype == wps.dtype == tok.dtype == wqkv.dtype == qkv.dtype and rv.dtype == bqkv.dtype == torch.float32
    assert wps.shape == (S, Cc, Cc) and wps.is_contiguous() and rv.shape == (S, Cc) and rv.stride(1) == 1 and M == S * rows_per_sample
    assert wqkv.shape == (3 * Cc, Cc) and wqkv.is_contiguous() and bqkv.shape == (3 * Cc,) and bqkv.is_contiguous()
    assert tok.shape == (M, Cc) and qkv.shape == (M, 3 * Cc) and x.stride(1) == tok.stride(1) == qkv.stride(1) == 1
    d = AttnInDesc()
    d.x, d.ldx, d.wpi, d.w_sample_stride = _p(x), x.stride(0), _p(wps), (Cc * Cc if S > 1 else 0)
    d.rowvec, d.ldv, d.rows_per_sample = _p(rv), rv.stride(0), int(rows_per_sample)
    d.tok, d.ldt, d.wqkv, d.bqkv, d.qkv, d.ldq = _p(tok), tok.stride(0), _p(wqkv), _p(bqkv), _p(qkv), qkv.stride(0)
    d.M, d.C, d.ln_eps, d.dtype = M, Cc, float(ln_eps), code(x.dtype)
    return Launch(lib.rf_attn_in, (C.byref(d),), (d, x, wps, rv, tok, wqkv, bqkv, qkv), name)


class Fp8Weight:
    """A weight matrix in the fp8 storage of BASELINE configs[4]: ``q`` [N, ldq] uint8 holding OCP e4m3fn bytes (rows zero-padded
    to a multiple of 128), ``scale`` [N] fp32 powers of two; w[n, k] = fp8(q[n, k]) * scale[n]."""
    __slots__ = ("q", "scale", "K")

    def __init__(self, q, scale, K):
        self.q, self.scale, self.K = q, scale, K

    @property
    def shape(self):
        return (self.q.shape[0], self.K)

    def dequant(self):
        """fp32 [N, K] values the kernel multiplies with (exact)."""
        return self.q[:, :self.K].view(torch.float8_e4m3fn).float() * self.scale[:, None]


class Fp8Act:
    """An activation tensor in the fp8 storage of the fp8 x fp8 GEMM path: ``q`` [..., Cp] uint8 e4m3fn bytes (Cp = C rounded up to a multiple of
    128; the pad bytes stay 0) and ``scale`` [..., Cp / 32] uint8 E8M0 codes, one per 32-channel block (pad blocks stay 127 = 1.0);
    x[..., c] = fp8(q[..., c]) * 2^(scale[..., c // 32] - 127)."""
    __slots__ = ("q", "scale", "C")

    def __init__(self, shape, device):
        *lead, c = shape
        cp = (c + 127) // 128 * 128
        self.q = torch.zeros((*lead, cp), dtype=torch.uint8, device=device)
        self.scale = torch.full((*lead, cp // 32), 127, dtype=torch.uint8, device=device)
        self.C = c

    @property
    def shape(self):
        return tuple(self.q.shape[:-1]) + (self.C,)

    @property
    def Cp(self):
        return self.q.shape[-1]

    def view(self, *shape):
        """Reshape of the leading dims: ``shape`` is the logical shape (..., C), as for a tensor."""
        assert shape[-1] == self.C, (shape, self.C)
        lead = shape[:-1]
        o = object.__new__(Fp8Act)
        o.q, o.scale, o.C = self.q.view(*lead, self.q.shape[-1]), self.scale.view(*lead, self.scale.shape[-1]), self.C
        return o

    def dequant(self):
        """fp32 [..., C] values the kernel multiplies with (exact)."""
        s = torch.pow(2.0, self.scale.float() - 127.0).repeat_interleave(32, dim=-1)
        return (self.q.view(torch.float8_e4m3fn).float() * s)[..., :self.C]


def quantize_act(x, out, name="quantize_fp8_act"):
    """x bf16 [..., C] -> out (Fp8Act of the same logical shape) through rf_quantize_fp8_act."""
    lib = _lib.load()
    _require_gpu(x, out.q, out.scale)
    Cc = x.shape[-1]
    M = x.numel() // Cc
    assert x.dtype == torch.bfloat16 and x.stride(-1) == 1 and out.C == Cc
    return Launch(lib.rf_quantize_fp8_act, (_p(x), M, Cc, x.stride(-2), _p(out.q), out.q.stride(-2), _p(out.scale), out.scale.stride(-2)),
                  (x, out.q, out.scale), name)


def quantize_fp8_padded(w2d, taps, cin):
    """GEMM weight [N, taps * cin] (tap-major K) -> Fp8Weight [N, taps * Cp] with every tap's channel run zero-padded to Cp = a multiple of
    128: the K order of an Fp8Act source (rf_conv_gemm fp8 x fp8 path)."""
    n, k = w2d.shape
    assert k == taps * cin, (k, taps, cin)
    cp = (cin + 127) // 128 * 128
    if cp != cin:
        w = torch.zeros((n, taps, cp), dtype=torch.float32, device=w2d.device)
        w[..., :cin] = w2d.float().reshape(n, taps, cin)
        w2d = w.reshape(n, taps * cp)
    return quantize_fp8(w2d)


def fp8_eligible(K, cin=None):
    """rf_conv_gemm takes fp8 weights on its direct-to-LDS main loop: K (and, for convolutions, the channel count) multiples
    of the 64-element bf16 K tile."""
    return K % 64 == 0 and (cin is None or cin % 64 == 0)


def quantize_fp8(w2d):
    """[N, K] floating weights (device) -> Fp8Weight through rf_quantize_fp8_rows (per-row power-of-two scale, RNE)."""
    lib = _lib.load()
    _require_gpu(w2d)
    w = w2d.detach().float().contiguous()
    N, K = w.shape
    ldq = (K + 127) // 128 * 128
    q = torch.empty((N, ldq), dtype=torch.uint8, device=w.device)
    sc = torch.empty((N,), dtype=torch.float32, device=w.device)
    _lib.check(lib.rf_quantize_fp8_rows(_p(w), N, K, ldq, _p(q), _p(sc), stream_ptr()), "rf_quantize_fp8_rows")
    return Fp8Weight(q, sc, K)


# ------------------------------------------------------------------------------------------------
# descriptor-based launches.  `Launch` objects are built once per (layer, shape) and replayed.
# ------------------------------------------------------------------------------------------------
class Launch:
    """A prepared kernel launch: ``fn(*args, stream)``.  Keeps the tensors it points at alive."""
    __slots__ = ("fn", "args", "keep", "name")

    def __init__(self, fn, args, keep, name):
        self.fn, self.args, self.keep, self.name = fn, args, keep, name

    def __call__(self, stream=None):
        rc = self.fn(*self.args, stream if stream is not None else stream_ptr())
        if rc != 0:
            _lib.check(rc, self.name)


_WORKSPACES = {}
SPLITK_WORKSPACE_BYTES = 96 << 20
_TLS = threading.local()          # per-thread stack of engine workspaces (workspace_scope): the CLI builds / runs engines from helper threads


def _scope_stack():
    st = getattr(_TLS, "stack", None)
    if st is None:
        st = _TLS.stack = []
    return st


class workspace_scope:
    """Launches prepared inside the scope use `ws` as their split-K scratch.  Each engine (UNet, VAE encoder / decoder, CLIP,
    ArcFace) owns one: launch lists of different engines may then run concurrently on different streams without sharing
    partial-sum memory; inside one engine the launches are serialised on a stream.  The scope is thread-local: an engine built lazily on
    a helper thread (landmark prefetch, PNG writer) never sees another thread's open scope."""

    def __init__(self, ws):
        self.ws = ws

    def __enter__(self):
        _scope_stack().append(self.ws)
        return self.ws

    def __exit__(self, *a):
        _scope_stack().pop()


def new_workspace(device, nbytes=SPLITK_WORKSPACE_BYTES):
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device)


def _default_workspace(device):
    """Split-K scratch of launches prepared outside any engine: one fp32 buffer per device -- single-stream use only."""
    st = _scope_stack()
    if st:
        return st[-1]
    ws = _WORKSPACES.get(device)
    if ws is None:
        ws = new_workspace(device)
        _WORKSPACES[device] = ws
    return ws


def conv_gemm(src0, W, out, *, M, N, K, C0, ld0, src1=None, C1=0, ld1=0, Hin=1, Win=1, Hout=1, Wout=1, KH=1, KW=1,
              stride=1, pad_t=0, pad_l=0, ups=0, bias=None, rowvec=None, rows_per_sample=0, ldv=0, residual=None, ldr=0,
              act=ACT_NONE, ldo=None, alpha=1.0, batch=1, sA=0, sW=0, sO=0, sR=0, ldw=0, act_vec=None, korder=0, workspace=None, x3=False,
              name="rf_conv_gemm"):
    """Prepare an rf_conv_gemm launch (see include/reface_hip.h).  x3: split-bf16 operands (src0 = [.., C0 hi | C0 lo] bf16, W from
    pack_x3, fp32 out); K / C0 are the REAL sizes."""
    lib = _lib.load()
    a8 = None
    if isinstance(src0, Fp8Act):          # fp8 activations + block scales: K / C0 / ld0 are the PADDED byte counts (the caller passes them)
        a8, src0 = src0, src0.q
        assert isinstance(W, Fp8Weight) and (isinstance(out, Fp8Act) or out.dtype == torch.bfloat16) and src1 is None and not x3
    oq = None
    if isinstance(out, Fp8Act):           # GEGLU of the fp8 x fp8 path writing fp8 + block scales for ff.net.2
        oq, out = out, out.q
        assert a8 is not None and act == ACT_GEGLU and residual is None and oq.C == N // 2
    _require_gpu(src0, W.q if isinstance(W, Fp8Weight) else W, out, src1, bias, rowvec, residual)
    d = ConvGemmDesc()
    d.dtype, d.out_dtype = (RF_FP8_E4M3 if a8 is not None else code(src0.dtype)), (RF_FP8_E4M3 if oq is not None else code(out.dtype))
    if oq is not None:
        d.oscale, d.os_ld = _p(oq.scale), oq.scale.stride(-2)
    if a8 is not None:
        d.ascale, d.as_ld = _p(a8.scale), a8.scale.stride(-2)
    wq = None
    if isinstance(W, Fp8Weight):             # fp8 weights: bf16 activations, bytes + per-row scales
        wq, W = W, W.q
        assert (src0.dtype == torch.bfloat16 or a8 is not None) and wq.K == K and ldw == 0, (src0.dtype, wq.K, K)
        d.w_dtype, d.wscale, ldw = RF_FP8_E4M3, _p(wq.scale), W.stride(0)
    else:
        assert W.dtype == src0.dtype
    if x3:
        assert src0.dtype == torch.bfloat16 and out.dtype == torch.float32 and wq is None and W.shape[-1] == 3 * K, (src0.dtype, out.dtype, tuple(W.shape), K)
        d.dtype = RF_BF16X3
    assert (src1 is None or src1.dtype == src0.dtype)
    assert residual is None or residual.dtype == out.dtype
    assert bias is None or bias.dtype == torch.float32
    assert rowvec is None or rowvec.dtype == torch.float32
    d.M, d.N, d.K = M, N, K
    d.src0, d.src1 = _p(src0), _p(src1)
    d.C0, d.C1, d.ld0, d.ld1 = C0, C1, ld0, ld1
    d.Hin, d.Win, d.Hout, d.Wout = Hin, Win, Hout, Wout
    d.KH, d.KW, d.stride, d.pad_t, d.pad_l, d.ups = KH, KW, stride, pad_t, pad_l, ups
    d.W, d.ldw, d.bias, d.rowvec = _p(W), ldw, _p(bias), _p(rowvec)
    d.rows_per_sample, d.ldv = rows_per_sample, ldv
    d.residual, d.ldr, d.act = _p(residual), ldr, act
    d.out, d.ldo, d.alpha = _p(out), (ldo if ldo is not None else (N // 2 if act == ACT_GEGLU else N)), alpha
    d.batch, d.sA, d.sW, d.sO, d.sR = batch, sA, sW, sO, sR
    d.act_vec = _p(act_vec)
    d.korder = korder
    ws = workspace if workspace is not None else _default_workspace(src0.device)
    d.workspace, d.workspace_bytes = _p(ws), (ws.numel() * ws.element_size() if ws is not None else 0)
    l = Launch(lib.rf_conv_gemm, (C.byref(d),), (d, src0, src1, W, out, bias, rowvec, residual, act_vec, ws, wq, a8, oq), name)
    if wq is not None and a8 is None:
        # fp8 weights run only on the direct-to-LDS main loop, whose preconditions (one source, 31-bit operand extents, <= 1024^2 outputs,
        # < 4095 samples, whole K tiles per tap) depend on the launch, not only on the weight: ask the library, and give a layer that
        # cannot take them its exactly dequantised bf16 weights instead of failing at the first replay
        bm, bn, sk = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        if lib.rf_conv_gemm_plan(C.byref(d), C.byref(bm), C.byref(bn), C.byref(sk)) != 0:
            wd = wq.dequant().to(torch.bfloat16).contiguous()
            d.w_dtype, d.wscale, d.W, d.ldw = 0, None, _p(wd), 0
            l = Launch(lib.rf_conv_gemm, (C.byref(d),), (d, src0, src1, wd, out, bias, rowvec, residual, act_vec, ws), name)
            _lib.check(lib.rf_conv_gemm_plan(C.byref(d), C.byref(bm), C.byref(bn), C.byref(sk)), name + ".plan")
    return l


def linear(x, W, out, bias=None, *, act=ACT_NONE, residual=None, rowvec=None, rows_per_sample=0, alpha=1.0, act_vec=None, x3=False, ln_u=None,
           w_per_sample=None, name="linear"):
    """out[M, N] = act(x[M, K] @ W[N, K]^T + bias) (+ residual).  x / out may be row-strided 2-D views.  x3: x is the split-bf16 form
    [M, 2K] of an fp32 matrix and W comes from pack_x3 ([N, 3K])."""
    if x3:
        M, K2 = x.shape
        K, N = K2 // 2, W.shape[0]
        assert W.shape[1] == 3 * K and x.stride(1) == 1 and out.stride(1) == 1
        return conv_gemm(x, W, out, M=M, N=N, K=K, C0=K, ld0=x.stride(0), Hin=1, Win=M, Hout=1, Wout=M, bias=bias, act=act, residual=residual,
                         ldr=(residual.stride(0) if residual is not None else 0), rowvec=rowvec, rows_per_sample=rows_per_sample,
                         ldv=(rowvec.stride(0) if rowvec is not None else 0), ldo=out.stride(0), alpha=alpha, act_vec=act_vec, x3=True, name=name)
    if isinstance(x, Fp8Act):             # K = the padded channel count (W from quantize_fp8_padded(w, 1, C))
        M, K, ld0 = x.q.shape[0], x.Cp, x.q.stride(0)
    else:
        (M, K), ld0 = x.shape, x.stride(0)
        assert x.stride(1) == 1
    N = W.shape[0]
    ldo = out.q.stride(0) if isinstance(out, Fp8Act) else out.stride(0)
    assert W.shape[1] == K and (isinstance(out, Fp8Act) or out.stride(1) == 1), (tuple(W.shape), K)
    l = conv_gemm(x, W, out, M=M, N=N, K=K, C0=K, ld0=ld0, Hin=1, Win=M, Hout=1, Wout=M, bias=bias, act=act,
                  residual=residual, ldr=(residual.stride(0) if residual is not None else 0), rowvec=rowvec,
                  rows_per_sample=rows_per_sample, ldv=(rowvec.stride(0) if rowvec is not None else 0),
                  ldo=ldo, alpha=alpha, act_vec=act_vec, name=name)
    if ln_u is not None:          # consumer of a folded LayerNorm (fold_layernorm_linear + layernorm_fold): sum_k W'[n, k]
        assert ln_u.dtype == torch.float32 and ln_u.is_contiguous() and ln_u.numel() == N
        l.keep[0].ln_u = _p(ln_u)
        l.keep = tuple(l.keep) + (ln_u,)
    if w_per_sample is not None:          # [S, N, K] weights, W = w_per_sample[0]: the rows of sample s multiply w_per_sample[s] (groupnorm_fold_linear)
        assert w_per_sample.is_contiguous() and w_per_sample.shape[1:] == W.shape and W.data_ptr() == w_per_sample.data_ptr() and rows_per_sample > 0
        assert M == w_per_sample.shape[0] * rows_per_sample
        l.keep[0].w_sample_stride = N * K
        l.keep = tuple(l.keep) + (w_per_sample,)
    return l


def groupnorm_fold_linear(W, gamma, beta, bias, partial, nchunks, *, B, HW, eps, dtype, name="groupnorm.fold"):
    """GroupNorm(32) folded into the Linear / 1x1 conv behind it (rf_groupnorm_fold_linear): W fp32 [N, C] -> (launch, W' [B, N, C] in `dtype`,
    per-sample vector fp32 [B, N]).  The GEMM then reads the UN-normalised tensor: linear(x, W'[0], out, None, rowvec=vec, rows_per_sample=HW,
    w_per_sample=W') -- the rf_groupnorm_apply pass (a read + a write of [B, HW, C]) and the normalised copy are gone."""
    lib = _lib.load()
    N, Cc = W.shape
    _require_gpu(W, gamma, beta, partial)
    assert W.dtype == torch.float32 and W.is_contiguous() and Cc % 32 == 0 and partial.dtype == torch.float64
    wout = torch.empty((B, N, Cc), dtype=dtype, device=W.device)
    rv = torch.empty((B, N), dtype=torch.float32, device=W.device)
    l = Launch(lib.rf_groupnorm_fold_linear, (_p(W), N, Cc, B, HW, nchunks, _p(partial), _p(gamma), _p(beta), _p(bias), float(eps), code(dtype), _p(wout), _p(rv)),
               (W, gamma, beta, bias, partial, wout, rv), name)
    return l, wout, rv


SMALLCONV_CHANNELS = (320, 128, 64)


def gn_silu_conv3x3_small(x, gamma, beta, partial, nchunks, W, bias, out, *, eps, silu=True, workspace=None, name="gn_silu_conv3x3"):
    """GroupNorm(32) + SiLU + 3x3 conv (stride 1, pad 1) to <= 4 channels on the RAW bf16 tensor x [B, H, W, C] (rf_gn_silu_conv3x3_small; the
    UNet's `out` head): W packed [No, 9 C] bf16 (pack_conv_weight), out [B, H, W, No] fp32 / bf16, partial / nchunks from fuse_groupnorm_stats or
    groupnorm_stats.  workspace: fp32 tensor of >= B*H*W*40 elements (per-tap partial products)."""
    lib = _lib.load()
    B, H, W_, Cc = x.shape
    No = W.shape[0]
    _require_gpu(x, gamma, beta, partial, W, bias, out)
    assert x.dtype in H16 and W.dtype == x.dtype and out.dtype in (torch.float32, x.dtype) and W.shape[1] == 9 * Cc and x.stride(3) == 1 and out.stride(3) == 1 and out.shape[:3] == x.shape[:3]
    assert Cc in SMALLCONV_CHANNELS and 1 <= No <= 4 and partial.dtype == torch.float64
    assert x.stride(0) == H * W_ * x.stride(2) and x.stride(1) == W_ * x.stride(2) and out.stride(0) == H * W_ * out.stride(2) and out.stride(1) == W_ * out.stride(2)
    if workspace is None:
        workspace = torch.empty((B * H * W_ * 40,), dtype=torch.float32, device=x.device)
    assert workspace.dtype == torch.float32 and workspace.numel() >= B * H * W_ * 40
    return Launch(lib.rf_gn_silu_conv3x3_small, (code(x.dtype), _p(x), B, H, W_, Cc, x.stride(2), nchunks, _p(partial), _p(gamma), _p(beta), float(eps), int(bool(silu)), _p(W), _p(bias),
                                                 No, code(out.dtype), _p(out), out.stride(2), _p(workspace), workspace.numel() * 4),
                  (x, gamma, beta, partial, W, bias, out, workspace), name)


def conv3x3_stem(x, W, bias, out, *, dup=None, name="conv_in"):
    """The UNet's stem (rf_conv3x3_stem): 3x3 conv, stride 1, pad 1, from the 16 stored input channels of x [B, H, W, >= 16] bf16 to
    out [B, H, W, C] bf16 (C in SMALLCONV_CHANNELS, H*W a multiple of 128), W packed [C, 144] (pack_conv_weight with cin_pad = 16).
    dup: a second [B, H, W, C] view with out's strides that receives the same rows (the CFG-duplicated batch half).  The launch can emit the GroupNorm
    statistics of `out` and of `dup` for up to three consumers (fuse_groupnorm_stats accepts it as a producer: 128-row blocks, all C columns)."""
    lib = _lib.load()
    _require_gpu(x, W, bias, out, dup)
    B, H, W_, Ci = x.shape
    Cc = W.shape[0]
    assert x.dtype == W.dtype == out.dtype and x.dtype in H16 and (bias is None or bias.dtype == torch.float32)
    assert Ci >= 16 and W.shape == (Cc, 144) and W.is_contiguous() and Cc in SMALLCONV_CHANNELS and (H * W_) % 128 == 0 and out.shape == (B, H, W_, Cc)
    for t in (x, out) + ((dup,) if dup is not None else ()):
        assert t.stride(3) == 1 and t.stride(1) == W_ * t.stride(2) and t.stride(0) == H * W_ * t.stride(2)
    d = StemDesc()
    d.x, d.ldx, d.B, d.H, d.W, d.C = _p(x), x.stride(2), B, H, W_, Cc
    d.w, d.bias, d.out, d.ldo = _p(W), _p(bias), _p(out), out.stride(2)
    d.dup_off = 0
    if dup is not None:
        assert dup.dtype == out.dtype and dup.shape == out.shape and dup.stride() == out.stride()
        off = dup.data_ptr() - out.data_ptr()
        assert off > 0 and off % 16 == 0
        d.dup_off = off // 2
    d.dtype = code(x.dtype)
    return Launch(lib.rf_conv3x3_stem, (C.byref(d),), (d, x, W, bias, out, dup), name)


def conv2d(x, W, out, bias=None, *, ksize=3, stride=1, pad=(1, 1), ups=0, x2=None, residual=None, rowvec=None,
           act=ACT_NONE, act_vec=None, korder=0, x3=False, name="conv2d"):
    """Channels-last convolution.  x: [B, Hin, Win, C0] (+ optional x2 [B, Hin, Win, C1] concatenated
    on channels); W: packed [Cout, k*k*(C0+C1)]; out: [B, Hout, Wout, Cout].  x3: x is the split-bf16 form [B, Hin, Win, 2*C0] of an
    fp32 tensor and W comes from pack_x3 ([Cout, 3*k*k*C0])."""
    if isinstance(x, Fp8Act):             # C0 = the padded channel count (W from quantize_fp8_padded(w, k*k, C))
        B, Hin, Win, C0 = x.q.shape
        ld0 = x.q.stride(2)
    else:
        B, Hin, Win, C0 = x.shape
        ld0 = x.stride(2)
        assert x.stride(3) == 1
    C1 = 0 if x2 is None else x2.shape[3]
    Bo, Hout, Wout, N = out.shape
    assert Bo == B and out.stride(3) == 1
    K = W.shape[1]
    if x3:
        assert x2 is None and C0 % 2 == 0 and K % 3 == 0
        C0, K = C0 // 2, K // 3
    return conv_gemm(x, W, out, M=B * Hout * Wout, N=N, K=K, C0=C0, ld0=ld0, src1=x2, C1=C1,
                     ld1=(x2.stride(2) if x2 is not None else 0), Hin=Hin, Win=Win, Hout=Hout, Wout=Wout, KH=ksize, KW=ksize,
                     stride=stride, pad_t=pad[0], pad_l=pad[1], ups=ups, bias=bias, residual=residual,
                     ldr=(residual.stride(2) if residual is not None else 0), rowvec=rowvec, rows_per_sample=Hout * Wout,
                     ldv=(rowvec.stride(0) if rowvec is not None else 0), act=act, act_vec=act_vec, ldo=out.stride(2), korder=korder, x3=x3, name=name)


GN_MAX_CHUNKS = 32
GN_FUSED_MAX_CHUNKS = 96     # chunk slots per sample rf_groupnorm_apply re-reduces itself; above that a finalize pass compacts them
GN_FUSED_MAX_SLOTS = 1 << 16


def gn_chunks(B, HW):
    """Pixel chunks of the GroupNorm statistics pass: >= 32 pixels per block, about two blocks per CU in total."""
    n = max(1, min(GN_MAX_CHUNKS, HW // 32))
    while B * n > 512 and n > 1:
        n //= 2
    return n


def _gn_out_code(x, out, split):
    if split:           # out = split-bf16 pairs [B, H, W, 2C] of the normalised fp32 values (RF_BF16X3 operand of the next conv)
        assert x.dtype == torch.float32 and out.dtype == torch.bfloat16 and out.shape[-1] == 2 * x.shape[-1]
        return RF_BF16X3
    return code(out.dtype)


def groupnorm_stats(x, partial, name="groupnorm"):
    """The statistics pass alone -> (launch, chunks per sample)."""
    lib = _lib.load()
    B, H, W_, Cc = x.shape
    n = gn_chunks(B, H * W_)
    _require_gpu(x, partial)
    assert partial.dtype == torch.float64 and partial.numel() >= B * n * 64
    return Launch(lib.rf_groupnorm_stats, (code(x.dtype), _p(x), B, H * W_, Cc, x.stride(2), n, _p(partial)), (x, partial), name + ".stats"), n


def groupnorm(x, gamma, beta, out, partial, *, eps, silu, split=False, name="groupnorm"):
    """GroupNorm(32)(+SiLU) over channels-last x [B, H, W, C] -> out.  ``partial``: fp64 scratch
    of at least B * GN_MAX_CHUNKS * 64 elements.  Returns the two launches (stats, apply)."""
    lib = _lib.load()
    B, H, W_, Cc = x.shape
    HW = H * W_
    n = gn_chunks(B, HW)
    assert partial.dtype == torch.float64 and partial.numel() >= B * n * 64
    a = Launch(lib.rf_groupnorm_stats, (code(x.dtype), _p(x), B, HW, Cc, x.stride(2), n, _p(partial)), (x, partial), name + ".stats")
    if isinstance(out, Fp8Act):
        return [a, groupnorm_apply(x, gamma, beta, out, partial, n, eps=eps, silu=silu, name=name)]
    _require_gpu(x, gamma, beta, out, partial)
    b = Launch(lib.rf_groupnorm_apply, (code(x.dtype), _p(x), B, HW, Cc, x.stride(2), n, _p(partial), _p(gamma), _p(beta),
                                        float(eps), int(bool(silu)), _gn_out_code(x, out, split), _p(out), out.stride(2)),
               (x, partial, gamma, beta, out), name + ".apply")
    return [a, b]


def gemm_plan(launch):
    """(rows, columns of the block that finishes an output tile, split-K factor) rf_conv_gemm will use for this prepared launch."""
    lib = _lib.load()
    bm, bn, sk = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    _lib.check(lib.rf_conv_gemm_plan(C.byref(launch.keep[0]), C.byref(bm), C.byref(bn), C.byref(sk)), launch.name + ".plan")
    return bm.value, bn.value, sk.value


def gemm_plan2(launch):
    """The whole tile plan of a prepared rf_conv_gemm launch: dict(stat_rows, stat_cols, splitk, bm, bn, wave_cols, direct, frag, gemm_kernels); raises
    RefaceHipError when the library rejects the descriptor (e.g. LayerNorm folding asked of a launch that cannot carry it)."""
    lib = _lib.load()
    info = (C.c_int32 * 8)()
    _lib.check(lib.rf_conv_gemm_plan2(C.byref(launch.keep[0]), info), launch.name + ".plan2")
    pl = dict(zip(("stat_rows", "stat_cols", "splitk", "bm", "bn", "wave_cols", "direct", "frag"), (int(v) for v in info)))
    # eighth word: bit 0 = split-K through fragment-ordered slabs, bit 1 = the call runs as TWO GEMM kernels (tail-round split along N)
    pl["gemm_kernels"] = 2 if pl["frag"] & 2 else 1
    pl["frag"] &= 1
    return pl


def fold_layernorm_linear(w, gamma, beta, bias, dtype):
    """LayerNorm folded into the Linear behind it, the consumer half of rf_conv_gemm's ln_* fields:
        (xhat * gamma + beta) W^T + b  =  rstd (x W'^T - mean u) + b'      with  W' = W diag(gamma),  u[n] = sum_k W'[n, k],  b' = b + W beta.
    w [N, C] fp32 (any row packing -- the fold is per column), gamma / beta [C], bias [N] or None -> (W' in `dtype`, u fp32 [N] summed over the
    ROUNDED W' so that the mean's term cancels exactly what the matrix pipe accumulates, b' fp32 [N])."""
    wf = w.float()
    w2 = (wf * gamma.float()[None, :]).to(dtype).contiguous()
    u = w2.float().sum(dim=1).contiguous()
    b2 = wf @ beta.float()
    if bias is not None:
        b2 = b2 + bias.float()
    return w2, u, b2.contiguous()


def layernorm_fold(producers, consumer, *, eps, C_):
    """Wire LayerNorm statistics from the producer GEMM(s) of a [M, C] tensor into the consumer GEMM that multiplies its normalised form.
    producers: [(launch, row0, rows)] prepared rf_conv_gemm launches that write rows [row0, row0 + rows) (all C columns each); consumer: the
    prepared launch whose A operand is that tensor (weights / bias / ln_u already folded: set_layernorm_consumer).  Returns the statistics
    tensor [M, parts, 2], or None (producers untouched; the consumer, whose weights assume the fold, is then to be discarded) when some launch
    cannot carry its half -- the caller keeps the rf_layernorm pass."""
    cd = consumer.keep[0]
    M = cd.M
    plans = []
    for l, row0, rows in producers:
        d = l.keep[0]
        if l.fn.__name__ != "rf_conv_gemm" or d.N != C_ or d.act != ACT_NONE or d.M != rows:
            return None
        pl = gemm_plan2(l)
        if not pl["direct"] or pl["splitk"] != 1 or C_ % pl["wave_cols"]:
            return None
        plans.append(pl["wave_cols"])
    if len(set(plans)) != 1 or sum(r for _, _, r in producers) != M:
        return None
    wc = plans[0]
    parts = C_ // wc
    stats = torch.zeros((M, parts, 2), dtype=torch.float32, device=consumer.keep[1].device)
    try:
        for l, row0, rows in producers:
            d = l.keep[0]
            d.ln_stats_out, d.ln_out_parts = stats[row0:].data_ptr(), parts
            gemm_plan2(l)                       # the library checks the producer half
        cd.ln_stats_in, cd.ln_in_parts, cd.ln_in_cols, cd.ln_eps = stats.data_ptr(), parts, wc, float(eps)
        gemm_plan2(consumer)                    # ... and the consumer half
    except _lib.RefaceHipError:
        for l, _, _ in producers:               # producers back to plain launches; the (folded-weight) consumer is the caller's to discard
            l.keep[0].ln_stats_out, l.keep[0].ln_out_parts = None, 0
        cd.ln_stats_in, cd.ln_in_parts, cd.ln_in_cols, cd.ln_eps = None, 0, 0, 0.0
        return None
    for l, _, _ in producers:
        l.keep = tuple(l.keep) + (stats,)
    consumer.keep = tuple(consumer.keep) + (stats,)
    return stats


def fuse_groupnorm_stats(x, producers):
    """Let the GEMMs that wrote ``x`` [B, H, W, C] also emit its GroupNorm(32) partial sums.

    ``producers``: [(launch, row0, rows, col0, cols)] -- prepared rf_conv_gemm launches that together tile the [B*H*W, C] matrix
    (column slices of a concat buffer, batch halves).  Returns (partial, nchunks, extra launches to run before the apply) for
    groupnorm_apply, or None when some
    producer cannot do it (tile rows straddling samples, GEGLU, both consumer slots taken, uneven tiling).
    """
    B, H, W_, Cc = x.shape
    HW, M = H * W_, B * H * W_
    if Cc % 32 or sum(r * c for _, _, r, _, c in producers) != M * Cc:
        return None
    plans, slot_of, need, nslots = [], {}, {}, 0
    for l, row0, rows, col0, cols in producers:
        d = l.keep[0]
        if l.fn.__name__ == "rf_ffn_block":          # the fused transformer tail: 128-token blocks over all C columns
            if (d.gn_part0 and d.gn_part1) or not d.wpo or row0 % HW or rows % HW or d.M != rows or d.C != cols:
                return None
            bm, bn = 128, cols
        elif l.fn.__name__ == "rf_conv3x3_stem":     # the stem: 128-pixel blocks over all C columns; three consumer slots (its output and the
            free = [i for i in range(3) if not getattr(d, f"gn_part{i}")]          # CFG duplicate are separate producer entries of one launch)
            if len(free) < sum(1 for q in producers if q[0] is l) or d.H * d.W != HW or rows != d.B * HW or row0 % HW or d.C != cols:
                return None
            bm, bn = 128, cols
        else:
            if l.fn.__name__ != "rf_conv_gemm" or d.act == ACT_GEGLU or d.batch != 1 or (d.gn_part0 and d.gn_part1):
                return None
            if row0 % HW or rows % HW or d.M != rows or d.N != cols:
                return None
            bm, bn, sk = gemm_plan(l)
        if HW % bm:
            return None
        per_sample = (HW // bm) * ((cols + bn - 1) // bn)
        key = (col0, cols)
        # batch slices of one column range share the slot numbering; slices with different tile plans (a launch split by samples) get the larger
        # count -- the slots a sample does not write stay zero
        need[key] = max(need.get(key, 0), per_sample)
        plans.append((l, row0, key))
    for key, n in need.items():
        slot_of[key] = (nslots, n)
        nslots += n
    if nslots > GN_FUSED_MAX_SLOTS:
        return None
    partial = torch.zeros((B, nslots, 32, 2), dtype=torch.float64, device=x.device)
    for l, row0, key in plans:
        d = l.keep[0]
        part = partial[row0 // HW:]
        if l.fn.__name__ == "rf_conv3x3_stem":
            i = next(i for i in range(3) if not getattr(d, f"gn_part{i}"))
            for f, v in (("part", part.data_ptr()), ("cpg", Cc // 32), ("coff", key[0]), ("slot", slot_of[key][0]), ("nchunks", nslots)):
                setattr(d, f"gn_{f}{i}", v)
            l.keep = tuple(l.keep) + (partial,)
            continue
        d.gn_rows = HW
        if not d.gn_part0:
            d.gn_part0, d.gn_cpg0, d.gn_coff0, d.gn_slot0, d.gn_nchunks0 = part.data_ptr(), Cc // 32, key[0], slot_of[key][0], nslots
        else:
            d.gn_part1, d.gn_cpg1, d.gn_coff1, d.gn_slot1, d.gn_nchunks1 = part.data_ptr(), Cc // 32, key[0], slot_of[key][0], nslots
        l.keep = tuple(l.keep) + (partial,)
    if nslots > GN_FUSED_MAX_CHUNKS:        # many tiles per sample (large images): compact to one slot per sample first
        lib = _lib.load()
        compact = torch.zeros((B, 1, 32, 2), dtype=torch.float64, device=x.device)
        fin = Launch(lib.rf_groupnorm_finalize, (_p(partial), B, nslots, _p(compact)), (partial, compact), "groupnorm.finalize")
        return compact, 1, [fin]
    return partial, nslots, []


def groupnorm_apply(x, gamma, beta, out, partial, nchunks, *, eps, silu, split=False, name="groupnorm"):
    """The normalisation pass alone, on statistics that already sit in ``partial`` (fuse_groupnorm_stats).  ``out`` may be an Fp8Act
    (bf16 input): the normalised values leave as e4m3fn bytes + block scales."""
    lib = _lib.load()
    B, H, W_, Cc = x.shape
    if isinstance(out, Fp8Act):
        _require_gpu(x, gamma, beta, out.q, out.scale, partial)
        assert x.dtype == torch.bfloat16 and out.C == Cc
        return Launch(lib.rf_groupnorm_apply_fp8, (_p(x), B, H * W_, Cc, x.stride(2), nchunks, _p(partial), _p(gamma), _p(beta), float(eps),
                                                   int(bool(silu)), _p(out.q), out.q.stride(2), _p(out.scale), out.scale.stride(2)),
                      (x, partial, gamma, beta, out.q, out.scale), name + ".apply")
    _require_gpu(x, gamma, beta, out, partial)
    return Launch(lib.rf_groupnorm_apply, (code(x.dtype), _p(x), B, H * W_, Cc, x.stride(2), nchunks, _p(partial), _p(gamma), _p(beta),
                                           float(eps), int(bool(silu)), _gn_out_code(x, out, split), _p(out), out.stride(2)),
                  (x, partial, gamma, beta, out), name + ".apply")


def layernorm(x, gamma, beta, out, *, eps=1e-5, name="layernorm"):
    lib = _lib.load()
    M, Cc = x.shape
    if isinstance(out, Fp8Act):
        _require_gpu(x, gamma, beta, out.q, out.scale)
        assert x.dtype == torch.bfloat16 and out.C == Cc
        return Launch(lib.rf_layernorm_fp8, (_p(x), M, Cc, x.stride(0), _p(gamma), _p(beta), float(eps), _p(out.q), out.q.stride(0), _p(out.scale),
                                             out.scale.stride(0)), (x, gamma, beta, out.q, out.scale), name)
    _require_gpu(x, gamma, beta, out)
    split = x.dtype == torch.float32 and out.dtype == torch.bfloat16 and out.shape[-1] == 2 * Cc          # split-bf16 pairs (RF_BF16X3 operand)
    return Launch(lib.rf_layernorm, (code(x.dtype), _p(x), M, Cc, x.stride(0), _p(gamma), _p(beta), float(eps),
                                     RF_BF16X3 if split else code(out.dtype), _p(out), out.stride(0)), (x, gamma, beta, out), name)


def attention(q, k, v, out, *, heads, scale, x3=False, name="attention"):
    """q/k/v/out: [B, N, heads*d] views (last dim contiguous; may be slices of a fused qkv buffer).  x3: fp32 tensors multiplied as
    split-bf16 operand pairs (hi hi + hi lo + lo hi on the bf16 MFMA, both contractions; fp32 softmax, accumulation and output) -- the
    attention of the "f32x3" parity mode, 2^-16 relative error per product instead of the exact-fp32 MFMA's 2^-24 at a fifth of its time."""
    lib = _lib.load()
    _require_gpu(q, k, v, out)
    B, Nq, Cc = q.shape
    Nk = k.shape[1]
    d = Cc // heads
    assert q.dtype == k.dtype == v.dtype == out.dtype and (not x3 or q.dtype == torch.float32)
    return Launch(lib.rf_attention, (RF_BF16X3 if x3 else code(q.dtype), _p(q), _p(k), _p(v), _p(out), B, heads, d, Nq, Nk, q.stride(1), k.stride(1),
                                     v.stride(1), out.stride(1), q.stride(0), k.stride(0), v.stride(0), out.stride(0), float(scale)),
                  (q, k, v, out), name)


def softmax_rows(x, name="softmax_rows"):
    lib = _lib.load()
    _require_gpu(x)
    rows, cols = x.shape
    return Launch(lib.rf_softmax_rows, (_p(x), rows, cols, x.stride(0)), (x,), name)


def ddim_pack_input(img, z_inpaint, mask, x_in, *, dup, name="ddim_pack"):
    lib = _lib.load()
    _require_gpu(img, z_inpaint, mask, x_in)
    B, _, h, w = img.shape
    return Launch(lib.rf_ddim_pack_input, (_p(img), _p(z_inpaint), _p(mask), B, h * w, dup, code(x_in.dtype), _p(x_in), x_in.shape[-1]),
                  (img, z_inpaint, mask, x_in), name)


def ddim_update(eps, img, pred_x0, noise, coefs, *, cfg, scale, name="ddim_update"):
    """coefs: device fp32 [5] = {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev-sigma^2), sigma}."""
    lib = _lib.load()
    _require_gpu(eps, img, pred_x0, noise, coefs)
    B, _, h, w = img.shape
    assert eps.dtype == torch.float32 and coefs.dtype == torch.float32
    return Launch(lib.rf_ddim_update, (_p(eps), eps.shape[-1], int(cfg), float(scale), _p(img), _p(pred_x0), _p(noise), B, h * w,
                                       _p(coefs)), (eps, img, pred_x0, noise, coefs), name)


def nchw_to_nhwc(x, out, name="nchw_to_nhwc"):
    lib = _lib.load()
    _require_gpu(x, out)
    B, Cc, H, W_ = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous()
    return Launch(lib.rf_nchw_to_nhwc, (_p(x), B, Cc, H * W_, code(out.dtype), _p(out), out.shape[-1]), (x, out), name)


def nhwc_to_nchw(x, out, C_=None, name="nhwc_to_nchw"):
    lib = _lib.load()
    _require_gpu(x, out)
    B, H, W_, ld = x.shape[0], x.shape[1], x.shape[2], x.stride(2)
    Cc = out.shape[1] if C_ is None else C_
    assert out.dtype == torch.float32 and out.is_contiguous()
    return Launch(lib.rf_nhwc_to_nchw, (code(x.dtype), _p(x), B, Cc, H * W_, ld, _p(out)), (x, out), name)


def cast(x, out, name="cast"):
    lib = _lib.load()
    _require_gpu(x, out)
    assert x.is_contiguous() and out.is_contiguous() and x.numel() == out.numel()
    return Launch(lib.rf_cast, (code(x.dtype), _p(x), code(out.dtype), _p(out), x.numel()), (x, out), name)


def timestep_embedding(t, freqs, out, name="timestep_embedding"):
    lib = _lib.load()
    _require_gpu(t, freqs, out)
    n, dim = out.shape
    assert t.dtype == torch.float32 and freqs.dtype == torch.float32 and out.dtype == torch.float32
    return Launch(lib.rf_timestep_embedding, (_p(t), n, dim, _p(freqs), _p(out)), (t, freqs, out), name)


def silu_f32(x, out, name="silu"):
    lib = _lib.load()
    _require_gpu(x, out)
    return Launch(lib.rf_silu_f32, (_p(x), _p(out), x.numel()), (x, out), name)


def to_image(x, out, name="to_image"):
    lib = _lib.load()
    _require_gpu(x, out)
    assert x.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous()
    return Launch(lib.rf_to_image, (_p(x), _p(out), x.numel()), (x, out), name)


def u8_to_norm(x_u8, mean, std, out, name="u8_to_norm"):
    """x_u8 [B, H, W, 3] uint8 -> out [B, 3, H, W] fp32 = (x / 255 - mean) / std (ToTensor + Normalize)."""
    lib = _lib.load()
    _require_gpu(x_u8, mean, std, out)
    B, H, W_, C3 = x_u8.shape
    assert C3 == 3 and x_u8.dtype == torch.uint8 and x_u8.is_contiguous() and out.is_contiguous() and out.dtype == torch.float32
    return Launch(lib.rf_u8_to_norm, (_p(x_u8), B, H * W_, _p(mean), _p(std), _p(out)), (x_u8, mean, std, out), name)


def compose_outputs_u8(result01, target, inpaint, mask, ref, out_u8, *, with_grid=True, name="compose_outputs_u8"):
    """The CLI's output panels (+ grid) of a batch as packed uint8 records out_u8 [B, record_bytes] (reface_amd/output.record_layout)."""
    lib = _lib.load()
    _require_gpu(result01, target, inpaint, mask, ref, out_u8)
    B, _, H, W_ = result01.shape
    for t in (result01, target, inpaint, mask, ref):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[0] == B and tuple(t.shape[2:]) == (H, W_), tuple(t.shape)
    assert out_u8.dtype == torch.uint8 and out_u8.is_contiguous() and out_u8.shape[0] == B
    return Launch(lib.rf_compose_outputs_u8, (_p(result01), _p(target), _p(inpaint), _p(mask), _p(ref), B, H, W_, int(bool(with_grid)), _p(out_u8),
                                              out_u8.stride(0)), (result01, target, inpaint, mask, ref, out_u8), name)


def resize_u8_linear(x_u8, out, name="resize_u8_linear"):
    """x_u8 [B, H, W, C] uint8 (HWC, contiguous images) -> out [B, Ho, Wo, C] uint8: cv2.resize(..., INTER_LINEAR) bit for bit."""
    lib = _lib.load()
    _require_gpu(x_u8, out)
    B, H, W_, Cc = x_u8.shape
    assert x_u8.dtype == torch.uint8 and out.dtype == torch.uint8 and out.is_contiguous() and out.shape[0] == B and out.shape[3] == Cc
    assert x_u8.stride(3) == 1 and x_u8.stride(2) == Cc and x_u8.stride(1) == W_ * Cc
    return Launch(lib.rf_resize_u8_linear, (_p(x_u8), B, H, W_, Cc, x_u8.stride(0), out.shape[1], out.shape[2], _p(out)), (x_u8, out), name)


def label_mask(labels_u8, lut256, out, *, invert, name="label_mask"):
    lib = _lib.load()
    _require_gpu(labels_u8, lut256, out)
    assert labels_u8.dtype == torch.uint8 and lut256.dtype == torch.uint8 and lut256.numel() == 256 and out.dtype == torch.float32
    assert labels_u8.is_contiguous() and out.is_contiguous() and out.numel() == labels_u8.numel()
    return Launch(lib.rf_label_mask, (_p(labels_u8), labels_u8.numel(), _p(lut256), int(bool(invert)), _p(out)), (labels_u8, lut256, out), name)


def mul_mask(x, mask, out, name="mul_mask"):
    lib = _lib.load()
    _require_gpu(x, mask, out)
    B, Cc, H, W_ = x.shape
    assert x.is_contiguous() and mask.is_contiguous() and out.is_contiguous() and mask.numel() == B * H * W_
    return Launch(lib.rf_mul_mask, (_p(x), _p(mask), B, Cc, H * W_, _p(out)), (x, mask, out), name)


def gaussian_sample(moments, eps, out, *, scale, name="gaussian_sample"):
    lib = _lib.load()
    _require_gpu(moments, eps, out)
    B, C2, H, W_ = moments.shape
    assert moments.is_contiguous() and out.is_contiguous() and (eps is None or eps.is_contiguous())
    return Launch(lib.rf_gaussian_sample, (_p(moments), _p(eps), float(scale), _p(out), B, C2 // 2, H * W_), (moments, eps, out), name)


def run(launches, stream=None):
    s = stream if stream is not None else stream_ptr()
    for l in launches:
        l(s)


# ------------------------------------------------------------------------------------------------
# conditioning-encoder side kernels (reface_amd/csrc/encoder.hip)
# ------------------------------------------------------------------------------------------------
def channel_affine(x, a, b, out, slope=None, name="channel_affine"):
    """out[..., c] = prelu(x[..., c] * a[c] + b[c]); x/out channels-last (any leading dims, uniform pixel pitch)."""
    lib = _lib.load()
    _require_gpu(x, a, b, out, slope)
    Cc = x.shape[-1]
    M = x.numel() // Cc
    return Launch(lib.rf_channel_affine, (code(x.dtype), _p(x), x.stride(-2), _p(a), _p(b), _p(slope), code(out.dtype), _p(out),
                                          out.stride(-2), M, Cc), (x, a, b, slope, out), name)


def spatial_mean(x, out, name="spatial_mean"):
    lib = _lib.load()
    _require_gpu(x, out)
    B, H, W_, Cc = x.shape
    assert out.dtype == torch.float32 and out.is_contiguous()
    return Launch(lib.rf_spatial_mean, (code(x.dtype), _p(x), B, H * W_, Cc, x.stride(2), _p(out)), (x, out), name)


def se_scale_add(r, s, shortcut, out, *, stride, name="se_scale_add"):
    lib = _lib.load()
    _require_gpu(r, s, shortcut, out)
    B, Ho, Wo, Cc = r.shape
    assert r.is_contiguous() and out.is_contiguous() and s.dtype == torch.float32 and shortcut.dtype == r.dtype
    return Launch(lib.rf_se_scale_add, (code(r.dtype), _p(r), _p(s), _p(shortcut), shortcut.stride(2), shortcut.shape[1], shortcut.shape[2],
                                        stride, _p(out), B, Ho, Wo, Cc), (r, s, shortcut, out), name)


def adaptive_avgpool(x, out, *, crop=None, a=None, b=None, nhwc=False, name="adaptive_avgpool"):
    """x: NCHW fp32; crop = (y0, x0, h, w) window (default full).  out: NCHW fp32 [B,C,Ho,Wo] or (nhwc) [B,Ho,Wo,Cpad]."""
    lib = _lib.load()
    _require_gpu(x, out, a, b)
    B, Cc, Hf, Wf = x.shape
    y0, x0, hc, wc = crop if crop is not None else (0, 0, Hf, Wf)
    Ho, Wo = (out.shape[1], out.shape[2]) if nhwc else (out.shape[2], out.shape[3])
    assert x.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous()
    return Launch(lib.rf_adaptive_avgpool, (_p(x), B, Cc, Hf, Wf, y0, x0, hc, wc, _p(a), _p(b), Ho, Wo, int(nhwc), code(out.dtype),
                                            out.shape[-1] if nhwc else Cc, _p(out)), (x, out, a, b), name)


def bilinear_resize(x, out, a=None, b=None, name="bilinear_resize"):
    lib = _lib.load()
    _require_gpu(x, out, a, b)
    B, Cc, Hi, Wi = x.shape
    assert x.dtype == torch.float32 and out.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous()
    return Launch(lib.rf_bilinear_resize, (_p(x), B, Cc, Hi, Wi, _p(a), _p(b), out.shape[2], out.shape[3], _p(out)), (x, out, a, b), name)


def clip_tokens(patch, cls, pos, out, name="clip_tokens"):
    lib = _lib.load()
    _require_gpu(patch, cls, pos, out)
    B, NP, Cc = patch.shape
    assert patch.is_contiguous() and out.is_contiguous() and patch.dtype == out.dtype
    return Launch(lib.rf_clip_tokens, (code(patch.dtype), _p(patch), _p(cls), _p(pos), _p(out), B, NP, Cc), (patch, cls, pos, out), name)


def l2norm_rows(x, out, name="l2norm_rows"):
    lib = _lib.load()
    _require_gpu(x, out)
    assert x.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous()
    return Launch(lib.rf_l2norm_rows, (_p(x), _p(out), x.shape[0], x.shape[1]), (x, out), name)


def combine3(a, b, c, out, *, wa, wb, wc, den, name="combine3"):
    lib = _lib.load()
    _require_gpu(a, b, c, out)
    for t in (a, b, c, out):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    return Launch(lib.rf_combine3, (_p(a), _p(b), _p(c), float(wa), float(wb), float(wc), float(den), _p(out), a.numel()), (a, b, c, out), name)
