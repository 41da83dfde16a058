"""GPU parity tests of the individual HIP kernels (through the C-ABI) against plain PyTorch fp32
CPU references of the same op.  fp32 kernels: round-off-class tolerances; bf16 kernels: compared
against the fp32 reference evaluated on the bf16-rounded inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

from reface_amd import ops
from reface_amd.params import seeded_randn as rnd

from e4m3_ref import e4m3fn_decode_np as _e4m3fn_decode_np, e4m3fn_encode_rne_sat_np as _e4m3fn_encode_rne_sat_np

pytestmark = pytest.mark.gpu
DEV = "cuda"


DTS = [torch.float32, torch.bfloat16, torch.float16]          # fp16 = the "fp16" throughput mode: the bf16 kernels instantiated for f16_t
H16 = [torch.bfloat16, torch.float16]
STEP = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}          # relative rounding step of the 16-bit storage types


def tol(dt):
    """(atol, rtol): ~8 / ~5 units of the storage type's rounding step 2^-8 (bf16) resp. 2^-11 (fp16)"""
    return (2e-5, 2e-5) if dt == torch.float32 else ((3e-2, 2e-2) if dt == torch.bfloat16 else (4e-3, 2.5e-3))


def check(got, ref, dt, scale=1.0):
    got = got.float().cpu()
    atol, rtol = tol(dt)
    err = (got - ref).abs()
    lim = atol * scale + rtol * ref.abs()
    assert torch.isfinite(got).all()
    assert (err <= lim).all(), f"max err {err.max():.3e} at ref absmax {ref.abs().max():.3e} (dtype {dt})"


def q(x, dt):
    """round to the storage dtype, return (device tensor in dt, fp32 cpu copy of the rounded values)"""
    xd = x.to(dt)
    return xd.to(DEV), xd.float()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 320, 320), (77, 200, 136), (16, 1280, 320), (257, 96, 1024), (1, 768, 512)])
def test_linear(dt, M, N, K):
    x, xr = q(rnd((M, K), 1), dt)
    w, wr = q(rnd((N, K), 2) / math.sqrt(K), dt)
    b = rnd((N,), 3)
    res, rr = q(rnd((M, N), 4), dt)
    out = torch.empty((M, N), dtype=dt, device=DEV)
    ops.linear(x, w, out, b.to(DEV), residual=res)()
    torch.cuda.synchronize()
    check(out, F.linear(xr, wr, b) + rr, dt)
    # asymmetric identity check (transposition detector): W = I-like with distinct rows
    out2 = torch.empty((M, N), dtype=torch.float32, device=DEV)
    ops.linear(x, w, out2, None, act=ops.ACT_SILU)()
    torch.cuda.synchronize()
    check(out2, F.silu(F.linear(xr, wr)), dt)


@pytest.mark.parametrize("dt", DTS)
def test_linear_strided_views_and_rowvec(dt):
    M, K, N, B = 96, 64, 160, 3
    buf, bufr = q(rnd((M, 3 * K), 5), dt)
    w, wr = q(rnd((N, K), 6) / 8, dt)
    rv = rnd((B, N), 7)
    outbuf = torch.zeros((M, 2 * N), dtype=dt, device=DEV)
    ops.linear(buf[:, K:2 * K], w, outbuf[:, N:], None, rowvec=rv.to(DEV), rows_per_sample=M // B)()
    torch.cuda.synchronize()
    ref = F.linear(bufr[:, K:2 * K], wr) + rv.repeat_interleave(M // B, 0)
    check(outbuf[:, N:], ref, dt)
    assert (outbuf[:, :N] == 0).all()


@pytest.mark.parametrize("dt", DTS)
def test_geglu(dt):
    M, Cc, Fh = 200, 64, 128
    x, xr = q(rnd((M, Cc), 8), dt)
    w = rnd((2 * Fh, Cc), 9) / 8
    b = rnd((2 * Fh,), 10)
    wp, bp = ops.pack_geglu(w, b, dt)
    out = torch.empty((M, Fh), dtype=dt, device=DEV)
    ops.linear(x, wp.to(DEV), out, bp.to(DEV), act=ops.ACT_GEGLU)()
    torch.cuda.synchronize()
    h = F.linear(xr, w.to(dt).float(), b)
    a, g = h.chunk(2, -1)
    check(out, a * F.gelu(g), dt)


def _conv_ref(xr, wr, b, stride, pad4, ups):
    x = xr.permute(0, 3, 1, 2)
    if ups:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    x = F.pad(x, pad4)
    return F.conv2d(x, wr, b, stride=stride).permute(0, 2, 3, 1)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", ["s1", "s2", "asym", "ups", "cat", "cin16", "k1"])
def test_conv(dt, case):
    B, H, W_, Ci, Co = 2, 12, 10, 64, 96
    stride, pad4, ups, ks, x2 = 1, (1, 1, 1, 1), 0, 3, None
    if case == "s2":
        stride = 2
    elif case == "asym":
        stride, pad4 = 2, (0, 1, 0, 1)
    elif case == "ups":
        ups = 1
    elif case == "cin16":
        Ci = 16
    elif case == "k1":
        ks, pad4 = 1, (0, 0, 0, 0)
    x, xr = q(rnd((B, H, W_, Ci), 11), dt)
    Ct = Ci
    if case == "cat":
        x2, x2r = q(rnd((B, H, W_, 32), 12), dt)
        Ct = Ci + 32
        xr = torch.cat([xr, x2r], -1)
    w = rnd((Co, Ct, ks, ks), 13) / math.sqrt(Ct * ks * ks)
    b = rnd((Co,), 14)
    wr = w.to(dt).float()
    ref = _conv_ref(xr, wr, b, stride, pad4, ups)
    out = torch.empty(ref.shape, dtype=dt, device=DEV)
    ops.conv2d(x, ops.pack_conv_weight(w, dt).to(DEV), out, b.to(DEV), ksize=ks, stride=stride, pad=(pad4[2], pad4[0]), ups=ups, x2=x2)()
    torch.cuda.synchronize()
    check(out, ref, dt)


@pytest.mark.parametrize("dt", DTS)
def test_conv_padded_cin_and_epilogue(dt):
    """9 real input channels stored in 16 (UNet conv_in), temb row-vector + residual epilogue."""
    B, H, W_, Co = 2, 8, 8, 320
    x9 = rnd((B, H, W_, 9), 15)
    x = torch.zeros((B, H, W_, 16))
    x[..., :9] = x9
    x, xr = q(x, dt)
    w = rnd((Co, 9, 3, 3), 16) / 9
    b = rnd((Co,), 17)
    rv = rnd((B, Co), 18)
    res, rr = q(rnd((B, H, W_, Co), 19), dt)
    out = torch.empty((B, H, W_, Co), dtype=dt, device=DEV)
    ops.conv2d(x, ops.pack_conv_weight(w, dt, cin_pad=16).to(DEV), out, b.to(DEV), rowvec=rv.to(DEV), residual=res)()
    torch.cuda.synchronize()
    ref = _conv_ref(xr[..., :9], w.to(dt).float(), b, 1, (1, 1, 1, 1), 0) + rv[:, None, None, :] + rr
    check(out, ref, dt)


@pytest.mark.parametrize("dt", DTS)
def test_batched_gemm(dt):
    Bt, M, N, K = 3, 100, 72, 64
    a, ar = q(rnd((Bt, M, K), 20), dt)
    w, wr = q(rnd((Bt, N, K), 21) / 8, dt)
    out = torch.empty((Bt, M, N), dtype=torch.float32, device=DEV)
    ops.conv_gemm(a, w, out, M=M, N=N, K=K, C0=K, ld0=K, Hin=1, Win=M, Hout=1, Wout=M, ldo=N, alpha=0.5, batch=Bt,
                  sA=M * K, sW=N * K, sO=M * N)()
    torch.cuda.synchronize()
    check(out, 0.5 * torch.einsum("bmk,bnk->bmn", ar, wr), dt)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("Cc,hw,silu,eps", [(320, 16, True, 1e-5), (640, 9, False, 1e-6), (1280, 8, True, 1e-5), (2560, 8, True, 1e-5), (128, 32, True, 1e-6)])
def test_groupnorm(dt, Cc, hw, silu, eps):
    B = 3
    x, xr = q(rnd((B, hw, hw, Cc), 22) * 2 + 0.7, dt)
    g = rnd((Cc,), 23) * 0.2 + 1
    be = rnd((Cc,), 24) * 0.2
    out = torch.empty((B, hw, hw, Cc), dtype=dt, device=DEV)
    part = torch.empty(B * ops.GN_MAX_CHUNKS * 64, dtype=torch.float64, device=DEV)
    ops.run(ops.groupnorm(x, g.to(DEV), be.to(DEV), out, part, eps=eps, silu=silu))
    torch.cuda.synchronize()
    ref = F.group_norm(xr.permute(0, 3, 1, 2), 32, g, be, eps)
    if silu:
        ref = F.silu(ref)
    check(out, ref.permute(0, 2, 3, 1), dt)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("c0,c1,hw,B,kin", [(320, 0, 32, 2, 64), (640, 320, 32, 2, 64), (1280, 640, 16, 4, 64), (1280, 640, 16, 4, 2048), (1280, 1280, 8, 4, 2048), (128, 0, 128, 2, 64)])
def test_groupnorm_stats_fused_into_gemm(dt, c0, c1, hw, B, kin):
    """GroupNorm whose statistics come out of the producing GEMMs' epilogues (rf_conv_gemm gn_* fields): a concat buffer
    [h | skip] written by two GEMMs (the skip one also feeds a second consumer with another grouping), vs torch on the stored tensor."""
    Cc = c0 + c1
    cat = torch.zeros((B, hw, hw, Cc), dtype=dt, device=DEV)
    prods = []
    for i, (off, c) in enumerate(((0, c0), (c0, c1))):
        if c == 0:
            continue
        xin, _ = q(rnd((B, hw, hw, kin), 90 + i), dt)
        w, _ = q(rnd((c, kin), 92 + i) / math.sqrt(kin), dt)
        bias = rnd((c,), 94 + i).to(DEV)
        res, _ = q(rnd((B, hw, hw, c), 96 + i), dt)
        out = cat[..., off:off + c]
        l = ops.conv2d(xin, w.reshape(c, kin), out, bias, ksize=1, pad=(0, 0), residual=res, name=f"prod{i}")
        prods.append((l, 0, B * hw * hw, off, c))
    fused = ops.fuse_groupnorm_stats(cat, prods)
    bm, bn, sk = ops.gemm_plan(prods[0][0])
    if (hw * hw) % bm:
        assert fused is None
        pytest.skip(f"plan bm={bm} splitk={sk} cannot fuse at this size")
    assert fused is not None
    second = None
    if c1:      # the skip half alone is a second consumer (different channels per group)
        second = ops.fuse_groupnorm_stats(cat[..., c0:], [(prods[1][0], 0, B * hw * hw, 0, c1)])
        assert second is not None
    ops.run([p_[0] for p_ in prods])
    ops.run(fused[2])
    g, be = rnd((Cc,), 98) * 0.2 + 1, rnd((Cc,), 99) * 0.2
    y = torch.empty_like(cat)
    ops.groupnorm_apply(cat, g.to(DEV), be.to(DEV), y, fused[0], fused[1], eps=1e-5, silu=True)()
    torch.cuda.synchronize()
    ref = F.silu(F.group_norm(cat.float().cpu().permute(0, 3, 1, 2), 32, g, be, 1e-5)).permute(0, 2, 3, 1)
    check(y, ref, dt)
    if second is not None:
        ops.run(second[2])
        y2 = torch.empty((B, hw, hw, c1), dtype=dt, device=DEV)
        ops.groupnorm_apply(cat[..., c0:], g[:c1].to(DEV), be[:c1].to(DEV), y2, second[0], second[1], eps=1e-5, silu=False)()
        torch.cuda.synchronize()
        ref2 = F.group_norm(cat[..., c0:].float().cpu().permute(0, 3, 1, 2), 32, g[:c1], be[:c1], 1e-5).permute(0, 2, 3, 1)
        check(y2, ref2, dt)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("Cc", [320, 768, 1024, 1280])
def test_layernorm(dt, Cc):
    M = 131
    x, xr = q(rnd((M, Cc), 25) * 1.5 - 0.3, dt)
    g = rnd((Cc,), 26) * 0.2 + 1
    be = rnd((Cc,), 27) * 0.2
    out = torch.empty((M, Cc), dtype=dt, device=DEV)
    ops.layernorm(x, g.to(DEV), be.to(DEV), out, eps=1e-5)()
    torch.cuda.synchronize()
    check(out, F.layer_norm(xr, (Cc,), g, be, 1e-5), dt)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("heads,d,N", [(8, 40, 256), (8, 80, 144), (8, 160, 64), (2, 160, 200), (16, 64, 257), (4, 8, 36), (8, 40, 1024)])
def test_attention(dt, heads, d, N):
    B = 2
    Cc = heads * d
    qkv, qkvr = q(rnd((B, N, 3 * Cc), 28), dt)
    out = torch.empty((B, N, Cc), dtype=dt, device=DEV)
    scale = d ** -0.5
    ops.attention(qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:], out, heads=heads, scale=scale)()
    torch.cuda.synchronize()
    sp = lambda t: t.reshape(B, N, heads, d).transpose(1, 2)
    qq, kk, vv = sp(qkvr[..., :Cc]), sp(qkvr[..., Cc:2 * Cc]), sp(qkvr[..., 2 * Cc:])
    att = torch.softmax(qq @ kk.transpose(-1, -2) * scale, -1)
    ref = (att @ vv).transpose(1, 2).reshape(B, N, Cc)
    check(out, ref, dt)


@pytest.mark.parametrize("heads,d,N", [(8, 40, 256), (8, 80, 144), (8, 160, 64), (2, 160, 200), (16, 64, 257), (4, 8, 36), (8, 40, 1024), (8, 40, 4096)])
def test_attention_x3_vs_fp64(heads, d, N):
    """rf_attention dtype RF_BF16X3 (the attention of the "f32x3" parity mode): fp32 tensors, both contractions on split-bf16 operand
    pairs (three bf16 MFMA passes), fp32 softmax -- against an fp64 reference (attention.py:206-220), beside the exact-fp32 kernel's
    distance to the same reference.  Logits of realistic spread (|s| up to ~10 in the exp2 domain)."""
    B = 2
    C_ = heads * d
    qkv = (rnd((B, N, 3 * C_), 130) * 1.0).to(DEV)
    qq, kk, vv = qkv[..., :C_], qkv[..., C_:2 * C_], qkv[..., 2 * C_:]
    scale = d ** -0.5 * 2.0
    out = torch.empty((B, N, C_), dtype=torch.float32, device=DEV)
    out32 = torch.empty_like(out)
    ops.attention(qq, kk, vv, out, heads=heads, scale=scale, x3=True)()
    ops.attention(qq, kk, vv, out32, heads=heads, scale=scale)()
    torch.cuda.synchronize()
    q64, k64, v64 = (t.double().cpu().reshape(B, N, heads, d).permute(0, 2, 1, 3) for t in (qq, kk, vv))
    att = torch.softmax(q64 @ k64.transpose(-1, -2) * scale, dim=-1)
    ref = (att @ v64).permute(0, 2, 1, 3).reshape(B, N, C_)
    e3 = (out.double().cpu() - ref).abs().max().item()
    e32 = (out32.double().cpu() - ref).abs().max().item()
    print(f"attention x3 (heads {heads}, d {d}, N {N}): max |d| vs fp64 = {e3:.2e} (exact-fp32 kernel: {e32:.2e}; |out| max {ref.abs().max().item():.2f})")
    assert torch.isfinite(out).all() and e3 < 1e-4 * max(1.0, ref.abs().max().item()), (e3, e32)


def test_attention_spike_forces_rescale():
    """Online-softmax rescale branch: one key dominates from a late tile (guide rule 26)."""
    B, heads, d, N = 1, 1, 40, 320
    qkv = rnd((B, N, 3 * d), 29)
    qkv[0, 5, :d] *= 6.0
    qkv[0, 300, d:2 * d] = qkv[0, 5, :d] * 2.0          # key 300 aligned with query 5
    x = qkv.to(DEV)
    out = torch.empty((B, N, d), dtype=torch.float32, device=DEV)
    ops.attention(x[..., :d], x[..., d:2 * d], x[..., 2 * d:], out, heads=1, scale=d ** -0.5)()
    torch.cuda.synchronize()
    att = torch.softmax((qkv[..., :d] @ qkv[..., d:2 * d].transpose(-1, -2)).double() * d ** -0.5, -1)
    ref = (att @ qkv[..., 2 * d:].double()).float()
    check(out, ref, torch.float32)


def _attn_ref(qr, kr, vr, heads, d, scale):
    B, Nq, Nk = qr.shape[0], qr.shape[1], kr.shape[1]
    sp = lambda t, n: t.reshape(B, n, heads, d).transpose(1, 2).double()
    att = torch.softmax(sp(qr, Nq) @ sp(kr, Nk).transpose(-1, -2) * scale, -1)
    return (att @ sp(vr, Nk)).transpose(1, 2).reshape(B, Nq, heads * d).float()


@pytest.mark.parametrize("d,Nq,Nk,mode", [(40, 4096, 4096, "plain"), (40, 4000, 1088, "plain"), (40, 3900, 1024, "hot"), (40, 4096, 1152, "cold"), (40, 4096, 1024, "late"),
                                          (80, 1024, 1024, "plain"), (80, 1000, 1152, "plain"), (80, 900, 1024, "hot"), (80, 1024, 384, "cold"), (80, 1024, 1024, "late"),
                                          (80, 2304, 2304, "plain"), (80, 70, 512, "hot"), (40, 4096, 1024, "frozen"), (80, 1024, 1024, "frozen")])
@pytest.mark.parametrize("dt", H16)
def test_attention_pipelined_kernels(d, Nq, Nk, mode, dt):
    """attention_dma_kernel (16-bit; d = 40 with Nk a multiple of 64 >= 1024 and a grid >= 512 blocks, d = 80 with Nk a multiple of 128 >= 384): ragged
    query counts, a key count that is not a multiple of 128, the shortest ring (3 tiles), scores far above 2^8 in the exp2 domain (the thresholded running
    max must move, more than once), scores that are all very negative (the first unit has to LOWER the reference point from 0) and a dominant key in the
    last tile.  d = 40 carries the reference point in a padded k-slot of Q, d = 80 as the C operand of a unit's first QK^T MFMA."""
    B, heads = 4, 8
    Cc = heads * d
    qx, kx, vx = rnd((B, Nq, Cc), 41), rnd((B, Nk, Cc), 42), rnd((B, Nk, Cc), 43)
    scale = d ** -0.5
    if mode == "hot":
        # logits in the tens (exp2 domain): several moves of the thresholded running max.  q arrives PRE-SCALED (scale = ln 2), the way
        # the UNet calls the kernel: with a plain scale the kernel rounds q * scale * log2(e) to bf16 once more, a 2^-9-relative
        # perturbation of the logits that peaked softmaxes turn into percent-level differences (covered by "plain" at unit variance)
        qx *= 2.5 * 2.5 * scale * ops.LOG2E
        scale = ops.LN2
    elif mode in ("cold", "frozen"):
        f = 3.0 if mode == "cold" else 5.0                   # "frozen": every score below -128 in the exp2 domain (2^-delta of the first unit's move overflows fp32)
        kx = -qx[:, :1].repeat(1, Nk, 1) * f + 0.1 * kx       # every key anti-aligned with query 0 ...
        qx = qx[:, :1].repeat(1, Nq, 1) * f + 0.1 * qx        # ... and every query close to it: all scores << 0
    elif mode == "late":
        kx[:, Nk - 3] = qx[:, 7] * 4.0                       # one key of the last tile dominates query 7
    qd, qr = q(qx, dt)
    kd, kr = q(kx, dt)
    vd, vr = q(vx, dt)
    out = torch.empty((B, Nq, Cc), dtype=dt, device=DEV)
    ops.attention(qd, kd, vd, out, heads=heads, scale=scale)()
    torch.cuda.synchronize()
    ref = _attn_ref(qr, kr, vr, heads, d, scale)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    k = STEP[dt] / STEP[torch.bfloat16]          # (fp16: the probabilities and the output carry 11 bits instead of 8)
    assert err.max() < 3e-2 * k and (err.norm() / ref.norm()) < 8e-3 * k, f"{mode} {dt}: max {err.max():.3e} rel {err.norm() / ref.norm():.3e}"


def test_softmax_rows():
    x = rnd((37, 4096), 30) * 3
    xd = x.to(DEV)
    ops.softmax_rows(xd)()
    torch.cuda.synchronize()
    check(xd, torch.softmax(x, -1), torch.float32, scale=0.01)


def test_ddim_glue():
    B, h = 2, 8
    img, z, m = rnd((B, 4, h, h), 31), rnd((B, 4, h, h), 32), (rnd((B, 1, h, h), 33) > 0).float()
    for dt in (torch.float32, torch.bfloat16):
        x_in = torch.empty((2 * B, h, h, 16), dtype=dt, device=DEV)
        ops.ddim_pack_input(img.to(DEV), z.to(DEV), m.to(DEV), x_in, dup=2)()
        torch.cuda.synchronize()
        ref = torch.cat([img, z, m], 1).permute(0, 2, 3, 1)
        got = x_in.float().cpu()
        assert torch.equal(got[:B, ..., :9], ref.to(dt).float()) and torch.equal(got[B:, ..., :9], ref.to(dt).float())
        assert (got[..., 9:] == 0).all()
    eps = rnd((2 * B, h, h, 64), 34)
    a_t, a_prev, sig = 0.5, 0.6, 0.1
    noise = rnd((B, 4, h, h), 35)
    imgd = img.to(DEV).clone()
    px0 = torch.empty_like(imgd)
    coefs = torch.tensor([math.sqrt(a_t), math.sqrt(1 - a_t), math.sqrt(a_prev), math.sqrt(1 - a_prev - sig ** 2), sig],
                         dtype=torch.float32, device=DEV)
    ops.ddim_update(eps.to(DEV), imgd, px0, noise.to(DEV), coefs, cfg=True, scale=3.5)()
    torch.cuda.synchronize()
    e = eps[..., :4].permute(0, 3, 1, 2)
    e = e[:B] + 3.5 * (e[B:] - e[:B])
    r0 = (img - math.sqrt(1 - a_t) * e) / math.sqrt(a_t)
    rp = math.sqrt(a_prev) * r0 + math.sqrt(1 - a_prev - sig ** 2) * e + sig * noise
    check(px0, r0, torch.float32)
    check(imgd, rp, torch.float32)


def test_layout_and_embedding():
    x = rnd((2, 5, 6, 7), 36)
    out = torch.empty((2, 6, 7, 8), dtype=torch.float32, device=DEV)
    ops.nchw_to_nhwc(x.to(DEV), out)()
    back = torch.empty((2, 5, 6, 7), dtype=torch.float32, device=DEV)
    ops.nhwc_to_nchw(out, back)()
    torch.cuda.synchronize()
    assert torch.equal(out.cpu()[..., :5], x.permute(0, 2, 3, 1)) and (out.cpu()[..., 5:] == 0).all()
    assert torch.equal(back.cpu(), x)
    t = torch.tensor([981.0, 1.0, 500.0, 21.0])
    half = 160
    freqs = torch.exp(-math.log(10000.0) * torch.arange(0, half, dtype=torch.float32) / half)
    emb = torch.empty((4, 320), dtype=torch.float32, device=DEV)
    ops.timestep_embedding(t.to(DEV), freqs.to(DEV), emb)()
    torch.cuda.synchronize()
    args = t[:, None] * freqs[None]
    check(emb, torch.cat([torch.cos(args), torch.sin(args)], -1), torch.float32, scale=0.1)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("stride,ups,Ci", [(1, 0, 128), (2, 0, 64), (1, 1, 192)])
def test_conv_channel_chunk_major_k(dt, stride, ups, Ci):
    """korder=1: K runs (channel chunk, tap, channel-in-chunk); weights packed to match (ops.pack_conv_weight)."""
    B, H, W_, Co = 2, 10, 12, 160
    assert ops.conv_korder(Ci, dt) == 1
    x, xr = q(rnd((B, H, W_, Ci), 40), dt)
    w = rnd((Co, Ci, 3, 3), 41) / math.sqrt(Ci * 9)
    b = rnd((Co,), 42)
    ref = _conv_ref(xr, w.to(dt).float(), b, stride, (1, 1, 1, 1), ups)
    out = torch.empty(ref.shape, dtype=dt, device=DEV)
    ops.conv2d(x, ops.pack_conv_weight(w, dt, korder=1).to(DEV), out, b.to(DEV), stride=stride, ups=ups, korder=1)()
    torch.cuda.synchronize()
    check(out, ref, dt)


@pytest.mark.parametrize("B,hw,Ci,Co", [(2, 32, 128, 320), (16, 8, 1280, 1280), (4, 16, 640, 1280), (2, 64, 320, 320), (16, 64, 320, 320), (3, 8, 128, 320),
                                        (16, 32, 640, 640), (1, 16, 64, 64), (5, 16, 1280, 640)])
@pytest.mark.parametrize("dt", H16)
def test_conv_row_extended_a_tiles(B, hw, Ci, Co, dt):
    """rf_conv_gemm korder 2 (gemm.hip HX, round 4): 3x3 stride-1 convolutions with the K order (filter row, channel chunk, filter column) -- the
    three horizontal taps of a (row, chunk) share ONE row-extended A tile (every image row of the output tile + a halo pixel on each side).
    Image borders (zero halo, top / bottom rows), tiles spanning several samples (8x8: 128-row tile = 2 samples), ragged M, split-K, and the
    epilogue's bias / per-sample vector / residual / fused GroupNorm statistics -- against F.conv2d on the bf16-rounded operands, and bit-equal
    to the tap-major launch of the same layer where the summation order per accumulator is the same K-tile sequence permuted (checked to 2 ulps)."""
    x, xr = q(rnd((B, hw, hw, Ci), 160) * 0.5, dt)
    w = rnd((Co, Ci, 3, 3), 161) / math.sqrt(Ci * 9)
    b = rnd((Co,), 162)
    rv = rnd((B, Co), 163)
    res, rr = q(rnd((B, hw, hw, Co), 164), dt)
    out = torch.empty((B, hw, hw, Co), dtype=dt, device=DEV)
    l = ops.conv2d(x, ops.pack_conv_weight(w, dt, korder=2).to(DEV), out, b.to(DEV), rowvec=rv.to(DEV), residual=res, korder=2)
    try:
        pl = ops.gemm_plan2(l)
    except Exception as e:
        pytest.skip(f"this launch's tile cannot take korder 2: {e}")
    assert pl["bm"] % hw == 0, pl
    fused = ops.fuse_groupnorm_stats(out, [(l, 0, B * hw * hw, 0, Co)])
    l()
    out0 = torch.empty_like(out)
    ops.conv2d(x, ops.pack_conv_weight(w, dt).to(DEV), out0, b.to(DEV), rowvec=rv.to(DEV), residual=res)()
    torch.cuda.synchronize()
    ref = (F.conv2d(xr.to(DEV).permute(0, 3, 1, 2), w.to(dt).float().to(DEV), b.to(DEV), padding=1).permute(0, 2, 3, 1).cpu() + rv[:, None, None, :] + rr)
    check(out, ref, dt)
    d01 = (out.float() - out0.float()).abs().max().item()
    print(f"korder 2 vs tap-major (B={B}, {hw}x{hw}, {Ci}->{Co}): tile {pl['bm']}x{pl['bn']} splitk {pl['splitk']}, max |d| = {d01:.3e} at |out| max {ref.abs().max().item():.2f}")
    assert d01 <= 4 * STEP[dt] * max(1.0, ref.abs().max().item())
    if fused is not None:
        ops.run(fused[2])
        g, be = rnd((Co,), 165) * 0.2 + 1, rnd((Co,), 166) * 0.2
        y = torch.empty_like(out)
        ops.groupnorm_apply(out, g.to(DEV), be.to(DEV), y, fused[0], fused[1], eps=1e-5, silu=True)()
        torch.cuda.synchronize()
        refn = F.silu(F.group_norm(out.float().cpu().permute(0, 3, 1, 2), 32, g, be, 1e-5)).permute(0, 2, 3, 1)
        check(y, refn, dt)


@pytest.mark.parametrize("dt", DTS)
def test_conv_split_k(dt):
    """Small-M / long-K conv (8x8 level of the UNet): takes the split-K path (partials in the workspace + reduce pass)."""
    B, H, W_, Ci, Co = 2, 8, 8, 1280, 320
    x, xr = q(rnd((B, H, W_, Ci), 43), dt)
    w = rnd((Co, Ci, 3, 3), 44) / math.sqrt(Ci * 9)
    b = rnd((Co,), 45)
    rv = rnd((B, Co), 46)
    res, rr = q(rnd((B, H, W_, Co), 47), dt)
    ref = _conv_ref(xr, w.to(dt).float(), b, 1, (1, 1, 1, 1), 0) + rv[:, None, None, :] + rr
    out = torch.empty(ref.shape, dtype=dt, device=DEV)
    ops.conv2d(x, ops.pack_conv_weight(w, dt).to(DEV), out, b.to(DEV), rowvec=rv.to(DEV), residual=res)()
    torch.cuda.synchronize()
    check(out, ref, dt)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,hw,Ci,Co", [(16, 16, 704, 1280), (16, 8, 1280, 1280), (4, 32, 1280, 640)])
def test_conv_split_k_fragment_slabs(dt, B, hw, Ci, Co):
    """Split-K through fragment-ordered slabs (direct-epilogue kernels + splitk_reduce_frag_kernel): the 3x3 convs of the 16x16 / 8x8 levels and the
    long-K N = 640 convs of the 32x32 level, with everything the reduce pass carries -- bias, per-sample vector, residual and the fused
    GroupNorm statistics of the stored values.  (A finish INSIDE the GEMM launch, spread over the K slices of a tile, was built and measured in
    round 5 -- correct, +1.5 % per batch: profiles/r05d_splitk_spread_finish.patch / .txt; the reduce pass stays.)"""
    x, xr = q(rnd((B, hw, hw, Ci), 143) * 0.5, dt)
    w = rnd((Co, Ci, 3, 3), 144) / math.sqrt(Ci * 9)
    b = rnd((Co,), 145)
    rv = rnd((B, Co), 146)
    res, rr = q(rnd((B, hw, hw, Co), 147), dt)
    out = torch.empty((B, hw, hw, Co), dtype=dt, device=DEV)
    l = ops.conv2d(x, ops.pack_conv_weight(w, dt).to(DEV), out, b.to(DEV), rowvec=rv.to(DEV), residual=res)
    bm, bn, sk = ops.gemm_plan(l)
    assert sk > 1 and bm == 32 and bn in (160, 128), (bm, bn, sk)          # the stripe of the fragment reduce pass
    assert ops.gemm_plan2(l)["frag"] == 1, ops.gemm_plan2(l)
    fused = ops.fuse_groupnorm_stats(out, [(l, 0, B * hw * hw, 0, Co)])
    assert fused is not None
    l()
    ops.run(fused[2])
    g, be = rnd((Co,), 148) * 0.2 + 1, rnd((Co,), 149) * 0.2
    y = torch.empty_like(out)
    ops.groupnorm_apply(out, g.to(DEV), be.to(DEV), y, fused[0], fused[1], eps=1e-5, silu=True)()
    torch.cuda.synchronize()
    wq = w.to(dt).float()
    ref = (F.conv2d(xr.to(DEV).permute(0, 3, 1, 2), wq.to(DEV), b.to(DEV), padding=1).permute(0, 2, 3, 1).cpu() + rv[:, None, None, :] + rr)
    check(out, ref, dt)
    refn = F.silu(F.group_norm(out.float().cpu().permute(0, 3, 1, 2), 32, g, be, 1e-5)).permute(0, 2, 3, 1)
    check(y, refn, dt)
    # run-to-run identical (fixed summation order over the z slices), also back to back
    o1 = out.clone()
    for _ in range(4):
        l()
    torch.cuda.synchronize()
    assert torch.equal(o1, out)


@pytest.mark.parametrize("dt", DTS)
def test_split_k_fragment_slabs_ragged_m_with_fused_stats(dt):
    """M < BM with split-K through fragment slabs AND fused GroupNorm statistics (CFG off, one sample at the 8x8 level: M = 64 on a 128-row
    tile): the reduce pass's stripes with row0 >= M have no statistics slot.  Two launches, one per sample, write one [2, 8, 8, C] tensor;
    sample 1's producer runs FIRST -- a stripe of sample 0's launch that wrote a slot beyond its M would zero sample 1's statistics
    (or, for the last sample, write past the end of the buffer: the round-3 advisor finding)."""
    B, hw, Ci, Co = 2, 8, 1280, 320
    x, xr = q(rnd((B, hw, hw, Ci), 150) * 0.5, dt)
    w = rnd((Co, Ci, 3, 3), 151) / math.sqrt(Ci * 9)
    b = rnd((Co,), 152)
    out = torch.empty((B, hw, hw, Co), dtype=dt, device=DEV)
    wp = ops.pack_conv_weight(w, dt).to(DEV)
    ls = [ops.conv2d(x[i:i + 1], wp, out[i:i + 1], b.to(DEV)) for i in range(B)]
    bm, bn, sk = ops.gemm_plan(ls[0])
    assert sk > 1 and bm == 32 and hw * hw < 128, (bm, bn, sk)          # fragment reduce pass (32-row stripes) under a 128-row tile
    fused = ops.fuse_groupnorm_stats(out, [(ls[i], i * hw * hw, hw * hw, 0, Co) for i in range(B)])
    assert fused is not None
    ls[1]()
    ls[0]()
    ops.run(fused[2])
    g, be = rnd((Co,), 153) * 0.2 + 1, rnd((Co,), 154) * 0.2
    y = torch.empty_like(out)
    ops.groupnorm_apply(out, g.to(DEV), be.to(DEV), y, fused[0], fused[1], eps=1e-5, silu=True)()
    torch.cuda.synchronize()
    ref = F.conv2d(xr.to(DEV).permute(0, 3, 1, 2), w.to(dt).float().to(DEV), b.to(DEV), padding=1).permute(0, 2, 3, 1).cpu()
    check(out, ref, dt)
    refn = F.silu(F.group_norm(out.float().cpu().permute(0, 3, 1, 2), 32, g, be, 1e-5)).permute(0, 2, 3, 1)
    check(y, refn, dt)


# ------------------------------------------------------------------------------------------------ fp8 weight path (BASELINE configs[4])
def _fp8_ref(w):
    """CPU reference of rf_quantize_fp8_rows: smallest power-of-two scale with amax / scale <= 448, RNE to OCP e4m3fn."""
    amax = w.abs().amax(dim=1)
    scale = torch.where(amax > 0, torch.exp2(torch.ceil(torch.log2(amax / 448.0))), torch.ones_like(amax))
    q = (w / scale[:, None]).to(torch.float8_e4m3fn)
    return q, scale


@pytest.mark.parametrize("N,K", [(64, 320), (96, 128), (33, 2880), (5, 64)])
def test_quantize_fp8_rows(N, K):
    w = rnd((N, K), 70) * torch.logspace(-3, 1, N)[:, None]          # rows of very different magnitude
    w[0, :5] = 0.0
    fw = ops.quantize_fp8(w.to(DEV))
    torch.cuda.synchronize()
    q_ref, s_ref = _fp8_ref(w)
    assert fw.q.shape == (N, (K + 127) // 128 * 128) and fw.K == K
    assert torch.equal(fw.scale.cpu(), s_ref)
    assert torch.equal(fw.q[:, :K].cpu().view(torch.float8_e4m3fn).float(), q_ref.float())          # byte-exact up to -0 / +0
    assert (fw.q[:, K:] == 0).all()
    deq = fw.dequant().cpu()
    assert (deq - w).abs().max() <= (w.abs().amax(dim=1) * 2.0 ** -3).max()                         # 3 mantissa bits, scale <= 2x amax/448


@pytest.mark.parametrize("k", [-9, 0, 5])
def test_quantize_fp8_rows_edge_rows_vs_numpy_e4m3(k):
    """rf_quantize_fp8_rows against an INDEPENDENT numpy statement of OCP e4m3fn round-to-nearest-even with saturation (the
    quantiser had only been compared with torch's float8 cast and bounded end to end).  Rows are built so that the row scale is exactly 2^k:
      row 0: every finite e4m3 value times 2^k (amax = 448 * 2^k: the frexp f == 0.5 branch; must reproduce its own code),
      row 1: every midpoint between neighbouring representable values (ties -> even mantissa), both signs, subnormal range included,
      row 2: the midpoints moved one float32 ulp up / down (must go to the upper / lower neighbour),
      row 3: values below half the smallest subnormal, exact quarter / half / three-quarter subnormal steps, +-0,
      row 4: amax one float32 ulp ABOVE 448 * 2^k -> the scale doubles, 448 * 2^k itself becomes 224 (no saturation inside a row by construction),
      row 5: a row of zeros (scale 1, all bytes 0)."""
    import numpy as np
    sc = 2.0 ** k
    codes = np.array([c for c in range(256) if (c & 0x7f) != 0x7f], dtype=np.int64)
    vals = _e4m3fn_decode_np(codes)
    pos = np.sort(np.unique(np.abs(vals)))
    mid = (pos[:-1] + pos[1:]) / 2
    K = 384
    rows = np.zeros((6, K), dtype=np.float32)

    def put(r, v):
        v = np.asarray(v, dtype=np.float64) * sc
        assert len(v) <= K - 1 and np.array_equal(v.astype(np.float32).astype(np.float64), v), "test values must be float32-exact"
        rows[r, :len(v)] = v.astype(np.float32)
        rows[r, K - 1] = np.float32(448.0 * sc)               # pins the row's amax (scale 2^k)
    put(0, vals)
    put(1, np.concatenate([mid, -mid]))
    m32 = (mid * sc).astype(np.float32)
    up, dn = np.nextafter(m32, np.float32(np.inf)), np.nextafter(m32, np.float32(0))
    rows[2, :len(up)] = up
    rows[2, len(up):2 * len(up)] = -dn
    rows[2, K - 1] = np.float32(448.0 * sc)
    put(3, [2.0 ** -11, 2.0 ** -10, 3 * 2.0 ** -11, 2.0 ** -10 * (1 - 2.0 ** -24), 2.0 ** -9, 1.5 * 2.0 ** -9, 2.5 * 2.0 ** -9, 7.5 * 2.0 ** -9, 0.0, -0.0, -2.0 ** -10, -3 * 2.0 ** -11, 2.0 ** -20])
    rows[4, 0] = np.nextafter(np.float32(448.0 * sc), np.float32(np.inf))
    rows[4, 1] = np.float32(448.0 * sc)
    rows[4, 2] = np.float32(-447.0 * sc)
    rows[4, 3:3 + len(vals)] = (vals * sc).astype(np.float32)
    w = torch.from_numpy(rows)
    fw = ops.quantize_fp8(w.to(DEV))
    torch.cuda.synchronize()
    got_q, got_s = fw.q[:, :K].cpu().numpy(), fw.scale.cpu().numpy()
    want_s = np.array([sc, sc, sc, sc, 2 * sc, 1.0], dtype=np.float32)
    assert np.array_equal(got_s, want_s), (got_s, want_s)
    want_q = _e4m3fn_encode_rne_sat_np(rows.astype(np.float64) / want_s[:, None].astype(np.float64))
    # (-0 / +0: the hardware conversion keeps the sign of a value that rounds to zero, so does the reference; compare bytes)
    bad = np.argwhere(got_q != want_q)
    assert len(bad) == 0, [(int(r), int(c), float(rows[r, c]), hex(int(got_q[r, c])), hex(int(want_q[r, c]))) for r, c in bad[:8]]
    assert (got_q[5] == 0).all() and (fw.q[:, K:] == 0).all()
    # row 0 reproduces its own codes (sign of zero aside)
    assert np.array_equal(got_q[0, :len(codes)] & 0x7f, (codes & 0x7f).astype(np.uint8))


@pytest.mark.parametrize("M,N,K,kind", [(256, 320, 320, "linear"), (4096, 640, 1280, "res"), (300, 160, 192, "linear"), (512, 1280, 11520, "splitk"),
                                         (2048, 512, 640, "geglu"), (16384, 320, 320, "big")])
def test_linear_fp8_weights(M, N, K, kind):
    """fp8 (e4m3fn) weights x bf16 activations: the kernel dequantises to exactly the bf16 values w = fp8 * 2^e, so it must agree with
    an fp32 reference on the dequantised weights to bf16-output rounding -- and bit for bit with the bf16 kernel fed those weights."""
    dt = torch.bfloat16
    x, xr = q(rnd((M, K), 71) * 0.5, dt)
    w = rnd((N, K), 72) / math.sqrt(K)
    b = rnd((N,), 73)
    if kind == "geglu":
        wp, bp = ops.pack_geglu(w, b, torch.float32)
        fw = ops.quantize_fp8(wp.to(DEV))
        out = torch.empty((M, N // 2), dtype=dt, device=DEV)
        ops.linear(x, fw, out, bp.to(DEV), act=ops.ACT_GEGLU)()
        out_b = torch.empty_like(out)
        ops.linear(x, fw.dequant().to(dt), out_b, bp.to(DEV), act=ops.ACT_GEGLU)()
        torch.cuda.synchronize()
        wd = fw.dequant().cpu()
        # undo the 32-row interleave of pack_geglu on the dequantised rows
        f = N // 2
        wv = wd.reshape(f // 32, 2, 32, K)[:, 0].reshape(f, K)
        wg = wd.reshape(f // 32, 2, 32, K)[:, 1].reshape(f, K)
        ref = (F.linear(xr, wv, b[:f]) * F.gelu(F.linear(xr, wg, b[f:])))
    else:
        fw = ops.quantize_fp8(w.to(DEV))
        res, rr = q(rnd((M, N), 74), dt) if kind == "res" else (None, 0.0)
        out = torch.empty((M, N), dtype=dt, device=DEV)
        l = ops.linear(x, fw, out, b.to(DEV), residual=res)
        l()
        out_b = torch.empty_like(out)
        ops.linear(x, fw.dequant().to(dt), out_b, b.to(DEV), residual=res)()
        torch.cuda.synchronize()
        if kind == "splitk":
            assert ops.gemm_plan(l)[2] > 1
        ref = F.linear(xr, fw.dequant().cpu(), b) + rr
    check(out, ref, dt)
    assert torch.equal(out, out_b), (out.float() - out_b.float()).abs().max().item()


def test_conv_fp8_weights():
    dt = torch.bfloat16
    B, H, W_, Ci, Co = 2, 16, 16, 128, 320
    x, xr = q(rnd((B, H, W_, Ci), 75), dt)
    w = rnd((Co, Ci, 3, 3), 76) / math.sqrt(9 * Ci)
    b = rnd((Co,), 77)
    fw = ops.quantize_fp8(ops.pack_conv_weight(w, torch.float32).to(DEV))
    out = torch.empty((B, H, W_, Co), dtype=dt, device=DEV)
    ops.conv2d(x, fw, out, b.to(DEV))()
    torch.cuda.synchronize()
    wd = fw.dequant().cpu().reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2).contiguous()
    ref = _conv_ref(xr, wd, b, 1, (1, 1, 1, 1), 0)
    check(out, ref, dt)


@pytest.mark.parametrize("M", [128, 300, 4096, 65536])          # 65536 = the benchmark's launch (512 blocks: two rounds of 256 CUs)
@pytest.mark.parametrize("dt", H16)
def test_ffn_geglu_fused(M, dt):
    """rf_ffn_geglu (C = 320): GEGLU projection + ff.net.2 + residual in one kernel, against an fp32 reference on the bf16-rounded operands
    (with the hidden activations rounded to bf16, as both the fused and the unfused path store / feed them) and against the two unfused
    rf_conv_gemm launches."""
    Cc = 320
    x, xr = q(rnd((M, Cc), 80) * 0.7, dt)
    w1 = rnd((8 * Cc, Cc), 81) / math.sqrt(Cc)
    b1 = rnd((8 * Cc,), 82) * 0.5
    w2 = rnd((Cc, 4 * Cc), 83) / math.sqrt(4 * Cc)
    b2 = rnd((Cc,), 84)
    res, rr = q(rnd((M, Cc), 85), dt)
    w1p, b1p = ops.pack_geglu(w1, b1, dt)
    out = torch.empty((M, Cc), dtype=dt, device=DEV)
    ops.ffn_geglu(x, w1p.to(DEV), b1p.to(DEV), ops.pack_ffn_w2(w2.to(DEV), dt), b2.to(DEV), out, residual=res)()
    # unfused: the same two GEMMs through rf_conv_gemm
    hid = torch.empty((M, 4 * Cc), dtype=dt, device=DEV)
    out_u = torch.empty_like(out)
    ops.linear(x, w1p.to(DEV), hid, b1p.to(DEV), act=ops.ACT_GEGLU)()
    ops.linear(hid, w2.to(dt).to(DEV), out_u, b2.to(DEV), residual=res)()
    torch.cuda.synchronize()
    h = F.linear(xr, w1.to(dt).float(), b1)
    a, g = h.chunk(2, -1)
    hid_ref = (a * F.gelu(g, approximate="tanh")).to(dt).float()
    ref = F.linear(hid_ref, w2.to(dt).float(), b2) + rr
    check(out, ref, dt)
    # ... and against the REFERENCE's GELU (attention.py:42-44: F.gelu, the erf form): the kernel's tanh form is an approximation of it, so the
    # bf16 tolerance must hold against the erf reference too (the two references differ by <= 4.8e-4 |value| per hidden element)
    ref_erf = F.linear((a * F.gelu(g)).to(dt).float(), w2.to(dt).float(), b2) + rr
    check(out, ref_erf, dt)
    assert (out.float() - out_u.float()).abs().max().item() <= 4 * STEP[dt] * max(1.0, ref.abs().max().item())       # <= 2 ulps apart


@pytest.mark.parametrize("M", [300, 4096, 65536])          # 65536 = the benchmark's launch: norm3 inside the kernel at all five C = 320 blocks
@pytest.mark.parametrize("dt", H16)
def test_ffn_geglu_fused_with_layernorm(M, dt):
    """rf_ffn_geglu with ln_eps > 0: `norm3` (attention.py:231-233, 243) runs inside the kernel -- two-pass fp32 statistics of the token's row in
    registers, the normalised values rounded to bf16, gamma / beta folded into W1 / b1 by ops.fold_layernorm_geglu.  Against an fp32
    reference with the same roundings, and against the unfused chain rf_layernorm -> rf_ffn_geglu (a few bf16 ulps: the affine is
    applied before resp. after the rounding of the normalised value)."""
    Cc = 320
    x, xr = q(rnd((M, Cc), 180) * 1.7 + 0.4, dt)                     # rows with a mean: the statistics matter
    gamma, beta = rnd((Cc,), 181) * 0.3 + 1.0, rnd((Cc,), 182) * 0.2
    w1 = rnd((8 * Cc, Cc), 81) / math.sqrt(Cc)
    b1 = rnd((8 * Cc,), 82) * 0.5
    w2 = rnd((Cc, 4 * Cc), 83) / math.sqrt(4 * Cc)
    b2 = rnd((Cc,), 84)
    w1f, b1f = ops.fold_layernorm_geglu(w1, b1, gamma, beta)
    w1p, b1p = ops.pack_geglu(w1f, b1f, dt)
    out = torch.empty((M, Cc), dtype=dt, device=DEV)
    ops.ffn_geglu(x, w1p.to(DEV), b1p.to(DEV), ops.pack_ffn_w2(w2.to(DEV), dt), b2.to(DEV), out, residual=x, ln_eps=1e-5)()
    # unfused chain: the LayerNorm pass (affine applied, bf16 output), then the fused feed-forward on it
    ln = torch.empty((M, Cc), dtype=dt, device=DEV)
    ops.layernorm(x, gamma.to(DEV), beta.to(DEV), ln)()
    w1u, b1u = ops.pack_geglu(w1, b1, dt)
    out_u = torch.empty_like(out)
    ops.ffn_geglu(ln, w1u.to(DEV), b1u.to(DEV), ops.pack_ffn_w2(w2.to(DEV), dt), b2.to(DEV), out_u, residual=x)()
    torch.cuda.synchronize()
    xhat = F.layer_norm(xr, (Cc,), None, None, 1e-5).to(dt).float()
    hsum = F.linear(xhat, w1f.to(dt).float(), b1f)
    a, g = hsum.chunk(2, -1)
    hid = (a * F.gelu(g, approximate="tanh")).to(dt).float()
    ref = F.linear(hid, w2.to(dt).float(), b2) + xr
    check(out, ref, dt)
    check(out, F.linear((a * F.gelu(g)).to(dt).float(), w2.to(dt).float(), b2) + xr, dt)          # the reference's erf GELU (attention.py:42-44)
    d = (out.float() - out_u.float()).abs().max().item()
    print(f"ffn + in-kernel LayerNorm vs LayerNorm pass + ffn (M = {M}, {dt}): max |d| = {d:.3e} at |out| max {ref.abs().max().item():.2f}")
    assert d <= 8 * STEP[dt] * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,hw,pair,concat", [(2, 32, False, False), (16, 64, False, False), (4, 32, True, False), (3, 16, False, True), (16, 64, True, True)])
@pytest.mark.parametrize("dt", H16)
def test_ffn_block_fused_tail(B, hw, pair, concat, dt):
    """rf_ffn_block: the token-resident tail of a SpatialTransformer block at C = 320 -- norm3 + GEGLU feed-forward + residual (attention.py:40-76,
    231-233, 243) + proj_out + `x + x_in` (attention.py:268-272, 288-289) in ONE kernel, with the GroupNorm statistics of the block's output from its
    epilogue.  Against the unfused chain on the GPU (rf_ffn_geglu, then proj_out as rf_conv_gemm with the residual): the same bf16 roundings (the
    feed-forward's output row is rounded to bf16 where the chain stores it), so a few bf16 steps at most; against an fp32 reference with those
    roundings; statistics through rf_groupnorm_apply against F.group_norm of the stored output.  pair: the residual x_in has half the rows and is
    shared by both batch halves (the CFG-shared first block).  concat: the output is the right half of a [.., 640] concat buffer whose left half
    another GEMM produces -- the decoder's GroupNorm over [h | skip] takes its statistics from both producers."""
    Cc = 320
    HW = hw * hw
    nb = 2 if pair else 1
    M = nb * B * HW
    x1, x1r = q(rnd((M, Cc), 820) * 1.3 + 0.3, dt)                       # the post-attention residual stream (input of norm3 / the feed-forward)
    xin, xinr = q(rnd((B * HW, Cc), 821), dt)                           # the transformer's input x_in (B samples; shared by the halves when pair)
    gamma, beta = rnd((Cc,), 822) * 0.3 + 1.0, rnd((Cc,), 823) * 0.2
    w1 = rnd((8 * Cc, Cc), 824) / math.sqrt(Cc)
    b1 = rnd((8 * Cc,), 825) * 0.5
    w2 = rnd((Cc, 4 * Cc), 826) / math.sqrt(4 * Cc)
    b2 = rnd((Cc,), 827)
    wpo = rnd((Cc, Cc), 828) / math.sqrt(Cc)
    bpo = rnd((Cc,), 829)
    w1f, b1f = ops.fold_layernorm_geglu(w1, b1, gamma, beta)
    w1p, b1p = ops.pack_geglu(w1f, b1f, dt)
    w2q = ops.pack_ffn_w2(w2.to(DEV), dt)
    Ct = 2 * Cc if concat else Cc
    buf = torch.zeros((nb * B, hw, hw, Ct), dtype=dt, device=DEV)
    y = buf[..., Ct - Cc:]
    y2 = y.as_strided((M, Cc), (y.stride(2), 1))
    lf = ops.ffn_block(x1, w1p.to(DEV), b1p.to(DEV), w2q, b2.to(DEV), y2, residual=x1, wpo=wpo.to(dt).to(DEV), bpo=bpo.to(DEV), res2=xin,
                       res2_rows=B * HW if pair else 0, ln_eps=1e-5)
    prods = [(lf, 0, M, Ct - Cc, Cc)]
    if concat:
        a0, _ = q(rnd((M, 64), 830), dt)
        wl = rnd((Cc, 64), 831) / 8.0
        left = buf[..., :Cc]
        ll = ops.linear(a0, wl.to(dt).to(DEV), left.as_strided((M, Cc), (left.stride(2), 1)), rnd((Cc,), 832).to(DEV))
        prods.insert(0, (ll, 0, M, 0, Cc))
    fused = ops.fuse_groupnorm_stats(buf, prods)
    assert fused is not None
    if concat:
        ll()
    lf()
    ops.run(fused[2])
    g2, be2 = rnd((Ct,), 833) * 0.2 + 1, rnd((Ct,), 834) * 0.2
    yn = torch.empty_like(buf)
    ops.groupnorm_apply(buf, g2.to(DEV), be2.to(DEV), yn, fused[0], fused[1], eps=1e-5, silu=True)()
    # the unfused chain: fused feed-forward kernel (its own output tensor), then proj_out + residual as a GEMM
    x2 = torch.empty((M, Cc), dtype=dt, device=DEV)
    ops.ffn_geglu(x1, w1p.to(DEV), b1p.to(DEV), w2q, b2.to(DEV), x2, residual=x1, ln_eps=1e-5)()
    yu = torch.empty((M, Cc), dtype=dt, device=DEV)
    for hf in range(nb):
        ops.linear(x2[hf * B * HW:(hf + 1) * B * HW], wpo.to(dt).to(DEV), yu[hf * B * HW:(hf + 1) * B * HW], bpo.to(DEV), residual=xin)()
    torch.cuda.synchronize()
    # fp32 reference with the chain's roundings
    xhat = F.layer_norm(x1r, (Cc,), None, None, 1e-5).to(dt).float()
    a, g = F.linear(xhat, w1f.to(dt).float(), b1f).chunk(2, -1)
    hid = (a * F.gelu(g, approximate="tanh")).to(dt).float()
    x2r = (F.linear(hid, w2.to(dt).float(), b2) + x1r).to(dt).float()
    ref = F.linear(x2r, wpo.to(dt).float(), bpo) + xinr.repeat(nb, 1)
    got = y2.float().cpu()
    check(got, ref, dt)
    x2e = (F.linear((a * F.gelu(g)).to(dt).float(), w2.to(dt).float(), b2) + x1r).to(dt).float()          # the reference's erf GELU (attention.py:42-44)
    check(got, F.linear(x2e, wpo.to(dt).float(), bpo) + xinr.repeat(nb, 1), dt)
    d = (got - yu.float().cpu()).abs().max().item()
    print(f"fused tail vs unfused chain (B {B}, {hw}x{hw}, pair {pair}, concat {concat}, {dt}): max |d| = {d:.3e} at |out| max {ref.abs().max().item():.2f}")
    assert d <= 8 * STEP[dt] * max(1.0, ref.abs().max().item())
    refn = F.silu(F.group_norm(buf.float().cpu().permute(0, 3, 1, 2), 32, g2, be2, 1e-5)).permute(0, 2, 3, 1)
    check(yn, refn, dt)


@pytest.mark.parametrize("dt", H16)
@pytest.mark.parametrize("B,hw,pair", [(2, 32, False), (16, 64, False), (4, 32, True), (3, 16, False)])
def test_ffn_block_with_out_projection_in_front(B, hw, pair, dt):
    """rf_ffn_block with attn1.to_out in FRONT (rf_ffn_desc.wo; attention.py:239-243): x1 = att Wo^T + bo + ctx[sample] + tok inside the kernel, then the token-resident
    tail on it.  Against the chain it replaces on the GPU -- the out-projection as rf_conv_gemm (one launch per CFG half under `pair`: both halves read the same attention
    output and tok, each its own cross-attention vectors), then rf_ffn_block on its output: the same roundings of x1, so a few storage steps at most -- and the stored x1."""
    Cc = 320
    HW = hw * hw
    nb = 2 if pair else 1
    Mt, M = B * HW, nb * B * HW
    att, _ = q(rnd((Mt, Cc), 1300) * 0.8, dt)
    tok, _ = q(rnd((Mt, Cc), 1301), dt)
    xin, _ = q(rnd((Mt, Cc), 1302), dt)
    wo, bo = rnd((Cc, Cc), 1303) / math.sqrt(Cc), rnd((Cc,), 1304) * 0.3
    ctx = (rnd((nb * B, Cc), 1305) * 0.5).to(DEV)
    gamma, beta = rnd((Cc,), 1306) * 0.3 + 1.0, rnd((Cc,), 1307) * 0.2
    w1, b1 = rnd((8 * Cc, Cc), 1308) / math.sqrt(Cc), rnd((8 * Cc,), 1309) * 0.5
    w2, b2 = rnd((Cc, 4 * Cc), 1310) / math.sqrt(4 * Cc), rnd((Cc,), 1311)
    wpo, bpo = rnd((Cc, Cc), 1312) / math.sqrt(Cc), rnd((Cc,), 1313)
    w1f, b1f = ops.fold_layernorm_geglu(w1, b1, gamma, beta)
    w1p, b1p = ops.pack_geglu(w1f, b1f, dt)
    w2q = ops.pack_ffn_w2(w2.to(DEV), dt)
    wod = wo.to(dt).to(DEV)
    common = dict(wpo=wpo.to(dt).to(DEV), bpo=bpo.to(DEV), res2=xin, res2_rows=Mt if pair else 0, ln_eps=1e-5)
    # fused: out-projection in front
    x1f = torch.zeros((M, Cc), dtype=dt, device=DEV)
    yf = torch.empty((M, Cc), dtype=dt, device=DEV)
    ops.ffn_block(att, w1p.to(DEV), b1p.to(DEV), w2q, b2.to(DEV), yf, residual=x1f,
                  front=dict(wo=wod, bo=bo.to(DEV), ctx=ctx, rows_per_sample=HW, res0=tok, front_rows=Mt if pair else 0, x1=x1f), **common)()
    # the chain
    x1c = torch.empty((M, Cc), dtype=dt, device=DEV)
    for hf in range(nb):
        ops.linear(att, wod, x1c[hf * Mt:(hf + 1) * Mt], bo.to(DEV), residual=tok, rowvec=ctx[hf * B:(hf + 1) * B], rows_per_sample=HW)()
    yc = torch.empty((M, Cc), dtype=dt, device=DEV)
    ops.ffn_block(x1c, w1p.to(DEV), b1p.to(DEV), w2q, b2.to(DEV), yc, residual=x1c, **common)()
    torch.cuda.synchronize()
    d1 = (x1f.float() - x1c.float()).abs().max().item()
    dy = (yf.float() - yc.float()).abs().max().item()
    s1, sy = x1c.float().abs().max().item(), yc.float().abs().max().item()
    print(f"to_out in front (B {B}, {hw}x{hw}, pair {pair}, {dt}): x1 vs chain max |d| {d1:.2e} at {s1:.1f}, block output {dy:.2e} at {sy:.1f}")
    assert torch.isfinite(yf.float()).all()
    assert d1 <= 2 * STEP[dt] * max(1.0, s1)          # same products, another order of the fp32 additions: a rounding boundary now and then
    assert dy <= 16 * STEP[dt] * max(1.0, sy)


@pytest.mark.parametrize("M,K0,Cc,N,geglu,res", [(4096, 320, 320, 960, False, False), (65536, 320, 320, 960, False, True), (16384, 640, 640, 5120, True, True),
                                                  (4096, 1280, 1280, 3840, False, True), (1000, 320, 320, 640, True, False),
                                                  (4096, 1280, 1280, 10240, True, True)])          # (the consumer is split along N: 2.5 rounds of 256 x 256 tiles)
@pytest.mark.parametrize("dt", H16)
def test_layernorm_folded_around_gemms(M, K0, Cc, N, geglu, res, dt):
    """LayerNorm folded around two bf16 GEMMs (rf_conv_gemm_desc.ln_*; attention.py:231-243 norm1 -> to_q/k/v, norm3 -> ff.net.0): the producer's
    direct epilogue writes per-row (mean, M2) records per wave-tile stripe, the consumer multiplies the UN-normalised tensor by W diag(gamma)
    and applies rstd (acc - mean u) + (b + W beta) in its epilogue.  Against fp32 torch on the stored producer output, and against the
    unfolded chain on the GPU (rf_layernorm pass + plain GEMM): a few bf16 ulps (the normalised operand is never rounded in the folded form)."""
    x0, _ = q(rnd((M, K0), 190) * 0.8, dt)
    wp = rnd((Cc, K0), 191) / math.sqrt(K0)
    bp = rnd((Cc,), 192) * 0.5 + 0.3                        # rows with a mean
    r0, _ = q(rnd((M, Cc), 193), dt)
    gamma, beta = rnd((Cc,), 194) * 0.3 + 1.0, rnd((Cc,), 195) * 0.2
    w = rnd((N, Cc), 196) / math.sqrt(Cc)
    b = rnd((N,), 197) * 0.3
    y = torch.empty((M, Cc), dtype=dt, device=DEV)
    l_prod = ops.linear(x0, wp.to(dt).to(DEV), y, bp.to(DEV), residual=r0 if res else None, name="producer")
    act = ops.ACT_GEGLU if geglu else ops.ACT_NONE
    if geglu:
        wpk, bpk = ops.pack_geglu(w, b, torch.float32)
    else:
        wpk, bpk = w, b
    w2, u2, b2 = ops.fold_layernorm_linear(wpk, gamma, beta, bpk, dt)
    out = torch.empty((M, N // 2 if geglu else N), dtype=dt, device=DEV)
    cons = ops.linear(y, w2.to(DEV), out, b2.to(DEV), act=act, ln_u=u2.to(DEV), name="consumer")
    stats = ops.layernorm_fold([(l_prod, 0, M)], cons, eps=1e-5, C_=Cc)
    pl = ops.gemm_plan2(l_prod)
    if stats is None:
        # the fold is refused when either launch has no direct epilogue (small grids take the 4-wave 128x128 tile): the caller keeps the pass
        cons.keep[0].ln_u = None
        plc = ops.gemm_plan2(cons)
        assert not pl["direct"] or pl["splitk"] != 1 or Cc % pl["wave_cols"] or not plc["direct"] or plc["splitk"] != 1, (pl, plc)
        assert l_prod.keep[0].ln_stats_out is None and l_prod.keep[0].ln_out_parts == 0
        pytest.skip(f"these plans cannot carry the fold: producer {pl}, consumer {plc}")
    assert stats.shape == (M, Cc // pl["wave_cols"], 2)
    l_prod()
    cons()
    # unfolded chain on the GPU: LayerNorm pass + plain GEMM on its bf16 output
    ln = torch.empty_like(y)
    ops.layernorm(y, gamma.to(DEV), beta.to(DEV), ln)()
    out_u = torch.empty_like(out)
    ops.linear(ln, wpk.to(dt).to(DEV), out_u, bpk.to(DEV), act=act)()
    torch.cuda.synchronize()
    yf = y.float().cpu()
    # the records: mean / M2 of the stored row values per stripe
    wc = pl["wave_cols"]
    ys = yf.reshape(M, Cc // wc, wc)
    assert (stats[..., 0].cpu() - ys.mean(-1)).abs().max().item() < 1e-4
    m2 = ((ys - ys.mean(-1, keepdim=True)) ** 2).sum(-1)
    assert ((stats[..., 1].cpu() - m2).abs() / (m2 + 1.0)).max().item() < 1e-4
    h = F.linear(F.layer_norm(yf, (Cc,), gamma, beta, 1e-5), w, b)
    if geglu:
        a, g = h.chunk(2, -1)
        ref = a * F.gelu(g, approximate="tanh")
        check(out, a * F.gelu(g), dt)          # the reference's erf GELU (attention.py:42-44)
    else:
        ref = h
    check(out, ref, dt)
    dmax = (out.float() - out_u.float()).abs().max().item()
    print(f"LayerNorm fold M={M} C={Cc} N={N} geglu={geglu}: folded vs LayerNorm pass + GEMM max |d| = {dmax:.3e} (|out| max {ref.abs().max().item():.2f}), "
          f"producer tile {pl['bm']}x{pl['bn']} stripe {wc}")
    assert dmax <= 8 * STEP[dt] * max(1.0, ref.abs().max().item())


# ------------------------------------------------------------------------------------------------ split-bf16 (RF_BF16X3) operands
def _split_ref(x):
    """(hi, lo) bf16 pair of an fp32 tensor, as rf_split_bf16 / rf_groupnorm_apply(out_dtype = RF_BF16X3) define it."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


@pytest.mark.parametrize("B,hw,c", [(4, 32, 320), (2, 32, 640), (3, 16, 320)])
@pytest.mark.parametrize("dt", H16)
def test_groupnorm_folded_into_linear(B, hw, c, dt):
    """SpatialTransformer `norm` (GroupNorm 32, eps 1e-6, no SiLU) folded into proj_in (attention.py:262-266, 276-279): rf_groupnorm_fold_linear
    scales W's columns per sample from the GroupNorm statistics, rf_conv_gemm multiplies the UN-normalised tensor with per-sample weights
    (w_sample_stride) and adds the per-sample vector.  Against the fp32 reference Linear(GroupNorm(x)) on the bf16-rounded x, beside the unfused
    bf16 pair (normalise pass + GEMM) -- the fold must not be less accurate than what it replaces."""
    HW = hw * hw
    x, xr = q(rnd((B, hw, hw, c), 310) * 1.5 + rnd((B, 1, 1, c), 311) * 2.0, dt)          # channel offsets: means far from 0
    w = rnd((c, c), 312) / math.sqrt(c)
    bias = rnd((c,), 313)
    g, be = rnd((c,), 314) * 0.3 + 1, rnd((c,), 315) * 0.3
    ref = F.linear(F.group_norm(xr.permute(0, 3, 1, 2), 32, g, be, 1e-6).permute(0, 2, 3, 1), w, bias)
    part = torch.empty(B * ops.GN_MAX_CHUNKS * 64, dtype=torch.float64, device=DEV)
    ls, n = ops.groupnorm_stats(x, part)
    fl, wps, rv = ops.groupnorm_fold_linear(w.to(DEV), g.to(DEV), be.to(DEV), bias.to(DEV), part, n, B=B, HW=HW, eps=1e-6, dtype=dt)
    out = torch.empty((B * HW, c), dtype=dt, device=DEV)
    lg = ops.linear(x.view(B * HW, c), wps[0], out, None, rowvec=rv, rows_per_sample=HW, w_per_sample=wps)
    plan = ops.gemm_plan2(lg)
    if HW % plan["bm"]:
        with pytest.raises(ops._lib.RefaceHipError):          # a tile would straddle samples: refused, not mis-computed
            lg()
        return
    ls(); fl(); lg()
    # the unfused pair
    xn = torch.empty_like(x)
    out2 = torch.empty_like(out)
    a, b_ = ops.groupnorm(x, g.to(DEV), be.to(DEV), xn, part, eps=1e-6, silu=False)
    a(); b_()
    ops.linear(xn.view(B * HW, c), w.to(dt).to(DEV), out2, bias.to(DEV))()
    torch.cuda.synchronize()
    e_fold = ((out.float().cpu().view(ref.shape) - ref).norm() / ref.norm()).item()
    e_pair = ((out2.float().cpu().view(ref.shape) - ref).norm() / ref.norm()).item()
    print(f"GroupNorm folded into Linear (B {B}, {hw}x{hw}, C {c}): rel L2 {e_fold:.2e} (normalise pass + GEMM: {e_pair:.2e})")
    assert e_fold < 8e-3 * (STEP[dt] / STEP[torch.bfloat16]) ** 0.5 and e_fold < 1.5 * e_pair + 1e-3, (e_fold, e_pair)
    # per-sample weights really are per sample
    assert not torch.equal(wps[0], wps[1])


@pytest.mark.parametrize("dt", H16)
@pytest.mark.parametrize("B,hw", [(2, 16), (3, 32), (8, 64)])
def test_attn_in_fused_front(B, hw, dt):
    """rf_attn_in: the token-resident front of a SpatialTransformer block at C = 320 -- GroupNorm `norm` (folded into per-sample proj_in weights) + proj_in + norm1 +
    to_q / to_k / to_v (attention.py:262-266, 276-279, 231-233, 239, 159-170) in ONE kernel.  Against the chain it replaces on the GPU (rf_groupnorm_fold_linear ->
    rf_conv_gemm with per-sample weights -> rf_layernorm -> rf_conv_gemm: the same roundings of tok and of the normalised row), and against an fp32 reference
    Linear(LayerNorm(Linear(GroupNorm(x)))) on the rounded x."""
    c = 320
    HW, M = hw * hw, B * hw * hw
    x, xr = q(rnd((B, hw, hw, c), 1200) * 1.5 + rnd((B, 1, 1, c), 1201) * 0.5, dt)
    g0, b0 = rnd((c,), 1202) * 0.3 + 1, rnd((c,), 1203) * 0.2
    wpi, bpi = rnd((c, c), 1204) / math.sqrt(c), rnd((c,), 1205) * 0.3
    g1, b1 = rnd((c,), 1206) * 0.3 + 1, rnd((c,), 1207) * 0.2
    wqkv = rnd((3 * c, c), 1208) / math.sqrt(c)
    part = torch.empty(B * ops.GN_MAX_CHUNKS * 64, dtype=torch.float64, device=DEV)
    ls, n = ops.groupnorm_stats(x, part)
    fl, wps, rv = ops.groupnorm_fold_linear(wpi.to(DEV), g0.to(DEV), b0.to(DEV), bpi.to(DEV), part, n, B=B, HW=HW, eps=1e-6, dtype=dt)
    wqf, bqf = ops.fold_layernorm_geglu(wqkv, torch.zeros(3 * c), g1, b1)
    tok = torch.empty((M, c), dtype=dt, device=DEV)
    qkv = torch.full((M, 3 * c), 9.0, dtype=dt, device=DEV)
    la = ops.attn_in(x.view(M, c), wps, rv, tok, wqf.to(dt).to(DEV), bqf.to(DEV), qkv, rows_per_sample=HW, ln_eps=1e-5)
    ls(); fl(); la()
    # the chain it replaces
    tok2 = torch.empty_like(tok)
    ops.linear(x.view(M, c), wps[0], tok2, None, rowvec=rv, rows_per_sample=HW, w_per_sample=wps)()
    lnb = torch.empty_like(tok)
    ops.layernorm(tok2, g1.to(DEV), b1.to(DEV), lnb, eps=1e-5)()
    qkv2 = torch.empty_like(qkv)
    ops.linear(lnb, wqkv.to(dt).to(DEV), qkv2, None)()
    torch.cuda.synchronize()
    # fp32 reference on the rounded x
    gn = F.group_norm(xr.permute(0, 3, 1, 2), 32, g0, b0, 1e-6).permute(0, 2, 3, 1).reshape(M, c)
    tok_ref = F.linear(gn, wpi, bpi)
    qkv_ref = F.linear(F.layer_norm(tok_ref, (c,), g1, b1, 1e-5), wqkv)
    d_tok = (tok.float() - tok2.float()).abs().max().item()
    d_qkv = (qkv.float() - qkv2.float()).abs().max().item()
    e_tok = ((tok.float().cpu() - tok_ref).norm() / tok_ref.norm()).item()
    e_qkv = ((qkv.float().cpu() - qkv_ref).norm() / qkv_ref.norm()).item()
    e_qkv2 = ((qkv2.float().cpu() - qkv_ref).norm() / qkv_ref.norm()).item()
    print(f"fused front (B {B}, {hw}x{hw}, {dt}): tok vs chain max |d| {d_tok:.2e}, qkv vs chain {d_qkv:.2e}; rel L2 vs fp32 reference tok {e_tok:.2e}, qkv {e_qkv:.2e} (chain {e_qkv2:.2e})")
    assert torch.isfinite(qkv.float()).all()
    assert d_tok <= 4 * STEP[dt] * max(1.0, tok_ref.abs().max().item())          # same products, another order of the fp32 additions
    assert d_qkv <= 16 * STEP[dt] * max(1.0, qkv_ref.abs().max().item())         # (a tok value on a rounding boundary moves one normalised operand by a step)
    lim = 6e-3 * (STEP[dt] / STEP[torch.bfloat16]) ** 0.5
    assert e_tok < lim and e_qkv < 1.5 * e_qkv2 + 1e-3 and e_qkv < 2 * lim, (e_tok, e_qkv, e_qkv2)


@pytest.mark.parametrize("B,H,W_,c,No,odt", [(2, 64, 64, 320, 4, torch.float32), (3, 24, 40, 320, 4, torch.float32), (2, 16, 16, 64, 4, torch.float32),
                                             (1, 9, 7, 128, 3, None), (2, 96, 96, 320, 4, torch.float32)])          # (odt None: the input's 16-bit type)
@pytest.mark.parametrize("dt", H16)
def test_gn_silu_conv3x3_small_fused(B, H, W_, c, No, odt, dt):
    """rf_gn_silu_conv3x3_small, the UNet's `out` head (openaimodel.py:737-741: GroupNorm32 -> SiLU -> 3x3 conv to 4 channels) in one pass over
    the raw tensor: against the fp32 reference conv(SiLU(GroupNorm(x))) with the activations rounded to bf16 where the kernel rounds them (the
    operand of the matrix pipe = what the normalisation pass stores), and against the unfused pair rf_groupnorm_apply + rf_conv_gemm (same
    products, another order of fp32 additions).  Odd image sizes: the last 128-pixel block of a sample is ragged, the border taps are skipped."""
    odt = odt or dt
    x, xr = q(rnd((B, H, W_, c), 700) * 1.3 + rnd((B, 1, 1, c), 701) * 0.8, dt)
    g, be = rnd((c,), 702) * 0.3 + 1, rnd((c,), 703) * 0.3
    w = rnd((No, c, 3, 3), 704) / math.sqrt(9 * c)
    bias = rnd((No,), 705)
    part = torch.empty(B * ops.GN_MAX_CHUNKS * 64, dtype=torch.float64, device=DEV)
    ls, n = ops.groupnorm_stats(x, part)
    wp = ops.pack_conv_weight(w, dt).to(DEV)
    out = torch.full((B, H, W_, No), 7.0, dtype=odt, device=DEV)
    lf = ops.gn_silu_conv3x3_small(x, g.to(DEV), be.to(DEV), part, n, wp, bias.to(DEV), out, eps=1e-5, silu=True)
    ls(); lf()
    # the unfused pair on the same statistics
    xn = torch.empty_like(x)
    out2 = torch.empty_like(out)
    ops.groupnorm_apply(x, g.to(DEV), be.to(DEV), xn, part, n, eps=1e-5, silu=True)()
    ops.conv2d(xn, wp, out2, bias.to(DEV))()
    torch.cuda.synchronize()
    act = F.silu(F.group_norm(xr.permute(0, 3, 1, 2), 32, g, be, 1e-5)).to(dt).float()          # rounded where the kernel rounds
    ref = F.conv2d(act, w.to(dt).float(), bias, padding=1).permute(0, 2, 3, 1)
    e_f = (out.float().cpu() - ref).abs().max().item()
    e_p = (out2.float().cpu() - ref).abs().max().item()
    e_fp = (out.float() - out2.float()).abs().max().item()
    print(f"fused out head (B {B}, {H}x{W_}, C {c} -> {No}): max |d| vs fp32 reference {e_f:.2e} (unfused pair {e_p:.2e}), fused vs pair {e_fp:.2e}")
    # the bf16 rounding of an activation near a rounding boundary may differ by one step between this kernel, the apply pass and torch (scale / shift
    # association): a handful of +-1 ulp operands out of 9 c per output -- bounded well below the output's own bf16 step
    tol = (4e-3 if odt == torch.float32 else 2e-2) * STEP[dt] / STEP[torch.bfloat16]
    assert e_f < tol and e_fp < tol, (e_f, e_p, e_fp)


@pytest.mark.parametrize("B,H,W_,c,dup", [(2, 64, 64, 320, True), (1, 96, 96, 320, False), (2, 16, 16, 64, True), (3, 16, 16, 128, False)])
@pytest.mark.parametrize("dt", H16)
def test_conv3x3_stem_pixels_on_lanes(B, H, W_, c, dup, dt):
    """rf_conv3x3_stem, the UNet's stem (openaimodel.py:666-671: 3x3 conv, pad 1, 9 input channels stored in 16 -> model_channels): against
    F.conv2d of the bf16-rounded operands in fp32 (output rounded to bf16: one bf16 step), BIT-compatible rows in the duplicate half, against the
    implicit GEMM it replaces, and its GroupNorm statistics for three consumers -- the un-duplicated half alone (the first ResBlock under cfg_pair)
    and both halves as the right column range of a [.., 2c] concat buffer whose left half another GEMM produces (the last decoder block) --
    through rf_groupnorm_apply against F.group_norm of the stored tensor."""
    nb = 2 if dup else 1
    x, xr = q(rnd((B, H, W_, 16), 910), dt)
    x[..., 9:] = 0
    xr[..., 9:] = 0
    w = rnd((c, 9, 3, 3), 911) / math.sqrt(81)
    bias = rnd((c,), 912)
    wp = ops.pack_conv_weight(w, dt, cin_pad=16).to(DEV)
    cat = torch.zeros((nb * B, H, W_, 2 * c), dtype=dt, device=DEV)
    y = cat[..., c:]
    l = ops.conv3x3_stem(x, wp, bias.to(DEV), y[:B], dup=y[B:] if dup else None)
    M = nb * B * H * W_
    a0, _ = q(rnd((M, 64), 913), dt)
    left = cat[..., :c]
    ll = ops.linear(a0, (rnd((c, 64), 914) / 8.0).to(dt).to(DEV), left.as_strided((M, c), (left.stride(2), 1)), rnd((c,), 915).to(DEV))
    # consumer 1: the first half alone; consumer 2: the concat of both halves
    f1 = ops.fuse_groupnorm_stats(y[:B], [(l, 0, B * H * W_, 0, c)])
    prods = [(ll, 0, M, 0, c), (l, 0, B * H * W_, c, c)] + ([(l, B * H * W_, B * H * W_, c, c)] if dup else [])
    f2 = ops.fuse_groupnorm_stats(cat, prods)
    assert f1 is not None and f2 is not None
    if dup:          # all three consumer slots of the launch are taken
        assert ops.fuse_groupnorm_stats(y[:B], [(l, 0, B * H * W_, 0, c)]) is None
    ll(); l()
    ops.run(f1[2]); ops.run(f2[2])
    g1, b1 = rnd((c,), 916) * 0.2 + 1, rnd((c,), 917) * 0.2
    g2, b2 = rnd((2 * c,), 918) * 0.2 + 1, rnd((2 * c,), 919) * 0.2
    n1 = torch.empty((B, H, W_, c), dtype=dt, device=DEV)
    n2 = torch.empty_like(cat)
    ops.groupnorm_apply(y[:B], g1.to(DEV), b1.to(DEV), n1, f1[0], f1[1], eps=1e-5, silu=True)()
    ops.groupnorm_apply(cat, g2.to(DEV), b2.to(DEV), n2, f2[0], f2[1], eps=1e-5, silu=False)()
    y2 = torch.empty((B, H, W_, c), dtype=dt, device=DEV)
    ops.conv2d(x, wp, y2, bias.to(DEV))()
    torch.cuda.synchronize()
    ref = F.conv2d(xr[..., :9].permute(0, 3, 1, 2), w.to(dt).float(), bias, padding=1).permute(0, 2, 3, 1)
    got = y[:B].float().cpu()
    check(got, ref, dt)
    d = (got - y2.float().cpu()).abs().max().item()
    print(f"stem (B {B}, {H}x{W_}, 9 -> {c}, dup {dup}): max |d| vs fp32 reference {(got - ref).abs().max().item():.2e}, vs the implicit GEMM {d:.2e}")
    assert d <= 4 * STEP[dt] * max(1.0, ref.abs().max().item())
    if dup:
        assert torch.equal(y[:B], y[B:])
    check(n1, F.silu(F.group_norm(got.permute(0, 3, 1, 2), 32, g1, b1, 1e-5)).permute(0, 2, 3, 1), dt)
    check(n2, F.group_norm(cat.float().cpu().permute(0, 3, 1, 2), 32, g2, b2, 1e-5).permute(0, 2, 3, 1), dt)


def test_split_bf16_kernel_bit_exact():
    x = rnd((3, 5, 7, 64), 900) * 3.0
    out = torch.zeros((3, 5, 7, 128), dtype=torch.bfloat16, device=DEV)
    ops.split_bf16(x.to(DEV), out)()
    torch.cuda.synchronize()
    hi, lo = _split_ref(x)
    assert torch.equal(out[..., :64].cpu(), hi) and torch.equal(out[..., 64:].cpu(), lo)
    # hi + lo carries 16 significant bits
    rec = out[..., :64].float().cpu() + out[..., 64:].float().cpu()
    assert ((rec - x).abs() <= 2.0 ** -16 * x.abs() + 1e-30).all()


@pytest.mark.parametrize("M,N,K", [(300, 192, 128), (4096, 128, 1152), (1024, 512, 4608)])
def test_linear_x3_vs_fp64(M, N, K):
    """out = x W^T on split-bf16 operands: hi*hi + hi*lo + lo*hi in fp32 -- 2^-16-class relative error per product, two orders of
    magnitude under the plain bf16 kernel, and a transposition-detecting reference (fp64 of the original fp32 operands)."""
    x = rnd((M, K), 901)
    w = rnd((N, K), 902) / math.sqrt(K)
    b = rnd((N,), 903)
    res = rnd((M, N), 904)
    xs = torch.empty((M, 2 * K), dtype=torch.bfloat16, device=DEV)
    ops.split_bf16(x.to(DEV), xs)()
    out = torch.empty((M, N), dtype=torch.float32, device=DEV)
    l = ops.conv_gemm(xs, ops.pack_x3(w).to(DEV), out, M=M, N=N, K=K, C0=K, ld0=2 * K, Hin=1, Win=M, Hout=1, Wout=M, bias=b.to(DEV),
                      residual=res.to(DEV), ldr=N, ldo=N, x3=True)
    l()
    torch.cuda.synchronize()
    ref = (x.double() @ w.double().T + b.double() + res.double())
    err = (out.cpu().double() - ref).abs().max().item()
    # sum over K of |x w| ~ 0.8 sqrt(K) * ... : bound the error by 2^-15 * sum|x||w| (three dropped / rounded terms) + fp32 accumulation
    bound = (x.abs().double() @ w.abs().double().T).max().item() * 2.0 ** -15 + 1e-5
    assert err < bound, (err, bound)
    assert err < 2e-4, err
    # the plain bf16 kernel on the same operands is >= 20x further away: the lo passes really are multiplied in
    outb = torch.empty((M, N), dtype=torch.float32, device=DEV)
    ops.linear(x.to(torch.bfloat16).to(DEV), w.to(torch.bfloat16).to(DEV), outb, b.to(DEV), residual=res.to(DEV))()
    torch.cuda.synchronize()
    errb = (outb.cpu().double() - ref).abs().max().item()
    assert errb > 20 * err, (errb, err)


@pytest.mark.parametrize("case", ["s1", "s2asym", "ups", "k1", "big"])
def test_conv_x3_vs_fp64(case):
    B, H, W_, Ci, Co = 2, 12, 10, 64, 96
    stride, pad4, ups, ks = 1, (1, 1, 1, 1), 0, 3
    if case == "s2asym":
        stride, pad4 = 2, (0, 1, 0, 1)
    elif case == "ups":
        ups = 1
    elif case == "k1":
        ks, pad4 = 1, (0, 0, 0, 0)
    elif case == "big":                       # 8-wave tiles + split-K-free long K
        B, H, W_, Ci, Co = 2, 64, 64, 128, 256
    x = rnd((B, H, W_, Ci), 905)
    w = rnd((Co, Ci, ks, ks), 906) / math.sqrt(Ci * ks * ks)
    b = rnd((Co,), 907)
    ref = _conv_ref(x.double(), w.double(), b.double(), stride, pad4, ups)
    xs = torch.empty((B, H, W_, 2 * Ci), dtype=torch.bfloat16, device=DEV)
    ops.split_bf16(x.to(DEV), xs)()
    out = torch.empty(ref.shape, dtype=torch.float32, device=DEV)
    ops.conv2d(xs, ops.pack_x3(ops.pack_conv_weight(w, torch.float32)).to(DEV), out, b.to(DEV), ksize=ks, stride=stride, pad=(pad4[2], pad4[0]),
               ups=ups, x3=True)()
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs().max().item()
    assert torch.isfinite(out).all() and err < 1e-4, err


def test_groupnorm_apply_split_output():
    """GroupNorm + SiLU of an fp32 tensor written as split-bf16 pairs == rf_split_bf16 of the fp32 result of the same kernel."""
    B, H, W_, Cc = 2, 16, 16, 128
    x = rnd((B, H, W_, Cc), 908).to(DEV)
    g, bt = rnd((Cc,), 909).to(DEV), rnd((Cc,), 910).to(DEV)
    part = torch.empty(B * ops.GN_MAX_CHUNKS * 64, dtype=torch.float64, device=DEV)
    y32 = torch.empty_like(x)
    ops.run(ops.groupnorm(x, g, bt, y32, part, eps=1e-6, silu=True))
    ys = torch.zeros((B, H, W_, 2 * Cc), dtype=torch.bfloat16, device=DEV)
    ops.run(ops.groupnorm(x, g, bt, ys, part, eps=1e-6, silu=True, split=True))
    torch.cuda.synchronize()
    hi, lo = _split_ref(y32.cpu())
    assert torch.equal(ys[..., :Cc].cpu(), hi) and torch.equal(ys[..., Cc:].cpu(), lo)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_geglu_negative_gates(dt):
    """The 16-bit GEGLU epilogues use the sigmoid-form GELU with a degree-5 argument (csrc/common.h gelu_sigmoid5: |form - gelu_erf| <= 2.6e-5
    absolute; the tanh form it replaced was 4.8e-4).  For negative gates gelu(g) is small, so the bound on the PRODUCT is absolute:
    |out - value * gelu_erf(gate)| <= 2.6e-5 * |value| + one ulp of the stored product.  Gates swept over [-8, 8] (beyond |7| the argument is clamped)."""
    M, Cc, Fh = 512, 64, 64
    x = torch.zeros((M, Cc))
    x[:, 0] = torch.linspace(-8.0, 8.0, M)          # channel 0 drives the gate, channel 1 the value
    x[:, 1] = rnd((M,), 920) * 2.0
    w = torch.zeros((2 * Fh, Cc))
    w[:Fh, 1] = 1.0                                  # value rows = x[:, 1]
    w[Fh:, 0] = 1.0                                  # gate rows  = x[:, 0]
    b = torch.zeros((2 * Fh,))
    xd, xr = q(x, dt)
    wp, bp = ops.pack_geglu(w, b, dt)
    out = torch.empty((M, Fh), dtype=dt, device=DEV)
    ops.linear(xd, wp.to(DEV), out, bp.to(DEV), act=ops.ACT_GEGLU)()
    torch.cuda.synchronize()
    val, gate = xr[:, 1:2].double(), xr[:, 0:1].double()
    ref = (val * 0.5 * gate * (1.0 + torch.erf(gate / math.sqrt(2.0)))).expand(M, Fh)
    err = (out.double().cpu() - ref).abs()
    ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    lim = 2.6e-5 * val.abs() + ulp * ref.abs() + 1e-6
    assert (err <= lim).all(), (err - lim).max().item()


# ------------------------------------------------------------------------------------------------ fp8 x fp8 (MX-scaled MFMA) path
def _quant_act_ref(x):
    """torch restatement of rf_quantize_fp8_act: per (row, 32-channel block) the smallest power-of-two scale with amax / scale <= 448,
    RNE conversion to e4m3fn.  Returns (q as float8 tensor, E8M0 codes)."""
    *lead, c = x.shape
    xb = x.float().reshape(*lead, c // 32, 32)
    amax = xb.abs().amax(-1)
    m, e = torch.frexp(amax)                                   # amax = m * 2^e, m in [0.5, 1)
    # amax = (2m) * 2^(e-1), 2m in [1, 2): code = (e - 1 + 127) - 8 + (2m > 1.75)
    code = (e - 1 + 127 - 8 + (2 * m > 1.75).int()).clamp(0, 253)
    code = torch.where(amax > 0, code, torch.zeros_like(code))
    scale = torch.pow(2.0, code.float() - 127.0)
    q = (xb / scale[..., None]).reshape(*lead, c).to(torch.float8_e4m3fn)
    return q, code.to(torch.uint8)


@pytest.mark.parametrize("M,Cc", [(300, 320), (64, 128), (1000, 960)])
def test_quantize_fp8_act(M, Cc):
    x = (rnd((M, Cc), 930) * torch.exp(rnd((M, 1), 931))).to(torch.bfloat16)          # rows of very different magnitude
    x[0, :32] = 0                                                                       # an all-zero block
    out = ops.Fp8Act((M, Cc), DEV)
    ops.quantize_act(x.to(DEV), out)()
    torch.cuda.synchronize()
    q, code = _quant_act_ref(x)
    assert torch.equal(out.scale[:, :Cc // 32].cpu(), code)
    assert torch.equal(out.q[:, :Cc].cpu(), q.view(torch.uint8))
    assert (out.q[:, Cc:] == 0).all() and (out.scale[:, Cc // 32:] == 127).all()       # pads untouched
    # round trip: 3 mantissa bits + block scaling
    deq = out.dequant().cpu()
    blockmax = x.float().reshape(M, Cc // 32, 32).abs().amax(-1).repeat_interleave(32, dim=1)
    assert ((deq - x.float()).abs() <= 2.0 ** -4 * x.float().abs() + 2.0 ** -9 * blockmax + 1e-30).all()


@pytest.mark.parametrize("M,N,K,kind", [(256, 320, 320, "plain"), (4096, 960, 320, "plain"), (2048, 640, 1280, "res"), (1024, 1280, 5120, "plain"),
                                        (512, 2560, 320, "geglu"), (77, 200, 128, "plain")])
def test_linear_fp8_act(M, N, K, kind):
    """fp8 activations (E8M0 block scales) x fp8 weights (per-row scales) on v_mfma_scale_f32_32x32x64_f8f6f4: products of fp8 values are
    exact in fp32 and the accumulation is fp32, so the result must match an fp32 reference on the DEQUANTISED operands to bf16 output
    rounding -- including K = 320 (padded to 384 against zero weights)."""
    dt = torch.bfloat16
    x = (rnd((M, K), 932) * 0.7).to(dt)
    xa = ops.Fp8Act((M, K), DEV)
    ops.quantize_act(x.to(DEV), xa)()
    w = rnd((N, K), 933) / math.sqrt(K)
    b = rnd((N,), 934)
    if kind == "geglu":
        wp, bp = ops.pack_geglu(w, b, torch.float32)
        fw = ops.quantize_fp8_padded(wp.to(DEV), 1, K)
        out = torch.empty((M, N // 2), dtype=dt, device=DEV)
        ops.linear(xa, fw, out, bp.to(DEV), act=ops.ACT_GEGLU)()
        torch.cuda.synchronize()
        wd = fw.dequant()[:, :K].cpu()
        f = N // 2
        wv, wg = wd.reshape(f // 32, 2, 32, K)[:, 0].reshape(f, K), wd.reshape(f // 32, 2, 32, K)[:, 1].reshape(f, K)
        xd = xa.dequant().cpu()
        ref = F.linear(xd, wv, b[:f]) * F.gelu(F.linear(xd, wg, b[f:]))
    else:
        fw = ops.quantize_fp8_padded(w.to(DEV), 1, K)
        res, rr = q(rnd((M, N), 935), dt) if kind == "res" else (None, 0.0)
        out = torch.empty((M, N), dtype=dt, device=DEV)
        l = ops.linear(xa, fw, out, b.to(DEV), residual=res)
        l()
        torch.cuda.synchronize()
        assert l.keep[0].dtype == 2 and l.keep[0].w_dtype == 2
        ref = F.linear(xa.dequant().cpu(), fw.dequant()[:, :K].cpu(), b) + rr
    check(out, ref, dt)


@pytest.mark.parametrize("case", ["c320", "c640s", "c128", "down", "up"])
def test_conv_fp8_act(case):
    dt = torch.bfloat16
    B, H, W_, Ci, Co = {"c320": (2, 16, 16, 320, 320), "c640s": (2, 8, 8, 640, 640), "c128": (1, 12, 10, 128, 96), "down": (2, 16, 16, 320, 320),
                        "up": (2, 8, 8, 640, 640)}[case]
    stride, ups = (2 if case == "down" else 1), (1 if case == "up" else 0)
    x = (rnd((B, H, W_, Ci), 936) * 0.8).to(dt)
    xa = ops.Fp8Act((B, H, W_, Ci), DEV)
    ops.quantize_act(x.to(DEV), xa)()
    w = rnd((Co, Ci, 3, 3), 937) / math.sqrt(9 * Ci)
    b = rnd((Co,), 938)
    rv = rnd((B, Co), 939)
    fw = ops.quantize_fp8_padded(ops.pack_conv_weight(w, torch.float32).to(DEV), 9, Ci)
    Ho, Wo = (2 * H, 2 * W_) if ups else (H // stride, W_ // stride)
    out = torch.empty((B, Ho, Wo, Co), dtype=dt, device=DEV)
    ops.conv2d(xa, fw, out, b.to(DEV), rowvec=rv.to(DEV), stride=stride, ups=ups)()
    torch.cuda.synchronize()
    Cp = xa.Cp
    wd = fw.dequant().cpu().reshape(Co, 9, Cp)[:, :, :Ci].reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2)
    ref = _conv_ref(xa.dequant().cpu(), wd, b, stride, (1, 1, 1, 1), ups) + rv[:, None, None, :]
    check(out, ref, dt)


def test_norms_fp8_output():
    """GroupNorm+SiLU and LayerNorm writing fp8 + block scales directly == rf_quantize_fp8_act semantics applied to the fp32 normalised values
    (the bf16-output kernels round to bf16 first: compare dequantised values with a 3-mantissa-bit + block-scale tolerance)."""
    B, H, W_, Cc = 2, 16, 16, 320
    x = rnd((B, H, W_, Cc), 940).to(torch.bfloat16).to(DEV)
    g, bt = rnd((Cc,), 941).to(DEV), rnd((Cc,), 942).to(DEV)
    part = torch.empty(B * ops.GN_MAX_CHUNKS * 64, dtype=torch.float64, device=DEV)
    y32 = torch.empty((B, H, W_, Cc), dtype=torch.float32, device=DEV)
    ops.run(ops.groupnorm(x, g, bt, y32, part, eps=1e-5, silu=True))
    ya = ops.Fp8Act((B, H, W_, Cc), DEV)
    ops.run(ops.groupnorm(x, g, bt, ya, part, eps=1e-5, silu=True))
    torch.cuda.synchronize()
    qr, code = _quant_act_ref(y32.cpu())
    assert (ya.scale[..., :Cc // 32].cpu().int() - code.int()).abs().max() <= 1          # (fp32 evaluation order may move an amax across a power of two)
    d = (ya.dequant().cpu() - y32.cpu()).abs()
    bm = y32.cpu().reshape(B, H, W_, Cc // 32, 32).abs().amax(-1).repeat_interleave(32, dim=-1)
    assert (d <= 2.0 ** -4 * y32.cpu().abs() + 2.0 ** -8 * bm + 1e-6).all(), d.max().item()
    M = 512
    xl = rnd((M, Cc), 943).to(torch.bfloat16).to(DEV)
    l32 = torch.empty((M, Cc), dtype=torch.float32, device=DEV)
    ops.layernorm(xl, g, bt, l32)()
    la = ops.Fp8Act((M, Cc), DEV)
    ops.layernorm(xl, g, bt, la)()
    torch.cuda.synchronize()
    d = (la.dequant().cpu() - l32.cpu()).abs()
    bm = l32.cpu().reshape(M, Cc // 32, 32).abs().amax(-1).repeat_interleave(32, dim=-1)
    assert (d <= 2.0 ** -4 * l32.cpu().abs() + 2.0 ** -8 * bm + 1e-6).all(), d.max().item()


@pytest.mark.parametrize("M,K,F4", [(2048, 320, 1280), (4096, 640, 2560)])
def test_geglu_fp8_output_feeds_ff2(M, K, F4):
    """fp8 x fp8 GEGLU whose direct epilogue writes the gated values as e4m3fn + E8M0 block scales (block amax across the lane pair that
    shares a 32-column block), consumed by the fp8 x fp8 ff.net.2 GEMM.  The quantised hidden tensor must be the rf_quantize_fp8_act
    quantisation of the exact GEGLU values (codes may differ by one where fp32 evaluation order moves an amax across a power of two),
    and ff.net.2 on it must match an fp32 reference on the dequantised operands."""
    dt = torch.bfloat16
    x = (rnd((M, K), 950) * 0.7).to(dt)
    xa = ops.Fp8Act((M, K), DEV)
    ops.quantize_act(x.to(DEV), xa)()
    w1 = rnd((2 * F4, K), 951) / math.sqrt(K)
    b1 = rnd((2 * F4,), 952)
    wp, bp = ops.pack_geglu(w1, b1, torch.float32)
    fw1 = ops.quantize_fp8_padded(wp.to(DEV), 1, K)
    hq = ops.Fp8Act((M, F4), DEV)
    l = ops.linear(xa, fw1, hq, bp.to(DEV), act=ops.ACT_GEGLU)
    l()
    torch.cuda.synchronize()
    assert l.keep[0].out_dtype == 2
    wd = fw1.dequant()[:, :K].cpu()
    wv, wg = wd.reshape(F4 // 32, 2, 32, K)[:, 0].reshape(F4, K), wd.reshape(F4 // 32, 2, 32, K)[:, 1].reshape(F4, K)
    xd = xa.dequant().cpu()
    gate = F.linear(xd, wg, b1[F4:])
    href = F.linear(xd, wv, b1[:F4]) * F.gelu(gate)            # erf GELU; the bf16-path kernels use the tanh form (<= 4.8e-4 * |value|)
    qr, code = _quant_act_ref(href)
    assert (hq.scale[:, :F4 // 32].cpu().int() - code.int()).abs().max() <= 1
    d = (hq.dequant().cpu() - href).abs()
    bm = href.reshape(M, F4 // 32, 32).abs().amax(-1).repeat_interleave(32, dim=-1)
    vmag = F.linear(xd, wv, b1[:F4]).abs()
    assert (d <= 2.0 ** -4 * href.abs() + 2.0 ** -8 * bm + 6e-4 * vmag + 1e-6).all(), d.max().item()
    # ff.net.2 on the quantised hidden tensor
    w2 = rnd((K, F4), 953) / math.sqrt(F4)
    b2 = rnd((K,), 954)
    res, rr = q(rnd((M, K), 955), dt)
    fw2 = ops.quantize_fp8_padded(w2.to(DEV), 1, F4)
    out = torch.empty((M, K), dtype=dt, device=DEV)
    ops.linear(hq, fw2, out, b2.to(DEV), residual=res)()
    torch.cuda.synchronize()
    check(out, F.linear(hq.dequant().cpu(), fw2.dequant().cpu(), b2) + rr, dt)
