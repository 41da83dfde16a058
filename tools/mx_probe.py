"""Pin the operand / scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 that the fp8 x fp8 GEMM path relies on:
   (1) lane l = (row l & 31, 32-byte K chunk l >> 5) for A and B alike; (2) C layout as the bf16 32x32 MFMA; (3) the E8M0 scale of
   (row, chunk) comes from byte `opsel` of the scale register of THAT lane."""
import ctypes, os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libmxprobe.so"))
g = torch.Generator().manual_seed(0)
A = (torch.randn(32, 64, generator=g) * 2).to(torch.float8_e4m3fn)
B = (torch.randn(32, 64, generator=g) * 2).to(torch.float8_e4m3fn)
ok_all = True
for opa, opb in ((0, 0), (1, 0), (0, 2), (3, 3)):
    sa = torch.randint(120, 134, (64, 4), generator=g, dtype=torch.int32)
    sb = torch.randint(120, 134, (64, 4), generator=g, dtype=torch.int32)
    pack = lambda s: (s[:, 0] | (s[:, 1] << 8) | (s[:, 2] << 16) | (s[:, 3] << 24)).to(torch.int32)
    out = torch.zeros(64 * 16, dtype=torch.float32, device="cuda")
    Ad, Bd = A.view(torch.uint8).cuda(), B.view(torch.uint8).cuda()
    sad, sbd = pack(sa).cuda(), pack(sb).cuda()
    rc = lib.mx_probe(ctypes.c_void_p(Ad.data_ptr()), ctypes.c_void_p(Bd.data_ptr()), ctypes.c_void_p(sad.data_ptr()), ctypes.c_void_p(sbd.data_ptr()),
                      ctypes.c_void_p(out.data_ptr()), opa, opb, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = out.cpu().view(64, 16)
    Af, Bf = A.double(), B.double()
    # hypothesis: scale of (row i, chunk h) = byte opsel of lane i + 32 h
    sA = torch.stack([2.0 ** (sa[:32, opa].double() - 127), 2.0 ** (sa[32:, opa].double() - 127)], 1)      # [row, chunk]
    sB = torch.stack([2.0 ** (sb[:32, opb].double() - 127), 2.0 ** (sb[32:, opb].double() - 127)], 1)
    C = torch.zeros(32, 32, dtype=torch.float64)
    for h in range(2):
        C += (Af[:, 32 * h:32 * h + 32] * sA[:, h:h + 1]) @ (Bf[:, 32 * h:32 * h + 32] * sB[:, h:h + 1]).T
    lane = torch.arange(64)
    want = torch.zeros(64, 16, dtype=torch.float64)
    for r in range(16):
        row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
        want[:, r] = C[row, lane & 31]
    err = (got.double() - want).abs().max().item()
    ok = err < 1e-3 * want.abs().max().item()
    ok_all &= ok
    print(f"rc {rc} opsel ({opa},{opb}): max err {err:.3e} of {want.abs().max().item():.3e} -> {'MATCH' if ok else 'MISMATCH'}")
print("MX_PROBE_OK" if ok_all else "MX_PROBE_FAIL")
