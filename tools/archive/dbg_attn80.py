import sys, torch
sys.path.insert(0, '.')
from reface_amd import ops
torch.manual_seed(0)
B, heads, d = 4, 8, 80
C = heads * d
for Nq, Nk, mode in [(1024, 384, "cold"), (1024, 1024, "cold"), (1024, 384, "plain"), (1024, 512, "cold"), (256, 384, "cold")]:
    qx, kx, vx = torch.randn(B, Nq, C), torch.randn(B, Nk, C), torch.randn(B, Nk, C)
    if mode == "cold":
        kx = -qx[:, :1].repeat(1, Nk, 1) * 3.0 + 0.1 * kx
        qx = qx[:, :1].repeat(1, Nq, 1) * 3.0 + 0.1 * qx
    q, k, v = (t.to(torch.bfloat16).cuda() for t in (qx, kx, vx))
    out = torch.empty(B, Nq, C, dtype=torch.bfloat16, device="cuda")
    ops.attention(q, k, v, out, heads=heads, scale=d ** -0.5)()
    torch.cuda.synchronize()
    o = out.float().cpu().reshape(B, Nq, heads, d)
    bad = ~torch.isfinite(o)
    sp = lambda t, n: t.to(torch.bfloat16).float().reshape(B, n, heads, d).transpose(1, 2).double()
    s = sp(qx, Nq) @ sp(kx, Nk).transpose(-1, -2) * d ** -0.5 * 1.4426950408889634
    print(mode, Nq, Nk, "bad", int(bad.sum()), "of", bad.numel(), "score range (exp2 domain)", float(s.min()), float(s.max()))
    if bad.any():
        idx = bad.nonzero()
        print("  first bad (b, q, h, c):", idx[:5].tolist(), " bad per head:", bad.sum((0, 1, 3)).tolist(), " bad q range:", int(idx[:, 1].min()), int(idx[:, 1].max()),
              " bad c range:", int(idx[:, 3].min()), int(idx[:, 3].max()))
