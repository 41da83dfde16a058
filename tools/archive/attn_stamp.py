#!/usr/bin/env python3
"""Phase stamps of attention_dma_kernel (a library built with -DRF_ATTN_STAMP writes five int32 over the first output row of every wave: cycles of prologue, first tile,
remaining tiles, tail + stores, and the start stamp).  REFACE_HIP_LIB=<stamp build> python tools/archive/attn_stamp.py [d N]"""
import sys
import torch
sys.path.insert(0, ".")
from reface_amd import ops
d, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (80, 1024)
B, heads = 16, 8
C = heads * d
qkv = torch.randn(B, N, 3 * C, device="cuda").to(torch.bfloat16)
out = torch.empty(B, N, C, dtype=torch.bfloat16, device="cuda")
l = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], out, heads=heads, scale=d ** -0.5)
for _ in range(3):
    l()
torch.cuda.synchronize()
o = out.view(B, N, heads, d)[:, ::64].contiguous().view(torch.int32).view(B, N // 64, heads, d // 2)[..., :5].cpu().reshape(-1, 5).double()
names = ["prologue (Q, ones, tiles 0-1 issued, tile 0 landed)", "first tile (4 units)", "remaining tiles", "tail PV + normalise + stores"]
tiles = N // 128
for i, n in enumerate(names):
    print(f"{n:58s} mean {o[:, i].mean():9.0f}  min {o[:, i].min():9.0f}  max {o[:, i].max():9.0f} cycles")
print(f"per unit in the steady loop: {o[:, 2].mean() / ((tiles - 1) * 4):.0f} cycles; wave total {o[:, :4].sum(1).mean():.0f}")
st = o[:, 4]
print(f"start stamps: spread {st.max() - st.min():.0f} cycles over {len(st)} waves; waves starting later than half the spread: {(st > st.min() + (st.max() - st.min()) / 2).sum().item():.0f}")
