"""Check the lane -> address -> result mapping of ds_read_b64_tr_b16 that attention's V fragments rely on."""
import ctypes, os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libtrprobe.so"))
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 80        # row bytes
lane = torch.arange(64)
grp, l = lane // 16, lane % 16
base = grp * 1024                                            # every 16-lane group reads its own [4 rows][16 cols] block
addr = (base + (l // 4) * stride + (l % 4) * 8).to(torch.int32).cuda()
out = torch.zeros(64 * 4, dtype=torch.int16, device="cuda")
rc = lib.tr_probe(ctypes.c_void_p(addr.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
got = out.cpu().view(64, 4).to(torch.int32) & 0xFFFF
j = torch.arange(4)
want = (base[:, None] + j[None, :] * stride) // 2 + l[:, None]          # lane i of the group: column i of rows 0..3
print("rc", rc, "match hypothesis (lane i <- column i of the 4 rows):", bool((got == want).all()))
if not (got == want).all():
    for L in range(0, 64, 1):
        print(L, got[L].tolist(), want[L].tolist())
