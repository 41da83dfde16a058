import ctypes, os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libmxprobe.so"))
g = torch.Generator().manual_seed(0)
A = (torch.randn(32, 64, generator=g) * 2).to(torch.float8_e4m3fn)
B = (torch.randn(32, 64, generator=g) * 2).to(torch.float8_e4m3fn)
Ad, Bd = A.view(torch.uint8).cuda(), B.view(torch.uint8).cuda()
def run(sa, sb, opa=0, opb=0):
    pack = lambda s: (s[:, 0] | (s[:, 1] << 8) | (s[:, 2] << 16) | (s[:, 3] << 24)).to(torch.int32)
    out = torch.zeros(64 * 16, dtype=torch.float32, device="cuda")
    sad, sbd = pack(sa).cuda(), pack(sb).cuda()
    lib.mx_probe(ctypes.c_void_p(Ad.data_ptr()), ctypes.c_void_p(Bd.data_ptr()), ctypes.c_void_p(sad.data_ptr()), ctypes.c_void_p(sbd.data_ptr()),
                 ctypes.c_void_p(out.data_ptr()), opa, opb, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = out.cpu().view(64, 16).double()
    C = torch.zeros(32, 32, dtype=torch.float64)          # un-permute with the 32x32 C layout
    lane = torch.arange(64)
    for r in range(16):
        row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
        C[row, lane & 31] = got[:, r]
    return C
ones = torch.full((64, 4), 127, dtype=torch.int32)
C0 = run(ones, ones)
ref = A.double() @ B.double().T
print("unit scales: max err", (C0 - ref).abs().max().item(), "of", ref.abs().max().item())
# per-chunk partial products
P = [A.double()[:, 32 * h:32 * h + 32] @ B.double()[:, 32 * h:32 * h + 32].T for h in range(2)]
# vary ONE lane's scale_a byte b: which C entries change, by what factor
for lane_id in (0, 5, 32, 37):
    for byte in range(4):
        sa = ones.clone(); sa[lane_id, byte] = 130            # x8
        C1 = run(sa, ones, 0, 0)
        d = C1 - C0
        rows = d.abs().sum(1).nonzero().flatten().tolist()
        if not rows:
            continue
        i = rows[0]
        # explain: d[i] = 7 * P[h][i] for which h?
        e = [((d[i] - 7 * P[h][i]).abs().max().item()) for h in range(2)]
        print(f"scale_a lane {lane_id} byte {byte} (opsel 0): rows changed {rows[:6]}{'...' if len(rows) > 6 else ''}; residual vs 7*P[h] h=0: {e[0]:.2e} h=1: {e[1]:.2e}")
for opa in (1, 2, 3):
    for lane_id in (0, 32):
        for byte in range(4):
            sa = ones.clone(); sa[lane_id, byte] = 130
            try:
                C1 = run(sa, ones, opa, 0)
            except Exception as ex:
                continue
            d = C1 - C0
            rows = d.abs().sum(1).nonzero().flatten().tolist()
            if rows:
                i = rows[0]
                e = [((d[i] - 7 * P[h][i]).abs().max().item()) for h in range(2)]
                print(f"opsel_a {opa}: lane {lane_id} byte {byte}: rows {rows[:6]}; residual h0 {e[0]:.2e} h1 {e[1]:.2e}")
# scale_b
for lane_id in (3, 35):
    for byte in range(4):
        sb = ones.clone(); sb[lane_id, byte] = 130
        C1 = run(ones, sb, 0, 0)
        d = C1 - C0
        cols = d.abs().sum(0).nonzero().flatten().tolist()
        if cols:
            j = cols[0]
            e = [((d[:, j] - 7 * P[h][:, j]).abs().max().item()) for h in range(2)]
            print(f"scale_b lane {lane_id} byte {byte} (opsel 0): cols changed {cols[:6]}; residual h0 {e[0]:.2e} h1 {e[1]:.2e}")
print("---- which 8-byte groups of the row's 64-byte slice does a lane's scale cover? (least squares over 8 groups)")
G8 = [A.double()[:, 8 * g:8 * g + 8] @ B.double()[:, 8 * g:8 * g + 8].T for g in range(8)]
for name, which in (("scale_a", 0), ("scale_b", 1)):
    for lane_id in (0, 32, 7, 39):
        s = ones.clone(); s[lane_id, 0] = 130
        C1 = run(s, ones) if which == 0 else run(ones, s)
        d = C1 - C0
        if which == 0:
            i = d.abs().sum(1).argmax().item()
            M = torch.stack([7 * G8[g][i] for g in range(8)], 1)
            sol = torch.linalg.lstsq(M, d[i].unsqueeze(1)).solution.flatten()
        else:
            j = d.abs().sum(0).argmax().item()
            M = torch.stack([7 * G8[g][:, j] for g in range(8)], 1)
            sol = torch.linalg.lstsq(M, d[:, j].unsqueeze(1)).solution.flatten()
        print(name, "lane", lane_id, "-> group weights", [round(v, 2) for v in sol.tolist()])
