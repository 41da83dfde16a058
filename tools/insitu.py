#!/usr/bin/env python3
"""Per-layer IN-SITU kernel times of one DDIM step: joins a rocprofv3 kernel trace of bench.py (graph replays) with the
launch list written by `bench.py --profile-json` (isolated timings).  Usage: tools/insitu.py <kernel_trace.csv> <prof.json> [--by-shape]"""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "ddim_pack_kernel" in n]
segs = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
L = collections.Counter(b - a for a, b in segs).most_common(1)[0][0]
segs = [s for s in segs if s[1] - s[0] == L][-45:]
dur = [0.0] * L
for a, b in segs:
    for k in range(L):
        r = rows[a + k]
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
dur = [d / len(segs) for d in dur]
iso = json.load(open(sys.argv[2]))["step_launches"]
merged = []
for k in range(L):
    n = names[segs[-1][0] + k]
    if "splitk_reduce" in n:
        merged[-1][1] += dur[k]
    elif "rf::" in n or "gn_finalize_kernel" in n:
        merged.append([n, dur[k]])
assert len(merged) == len(iso), (len(merged), len(iso))
print(f"step: {sum(dur):.1f} us in-situ over {len(segs)} steps, {L} kernels/step")
fam = collections.OrderedDict()
shape = collections.OrderedDict()
for (n, d), l in zip(merged, iso):
    f = fam.setdefault(l["family"], [0, 0.0, 0.0, 0.0])
    f[0] += 1; f[1] += d; f[2] += l["ms"] * 1e3; f[3] += l["flop"]
    if l["family"].startswith("rf_conv_gemm"):
        key = (l["M"], l["N"], l["K"], l["act"])
    else:
        key = (l["family"],)
    v = shape.setdefault(key, [0, 0.0, 0.0, 0.0])
    v[0] += 1; v[1] += d; v[2] += l["ms"] * 1e3; v[3] += l["flop"]
for k, v in fam.items():
    print(f"{k:26s} x{v[0]:3d}  in-situ {v[1]:9.1f} us  isolated {v[2]:9.1f} us  {v[3] / v[1] / 1e6 if v[3] else 0:7.0f} TF in-situ")
if "--by-shape" in sys.argv:
    for k, v in sorted(shape.items(), key=lambda kv: -kv[1][1]):
        print(f"{str(k):42s} x{v[0]:3d}  in-situ {v[1]:8.1f} ({v[1] / v[0]:7.1f} each)  isolated {v[2] / v[0]:7.1f}  {v[3] / v[1] / 1e6 if v[3] else 0:6.0f} TF")
