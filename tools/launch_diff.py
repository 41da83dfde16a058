#!/usr/bin/env python3
"""Per-launch-class difference of two `bench.py --profile-json` tables (same config, two libraries / settings on one box):
   python tools/launch_diff.py base.json other.json [min_us]"""
import collections
import json
import sys

def table(fn):
    t = collections.OrderedDict()
    for r in json.load(open(fn))["step_launches"]:
        key = (r["family"],) + ((r["M"], r["N"], r["K"], r.get("act"), f"{r.get('bm')}x{r.get('bn')}" + (f" sk{r['splitk']}" if r.get("splitk", 1) > 1 else "")) if "M" in r else ())
        c = t.setdefault(key, [0, 0.0])
        c[0] += 1
        c[1] += r["ms"]
    return t

a, b = table(sys.argv[1]), table(sys.argv[2])
lim = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = []
for k in list(a) + [k for k in b if k not in a]:
    na, ma = a.get(k, [0, 0.0])
    nb, mb = b.get(k, [0, 0.0])
    rows.append((mb - ma, k, na, ma, nb, mb))
print(f"{'class':78s} {'n':>3s} {'base us':>9s} {'other us':>9s} {'delta us (sum)':>15s}")
for d, k, na, ma, nb, mb in sorted(rows):
    if abs(d) * 1e3 >= lim:
        print(f"{str(k)[:78]:78s} {max(na, nb):3d} {ma / max(na, 1) * 1e3:9.1f} {mb / max(nb, 1) * 1e3:9.1f} {d * 1e3:15.1f}")
print(f"total: base {sum(v[1] for v in a.values()):.3f} ms, other {sum(v[1] for v in b.values()):.3f} ms")
