#!/usr/bin/env python3
"""The ceiling budget of one DDIM step: for every launch class of bench.py's `--profile-json` table

    floor = max(FLOP / R_best, compulsory bytes / BW) + T_launch          R_best = 1.24 PFLOP/s  (the best ANY kernel has shown on this chip for this kind of GEMM:
                                                                                   hipBLASLt on 16384 x 640 x 5760, profiles/r04k_vs_vendor_libraries.txt)
                                                                          BW     = 6.5 TB/s      (what L2 misses are served at, tools/fill_probe.py)
                                                                          T_launch = 3 us        (launch + prologue + epilogue drain of a one-round grid)

against the measured time, and the sum of the floors = the whole-step time THIS DESIGN (this launch list, these algebraic reductions) could reach if every
kernel ran at the best rate seen on the chip -- and the MFMA utilisation that would be, priced like bench.py prices `unet_mfma_util_*` (reference FLOP count
over 2.5 PF).   python tools/ceiling_budget.py profile.json [--alg-flop 12.751e12] > profiles/rNN_ceiling_budget.txt"""
import argparse
import collections
import json

ap = argparse.ArgumentParser()
ap.add_argument("profile")
ap.add_argument("--rbest", type=float, default=1.24e15)
ap.add_argument("--bw", type=float, default=6.5e12)
ap.add_argument("--tlaunch", type=float, default=3e-6)
ap.add_argument("--alg-flop", type=float, default=2 * 8 * 796.94e9, help="reference FLOP of one UNet evaluation on the CFG batch (bench.py F_UNET x 2B)")
ap.add_argument("--peak", type=float, default=2.5e15)
a = ap.parse_args()
rows = json.load(open(a.profile))["step_launches"]
cls = collections.OrderedDict()
for r in rows:
    if r["family"].startswith("rf_conv_gemm"):
        key = f"gemm {r['M']}x{r['N']}x{r['K']}" + (" 3x3" if r.get("KH", 1) == 3 else "") + (" geglu" if r.get("act") == 1 else "") + (" +res" if r.get("residual") else "") + \
              f" [{r.get('bm', '?')}x{r.get('bn', '?')}" + (f" sk{r['splitk']}" if r.get("splitk", 1) > 1 else "") + "]"
    else:
        key = r["family"]
    c = cls.setdefault(key, dict(n=0, ms=0.0, flop=0.0, bytes=0.0, floor=0.0, kernels=0))
    c["n"] += 1
    c["ms"] += r["ms"]
    c["flop"] += r["flop"]
    c["bytes"] += r.get("bytes", 0)
    nk = r.get("gemm_kernels", 1) + (1 if r.get("splitk", 1) > 1 else 0)
    c["kernels"] += nk
    c["floor"] += max(r["flop"] / a.rbest, r.get("bytes", 0) / a.bw) + a.tlaunch          # (one launch cost per CALL: a split-K reduce pass or a tail launch is this design's choice)
tot_ms = sum(c["ms"] for c in cls.values())
tot_floor = sum(c["floor"] for c in cls.values()) * 1e3
print(f"# ceiling budget of one DDIM step: {len(rows)} launches, measured sum {tot_ms:.3f} ms; floor = max(FLOP / {a.rbest / 1e15:.2f} PF, bytes / {a.bw / 1e12:.1f} TB/s) + {a.tlaunch * 1e6:.0f} us per call")
print(f"{'class':58s} {'n':>3s} {'us/call':>8s} {'floor us':>9s} {'ratio':>6s} {'ms':>7s} {'floor ms':>9s} {'bound':>6s} {'TF/s':>6s}")
for k, c in sorted(cls.items(), key=lambda kv: -kv[1]["ms"]):
    fl_c, by_c = c["flop"] / a.rbest, c["bytes"] / a.bw
    print(f"{k[:58]:58s} {c['n']:3d} {c['ms'] / c['n'] * 1e3:8.1f} {c['floor'] / c['n'] * 1e6:9.1f} {c['ms'] * 1e-3 / c['floor']:6.2f} {c['ms']:7.3f} {c['floor'] * 1e3:9.3f} "
          f"{'mfma' if fl_c >= by_c else 'hbm':>6s} {c['flop'] / (c['ms'] * 1e-3) / 1e12 if c['ms'] else 0:6.0f}")
print(f"{'TOTAL':58s} {len(rows):3d} {'':8s} {'':9s} {tot_ms / tot_floor:6.2f} {tot_ms:7.3f} {tot_floor:9.3f}")
print(f"# whole-step MFMA utilisation (reference FLOP {a.alg_flop / 1e12:.3f} T over {a.peak / 1e15:.1f} PF): measured sum of kernels {a.alg_flop / (tot_ms * 1e-3) / a.peak:.3f}, "
      f"at the sum of the floors {a.alg_flop / (tot_floor * 1e-3) / a.peak:.3f}")
