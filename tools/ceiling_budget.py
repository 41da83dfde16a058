#!/usr/bin/env python3
"""The ceiling budget of one DDIM step: for every launch class of bench.py's `--profile-json` table

    floor = max(FLOP / R_best, compulsory bytes / BW [, exponentials / R_exp]) + T_launch
                                                                          R_best = 1.24 PFLOP/s  (the best ANY kernel has shown on this chip for this kind of GEMM:
                                                                                   hipBLASLt on 16384 x 640 x 5760, profiles/r04k_vs_vendor_libraries.txt)
                                                                          BW     = 6.5 TB/s      (what L2 misses are served at, tools/fill_probe.py)
                                                                          T_launch = 3 us        (launch + prologue + epilogue drain of a one-round grid)
                                                                          R_exp  = 1.97e13 / s   (attention only: one v_exp_f32 per score at 8 cycles per wave instruction --
                                                                                   tools/exp_probe.py: 40 cycles per 4; never the larger term: the attention floors are FLOP floors)

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
ap.add_argument("--exp-rate", type=float, default=1024 * 8 * 2.4e9, help="v_exp_f32 per second of the chip: 1024 SIMDs x 64 lanes / 8 cycles per instruction x 2.4 GHz")
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
    c["exp_s"] = c.get("exp_s", 0.0) + r.get("exps", 0.0) / a.exp_rate
    c["floor"] += max(r["flop"] / a.rbest, r.get("bytes", 0) / a.bw, r.get("exps", 0.0) / a.exp_rate) + a.tlaunch          # (one launch cost per CALL: a split-K reduce pass or a tail launch is this design's choice)
tot_ms = sum(c["ms"] for c in cls.values())
tot_floor = sum(c["floor"] for c in cls.values()) * 1e3
print(f"# ceiling budget of one DDIM step: {len(rows)} launches, measured sum {tot_ms:.3f} ms; floor = max(FLOP / {a.rbest / 1e15:.2f} PF, bytes / {a.bw / 1e12:.1f} TB/s) + {a.tlaunch * 1e6:.0f} us per call")
print(f"{'class':58s} {'n':>3s} {'us/call':>8s} {'floor us':>9s} {'ratio':>6s} {'ms':>7s} {'floor ms':>9s} {'bound':>6s} {'TF/s':>6s}")
for k, c in sorted(cls.items(), key=lambda kv: -kv[1]["ms"]):
    fl_c, by_c, ex_c = c["flop"] / a.rbest, c["bytes"] / a.bw, c.get("exp_s", 0.0)
    bound = "exp" if ex_c >= max(fl_c, by_c) and ex_c > 0 else ("mfma" if fl_c >= by_c else "hbm")
    print(f"{k[:58]:58s} {c['n']:3d} {c['ms'] / c['n'] * 1e3:8.1f} {c['floor'] / c['n'] * 1e6:9.1f} {c['ms'] * 1e-3 / c['floor']:6.2f} {c['ms']:7.3f} {c['floor'] * 1e3:9.3f} "
          f"{bound:>6s} {c['flop'] / (c['ms'] * 1e-3) / 1e12 if c['ms'] else 0:6.0f}")
def group_of(k):
    if not k.startswith("gemm "):
        return {"rf_attention": "self-attention (d = 40 / 80 / 160)", "rf_ffn_geglu": "token-resident feed-forward + proj_out (C = 320)", "rf_attn_in": "token-resident proj_in + norm1 + qkv (C = 320)", "rf_groupnorm_apply": "GroupNorm + SiLU apply passes"}.get(k, "other passes (fold, finalize, stem, out head, LayerNorm, DDIM glue)")
    M, N, K = (int(v) for v in k.split()[1].split("x"))
    if "geglu" in k:
        return "GEGLU projections (C = 640 / 1280)"
    if M <= 1024:
        return "8x8 level (M = 1024)"
    if K == 5 * N or (K == 1600 and N == 320):
        return "ff.net.2 + proj_out folded (K = 5 C)"
    if K >= 2560:
        return "long-K convolutions (K >= 2560, M >= 4096)"
    return "short-K projections / convolutions (K < 2560, M >= 4096)"
grp = collections.OrderedDict()
for k, c in cls.items():
    g = grp.setdefault(group_of(k), dict(n=0, ms=0.0, floor=0.0, flop=0.0))
    g["n"] += c["n"]; g["ms"] += c["ms"]; g["floor"] += c["floor"]; g["flop"] += c["flop"]
print()
print(f"{'group':70s} {'calls':>5s} {'ms':>7s} {'floor ms':>9s} {'ratio':>6s} {'gap ms':>7s} {'TF/s':>6s}")
for k, g in sorted(grp.items(), key=lambda kv: -(kv[1]["ms"] - kv[1]["floor"] * 1e3)):
    print(f"{k:70s} {g['n']:5d} {g['ms']:7.3f} {g['floor'] * 1e3:9.3f} {g['ms'] * 1e-3 / g['floor']:6.2f} {g['ms'] - g['floor'] * 1e3:7.3f} {g['flop'] / (g['ms'] * 1e-3) / 1e12 if g['ms'] else 0:6.0f}")
print()
print(f"{'TOTAL':58s} {len(rows):3d} {'':8s} {'':9s} {tot_ms / tot_floor:6.2f} {tot_ms:7.3f} {tot_floor:9.3f}")
print(f"# whole-step MFMA utilisation (reference FLOP {a.alg_flop / 1e12:.3f} T over {a.peak / 1e15:.1f} PF): measured sum of kernels {a.alg_flop / (tot_ms * 1e-3) / a.peak:.3f}, "
      f"at the sum of the floors {a.alg_flop / (tot_floor * 1e-3) / a.peak:.3f}")
