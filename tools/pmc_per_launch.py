#!/usr/bin/env python3
"""Diagnostics: per-LAUNCH fabric traffic of one DDIM step.  Run under rocprofv3 (one counter per pass):

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/X_fetch -- python3 tools/pmc_per_launch.py --config c1 --list gpurun_out/X_launches.json
    python3 tools/pmc_per_launch.py --join gpurun_out/X_launches.json gpurun_out/X_fetch [gpurun_out/X_write]

The first form builds the engine of a BASELINE config, runs its launch list eagerly three times (warm caches as inside the loop); the LAST pass (from the last
ddim_pack_input dispatch on) is what --join reports.  The second form (no GPU) joins the tail of the
counter CSV with the launch names (a split-K rf_conv_gemm is two kernels) and prints MB read per launch beside the operand sizes."""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def collect(args):
    import torch
    import bench
    from reface_amd import ops
    from reface_amd.ddim import DDIMSampler
    conf = bench.CONFIGS[args.config]
    B, h, dname = conf["batch"], conf["latent"], conf["dtype"]
    dtype = {"bf16": torch.bfloat16, "fp8": "fp8", "fp8c": "fp8c"}[dname]
    torch.cuda.set_device(0)
    unet, vae, ldm, _ = bench.build_models(dtype, "cuda:0", 0, 1, False)
    os.environ["REFACE_NO_GRAPH"] = "1"          # (counter collection cannot trace graph replays)
    sampler = DDIMSampler(ldm)
    x_T, z_inp, mask, c, uc = bench.synthetic_inputs(B, h, 42, "cuda:0")
    sampler.sample(S=2, conditioning=c, batch_size=B, shape=[4, h, h], verbose=False, unconditional_guidance_scale=3.5, unconditional_conditioning=uc, eta=0.0,
                   x_T=x_T, test_model_kwargs={"inpaint_image": z_inp, "inpaint_mask": mask})
    plan = list(sampler._plans.values())[0]
    rows = []
    for l in plan["step"]:
        r = {"name": l.name, "fn": l.fn.__name__, "kernels": 1}
        if l.fn.__name__ == "rf_conv_gemm":
            d = l.keep[0]
            pl = ops.gemm_plan2(l)
            r.update(M=d.M, N=d.N, K=d.K, KH=d.KH, C0=d.C0, act=d.act, splitk=pl["splitk"], bm=pl["bm"], bn=pl["bn"], res=bool(d.residual))
            # (split-K: GEMM + reduce pass; tail-round split along N: two GEMM kernels behind one call -- the library's plan says which)
            r["kernels"] = (2 if pl["splitk"] > 1 else 1) + (pl["gemm_kernels"] - 1)
        elif l.fn.__name__ == "rf_gn_silu_conv3x3_small":
            r["kernels"] = 2
        rows.append(r)
    with open(args.list, "w") as f:
        json.dump(rows, f)
    sp = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        for l in plan["step"]:
            l(sp)
        torch.cuda.synchronize()
    sys.stdout.flush()          # (normal exit: the profiler flushes its tables at exit; --join finds the last pass by its first kernel)


def join(args):
    rows = json.load(open(args.join[0]))
    nk = sum(r["kernels"] for r in rows)
    cols = []
    for d in args.join[1:]:
        recs = []
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                recs.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"]), r["Kernel_Name"], int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)))
        recs.sort()
        # one record per dispatch and counter name; keep the first counter name seen
        cname = recs[-1][1]
        recs = [x for x in recs if x[1] == cname]
        first = max(i for i, x in enumerate(recs) if "ddim_pack" in x[3])          # the last pass starts at the last ddim_pack_input dispatch
        assert first + nk <= len(recs), (first, nk, len(recs))
        cols.append((cname, recs[first:first + nk]))
    out = []
    pos = 0
    print(f"{'launch':58s} {'M':>6s} {'N':>6s} {'K':>6s} sk  blocks " + " ".join(f"{c[0]+' MB':>16s}" for c in cols) + "   A MB   W MB   out MB")
    tot = [0.0] * len(cols)
    for r in rows:
        vals = []
        for ci, (cname, recs) in enumerate(cols):
            v = sum(recs[pos + k][2] for k in range(r["kernels"]))
            mb = v * 1024 / 1e6 * (2.0 if cname == "FETCH_SIZE" else 1.0)          # KB units; gfx950 FETCH_SIZE counts 128-B requests as 64 B (guide)
            vals.append(mb)
            tot[ci] += mb
        blocks = cols[0][1][pos][4] if cols else 0
        kn = cols[0][1][pos][3] if cols else ""
        pos += r["kernels"]
        if r["fn"] == "rf_conv_gemm":
            a_mb = r["M"] * (r["C0"] if r["KH"] > 1 else r["K"]) * 2 / 1e6
            w_mb = r["N"] * r["K"] * 2 / 1e6
            o_mb = r["M"] * (r["N"] // 2 if r["act"] == 1 else r["N"]) * 2 / 1e6 * (2 if r["res"] else 1)
            print(f"{r['name'][:58]:58s} {r['M']:6d} {r['N']:6d} {r['K']:6d} {r['splitk']:2d} {blocks:6d} " + " ".join(f"{v:16.1f}" for v in vals) + f" {a_mb:6.1f} {w_mb:6.1f} {o_mb:6.1f}")
            assert "conv_gemm" in kn, (r["name"], kn)
        else:
            print(f"{r['name'][:58]:58s} {'':6s} {'':6s} {'':6s}    {blocks:6d} " + " ".join(f"{v:16.1f}" for v in vals) + f"   [{r['fn']}]")
        out.append(dict(r, **{c[0]: v for c, v in zip(cols, vals)}))
    print("total MB per step: " + ", ".join(f"{c[0]} {t:.0f}" for c, t in zip(cols, tot)))
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c1")
    ap.add_argument("--list", default=None)
    ap.add_argument("--join", nargs="+", default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.join:
        join(a)
    else:
        collect(a)
