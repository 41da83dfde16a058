#!/bin/bash
# Hardware-counted matrix-pipe occupancy of the bench workload per kernel family:  SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's matrix
# pipe is busy, summed over the 1024 SIMDs) against the kernel's duration x 1024 SIMDs, plus the MFMA instruction count.  One
# rocprofv3 --pmc pass (no trace domains) for the counters, one --kernel-trace pass of the same command for the durations.
#   tools/pmc_mfma.sh <tag> [cfg]   ->  gpurun_out/<tag>_mfma_busy.json   (copy to profiles/)
tag=$1
cfg=${2:-c1}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-conditioning --no-parity --no-other-configs $EXTRA"
REFACE_NO_GRAPH=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d gpurun_out/${tag}_pmc_mfma -- $CMD > gpurun_out/${tag}_pmc_mfma.log 2>&1
REFACE_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_kt_mfma -- $CMD > gpurun_out/${tag}_kt_mfma.log 2>&1
python3 - <<PY
import csv, glob, collections, json, re
def fam(n):
    if "conv_gemm_kernel<unsigned char" in n: return "rf_conv_gemm[fp8]"          # fp8 activations x fp8 weights (MX-scaled MFMA)
    if "conv_gemm_kernel<_Float16" in n: return "rf_conv_gemm[f16]"               # fp16 operands (the fp16 throughput mode)
    mm = re.match(r"_ZN2rf\d+([a-z0-9_]+?)_kernelI", n)                             # (rocprofv3 leaves names with a _Float16 template argument -- DF16_ -- mangled)
    if mm: return "rf_conv_gemm[f16]" if mm.group(1) == "conv_gemm" else "rf_" + mm.group(1)
    m = re.search(r"conv_gemm_kernel<(unsigned short|float), (unsigned short|float)", n)
    if m:
        if m.group(1) == "unsigned short" and m.group(2) == "float": return "rf_conv_gemm[bf16x3]"      # bf16 operands, fp32 out: the split-bf16 VAE convs (+ the UNet's 4-channel out conv)
        if re.search(r", true, \d+>\(", n): return "rf_conv_gemm[fp8w]"          # template arguments end with ..., W8, LNF
        return "rf_conv_gemm[%s]" % ("bf16" if m.group(1) == "unsigned short" else "f32")
    m = re.search(r"rf::(\w+?)_kernel", n)
    return "rf_" + m.group(1) if m else "other"
busy, insts, cnt = collections.Counter(), collections.Counter(), collections.Counter()
for f in glob.glob("gpurun_out/${tag}_pmc_mfma/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = fam(r["Kernel_Name"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": busy[k] += float(r["Counter_Value"]); cnt[k] += 1
        elif r["Counter_Name"] == "SQ_INSTS_MFMA": insts[k] += float(r["Counter_Value"])
dur = collections.Counter()
for f in glob.glob("gpurun_out/${tag}_kt_mfma/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        dur[fam(r["Name"])] += float(r["TotalDurationNs"])
out = {}
CLK = 2.4e9          # shader clock the utilisation is quoted at (the peak figures are at this clock)
for k in busy:
    if dur[k] <= 0: continue
    out[k] = {"launches": cnt[k], "mfma_busy_cycles": busy[k], "mfma_insts": insts[k], "duration_ms_total": dur[k] / 1e6,
              "matrix_pipe_busy_frac_at_2.4GHz": busy[k] / (dur[k] * 1e-9 * CLK * 1024)}
tb, td = sum(v["mfma_busy_cycles"] for k, v in out.items() if "f32" not in k), sum(v["duration_ms_total"] for k, v in out.items() if "f32" not in k)
import sys
sys.path.insert(0, ".")
import bench
c = bench.CONFIGS["$cfg"]
out["_meta"] = {"lib_digest": bench.lib_digest(), "workload": "$cfg:%dx%d:S50:B%d:%s" % (8 * c["latent"], 8 * c["latent"], c["batch"], ("$EXTRA".split("--dtype ")[1].split()[0] if "--dtype " in "$EXTRA" else c["dtype"])),
                "command": "tools/pmc_mfma.sh ${tag} $cfg (REFACE_NO_GRAPH=1, one batch of eager launches; counters and durations from two passes of the same command)",
                "note": "a v_mfma_f32_32x32x16_bf16 keeps the pipe busy for 32 cycles; busy_frac = busy cycles / (duration x 2.4 GHz x 1024 SIMDs); fp32 families are the VAE decode"}
json.dump(out, open("gpurun_out/${tag}_mfma_busy.json", "w"), indent=1)
for k, v in sorted(((k, v) for k, v in out.items() if k != "_meta"), key=lambda kv: -kv[1]["duration_ms_total"]):
    print(f"{k:28s} launches {v['launches']:6d}  duration {v['duration_ms_total']:9.1f} ms  matrix pipe busy {100 * v['matrix_pipe_busy_frac_at_2.4GHz']:5.1f} %")
PY
rm -rf gpurun_out/${tag}_pmc_mfma gpurun_out/${tag}_kt_mfma
