#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh output directory into profiles/<tag>_summary.md (+ copies of the small CSVs)."""
import collections
import csv
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)
lines = [f"# rocprofv3 summary `{tag}` (kernel trace: `python bench.py --cpu-sample 0`; PMC passes: the headline workload "
         f"only, `--steps 20 --warmup 3 --no-extra --reach-ns 0`; MI355X gfx950)", ""]
ks = os.path.join(src, "trace", "trace_kernel_stats.csv")
shutil.copy(ks, os.path.join(dst, f"{tag}_kernel_stats.csv"))
lines += ["## kernel trace (--kernel-trace --stats)", "", "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
for r in csv.DictReader(open(ks)):
    if "gpmpc" in r["Name"] or float(r["Percentage"]) > 1.0:
        lines.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")
lines += ["", "## PMC counters (separate --pmc passes), mean per launch of the gpmpc kernels", "",
          "| kernel | counter | launches | mean |", "|---|---|---|---|"]
for sub in ["pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"]:
    f = os.path.join(src, sub, "pmc_counter_collection.csv")
    if not os.path.exists(f):
        continue
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "gpmpc" in r["Kernel_Name"]:
            d[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        for c, vals in v.items():
            lines.append(f"| `{k}` | {c} | {len(vals)} | {sum(vals)/len(vals):.6g} |")
lines += ["", "FETCH_SIZE / WRITE_SIZE are in KiB per launch as rocprofv3 reports them; per MI355X_MICROARCH.md (HBM section)",
          "FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, so HBM bytes <= (2*FETCH_SIZE + WRITE_SIZE) * 1024.", ""]
# the counters of the headline launch's rollout kernel, for bench.py's `roofline` (PMC counters cannot be read in-process)
import json
def _mean(sub, counter, kernel_key):
    f = os.path.join(src, sub, "pmc_counter_collection.csv")
    if not os.path.exists(f):
        return None, None
    vals, name = [], None
    for r in csv.DictReader(open(f)):
        if kernel_key in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
            name = r["Kernel_Name"]
    return (sum(vals) / len(vals) if vals else None), name
for key in ("rollout_one_kernel", "rollout_fast_kernel", "rollout_tiles_kernel"):
    fetch, kname = _mean("pmc_fetch", "FETCH_SIZE", key)
    if fetch is None:
        continue
    write, _ = _mean("pmc_write", "WRITE_SIZE", key)
    valu, _ = _mean("pmc_sq", "SQ_INSTS_VALU", key)
    act, _ = _mean("pmc_sq", "SQ_ACTIVE_INST_VALU", key)
    cyc, _ = _mean("pmc_sq", "SQ_WAVE_CYCLES", key)
    busy, _ = _mean("pmc_sq", "SQ_BUSY_CYCLES", key)
    mfma_busy, _ = _mean("pmc_sq2", "SQ_VALU_MFMA_BUSY_CYCLES", key)
    mops, _ = _mean("pmc_sq2", "SQ_INSTS_VALU_MFMA_MOPS_F64", key)
    out = {"source": f"profiles/{tag}_summary.md (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 20)",
           "kernel": kname[:80], "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
           "hbm_bytes_per_launch_uncorrected": (fetch + (write or 0.0)) * 1024,
           "hbm_bytes_per_launch_gfx950_corrected": (2 * fetch + (write or 0.0)) * 1024,
           "workload": {"Ns": 1024, "H": 30}, "valu_insts_per_launch": valu,
           "valu_active_frac": (act / cyc) if (act and cyc) else None,
           # SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_WAVE_CYCLES quad-cycles; one wave per SIMD: wave time = SIMD time
           "mfma_busy_frac": (mfma_busy / (4.0 * cyc)) if (mfma_busy and cyc) else None,
           "mfma_mops_f64_per_launch": mops,
           "valu_source": f"profiles/{tag}_summary.md (SQ_INSTS_VALU; SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES; SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_WAVE_CYCLES))"}
    json.dump(out, open(os.path.join(dst, "latest_traffic.json"), "w"), indent=1)
    lines += ["", "latest_traffic.json <- " + json.dumps(out)]
    break
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines))
