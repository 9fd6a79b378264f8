#!/usr/bin/env python3
"""Condense a tools/profile_extra.sh output directory into profiles/<tag>_other_workloads.md."""
import csv
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
lines = [f"# rocprofv3 kernel traces `{tag}` of the non-bench workloads (tools/bench_configs.py, tools/bench_joint.py; MI355X gfx950)", "",
         "| workload | kernel | calls | avg ms | min ms | max ms |", "|---|---|---|---|---|---|"]
for wl in ("configs", "joint"):
    f = os.path.join(src, wl, "t_kernel_stats.csv")
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        if "gpmpc" in r["Name"]:
            lines.append(f"| {wl} | `{r['Name'][:80]}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | "
                         f"{float(r['MinNs'])/1e6:.3f} | {float(r['MaxNs'])/1e6:.3f} |")
lines += ["", "## per-dispatch durations (kernel trace)", ""]
for wl in ("configs", "joint"):
    f = os.path.join(src, wl, "t_kernel_trace.csv")
    if not os.path.exists(f):
        continue
    lines += [f"### {wl}", "```"]
    for r in csv.DictReader(open(f)):
        if "gpmpc" in r["Kernel_Name"] and "plan_kernel" not in r["Kernel_Name"]:
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            lines.append(f"{r['Kernel_Name'][12:60]:48s} grid={r['Grid_Size_X']:>8s} wg={r['Workgroup_Size_X']:>4s} vgpr={r['VGPR_Count']:>3s}+{r['Accum_VGPR_Count']:>3s} lds={r['LDS_Block_Size']:>6s} scratch={r['Scratch_Size']:>4s}: {dur:8.3f} ms")
    lines += ["```", ""]
for wl in ("configs", "joint"):
    f = os.path.join(src, wl + ".log")
    if os.path.exists(f):
        lines += [f"### {wl}: tool output", "```"]
        lines += [l.rstrip() for l in open(f) if ("ms" in l and ("Ns=" in l)) ]
        lines += ["```", ""]
open(os.path.join(dst, f"{tag}_other_workloads.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:14]))
