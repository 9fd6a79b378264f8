#!/usr/bin/env python3
"""Condense a tools/profile_closed_loop.sh directory into profiles/<tag>_closed_loop_trace.md: per HIP API function the
calls / total / max, the longest single calls, the longest gaps BETWEEN consecutive API calls of the main thread (host time
outside the runtime) and what ran on the GPU meanwhile."""
import collections
import csv
import glob
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(pattern):
    fs = glob.glob(os.path.join(src, "trace", "**", pattern), recursive=True)
    return fs[0] if fs else None


def col(row, *names):
    for n in names:
        if n in row:
            return row[n]
    raise KeyError(names)


api_f, ker_f = find("*hip_api_trace.csv"), find("*kernel_trace.csv")
lines = [f"# Closed loop under `rocprofv3 --hip-trace --kernel-trace --stats` (`{tag}`; tools/profile_closed_loop.sh, MI355X gfx950)", ""]
for name in ("untraced.log", "traced.log"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        lines += [f"## {name} (wall clock of the GP side per SQP iteration, ms)", "", "```"]
        lines += [ln.rstrip() for ln in open(p) if ln.startswith(("MPC step", "params_"))]
        lines += ["```", ""]
if api_f is None:
    lines.append("no HIP API trace found")
else:
    calls = []
    for r in csv.DictReader(open(api_f)):
        calls.append((int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Function"), col(r, "Thread_Id")))
    calls.sort()
    t_first, t_last = calls[0][0], max(c[1] for c in calls)
    by_thread = collections.Counter(c[3] for c in calls)
    main = by_thread.most_common(1)[0][0]
    agg = collections.defaultdict(lambda: [0, 0, 0])
    for s, e, f, th in calls:
        a = agg[f]
        a[0] += 1
        a[1] += e - s
        a[2] = max(a[2], e - s)
    lines += [f"## HIP API calls: {len(calls)} over {(t_last - t_first) * 1e-6:.0f} ms ({len(by_thread)} threads; main thread {by_thread[main]} calls)", "",
              "| function | calls | total ms | mean us | max ms |", "|---|---|---|---|---|"]
    for f, (n, tot, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        lines.append(f"| `{f}` | {n} | {tot * 1e-6:.2f} | {tot / n * 1e-3:.1f} | {mx * 1e-6:.3f} |")
    kernels = []
    if ker_f:
        for r in csv.DictReader(open(ker_f)):
            kernels.append((int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Kernel_Name")))
        kernels.sort()

    def gpu_busy(a, b):
        tot, names = 0, collections.Counter()
        for s, e, n in kernels:
            if e <= a or s >= b:
                continue
            d = min(e, b) - max(s, a)
            tot += d
            names[n.split("(")[0][-48:]] += d
        return tot, names

    lines += ["", "## the ten longest single API calls", "", "| ms | function | at ms | GPU busy inside the call (ms) | longest kernel inside |", "|---|---|---|---|---|"]
    for s, e, f, th in sorted(calls, key=lambda c: c[0] - c[1])[:10]:
        busy, names = gpu_busy(s, e)
        top = names.most_common(1)[0][0] if names else "-"
        lines.append(f"| {(e - s) * 1e-6:.2f} | `{f}` | {(s - t_first) * 1e-6:.0f} | {busy * 1e-6:.2f} | `{top}` |")
    mc = [c for c in calls if c[3] == main]
    gaps = []
    for (s0, e0, f0, _), (s1, e1, f1, _) in zip(mc, mc[1:]):
        if s1 > e0:
            gaps.append((s1 - e0, e0, s1, f0, f1))
    gaps.sort(reverse=True)
    host_total = sum(g[0] for g in gaps)
    lines += ["", f"## host time between consecutive API calls of the main thread: {host_total * 1e-6:.0f} ms in total; the ten longest gaps", "",
              "| ms | after | before | at ms | GPU busy during the gap (ms) |", "|---|---|---|---|---|"]
    for d, a, b, f0, f1 in gaps[:10]:
        busy, _ = gpu_busy(a, b)
        lines.append(f"| {d * 1e-6:.2f} | `{f0}` | `{f1}` | {(a - t_first) * 1e-6:.0f} | {busy * 1e-6:.2f} |")
    big_calls = [c for c in calls if c[1] - c[0] > 20e6]
    big_gaps = [g for g in gaps if g[0] > 20e6]
    lines += ["", f"API calls longer than 20 ms: {len(big_calls)}; host gaps longer than 20 ms: {len(big_gaps)}.", ""]
    if kernels:
        kagg = collections.defaultdict(lambda: [0, 0])
        for s, e, n in kernels:
            kagg[n.split("(")[0][-60:]][0] += 1
            kagg[n.split("(")[0][-60:]][1] += e - s
        lines += ["## kernels (total over the run)", "", "| kernel | launches | total ms |", "|---|---|---|"]
        for n, (c, t) in sorted(kagg.items(), key=lambda kv: -kv[1][1])[:10]:
            lines.append(f"| `{n}` | {c} | {t * 1e-6:.2f} |")
dst = os.path.join(root, "gpurun_out", f"{tag}_closed_loop_trace.md")
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
