#!/bin/bash
# Run ON the GPU box: kernel launches per SQP iteration of the closed loop (configs[4] shard as shipped), by kernel name:
# the difference of two kernel traces (2 and 6 of 6 prepared MPC steps run) over the 16 SQP iterations in between, so that the
# constructor's launches (base samples of all 6 steps, plan) cancel.   bash tools/count_closed_loop_launches.sh <tag>
set -u
TAG=${1:-r4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/cl_launches_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 2 6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$n -o cl -- python3 $ROOT/tools/bench_closed_loop.py --mpc-steps 6 --run-steps $n > $OUT/run$n.log 2>&1
done
cd $ROOT
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections, os
out, tag = sys.argv[1], sys.argv[2]
def counts(n):
    f = glob.glob(os.path.join(out, f"t{n}", "**", "*kernel_stats.csv"), recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = counts(2), counts(6)
rows = []
for k, (c6, t6) in b.items():
    c2, t2 = a.get(k, (0, 0.0))
    if c6 - c2 > 0:
        rows.append((k, (c6 - c2) / 16.0, (t6 - t2) / 16.0 / 1e3))
rows.sort(key=lambda r: -r[1])
lines = [f"# kernel launches per SQP iteration of the closed loop ({tag}): traces of 6 and 2 MPC steps, difference / 16 iterations", "",
         "| kernel | launches / iteration | us / iteration |", "|---|---|---|"]
tot = gp = 0.0
for k, c, t in rows:
    lines.append(f"| `{k[:100]}` | {c:.2f} | {t:.1f} |")
    tot += c
    gp += c if "gpmpc" in k else 0.0
lines += ["", f"total {tot:.1f} launches per iteration, of which gpmpc kernels {gp:.1f}, others {tot - gp:.1f}", ""]
for n in (2, 6):
    lines += [f"## run{n}.log", "```"] + open(os.path.join(out, f"run{n}.log")).read().splitlines()[-8:] + ["```"]
open(os.path.join(out, "launches.md"), "w").write("\n".join(lines))
print("\n".join(lines))
PY
rm -rf $OUT/t2 $OUT/t6
