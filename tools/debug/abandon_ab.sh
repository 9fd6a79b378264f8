#!/bin/bash
# Run ON the GPU box: kernel-trace A/B of GPMPC_JOINT_ABANDON (0 / 1) on the k = 0 joint draw of the car closed loop
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  export GPMPC_JOINT_ABANDON=$v
  python3 $ROOT/tools/debug/k0_draw.py 2>/dev/null | grep "k=0"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab$v -o t -- python3 $ROOT/tools/debug/k0_draw.py > /tmp/ab$v.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/ab$v/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "gpmpc" in r["Name"]:
        print("ABANDON=$v", r["Name"][:60], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1), "max", round(float(r["MaxNs"]) / 1e3, 1))
PY
done
