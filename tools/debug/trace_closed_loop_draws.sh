#!/bin/bash
# Run ON the GPU box: kernel trace of the closed loop's joint draws (tools/debug/eigh_phases_closed_loop.py, two MPC steps on the
# loop's own linearisation points), our kernels in launch order with their durations.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_cl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $ROOT/tools/debug/eigh_phases_closed_loop.py > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gpmpc" in n:
            rows.append((int(r["Start_Timestamp"]), n.split("(")[0].replace("void gpmpc::", "")[:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
for t, n, d in rows:
    if "joint" in n:
        print(f"{n:62s} {d:9.1f} us")
PY
grep n_ho $OUT/run.log | cut -c1-30
