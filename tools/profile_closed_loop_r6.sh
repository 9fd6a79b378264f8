#!/bin/bash
# Run ON the GPU box: kernel trace + PMC passes of the car closed loop AS SHIPPED in its own sequence (configs[4] shard: Ns = 1024, H = 40,
# three MPC steps x four SQP iterations through Agent.sqp_linearisation, tools/debug/closed_loop_fused_phases.py): from round 6 a draw
# leaves pending rows for the next one, so the draws cannot be profiled as repetitions of one iteration any more.  Every pass its own
# process (FETCH_SIZE and WRITE_SIZE alone, as MI355X_MICROARCH.md prescribes); the summary is PER KERNEL: launches, mean duration,
# mean counter values over ALL launches of the run (the throw-away Agent's included).
#   tools/profile_closed_loop_r6.sh [tag]
set -u
TAG=${1:-r6}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_cl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/tools/debug/closed_loop_fused_phases.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o pmc -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o pmc -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -o pmc -- python3 $ARGS > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq2 -o pmc -- python3 $ARGS > $OUT/sq2.log 2>&1
grep "step" $OUT/trace.log
python3 - <<PY | tee $OUT/summary.txt
import collections, csv, glob
def short(n):
    n = n.replace("void gpmpc::", "")
    return n.split("(")[0][:58]
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/trace/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "gpmpc" in r["Kernel_Name"]:
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write", "sq", "sq2"):
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            if "gpmpc" in r["Kernel_Name"]:
                cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    v = dur[k]
    print("%-60s launches %4d  mean %9.1f us  total %9.1f us" % (k, len(v), sum(v) / len(v), sum(v)))
    for c, vals in sorted(cnt[k].items()):
        print("      %-28s mean %.4g" % (c, sum(vals) / len(vals)))
PY
