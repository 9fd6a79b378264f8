#!/bin/bash
# Run ON the GPU box: where the headline kernel's wave cycles go by instruction class (one PMC pass each).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_issue
mkdir -p $OUT $ROOT/gpurun_out/ret
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-extra --reach-ns 0 --prewarm 200"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -o pmc -- python3 $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/b -o pmc -- python3 $ARGS > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS --output-format csv -d $OUT/c -o pmc -- python3 $ARGS > $OUT/c.log 2>&1
python3 - <<PY
import csv, collections, glob
tot = {}
for sub in "abc":
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rollout_one" in r["Kernel_Name"]:
                d[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in d.items():
            tot[k] = sum(v) / len(v)
wc = tot.get("SQ_WAVE_CYCLES", 1.0)
per = 1024 * 30.0
with open("$ROOT/gpurun_out/ret/one_issue_counters.txt", "w") as o:
    for k in sorted(tot):
        line = f"{k:32s} {tot[k]:14.4g} per launch   {tot[k] / per:9.1f} per wave-step   {tot[k] / wc:7.3f} of SQ_WAVE_CYCLES"
        print(line); o.write(line + "\n")
PY
rm -rf $OUT
