#!/bin/bash
# Run ON the GPU box: PMC passes of the mode-J joint draws as shipped (eigh root), mean per launch by kernel.
set -u
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_eigh_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/tools/bench_joint.py --car-only"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o pmc -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/pmc_write.log 2>&1
python3 - <<PY
import csv, collections, glob
for sub in ["pmc_sq","pmc_sq2","pmc_fetch","pmc_write"]:
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % sub):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "joint" in r["Kernel_Name"]:
                d[(r["Counter_Name"], r["Kernel_Name"][12:46])].append(float(r["Counter_Value"]))
        for k, v in sorted(d.items()):
            print(k, len(v), "mean %.4g max %.4g" % (sum(v)/len(v), max(v)))
PY
