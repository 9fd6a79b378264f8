#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_jm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/tools/debug/joint_paths.py --ns 1024"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -o pmc -- python3 $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/b -o pmc -- python3 $ARGS > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/c -o pmc -- python3 $ARGS > $OUT/c.log 2>&1
python3 - <<PY
import csv, collections, glob
for sub in "abc":
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % sub):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "joint_test_mfma" in r["Kernel_Name"]:
                d[(r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
        for k, v in sorted(d.items()):
            print(k, len(v), "mean %.4g max %.4g" % (sum(v)/len(v), max(v)))
PY
tail -3 $OUT/c.log
