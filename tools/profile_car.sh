#!/bin/bash
# Run ON the GPU box: PMC passes of the config-3 rollout (car, mode R, Ns=4096, H=40), mean per launch.
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_car_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/tools/bench_configs.py --car"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -o pmc -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -o pmc -- python3 $ARGS > $OUT/pmc_tcc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_mfma -o pmc -- python3 $ARGS > $OUT/pmc_mfma.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum --output-format csv -d $OUT/pmc_tcp -o pmc -- python3 $ARGS > $OUT/pmc_tcp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/trace.log 2>&1
python3 - <<PY
import csv, collections, glob
for sub in ["pmc_sq","pmc_sq2","pmc_fetch","pmc_write","pmc_tcc","pmc_mfma","pmc_tcp"]:
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % sub):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rollout_fast" in r["Kernel_Name"] or "rollout_tiles" in r["Kernel_Name"]:
                d[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(d.items()):
            print(k, len(v), "mean %.4g" % (sum(v)/len(v)))
PY
grep -h "rollout_" $OUT/trace/*kernel_stats.csv 2>/dev/null | head -5
