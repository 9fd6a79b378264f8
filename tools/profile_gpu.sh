#!/bin/bash
# Run ON the GPU box (through gpurun): kernel trace + separate PMC passes for bench.py's workload.
# Usage: bash tools/profile_gpu.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the kernel trace is of bench.py's default command (its CPU leg skipped); the PMC passes replay a shorter run of the
# headline workload only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/bench.py --cpu-sample 0 > $OUT/trace.log 2>&1
ARGS="$ROOT/bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-extra --reach-ns 0 --prewarm 200"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o pmc -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1
find $OUT -name "*.csv" | head -40
