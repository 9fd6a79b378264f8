#!/bin/bash
# Run ON the GPU box: kernel trace + PMC passes of the k = 3 joint draw of the car closed loop (configs[4] shard: Ns = 1024,
# H = 40, 360 hallucinated slots to condition on), once with the VALU path pinned ("before") and once with the matrix-pipe path
# ("after").  Every pass is its own process (FETCH_SIZE and WRITE_SIZE alone, as MI355X_MICROARCH.md prescribes); the numbers are
# PER DRAW: the sum over the kernels of one draw, mean of the run's last five draws (tools/bench_joint.py --only-k 3).
#   tools/profile_joint_r5.sh [tag]
set -u
TAG=${1:-r5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_joint_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in valu mfma; do
  ARGS="$ROOT/tools/bench_joint.py --car-only --only-k 3 --path $P"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${P}_trace -o t -- python3 $ARGS > $OUT/${P}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${P}_fetch -o pmc -- python3 $ARGS > $OUT/${P}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${P}_write -o pmc -- python3 $ARGS > $OUT/${P}_write.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/${P}_sq -o pmc -- python3 $ARGS > $OUT/${P}_sq.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/${P}_sq2 -o pmc -- python3 $ARGS > $OUT/${P}_sq2.log 2>&1
  grep "k=3" $OUT/${P}_trace.log
done
python3 $ROOT/tools/summarise_joint_r5.py $OUT | tee $OUT/summary.txt
