#!/bin/bash
# Run ON the GPU box: HIP API + kernel trace of the closed loop (tools/bench_closed_loop.py, configs[4] shard as shipped),
# condensed by tools/summarise_closed_loop_trace.py into profiles/<tag>_closed_loop_trace.md: which API call owns the gaps.
set -u
TAG=${1:-r3}
STEPS=${2:-8}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_cl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# untraced reference run first (wall clock per SQP iteration without the tracer attached)
python3 $ROOT/tools/bench_closed_loop.py --mpc-steps $STEPS > $OUT/untraced.log 2>&1
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $OUT/trace -o cl -- python3 $ROOT/tools/bench_closed_loop.py --mpc-steps $STEPS > $OUT/traced.log 2>&1
cd $ROOT
python3 tools/summarise_closed_loop_trace.py $OUT $TAG
rm -f $OUT/trace/*hip_api_trace.csv $OUT/trace/*kernel_trace.csv     # tens of MB; the summary and the *_stats.csv stay
