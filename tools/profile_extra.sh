#!/bin/bash
# kernel traces (no counters) of the non-bench workloads: cfg3 / cfg4 rollouts and the mode-J SQP iterations
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_extra_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/configs -o t -- python3 $ROOT/tools/bench_configs.py > $OUT/configs.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/joint -o t -- python3 $ROOT/tools/bench_joint.py > $OUT/joint.log 2>&1
grep -v amdgpu $OUT/configs.log | tail -5
grep -v amdgpu $OUT/joint.log | tail -7
