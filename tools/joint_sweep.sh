#!/bin/bash
# build + time the joint kernel for a list of "-D..." knob sets (run on the GPU box: hipcc is there too)
for defs in "$@"; do
  echo "=== $defs"
  GPMPC_EXTRA_DEFS="$defs" python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1 || { echo build failed; continue; }
  python tools/bench_joint.py 2>&1 | grep "car_residual\|pendulum" | grep -v "car.*k=1\|pendulum.*k=0"
  if [ -n "$JOINT_TESTS" ]; then python -m pytest tests -x -q -m gpu -k "joint or jacobians or model_i" 2>&1 | grep -E "passed|failed|Error" | tail -2; fi
done
