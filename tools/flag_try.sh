#!/bin/bash
# run on the GPU box: time the bench rollout with alternative code-generation flags for rollout_fast.hip
for f in "$@"; do
  GPMPC_FAST_FLAGS="$f" python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  printf "%-90s " "[$f]"; python tools/phase_cycles.py ${ABLATE_ARGS} 2>/dev/null | grep "us per rollout"
done
