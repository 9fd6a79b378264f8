#!/bin/bash
# run on the GPU box: time the configs[2] car rollout (tools/bench_configs.py --car --sustained) with alternative
# code-generation flags for rollout_fast.hip
for f in "$@"; do
  GPMPC_FAST_FLAGS="$f" python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  printf "%-100s " "[$f]"
  python tools/bench_configs.py --car --sustained 2>/dev/null | grep "Ns=" | cut -c50-110
done
