#!/bin/bash
# run on the GPU box: kernel time of the bench rollout with individual phases compiled out (results are wrong then;
# the difference to the full kernel is that phase's cost).  usage: tools/ablate.sh "" -DGPMPC_ABLATE_SUBST ...
mkdir -p gpurun_out
for d in "$@"; do
  GPMPC_EXTRA_DEFS="$d" python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1 || { echo "build failed: $d"; continue; }
  printf "%-70s " "[$d]"; python tools/phase_cycles.py ${ABLATE_ARGS} 2>/dev/null | grep "us per rollout"
done | tee gpurun_out/ablate.log
