#!/bin/bash
# run on the GPU box: time the bench rollout (sustained clocks, bench.py's own timed region) with alternative
# code-generation flags for rollout_fast.hip
for f in "$@"; do
  GPMPC_FAST_FLAGS="$f" python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  printf "%-100s " "[$f]"
  python bench.py --cpu-sample 0 --reach-ns 0 --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.5f ms/step  kernel %.5f ms  frac %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
