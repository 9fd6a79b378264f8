#!/bin/bash
# build + time the workspace-factor variant of the tuned rollout kernel for several prefetch-ring depths (GPU box)
for rg in "$@"; do
  echo "=== GPMPC_FAST_RING_GLOBAL=$rg"
  GPMPC_EXTRA_DEFS="-DGPMPC_FAST_RING_GLOBAL=$rg" python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1 || { echo build failed; continue; }
  python tools/bench_configs.py 2>&1 | grep "mode=R" 
  python - <<'PY' 2>&1 | grep "H=43"
import sys, os
sys.path.insert(0, "tools")
import bench_configs as bc
bc.run("params_pendulum1D_samples", 4096, 43, False, 5)
PY
done
