#!/bin/bash
# run on the GPU box: the L2-prefetch budget of rollout_tiles_kernel (GPMPC_TILES_PF, live KB per wave and step) swept, tiled kernel only
#   tools/tiles_pf_sweep.sh 0 16 32 48 64 96
for pf in "$@"; do
  f="-DGPMPC_TILES_PF=$pf"
  GPMPC_EXTRA_DEFS="$f" python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  echo "== [$f]"
  GPMPC_EXTRA_DEFS="$f" python tools/bench_tiles.py --quick --tiles-only 2>/dev/null | grep "Ns=" | cut -c1-120
done
python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1
