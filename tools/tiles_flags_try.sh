#!/bin/bash
# run on the GPU box: rebuild rollout_tiles.hip with alternative code-generation flags and time the tiled kernel
for f in "$@"; do
  GPMPC_TILES_FLAGS="$f" python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  echo "== [$f]"
  GPMPC_TILES_FLAGS="$f" python tools/bench_tiles.py --quick --tiles-only 2>/dev/null | grep "Ns=" | cut -c1-120
done
python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1
