#!/bin/bash
# run on the GPU box: rebuild with alternative -D knobs for rollout_tiles.hip and time the tiled kernel (tools/bench_tiles.py --quick)
#   tools/tiles_try.sh "-DGPMPC_TILES_RC=32" "-DGPMPC_TILES_RC=48"
for f in "$@"; do
  GPMPC_EXTRA_DEFS="$f" python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  echo "== [$f]"
  GPMPC_EXTRA_DEFS="$f" python tools/bench_tiles.py --quick --tiles-only 2>/dev/null | grep "Ns=" | cut -c1-120
done
python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1
