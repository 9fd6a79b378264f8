#!/bin/bash
# Run ON the GPU box: the headline kernel (rollout_one.hip) rebuilt with code-alignment flags, bench.py --steps 200 twice each.
cd ${GRAFT_REPO_ROOT:-.}
for F in "-mllvm -disable-machine-licm" "-mllvm -disable-machine-licm -falign-loops=64" "-mllvm -disable-machine-licm -mllvm -align-all-blocks=5" "-mllvm -disable-machine-licm -mllvm -align-all-nofallthru-blocks=6" "-mllvm -disable-machine-licm -mllvm -amdgpu-s-branch-bits=16"; do
  GPMPC_ONE_FLAGS="$F" python sampling_gpmpc_amd/csrc/build.py > /tmp/b.log 2>&1 || { echo "build failed: $F"; tail -3 /tmp/b.log; continue; }
  echo "== $F"
  for i in 1 2; do python bench.py --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"; done
done
