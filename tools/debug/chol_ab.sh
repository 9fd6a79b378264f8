#!/bin/bash
# run on the GPU box: joint_chol_mfma_kernel - waves per SIMD of the eight-tile instance: parity (soaks of the matrix-pipe path), phase cycles, kernel trace
for f in "-DGPMPC_JC_OCC8=2" "-DGPMPC_JC_OCC8=1"; do
  echo "== [$f]"
  GPMPC_EXTRA_DEFS="$f" GPMPC_PHASE_TIMERS=1 python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1 || echo "timer build failed"
  GPMPC_EXTRA_DEFS="$f" GPMPC_PHASE_TIMERS=1 python tools/debug/chol_phases.py 2>&1 | grep "k=[23]"
  GPMPC_EXTRA_DEFS="$f" python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1
  GPMPC_EXTRA_DEFS="$f" python -m pytest tests/test_hip_joint_soak.py -x -q -m gpu -k "determinism or poisoning" 2>&1 | tail -1
  GPMPC_EXTRA_DEFS="$f" bash tools/debug/trace_closed_loop_draws.sh 2>&1 | grep chol | tail -3
done
python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1
