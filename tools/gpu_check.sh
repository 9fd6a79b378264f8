#!/bin/bash
# run on the GPU box: rollout parity tests, bench line, per-phase cycle counters of the bench kernel (debug build)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q ${GPU_CHECK_K:+-k "$GPU_CHECK_K"} > gpurun_out/test.log 2>&1
python bench.py --steps 50 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/bench_quick.json
if [ "$GPU_CHECK_PHASES" = "1" ]; then
  GPMPC_PHASE_TIMERS=1 python sampling_gpmpc_amd/csrc/build.py --force > /dev/null 2>&1
  python tools/phase_cycles.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/phases.log
fi
grep -E "passed|failed|error" gpurun_out/test.log | tail -3
python -c "import json;d=json.load(open('gpurun_out/bench_quick.json'));print('bench ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"
[ -f gpurun_out/phases.log ] && cat gpurun_out/phases.log
