#!/bin/bash
# Run ON the GPU box: is the closed loop's stall pattern the container's CPU quota?  Prints the cgroup's cpu.max and the
# throttling counters around runs of tools/bench_closed_loop.py with the host thread pools as they come and limited to one thread.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cg=/sys/fs/cgroup
echo "nproc $(nproc); cpu.max: $(cat $cg/cpu.max 2>/dev/null || echo n/a); cpuset: $(cat $cg/cpuset.cpus.effective 2>/dev/null || echo n/a)"
stat() { grep -E "nr_periods|nr_throttled|throttled_usec" $cg/cpu.stat 2>/dev/null | tr '\n' ' '; echo; }
cd /tmp
for v in default one-thread; do
  echo "=== $v"; echo -n "before: "; stat
  if [ $v = default ]; then python3 $ROOT/tools/bench_closed_loop.py --mpc-steps 6 2>&1 | grep -E "MPC step|host waits"
  else OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 python3 $ROOT/tools/bench_closed_loop.py --mpc-steps 6 2>&1 | grep -E "MPC step|host waits"; fi
  echo -n "after:  "; stat
done
