"""LDS-resident vs HBM/L2-resident factor of the tuned rollout kernel over the horizon (GPMPC_FORCE_GLOBAL_FACTOR=1
selects the latter); decides the heuristic in rollout_fast.hip:fast_plan."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc
bc.run("params_pendulum1D_samples", 1024, 30, False, 10)
for h in (24, 30, 31, 36, 43):
    bc.run("params_pendulum1D_samples", 4096, h, False, 10)
