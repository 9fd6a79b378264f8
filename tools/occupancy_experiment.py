"""Experiment: does a second wave per SIMD come for free for the tuned rollout kernel?

Needs a library built with half the register budget so that two 4-sample workgroups can share a CU:
    GPMPC_EXTRA_DEFS="-DGPMPC_FAST_MAXTHREADS=512" python sampling_gpmpc_amd/csrc/build.py --force
(the kernel then spills, so absolute times are worse than the shipped build; the point is the ratio) and a horizon short
enough for 8 samples' factors in LDS (H <= 15).  Measured on MI355X: Ns=1024 (one wave per SIMD) 305 / 386 us at
H = 12 / 15, Ns=2048 (two waves per SIMD) 394 / 499 us: 2x the work in 1.29x the time.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc
for H in (12, 15):
    bc.run("params_pendulum1D_samples", 1024, H, False, 20)
    bc.run("params_pendulum1D_samples", 2048, H, False, 20)
