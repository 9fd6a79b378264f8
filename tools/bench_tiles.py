"""Tiled throughput kernel (rollout_tiles.hip) against the one-chain-per-wave kernels at the same sizes, sustained clocks."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
code = r'''
import sys, os
sys.path.insert(0, os.path.dirname(%r))
sys.argv = ["x", "--sustained"]
sys.path.insert(0, %r)
import bench_configs as bc
from sampling_gpmpc_amd import _lib
for (pn, ns, h, g) in CASES:
    bc.run(pn, ns, h, False, 20, n_data_x=g)
    print("   path", _lib.load().gpmpc_debug_last_rollout_path(), flush=True)
'''
cases = [("params_car_residual_fs", 4096, 40, None), ("params_pendulum1D_samples", 4096, 30, None),
         ("params_pendulum1D_samples", 16384, 30, None), ("params_car_residual_fs", 1024, 40, None),
         ("params_pendulum1D_samples", 1024, 30, None), ("params_car_residual_fs", 4096, 20, None),
         # shapes without a tuned one-chain kernel (GPMPC_ROLLOUT_TILES=0: the generic kernel)
         ("params_car_residual_fs", 4096, 50, None), ("params_car_residual_fs", 4096, 40, 6),
         ("params_pendulum1D_samples", 4096, 30, 5), ("params_pendulum1D_samples", 4096, 50, None),
         ("params_car_residual_fs", 1024, 65, None)]
if "--quick" in sys.argv:
    cases = cases[:2] + cases[6:8]
for mode in (("1",) if "--tiles-only" in sys.argv else ("1", "0")):
    env = dict(os.environ, GPMPC_ROLLOUT_TILES=mode)
    print("== GPMPC_ROLLOUT_TILES=%s" % mode, flush=True)
    src = code.replace("CASES", repr(cases)) % (HERE, HERE)
    subprocess.run([sys.executable, "-c", src], env=env)
