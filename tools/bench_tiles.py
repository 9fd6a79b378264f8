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
for (pn, ns, h) in CASES:
    bc.run(pn, ns, h, False, 20)
    print("   path", _lib.load().gpmpc_debug_last_rollout_path(), flush=True)
'''
cases = [("params_car_residual_fs", 4096, 40), ("params_pendulum1D_samples", 4096, 30), ("params_pendulum1D_samples", 16384, 30),
         ("params_car_residual_fs", 1024, 40), ("params_pendulum1D_samples", 1024, 30), ("params_car_residual_fs", 4096, 20)]
for mode in ("1", "0"):
    env = dict(os.environ, GPMPC_ROLLOUT_TILES=mode)
    print("== GPMPC_ROLLOUT_TILES=%s" % mode, flush=True)
    src = code.replace("CASES", repr(cases)) % (HERE, HERE)
    subprocess.run([sys.executable, "-c", src], env=env)
