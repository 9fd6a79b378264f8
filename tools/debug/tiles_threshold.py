import os, subprocess, sys
HERE = "/root/repo/tools"
code = r'''
import sys, os
sys.path.insert(0, "/root/repo"); sys.argv = ["x", "--sustained"]; sys.path.insert(0, "/root/repo/tools")
import bench_configs as bc
from sampling_gpmpc_amd import _lib
for (pn, ns, h) in CASES:
    bc.run(pn, ns, h, False, 20)
'''
cases = [("params_car_residual_fs", 256, 40), ("params_car_residual_fs", 384, 40), ("params_car_residual_fs", 512, 40), ("params_car_residual_fs", 768, 40),
         ("params_pendulum1D_samples", 1536, 30), ("params_pendulum1D_samples", 2048, 30), ("params_pendulum1D_samples", 3072, 30)]
for mode in ("1", "0"):
    print("== GPMPC_ROLLOUT_TILES=%s" % mode, flush=True)
    subprocess.run([sys.executable, "-c", code.replace("CASES", repr(cases))], env=dict(os.environ, GPMPC_ROLLOUT_TILES=mode))
