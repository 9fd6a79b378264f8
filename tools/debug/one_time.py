"""rollout_one_kernel against rollout_fast_kernel at the headline shape and around it, sustained clocks; phase cycles of
wave 0 when the library was built with GPMPC_PHASE_TIMERS=1."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
TOOLS = os.path.dirname(HERE)
code = r'''
import sys, os, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, os.path.dirname(%r))
sys.argv = ["x", "--sustained"]
import bench_configs as bc
from sampling_gpmpc_amd import _lib
for (ns, h) in CASES:
    bc.run("params_pendulum1D_samples", ns, h, False, 20)
    print("   path", _lib.load().gpmpc_debug_last_rollout_path(), flush=True)
    if os.environ.get("GPMPC_PHASE_TIMERS") == "1" and _lib.load().gpmpc_debug_last_rollout_path() == 4:
        raw = C.CDLL(_lib.LIB_PATH); out = (C.c_longlong * 16)(); raw.gpmpc_debug_read_one_phases(out)
        names = ["entries+vr+lds", "-", "solve", "gram+extract", "sample", "append sets", "append diag", "state"]
        tot = sum(out[:8])
        print("   total cycles", tot, "per step", tot // h)
        for n, v in zip(names, out[:8]): print(f"   {n:16s} {v:9d} {v // h:7d}/step {100.0 * v / max(tot, 1):5.1f}%%")
        print(f"   prologue (kernel entry -> step loop) {out[8]} cycles, whole kernel of wave 0 {out[9]} cycles (the step loop {tot})")
'''
cases = [(1024, 30), (1024, 15), (2048, 30), (256, 30)]
for mode in ("1", "0"):
    env = dict(os.environ, GPMPC_ROLLOUT_ONE=mode)
    print("== GPMPC_ROLLOUT_ONE=%s" % mode, flush=True)
    subprocess.run([sys.executable, "-c", code.replace("CASES", repr(cases)) % (TOOLS, TOOLS)], env=env)
