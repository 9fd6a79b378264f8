"""Per-phase shader-cycle totals of block 0 / wave 0 for the BASELINE configs[1] rollout (debug aid)."""
import sys, ctypes as C, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import _lib
from sampling_gpmpc_amd.rollout import RolloutRunner
from tests.helpers import fs_params, synthetic_u_ff
pname = sys.argv[1] if len(sys.argv) > 1 else "params_pendulum1D_samples"
Ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H = int(sys.argv[3]) if len(sys.argv) > 3 else 30
p = fs_params(pname, Ns, H, nograd=False, beta=(3.0 if "car" in pname else None)); p["common"]["use_cuda"] = True; p["agent"]["base_sample_generator"] = "vectorized"
agent = sg.Agent(p, sg.make_env(p)); u_ff = synthetic_u_ff(agent.nu, H); erv = agent.epistimic_random_vector; per = Ns * agent.g_ny * 3
r = RolloutRunner(agent, u_ff, erv.reshape(-1)[per:], erv.shape[1] * per, H, _lib.MODE_RECONDITIONED, False)
for _ in range(3): r.launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): r.launch()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"{pname} Ns={Ns} H={H}: {ms*1e3:.1f} us per rollout -> {Ns*H/ms*1e3/1e6:.1f} M traj-steps/s")
lib = _lib.load(); out = (C.c_longlong * 16)()
fn = lib.gpmpc_debug_read_fast_phases if os.environ.get("GPMPC_DISABLE_FAST_ROLLOUT") != "1" else lib.gpmpc_debug_read_phases
fn(out)
names = ["kr", "vr", "rhs", "subst", "reduce", "sample", "append", "step"]
tot = sum(out[:8]); print("total cycles", tot, "=> us at 2.4GHz", tot / 2400)
for n, v in zip(names, out[:8]): print(f"{n:8s} {v:10d} {100*v/max(tot,1):5.1f}%")
