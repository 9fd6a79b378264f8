"""Per-phase shader-cycle totals of block 0 / wave 0 of the mode-I grid kernel (debug build: GPMPC_PHASE_TIMERS=1)."""
import sys, ctypes as C, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import _lib
from sampling_gpmpc_amd.rollout import RolloutRunner
from tests.helpers import fs_params, synthetic_u_ff
Ns = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
H = 40
p = fs_params("params_car_residual_fs", Ns, H, nograd=True); p["common"]["use_cuda"] = True; p["agent"]["base_sample_generator"] = "vectorized"
agent = sg.Agent(p, sg.make_env(p)); u_ff = synthetic_u_ff(agent.nu, H); erv = agent.epistimic_random_vector; per = Ns * agent.g_ny
r = RolloutRunner(agent, u_ff, erv.reshape(-1)[per:], erv.shape[1] * per, H, _lib.MODE_INDEPENDENT, True)
for _ in range(3): r.launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): r.launch()
e1.record(); torch.cuda.synchronize()
print(f"mode I Ns={Ns} H={H}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per rollout")
lib = _lib.load(); out = (C.c_longlong * 16)()
lib.gpmpc_debug_read_indep_phases(out)
names = ["head", "exps", "fma", "tail", "exchange", "flush+env"]
tot = sum(out[:6]); print("total cycles", tot, "per step", tot / H)
for n, v in zip(names, out[:6]): print(f"{n:10s} {v:10d} {v / H:9.0f}/step {100*v/max(tot,1):5.1f}%")
