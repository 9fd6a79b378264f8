"""Phase cycles of joint_kernel / joint_eigh_kernel on the CLOSED LOOP's linearisation points (configs[4] shard as shipped);
build with GPMPC_PHASE_TIMERS=1.  tools/bench_joint.py prints the same for scattered points (ranks ~50)."""
import ctypes as C, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd.closed_loop import ClosedLoop, SurrogateSolver
from sampling_gpmpc_amd.workloads import closed_loop_params

p = closed_loop_params("params_car_residual", 1024, 40, 2, 4)
p["common"]["use_cuda"] = True
p["agent"]["base_sample_generator"] = "counter"
p["optimizer"]["SEMPC"]["tol_nlp"] = 0.0
agent = sg.Agent(p, sg.make_env(p))
agent.update_current_state(np.asarray(p["env"]["start"], dtype=np.float64))
lib = sg._lib.load()
orig = agent.sample_gp


def traced(x, base_samples=None):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = orig(x, base_samples=base_samples)
    e1.record()
    torch.cuda.synchronize()
    j = (C.c_longlong * 16)(); lib.gpmpc_debug_read_joint_phases(j)
    e = (C.c_longlong * 8)(); lib.gpmpc_debug_read_eigh_phases(e)
    jn = ["realcols", "init", "update", "factor", "solve", "mean+S", "root", "sample"]
    en = ["pivchol", "gram", "jacobi", "reverse", "sample", "ticks_100MHz", "sweeps", "rank"]
    print(f"n_ho={agent.model_i.h_slots.numel():4d} draw {e0.elapsed_time(e1):6.2f} ms | root: {j[9]} attempts, {j[8]} column blocks | joint", {n: j[i] for i, n in enumerate(jn)},
          "| eigh", {n: e[i] for i, n in enumerate(en)}, flush=True)
    return y


agent.sample_gp = traced
loop = ClosedLoop(p, agent, SurrogateSolver(p))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for step in range(2):
        agent.mpc_iteration(step)
        st = np.array(np.asarray(agent.current_state[: agent.nx], dtype=np.float64).tolist() * 1024)
        loop.one_step_planner(st)
