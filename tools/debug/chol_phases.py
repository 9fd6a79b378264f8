"""Cycles of joint_chol_mfma_kernel per phase (chain 0's wave), car closed loop as shipped, Ns = 1024, H = 40: needs a library built with
GPMPC_PHASE_TIMERS=1 (python sampling_gpmpc_amd/csrc/build.py --force under that environment)."""
import ctypes as C, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import workloads as wl
Ns, H = 1024, 40
p = wl.closed_loop_params("params_car_residual", Ns, H, 1, 4)
p["common"]["use_cuda"] = True
p["agent"]["base_sample_generator"] = "counter"
agent = sg.Agent(p, sg.make_env(p))
x0 = np.asarray(p["env"]["start"], dtype=np.float64)[: agent.nx]
u_h, x_h = wl.synthetic_u_ff(agent.nu, H), np.tile(x0, (H, Ns))
raw = sg._lib.load()
out = (C.c_longlong * 8)()
agent.mpc_iteration(0)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for k in range(4):
        agent.train_hallucinated_dynGP(k)
        gv, _, _ = agent.dyn_fg_jacobians(agent.get_batch_x_hat(x_h, u_h), k)
        torch.cuda.synchronize()
        raw.gpmpc_debug_read_joint_chol_phases(out)
        v = list(out)[:5]
        print(f"k={k}: updates {v[1]} | diagonal tiles {v[2]} | panel + next loads {v[3]} | stores {v[4]} | total {sum(v)} cycles", flush=True)
        mean_next = gv[:, :, :, 0].mean(axis=0).T
        x_h = np.tile(np.vstack([x0[None, :], mean_next[:-1]]), (1, Ns))
