"""Cycles of joint_real_mfma_kernel per phase (chain 0's wave) in its test use (the k = 0 draw of the first MPC step: no hallucinated slot),
car closed loop as shipped, Ns = 1024, H = 40; needs a library built with GPMPC_PHASE_TIMERS=1."""
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
    for rep in range(3):
        agent.train_hallucinated_dynGP(0)
        agent.dyn_fg_jacobians(agent.get_batch_x_hat(x_h, u_h), 0)
        torch.cuda.synchronize()
        raw.gpmpc_debug_read_joint_chol_phases(out)
        v = list(out)[:8]
        print(f"[LinvT requests + descriptors {v[5]} | points to LDS {v[6]} | pair pass {v[0]}] descriptors + pair tables {v[5] + v[6] + v[0]} | LinvT tiles {v[1]} | X (entries + products) {v[2]} | mean / X^T stores {v[3]} | S (entries, products, stores) {v[4]} | "
              f"total {sum(v[:8])} cycles", flush=True)
