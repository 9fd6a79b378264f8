"""cProfile of the facade's Python in steady-state SQP iterations (car closed loop as shipped, Ns = 1024, H = 40): where the host time of
Agent.sqp_linearisation goes."""
import cProfile, os, pstats, sys, warnings, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import workloads as wl
Ns, H, iters = 1024, 40, 4
p = wl.closed_loop_params("params_car_residual", Ns, H, 12, iters)
p["common"]["use_cuda"] = True
p["agent"]["base_sample_generator"] = "counter"
agent = sg.Agent(p, sg.make_env(p))
x0 = np.asarray(p["env"]["start"], dtype=np.float64)[: agent.nx]
u_h, x_h = wl.synthetic_u_ff(agent.nu, H), np.tile(x0, (H, Ns))
xg, w = np.zeros(H), np.zeros(H)
pr = cProfile.Profile()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for step in range(10):
        agent.mpc_iteration(step)
        for k in range(iters):
            if step == 2 and k == 0:
                torch.cuda.synchronize()
                pr.enable()
            agent.sqp_linearisation(x_h, u_h, k, xg, w)
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
