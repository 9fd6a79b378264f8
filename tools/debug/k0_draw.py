"""The k = 0 joint draw of the car closed loop as shipped (Ns=1024, H=40), repeated: for kernel traces / A-B runs."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import workloads as wl
Ns, H = 1024, 40
p = wl.closed_loop_params("params_car_residual", Ns, H, 2, 4)
p["common"]["use_cuda"] = True
p["agent"]["base_sample_generator"] = "counter"
agent = sg.Agent(p, sg.make_env(p))
x0 = np.asarray(p["env"]["start"], dtype=np.float64)[: agent.nx]
u_h = wl.synthetic_u_ff(agent.nu, H)
x_h = np.tile(x0, (H, Ns))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    agent.mpc_iteration(0)
    agent.train_hallucinated_dynGP(0)
    bx = agent.get_batch_x_hat(x_h, u_h)
    g_xu = agent.env_model.get_g_xu_hat(bx).contiguous()
    z = agent.epistimic_random_vector[0][0]
    for _ in range(60):
        agent.sample_gp(g_xu, base_samples=z)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    best = 1e9
    for _ in range(10):
        ev[0].record(); agent.sample_gp(g_xu, base_samples=z); ev[1].record(); torch.cuda.synchronize()
        best = min(best, ev[0].elapsed_time(ev[1]))
    print(f"k=0 draw best of 10: {best:.3f} ms  ABANDON={os.environ.get('GPMPC_JOINT_ABANDON')}")
