"""Wall clock of one SQP iteration through the FUSED call of round 6 (Agent.sqp_linearisation: one upload, gpmpc_build_x_hat, the
joint draw, gpmpc_assemble_jacobians_plin, one download of p_lin) on the car closed loop as shipped (Ns = 1024, H = 40), against the draw
alone (HIP events around get_batch_gp_sensitivities inside the call): what of an iteration is NOT the draw."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import workloads as wl

Ns, H, iters = 1024, 40, 4
p = wl.closed_loop_params("params_car_residual", Ns, H, 3, iters)
p["common"]["use_cuda"] = True
p["agent"]["base_sample_generator"] = "counter"


def make():
    return sg.Agent(p, sg.make_env(p))


warm = make()                                          # process-wide first-use costs go to a throw-away Agent
warm.mpc_iteration(0)
x0 = np.asarray(p["env"]["start"], dtype=np.float64)[: warm.nx]
u_h = wl.synthetic_u_ff(warm.nu, H)
xg, w = np.zeros(H), np.zeros(H)
for kw in range(2):
    warm.sqp_linearisation(np.tile(x0, (H, Ns)), u_h, kw, xg, w)
del warm
agent = make()
x_h = np.tile(x0, (H, Ns))
sync = torch.cuda.synchronize
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
inner = agent.get_batch_gp_sensitivities


def timed_draw(xu, k):
    e0.record()
    y = inner(xu, k)
    e1.record()
    return y


agent.get_batch_gp_sensitivities = timed_draw
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for step in range(3):
        agent.mpc_iteration(step)
        for k in range(iters):
            sync()
            t0 = time.perf_counter()
            pl = agent.sqp_linearisation(x_h, u_h, k, xg, w)
            sync()
            wall = (time.perf_counter() - t0) * 1e3
            draw = e0.elapsed_time(e1)
            gv = agent._last_device_jacobians[0]
            mean_next = gv[:, :, :, 0].mean(dim=0).T.cpu().numpy()
            x_h = np.tile(np.vstack([x0[None, :], mean_next[:-1]]), (1, Ns))
            print(f"step {step} k={k}: iteration {wall:7.2f} ms  draw + append (HIP events) {draw:7.2f}  everything else {wall - draw:6.2f}  "
                  f"(p_lin {pl.nbytes / 1e6:.1f} MB to the host)", flush=True)
