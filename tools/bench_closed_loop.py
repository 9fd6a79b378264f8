"""One MPC step's worth of SQP iterations on the per-GPU shard of BASELINE configs[4] (car_residual, Ns=8192 over 8
GPUs = 1024 per GPU, H=40), timed end to end through the Agent surface exactly as src/solver.py:84-131 drives it:

    train_hallucinated_dynGP(k) -> get_batch_x_hat(x_h, u_h) -> dyn_fg_jacobians (joint draw + Jacobian assembly +
    D2H of the three arrays) -> pack_p_lin (stage parameter vectors, packed on the device + D2H)

acados is not installable here, so the solver's update of the linearisation points is replaced by the deterministic
surrogate SURVEY.md §8d names: x_h of iteration k+1 := the sample mean of iteration k's gp_val (shifted one stage),
inputs stay at the nominal sequence.  `--ns`, `--horizon`, `--iters` change the size.
"""
import argparse, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from tests.helpers import load_params


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--params", default="params_car_residual")
    ap.add_argument("--ns", type=int, default=1024)
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--mpc-steps", type=int, default=4)
    ap.add_argument("--jitter", type=float, default=None, help="override Dyn_gp_jitter (default: as shipped, 1e-20 -> eigh root)")
    a = ap.parse_args()
    p = load_params(a.params)
    p["common"]["use_cuda"] = True
    p["agent"]["num_dyn_samples"], p["optimizer"]["H"] = a.ns, a.horizon
    p["agent"]["true_dyn_as_sample"] = False
    p["agent"]["base_sample_generator"] = "vectorized"
    p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = 1, a.iters
    if a.jitter is not None:
        p["agent"]["Dyn_gp_jitter"] = a.jitter
    torch.manual_seed(3)
    agent = sg.Agent(p, sg.make_env(p))
    Ns, H, nx, nu = a.ns, a.horizon, agent.nx, agent.nu
    x0 = np.asarray(p["env"]["start"], dtype=np.float64)[:nx]
    u_h = np.zeros((H, nu)); u_h[:, 0] = 0.05 * np.sin(2 * np.pi * np.arange(H) / H)
    xg, w = np.zeros(H), np.ones(H)
    print(f"{a.params}: Ns={Ns} H={H} ({Ns * agent.g_ny} chains), {a.iters} SQP iterations; times in ms")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for mpc_step in range(a.mpc_steps):   # step 0 pays the one-off costs (workspace / tensor allocations)
            agent.mpc_iteration(0)
            x_h = np.tile(x0, (H, Ns))                           # (H, Ns*nx): every sample starts from the nominal point
            print(f"MPC step {mpc_step}" + ("  (cold: allocations)" if mpc_step == 0 else "  (steady state)"))
            print(f"{'k':>2} {'n_o':>5} {'train':>7} {'x_hat':>7} {'fg_jac':>8} {'p_lin':>7} {'total':>8}   traj-steps/s")
            tot = 0.0
            for k in range(a.iters):
                torch.cuda.synchronize(); t = [time.perf_counter()]
                agent.train_hallucinated_dynGP(k); torch.cuda.synchronize(); t.append(time.perf_counter())
                bx = agent.get_batch_x_hat(x_h, u_h); torch.cuda.synchronize(); t.append(time.perf_counter())
                gp_val, y_grad, u_grad = agent.dyn_fg_jacobians(bx, k); t.append(time.perf_counter())
                p_lin = agent.pack_p_lin(x_h, u_h, xg, w); t.append(time.perf_counter())
                d = np.diff(t) * 1e3
                n_o = agent.model_i.plan.n_r + agent.model_i.n_h * 3
                tot += d.sum()
                print(f"{k:2d} {n_o:5d} {d[0]:7.2f} {d[1]:7.2f} {d[2]:8.2f} {d[3]:7.2f} {d.sum():8.2f}   {Ns * H / d.sum() * 1e3:.3e}"
                      f"   finite={np.isfinite(gp_val).all() and np.isfinite(p_lin).all()}", flush=True)
                mean_next = gp_val[:, :, :, 0].mean(axis=0).T    # (H, nx): sample mean of f + B_d g at every stage
                x_new = np.vstack([x0[None, :], mean_next[:-1]]) # stage j+1 starts where the mean prediction of j ends
                x_h = np.tile(x_new, (1, Ns))
            print(f"  {a.iters} SQP iterations: {tot:.1f} ms of GP work per GPU")

if __name__ == "__main__":
    main()
