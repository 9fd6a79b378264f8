"""MPC steps of the closed loop on the per-GPU shard of BASELINE configs[4] (car_residual AS SHIPPED, Ns = 8192 / 8 GPUs
= 1024 per GPU, H = 40), timed end to end through the package's driver (sampling_gpmpc_amd.closed_loop: reference
src/DEMPC.py:39-80 around the SQP loop of src/solver.py:56-131 with the surrogate QP step of SURVEY.md 8d):

    train_hallucinated_dynGP(k) -> get_batch_x_hat[_u_diff] -> dyn_fg_jacobians (joint draw + Jacobian assembly + D2H of
    the three arrays) -> pack_p_lin (stage parameter vectors, packed on the device + D2H) -> plant step

`--ns`, `--horizon`, `--iters`, `--mpc-steps` change the size; `--jitter` overrides Dyn_gp_jitter (default: as shipped,
1e-20 -> the eigendecomposition root on every draw).
"""
import argparse, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd.closed_loop import ClosedLoop, SurrogateSolver
from sampling_gpmpc_amd.workloads import closed_loop_params


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--params", default="params_car_residual")
    ap.add_argument("--ns", type=int, default=1024)
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--mpc-steps", type=int, default=4)
    ap.add_argument("--run-steps", type=int, default=None, help="MPC steps actually run (default: all; the base samples are drawn for --mpc-steps either way)")
    ap.add_argument("--jitter", type=float, default=None, help="override Dyn_gp_jitter (default: as shipped, 1e-20 -> eigh root)")
    ap.add_argument("--block", action="store_true", help="let the runtime wait for completions (GPMPC_HOST_WAIT=block) instead of polling")
    a = ap.parse_args()
    if a.block:
        os.environ["GPMPC_HOST_WAIT"] = "block"
    p = closed_loop_params(a.params, a.ns, a.horizon, a.mpc_steps, a.iters)
    p["common"]["use_cuda"] = True
    p["agent"]["base_sample_generator"] = "counter"
    p["optimizer"]["SEMPC"]["tol_nlp"] = 0.0
    if a.jitter is not None:
        p["agent"]["Dyn_gp_jitter"] = a.jitter
    agent = sg.Agent(p, sg.make_env(p))
    agent.update_current_state(np.asarray(p["env"]["start"], dtype=np.float64))
    loop = ClosedLoop(p, agent, SurrogateSolver(p))
    print(f"host waits: {'runtime (interrupt)' if a.block else 'polled'}; HSA_ENABLE_INTERRUPT={os.environ.get('HSA_ENABLE_INTERRUPT', 'unset')}")
    print(f"{a.params}: Ns={a.ns} H={a.horizon} ({a.ns * agent.g_ny} chains), {a.iters} SQP iterations per MPC step; "
          f"jitter {p['agent']['Dyn_gp_jitter']:g}; GP side per SQP iteration in ms (train + x_hat + fg_jac + p_lin)")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for step in range(a.mpc_steps if a.run_steps is None else min(a.run_steps, a.mpc_steps)):
            agent.mpc_iteration(step)
            x_curr = np.asarray(agent.current_state[: agent.nx], dtype=np.float64)
            st = np.array(x_curr.tolist() * a.ns)
            torch.cuda.synchronize()
            loop.one_step_planner(st)
            ms = loop.solver.gp_ms
            eigh = bool((agent.model_i_call.last_info & sg._lib.INFO_ROOT_EIGH).all().item())
            print(f"MPC step {step}{' (cold: allocations)' if step == 0 else ''}: " + "  ".join(f"k={k}: {t:7.2f}" for k, t in enumerate(ms))
                  + f"   total {sum(ms):7.1f} ms   eigh root={eigh}   finite={bool(np.isfinite(loop.solver.p_lin).all())}", flush=True)


if __name__ == "__main__":
    main()
