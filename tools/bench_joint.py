"""Time mode J (joint-horizon draw per SQP iteration, BASELINE config 5 shape) through Agent.dyn_fg_jacobians.

    python tools/bench_joint.py [--car-only] [--cholesky-branch] [--sustained] [--path valu|mfma] [--only-k K]

--path pins gpmpc_joint_sample's path for the whole run (default: the dispatcher's choice); --only-k times the draw of SQP
iteration K only (the counter passes of tools/profile_joint_r5.sh: the last five draws of the run are that iteration's).
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings
import torch, numpy as np
import sampling_gpmpc_amd as sg
from tests.helpers import load_params

ONLY_K = int(sys.argv[sys.argv.index("--only-k") + 1]) if "--only-k" in sys.argv else None
PATH = sys.argv[sys.argv.index("--path") + 1] if "--path" in sys.argv else "auto"


def run(pname, Ns, H, iters, jitter=None):
    p = load_params(pname)
    p["common"]["use_cuda"] = True
    p["agent"]["num_dyn_samples"], p["optimizer"]["H"] = Ns, H
    p["agent"]["true_dyn_as_sample"] = False
    p["agent"]["base_sample_generator"] = "vectorized"
    p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = 1, iters
    if jitter is not None:
        p["agent"]["Dyn_gp_jitter"] = jitter
    torch.manual_seed(3)
    agent = sg.Agent(p, sg.make_env(p))
    sg._lib.load().gpmpc_joint_pin_path({"auto": sg._lib.JOINT_AUTO, "valu": sg._lib.JOINT_VALU, "mfma": sg._lib.JOINT_MFMA}[PATH])
    nx, nu = agent.nx, agent.nu
    g = torch.Generator().manual_seed(5)
    x0 = np.array(p["env"]["start"], dtype=np.float64)
    agent.mpc_iteration(0)
    for it in range(iters):
        x_h = np.tile(x0, (H, Ns)) + 0.05 * torch.randn(H, Ns * nx, generator=g, dtype=torch.float64).numpy() + 0.02 * np.arange(H)[:, None]
        u_h = 0.3 * torch.randn(H, Ns, nu, generator=g, dtype=torch.float64).numpy()
        agent.train_hallucinated_dynGP(it)
        bx = agent.get_batch_x_hat_u_diff(x_h, u_h)
        # time the draw itself (joint kernel + post-processing) on repeats that leave the Agent state alone ...
        g_xu = agent.env_model.get_g_xu_hat(bx).contiguous()
        z = agent.epistimic_random_vector[agent.mpc_iter][it]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        best = float("inf")
        # every timed draw starts from the factor-cache state the closed loop has at this iteration (rows of the slots
        # the previous iteration's draw conditioned on), not from the state the previous timed draw left
        cache = agent._ws_cache.get("joint_factor_cache")
        held = cache.n_valid if cache is not None else 0

        def rewind():
            c = agent._ws_cache.get("joint_factor_cache")
            if c is not None:
                c.rewind(held)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if "--sustained" in sys.argv:            # bring the GPU to its sustained clocks first (tools/clock_check.py)
                t_end = time.perf_counter() + 0.5
                while time.perf_counter() < t_end:
                    rewind(); agent.sample_gp(g_xu, base_samples=z)
                torch.cuda.synchronize()
            for rep in range(4 if ONLY_K in (None, it) else 0):
                rewind()
                ev[0].record(); agent.sample_gp(g_xu, base_samples=z); ev[1].record(); torch.cuda.synchronize()
                if rep:
                    best = min(best, ev[0].elapsed_time(ev[1]))
            rewind()
            # ... then the real call, which also appends the draw to the hallucinated data set
            y = agent.get_batch_gp_sensitivities(bx, it)
        t0, t1 = 0.0, best * 1e-3
        n_h = agent.model_i.n_h
        lvl = int(((agent.model_i_call.last_info >> 1) & 7).max().item())
        eigh = bool((agent.model_i_call.last_info & sg._lib.INFO_ROOT_EIGH).all().item())
        if os.environ.get("GPMPC_PHASE_TIMERS") == "1":
            import ctypes as C
            out = (C.c_longlong * 16)(); sg._lib.load().gpmpc_debug_read_joint_phases(out)
            names = ["realcols", "init", "update", "factor", "solve", "mean+S", "root", "sample"]
            print("   phases(cycles):", {n: out[i] for i, n in enumerate(names)})
            out = (C.c_longlong * 8)(); sg._lib.load().gpmpc_debug_read_eigh_phases(out)
            names = ["pivchol", "gram", "jacobi", "reverse", "sample", "ticks_100MHz", "sweeps", "rank"]
            cyc = sum(out[i] for i in range(5))
            print("   eigh phases(cycles, chain 0 or 1000):", {n: out[i] for i, n in enumerate(names)},
                  f"=> {cyc} cycles in {out[5] * 10} ns: shader clock {cyc / max(out[5], 1) / 10:.2f} GHz")
        print(f"{pname:26s} Ns={Ns} H={H} k={it}: n_o={agent.model_i.plan.n_r + n_h*3:4d} m*T={H*3} cached rows {agent.model_i_call.n_cached_rows}: sample_gp {1e3*(t1-t0):8.2f} ms (best of 3) "
              f"({Ns*H/(t1-t0)/1e6:7.2f} M traj-steps/s), path {sg._lib.load().gpmpc_joint_last_path()}, max jitter level {lvl}, eigh root={eigh}, finite={bool(torch.isfinite(y).all())}", flush=True)

if __name__ == "__main__":
    if "--car-only" not in sys.argv:
        run("params_pendulum1D_samples", 1024, 30, 2)
    run("params_car_residual", 1024, 40, 4)                     # as shipped: Dyn_gp_jitter 1e-20 -> eigendecomposition root
    if "--cholesky-branch" in sys.argv:
        run("params_car_residual", 1024, 40, 4, jitter=1e-9)    # comparison: the car kept on the Cholesky branch
