"""Does a joint draw read workspace it has not written?  The closed loop's draws (car as shipped, Ns = 64) with the joint workspace
filled with zeros and with NaNs in front of every call: the outputs must be finite and bit-equal."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from tests.helpers import load_params
Ns, H, iters = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 40, 5
p = load_params("params_car_residual")
p["common"]["use_cuda"] = True
p["agent"]["num_dyn_samples"], p["optimizer"]["H"] = Ns, H
p["agent"]["true_dyn_as_sample"] = False
p["agent"]["base_sample_generator"] = "vectorized"
p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = 1, iters
out = {}
for fill in (0.0, float("nan")):
    torch.manual_seed(11)
    agent = sg.Agent(p, sg.make_env(p))
    x0 = np.array(p["env"]["start"], dtype=np.float64)
    u_h = np.zeros((H, 2)); u_h[:, 0] = 0.05 * np.sin(2 * np.pi * np.arange(H) / H)
    x_h = np.tile(x0, (H, Ns))
    res = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k in range(iters):
            agent.train_hallucinated_dynGP(k)
            ws = agent._ws_cache.get("joint")
            if ws is not None:
                ws.fill_(fill)
            fcache = agent._ws_cache.get("joint_factor_cache")
            gp_val, y_grad, u_grad = agent.dyn_fg_jacobians(agent.get_batch_x_hat(x_h, u_h), k)
            res.append((gp_val.copy(), y_grad.copy()))
            mean_next = gp_val[:, :, :, 0].mean(axis=0).T
            x_h = np.tile(np.vstack([x0[None, :], mean_next[:-1]]), (1, Ns))
    out[str(fill)] = res
for k in range(iters):
    a, b = out["0.0"][k], out["nan"][k]
    print(f"k={k}: finite with NaN-filled workspace: {bool(np.isfinite(b[0]).all() and np.isfinite(b[1]).all())}; bit-equal to the zero-filled run: {bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]))}; max |diff| {np.nanmax(np.abs(a[0] - b[0])):.2e}")
