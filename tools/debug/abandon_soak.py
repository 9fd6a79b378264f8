"""GPMPC_JOINT_ABANDON = 0 against 1 (forced on): the joint draws of the car closed loop as shipped must be BIT-identical - samples,
means, variances, info words - whatever the launch size (a chain that abandons its Cholesky attempts is redrawn by the eigh kernel).
One subprocess per setting (the library reads the knob once)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
code = r'''
import sys, os, warnings
sys.path.insert(0, %r)
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import workloads as wl
out = {}
for Ns in (8, 300, 1024):
    H, iters = 40, 3
    p = wl.closed_loop_params("params_car_residual", Ns, H, 2, iters)
    p["common"]["use_cuda"] = True
    p["agent"]["base_sample_generator"] = "counter"
    agent = sg.Agent(p, sg.make_env(p))
    x0 = np.asarray(p["env"]["start"], dtype=np.float64)[: agent.nx]
    u_h = wl.synthetic_u_ff(agent.nu, H)
    x_h = np.tile(x0, (H, Ns))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        agent.mpc_iteration(0)
        for k in range(iters):
            agent.train_hallucinated_dynGP(k)
            gv, yg, ug = agent.dyn_fg_jacobians(agent.get_batch_x_hat(x_h, u_h), k)
            out[f"y_{Ns}_{k}"] = agent.model_i_samples.cpu().numpy()
            out[f"info_{Ns}_{k}"] = agent.model_i_call.last_info.cpu().numpy()
            out[f"gv_{Ns}_{k}"] = gv
            mean_next = gv[:, :, :, 0].mean(axis=0).T
            x_h = np.tile(np.vstack([x0[None, :], mean_next[:-1]]), (1, Ns))
np.savez(sys.argv[1], **out)
'''
res = {}
for mode in ("0", "1"):
    f = f"/tmp/abandon_{mode}.npz"
    subprocess.run([sys.executable, "-c", code % ROOT, f], env=dict(os.environ, GPMPC_JOINT_ABANDON=mode), check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import numpy as np
    res[mode] = dict(np.load(f))
bad = 0
for k in sorted(res["0"]):
    a, b = res["0"][k], res["1"][k]
    same = a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
    if k.startswith("info"):
        print(f"{k:14s} abandon off / on identical: {same}   words 0x{int(a.max()):x} / 0x{int(b.max()):x}, distinct values {len(np.unique(a))} / {len(np.unique(b))}")
    else:
        print(f"{k:14s} abandon off / on identical: {same}")
    bad += 0 if same else 1
print("ALL IDENTICAL" if bad == 0 else f"{bad} arrays differ")
sys.exit(1 if bad else 0)
