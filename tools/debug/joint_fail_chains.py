import os, sys, warnings
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import _lib
from tests.helpers import load_params
Ns, H = int(sys.argv[1]), 40
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3      # 5: the last draw conditions on 45 + 480 slots (TOP + BOTTOM launches)
p = load_params("params_car_residual")
p["common"]["use_cuda"] = True
p["agent"]["num_dyn_samples"], p["optimizer"]["H"] = Ns, H
p["agent"]["true_dyn_as_sample"] = False
p["agent"]["base_sample_generator"] = "vectorized"
p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = 1, iters
torch.manual_seed(3)
agent = sg.Agent(p, sg.make_env(p))
lib = _lib.load()
g = torch.Generator().manual_seed(5)
x0 = np.array(p["env"]["start"], dtype=np.float64)
agent.mpc_iteration(0)
for it in range(iters):
    x_h = np.tile(x0, (H, Ns)) + 0.05 * torch.randn(H, Ns * agent.nx, generator=g, dtype=torch.float64).numpy() + 0.02 * np.arange(H)[:, None]
    u_h = 0.3 * torch.randn(H, Ns, agent.nu, generator=g, dtype=torch.float64).numpy()
    agent.train_hallucinated_dynGP(it)
    bx = agent.get_batch_x_hat_u_diff(x_h, u_h)
    g_xu = agent.env_model.get_g_xu_hat(bx).contiguous()
    z = agent.epistimic_random_vector[agent.mpc_iter][it]
    cache = agent._ws_cache.get("joint_factor_cache")
    held = cache.n_valid if cache is not None else 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for path in (1, 2):
            lib.gpmpc_joint_pin_path(path)
            c = agent._ws_cache.get("joint_factor_cache")
            if c is not None: c.rewind(held)
            post = agent.model_i(g_xu)
            y, bits = post._run(z, True, 2.0, 1e-9, raise_chol_fail=False)
            info = post.last_info
            bad = (info & _lib.INFO_TRAIN_CHOL_FAIL) != 0
            if path == 1:
                truth = (post.mean.clone(), post.variance.clone())
            print(f"k={it} path {path}: chains with TRAIN_CHOL_FAIL: {int(bad.sum())} of {bad.numel()}  first: {bad.nonzero()[:6].tolist()}  nan mean {int(torch.isnan(post.mean).sum())}", flush=True)
        if it >= 1:      # run-to-run determinism of the matrix-pipe path
            lib.gpmpc_joint_pin_path(2)
            ref = None
            for rep in range(int(os.environ.get("REPS", "12"))):
                c = agent._ws_cache.get("joint_factor_cache")
                if c is not None: c.rewind(held)
                post = agent.model_i(g_xu)
                y, bits = post._run(z, True, 2.0, 1e-9, raise_chol_fail=False)
                cur = (post.mean.clone(), post.variance.clone())
                em = ((cur[0] - truth[0]).abs() / truth[0].abs().max()).flatten(2)      # (Ns, g_ny, m*T)
                ev = ((cur[1] - truth[1]).abs() / truth[1].abs().max()).flatten(2)
                wrongm, wrongv = em > 1e-8, ev > 1e-8
                if wrongm.any() or wrongv.any():
                    ch = (wrongm.any(-1) | wrongv.any(-1)).nonzero()[:3].tolist()
                    for c in ch:
                        print(f"    rep {rep} chain {c}: mean wrong at columns {wrongm[c[0], c[1]].nonzero().flatten().tolist()}  var wrong at {wrongv[c[0], c[1]].nonzero().flatten().tolist()}  max rel {float(em[c[0], c[1]].max()):.1e} {float(ev[c[0], c[1]].max()):.1e}")
                bad = (post.last_info & _lib.INFO_TRAIN_CHOL_FAIL) != 0
                if ref is None:
                    ref = cur
                else:
                    dm = (cur[0] != ref[0]).flatten(2).any(-1)
                    dv = (cur[1] != ref[1]).flatten(2).any(-1)
                    d0 = (cur[0] - ref[0]).abs()
                    worst = int(d0.flatten(2).max(-1).values.flatten().argmax())
                    wc = (worst // 3, worst % 3)
                    tp = d0[wc[0], wc[1]].max(-1).values
                    print(f"  k={it} rep {rep}: chains whose mean differs from rep 0: {int(dm.sum())}, variance: {int(dv.sum())}; max |diff| mean {float(d0.max()):.2e} var {float((cur[1]-ref[1]).abs().max()):.2e}; worst chain {wc}: test points with a differing mean {(tp > 0).nonzero().flatten().tolist()[:45]}", flush=True)
        lib.gpmpc_joint_pin_path(int(os.environ.get("FINAL_PATH", "0")))
        c = agent._ws_cache.get("joint_factor_cache")
        if c is not None: c.rewind(held)
        try:
            agent.get_batch_gp_sensitivities(bx, it)
        except Exception as e:
            print("final call failed:", type(e).__name__)
            break
