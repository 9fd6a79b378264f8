"""gpmpc_joint_sample, VALU path against matrix-pipe path (gpmpc_joint_pin_path) on the same inputs: mean, variance, covariance,
sample, and the time of either draw.  A kernel-vs-kernel comparison for bring-up and timing - the parity evidence is the test
suite (both paths against the oracle).

    python tools/debug/joint_paths.py [--pendulum] [--ns N] [--H H] [--iters K] [--cholesky] [--no-cache] [--time]
"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import _lib
from tests.helpers import load_params


def arg(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


pname = "params_pendulum1D_samples" if "--pendulum" in sys.argv else "params_car_residual"
Ns, H, iters = arg("--ns", 16), arg("--H", 40 if "car" in pname else 30), arg("--iters", 4)
p = load_params(pname)
p["common"]["use_cuda"] = True
p["agent"]["num_dyn_samples"], p["optimizer"]["H"] = Ns, H
p["agent"]["true_dyn_as_sample"] = False
p["agent"]["base_sample_generator"] = "vectorized"
p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = 1, iters
if "--cholesky" in sys.argv:
    p["agent"]["Dyn_gp_jitter"] = 1e-9
torch.manual_seed(3)
agent = sg.Agent(p, sg.make_env(p))
lib = _lib.load()
nx, nu = agent.nx, agent.nu
g = torch.Generator().manual_seed(5)
x0 = np.array(p["env"]["start"], dtype=np.float64)
agent.mpc_iteration(0)
worst = 0.0
for it in range(iters):
    x_h = np.tile(x0, (H, Ns)) + 0.05 * torch.randn(H, Ns * nx, generator=g, dtype=torch.float64).numpy() + 0.02 * np.arange(H)[:, None]
    u_h = 0.3 * torch.randn(H, Ns, nu, generator=g, dtype=torch.float64).numpy()
    agent.train_hallucinated_dynGP(it)
    bx = agent.get_batch_x_hat_u_diff(x_h, u_h)
    g_xu = agent.env_model.get_g_xu_hat(bx).contiguous()
    z = agent.epistimic_random_vector[agent.mpc_iter][it]
    cache = agent._ws_cache.get("joint_factor_cache")
    if cache is not None and "--no-cache" in sys.argv:
        cache.enabled = False
    held = cache.n_valid if cache is not None else 0
    res = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for path in (_lib.JOINT_VALU, _lib.JOINT_MFMA):
            lib.gpmpc_joint_pin_path(path)
            c = agent._ws_cache.get("joint_factor_cache")
            if c is not None:
                c.rewind(held)
            post = agent.model_i(g_xu)
            try:
                y = post._sample(z, clip=True, beta=p["agent"]["Dyn_gp_beta"], var_zero_thr=p["agent"]["Dyn_gp_variance_is_zero"])
            except Exception as e:
                bad = (post.last_info & _lib.INFO_TRAIN_CHOL_FAIL) != 0
                print(f"k={it} path {path}: {type(e).__name__}; cached rows {post.n_cached_rows if hasattr(post, 'n_cached_rows') else '?'}; chains flagged {int(bad.sum())} of {bad.numel()}: {bad.nonzero()[:8].tolist()}", flush=True)
                raise
            took = lib.gpmpc_joint_last_path()
            if c is not None:
                c.rewind(held)
            cov = post.covariance_matrix
            ms = None
            if "--time" in sys.argv:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                best = 1e9
                for rep in range(6):
                    if c is not None:
                        c.rewind(held)
                    ev[0].record(); agent.sample_gp(g_xu, base_samples=z); ev[1].record(); torch.cuda.synchronize()
                    if rep:
                        best = min(best, ev[0].elapsed_time(ev[1]))
                ms = best
            res[path] = (post.mean.clone(), post.variance.clone(), cov.clone(), y.clone(), post.last_info.clone(), took, ms)
        lib.gpmpc_joint_pin_path(_lib.JOINT_AUTO)
        c = agent._ws_cache.get("joint_factor_cache")
        if c is not None:
            c.rewind(held)
        agent.get_batch_gp_sensitivities(bx, it)
    a, b = res[_lib.JOINT_VALU], res[_lib.JOINT_MFMA]
    rel = lambda u, v: float((u - v).abs().max() / v.abs().max())
    e = [rel(b[i], a[i]) for i in range(4)]
    worst = max(worst, e[0], e[2])
    n_ho = int(agent.model_i.h_slots.numel()) - H * agent.in_dim_y if False else None
    print(f"{pname} Ns={Ns} H={H} k={it}: paths {a[5]}/{b[5]}  rel diff mean {e[0]:.2e} var {e[1]:.2e} cov {e[2]:.2e} y {e[3]:.2e}"
          f"  info {int(a[4].max())}/{int(b[4].max())} finite {bool(torch.isfinite(b[3]).all())}"
          + (f"  ms {a[6]:.3f} / {b[6]:.3f}" if a[6] is not None else ""), flush=True)
print("worst mean/cov rel diff", worst)
if os.environ.get("GPMPC_PHASE_TIMERS") == "1":
    import ctypes as C
    modes = {0: "test", 1: "factor", 2: "top", 3: "bottom"}
    for md in ([1] if "--factor-phases" in sys.argv else []) + ([2, 3] if iters > 4 else []):
        om = (C.c_longlong * 40)()
        lib.gpmpc_debug_read_joint_mfma_phases_of(md, om)
        print("joint_test_mfma_kernel, %s-mode launch (wave 0 / wave 7): prologue %d / %d entries %d / %d solve %d / %d | gram: init %d / %d publish %d / %d products %d / %d out %d / %d"
              % ((modes[md],) + tuple(v for i in range(7) for v in (om[i], om[8 + i]))))
    out = (C.c_longlong * 40)()
    lib.gpmpc_debug_read_joint_mfma_phases(out)
    for w, o in ((0, 0), (7, 8)):
        print("joint_test_mfma_kernel phases (cycles, wave %d of block 0, last launch): prologue %d entries %d solve %d | gram: init %d publish %d products %d out %d"
              % ((w,) + tuple(out[o:o + 7])))
    print("  inside the prologue (wave 0): tables + first chunk %d, real-row tiles %d, y' %d, point runs + barrier %d, diagonal-tile requests %d, pad + barrier %d, K_cc %d, wait + inversion %d" % tuple(out[20:28]))
    print("  inside the kernel entries (wave 0, the first 10 chunks): pairs %d, barrier %d, into the accumulators %d, barrier %d" % tuple(out[28:32]))
    print("  inside the substitution: wave 0: diagonal steps %d, hand-overs + first tiles %d, runs %d; wave 7: %d, %d, %d" % (out[7], out[16], out[17], out[15], out[18], out[19]))
    print("  hand-overs (the three parts that are NOT in the 'hand-overs + first tiles' figure above): wave 0: own pieces of the chunk %d, barrier %d, next chunk's requests %d; wave 7: %d, %d, %d" % tuple(out[32:38]))
    jo = (C.c_longlong * 16)()
    lib.gpmpc_debug_read_joint_phases(jo)
    print("joint_kernel, the last factor-only launch (CHOL phase; cycles, block 0): real columns %d, block start (descriptors, S / kernel entries) %d, update %d, diagonal block %d, row solve + stores %d" % tuple(jo[10:15]))
