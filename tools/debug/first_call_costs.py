import os, sys, time, warnings
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import workloads as wl
p = wl.closed_loop_params("params_car_residual", 1024, 40, 2, 4)
p["common"]["use_cuda"] = True
p["agent"]["base_sample_generator"] = "counter"
H = 40
def run(tag):
    agent = sg.Agent(p, sg.make_env(p))
    x0 = np.asarray(p["env"]["start"], dtype=np.float64)[:agent.nx]
    u_h = wl.synthetic_u_ff(agent.nu, H)
    x_h = np.tile(x0, (H, 1024))
    agent.mpc_iteration(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            agent.train_hallucinated_dynGP(k)
            t1 = time.perf_counter()
            bx = agent.get_batch_x_hat(x_h, u_h)
            t2 = time.perf_counter()
            gp_val, yg, ug = agent.dyn_fg_jacobians(bx, k)
            torch.cuda.synchronize(); t3 = time.perf_counter()
            print(f"{tag} k={k}: train {1e3*(t1-t0):.2f} x_hat {1e3*(t2-t1):.2f} draw+jac {1e3*(t3-t2):.2f} ms", flush=True)
            mean_next = gp_val[:, :, :, 0].mean(axis=0).T
            x_h = np.tile(np.vstack([x0[None, :], mean_next[:-1]]), (1, 1024))
    return agent
a1 = run("first agent")
del a1
a2 = run("second agent")
# sizes: the joint workspace and the factor cache the facade asks for at every iteration
from sampling_gpmpc_amd import _lib
lib = _lib.load()
mdl = a2.model_i
for n_ho in (0, 120, 240, 360, 480):
    wb = lib.gpmpc_joint_workspace_bytes(mdl.plan.desc, 1024, n_ho, 40)
    cb = lib.gpmpc_joint_cache_bytes(mdl.plan.desc, 1024, max(256, -(-4 * n_ho // 128) * 128)) if n_ho else 0
    print(f"n_ho={n_ho}: workspace {wb / 2**30:.2f} GiB, factor cache (4x headroom rows) {cb / 2**30:.2f} GiB")
for gb in (1, 4, 8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = torch.empty(gb << 27, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    del t; torch.cuda.empty_cache()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"torch.empty of {gb} GiB: {1e3 * (t1 - t0):.1f} ms, del + empty_cache: {1e3 * (t2 - t1):.1f} ms")
