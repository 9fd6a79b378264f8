import os, sys, time, warnings, gc
if os.environ.get("NOGC") == "1":
    gc.disable()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd.workloads import closed_loop_params, synthetic_u_ff
p = closed_loop_params("params_car_residual", 1024, 40, 4, 4)
p["common"]["use_cuda"] = True; p["agent"]["base_sample_generator"] = "counter"
agent = sg.Agent(p, sg.make_env(p))
H, Ns = 40, 1024
x0 = np.asarray(p["env"]["start"], dtype=np.float64)[:4]
u_h = np.zeros((H, 2)); xg, w = np.zeros(H), np.ones(H)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for step in range(4):
        agent.mpc_iteration(step)
        x_h = np.tile(x0, (H, Ns))
        for k in range(4):
            t = [T()]
            agent.train_hallucinated_dynGP(k); t.append(T())
            bx = agent.get_batch_x_hat(x_h, u_h); t.append(T())
            g_xu = agent.env_model.get_g_xu_hat(bx).contiguous(); t.append(T())
            ms0 = torch.cuda.memory_stats(); a0, f0 = ms0.get("num_device_alloc", 0), ms0.get("num_device_free", 0)
            y = agent.sample_gp(g_xu, base_samples=agent.epistimic_random_vector[step][k]); t.append(T())
            ms1 = torch.cuda.memory_stats(); a1, f1 = ms1.get("num_device_alloc", 0), ms1.get("num_device_free", 0)
            agent.update_hallucinated_Dyn_dataset(g_xu, y); t.append(T())
            d = np.diff(t) * 1e3
            print(f"step {step} k={k}: train {d[0]:6.2f} x_hat {d[1]:6.2f} g_xu {d[2]:6.2f} sample_gp {d[3]:7.2f} update {d[4]:6.2f}   hipMalloc {a1 - a0} hipFree {f1 - f0} reserved {ms1.get('reserved_bytes.all.current', 0) / 2**30:.1f} GiB", flush=True)
