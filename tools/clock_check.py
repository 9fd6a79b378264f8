"""Sustained per-launch time, shader clock and socket power (rocm-smi) of the mode-I full-size launch and of the bench
kernel over ~2 s of back-to-back launches: a cold 10..50-launch measurement reads 5..25 % slower than this steady state
(MI355X: mode I Ns=262144 0.543 ms sustained vs 0.68 ms over 10 launches; bench kernel 0.1088 vs 0.1166 ms)."""
import sys, os, time, subprocess, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import _lib
from sampling_gpmpc_amd.rollout import RolloutRunner
from tests.helpers import fs_params, synthetic_u_ff
def mk(pname, Ns, H, nograd):
    p = fs_params(pname, Ns, H, nograd=nograd, beta=(3.0 if ("car" in pname and not nograd) else None)); p["common"]["use_cuda"] = True; p["agent"]["base_sample_generator"] = "vectorized"
    agent = sg.Agent(p, sg.make_env(p)); u_ff = synthetic_u_ff(agent.nu, H); erv = agent.epistimic_random_vector
    T = 1 if nograd else 3; per = Ns * agent.g_ny * T
    return RolloutRunner(agent, u_ff, erv.reshape(-1)[per:], erv.shape[1] * per, H, _lib.MODE_INDEPENDENT if nograd else _lib.MODE_RECONDITIONED, nograd)
def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        return " | ".join(l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l or "power" in l))[:300]
    except Exception as e:
        return repr(e)
for name, r, reps in (("mode I Ns=262144", mk("params_car_residual_fs", 262144, 40, True), 3000), ("mode R pendulum Ns=1024", mk("params_pendulum1D_samples", 1024, 30, False), 20000)):
    for _ in range(3): r.launch()
    torch.cuda.synchronize()
    res = []
    th = threading.Thread(target=lambda: res.append(smi()))
    t0 = time.perf_counter()
    for i in range(reps):
        r.launch()
        if i == reps // 4: th.start()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0; th.join()
    print(name, f"{dt/reps*1e3:.4f} ms/launch over {dt:.1f} s;", res[0] if res else "no smi")
print("idle:", smi())
