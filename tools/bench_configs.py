"""Time the other BASELINE configs (not the bench.py line): cfg3 car mode R, cfg4 car mode I (as shipped);
`--sweep` adds a sample-count sweep of the configs[1] workload (how the one-sample-per-SIMD latency bound of Ns=1024
turns into throughput once several waves share a SIMD / several rounds of workgroups run)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import _lib
from sampling_gpmpc_amd.rollout import RolloutRunner
from tests.helpers import fs_params, synthetic_u_ff

def run(pname, Ns, H, nograd, reps=10, n_data_x=None, hall_tasks=None):
    p = fs_params(pname, Ns, H, nograd=nograd, beta=(3.0 if ("car" in pname and not nograd) else None))
    if n_data_x is not None:
        p["env"]["n_data_x"] = n_data_x
    p["common"]["use_cuda"] = True; p["agent"]["base_sample_generator"] = "vectorized"
    torch.manual_seed(1)
    agent = sg.Agent(p, sg.make_env(p)); u_ff = synthetic_u_ff(agent.nu, H); erv = agent.epistimic_random_vector
    T = 1 if nograd else 3
    per = Ns * agent.g_ny * T
    mode = _lib.MODE_INDEPENDENT if nograd else _lib.MODE_RECONDITIONED
    if H > 45:
        u_ff = u_ff * 0.5
    r = RolloutRunner(agent, u_ff, erv.reshape(-1)[per:], erv.shape[1] * per, H, mode, nograd, hall_tasks=hall_tasks)
    for _ in range(2): r.launch()
    if "--sustained" in sys.argv:                      # bring the GPU to its sustained clocks first (tools/clock_check.py)
        for _ in range(max(int(0.4 / 2e-4), 200)): r.launch()
        reps = max(reps, 200)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): r.launch()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ok = bool(torch.isfinite(r.X_traj).all()); bits = int(r.info.max().item())
    kern = _lib.load().gpmpc_rollout_last_kernel()
    print(f"{pname:28s} grid={n_data_x or '-'} Ns={Ns:7d} H={H} mode={'I' if nograd else 'R'}{'' if hall_tasks is None else f' labels={hall_tasks}'} kernel={kern}: {ms:9.3f} ms/rollout  {Ns*H/ms*1e3/1e6:10.1f} M traj-steps/s  finite={ok} info=0x{bits:x}", flush=True)

if __name__ == "__main__":
    if "--sweep" in sys.argv:
        for ns in (256, 1024, 2048, 4096, 16384, 65536):
            run("params_pendulum1D_samples", ns, 30, False, 10)
        for h in (10, 15, 20, 43):
            run("params_pendulum1D_samples", 4096, h, False, 10)
        sys.exit(0)
    if "--car" in sys.argv:
        run("params_car_residual_fs", 4096, 40, False, 5)
        sys.exit(0)
    if "--value-only" in sys.argv:
        # SURVEY 8d, cfg4 "R value-only labels": the T = 3 model reconditioned on the VALUE of each drawn point only
        # (src/agent.py:402 NaNs the gradient labels); the per-GPU share of the 8-GPU job and, memory permitting, the whole job
        run("params_car_residual_fs", 4096, 40, False, 5, hall_tasks=1)
        run("params_car_residual_fs", 32768, 40, False, 5, hall_tasks=1)
        run("params_car_residual_fs", 4096, 40, False, 5)
        run("params_car_residual_fs", 32768, 40, False, 5)
        sys.exit(0)
    if "--mode-i" in sys.argv:
        run("params_car_residual_fs", 32768, 40, True, 10)
        run("params_car_residual_fs", 262144, 40, True, 10)
        sys.exit(0)
    run("params_pendulum1D_samples", 1024, 30, False, 20)
    run("params_car_residual_fs", 4096, 40, False, 5)
    run("params_car_residual_fs", 32768, 40, True, 5)
    run("params_car_residual_fs", 262144, 40, True, 3)
