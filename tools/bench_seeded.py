"""Seeded rollouts (gpmpc_rollout_seeded without a kept factor state) on the tiled kernel against the generic kernel:
the second rollout on an Agent (seed points with all tasks) and the fused prepare_dynamics_set's launch shape (seed points
+ value-only seeds, hall_tasks = 1).   python tools/bench_seeded.py [Ns]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import _lib
from sampling_gpmpc_amd.rollout import rollout_device
from sampling_gpmpc_amd.workloads import fs_params, synthetic_u_ff

F64 = torch.float64
Ns = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = _lib.load()
for pname, nu in (("params_pendulum1D_samples", 1), ("params_car_residual_fs", 2)):
    for (H, n0, nv, ht) in ((20, 19, 0, 3), (8, 24, 1, 1)):
        p = fs_params(pname, Ns, H, nograd=False, beta=(3.0 if "car" in pname else None))
        p["common"]["use_cuda"] = True
        agent = sg.Agent(p, sg.make_env(p))
        dev = agent.torch_device
        g = torch.Generator().manual_seed(3)
        gny = agent.g_ny
        # seed points as a closed loop has them: the points and draws of a first (unseeded) rollout of this agent
        z0 = torch.randn(n0 + nv, Ns * gny * 3, generator=g, dtype=F64).clamp(-2, 2).to(dev)
        r0 = rollout_device(agent, synthetic_u_ff(nu, n0 + nv), z0.reshape(-1), z0.shape[1], H=n0 + nv, mode=_lib.MODE_RECONDITIONED,
                            use_model_without_derivatives=False)
        Xall = r0.Xi[:, None].expand(-1, gny, -1, -1).contiguous()
        Xs, Ys = Xall[:, :, :n0].contiguous(), r0.Y[:, :, :n0].contiguous()
        vs = (Xall[:, :, n0:].contiguous(), r0.Y[:, :, n0:].contiguous()) if nv else None
        z = torch.randn(H, Ns * gny * 3, generator=g, dtype=F64).clamp(-2, 2).to(dev)
        u_ff = synthetic_u_ff(nu, H)
        ms = {}
        for kern in (_lib.KERNEL_TILES, _lib.KERNEL_GENERIC):
            lib.gpmpc_rollout_pin_kernel(kern)
            def run():
                return rollout_device(agent, u_ff, z.reshape(-1), z.shape[1], H=H, mode=_lib.MODE_RECONDITIONED,
                                      use_model_without_derivatives=False, hall_tasks=ht, seeds=(Xs, Ys), value_seeds=vs)
            for _ in range(3):
                res = run()
            torch.cuda.synchronize()
            path = lib.gpmpc_debug_last_rollout_path()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(5):
                res = run()
            ev[1].record()
            torch.cuda.synchronize()
            ms[kern] = ev[0].elapsed_time(ev[1]) / 5
            fin = bool(torch.isfinite(res.X_traj).all())
            keep = res.X_traj.clone()
            if kern == _lib.KERNEL_GENERIC:
                err = float((keep - first).abs().max() / first.abs().max())
            else:
                first, err = keep, 0.0
            print(f"{pname:28s} Ns={Ns} H={H} seeds={n0}+{nv} hall_tasks={ht}: kernel path {path} {ms[kern]:8.3f} ms  finite={fin} "
                  f"info=0x{int(torch.bitwise_or(res.info, res.info).max().item()):x} rel diff to the tiled run {err:.1e}", flush=True)
        lib.gpmpc_rollout_pin_kernel(-1)
        print(f"   tiled / generic speed-up: {ms[_lib.KERNEL_GENERIC] / ms[_lib.KERNEL_TILES]:.2f}x", flush=True)
