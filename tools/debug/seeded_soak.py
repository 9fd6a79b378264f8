"""Soak of seeded / value-only rollouts on the tiled kernel (path 3) against the generic kernel (path 0): random seed counts,
value-seed counts, hall_tasks, horizons, launch sizes, pendulum and car; seed points from a first rollout of the same agent."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import _lib
from sampling_gpmpc_amd.rollout import rollout_device
from sampling_gpmpc_amd.workloads import fs_params, synthetic_u_ff

F64 = torch.float64
lib = _lib.load()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 30
worst = 0.0
for case in range(ncase):
    car = bool(rng.randint(2))
    pname, nu = ("params_car_residual_fs", 2) if car else ("params_pendulum1D_samples", 1)
    Ns = int(rng.choice([1, 2, 3, 5, 9, 33, 64, 257]))
    n0 = int(rng.randint(0, 40))
    nv = int(rng.randint(0, 6))
    ht = int(rng.choice([1, 3]))
    Hmax = 64 - n0 - nv                                   # 3 (n0 + nv + H - 1) <= 192
    H = int(rng.randint(1, min(Hmax, 30) + 1))
    if n0 + nv == 0 and (H < 2 or ht == 3):
        ht, H = 1, max(H, 2)                              # (the unseeded T-task call has its own tests)
    p = fs_params(pname, Ns, max(H, n0 + nv, 2), nograd=False, beta=(3.0 if car else None))
    p["common"]["use_cuda"] = True
    p["agent"]["base_sample_generator"] = "counter"
    agent = sg.Agent(p, sg.make_env(p))
    dev, gny = agent.torch_device, agent.g_ny
    g = torch.Generator().manual_seed(100 + case)
    seeds = vseeds = None
    if n0 + nv:
        z0 = torch.randn(n0 + nv, Ns * gny * 3, generator=g, dtype=F64).clamp(-2, 2).to(dev)
        lib.gpmpc_rollout_pin_kernel(_lib.KERNEL_GENERIC)
        r0 = rollout_device(agent, synthetic_u_ff(nu, n0 + nv), z0.reshape(-1), z0.shape[1], H=n0 + nv, mode=_lib.MODE_RECONDITIONED,
                            use_model_without_derivatives=False)
        Xall = r0.Xi[:, None].expand(-1, gny, -1, -1).contiguous()
        if n0:
            seeds = (Xall[:, :, :n0].contiguous(), r0.Y[:, :, :n0].contiguous())
        if nv:
            vseeds = (Xall[:, :, n0:].contiguous(), r0.Y[:, :, n0:].contiguous())
    z = torch.randn(H, Ns * gny * 3, generator=g, dtype=F64).clamp(-2, 2).to(dev)
    u_ff = synthetic_u_ff(nu, H) * float(rng.uniform(0.3, 1.2))
    out = {}
    for kern in (_lib.KERNEL_TILES, _lib.KERNEL_GENERIC):
        lib.gpmpc_rollout_pin_kernel(kern)
        res = rollout_device(agent, u_ff, z.reshape(-1), z.shape[1], H=H, mode=_lib.MODE_RECONDITIONED,
                             use_model_without_derivatives=False, hall_tasks=ht, seeds=seeds, value_seeds=vseeds)
        assert lib.gpmpc_debug_last_rollout_path() == kern, (case, lib.gpmpc_debug_last_rollout_path(), kern)
        out[kern] = (res.X_traj.cpu().numpy(), res.Y.cpu().numpy(), res.info.cpu().numpy())
    lib.gpmpc_rollout_pin_kernel(-1)
    Xt, Yt, it = out[_lib.KERNEL_TILES]
    Xg, Yg, ig = out[_lib.KERNEL_GENERIC]
    ex = float(np.abs(Xt - Xg).max() / max(np.abs(Xg).max(), 1e-300))
    ey = float(np.abs(Yt - Yg).max() / max(np.abs(Yg).max(), 1e-300))
    worst = max(worst, ex, ey)
    print(f"case {case:2d}: {'car     ' if car else 'pendulum'} Ns={Ns:4d} H={H:2d} seeds={n0:2d}+{nv} hall_tasks={ht}: X {ex:.1e} Y {ey:.1e} "
          f"info 0x{int(it.max()):x}/0x{int(ig.max()):x}", flush=True)
    assert np.isfinite(Xt).all() and ex < 1e-7 and ey < 1e-5, (case, ex, ey)
    assert ((it & ~_lib.INFO_VAR_CLAMPED) == (ig & ~_lib.INFO_VAR_CLAMPED)).all()
print(f"{ncase} cases, worst relative difference {worst:.2e}")
