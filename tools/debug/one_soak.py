"""Soak of rollout_one_kernel (path 4) against rollout_fast_kernel (path 1) and the generic kernel (path 0): random launch
sizes, horizons, options (feedback, per-sample start states, sampling clip, variance-is-zero threshold, outputs on / off) on the
same base samples.  Two independent implementations of the same arithmetic must agree to round-off on EVERY case."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import _lib
from sampling_gpmpc_amd.rollout import rollout_device
from sampling_gpmpc_amd.workloads import fs_params, synthetic_u_ff

F64 = torch.float64
lib = _lib.load()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = 0.0
for case in range(ncase):
    Ns = int(rng.choice([1, 2, 3, 5, 17, 64, 255, 256, 700, 1024, 1500, 2048]))
    H = int(rng.randint(2, 31))
    fb = bool(rng.randint(2))
    beta = float(rng.choice([1e9, 3.0, 0.7, 0.2]))
    vz = float(rng.choice([-1.0, -1.0, 1e-7, 1e-5]))
    per_sample = bool(rng.randint(2))
    want = bool(rng.randint(3) > 0)
    p = fs_params("params_pendulum1D_samples", Ns, H, nograd=False, feedback=fb)
    p["common"]["use_cuda"] = True
    p["agent"]["base_sample_generator"] = "counter"
    agent = sg.Agent(p, sg.make_env(p))
    dev = agent.torch_device
    g = torch.Generator().manual_seed(1000 + case)
    z = torch.randn(H, Ns * 3, generator=g, dtype=F64).clamp(-2.5, 2.5).to(dev)
    u_ff = synthetic_u_ff(1, H) * float(rng.uniform(0.3, 1.5))
    x0 = None
    if per_sample:
        x0 = (torch.tensor(p["env"]["start"][:2], dtype=F64) + 0.3 * torch.randn(Ns, 2, generator=g, dtype=F64)).to(dev)
    out = {}
    for kern in (_lib.KERNEL_ONE, _lib.KERNEL_FAST, _lib.KERNEL_GENERIC):
        lib.gpmpc_rollout_pin_kernel(kern)
        res = rollout_device(agent, u_ff, z.reshape(-1), z.shape[1], H=H, mode=_lib.MODE_RECONDITIONED,
                             use_model_without_derivatives=False, use_feedback=fb, x0=x0, var_zero_thr=vz, beta=beta, want_samples=want)
        path = lib.gpmpc_debug_last_rollout_path()
        assert path == kern, (path, kern)
        out[kern] = (res.X_traj.cpu().numpy(), res.Y.cpu().numpy() if want else None, res.info.cpu().numpy())
    lib.gpmpc_rollout_pin_kernel(-1)
    X4, Y4, i4 = out[_lib.KERNEL_ONE]
    errs = []
    for other in (_lib.KERNEL_FAST, _lib.KERNEL_GENERIC):
        Xo, Yo, io = out[other]
        ex = float(np.abs(X4 - Xo).max() / max(np.abs(Xo).max(), 1e-300))
        ey = float(np.abs(Y4 - Yo).max() / max(np.abs(Yo).max(), 1e-300)) if want else 0.0
        errs += [ex, ey]
        assert np.isfinite(X4).all() and ex < 1e-8 and ey < 1e-6, (case, Ns, H, fb, beta, vz, per_sample, ex, ey)
        assert ((i4 & ~_lib.INFO_VAR_CLAMPED) == (io & ~_lib.INFO_VAR_CLAMPED)).all(), (case, i4.max(), io.max())
    worst = max(worst, max(errs))
    print(f"case {case:2d}: Ns={Ns:5d} H={H:2d} feedback={int(fb)} beta={beta:g} var_zero={vz:g} x0/sample={int(per_sample)} outputs={int(want)}: "
          f"vs fast X {errs[0]:.1e} Y {errs[1]:.1e}; vs generic X {errs[2]:.1e} Y {errs[3]:.1e}; info 0x{int(i4.max()):x}", flush=True)
print(f"{ncase} cases, worst relative difference {worst:.2e}")
