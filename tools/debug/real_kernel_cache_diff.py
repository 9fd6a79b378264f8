"""The factor-cache rows of the second SQP iteration (nothing cached, 120 new slots) as joint_real_mfma_kernel + joint_chol_mfma_kernel leave
them against the factor-mode launch's: max differences of the real columns (X^T) and of the new rows' own block."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import workloads as wl
Ns, H = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 40
p = wl.closed_loop_params("params_car_residual", Ns, H, 2, 4)
p["common"]["use_cuda"] = True
p["agent"]["base_sample_generator"] = "counter"
lib = sg._lib.load()
res = {}
for mode in (0, 1):
    lib.gpmpc_debug_joint_real_kernel(mode)
    agent = sg.Agent(p, sg.make_env(p))
    x0 = np.asarray(p["env"]["start"], dtype=np.float64)[:agent.nx]
    u_h, x_h = wl.synthetic_u_ff(agent.nu, H), np.tile(x0, (H, agent.ns))
    agent.mpc_iteration(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k in range(2):
            agent.train_hallucinated_dynGP(k)
            agent.dyn_fg_jacobians_device(agent.get_batch_x_hat(x_h, u_h), k)
    torch.cuda.synchronize()
    c = agent._ws_cache["joint_factor_cache"]
    rows, n_r = c.rows, agent.model_i.plan.n_r
    CS = (n_r + rows + 1) & ~1
    per = rows * (CS + 1)
    buf = c.buf[: per * Ns * 3].reshape(Ns * 3, per)
    blk = buf[:, : rows * CS].reshape(Ns * 3, rows, CS)[:, :120, : n_r + 120].clone()
    res[mode] = (blk, buf[:, rows * CS: rows * CS + 120].clone())
    del agent
a, b = res[0], res[1]
n_r = 45
print("X^T (real columns): max |diff|", float((a[0][:, :, :n_r] - b[0][:, :, :n_r]).abs().max()), " scale", float(a[0][:, :, :n_r].abs().max()))
L0, L1 = torch.tril(a[0][:, :, n_r:]), torch.tril(b[0][:, :, n_r:])
print("new rows' own block (lower): max |diff|", float((L0 - L1).abs().max()), " scale", float(L0.abs().max()))
print("1 / diag: max |diff|", float((a[1] - b[1]).abs().max()))
d = (L0 - L1).abs().amax(dim=0)
bad = torch.nonzero(d > 1e-6)
print("first differing (row, col):", bad[:10].tolist(), " count", int(bad.shape[0]))
dx = (a[0][:, :, :n_r] - b[0][:, :, :n_r]).abs().amax(dim=0)
badx = torch.nonzero(dx > 1e-9)
print("X^T differing (row, col):", badx[:10].tolist(), " count", int(badx.shape[0]))
torch.set_printoptions(precision=6, linewidth=200)
print("factor mode, chain 0 row 0, real columns 0..15:", a[0][0, 0, :16])
print("real kernel, chain 0 row 0, real columns 0..15:", b[0][0, 0, :16])
print("factor mode, chain 0 row 17:", a[0][0, 17, :16])
print("real kernel, chain 0 row 17:", b[0][0, 17, :16])
