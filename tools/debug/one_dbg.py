"""Debug aid: per-step draws of rollout_one_kernel (GPMPC_ROLLOUT_ONE=1) against the oracle."""
import os, sys
os.environ.setdefault("GPMPC_ROLLOUT_ONE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from oracle import agent_oracle as ao
from sampling_gpmpc_amd.rollout import forward_sampling_rollout
from sampling_gpmpc_amd.workloads import fs_params, synthetic_u_ff
pname, Ns, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
p = fs_params(pname, Ns, H, nograd=False, beta=(3.0 if "car" in pname else None))
pg = {**p, "common": {**p["common"], "use_cuda": True}}
torch.manual_seed(123456)
agent = sg.Agent(pg, sg.make_env(pg))
erv = agent.epistimic_random_vector.cpu()
oagent = ao.OracleAgent(p, ao.make_oracle_env(p), erv)
u = synthetic_u_ff(agent.nu, H)
X, Y = forward_sampling_rollout(agent, u, return_samples=True, check=False)
print("path", sg._lib.load().gpmpc_debug_last_rollout_path())
Xo, Yo = ao.forward_sampling_rollout(oagent, u, return_samples=True)
np.set_printoptions(precision=6, linewidth=200)
for t in range(H):
    e = np.abs(Y[:, :, t] - Yo[:, :, t]).max() / (np.abs(Yo[:, :, t]).max() + 1e-300)
    print("step", t, "n_h", 3 * t, "rel err Y", f"{e:.2e}", "gpu", Y[0, :, t].ravel(), "oracle", Yo[0, :, t].ravel())
print("X_traj rel err", np.abs(X - Xo).max() / np.abs(Xo).max())

# ---- dense reference of chain 0 at the debug step against the kernel's dump (GPMPC_ONE_DEBUG build) ------------------------
import ctypes as C
raw = C.CDLL(sg._lib.LIB_PATH)
if hasattr(raw, "gpmpc_debug_read_one") and os.environ.get("ONE_DBG_STEP"):
    from oracle.gp_oracle import GPHyper, scaled_rbf_kernel
    buf = (C.c_double * 4096)()
    raw.gpmpc_debug_read_one(buf)
    dump = np.array(buf).reshape(64, 64)
    step = int(os.environ["ONE_DBG_STEP"])
    hy = GPHyper.from_params(p, True)
    Xr = oagent.Dyn_gp_X_train.double()
    Xh = oagent.Hallcinated_X_train[0, 0, :step].double()
    xt = oagent.Hallcinated_X_train[0, 0, step:step + 1].double()
    ell, osc = hy.ell[0], hy.outputscale[0]
    allx = torch.cat([Xr, Xh], 0)
    K = scaled_rbf_kernel(allx, allx, ell, osc, True)
    nr = Xr.shape[0]
    obs = [3 * i for i in range(nr)] + [3 * nr + k for k in range(3 * step)]
    Koo = K[obs][:, obs].clone()
    nz = torch.cat([torch.full((nr,), float(hy.noise_diag[0])), hy.noise_diag.repeat(step)])
    Koo += torch.diag(nz)
    Lf = torch.linalg.cholesky(Koo)
    Kot = scaled_rbf_kernel(allx, xt, ell, osc, True)[obs]
    v = torch.linalg.solve_triangular(Lf, Kot, upper=False)
    Lhh = Lf[nr:, nr:].numpy()
    vh = v[nr:].numpy()
    kh = Kot[nr:].numpy()
    n_h = 3 * step
    i0 = n_h & 3
    cols = [(i0 + b) & 3 for b in range(3)]
    def nat(slot, b=0):
        return np.array([[dump[slot, 16 * k + 4 * b + j] for j in range(4)] for k in range(4)])
    print("n_h", n_h, "i0", i0, "task columns", cols, "label column", (i0 + 3) & 3)
    for b in range(4):
        print("RN[0] block", b); print(nat(5, b))
    print("dense k_h rows (task columns):"); print(kh)
    for b in range(4):
        print("acc after off, block", b); print(nat(10, b))
    print("dense rhs = L_hh v_h:"); print(Lhh @ vh)
    for b in range(4):
        print("W0 before replicate block", b); print(nat(11, b))
    for tl in range(4):
        for b in range(4):
            print("Vrep[%d] block %d" % (tl, b)); print(nat(tl, b))
    print("dense v_h:"); print(vh)
    for b in range(4):
        print("GD[0] as operator (L^-1), block", b); print(nat(12, b))
    print("dense L_hh:"); print(Lhh)
    print("dense inv of diag tiles:")
    for tl in range((n_h + 3) // 4):
        r0 = 4 * tl; m = min(4, n_h - r0); Ld = np.eye(4); Ld[:m, :m] = Lhh[r0:r0 + m, r0:r0 + m]; print(np.linalg.inv(Ld))
    print("Stot block0"); print(nat(6, 0)); print("Srr block0"); print(nat(7, 0))
