"""Debug aid: per-step draws of the tiled rollout kernel against the oracle (GPMPC_ROLLOUT_TILES=1)."""
import os, sys
os.environ["GPMPC_ROLLOUT_TILES"] = "1"
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
    print("step", t, "rel err Y", f"{e:.2e}", "gpu", Y[0, :, t].ravel(), "oracle", Yo[0, :, t].ravel())

# ---- dense reference of chain 0 at the debug step against the kernel's dumped tiles (GPMPC_TILES_DEBUG build) -------------
import ctypes as C
lib = sg._lib.load()
if hasattr(C.CDLL(sg._lib.LIB_PATH), "gpmpc_debug_read_tiles"):
    from oracle.gp_oracle import GPHyper, scaled_rbf_kernel
    raw = C.CDLL(sg._lib.LIB_PATH)
    buf = (C.c_double * 4096)()
    raw.gpmpc_debug_read_tiles(buf)
    dump = np.array(buf).reshape(64, 64)
    step = int(os.environ.get("TILES_DBG_STEP", "2"))
    hy = GPHyper.from_params(p, True)
    Xr = oagent.Dyn_gp_X_train.double()
    Xh = oagent.Hallcinated_X_train[0, 0, :step].double()          # points appended before the step
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
    Kot = scaled_rbf_kernel(allx, xt, ell, osc, True)[obs]        # (n_o, 3)
    v = torch.linalg.solve_triangular(Lf, Kot, upper=False)
    Lhh = Lf[nr:, nr:].numpy()
    vh = v[nr:].numpy()
    n_h = 3 * step
    i0 = n_h & 3
    cols = [(i0 + b) & 3 for b in range(3)]
    def nat(tile, b=0):                                            # natural layout of chain b: [row kq][col jq]
        return np.array([[dump[tile, 16 * k + 4 * b + j] for j in range(4)] for k in range(4)])
    print("n_h", n_h, "i0", i0, "task columns", cols)
    for tl in range((n_h + 3) // 4):
        Vt = nat(4 + tl)
        ref = np.zeros((4, 4))
        for k in range(4):
            if 4 * tl + k < n_h:
                ref[k, cols] = vh[4 * tl + k]
        print("solved tile", tl, "max abs diff (task columns)", np.abs(Vt[:, cols] - ref[:, cols]).max())
        print(Vt[:, cols]); print(ref[:, cols])
    for tl in range(min(3, (n_h + 3) // 4)):
        G = nat(9 + 4 * tl)                                        # (U^-1) natural = L_rr^-T
        r0 = 4 * tl
        Ld = np.eye(4)
        m = min(4, n_h - r0)
        Ld[:m, :m] = Lhh[r0:r0 + m, r0:r0 + m]
        print("diag tile", tl, "max abs diff of (L^-1)^T", np.abs(G - np.linalg.inv(Ld).T).max(), "of max", np.abs(G).max())
        print(G); print(np.linalg.inv(Ld).T)
        if tl > 0:
            X = nat(10 + 4 * tl)                                   # tile (tl, 0): -L^T natural
            ref = np.zeros((4, 4))
            ref[:, :m] = -Lhh[r0:r0 + m, 0:4].T
            print("tile (%d,0) max abs diff" % tl, np.abs(X - ref).max()); print(X); print(ref)

    if os.environ.get("TILES_DBG_APPEND"):
        for nm, sl in (("U", 26), ("drow", 27), ("dcol", 28), ("G", 29), ("U1", 30), ("drow1", 31), ("dcol1", 32), ("G1", 33), ("X", 34)):
            print(nm); print(nat(sl))
