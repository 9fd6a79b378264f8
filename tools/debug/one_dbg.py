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

# ---- raw dump of chain 0 at the debug step (GPMPC_ONE_DEBUG build): natural-map registers, block by block ------------------
import ctypes as C
raw = C.CDLL(sg._lib.LIB_PATH)
if hasattr(raw, "gpmpc_debug_read_one") and os.environ.get("ONE_DBG_STEP"):
    buf = (C.c_double * 4096)()
    raw.gpmpc_debug_read_one(buf)
    dump = np.array(buf).reshape(64, 64)
    def nat(slot, b=0):
        return np.array([[dump[slot, 16 * k + 4 * b + j] for j in range(4)] for k in range(4)])
    for slot, name in ((0, "Vu[0]"), (1, "Vu[1]"), (2, "Vu[2]"), (3, "Vu[3]"), (5, "RN[2]"), (10, "acc of the debug row"), (11, "W"), (6, "Stot"), (8, "ud"), (9, "Gt")):
        for b in range(4):
            print(name, "block", b)
            print(nat(slot, b))
