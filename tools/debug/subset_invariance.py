"""A chain's joint draw must not depend on which other chains are in the launch: the car closed loop as shipped with Ns samples
against its first 8 samples alone, SQP iterations k = 0 ..; optional path pin (1 VALU, 2 matrix pipe).
    python tools/debug/subset_invariance.py [Ns] [pin] [iters]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import _lib
from tests.helpers import load_params
Ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pin = int(sys.argv[2]) if len(sys.argv) > 2 else 0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
H, sub = 40, 8
def make(n, erv=None):
    p = load_params("params_car_residual")
    p["common"]["use_cuda"] = True
    p["agent"]["num_dyn_samples"], p["optimizer"]["H"] = n, H
    p["agent"]["true_dyn_as_sample"] = False
    p["agent"]["base_sample_generator"] = "vectorized"
    p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = 1, iters
    a = sg.Agent(p, sg.make_env(p))
    if erv is not None:
        a.epistimic_random_vector = erv
    return a, p
torch.manual_seed(11)
agent, p = make(Ns)
small, _ = make(sub, agent.epistimic_random_vector[:, :, :sub].clone())
lib = _lib.load()
if os.environ.get("KEEP_ROOT") == "1":
    small.debug_keep_root = True
lib.gpmpc_joint_pin_path(pin)
x0 = np.array(p["env"]["start"], dtype=np.float64)
u_h = np.zeros((H, 2)); u_h[:, 0] = 0.05 * np.sin(2 * np.pi * np.arange(H) / H)
x_h = np.tile(x0, (H, Ns))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for k in range(iters):
        agent.train_hallucinated_dynGP(k); small.train_hallucinated_dynGP(k)
        xs = x_h.reshape(H, Ns, 4)[:, :sub].reshape(H, sub * 4)
        gv, yg, ug = agent.dyn_fg_jacobians(agent.get_batch_x_hat(x_h, u_h), k)
        pa = lib.gpmpc_joint_last_path()
        ca = agent.model_i_call.n_cached_rows
        sv, sy, su = small.dyn_fg_jacobians(small.get_batch_x_hat(xs, u_h), k)
        ps, cs = lib.gpmpc_joint_last_path(), small.model_i_call.n_cached_rows
        m_eq = bool(torch.equal(agent.model_i_call.mean[:sub], small.model_i_call.mean))
        v_eq = bool(torch.equal(agent.model_i_call.variance[:sub], small.model_i_call.variance))
        d = np.abs(gv[:sub] - sv)
        print(f"k={k}: paths {pa}/{ps} cached rows {ca}/{cs}: mean equal {m_eq} variance equal {v_eq} samples equal {bool(np.array_equal(gv[:sub], sv))} "
              f"(max |diff| {d.max():.2e}, samples that differ {np.nonzero(d.reshape(sub, -1).max(axis=1))[0].tolist()})", flush=True)
        mean_next = gv[:, :, :, 0].mean(axis=0).T
        x_h = np.tile(np.vstack([x0[None, :], mean_next[:-1]]), (1, Ns))
