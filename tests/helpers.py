"""Shared helpers for the test-suite (config loading, synthetic inputs of SURVEY.md section 8d)."""
import copy
import os

import numpy as np
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
PARAMS = os.path.join(REPO, "sampling_gpmpc_amd", "params")


def load_params(name):
    with open(os.path.join(PARAMS, name + ".yaml")) as f:
        return yaml.safe_load(f)


def fs_params(name, Ns, H_traj, nograd=None, feedback=None, beta=None):
    """Params prepared the way the reference forward-sampling script needs them (H == 1, [H_idx][1] indexing)."""
    p = copy.deepcopy(load_params(name))
    p["common"]["use_cuda"] = False
    p["agent"]["num_dyn_samples"] = Ns
    p["agent"]["true_dyn_as_sample"] = False
    p["optimizer"]["H"] = 1
    p["common"]["num_MPC_itrs"] = H_traj
    p["optimizer"]["SEMPC"]["max_sqp_iter"] = 2
    if nograd is not None:
        p["env"]["use_model_without_derivatives"] = bool(nograd)
    if feedback is not None:
        p["agent"]["feedback"]["use"] = bool(feedback)
    if beta is not None:
        p["agent"]["Dyn_gp_beta"] = float(beta)
    return p


def synthetic_u_ff(nu, H):
    t = np.arange(H)
    if nu == 1:
        return np.linspace(-1, 1, H).reshape(H, 1)
    return np.stack([0.05 * np.sin(2 * np.pi * t / H), np.zeros(H)], axis=1)
