"""On-disk formats of the forward-sampling campaign, write-compatible with the reference so
that its post-processing scripts can consume GPU results:

* ``data_X_traj_<idx>.pkl`` - reference ``benchmarking/simulate_forward_sampling_car.py:157-161``: a pickled float64
  numpy array ``(Ns, nx, H+1)``; consumed by ``benchmarking/generate_convex_hull.py:76-83`` with ``pickle.load`` and
  concatenated over files along axis 0.
* ``data_epistemic_vector_<idx>.pkl`` - the optional base-sample input of the same script (``:78-80``): the pickled
  ``epistimic_random_vector`` tensor.
* ``data.pkl`` - reference ``src/visu.py:497-517``: the closed-loop record, a dict of lists (``state_traj``,
  ``input_traj``, ``mean_state_traj``, ``true_state_traj``, ``physical_state_traj``, ``solver_time``,
  ``gp_model_after_solve_train_X`` / ``_Y``) plus ``tilde_eps_list`` / ``ci_list``; read back by ``visu.py:519-521`` and
  by both benchmarking harnesses (``simulate_forward_sampling_car.py:91-98`` takes ``input_traj[-1]`` from it).
* ``X_traj_list_<k>.pkl`` - reference ``benchmarking/simulate_true_reachable_set.py:263-273``: a list of ``H+1``
  tensors ``(N, g_ny, 1, nx+nu)`` (state in ``[..., :nx]``, applied input in ``[..., nx:]``, the state replicated over the
  ``g_ny`` batch axis), consumed by ``extra/cdc_plt.py:155-176``.

The reference pickles with ``dill`` (imported as ``pickle``); plain numpy arrays / torch CPU tensors pickled with either
module are mutually readable, so ``dill`` is used when importable and the standard library otherwise.
"""
from __future__ import annotations

import os

import numpy as np
import torch

try:                                    # the reference: `import dill as pickle`
    import dill as _pickle
except Exception:                       # pragma: no cover
    import pickle as _pickle


def x_traj_path(save_dir: str, epistemic_idx: int) -> str:
    return os.path.join(save_dir, f"data_X_traj_{epistemic_idx}.pkl")


def save_x_traj(save_dir: str, epistemic_idx: int, X_traj) -> str:
    """Write the reachable tube of one forward-sampling job exactly as the reference does."""
    X = np.ascontiguousarray(X_traj.detach().cpu().numpy() if torch.is_tensor(X_traj) else X_traj, dtype=np.float64)
    if X.ndim != 3:
        raise ValueError("X_traj must have shape (Ns, nx, H+1)")
    os.makedirs(save_dir, exist_ok=True)
    path = x_traj_path(save_dir, epistemic_idx)
    with open(path, "wb") as f:
        _pickle.dump(X, f)
    return path


def load_x_traj(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        return _pickle.load(f)


def merge_x_traj(save_dir: str, indices) -> np.ndarray:
    """Concatenate job files over the sample axis (what generate_convex_hull.py does before taking hulls)."""
    return np.concatenate([load_x_traj(x_traj_path(save_dir, i)) for i in indices], axis=0)


def save_epistemic_vector(save_dir: str, epistemic_idx: int, erv: torch.Tensor) -> str:
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, f"data_epistemic_vector_{epistemic_idx}.pkl")
    with open(path, "wb") as f:
        _pickle.dump(erv.detach().cpu(), f)
    return path


def load_epistemic_vector(path: str) -> torch.Tensor:
    with open(path, "rb") as f:
        return _pickle.load(f)


DATA_PKL_KEYS = ("state_traj", "input_traj", "mean_state_traj", "true_state_traj", "physical_state_traj", "solver_time",
                 "gp_model_after_solve_train_X", "gp_model_after_solve_train_Y", "tilde_eps_list", "ci_list")


def save_data_pkl(save_dir: str, data: dict) -> str:
    """Write the closed-loop record as reference ``src/visu.py:497-517`` does (same keys, ``data.pkl``)."""
    missing = [k for k in DATA_PKL_KEYS if k not in data]
    if missing:
        raise ValueError(f"data.pkl needs the keys {missing}")
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, "data.pkl")
    with open(path, "wb") as f:
        _pickle.dump({k: data[k] for k in DATA_PKL_KEYS}, f)
    return path


def load_data_pkl(path: str) -> dict:
    with open(path, "rb") as f:
        return _pickle.load(f)


def x_traj_list_path(save_dir: str, k: int) -> str:
    return os.path.join(save_dir, f"X_traj_list_{k}.pkl")


def save_x_traj_list(save_dir: str, k: int, X_traj, U, g_ny: int) -> str:
    """Write a true-reachable-set file from the tube ``X_traj (N, nx, H+1)`` and the applied inputs ``U (H, nu)`` in the
    layout of reference ``simulate_true_reachable_set.py:110-116, 189-196, 263-273``: element i of the list is the
    ``(N, g_ny, 1, nx+nu)`` tensor of step i (input columns of the last element stay zero, as in the reference)."""
    X = torch.as_tensor(np.asarray(X_traj.detach().cpu() if torch.is_tensor(X_traj) else X_traj), dtype=torch.float64)
    U = torch.as_tensor(np.asarray(U), dtype=torch.float64)
    N, nx, H1 = X.shape
    if U.shape[0] != H1 - 1:
        raise ValueError("U must have one row per step (H rows for an (N, nx, H+1) tube)")
    nu = U.shape[1]
    out = []
    for i in range(H1):
        t = torch.zeros(N, g_ny, 1, nx + nu, dtype=torch.float64)
        t[:, :, 0, :nx] = X[:, None, :, i]
        if i < H1 - 1:
            t[:, :, 0, nx:] = U[i]
        out.append(t)
    os.makedirs(save_dir, exist_ok=True)
    path = x_traj_list_path(save_dir, k)
    with open(path, "wb") as f:
        _pickle.dump(out, f)
    return path


def load_x_traj_list(path: str):
    with open(path, "rb") as f:
        return _pickle.load(f)
