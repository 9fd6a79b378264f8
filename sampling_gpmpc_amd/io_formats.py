"""On-disk formats of the forward-sampling campaign, write-compatible with the reference so
that its post-processing scripts can consume GPU results:

* ``data_X_traj_<idx>.pkl`` - reference ``benchmarking/simulate_forward_sampling_car.py:157-161``: a pickled float64
  numpy array ``(Ns, nx, H+1)``; consumed by ``benchmarking/generate_convex_hull.py:76-83`` with ``pickle.load`` and
  concatenated over files along axis 0.
* ``data_epistemic_vector_<idx>.pkl`` - the optional base-sample input of the same script (``:78-80``): the pickled
  ``epistimic_random_vector`` tensor.

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
