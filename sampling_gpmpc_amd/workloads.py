"""Shipped configurations and the synthetic inputs of SURVEY.md section 8d (what bench.py, the tools and the tests run).

The three YAMLs under ``params/`` are the reference's ``params_pendulum1D_samples``, ``params_car_residual`` and
``params_car_residual_fs`` (the only ones consistent with the reference's current ``src/``).
The input sequences stand in for the reference's ``data.pkl`` solution, which needs acados to produce
(reference ``benchmarking/simulate_forward_sampling_car.py:91-98``).
"""
from __future__ import annotations

import copy
import os

import numpy as np
import yaml

PARAMS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "params")


def load_params(name: str) -> dict:
    with open(os.path.join(PARAMS_DIR, name + ".yaml")) as f:
        return yaml.safe_load(f)


def fs_params(name, Ns, H_traj, nograd=None, feedback=None, beta=None) -> dict:
    """Params prepared the way the reference forward-sampling script needs them (``optimizer.H == 1``, base samples
    indexed ``[H_idx][1]``: reference ``params_car_residual_fs.yaml:71,75,88``)."""
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


def closed_loop_params(name, Ns, H, n_mpc=1, n_sqp=4) -> dict:
    """Params of the closed-loop SQP use (mode J, reference ``src/solver.py:56-94``) at a given size, otherwise as
    shipped (in particular ``Dyn_gp_jitter``: 1e-20 for the car, i.e. the eigendecomposition root)."""
    p = copy.deepcopy(load_params(name))
    p["agent"]["num_dyn_samples"], p["optimizer"]["H"] = Ns, H
    p["agent"]["true_dyn_as_sample"] = False
    p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = n_mpc, n_sqp
    return p


def synthetic_u_ff(nu: int, H: int) -> np.ndarray:
    """Open-loop input sequence (H, nu): pendulum ``linspace(-1, 1, H)``; car ``delta = 0.05 sin(2 pi t / H)``, ``a = 0``."""
    t = np.arange(H)
    if nu == 1:
        return np.linspace(-1, 1, H).reshape(H, 1)
    return np.stack([0.05 * np.sin(2 * np.pi * t / H), np.zeros(H)], axis=1)


# algorithmic work per unit (unit = one sampled trajectory-step = one sample, one time step, all
# g_ny outputs; append-row algorithm for mode R, from-scratch algebra for mode J)
def flop_mode_r(g_ny, T, n_real_obs, N_r, H) -> float:
    """Mean FP64 FLOP per trajectory-step of an H-step re-conditioned rollout (SURVEY 8d formula for mode R)."""
    tot = 0.0
    for t in range(H):
        n = n_real_obs + T * t
        tot += g_ny * (T * n * n + 2 * T * n + 2 * T * T * n + 14 * (N_r + t) + 4 * T * T * (N_r + t) + 3 * T ** 3)
    return tot / H


def flop_mode_j(g_ny, T, n_real_obs, H, k) -> float:
    """FP64 FLOP of ONE joint draw of one sample (all outputs) at SQP iteration k: n = N_r,obs + k H T, m = H T."""
    m = H * T
    n = n_real_obs + k * H * T
    return g_ny * (n ** 3 / 3 + 2 * n * n + m * n * n + m * m * n + m ** 3 / 3 + 2 * m * m + 2 * m * n)


def flop_mode_j_cached_rows(g_ny, n_real_obs, n_cached) -> float:
    """FLOP of `flop_mode_j` that a call with `n_cached` factor rows in the caller's cache does not execute: the
    factorisation of those rows (columns up to each row's own)."""
    return g_ny * ((n_real_obs + n_cached) ** 3 - n_real_obs ** 3) / 3


def flop_chol(g_ny, n) -> float:
    """FLOP of the Cholesky factorisation of an n x n block per sample (all outputs)."""
    return g_ny * n ** 3 / 3


def flop_mode_i(g_ny, N_r) -> float:
    return g_ny * (14 * N_r + N_r * N_r + 2 * N_r + 10)


def flop_mode_i_grid_root(g_ny, n0, n1) -> float:
    """FP64 FLOP per trajectory-step of the mode-I algorithm the grid-root kernel EXECUTES (csrc/rollout_indep.hip; DESIGN
    4.2), per output: the separable products  A = Qa^T ea (n0^2 FMA), B = Qb^T eb (n1^2), mu = sum_c B_c sum_a m1_ac A_a
    and k^T (K+s2 I)^-1 k = sum_c B_c^2 sum_a m2_ac A_a^2 (2 n0 n1 + 2 n1 FMA) = 2 FLOP each; the kernel factors by the
    equispaced-axis recurrence: 4 exponentials (22 FLOP each: Cody-Waite + degree-11 polynomial) and 2 (n0 + n1)
    multiplications; ~10 for variance floor, root, sample and clip.  Car (5 x 9): 3 x 554 = 1.66e3 (SURVEY 8d counts the
    triangular algorithm it replaces: 8.3e3)."""
    return g_ny * (2.0 * (n0 * n0 + n1 * n1 + 2 * n0 * n1 + 2 * n1) + 4 * 22 + 2 * (n0 + n1) + 10)


def min_hbm_bytes(nx, g_ny, T) -> int:
    """Unavoidable HBM bytes per trajectory-step with the factor on chip: state in/out, z in, y out."""
    return 8 * (2 * nx + 2 * g_ny * T)
