"""Constraint-tightening tables (host-side constants; reference ``src/utils/reachable_set.py:3-38``).

``tilde_eps_list[k] = [sqrt(diag(P^-1)) B_k (nx), sqrt(diag(K P^-1 K^T)) B_k (nu), B_k]`` with
``B_k = (dyn_eps + w_bound) * c_P * V_k[k-1] * sum_{j<k} L^j``, ``B_0 = 0`` and ``ci_list[k-1] = B_k``;
``c_P`` sums the square roots of the first (up to three) diagonal entries of P, as the reference's
``np.diag(P[:3][:3])`` does.
"""
from __future__ import annotations

import numpy as np


def get_reachable_set_ball(params, V_k, eps_vec=None):
    H = params["optimizer"]["H"]
    V_k = np.asarray(V_k, dtype=np.float64)
    assert V_k.shape[0] == H + 1
    tt = params["optimizer"]["terminal_tightening"]
    P = np.array(tt["P"], dtype=np.float64)
    K = np.array(tt["K"], dtype=np.float64)
    tight = params["agent"]["tight"]
    L = tight["Lipschitz"]
    var_eps = tight["dyn_eps"] + tight["w_bound"]
    root_diag = np.sqrt(np.diag(P)[: min(3, P.shape[0])])
    if eps_vec is not None:
        B_d_norm = (np.dot(root_diag, eps_vec) / var_eps) * V_k
    else:
        B_d_norm = np.sum(root_diag) * V_k
    P_inv = np.linalg.inv(P)
    sx = np.sqrt(np.diag(P_inv))
    su = np.sqrt(np.diag(K @ P_inv @ K.T))
    geom = np.concatenate([[0.0], [np.sum(np.power(L, np.arange(0, k))) for k in range(1, H + 1)]])
    B = np.concatenate([[0.0], var_eps * B_d_norm[:H] * geom[1:]])
    tilde_eps_list = [np.concatenate([sx * b, su * b, [b]]) for b in B]
    ci_list = [float(b) for b in B[1:]]
    return tilde_eps_list, ci_list
