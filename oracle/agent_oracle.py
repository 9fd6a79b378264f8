"""CPU oracle for the Agent-level part of the rollout hot path.  TEST INFRASTRUCTURE ONLY (see gp_oracle.py).

Restates, op for op in torch CPU FP64, what the reference does around the GP algebra:

* environments .......... reference ``src/environments/pendulum1D.py`` / ``car_model_residual.py``
* Agent ................. reference ``src/agent.py`` (line ranges cited per method)
* forward-sampling loop . reference ``benchmarking/simulate_forward_sampling_car.py:108-138``
* true-reachable loop ... reference ``benchmarking/simulate_true_reachable_set.py:179-258``

All of this part IS pinned: ``tests/golden/make_goldens.py`` runs the reference's own ``Agent`` / environment
code (gpytorch replaced by an import stub whose model class delegates to ``gp_oracle.OracleGP``) and
``tests/test_oracle_golden.py`` compares this file against the captured tensors.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from .gp_oracle import F64, GPHyper, OracleGP, OracleSemantics


# ----------------------------------------------------------------------------------------------------------------
# environments (only the maps on the path)
# ----------------------------------------------------------------------------------------------------------------
class OraclePendulum1D:
    """reference ``src/environments/pendulum1D.py`` (class Pendulum)."""
    name = "Pendulum1D"

    def __init__(self, params):
        self.params = params
        self.nx, self.nu = params["agent"]["dim"]["nx"], params["agent"]["dim"]["nu"]
        self.g_ny = params["agent"]["g_dim"]["ny"]
        self.g_nx, self.g_nu = params["agent"]["g_dim"]["nx"], params["agent"]["g_dim"]["nu"]
        self.pad_g = [0, 1, 3]                      # :15
        self.g_idx_inputs = [0, 2]                  # :16
        self.B_d = torch.tensor([0.0, 1.0], dtype=F64).reshape(self.nx, self.g_ny)   # :26-28

    def unknown_dyn(self, xu):                      # :127-135
        l, g = self.params["env"]["params"]["l"], self.params["env"]["params"]["g"]
        dt = self.params["optimizer"]["dt"]
        return -g * torch.sin(xu[:, [0]]) * dt / l + xu[:, [1]] * dt

    def get_prior_data(self, x_hat):                # :58-84
        l, g = self.params["env"]["params"]["l"], self.params["env"]["params"]["g"]
        dt = self.params["optimizer"]["dt"]
        y = torch.zeros((self.g_ny, x_hat.shape[0], 1 + self.g_nx + self.g_nu), dtype=F64)
        y[0, :, 0] = self.unknown_dyn(x_hat)[:, 0]
        y[0, :, 1] = (-g * torch.cos(x_hat[:, 0]) / l) * dt
        y[0, :, 2] = torch.ones(x_hat.shape[0], dtype=F64) * dt
        return y

    def initial_training_data(self):                # :30-56
        p = self.params
        x1 = torch.linspace(p["optimizer"]["x_min"][0], p["optimizer"]["x_max"][0], p["env"]["n_data_x"], dtype=F64)
        u = torch.linspace(p["optimizer"]["u_min"][0], p["optimizer"]["u_max"][0], p["env"]["n_data_u"], dtype=F64)
        X1, U = torch.meshgrid(x1, u, indexing="ij")
        X = torch.hstack([X1.reshape(-1, 1), U.reshape(-1, 1)])
        Y = self.get_prior_data(X)
        if not p["env"]["train_data_has_derivatives"]:
            Y[:, :, 1:] = torch.nan
        return X, Y

    def known_dyn(self, xu):                        # :172-188   xu (ns, nx, H, nx+nu) -> (ns, nx, H)
        dt = self.params["optimizer"]["dt"]
        th, om = xu[:, [0], :, 0], xu[:, [0], :, 1]
        return torch.cat([th + om * dt, om], 1)

    def get_f_known_jacobian(self, xu):             # :137-163
        ns, nH = xu.shape[0], xu.shape[2]
        dt = self.params["optimizer"]["dt"]
        df = torch.zeros((ns, self.nx, nH, 1 + self.nx + self.nu), dtype=F64)
        df[:, :, :, 0] = self.known_dyn(xu)
        df[:, 0, :, 1] = 1.0
        df[:, 0, :, 2] = dt
        df[:, 1, :, 2] = 1.0
        return df

    def get_g_xu_hat(self, xu_hat):                 # :165-170
        return xu_hat[:, 0:self.g_ny, :, self.g_idx_inputs]

    def transform_sensitivity(self, dg, xu_hat):    # :240-241
        return dg

    def discrete_dyn(self, xu):                     # :115-125   xu (1, nx+nu)
        f = self.known_dyn(xu.reshape(1, 1, 1, -1).tile((1, self.nx, 1, 1)))[0, :, :]
        g = self.unknown_dyn(xu[:, self.g_idx_inputs]).transpose(0, 1)
        return f + self.B_d @ g


class OracleCarResidual:
    """reference ``src/environments/car_model_residual.py`` (class CarKinematicsModel, alias bicycle_Bdx)."""
    name = "bicycle_Bdx"

    def __init__(self, params):
        self.params = params
        self.nx, self.nu = params["agent"]["dim"]["nx"], params["agent"]["dim"]["nu"]
        self.g_ny = params["agent"]["g_dim"]["ny"]
        self.g_nx, self.g_nu = params["agent"]["g_dim"]["nx"], params["agent"]["g_dim"]["nu"]
        self.pad_vg = [0, 1, 3]                     # :14
        self.pad_g = [0, 3, 4, 5]                   # :15
        self.g_idx_inputs = [2, 4]                  # :16
        self.B_d = torch.eye(self.nx, self.g_ny, dtype=F64)      # :26

    def unknown_dyn(self, xu):                      # :167-182
        lf, lr = self.params["env"]["params"]["lf"], self.params["env"]["params"]["lr"]
        dt = self.params["optimizer"]["dt"]
        phi, delta = xu[:, [0]], xu[:, [1]]
        beta = torch.atan(torch.tan(delta) * lr / (lr + lf))
        return torch.hstack([torch.cos(phi + beta) * dt, torch.sin(phi + beta) * dt, torch.sin(beta) * dt / lr])

    def get_prior_data(self, xu):                   # :62-99
        lf, lr = self.params["env"]["params"]["lf"], self.params["env"]["params"]["lr"]
        dt = self.params["optimizer"]["dt"]
        phi, delta = xu[:, 0], xu[:, 1]
        g = self.unknown_dyn(xu)
        y = torch.zeros((self.g_ny, xu.shape[0], 1 + self.g_nx + self.g_nu), dtype=F64)
        y[0, :, 0], y[1, :, 0], y[2, :, 0] = g[:, 0], g[:, 1], g[:, 2]
        beta_in = (lr * torch.tan(delta)) / (lf + lr)
        beta = torch.atan(beta_in)
        term = ((lr / (torch.cos(delta) ** 2)) / (lf + lr)) / (1 + beta_in ** 2)
        y[0, :, 1] = -torch.sin(phi + beta) * dt
        y[0, :, 2] = -torch.sin(phi + beta) * dt * term
        y[1, :, 1] = torch.cos(phi + beta) * dt
        y[1, :, 2] = torch.cos(phi + beta) * dt * term
        y[2, :, 2] = torch.cos(beta) * dt * term / lr
        return y

    def initial_training_data(self):                # :29-60
        p = self.params
        phi = torch.linspace(p["optimizer"]["x_min"][2], p["optimizer"]["x_max"][2], p["env"]["n_data_x"], dtype=F64)
        delta = torch.linspace(p["optimizer"]["u_min"][0], p["optimizer"]["u_max"][0], p["env"]["n_data_u"], dtype=F64)
        Phi, Delta = torch.meshgrid(phi, delta, indexing="ij")
        X = torch.hstack([Phi.reshape(-1, 1), Delta.reshape(-1, 1)])
        Y = self.get_prior_data(X)
        if not p["env"]["train_data_has_derivatives"]:
            Y[:, :, 1:] = torch.nan
        return X, Y

    def known_dyn(self, xu):                        # :139-161
        dt = self.params["optimizer"]["dt"]
        return torch.cat([xu[:, [0], :, 0], xu[:, [0], :, 1], xu[:, [0], :, 2],
                          xu[:, [0], :, 3] + xu[:, [0], :, 5] * dt], 1)

    def get_f_known_jacobian(self, xu):             # :101-130
        ns, nH = xu.shape[0], xu.shape[2]
        dt = self.params["optimizer"]["dt"]
        df = torch.zeros((ns, self.nx, nH, 1 + self.nx + self.nu), dtype=F64)
        df[:, :, :, 0] = self.known_dyn(xu)
        for i in range(4):
            df[:, i, :, 1 + i] = 1.0
        df[:, 3, :, 6] = dt
        return df

    def get_g_xu_hat(self, xu_hat):                 # :132-137
        return xu_hat[:, 0:self.g_ny, :, self.g_idx_inputs]

    def transform_sensitivity(self, dg, xu_hat):    # :211-224
        ns, nH = dg.shape[0], dg.shape[2]
        out = torch.zeros((ns, self.g_ny, nH, 4), dtype=F64)
        out[:, :, :, self.pad_vg] = xu_hat[:, 0:3, :, [3]] * dg
        out[:, :, :, 2] = dg[:, :, :, 0]
        return out

    def discrete_dyn(self, xu):                     # :188-196
        f = self.known_dyn(xu.reshape(1, 1, 1, -1).tile((1, self.nx, 1, 1)))[0, :, :]
        g = self.unknown_dyn(xu[:, self.g_idx_inputs]).transpose(0, 1)
        B = xu[:, [3]] * torch.eye(self.nx, self.g_ny, dtype=F64)
        return f + B @ g


def make_oracle_env(params):
    return {"Pendulum1D": OraclePendulum1D, "bicycle_Bdx": OracleCarResidual}[params["env"]["dynamics"]](params)


# ----------------------------------------------------------------------------------------------------------------
# base samples (a2) and tightenings (f3)
# ----------------------------------------------------------------------------------------------------------------
def random_vector_within_bounds(params, g_ny: int, T: int) -> torch.Tensor:
    """reference ``src/agent.py:76-104``: whole (g_ny,H,T) vectors are rejected unless every entry is in [-b, b];
    one ``torch.normal`` call per candidate on the global CPU generator (call-for-call, so the stream matches)."""
    H = params["optimizer"]["H"]
    n_dyn = params["agent"]["num_dyn_samples"]
    beta = params["agent"]["Dyn_gp_beta"]
    n_mpc = params["common"]["num_MPC_itrs"]
    n_itrs = params["optimizer"]["SEMPC"]["max_sqp_iter"]
    out = torch.empty(n_mpc, n_itrs, n_dyn, g_ny, H, T, dtype=F64)
    for j in range(n_mpc):
        for i in range(n_itrs):
            k = 0
            while k < n_dyn:
                w = torch.normal(0, 1, size=(1, g_ny, H, T), dtype=F64)
                if torch.all(w >= -beta) and torch.all(w <= beta):
                    out[j, i, k] = w[0]
                    k += 1
    return out


def get_reachable_set_ball(params, V_k):
    """reference ``src/utils/reachable_set.py:3-38`` (eps_vec=None branch)."""
    H = params["optimizer"]["H"]
    assert V_k.shape[0] == H + 1
    P = np.array(params["optimizer"]["terminal_tightening"]["P"])
    L = params["agent"]["tight"]["Lipschitz"]
    var_eps = params["agent"]["tight"]["dyn_eps"] + params["agent"]["tight"]["w_bound"]
    B_d_norm = np.sum(np.sqrt(np.diag(P[:3][:3]))) * V_k
    P_inv = np.linalg.inv(P)
    K = np.array(params["optimizer"]["terminal_tightening"]["K"])
    sx, su = np.sqrt(np.diag(P_inv)), np.sqrt(np.diag(K @ P_inv @ K.T))
    tilde, ci = [np.concatenate([sx * 0, su * 0, [0]])], []
    for stage in range(1, H + 1):
        B = var_eps * B_d_norm[stage - 1] * np.sum(np.power(L, np.arange(0, stage)))
        tilde.append(np.concatenate([sx * B, su * B, [B]]))
        ci.append(B)
    return tilde, ci


# ----------------------------------------------------------------------------------------------------------------
# Agent (only the hot-path methods)
# ----------------------------------------------------------------------------------------------------------------
class OracleAgent:
    def __init__(self, params, env, epistimic_random_vector: Optional[torch.Tensor] = None,
                 semantics: Optional[OracleSemantics] = None):
        """reference ``src/agent.py:18-74``.  Base samples are an *input* (generated by the caller with
        ``random_vector_within_bounds`` for reference-exact runs).  ``semantics``: the switchable gpytorch behaviours
        (``gp_oracle.OracleSemantics``; defaults = what the HIP kernels implement)."""
        self.semantics = semantics if semantics is not None else OracleSemantics()
        self.params, self.env_model = params, env
        ag = params["agent"]
        self.g_nx, self.g_nu, self.g_ny = ag["g_dim"]["nx"], ag["g_dim"]["nu"], ag["g_dim"]["ny"]
        self.ns = ag["num_dyn_samples"]
        self.nx, self.nu = ag["dim"]["nx"], ag["dim"]["nu"]
        self.in_dim_x = self.g_nx + self.g_nu
        self.in_dim_y = 1 if params["env"]["use_model_without_derivatives"] else 1 + self.in_dim_x
        self.batch_shape = torch.Size([self.ns, self.g_ny])
        self.Hallcinated_X_train = torch.empty(self.ns, self.g_ny, 0, self.in_dim_x, dtype=F64)
        self.Hallcinated_Y_train = torch.empty(self.ns, self.g_ny, 0, self.in_dim_y, dtype=F64)
        self.Dyn_gp_X_train, self.Dyn_gp_Y_train = env.initial_training_data()
        if self.in_dim_y == 1:
            self.Dyn_gp_Y_train = self.Dyn_gp_Y_train[:, :, [0]]
        # real_data_batch :204-214 (tiled Ns times, as the reference does)
        self.Dyn_gp_X_train_batch = torch.tile(self.Dyn_gp_X_train, dims=(self.ns, self.g_ny, 1, 1))
        self.Dyn_gp_Y_train_batch = torch.tile(self.Dyn_gp_Y_train, dims=(self.ns, 1, 1, 1))
        self.epistimic_random_vector = epistimic_random_vector
        self.model_i = None
        self.mpc_iter = 0
        if "terminal_tightening" in params["optimizer"]:
            self.tilde_eps_list, self.ci_list = get_reachable_set_ball(params, np.ones(params["optimizer"]["H"] + 1))

    def mpc_iteration(self, i):
        self.mpc_iter = i

    def update_current_state(self, state):          # :157-162
        self.current_state = state
        self.current_location = state[: self.nx]

    def get_next_to_go_loc(self):                   # :529-530
        return np.array([2])

    def concatenate_real_hallucinated_data(self):   # :274-281
        return (torch.concat([self.Dyn_gp_X_train_batch, self.Hallcinated_X_train], dim=2),
                torch.concat([self.Dyn_gp_Y_train_batch, self.Hallcinated_Y_train], dim=2))

    def train_hallucinated_dynGP(self, sqp_iter, use_model_without_derivatives=False):   # :216-272
        if use_model_without_derivatives:
            data_X, data_Y = self.Dyn_gp_X_train_batch, self.Dyn_gp_Y_train_batch[:, :, :, [0]]
        else:
            data_X, data_Y = self.concatenate_real_hallucinated_data()
        hyper = GPHyper.from_params(self.params, use_grad=not use_model_without_derivatives)
        hyper.semantics = self.semantics
        if not use_model_without_derivatives and self.in_dim_y == 1:
            # the reference would build a T=1+D likelihood against T=1 labels here; not a runnable combination
            raise RuntimeError("in_dim_y == 1 requires use_model_without_derivatives=True")
        self.model_i = OracleGP(data_X, data_Y, hyper)
        if sqp_iter == 0:                            # reset AFTER the model was built (quirk, :261-272)
            self.Hallcinated_X_train = torch.empty(self.ns, self.g_ny, 0, self.in_dim_x, dtype=F64)
            self.Hallcinated_Y_train = torch.empty(self.ns, self.g_ny, 0, self.in_dim_y, dtype=F64)

    def train_forward_sampling_dynGP(self):           # :283-329  real ++ forward-sampling ++ hallucinated data
        data_X = torch.concat([self.Dyn_gp_X_train_batch, self.FS_X_train_batch, self.Hallcinated_X_train], dim=2)
        data_Y = torch.concat([self.Dyn_gp_Y_train_batch, self.FS_Y_train_batch, self.Hallcinated_Y_train], dim=2)
        hyper = GPHyper.from_params(self.params, use_grad=True)
        hyper.semantics = self.semantics
        self.model_i = OracleGP(data_X, data_Y, hyper)

    def prepare_dynamics_set(self, X_soln, U_soln, X_kp1, base_samples=None, rng=None):   # :331-443
        """Forward sampling with rejection.  ``base_samples``: list of (Ns, g_ny, 1, T) tensors, one per propagation
        step (the reference draws them internally with ``.sample()``); ``rng``: numpy RandomState for the survivor
        choice (the reference uses the global ``np.random``).  g_ny == 1 only, like the reference (its ``squeeze`` /
        ``.t()`` arithmetic does not type-check for more outputs)."""
        n_sample = self.ns
        tight = self.params["agent"]["tight"]
        B_d_norm = np.sqrt(self.params["optimizer"]["terminal_tightening"]["P"][1][1])
        var_eps = (tight["dyn_eps"] + tight["w_bound"]) * B_d_norm
        rng = np.random if rng is None else rng
        self.FS_X_train_batch = torch.empty(n_sample, self.g_ny, 0, self.in_dim_x, dtype=F64)
        self.FS_Y_train_batch = torch.empty(n_sample, self.g_ny, 0, self.in_dim_y, dtype=F64)
        X_soln = torch.as_tensor(X_soln, dtype=F64).reshape(X_soln.shape[0], n_sample, self.nx)
        X_kp1 = torch.as_tensor(X_kp1, dtype=F64).transpose(0, 1)
        U_soln = torch.as_tensor(U_soln, dtype=F64)
        samples_left = torch.prod(torch.abs(X_soln[1, :, :] - X_kp1) - var_eps < 0, dim=1)
        xu_hat = torch.tile(torch.cat([X_kp1, U_soln[[1]]], dim=-1), dims=(n_sample, self.nx, 1, 1))
        self.rejection_trace = [samples_left.clone()]
        for i in range(1, X_soln.shape[0] - 1):
            g_xu_hat = self.env_model.get_g_xu_hat(xu_hat)
            z = None if base_samples is None else base_samples[i - 1]
            Y_sample = self.model_i(g_xu_hat).sample(z)
            g_val = Y_sample[:, :, :].squeeze()[:, : self.g_ny]
            f_val = self.env_model.known_dyn(xu_hat).squeeze()
            x_next = f_val + torch.matmul(self.env_model.B_d, g_val.t()).t()
            samples_left = samples_left * torch.prod(torch.abs(X_soln[i + 1, :, :] - x_next) - self.ci_list[i] < 0, dim=1)
            self.rejection_trace.append(samples_left.clone())
            if i == X_soln.shape[0] - 2:
                break
            self.FS_X_train_batch = torch.cat([self.FS_X_train_batch, g_xu_hat], dim=2)
            Y_sample = Y_sample.clone()
            Y_sample[:, :, :, 1:] = float("nan")
            self.FS_Y_train_batch = torch.cat([self.FS_Y_train_batch, Y_sample], dim=2)
            self.train_forward_sampling_dynGP()
            xu_hat = torch.cat([torch.stack([x_next] * self.nx, dim=1)[:, :, None, :],
                                torch.tile(U_soln[[i + 1]], dims=(n_sample, self.nx, 1, 1))], dim=-1)
        if torch.sum(samples_left) > 0:
            n_rep = int(torch.sum(samples_left == 0).item())
            remaining = torch.arange(n_sample)[samples_left > 0].numpy()
            dead = samples_left == 0
            self.Hallcinated_X_train[dead] = self.Hallcinated_X_train[rng.choice(remaining, n_rep).tolist()]
            self.Hallcinated_Y_train[dead] = self.Hallcinated_Y_train[rng.choice(remaining, n_rep).tolist()]
        self.train_hallucinated_dynGP(sqp_iter=self.params["optimizer"]["SEMPC"]["max_sqp_iter"])

    def update_hallucinated_Dyn_dataset(self, newX, newY):   # :164-202
        min_distance = self.params["agent"]["Dyn_gp_min_data_dist"]
        X_cond, _ = self.concatenate_real_hallucinated_data()
        dist = newX[:, :, None, :, :] - X_cond[:, :, :, None, :]
        dist_norm = torch.linalg.vector_norm(dist, dim=-1)
        filt = torch.any(dist_norm <= min_distance, dim=2)
        filt_y = filt.unsqueeze(-1).tile(1, 1, 1, self.in_dim_y)
        newY_f = newY.clone()
        newY_f[filt_y] = torch.nan
        filt_all = torch.any(torch.all(filt, dim=0), dim=0)
        self.Hallcinated_X_train = torch.cat([self.Hallcinated_X_train, newX[:, :, filt_all == False, :]], 2)
        self.Hallcinated_Y_train = torch.cat([self.Hallcinated_Y_train, newY_f[:, :, filt_all == False, :]], 2)

    def get_batch_x_hat_u_diff(self, x_h, u_h):     # :480-501
        H = self.params["optimizer"]["H"]
        x_h, u_h = torch.as_tensor(x_h, dtype=F64), torch.as_tensor(u_h, dtype=F64)
        xb = x_h.transpose(0, 1).reshape(self.ns, self.nx, H).transpose(1, 2)
        ub = u_h.transpose(0, 1).reshape(self.ns, H, self.nu)
        ret = torch.cat([xb, ub], 2)
        return torch.stack([ret] * self.nx, dim=1)

    def get_batch_x_hat(self, x_h, u_h):            # :503-527
        H = self.params["optimizer"]["H"]
        x_h, u_h = torch.as_tensor(x_h, dtype=F64), torch.as_tensor(u_h, dtype=F64)
        xb = x_h.transpose(0, 1).reshape(self.ns, self.nx, H).transpose(1, 2)
        ub = torch.ones(self.ns, H, 1, dtype=F64) * u_h
        ret = torch.cat([xb, ub], 2)
        return torch.stack([ret] * self.nx, dim=1)

    def sample_gp(self, x_input, base_samples=None):   # :629-730
        ag = self.params["agent"]
        self.model_i_call = self.model_i(x_input)
        y = self.model_i_call.sample(base_samples=base_samples)
        y_train, x_train = self.model_i.train_targets, self.model_i.train_inputs[0]
        mean, var = self.model_i_call.mean, self.model_i_call.variance
        if ag["Dyn_gp_variance_is_zero"] >= 0.0:                                   # :646-663
            z_all = torch.all(var <= ag["Dyn_gp_variance_is_zero"], dim=-1, keepdim=True).tile(1, 1, 1, self.in_dim_y)
            num = torch.zeros_like(var)
            num[z_all] = 1
            y = num * mean + (1 - num) * y
        if ag["Dyn_gp_min_data_dist"] >= 0.0:                                      # :666-698
            m = x_input.shape[2]
            dist = x_input[:, :, None, :, :] - x_train[:, :, :, None, :]
            isnan = torch.any(torch.isnan(y_train), dim=3).unsqueeze(-1).tile(1, 1, 1, m)
            dn = torch.linalg.vector_norm(dist, dim=-1)
            dn[isnan] = float("inf")
            too_small = torch.any(dn <= ag["Dyn_gp_min_data_dist"], dim=2).unsqueeze(-1).tile(1, 1, 1, self.in_dim_y)
            _, idx = torch.min(dn, dim=2)
            A, B = y_train.shape[0], y_train.shape[1]
            i1 = torch.arange(A).view(A, 1, 1).expand(A, B, m)
            i2 = torch.arange(B).view(1, B, 1).expand(A, B, m)
            y = torch.where(too_small, y_train[i1, i2, idx, :], y)
            assert not torch.any(torch.isnan(y))
        y = torch.min(torch.max(y, mean - ag["Dyn_gp_beta"] * torch.sqrt(var)),
                      mean + ag["Dyn_gp_beta"] * torch.sqrt(var))               # :701-708
        self.model_i_samples = y
        return y

    def get_batch_gp_sensitivities(self, xu_hat, sqp_iter):   # :566-627
        ag = self.params["agent"]
        g_xu_hat = self.env_model.get_g_xu_hat(xu_hat)
        H = self.params["optimizer"]["H"]
        update = True
        if (ag["true_dyn_as_sample"] or ag["mean_as_dyn_sample"]) and self.ns == 1:
            y = torch.zeros((1, self.g_ny, H, self.in_dim_y), dtype=F64)
            update = False
        elif (ag["true_dyn_as_sample"] and ag["mean_as_dyn_sample"]) and self.ns == 2:
            y = torch.zeros((2, self.g_ny, H, self.in_dim_y), dtype=F64)
            update = False
        else:
            y = self.sample_gp(g_xu_hat, base_samples=self.epistimic_random_vector[self.mpc_iter][sqp_iter])
        if not update:
            self.model_i_call = self.model_i(g_xu_hat)
        idx = 0
        if ag["true_dyn_as_sample"]:
            t = self.env_model.get_prior_data(g_xu_hat[idx, 0, :, :])
            if self.in_dim_y == 1:
                t = t[:, :, [0]]
            y[idx, :, :, :] = t
            idx += 1
        if ag["mean_as_dyn_sample"]:
            y[[idx], :, :, :] = self.model_i_call.mean[[idx], :, :, :]
            idx += 1
        if update:
            self.update_hallucinated_Dyn_dataset(g_xu_hat, y)
        return y

    def dyn_fg_jacobians(self, xu_hat, sqp_iter, injected_sample=None):   # :532-564
        ns, nH = xu_hat.shape[0], xu_hat.shape[2]
        df = self.env_model.get_f_known_jacobian(xu_hat)
        dg = injected_sample if injected_sample is not None else self.get_batch_gp_sensitivities(xu_hat, sqp_iter)
        pad = torch.zeros(ns, self.g_ny, nH, 1 + self.nx + self.nu, dtype=F64)
        dg = self.env_model.transform_sensitivity(dg, xu_hat)
        pad[:, :, :, self.env_model.pad_g] = dg
        y = df + torch.matmul(self.env_model.B_d, pad.transpose(1, 2)).transpose(1, 2)
        return (y[:, :, :, [0]].numpy(), y[:, :, :, 1:1 + self.nx].numpy(),
                y[:, :, :, 1 + self.nx:1 + self.nx + self.nu].numpy())


# ----------------------------------------------------------------------------------------------------------------
# rollout drivers (a14)
# ----------------------------------------------------------------------------------------------------------------
def forward_sampling_rollout(agent: OracleAgent, u_ff: np.ndarray, x0=None, return_samples: bool = False):
    """reference ``benchmarking/simulate_forward_sampling_car.py:108-138``.

    ``u_ff`` (H_traj, nu) is the open-loop input sequence the reference loads from ``data.pkl``.  Requires
    ``optimizer.H == 1``, ``num_MPC_itrs >= H_traj``, ``max_sqp_iter >= 2`` exactly as the reference script does
    (base samples are indexed ``[H_idx][1]``).  Returns ``X_traj (Ns, nx, H_traj+1)`` float64 numpy.
    """
    p = agent.params
    ns, nx = agent.ns, agent.nx
    H = u_ff.shape[0]
    K = np.array(p["optimizer"]["terminal_tightening"]["K"])
    x_equi = np.array(p["env"]["goal_state"])
    x_curr = np.array(p["env"]["start"] if x0 is None else x0, dtype=np.float64)[:nx].reshape(nx)
    x_h = np.tile(x_curr, (1, ns))
    X_traj = torch.empty((ns, nx, H + 1), dtype=F64)
    Y = []
    flag = p["env"]["use_model_without_derivatives"]
    for H_idx in range(H):
        agent.train_hallucinated_dynGP(1, use_model_without_derivatives=flag)
        agent.mpc_iteration(H_idx)
        u_h = u_ff[H_idx].reshape(1, -1)
        if p["agent"]["feedback"]["use"]:
            bx = agent.get_batch_x_hat_u_diff(
                x_h, -(x_equi - x_h.reshape(1, ns, -1)) @ K.T + np.tile(u_h[:, None, :], (ns, 1)))
        else:
            bx = agent.get_batch_x_hat(x_h, u_h)
        gp_val, _, _ = agent.dyn_fg_jacobians(bx, 1)
        if return_samples:
            Y.append(agent.model_i_samples.clone())
        X_traj[:, :, H_idx] = bx[:, 0, 0, :nx]
        x_h = gp_val[:, :, 0, 0].reshape(1, -1)
    X_traj[:, :, H] = torch.tensor(gp_val[:, :, 0, 0])
    if return_samples:
        return X_traj.numpy(), torch.cat(Y, dim=2).numpy()
    return X_traj.numpy()
