"""Receding-horizon closed loop around the HIP ``Agent``.

``ClosedLoop`` stands where reference ``src/DEMPC.py`` stands (``receding_horizon`` :39-80, ``one_step_planner`` :82-112):
per MPC step plan with the solver, apply the first input (plus the feedback law when ``agent.feedback.use``) to the true
plant ``env.discrete_dyn``, move the agent, optionally run ``prepare_dynamics_set`` (``common.dynamics_rejection``),
record what ``src/visu.py:475-517`` records.  It drives any solver object with the reference solver's surface
(``solve(agent)``, ``get_solution()`` / ``get_and_shift_solution()``, ``set_initial_state``).

acados / casadi are not installable in this image, so the solver shipped here is ``SurrogateSolver``: the SQP loop of
reference ``src/solver.py:56-156`` as far as it touches the hot path - ``train_hallucinated_dynGP(sqp_iter)`` ->
``get_batch_x_hat[_u_diff]`` -> ``dyn_fg_jacobians`` (-> ``y_grad += u_grad K`` under feedback) -> stage parameter vectors
``p_lin`` in the acados layout, packed on the device - with the QP step replaced by the deterministic surrogate SURVEY.md
section 8d names: the next iterate's linearisation point of stage j+1 is the sample mean of stage j's predicted next
state, inputs stay at the nominal sequence.  With acados present a ``DEMPC_solver`` can be dropped in unchanged.
"""
from __future__ import annotations

import time
from typing import Optional

import numpy as np
import torch


class SurrogateSolver:
    """SQP driver with the surface reference ``src/DEMPC.py`` uses on ``DEMPC_solver``; QP step = SURVEY 8d surrogate."""

    def __init__(self, params, u_nominal: Optional[np.ndarray] = None, pack_p_lin: bool = True):
        self.params = params
        self.H = params["optimizer"]["H"]
        self.max_sqp_iter = params["optimizer"]["SEMPC"]["max_sqp_iter"]
        self.tol_nlp = params["optimizer"]["SEMPC"]["tol_nlp"]
        self.nx, self.nu = params["agent"]["dim"]["nx"], params["agent"]["dim"]["nu"]
        self.ns = params["agent"]["num_dyn_samples"]
        self.pack = pack_p_lin
        self.x_h = np.zeros((self.H + 1, self.nx * self.ns))      # iterate: states of stages 0..H (stage H = terminal)
        self.u_h = np.zeros((self.H, self.nu)) if u_nominal is None else np.array(u_nominal, dtype=np.float64).reshape(self.H, self.nu)
        self.p_lin = None
        self.iterations = 0
        self.gp_ms = []                                           # per SQP iteration: GP side wall time (ms)

    # the reference pins stage 0 with ocp_solver.set(0, "lbx"/"ubx", st_curr) (src/DEMPC.py:89-90)
    def set_initial_state(self, st_curr):
        st = np.asarray(st_curr, dtype=np.float64).reshape(-1)
        if st.size != self.nx * self.ns:
            raise ValueError("st_curr must hold the current state once per sample (Ns * nx values)")
        if not np.any(self.x_h):
            self.x_h[:] = st[None, :]                             # first call: every stage starts at the current state
        self.x_h[0] = st

    def solve(self, player) -> int:
        p = self.params
        H, ns, nx = self.H, self.ns, self.nx
        fb = p["agent"]["feedback"]["use"]
        K = np.array(p["optimizer"]["terminal_tightening"]["K"]) if fb else None
        x_equi = np.array(p["env"]["goal_state"])
        w = np.ones(H + 1) * p["optimizer"].get("w", 1.0)
        xg = np.ones(H + 1) * np.asarray(player.get_next_to_go_loc(), dtype=np.float64).reshape(-1)[0]
        self.gp_ms = []
        for sqp_iter in range(self.max_sqp_iter):
            x_old = self.x_h.copy()
            xs = self.x_h[:H]
            t0 = time.perf_counter()
            u_fb = (-(x_equi - xs.reshape(H, ns, -1)) @ K.T + np.tile(self.u_h[:, None, :], (ns, 1))) if fb else None
            if self.pack and hasattr(player, "sqp_linearisation"):
                # the HIP Agent's fused call (what a solver that consumes p_lin needs, src/solver.py:84-131): one upload, the
                # draw, Jacobians + p_lin in one launch, one download; the three arrays stay on the device
                self.p_lin = player.sqp_linearisation(xs, u_fb if fb else self.u_h, sqp_iter, xg, w, K=K, u_nominal=self.u_h)
                self.gp_ms.append((time.perf_counter() - t0) * 1e3)
                mean_next = player._last_device_jacobians[0][:, :, :, 0].mean(dim=0).T.cpu().numpy()
            else:
                player.train_hallucinated_dynGP(sqp_iter)
                if fb:
                    gp_val, y_grad, u_grad = player.dyn_fg_jacobians(player.get_batch_x_hat_u_diff(xs, u_fb), sqp_iter)
                else:
                    gp_val, y_grad, u_grad = player.dyn_fg_jacobians(player.get_batch_x_hat(xs, self.u_h), sqp_iter)
                if self.pack:
                    self.p_lin = player.pack_p_lin(xs, self.u_h, xg, w, K=K)
                self.gp_ms.append((time.perf_counter() - t0) * 1e3)
                mean_next = gp_val[:, :, :, 0].mean(axis=0).T       # (H, nx)
            # surrogate QP step: stage j+1 is linearised where the sample mean of stage j's prediction lands
            self.x_h[1:] = np.tile(mean_next, (1, ns))
            self.iterations = sqp_iter + 1
            x_diff = np.linalg.norm(self.x_h - x_old) / (np.linalg.norm(x_old) + 1e-6)
            if sqp_iter >= 1 and x_diff < self.tol_nlp:
                break
        return 0

    def get_solution(self):
        return self.x_h.copy(), self.u_h.copy(), np.zeros(self.H + 1)

    def get_and_shift_solution(self):
        X, U, Sl = self.get_solution()
        self.x_h[:-1] = X[1:]
        self.u_h[:-1] = U[1:]
        return X, U, Sl


class Recorder:
    """The lists reference ``src/visu.py:475-517`` records and pickles into ``data.pkl``."""

    def __init__(self, params, agent):
        self.params, self.agent = params, agent
        self.state_traj, self.input_traj, self.mean_state_traj, self.true_state_traj = [], [], [], []
        self.physical_state_traj, self.solver_time = [], []
        self.gp_model_after_solve_train_X, self.gp_model_after_solve_train_Y = [], []
        self.tilde_eps_list = getattr(agent, "tilde_eps_list", None)
        self.ci_list = getattr(agent, "ci_list", None)

    def propagate_true_dynamics(self, x_init, U):
        """true plant along the planned inputs (+ feedback law), reference ``src/visu.py:195-218``"""
        p = self.params
        K = np.array(p["optimizer"]["terminal_tightening"]["K"])
        x_equi = np.array(p["env"]["goal_state"])
        states = [np.asarray(x_init, dtype=np.float64).reshape(-1)]
        for j in range(U.shape[0]):
            cur = states[-1]
            u = -(x_equi - cur) @ K.T + U[j] if p["agent"]["feedback"]["use"] else U[j]
            xu = torch.as_tensor(np.hstack([cur, u])).reshape(1, -1)
            states.append(np.asarray(self.agent.env_model.discrete_dyn(xu)).reshape(-1))
        return np.stack(states)

    def record(self, x_curr, X, U, dt, record_gp_model=True):
        self.physical_state_traj.append(x_curr)
        self.state_traj.append(X)
        self.input_traj.append(U)
        self.solver_time.append(dt)
        if record_gp_model and self.agent.model_i is not None:
            self.gp_model_after_solve_train_X.append(self.agent.model_i.train_inputs[0].detach().cpu())
            self.gp_model_after_solve_train_Y.append(self.agent.model_i.train_targets.detach().cpu())
        nx = self.agent.nx
        self.true_state_traj.append(self.propagate_true_dynamics(X[0][:nx], U))

    def data_dict(self) -> dict:
        return {"state_traj": self.state_traj, "input_traj": self.input_traj, "mean_state_traj": self.mean_state_traj,
                "true_state_traj": self.true_state_traj, "physical_state_traj": self.physical_state_traj,
                "solver_time": self.solver_time,
                "gp_model_after_solve_train_X": self.gp_model_after_solve_train_X,
                "gp_model_after_solve_train_Y": self.gp_model_after_solve_train_Y,
                "tilde_eps_list": self.tilde_eps_list, "ci_list": self.ci_list}

    def save_data(self, save_dir: str) -> str:
        from .io_formats import save_data_pkl
        return save_data_pkl(save_dir, self.data_dict())


class ClosedLoop:
    """reference ``src/DEMPC.py`` with a pluggable solver."""

    def __init__(self, params, agent, solver=None, recorder: Optional[Recorder] = None, rng=None):
        self.params, self.agent = params, agent
        self.solver = SurrogateSolver(params) if solver is None else solver
        self.recorder = Recorder(params, agent) if recorder is None else recorder
        self.nx = params["agent"]["dim"]["nx"]
        self.rng = rng

    def one_step_planner(self, st_curr):
        self.solver.set_initial_state(st_curr)
        t0 = time.perf_counter()
        self.solver.solve(self.agent)
        dt = time.perf_counter() - t0
        if self.params["agent"].get("shift_soln", False):
            X, U, _ = self.solver.get_and_shift_solution()
        else:
            X, U, _ = self.solver.get_solution()
        self.recorder.record(st_curr, X, U, dt)
        return torch.from_numpy(X), torch.from_numpy(U)

    def receding_horizon(self, n_steps: Optional[int] = None):
        p = self.params
        n = p["common"]["num_MPC_itrs"] if n_steps is None else n_steps
        for i in range(n):
            self.agent.mpc_iteration(i)
            x_curr = np.asarray(self.agent.current_state[: self.nx], dtype=np.float64).reshape(self.nx)
            st_curr = np.array(x_curr.tolist() * p["agent"]["num_dyn_samples"])
            X, U = self.one_step_planner(st_curr)
            if p["agent"]["feedback"]["use"]:
                K = torch.tensor(p["optimizer"]["terminal_tightening"]["K"], dtype=X.dtype)
                x_equi = torch.tensor(p["env"]["goal_state"], dtype=X.dtype)
                U_i = -(x_equi - X[0][: self.nx]) @ K.T + U[0]
            else:
                U_i = U[0]
            state_input = torch.hstack([X[0][: self.nx], U_i]).reshape(1, -1)
            state_kp1 = self.agent.env_model.discrete_dyn(state_input)
            self.agent.update_current_state(np.asarray(state_kp1, dtype=np.float64).reshape(-1))
            if p["common"].get("dynamics_rejection", False):
                self.agent.prepare_dynamics_set(X, U, torch.as_tensor(state_kp1), rng=self.rng)
        return False

    def run(self):
        """``dempc_main`` of the reference: loop ``receding_horizon`` until it says stop."""
        while self.receding_horizon():
            pass
        return self.recorder
