#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own Python code.

Runs only in the build container (needs /root/reference); nothing of the reference travels: the outputs are
small .npz files of inputs and expected outputs.  Usage:  python tests/golden/make_goldens.py

What is captured, and from which reference code:

* ``env_*.npz`` ............. reference ``src/environments/pendulum1D.py`` / ``car_model_residual.py`` (real code)
* ``tightenings.npz`` ....... reference ``src/utils/reachable_set.py:3-38`` (real code)
* ``agent_plumbing_*.npz`` .. reference ``src/agent.py`` (real code; ``gpytorch`` replaced by an import stub):
                              seeded base samples, real-data tiling, ``get_batch_x_hat[_u_diff]``,
                              ``update_hallucinated_Dyn_dataset``, ``dyn_fg_jacobians`` with an injected sample
* ``agent_e2e_*.npz`` ....... reference ``src/agent.py`` driven exactly as reference
                              ``benchmarking/simulate_forward_sampling_car.py:108-138`` and
                              ``src/solver.py:84-94`` drive it, with the stub's model class delegating the GP algebra
                              to ``oracle/gp_oracle.py``.  These pin everything AROUND the GP algebra (sample
                              post-processing, hallucinated-set growth, B_d / padding / velocity transform, state
                              hand-over).  They do NOT pin the GP algebra itself (gpytorch is not installable here;
                              "parity unpinned", see oracle/gp_oracle.py).
* ``agent_e2e_prepare_dynamics_set_pendulum1D.npz`` .. reference ``Agent.prepare_dynamics_set`` (src/agent.py:331-443) itself,
                              ``Tensor.cuda`` neutralised, its internal ``randn`` draws recorded as base samples
* ``agent_e2e_pinned_samples.npz`` .. reference ``get_batch_gp_sensitivities`` (src/agent.py:582-624) with
                              ``true_dyn_as_sample`` / ``mean_as_dyn_sample`` and the Ns = 1 / Ns = 2 short-circuits
* ``conditioning_gp.npz`` ... reference ``extra/conditioning_gp.py`` executed as is (numpy RBF posterior sampler,
                              value-only) - an in-reference cross-check of kernel + Cholesky conditioning + sampling.
* ``configs`` ............... the three runnable reference YAMLs re-serialised into sampling_gpmpc_amd/params/.
* ``formats_check.npz`` ..... files written by ``sampling_gpmpc_amd.io_formats`` (``data.pkl``, ``X_traj_list_<k>.pkl``,
                              ``data_X_traj_<idx>.pkl``) read back by the reference's own loader lines
                              (``benchmarking/simulate_forward_sampling_car.py:91-98``, ``extra/cdc_plt.py:162-176``,
                              ``benchmarking/generate_convex_hull.py:76-83``), executed from the reference sources here;
                              what they produced is the fixture (``python tests/golden/make_goldens.py --only formats``).

``--real-gpytorch``: the one-command pin of the GP algebra for a machine that HAS gpytorch (this image does not; the script
then says so and exits with status 3 without touching anything).  It drives the reference's real ``src/agent.py`` with the
GENUINE library through the same end-to-end cases, writes ``agent_e2e_*_gpytorch.npz`` next to the stub-generated files,
prints a per-tensor difference against the oracle, and tries the ``OracleSemantics`` switches to say which setting the
library matches (``tests/test_oracle_golden.py`` picks the files up when they exist).
"""
import contextlib
import copy
import io
import os
import runpy
import sys
import tempfile
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from oracle.gp_oracle import GPHyper, OracleGP  # noqa: E402


# ------------------------------------------------------------------------------------------------------------
# import stub for gpytorch: just enough names for ``src.GP_model`` / ``src.agent`` to import and for Agent's
# context managers to run.  The model class used by Agent is replaced below by an adapter around OracleGP.
# ------------------------------------------------------------------------------------------------------------
def install_gpytorch_stub():
    g = types.ModuleType("gpytorch")
    for sub in ["models", "kernels", "means", "likelihoods", "distributions", "settings", "constraints", "mlls"]:
        m = types.ModuleType("gpytorch." + sub)
        setattr(g, sub, m)
        sys.modules["gpytorch." + sub] = m
    g.models.ExactGP = type("ExactGP", (torch.nn.Module,), {})
    g.kernels.RBFKernel = g.kernels.ScaleKernel = object

    class _Lik:
        def __init__(self, *a, **k):
            pass

        def eval(self):
            return self

    g.likelihoods.MultitaskGaussianLikelihood = _Lik
    g.constraints.GreaterThan = lambda *a, **k: None
    for name in ["observation_nan_policy", "fast_computations", "cholesky_jitter"]:
        setattr(g.settings, name, lambda *a, **k: contextlib.nullcontext())
    sys.modules["gpytorch"] = g


class OracleModelAdapter:
    """Has the constructor signature of reference ``BatchMultitaskGPModelWithDerivatives_fromParams``."""

    def __new__(cls, train_x, train_y, likelihood, params, batch_shape=None, use_grad=True):
        return OracleGP(train_x, train_y, GPHyper.from_params(params, use_grad))


def load_params(name):
    with open(f"{REF}/params/{name}.yaml") as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def formats_fixture():
    """Write with sampling_gpmpc_amd.io_formats, read with the reference's loader lines (taken from its sources at run time)."""
    import pickle
    import re
    from sampling_gpmpc_amd import io_formats as io
    g = torch.Generator().manual_seed(21)
    Ns, nx, nu, H, g_ny = 5, 2, 1, 6, 1
    X_traj = torch.randn(Ns, nx, H + 1, generator=g, dtype=torch.float64).numpy()
    U = torch.randn(H, nu, generator=g, dtype=torch.float64).numpy()
    data = {"state_traj": [torch.randn(H + 1, Ns * nx, generator=g, dtype=torch.float64).numpy() for _ in range(2)],
            "input_traj": [U * 0.5, U], "mean_state_traj": [], "true_state_traj": [X_traj[0].T.copy(), X_traj[1].T.copy()],
            "physical_state_traj": [np.zeros(Ns * nx), np.ones(Ns * nx)], "solver_time": [0.1, 0.2],
            "gp_model_after_solve_train_X": [torch.zeros(Ns, g_ny, 3, 2)], "gp_model_after_solve_train_Y": [torch.zeros(Ns, g_ny, 3, 3)],
            "tilde_eps_list": [np.arange(4.0)], "ci_list": [0.5]}
    out = {"X_traj": X_traj, "U": U, "input_traj_0": data["input_traj"][0], "input_traj_1": data["input_traj"][1]}
    with tempfile.TemporaryDirectory() as td:
        io.save_data_pkl(td, data)
        io.save_x_traj_list(td, 3, X_traj, U, g_ny)
        for idx in (1, 2):
            io.save_x_traj(td, idx, X_traj + idx)
        # (1) reference benchmarking/simulate_forward_sampling_car.py:91-98, executed from its source: data.pkl -> the
        #     open-loop input sequence the forward-sampling harness replays
        src = open(f"{REF}/benchmarking/simulate_forward_sampling_car.py").read().split("\n")
        code = "\n".join(src[90:98])                      # `with open(input_data_path, "rb") ...` .. `input_gpmpc_input_traj = ...`
        assert "pickle.load(input_data_file)" in code and "input_gpmpc_input_traj" in code, "reference lines moved"
        ns = {"pickle": pickle, "input_data_path": f"{td}/data.pkl", "np": np}
        exec(code, ns)
        out["ref_input_traj_last"] = np.asarray(ns["input_gpmpc_input_traj"])
        # (2) reference extra/cdc_plt.py:162-176: X_traj_list_<k>.pkl -> (H+1, N, g_ny, 1, nx+nu) array, state_traj slice
        with open(f"{td}/X_traj_list_3.pkl", "rb") as a_file:
            sampling_gpmpc_data = pickle.load(a_file)
        sampling_gpmpc_data_np = np.array([np.array(x.cpu()) for x in sampling_gpmpc_data])       # cdc_plt.py:169-171
        H_GT, N_samples, nx_, _, nxu = sampling_gpmpc_data_np.shape                                 # cdc_plt.py:172
        out["ref_list_shape"] = np.array(sampling_gpmpc_data_np.shape)
        out["ref_state_traj"] = sampling_gpmpc_data_np[:, :, 0, 0, 0:2]                             # cdc_plt.py:176
        # (3) reference benchmarking/generate_convex_hull.py:76-83, executed from its source: the per-job tube files are
        #     concatenated over the sample axis
        src = open(f"{REF}/benchmarking/generate_convex_hull.py").read().split("\n")
        code = "\n".join(src[75:83]).replace("range(1)", "range(1, 3)")
        assert "data_X_traj_" in code and "np.vstack(input_gpmpc_data_list)" in code, "reference lines moved"

        class _A:
            i = "job"
        os.makedirs(f"{td}/job", exist_ok=True)
        for idx in (1, 2):
            os.replace(f"{td}/data_X_traj_{idx}.pkl", f"{td}/job/data_X_traj_{idx}.pkl")
        ns = {"pickle": pickle, "np": np, "save_path": td + "/", "args": _A, "input_gpmpc_data_list": [], "print": lambda *a: None}
        exec(code, ns)
        out["ref_merged"] = np.asarray(ns["X_traj"])
    np.savez(f"{HERE}/formats_check.npz", **out)
    print("formats_check.npz written: data.pkl / X_traj_list_3.pkl / data_X_traj_{1,2}.pkl read back by the reference's loaders")


def real_gpytorch_pin():
    """Needs a machine with gpytorch==1.13 (reference requirements.txt:6).  See the module docstring."""
    try:
        import gpytorch                                                              # noqa: F401
    except Exception as e:                                                           # noqa: BLE001
        print(f"--real-gpytorch: `import gpytorch` failed ({e!r}).  Nothing was written.  Install gpytorch==1.13 "
              f"(reference requirements.txt:6) next to /root/reference and rerun this command; parity of the GP algebra "
              f"stays 'unpinned' until then.")
        return 3
    import itertools
    import src.agent as ref_agent
    from src.environments.pendulum1D import Pendulum as RefPendulum1D
    from src.environments.car_model_residual import CarKinematicsModel as RefCar
    from oracle import agent_oracle as ao
    from oracle.gp_oracle import OracleSemantics
    torch.set_default_dtype(torch.float64)
    report = []
    cases = [("R_pendulum1D", RefPendulum1D, "params_pendulum1D_samples", 8, 10, False, True),
             ("I_car", RefCar, "params_car_residual_fs", 8, 8, True, True),
             ("R_car", RefCar, "params_car_residual_fs", 4, 8, False, True)]
    for tag, cls, pname, Ns, Ht, nograd, fb in cases:
        p = load_params(pname)
        p["common"]["use_cuda"] = False
        p["agent"]["num_dyn_samples"], p["optimizer"]["H"] = Ns, 1
        p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = Ht, 2
        p["env"]["use_model_without_derivatives"], p["agent"]["feedback"]["use"] = nograd, fb
        torch.manual_seed(123456)
        agent = quiet(ref_agent.Agent, p, cls(p))
        u_ff = np.linspace(-1, 1, Ht).reshape(Ht, 1) if agent.nu == 1 else \
            np.stack([0.05 * np.sin(2 * np.pi * np.arange(Ht) / Ht), np.zeros(Ht)], axis=1)
        X_ref, Y_ref = fs_loop(agent, p, u_ff)
        erv = agent.epistimic_random_vector.clone()
        np.savez(f"{HERE}/agent_e2e_{tag}_gpytorch.npz", Ns=Ns, H_traj=Ht, nograd=nograd, feedback=fb, u_ff=u_ff,
                 beta=p["agent"]["Dyn_gp_beta"], epistimic_random_vector=erv.numpy(), X_traj=X_ref, Y=Y_ref)
        best = None
        for jp, ef, nm in itertools.product(("failed_elements", "whole_batch", "always"), ("whole_batch", "failed_elements"),
                                            (True, False)):
            sem = OracleSemantics(jitter_policy=jp, eigh_fallback=ef, nan_mask_batch_collapse=nm)
            oa = ao.OracleAgent(p, ao.make_oracle_env(p), erv, semantics=sem)
            Xo, Yo = ao.forward_sampling_rollout(oa, u_ff, return_samples=True)
            err = float(np.max(np.abs(Xo - X_ref)) / np.max(np.abs(X_ref)))
            report.append(f"{tag:14s} jitter_policy={jp:16s} eigh_fallback={ef:16s} nan_collapse={nm!s:5s}: "
                          f"rel err X_traj {err:.3e}  Y {float(np.max(np.abs(Yo - Y_ref))):.3e}")
            if best is None or err < best[0]:
                best = (err, sem)
        report.append(f"{tag:14s} best match: {best[1]} ({best[0]:.3e})")
    n_split = joint_car_split_capture(ref_agent, RefCar, "_gpytorch")
    report.append(f"J_car_split    written: k = 0..3 of MPC step 0 and k = 0 of MPC step 1 (conditioning set {n_split} points), "
                  f"mean / variance / samples of the reference's model_i_call")
    txt = "\n".join(report)
    print(txt)
    with open(f"{HERE}/gpytorch_pin_report.txt", "w") as f:
        f.write(txt + "\n")
    return 0


def joint_car_split_capture(ref_agent, RefCar, suffix):
    """Mode J as the closed loop runs it (src/solver.py:84-94): the car AS SHIPPED (Dyn_gp_jitter 1e-20 -> the eigendecomposition
    root), H = 40, SQP iterations k = 0..3 of MPC step 0 and k = 0 of MPC step 1, whose draw conditions on 45 + 480 slots - the size
    the HIP matrix-pipe path serves with its TOP + BOTTOM pair of launches (include/gpmpc_hip.h, ABI 7).  Driven through the
    reference's own Agent; with ``suffix == "_gpytorch"`` the algebra behind it is the genuine library, else the import stub's
    (the oracle).  Eigenvector signs are solver specific: consumers compare mean / variance and continue from these labels."""
    p = load_params("params_car_residual")
    Ns, H, iters = 4, 40, 4
    p["common"]["use_cuda"] = False
    p["agent"]["num_dyn_samples"], p["optimizer"]["H"] = Ns, H
    p["agent"]["true_dyn_as_sample"] = False
    p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = 2, iters
    torch.manual_seed(123456)
    agent = quiet(ref_agent.Agent, p, RefCar(p))
    x0 = np.array(p["env"]["start"], dtype=np.float64)[: agent.nx]
    u_h = np.stack([0.05 * np.sin(2 * np.pi * np.arange(H) / H), np.zeros(H)], axis=1)
    x_h = np.tile(x0, (H, Ns))
    rec = {"Ns": Ns, "H": H, "iters": iters, "u_h": u_h, "epistimic_random_vector": agent.epistimic_random_vector.clone().numpy()}
    n_last = 0
    for step, k in [(0, kk) for kk in range(iters)] + [(1, 0)]:
        agent.mpc_iteration(step)
        quiet(agent.train_hallucinated_dynGP, k)
        n_last = int(agent.model_i.train_inputs[0].shape[-2])
        gp_val, y_grad, u_grad = quiet(agent.dyn_fg_jacobians, agent.get_batch_x_hat(x_h, u_h), k)
        post = agent.model_i_call
        key = f"s{step}k{k}"
        rec.update({f"x_h_{key}": x_h.copy(), f"mean_{key}": post.mean.detach().numpy().copy(),
                    f"var_{key}": post.variance.detach().numpy().copy(), f"y_{key}": agent.model_i_samples.detach().numpy().copy(),
                    f"gp_val_{key}": np.asarray(gp_val).copy(), f"n_train_{key}": n_last})
        mean_next = np.asarray(gp_val)[:, :, :, 0].mean(axis=0).T
        x_h = np.tile(np.vstack([x0[None, :], mean_next[:-1]]), (1, Ns))
    np.savez(f"{HERE}/agent_e2e_J_car_split{suffix}.npz", **rec)
    return n_last


def fs_loop(agent, p, u_ff):
    """reference benchmarking/simulate_forward_sampling_car.py:108-138, same calls in the same order."""
    ns, nx = agent.ns, agent.nx
    K = np.array(p["optimizer"]["terminal_tightening"]["K"])
    x_equi = np.array(p["env"]["goal_state"])
    agent.update_current_state(np.array(p["env"]["start"]))
    x_curr = agent.current_state[:nx].reshape(nx)
    H = u_ff.shape[0]
    x_h = np.tile(x_curr, (1, ns))
    X_traj = torch.empty((ns, nx, H + 1))
    Ys = []
    for H_idx in range(H):
        agent.train_hallucinated_dynGP(1, use_model_without_derivatives=p["env"]["use_model_without_derivatives"])
        agent.mpc_iteration(H_idx)
        u_h = u_ff[H_idx].reshape(1, -1)
        if p["agent"]["feedback"]["use"]:
            bx = agent.get_batch_x_hat_u_diff(
                x_h, -(x_equi - x_h.reshape(1, ns, -1)) @ K.T + np.tile(u_h[:, None, :], (ns, 1)))
        else:
            bx = agent.get_batch_x_hat(x_h, u_h)
        gp_val, _, _ = quiet(agent.dyn_fg_jacobians, bx, 1)
        Ys.append(agent.model_i_samples.clone())
        X_traj[:, :, H_idx] = bx[:, 0, 0, :nx]
        x_h = gp_val[:, :, 0, 0].reshape(1, -1)
    X_traj[:, :, H_idx + 1] = torch.tensor(gp_val[:, :, 0, 0])
    return X_traj.numpy(), torch.cat(Ys, dim=2).numpy()


class RecordDraws:
    """The reference's ``prepare_dynamics_set`` calls ``.sample()`` WITHOUT base samples (src/agent.py:376): the library draws
    ``randn`` internally.  While active, the stub's posterior draws the same ``randn`` (global torch generator), records it in
    the layout of ``base_samples`` (batch..., m, T) and passes it on - so the capture holds exactly what was drawn."""

    def __enter__(self):
        import oracle.gp_oracle as go
        self.go, self.orig, self.draws = go, go.OraclePosterior.sample, []
        rec = self

        def sample(post, base_samples=None):
            if base_samples is None:
                m, T = post.mean.shape[-2], post.mean.shape[-1]
                z = torch.randn(*post.mean.shape[:-2], m * T, 1, dtype=post.mean.dtype)
                base_samples = z.reshape(*post.mean.shape[:-2], m, T)
                rec.draws.append(base_samples.clone())
            return rec.orig(post, base_samples)

        go.OraclePosterior.sample = sample
        return self

    def __exit__(self, *exc):
        self.go.OraclePosterior.sample = self.orig


def prepare_dynamics_set_fixture(ref_agent, RefPendulum1D):
    """``agent_e2e_prepare_dynamics_set_pendulum1D.npz``: the reference's REAL ``Agent.prepare_dynamics_set`` (src/agent.py:331-443:
    forward sampling with box rejection, re-training on real + forward-sampled value-only + hallucinated data at every step,
    survivor replacement through the global ``np.random``), ``Tensor.cuda`` neutralised (the method is CUDA-only, no GPU here),
    the GP algebra from the stub (oracle).  Two calls: a tube nobody leaves, then a tube on theta at the third state that
    rejects part of the samples."""
    import re
    Ns, H = 10, 6
    p = load_params("params_pendulum1D_samples")
    p["common"]["use_cuda"] = False
    p["agent"]["num_dyn_samples"], p["optimizer"]["H"] = Ns, H
    p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = 1, 2
    torch.manual_seed(123456)
    agent = quiet(ref_agent.Agent, p, RefPendulum1D(p))
    nx, nu = agent.nx, agent.nu
    g = torch.Generator().manual_seed(21)
    x0 = np.array(p["env"]["start"], dtype=np.float64)
    x_h = np.tile(x0, (H, Ns)) + 0.05 * torch.randn(H, Ns * nx, generator=g, dtype=torch.float64).numpy()
    u_h = 0.3 * torch.randn(H, Ns, nu, generator=g, dtype=torch.float64).numpy()
    agent.mpc_iteration(0)
    agent.train_hallucinated_dynGP(0)
    quiet(agent.dyn_fg_jacobians, agent.get_batch_x_hat_u_diff(x_h, u_h), 0)
    agent.train_hallucinated_dynGP(1)
    out = {"Ns": Ns, "H": H, "epistimic_random_vector": agent.epistimic_random_vector.numpy(), "x_h": x_h, "u_h": u_h,
           "hall_X_0": agent.Hallcinated_X_train.numpy().copy(), "hall_Y_0": agent.Hallcinated_Y_train.numpy().copy()}
    U_soln = 0.5 * torch.randn(H + 1, nu, generator=g, dtype=torch.float64)
    X_kp1 = torch.tensor(x0[:nx]).reshape(nx, 1)
    X_soln = torch.zeros(H + 1, Ns * nx, dtype=torch.float64)
    X_soln[1] = torch.tensor(x0[:nx]).repeat(Ns)
    out.update({"U_soln": U_soln.numpy(), "X_kp1": X_kp1.numpy(), "X_soln_1": X_soln.numpy()})
    cuda_orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        def call(tag, Xs, ci, np_seed, torch_seed):
            agent.ci_list = ci
            agent.train_hallucinated_dynGP(1)
            np.random.seed(np_seed)
            torch.manual_seed(torch_seed)
            buf = io.StringIO()
            with RecordDraws() as rec, contextlib.redirect_stdout(buf):
                agent.prepare_dynamics_set(Xs.clone(), U_soln, X_kp1)
            counts = [int(c) for c in re.findall(r"are\s+tensor\((\d+)", buf.getvalue())]
            assert len(rec.draws) == H - 1 and len(counts) == H, (len(rec.draws), counts)
            out.update({f"z_{tag}": torch.stack(rec.draws).numpy(), f"survivors_{tag}": np.array(counts),
                        f"FS_X_{tag}": agent.FS_X_train_batch.numpy().copy(), f"FS_Y_{tag}": agent.FS_Y_train_batch.numpy().copy(),
                        f"hall_X_{tag}": agent.Hallcinated_X_train.numpy().copy(),
                        f"hall_Y_{tag}": agent.Hallcinated_Y_train.numpy().copy(), f"np_seed_{tag}": np_seed})
        call("1", X_soln, [1e9] * (H + 1), 5, 100)
        x2 = agent.FS_X_train_batch[:, 0, 2, 0]                    # theta after two sampled steps
        med = float(x2.median())
        tol = float((x2 - med).abs().median()) + 1e-12
        X_soln2 = X_soln.clone()
        X_soln2[3] = torch.stack([torch.full((Ns,), med, dtype=torch.float64), torch.zeros(Ns, dtype=torch.float64)], dim=1).reshape(-1)
        ci = [1e9] * (H + 1)
        ci[2] = torch.tensor([tol, 1e9], dtype=torch.float64)
        call("2", X_soln2, ci, 7, 100)                             # same internal draws as the first call: same sampled states
        out.update({"X_soln_2": X_soln2.numpy(), "tube_med": med, "tube_tol": tol})
        assert 0 < out["survivors_2"][-1] < Ns, out["survivors_2"]
    finally:
        torch.Tensor.cuda = cuda_orig
    np.savez(f"{HERE}/agent_e2e_prepare_dynamics_set_pendulum1D.npz", **out)


def pinned_samples_fixture(ref_agent, RefPendulum1D, RefCar):
    """``agent_e2e_pinned_samples.npz``: the branches of ``get_batch_gp_sensitivities`` (src/agent.py:582-624) that overwrite
    samples after the draw - ``true_dyn_as_sample`` (set by the shipped params_car_residual.yaml:50), ``mean_as_dyn_sample`` -
    and their short-circuits (Ns == 1 with either flag, Ns == 2 with both: no draw, nothing appended), through the reference's
    real ``dyn_fg_jacobians``."""
    out = {}
    cases = [("pend_true_ns1", "params_pendulum1D_samples", RefPendulum1D, 1, True, False),
             ("pend_true_ns3", "params_pendulum1D_samples", RefPendulum1D, 3, True, False),
             ("pend_mean_ns3", "params_pendulum1D_samples", RefPendulum1D, 3, False, True),
             ("pend_both_ns2", "params_pendulum1D_samples", RefPendulum1D, 2, True, True),
             ("pend_both_ns4", "params_pendulum1D_samples", RefPendulum1D, 4, True, True),
             ("car_true_ns1", "params_car_residual", RefCar, 1, True, False),
             ("car_true_ns3", "params_car_residual", RefCar, 3, True, False)]
    for tag, pname, Env, Ns, td, md in cases:
        H = 5
        p = load_params(pname)
        p["common"]["use_cuda"] = False
        p["agent"]["num_dyn_samples"], p["optimizer"]["H"] = Ns, H
        p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = 1, 2
        p["agent"]["true_dyn_as_sample"], p["agent"]["mean_as_dyn_sample"] = td, md
        if "car" in pname:
            p["agent"]["Dyn_gp_jitter"] = 1e-8                     # the Cholesky-retry branch (the eigh root's signs are solver specific)
        torch.manual_seed(123456)
        agent = quiet(ref_agent.Agent, p, Env(p))
        nx, nu = agent.nx, agent.nu
        g = torch.Generator().manual_seed(17)
        x0 = np.array(p["env"]["start"], dtype=np.float64)
        x_h = np.tile(x0, (H, Ns)) + 0.02 * torch.randn(H, Ns * nx, generator=g, dtype=torch.float64).numpy()
        u_h = 0.1 * torch.randn(H, Ns, nu, generator=g, dtype=torch.float64).numpy()
        agent.mpc_iteration(0)
        rec = {"Ns": Ns, "H": H, "true_dyn": td, "mean": md, "epistimic_random_vector": agent.epistimic_random_vector.numpy(),
               "x_h": x_h, "u_h": u_h}
        for it in range(2):
            agent.train_hallucinated_dynGP(it)
            gp_val, y_grad, u_grad = quiet(agent.dyn_fg_jacobians, agent.get_batch_x_hat_u_diff(x_h, u_h), it)
            rec.update({f"gp_val_{it}": gp_val, f"y_grad_{it}": y_grad, f"u_grad_{it}": u_grad,
                        f"hall_X_{it}": agent.Hallcinated_X_train.numpy().copy(), f"hall_Y_{it}": agent.Hallcinated_Y_train.numpy().copy(),
                        f"mean_{it}": agent.model_i_call.mean.numpy().copy()})
        for k, v in rec.items():
            out[f"{tag}__{k}"] = v
    out["cases"] = np.array([c[0] for c in cases])
    out["case_params"] = np.array([c[1] for c in cases])
    np.savez(f"{HERE}/agent_e2e_pinned_samples.npz", **out)


def main():
    if "--real-gpytorch" in sys.argv:
        sys.exit(real_gpytorch_pin())
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "formats":
        formats_fixture()
        return
    install_gpytorch_stub()
    import src.agent as ref_agent                                                     # noqa
    from src.environments.pendulum1D import Pendulum as RefPendulum1D                 # noqa
    from src.environments.car_model_residual import CarKinematicsModel as RefCar      # noqa
    from src.utils.reachable_set import get_reachable_set_ball as ref_ball            # noqa
    ref_agent.BatchMultitaskGPModelWithDerivatives_fromParams = OracleModelAdapter
    torch.set_default_dtype(torch.float64)
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "joint_split":
        print("agent_e2e_J_car_split.npz:", joint_car_split_capture(ref_agent, RefCar, ""), "points behind the last draw")
        return

    # ---------------- configs -------------------------------------------------------------------------------
    hdr = ("# Configuration values for the GP-rollout hot path; key names and numeric values follow the\n"
           "# reference's params/{name}.yaml (re-serialised by tests/golden/make_goldens.py; data, not code).\n")
    for name in ["params_pendulum1D_samples", "params_car_residual", "params_car_residual_fs"]:
        p = load_params(name)
        with open(f"{REPO}/sampling_gpmpc_amd/params/{name}.yaml", "w") as f:
            f.write(hdr.format(name=name))
            yaml.safe_dump(p, f, default_flow_style=None, sort_keys=True, width=100)

    # ---------------- environments --------------------------------------------------------------------------
    for tag, cls, pname in [("pendulum1D", RefPendulum1D, "params_pendulum1D_samples"),
                            ("car_residual", RefCar, "params_car_residual")]:
        p = load_params(pname)
        p["common"]["use_cuda"] = False
        env = cls(p)
        X, Y = env.initial_training_data()
        g = torch.Generator().manual_seed(7)
        nx, nu = env.nx, env.nu
        xu = torch.rand(5, nx, 3, nx + nu, generator=g, dtype=torch.float64) * 2 - 1
        xu = xu[:, [0]].tile(1, nx, 1, 1)                                    # state row replicated over dim 1
        if tag == "car_residual":
            xu[..., 3] += 12.0
        dg = torch.rand(5, env.g_ny, 3, 1 + env.g_nx + env.g_nu, generator=g, dtype=torch.float64)
        one = torch.tensor(p["env"]["start"] + [0.1] * nu, dtype=torch.float64).reshape(1, -1)
        if tag == "car_residual":
            one = torch.tensor([[0.0, 1.95, 0.0, 14.0, 0.1, 0.5]], dtype=torch.float64)
        np.savez(f"{HERE}/env_{tag}.npz",
                 X_train=X.numpy(), Y_train=Y.numpy(), xu=xu.numpy(), dg=dg.numpy(),
                 prior_data=env.get_prior_data(X).numpy(),
                 unknown_dyn=env.unknown_dyn(X).numpy(),
                 known_dyn=env.known_dyn(xu).numpy(),
                 f_jac=env.get_f_known_jacobian(xu).numpy(),
                 g_xu_hat=env.get_g_xu_hat(xu).numpy(),
                 transform=env.transform_sensitivity(dg, xu).numpy(),
                 B_d=env.B_d.numpy(), pad_g=np.array(env.pad_g), g_idx=np.array(env.g_idx_inputs),
                 one_xu=one.numpy(), discrete_dyn=env.discrete_dyn(one).numpy())

    # ---------------- tightenings ---------------------------------------------------------------------------
    out = {}
    for tag, pname, H in [("P17", "params_pendulum1D_samples", 17), ("P30", "params_pendulum1D_samples", 30),
                          ("C50", "params_car_residual", 50)]:
        p = load_params(pname)
        p["optimizer"]["H"] = H
        te, ci = quiet(ref_ball, p, np.ones(H + 1))
        out[f"{tag}_tilde_eps"] = np.stack(te)
        out[f"{tag}_ci"] = np.array(ci)
    np.savez(f"{HERE}/tightenings.npz", **out)

    # ---------------- Agent plumbing (no GP algebra involved) -----------------------------------------------
    for tag, cls, pname, Ns, H, n_mpc, n_itr in [("pendulum1D", RefPendulum1D, "params_pendulum1D_samples", 4, 5, 2, 1),
                                                  ("car_residual", RefCar, "params_car_residual", 3, 4, 1, 2)]:
        p = load_params(pname)
        p["common"]["use_cuda"] = False
        p["agent"]["num_dyn_samples"] = Ns
        p["agent"]["true_dyn_as_sample"] = False
        p["optimizer"]["H"] = H
        p["common"]["num_MPC_itrs"] = n_mpc
        p["optimizer"]["SEMPC"]["max_sqp_iter"] = n_itr
        torch.manual_seed(123456)
        env = cls(p)
        agent = quiet(ref_agent.Agent, p, env)
        nx, nu = agent.nx, agent.nu
        g = torch.Generator().manual_seed(11)
        x_h = torch.rand(H, Ns * nx, generator=g, dtype=torch.float64).numpy()
        if tag == "car_residual":
            x_h.reshape(H, Ns, nx)[:, :, 3] += 13.0
        u_h = torch.rand(H, nu, generator=g, dtype=torch.float64).numpy() - 0.5
        u_diff = torch.rand(H, Ns, nu, generator=g, dtype=torch.float64).numpy() - 0.5
        bx = agent.get_batch_x_hat(x_h, u_h)
        bxd = agent.get_batch_x_hat_u_diff(x_h, u_diff)
        y_inj = torch.rand(Ns, agent.g_ny, H, agent.in_dim_y, generator=g, dtype=torch.float64) * 0.1
        agent.get_batch_gp_sensitivities = lambda xu, it, _y=y_inj: _y.clone()
        gp_val, y_grad, u_grad = agent.dyn_fg_jacobians(bxd.clone(), 0)
        assert gp_val.dtype == np.float64 and isinstance(gp_val, np.ndarray)
        # hallucinated-set update (min-dist filter disabled, as shipped) and with the filter switched on
        g_xu = env.get_g_xu_hat(bxd)
        quiet(agent.update_hallucinated_Dyn_dataset, g_xu, y_inj)
        hx0, hy0 = agent.Hallcinated_X_train.clone(), agent.Hallcinated_Y_train.clone()
        p["agent"]["Dyn_gp_min_data_dist"] = 0.25
        quiet(agent.update_hallucinated_Dyn_dataset, g_xu + 0.2, y_inj * 2)
        np.savez(f"{HERE}/agent_plumbing_{tag}.npz",
                 Ns=Ns, H=H, n_mpc=n_mpc, n_itr=n_itr,
                 epistimic_random_vector=agent.epistimic_random_vector.numpy(),
                 X_train_batch=agent.Dyn_gp_X_train_batch.numpy(), Y_train_batch=agent.Dyn_gp_Y_train_batch.numpy(),
                 x_h=x_h, u_h=u_h, u_diff=u_diff, batch_x_hat=bx.numpy(), batch_x_hat_u_diff=bxd.numpy(),
                 y_inj=y_inj.numpy(), gp_val=gp_val, y_grad=y_grad, u_grad=u_grad,
                 hall_X_0=hx0.numpy(), hall_Y_0=hy0.numpy(),
                 hall_X_1=agent.Hallcinated_X_train.numpy(), hall_Y_1=agent.Hallcinated_Y_train.numpy(),
                 min_dist_1=0.25)

    # ---------------- Agent end to end, GP algebra delegated to the oracle ----------------------------------
    cases = [
        # tag, class, yaml, Ns, H_traj, use_model_without_derivatives, feedback
        ("R_pendulum1D", RefPendulum1D, "params_pendulum1D_samples", 8, 10, False, True),
        ("R_pendulum1D_nofb", RefPendulum1D, "params_pendulum1D_samples", 4, 6, False, False),
        ("I_car", RefCar, "params_car_residual_fs", 8, 8, True, True),
        ("R_car", RefCar, "params_car_residual_fs", 4, 8, False, True),
    ]
    for tag, cls, pname, Ns, Ht, nograd, fb in cases:
        p = load_params(pname)
        p["common"]["use_cuda"] = False
        p["agent"]["num_dyn_samples"] = Ns
        p["optimizer"]["H"] = 1
        p["common"]["num_MPC_itrs"] = Ht
        p["optimizer"]["SEMPC"]["max_sqp_iter"] = 2
        p["env"]["use_model_without_derivatives"] = nograd
        p["agent"]["feedback"]["use"] = fb
        if tag == "R_car":
            p["agent"]["Dyn_gp_beta"] = 3.0
        torch.manual_seed(123456)
        env = cls(p)
        agent = quiet(ref_agent.Agent, p, env)
        t = np.arange(Ht)
        if env.nu == 1:
            u_ff = np.linspace(-1, 1, Ht).reshape(Ht, 1)
        else:
            u_ff = np.stack([0.05 * np.sin(2 * np.pi * t / Ht), np.zeros(Ht)], axis=1)
        X_traj, Y = fs_loop(agent, p, u_ff)
        np.savez(f"{HERE}/agent_e2e_{tag}.npz", Ns=Ns, H_traj=Ht, nograd=nograd, feedback=fb, u_ff=u_ff,
                 beta=p["agent"]["Dyn_gp_beta"],
                 epistimic_random_vector=agent.epistimic_random_vector.numpy(), X_traj=X_traj, Y=Y,
                 hall_X=agent.Hallcinated_X_train.numpy(), hall_Y=agent.Hallcinated_Y_train.numpy())

    # mode J as the SQP loop drives it (reference src/solver.py:84-94), two SQP iterations, pendulum1D
    p = load_params("params_pendulum1D_samples")
    p["common"]["use_cuda"] = False
    Ns, H = 6, 8
    p["agent"]["num_dyn_samples"] = Ns
    p["optimizer"]["H"] = H
    p["common"]["num_MPC_itrs"] = 2
    p["optimizer"]["SEMPC"]["max_sqp_iter"] = 2
    torch.manual_seed(123456)
    env = RefPendulum1D(p)
    agent = quiet(ref_agent.Agent, p, env)
    K = np.array(p["optimizer"]["terminal_tightening"]["K"])
    x_equi = np.array(p["env"]["goal_state"])
    g = torch.Generator().manual_seed(3)
    outJ = {"Ns": Ns, "H": H, "epistimic_random_vector": agent.epistimic_random_vector.numpy()}
    x_h = np.tile(np.array(p["env"]["start"]), (H, Ns)) + 0.05 * torch.randn(H, Ns * 2, generator=g).numpy()
    u_h = np.linspace(-1, 1, H).reshape(H, 1)
    agent.mpc_iteration(0)
    for it in range(2):
        agent.train_hallucinated_dynGP(it)
        bx = agent.get_batch_x_hat_u_diff(x_h, -(x_equi - x_h.reshape(H, Ns, -1)) @ K.T + np.tile(u_h[:, None, :], (Ns, 1)))
        gp_val, y_grad, u_grad = quiet(agent.dyn_fg_jacobians, bx, it)
        outJ.update({f"x_h_{it}": x_h.copy(), f"gp_val_{it}": gp_val, f"y_grad_{it}": y_grad, f"u_grad_{it}": u_grad,
                     f"mean_{it}": agent.model_i_call.mean.numpy(), f"var_{it}": agent.model_i_call.variance.numpy(),
                     f"y_{it}": agent.model_i_samples.numpy(),
                     f"jitter_{it}": agent.model_i_call.root_info.jitter_added.numpy()})
        x_h = x_h + 0.01 * torch.randn(H, Ns * 2, generator=g).numpy()
    outJ["u_h"] = u_h
    np.savez(f"{HERE}/agent_e2e_J_pendulum1D.npz", **outJ)

    # ---------------- Agent.prepare_dynamics_set (src/agent.py:331-443) and the pinned-sample branches -------
    prepare_dynamics_set_fixture(ref_agent, RefPendulum1D)
    pinned_samples_fixture(ref_agent, RefPendulum1D, RefCar)
    joint_car_split_capture(ref_agent, RefCar, "")

    # ---------------- extra/conditioning_gp.py executed as is -----------------------------------------------
    import matplotlib
    matplotlib.use("Agg")
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            np.random.seed(20240)
            ns_ = quiet(runpy.run_path, f"{REF}/extra/conditioning_gp.py")
        finally:
            os.chdir(cwd)
    np.savez(f"{HERE}/conditioning_gp.npz", X=ns_["X"], y=ns_["y"], Xtest=ns_["Xtest"], Xtest2=ns_["Xtest2"],
             random_weights=ns_["random_weights"], f_post=ns_["f_post"], f_post2=ns_["f_post2"],
             kernel_parameter=0.1, post_jitter=1e-6)
    formats_fixture()
    print("goldens written to", HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f"  {f:40s} {os.path.getsize(os.path.join(HERE, f)):8d} B")


if __name__ == "__main__":
    main()
