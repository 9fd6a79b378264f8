"""GPU: the receding-horizon driver (reference src/DEMPC.py:39-80 + the SQP loop of
src/solver.py:56-131 with the surrogate QP step) on the HIP Agent against the same driver on the oracle Agent."""
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import agent_oracle as ao
from tests.helpers import closed_loop_params
from tests.test_hip_parity import make_agents, relerr, sg  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pname,Ns,H,n_mpc,n_sqp", [("params_pendulum1D_samples", 6, 12, 3, 3),
                                                    ("params_car_residual", 4, 10, 2, 3)])
def test_closed_loop_driver_against_oracle(sg, pname, Ns, H, n_mpc, n_sqp, tmp_path):
    """Three MPC steps x three SQP iterations: GP re-definition incl. the reset-after-build quirk, joint draws (car: the
    shipped jitter -> eigendecomposition root), Jacobian assembly, p_lin packing, plant step, recording, data.pkl."""
    from sampling_gpmpc_amd.closed_loop import ClosedLoop, SurrogateSolver
    from sampling_gpmpc_amd import io_formats as io
    p = closed_loop_params(pname, Ns, H, n_mpc, n_sqp)
    p["optimizer"]["SEMPC"]["tol_nlp"] = 0.0                       # run every SQP iteration
    if "car" in pname:
        p["agent"]["Dyn_gp_jitter"] = 1e-9                         # sample parity needs the sign-free Cholesky branch
    agent, oagent = make_agents(sg, p)
    loops = []
    for a in (agent, oagent):
        a.update_current_state(np.array(p["env"]["start"], dtype=np.float64))
        loops.append(ClosedLoop(p, a, SurrogateSolver(p, pack_p_lin=hasattr(a, "pack_p_lin"))))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rec, orec = loops[0].run(), loops[1].run()
    assert len(rec.state_traj) == n_mpc and loops[0].solver.iterations == n_sqp
    for k in range(n_mpc):
        e = relerr(rec.state_traj[k], orec.state_traj[k])
        print(f"{pname} MPC step {k}: rel err of the planned states {e:.2e}, plant state {np.asarray(rec.physical_state_traj[k])[:agent.nx]}")
        assert e < 1e-6
        np.testing.assert_allclose(rec.true_state_traj[k], orec.true_state_traj[k], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(np.asarray(agent.current_state), np.asarray(oagent.current_state), rtol=1e-9)
    # the stage parameter vectors of the last iteration exist in the acados layout
    assert loops[0].solver.p_lin.shape == (H, Ns * (agent.nx ** 2 + agent.nx * agent.nu + 2 * agent.nx) + 2 * agent.nu + agent.nx + 3)
    # data.pkl round trip with the reference's keys
    path = rec.save_data(str(tmp_path))
    back = io.load_data_pkl(path)
    assert set(back) == set(io.DATA_PKL_KEYS) and len(back["input_traj"]) == n_mpc
    assert back["gp_model_after_solve_train_X"][0].shape[0] == Ns


def test_closed_loop_as_shipped_car_runs_the_eigh_root(sg):
    """The shipped car configuration (Dyn_gp_jitter 1e-20) through the driver: every joint draw takes the eigendecomposition
    root, results finite, the oracle's plant trajectory matches to the tolerance the sign ambiguity leaves (the surrogate
    uses sample means, which do not depend on eigenvector signs only in distribution - so this asserts finiteness, info
    bits and the mean / variance of the last posterior)."""
    from sampling_gpmpc_amd.closed_loop import ClosedLoop, SurrogateSolver
    p = closed_loop_params("params_car_residual", 16, 40, 1, 4)
    p["optimizer"]["SEMPC"]["tol_nlp"] = 0.0
    p["agent"]["base_sample_generator"] = "counter"
    agent, _ = make_agents(sg, p)
    agent.update_current_state(np.array(p["env"]["start"], dtype=np.float64))
    loop = ClosedLoop(p, agent, SurrogateSolver(p))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        loop.run()
    info = agent.model_i_call.last_info.cpu().numpy()
    assert (info & sg._lib.INFO_ROOT_EIGH).all() and not (info & sg._lib.INFO_TRAIN_CHOL_FAIL).any()
    assert np.isfinite(loop.solver.p_lin).all() and np.isfinite(loop.recorder.state_traj[0]).all()
    assert agent.model_i.n_h == 3 * 40                              # three iterations' points behind the fourth draw
    print("as shipped car closed loop: GP side per SQP iteration (ms):", [round(t, 2) for t in loop.solver.gp_ms])


@pytest.mark.parametrize("path", ["valu", "mfma"])
@pytest.mark.parametrize("pname,Ns,H,budget", [("params_pendulum1D_samples", 5, 30, None), ("params_car_residual", 4, 40, None),
                                                ("params_car_residual", 5, 40, 0.5),        # room for 2 of 5 samples only
                                                ("params_pendulum1D_samples", 7, 30, 0.3)])         # all samples while the set is small, then 3 of 7
def test_joint_factor_cache_is_bit_exact_and_used(sg, pname, Ns, H, budget, path):
    """The factor cache of gpmpc_joint_sample: between two resets the hallucinated set only grows, so the rows of the slots
    that were already there are reused (the reference re-factorises everything on every call).  Two MPC steps x four SQP
    iterations with and without the cache: bit-identical Jacobians; the cache is hit at k >= 1 and at k = 0 of the second
    MPC step (the reset-after-build quirk conditions on the previous step's whole set), and dropped when the points change.

    ``path`` (ABI 7, gpmpc_joint_pin_path): on the one-launch VALU path every factor row is the same sequence of FMAs whatever
    is cached - bit-identical, as before.  The matrix-pipe path extends the factor by TILES of new rows against old columns:
    with and without the cache the same entries are summed in a different grouping, so the two runs agree to rounding, not bit
    for bit (include/gpmpc_hip.h says so; a caller that needs bit-equality pins the VALU path) - asserted at 1e-8 on the
    pendulum's Cholesky-root draws and at the north-star tolerance on the car, whose eigendecomposition-root samples move by
    1e-5 under a one-ulp change of the covariance (tests/test_hip_eigh.py prints that figure)."""
    from sampling_gpmpc_amd.gp_model import JointFactorCache
    lib = sg._lib.load()
    lib.gpmpc_joint_pin_path(sg._lib.JOINT_VALU if path == "valu" else sg._lib.JOINT_MFMA)
    try:
        _factor_cache_case(sg, pname, Ns, H, budget, path)
    finally:
        lib.gpmpc_joint_pin_path(sg._lib.JOINT_AUTO)


def _factor_cache_case(sg, pname, Ns, H, budget, path):
    from sampling_gpmpc_amd.gp_model import JointFactorCache

    def same(u, v):
        if path == "valu":
            np.testing.assert_array_equal(u, v)
        else:
            tol = 1e-8 if "pendulum" in pname else 1e-4
            np.testing.assert_allclose(u, v, rtol=tol, atol=tol * max(float(np.abs(v).max()), 1e-300))
    iters = 4
    p = closed_loop_params(pname, Ns, H, 2, iters)
    agent, _ = make_agents(sg, p)
    plain, _ = make_agents(sg, p, erv=agent.epistimic_random_vector.cpu())
    off = JointFactorCache()
    off.enabled = False
    plain._ws_cache["joint_factor_cache"] = off
    if budget is not None:
        # a budget too small for the batch: the factor rows of a PREFIX of the samples are cached, the draw is two launches
        # (car as shipped: the eigh-for-the-whole-batch rule spans both)
        small = JointFactorCache()
        small.MAX_BYTES = int(budget * sg._lib.load().gpmpc_joint_cache_bytes(agent._plan(use_grad=True).desc, Ns, 512))
        agent._ws_cache["joint_factor_cache"] = small
    x0 = np.array(p["env"]["start"], dtype=np.float64)[: agent.nx]
    u_h = np.zeros((H, agent.nu))
    hits, cached_samples = [], []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for step in range(2):
            x_h = np.tile(x0 + 0.01 * step, (H, Ns))
            for k in range(iters):
                outs = []
                for a in (agent, plain):
                    a.mpc_iteration(step)
                    a.train_hallucinated_dynGP(k)
                    outs.append(a.dyn_fg_jacobians(a.get_batch_x_hat(x_h, u_h), k))
                for u, v in zip(*outs):
                    same(u, v)
                hits.append(agent.model_i_call.n_cached_rows)
                cached_samples.append(agent._ws_cache["joint_factor_cache"].n_samples)
                assert plain.model_i_call.n_cached_rows == 0
                mean_next = outs[0][0][:, :, :, 0].mean(axis=0).T
                x_h = np.tile(np.vstack([x0[None, :], mean_next[:-1]]), (1, Ns))
        T = 3
        fc = agent._ws_cache["joint_factor_cache"]
        print(f"{pname}: cached rows per call {hits}; samples cached {fc.n_samples} of {Ns}")
        if budget is None:
            assert fc.n_samples == Ns
        else:
            per_sample = sg._lib.load().gpmpc_joint_cache_bytes(agent.model_i.plan.desc, 1, fc.rows)
            assert 0 < fc.n_samples < Ns and fc.n_samples == fc.MAX_BYTES // per_sample
        assert hits[0] == 0 and hits[1] == 0                         # k = 0: empty set; k = 1: its rows are new
        if budget is None:
            assert hits[2] == H * T and hits[3] == 2 * H * T
            assert hits[4] == 3 * H * T                              # next MPC step, k = 0: the pre-reset set
            assert hits[5] == 0                                      # after the reset: other points
        else:                                                        # (a buffer that had to grow starts empty: fewer hits)
            print(f"samples cached per call {cached_samples}")
            assert any(h > 0 and c < Ns for h, c in zip(hits, cached_samples)), "the prefix cache was never hit"
        # changing a cached point IN PLACE (survivor replacement does that) invalidates the cache: the Agent's lineage counter
        # is told (Agent.invalidate_factor_cache - prepare_dynamics_set calls it; assigning the attribute does it by itself) ...
        agent.Hallcinated_X_train[0, :, 0, :] += 1e-3
        plain.Hallcinated_X_train[0, :, 0, :] += 1e-3
        # (round 5: nobody has to announce the edit - the Agent compares the tensors' in-place version counters when it builds
        # the next model; Agent.invalidate_factor_cache() remains for code that wants to say so)
        outs = []
        for a in (agent, plain):
            a.train_hallucinated_dynGP(iters)
            outs.append(a.dyn_fg_jacobians(a.get_batch_x_hat(x_h, u_h), iters - 1))
        assert agent.model_i_call.n_cached_rows == 0
        for u, v in zip(*outs):
            same(u, v)
        # ... and with GPMPC_VERIFY_FACTOR_CACHE=1 the cache compares the points themselves: an in-place edit nobody announced
        # is caught too
        import sampling_gpmpc_amd.gp_model as gm
        old_flag = gm._VERIFY_CACHE
        gm._VERIFY_CACHE = True
        try:
            agent.Hallcinated_X_train[0, :, 1, :] += 1e-3
            plain.Hallcinated_X_train[0, :, 1, :] += 1e-3
            outs = []
            for a in (agent, plain):
                a.train_hallucinated_dynGP(iters)
                outs.append(a.dyn_fg_jacobians(a.get_batch_x_hat(x_h, u_h), iters - 1))
            assert agent.model_i_call.n_cached_rows == 0
            for u, v in zip(*outs):
                same(u, v)
        finally:
            gm._VERIFY_CACHE = old_flag
