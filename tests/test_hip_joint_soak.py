"""GPU: soaks of the matrix-pipe path of gpmpc_joint_sample (joint_test_mfma_kernel, csrc/joint_mfma.hip) at FULL WIDTH.

The one bug class that kernel has had - an issued FP64 MFMA queued behind the SIMD's other wave reading SrcA / SrcB late
 - produced ~10 % wrong chains at Ns >= 128, different ones run to run, and is invisible below 128 chains: the
oracle cases of tests/test_hip_parity.py run 4-48 chains.  What guards it is here, in the GPU suite: the car closed loop AS
SHIPPED (params_car_residual.yaml, Dyn_gp_jitter 1e-20; reference src/agent.py:629-641 per src/solver.py:84-94) at the per-GPU
shard of BASELINE configs[4] (Ns = 1024, H = 40), the path PINNED and asserted:
  (a) repeat determinism: every draw at k = 1..4 (k = 4: 45 + 480 slots, the TOP + BOTTOM pair of launches) four times from the
      same factor-cache state - every chain's mean / variance / sample bit-equal to repetition 0 - and mean / variance within
      1e-8 of the one-launch VALU path;
  (b) workspace poisoning: the joint workspace NaN-filled in front of every draw - finite, bit-equal to the zero-filled run;
  (c) subset invariance: Ns = 1024 against its first 8 samples alone, bit-equal.
"""
import warnings

import numpy as np
import pytest
import torch

from tests.helpers import load_params
from tests.test_hip_parity import sg  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

H = 40


def _car_agent(sg, Ns, iters, erv=None):
    p = load_params("params_car_residual")
    p["common"]["use_cuda"] = True
    p["agent"]["num_dyn_samples"], p["optimizer"]["H"] = Ns, H
    p["agent"]["true_dyn_as_sample"] = False
    p["agent"]["base_sample_generator"] = "vectorized"
    p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = 1, iters
    a = sg.Agent(p, sg.make_env(p))
    if erv is not None:
        a.epistimic_random_vector = erv
    return a, p


def _u_ff():
    u_h = np.zeros((H, 2))
    u_h[:, 0] = 0.05 * np.sin(2 * np.pi * np.arange(H) / H)
    return u_h


@pytest.fixture
def pinned_mfma(sg):
    lib = sg._lib.load()
    lib.gpmpc_joint_pin_path(sg._lib.JOINT_MFMA)
    yield lib
    lib.gpmpc_joint_pin_path(sg._lib.JOINT_AUTO)


def test_matrix_pipe_repeat_determinism_full_width(sg, pinned_mfma):
    lib, Ns, iters, reps = pinned_mfma, 1024, 5, 4
    torch.manual_seed(3)
    agent, p = _car_agent(sg, Ns, iters)
    g = torch.Generator().manual_seed(5)
    x0 = np.array(p["env"]["start"], dtype=np.float64)
    agent.mpc_iteration(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k in range(iters):
            # scattered linearisation points (every chain its own), as tools/debug/joint_fail_chains.py
            x_h = np.tile(x0, (H, Ns)) + 0.05 * torch.randn(H, Ns * agent.nx, generator=g, dtype=torch.float64).numpy() \
                + 0.02 * np.arange(H)[:, None]
            u_h = 0.3 * torch.randn(H, Ns, agent.nu, generator=g, dtype=torch.float64).numpy()
            agent.train_hallucinated_dynGP(k)
            bx = agent.get_batch_x_hat_u_diff(x_h, u_h)
            g_xu = agent.env_model.get_g_xu_hat(bx).contiguous()
            z = agent.epistimic_random_vector[agent.mpc_iter][k]
            cache = agent._ws_cache.get("joint_factor_cache")
            held = cache.n_valid if cache is not None else 0

            def draw(path):
                lib.gpmpc_joint_pin_path(path)
                c = agent._ws_cache.get("joint_factor_cache")
                if c is not None:
                    c.rewind(held)                                  # every repetition from the same cache state
                post = agent.model_i(g_xu)
                y, _ = post._run(z, True, 2.0, 1e-9, raise_chol_fail=False)
                return post.mean.clone(), post.variance.clone(), y.clone(), post.last_info.clone()

            if k >= 0:      # (k = 0: no hallucinated slot - joint_real_mfma_kernel's test use; k = 1: nothing cached - its factor use)
                truth = draw(sg._lib.JOINT_VALU)
                assert lib.gpmpc_joint_last_path() == sg._lib.JOINT_VALU
                assert not (truth[3] & sg._lib.INFO_TRAIN_CHOL_FAIL).any()
                ref = None
                for rep in range(reps):
                    cur = draw(sg._lib.JOINT_MFMA)
                    assert lib.gpmpc_joint_last_path() == sg._lib.JOINT_MFMA, "the pinned matrix-pipe path did not run"
                    assert not (cur[3] & sg._lib.INFO_TRAIN_CHOL_FAIL).any()
                    if ref is None:
                        ref = cur
                        for i, name in ((0, "mean"), (1, "variance")):
                            e = float(((cur[i] - truth[i]).abs() / truth[i].abs().max()).max())
                            assert e < 1e-8, f"k={k}: {name} of the matrix-pipe path is {e:.1e} from the VALU path"
                    else:
                        for i, name in ((0, "mean"), (1, "variance"), (2, "sample")):
                            ndiff = int((cur[i] != ref[i]).flatten(1).any(-1).sum())
                            assert ndiff == 0, f"k={k} repetition {rep}: {ndiff} samples whose {name} differs from repetition 0"
            # continue the loop on the dispatcher's own choice, from the same cache state
            lib.gpmpc_joint_pin_path(sg._lib.JOINT_MFMA)
            c = agent._ws_cache.get("joint_factor_cache")
            if c is not None:
                c.rewind(held)
            agent.get_batch_gp_sensitivities(bx, k)
    assert agent.model_i.n_h == (iters - 1) * H                     # the last draw conditioned on 45 + 480 slots


def test_matrix_pipe_workspace_poisoning(sg, pinned_mfma):
    lib, Ns, iters = pinned_mfma, 128, 5
    out = {}
    for fill in (0.0, float("nan")):
        torch.manual_seed(11)
        agent, p = _car_agent(sg, Ns, iters)
        x0 = np.array(p["env"]["start"], dtype=np.float64)
        u_h, x_h, res = _u_ff(), np.tile(x0, (H, Ns)), []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for k in range(iters):
                agent.train_hallucinated_dynGP(k)
                ws = agent._ws_cache.get("joint")
                if ws is not None:
                    ws.fill_(fill)
                gp_val, y_grad, _ = agent.dyn_fg_jacobians(agent.get_batch_x_hat(x_h, u_h), k)
                if k >= 1:
                    assert lib.gpmpc_joint_last_path() == sg._lib.JOINT_MFMA
                res.append((gp_val.copy(), y_grad.copy()))
                mean_next = gp_val[:, :, :, 0].mean(axis=0).T
                x_h = np.tile(np.vstack([x0[None, :], mean_next[:-1]]), (1, Ns))
        out[str(fill)] = res
    for k in range(iters):
        a, b = out["0.0"][k], out["nan"][k]
        assert np.isfinite(b[0]).all() and np.isfinite(b[1]).all(), f"k={k}: NaN read from the workspace"
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])


def test_matrix_pipe_subset_invariance_full_width(sg, pinned_mfma):
    lib, Ns, sub, iters = pinned_mfma, 1024, 8, 4
    torch.manual_seed(11)
    agent, p = _car_agent(sg, Ns, iters)
    small, _ = _car_agent(sg, sub, iters, agent.epistimic_random_vector[:, :, :sub].clone())
    x0 = np.array(p["env"]["start"], dtype=np.float64)
    u_h, x_h = _u_ff(), np.tile(x0, (H, Ns))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k in range(iters):
            agent.train_hallucinated_dynGP(k)
            small.train_hallucinated_dynGP(k)
            xs = x_h.reshape(H, Ns, 4)[:, :sub].reshape(H, sub * 4)
            gv, yg, _ = agent.dyn_fg_jacobians(agent.get_batch_x_hat(x_h, u_h), k)
            pa = lib.gpmpc_joint_last_path()
            sv, sy, _ = small.dyn_fg_jacobians(small.get_batch_x_hat(xs, u_h), k)
            ps = lib.gpmpc_joint_last_path()
            if k >= 1:
                assert pa == sg._lib.JOINT_MFMA and ps == sg._lib.JOINT_MFMA
            assert torch.equal(agent.model_i_call.mean[:sub], small.model_i_call.mean), f"k={k}: mean depends on the batch"
            assert torch.equal(agent.model_i_call.variance[:sub], small.model_i_call.variance), f"k={k}: variance depends on the batch"
            np.testing.assert_array_equal(gv[:sub], sv)
            np.testing.assert_array_equal(yg[:sub], sy)
            mean_next = gv[:, :, :, 0].mean(axis=0).T
            x_h = np.tile(np.vstack([x0[None, :], mean_next[:-1]]), (1, Ns))


@pytest.mark.parametrize("Hs", [11, 12, 13])
def test_eigh_root_full_rank_just_beyond_the_narrow_cap(sg, Hs):
    """m T in 33..39 (T = 3, H = 11..13) with FULL-RANK covariances under a forced eigendecomposition root.  A pass of the
    pivoted Cholesky adds up to 8 pivots, so such a chain reaches rank m T > 32 (the narrow launch's LDS cap) without ever
    passing the deferral check at the top of a pass; it has to be handed to the second launch all the same (it used to take the
    HBM Gram path with an aliased buffer).  Scattered, well separated test points make the covariance full rank; asserted:
    R R^T == max(Sigma, 0), the draw reproduces mean + R z, and the chains WERE deferred."""
    import ctypes as C
    from tests.test_hip_eigh import _car
    from tests.test_hip_parity import make_agents
    Ns = 6
    p = _car(Ns, Hs, 1, jitter=1e-9)
    agent, _ = make_agents(sg, p)
    g = torch.Generator().manual_seed(7)
    x_h = np.tile(np.array(p["env"]["start"]), (Hs, Ns))
    x_h = x_h + np.repeat(np.linspace(-0.9, 0.9, Hs)[:, None], Ns * 4, axis=1) * np.tile([0, 0, 1.0, 0], Ns)[None, :] \
        + 0.05 * torch.randn(Hs, Ns * 4, generator=g, dtype=torch.float64).numpy()
    u_h = 0.5 * torch.randn(Hs, Ns, 2, generator=g, dtype=torch.float64).numpy()
    agent.train_hallucinated_dynGP(0)
    gx = agent.env_model.get_g_xu_hat(agent.get_batch_x_hat_u_diff(x_h, u_h)).contiguous()
    z = agent.epistimic_random_vector[0][0]
    post = agent.model_i(gx)
    raw = sg._lib.load()
    raw.gpmpc_debug_eigh_deferred.restype = C.c_longlong
    raw.gpmpc_debug_eigh_deferred(1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y, bits = post._run(z, clip=False, want_root=True, root_mode=sg._lib.ROOT_EIGH)
    deferred = int(raw.gpmpc_debug_eigh_deferred(0))
    assert bits & sg._lib.INFO_ROOT_EIGH and not bits & sg._lib.INFO_EIGH_NOCONV
    R, S = post.root.cpu(), post.covariance_matrix.cpu()
    rank = int((R.abs().amax(dim=-2) > 0).sum(dim=-1).max())
    ev, U = torch.linalg.eigh(S)
    Splus = (U * ev.clamp_min(0).unsqueeze(-2)) @ U.transpose(-1, -2)
    err = float((R @ R.transpose(-1, -2) - Splus).abs().max())
    print(f"H={Hs}: m T = {3 * Hs}, max rank {rank}, chains deferred to the second launch {deferred}, |RR^T - S+| {err:.1e}")
    assert err < 1e-8 * float(S.abs().max())
    zz = z.reshape(Ns, 3, -1, 1).cpu()
    np.testing.assert_allclose((y.cpu() - post.mean.cpu()).reshape(Ns, 3, -1), (R @ zz).squeeze(-1), rtol=0, atol=1e-12)
    if rank > 32:
        assert deferred > 0, "a chain of rank > 32 stayed in the narrow launch"


@pytest.mark.parametrize("pname,Ns,Hh", [("params_car_residual", 64, 40), ("params_pendulum1D_samples", 24, 30)])
def test_pending_rows_of_the_factor_cache(sg, pname, Ns, Hh, monkeypatch):
    """gpmpc_joint_sample_pending (ABI 9): in the SQP loop a draw's test points become the next call's new hallucinated points
    (reference src/agent.py:629-641 then :164-202), so the draw leaves X^T = the new rows against the old columns and S = their Schur
    complement (before the likelihood noise) in the cache, and the next call only factorises (S + noise) in place.  Two MPC steps x four
    SQP iterations with the matrix-pipe path pinned, pending rows on against off (GPMPC_JOINT_PENDING=0): the rows are USED from the
    third iteration of a step on and at iteration 0 of the next step (the reset-after-build quirk), never right after the reset; mean /
    variance agree to rounding, the samples to the tolerance of the configuration's root (Cholesky: 1e-8; the car's eigh root: 1e-4)."""
    from tests.helpers import closed_loop_params
    lib = sg._lib.load()
    iters = 4
    p = closed_loop_params(pname, Ns, Hh, 2, iters)
    p["common"]["use_cuda"] = True
    p["agent"]["base_sample_generator"] = "counter"
    x0 = np.array(p["env"]["start"], dtype=np.float64)
    runs = {}
    lib.gpmpc_joint_pin_path(sg._lib.JOINT_MFMA)
    try:
        for mode in ("1", "0"):
            monkeypatch.setenv("GPMPC_JOINT_PENDING", mode)
            agent = sg.Agent(p, sg.make_env(p))
            x0a = x0[: agent.nx]
            u_h, x_h = np.zeros((Hh, agent.nu)), np.tile(x0a, (Hh, Ns))
            rec = []
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for step in range(2):
                    agent.mpc_iteration(step)
                    for k in range(iters):
                        agent.train_hallucinated_dynGP(k)
                        gv, yg, _ = agent.dyn_fg_jacobians(agent.get_batch_x_hat(x_h, u_h), k)
                        post = agent.model_i_call
                        rec.append((bool(post.used_pending), post.mean.clone(), post.variance.clone(), gv.copy(),
                                    int(lib.gpmpc_joint_last_path())))
                        assert not (post.last_info & sg._lib.INFO_TRAIN_CHOL_FAIL).any()
                        mean_next = gv[:, :, :, 0].mean(axis=0).T
                        x_h = np.tile(np.vstack([x0a[None, :], mean_next[:-1]]), (1, Ns))
            runs[mode] = rec
    finally:
        lib.gpmpc_joint_pin_path(sg._lib.JOINT_AUTO)
    used = [r[0] for r in runs["1"]]
    print(f"{pname}: pending rows used per call {used}")
    assert used == [False, False, True, True, True, False, True, True] and not any(r[0] for r in runs["0"])
    # (the car's eigh-root samples move by ~1e-5 under a one-ulp change of the covariance, and the samples are the next iteration's
    # labels and - through their mean - linearisation points: from the fourth call on the two runs are not fed the same inputs)
    tol_y, tol_m = (1e-8, 1e-9) if "pendulum" in pname else (1e-4, 1e-6)
    for i, (a, b) in enumerate(zip(runs["1"], runs["0"])):
        for j, name in ((1, "mean"), (2, "variance")):
            e = float(((a[j] - b[j]).abs() / b[j].abs().max()).max())
            assert e < tol_m, f"call {i}: {name} with pending rows is {e:.1e} from the recomputed rows"
        ey = float(np.abs(a[3] - b[3]).max() / np.abs(b[3]).max())
        assert ey < tol_y, f"call {i}: samples {ey:.1e}"


@pytest.mark.parametrize("pname,Ns,Hh", [("params_car_residual", 96, 40), ("params_car_residual", 8, 13), ("params_pendulum1D_samples", 40, 30),
                                         ("params_pendulum1D_samples", 5, 7)])
def test_real_data_kernel_against_the_launches_it_replaces(sg, pname, Ns, Hh):
    """joint_real_mfma_kernel (columns conditioned on the real data alone, one wave per chain): its TEST use - the draw without hallucinated
    slots, k = 0 of the first MPC step - against joint_kernel's head, and its FACTOR use - the factor extension with nothing cached, k = 1 -
    against joint_test_mfma_kernel's factor mode (matrix-pipe path pinned so that every shape takes it): same Agent sequence with the kernel
    on and off (gpmpc_debug_joint_real_kernel), mean / variance to rounding, samples to the tolerance of the configuration's root.  Odd
    point counts (H = 13, 7) put the pair tables' point block on an odd offset."""
    from tests.helpers import closed_loop_params
    lib = sg._lib.load()
    p = closed_loop_params(pname, Ns, Hh, 1, 3)
    p["common"]["use_cuda"] = True
    p["agent"]["base_sample_generator"] = "counter"
    x0 = np.array(p["env"]["start"], dtype=np.float64)
    runs = {}
    lib.gpmpc_joint_pin_path(sg._lib.JOINT_MFMA)
    try:
        for mode in (1, 0):
            lib.gpmpc_debug_joint_real_kernel(mode)
            agent = sg.Agent(p, sg.make_env(p))
            x0a = x0[: agent.nx]
            u_h, x_h = 0.1 * np.ones((Hh, agent.nu)), np.tile(x0a, (Hh, Ns)) + 0.01 * np.arange(Hh)[:, None]
            rec = []
            agent.mpc_iteration(0)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for k in range(2):
                    agent.train_hallucinated_dynGP(k)
                    gv, _, _ = agent.dyn_fg_jacobians(agent.get_batch_x_hat(x_h, u_h), k)
                    post = agent.model_i_call
                    assert not (post.last_info & sg._lib.INFO_TRAIN_CHOL_FAIL).any()
                    rec.append((post.mean.clone(), post.variance.clone(), gv.copy(), int(lib.gpmpc_joint_last_path())))
            runs[mode] = rec
            del agent
    finally:
        lib.gpmpc_debug_joint_real_kernel(-1)
        lib.gpmpc_joint_pin_path(sg._lib.JOINT_AUTO)
    # k = 0 without the kernel is joint_kernel's head (the VALU path: nothing else is instantiated for an empty hallucinated set)
    assert [r[3] for r in runs[1]] == [sg._lib.JOINT_MFMA] * 2 and runs[0][0][3] == sg._lib.JOINT_VALU
    # (same blocked substitution over the real block's tiles as the launches replaced, other summation order: 1e-10 of the largest variance on
    # the car - whose posterior variances are 1e-4 of the prior's: S = K - X^T X cancels - and 1e-13 on the mean)
    tol_y, tol_m = (1e-8, 1e-10) if "pendulum" in pname else (1e-4, 1e-9)
    for k, (a, b) in enumerate(zip(runs[1], runs[0])):
        for j, name in ((0, "mean"), (1, "variance")):
            e = float(((a[j] - b[j]).abs() / b[j].abs().max()).max())
            assert e < tol_m, f"k = {k}: {name} with joint_real_mfma_kernel is {e:.1e} from the launch it replaces"
        ey = float(np.abs(a[2] - b[2]).max() / np.abs(b[2]).max())
        assert ey < tol_y, f"k = {k}: samples {ey:.1e}"


def test_closed_loop_sequence_is_deterministic_at_full_width(sg):
    """The closed loop's OWN sequence at the configs[4] shard (Ns = 1024, H = 40, two MPC steps x four SQP iterations, dispatcher's choice
    of path, pending rows on): two fresh Agents with the same base samples give bit-equal means, variances and Jacobians at every call -
    joint_chol_mfma_kernel (in place on the pending block), the test-mode launch that writes the pending rows, joint_tail_mfma_kernel and
    the TOP + BOTTOM pair all run at full width here (the repeat-determinism test above rewinds the cache, which drops the pending rows)."""
    from tests.helpers import closed_loop_params
    lib = sg._lib.load()
    Ns, iters = 1024, 4
    p = closed_loop_params("params_car_residual", Ns, H, 2, iters)
    p["common"]["use_cuda"] = True
    p["agent"]["base_sample_generator"] = "counter"
    x0 = np.array(p["env"]["start"], dtype=np.float64)[:4]
    runs = []
    for rep in range(2):
        agent = sg.Agent(p, sg.make_env(p))
        u_h, x_h, rec = _u_ff(), np.tile(x0, (H, Ns)), []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for step in range(2):
                agent.mpc_iteration(step)
                for k in range(iters):
                    agent.train_hallucinated_dynGP(k)
                    if agent.model_i.n_h:   # the Agent skips the observed-slot scan when every label came out of a finite draw (the eigh
                        from sampling_gpmpc_amd.gp_model import _observed_slots    # redraw of the car included): the scan has to agree
                        assert torch.equal(agent.model_i.h_slots, _observed_slots(agent.model_i.hall_Y)) and (k == 0 or agent._hall_all_observed)
                    gv, yg, _ = agent.dyn_fg_jacobians(agent.get_batch_x_hat(x_h, u_h), k)
                    post = agent.model_i_call
                    rec.append((post.mean.clone(), post.variance.clone(), gv.copy(), yg.copy(), bool(post.used_pending),
                                int(lib.gpmpc_joint_last_path())))
                    assert np.isfinite(gv).all() and not (post.last_info & sg._lib.INFO_TRAIN_CHOL_FAIL).any()
                    mean_next = gv[:, :, :, 0].mean(axis=0).T
                    x_h = np.tile(np.vstack([x0[None, :], mean_next[:-1]]), (1, Ns))
        runs.append(rec)
        del agent
    assert [r[4] for r in runs[0]] == [False, False, True, True, True, False, True, True]
    assert [r[5] for r in runs[0]] == [sg._lib.JOINT_MFMA] * 8
    for i, (a, b) in enumerate(zip(*runs)):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), f"call {i}: mean / variance differ between two runs"
        np.testing.assert_array_equal(a[2], b[2])
        np.testing.assert_array_equal(a[3], b[3])
