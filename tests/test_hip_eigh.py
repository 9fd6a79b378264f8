"""GPU parity of the in-kernel eigendecomposition root: the branch the SHIPPED
params_car_residual.yaml (Dyn_gp_jitter 1e-20, yaml:51) takes on every mode-J draw (reference src/agent.py:629-708).

What is well defined and therefore asserted (eigenvector signs are solver specific - in LAPACK too - and the
eigenvectors of the +-1e-16 round-off eigenvalues of the numerically singular covariance are arbitrary):
  * R R^T == max(Sigma, 0) for the root the kernel used;
  * the samples equal the oracle's (torch.linalg.eigh root) after aligning the sign of every column of the oracle's root
    with the kernel's, to the noise floor the oracle itself shows when Sigma is perturbed by one ulp;
  * mean / variance as for the Cholesky branch; info bits; whole-batch fallback semantics.
"""
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import agent_oracle as ao
from oracle.gp_oracle import F64
from tests.helpers import load_params
from tests.test_hip_parity import make_agents, relerr, sg  # noqa: F401  (sg is a fixture)

pytestmark = pytest.mark.gpu


def _car(Ns, H, iters, jitter=None):
    p = load_params("params_car_residual")
    p["agent"]["num_dyn_samples"], p["optimizer"]["H"] = Ns, H
    p["agent"]["true_dyn_as_sample"] = False
    p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = 1, iters
    if jitter is not None:
        p["agent"]["Dyn_gp_jitter"] = jitter
    return p


def _sign_aligned_oracle_sample(opost, R_hip, z, beta):
    """Oracle sample with every column of its eigh root given the sign of the kernel's column; clipped like sample_gp."""
    S = opost.covariance_matrix
    evals, evecs = torch.linalg.eigh(S)
    Ro = evecs * evals.clamp_min(0.0).sqrt().unsqueeze(-2)
    sgn = torch.sign((Ro * R_hip).sum(dim=-2, keepdim=True))
    sgn = torch.where(sgn == 0, torch.ones_like(sgn), sgn)
    bshape = opost.mean.shape[:2]
    y = ((Ro * sgn) @ z.reshape(*bshape, -1, 1)).squeeze(-1).reshape(opost.mean.shape) + opost.mean
    sd = beta * opost.variance.sqrt()
    return torch.min(torch.max(y, opost.mean - sd), opost.mean + sd), Ro


def _one_ulp_floor(opost, z, beta):
    """How far the oracle's own sample moves when Sigma is perturbed by one ulp per entry (sign aligned)."""
    S = opost.covariance_matrix
    g = torch.Generator().manual_seed(1)
    P = S * (1 + np.finfo(np.float64).eps * (2 * torch.rand(S.shape, generator=g, dtype=F64) - 1))
    P = (P + P.transpose(-1, -2)) / 2
    ev, U = torch.linalg.eigh(P)
    R2 = U * ev.clamp_min(0.0).sqrt().unsqueeze(-2)
    y1, Ro = _sign_aligned_oracle_sample(opost, R2, z, beta)
    bshape = opost.mean.shape[:2]
    y2 = (R2 @ z.reshape(*bshape, -1, 1)).squeeze(-1).reshape(opost.mean.shape) + opost.mean
    sd = beta * opost.variance.sqrt()
    y2 = torch.min(torch.max(y2, opost.mean - sd), opost.mean + sd)
    return float((y1 - y2).abs().max())


@pytest.mark.parametrize("Ns,H,iters,global_g", [(6, 40, 4, False), (4, 40, 2, True), (5, 12, 3, False), (3, 70, 1, False)])
def test_eigh_root_against_oracle(sg, Ns, H, iters, global_g, monkeypatch):
    """Shipped car_residual jitter: every chain fails the jitter chain, the whole batch takes the eigh root - in-kernel.
    H=70 has m*T = 210 > the LDS rank cap only when the rank exceeds 60; global_g forces the Gram matrix into HBM/L2."""
    if global_g:
        monkeypatch.setenv("GPMPC_EIGH_GLOBAL_G", "1")
    p = _car(Ns, H, iters)
    agent, oagent = make_agents(sg, p)
    agent.debug_keep_root = True
    beta = p["agent"]["Dyn_gp_beta"]
    g = torch.Generator().manual_seed(5)
    x0 = np.array(p["env"]["start"], dtype=np.float64)
    for it in range(iters):
        x_h = np.tile(x0, (H, Ns)) + 0.05 * torch.randn(H, Ns * 4, generator=g, dtype=F64).numpy() \
            + 0.02 * np.arange(H)[:, None]
        u_h = 0.3 * torch.randn(H, Ns, 2, generator=g, dtype=F64).numpy()
        agent.train_hallucinated_dynGP(it)
        oagent.train_hallucinated_dynGP(it)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            gp_val, y_grad, u_grad = agent.dyn_fg_jacobians(agent.get_batch_x_hat_u_diff(x_h, u_h), it)
        oagent.dyn_fg_jacobians(oagent.get_batch_x_hat_u_diff(x_h, u_h), it)
        post, opost = agent.model_i_call, oagent.model_i_call
        assert opost.root_info.used_eigh, "oracle expected to take the eigh branch at jitter 1e-20"
        info = post.last_info.cpu().numpy()
        assert (info & sg._lib.INFO_ROOT_EIGH).all() and not (info & sg._lib.INFO_EIGH_NOCONV).any()
        assert (info & sg._lib.INFO_ROOT_FAIL).all() and (((info >> 1) & 7) == 3).all()
        np.testing.assert_allclose(post.mean.cpu().numpy(), opost.mean.numpy(), rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(post.variance.cpu().numpy(), opost.variance.numpy(), rtol=1e-4, atol=1e-12)
        R = post.root.cpu()
        So = opost.covariance_matrix
        z = oagent.epistimic_random_vector[0][it]
        yo, Ro = _sign_aligned_oracle_sample(opost, R, z, beta)
        Splus = Ro @ Ro.transpose(-1, -2)
        err_rrt = float((R @ R.transpose(-1, -2) - Splus).abs().max())
        smax = float(So.abs().max())
        y = agent.model_i_samples.cpu()
        floor = _one_ulp_floor(opost, z, beta)
        err_y = float((y - yo).abs().max())
        rank = int((R.abs().amax(dim=-2) > 0).sum(dim=-1).max())
        print(f"car eigh Ns={Ns} H={H} it={it}: |RR^T - S+| {err_rrt:.1e} (max|S| {smax:.1e}), sample err {err_y:.2e} "
              f"(oracle's own 1-ulp floor {floor:.2e}; max|y| {float(yo.abs().max()):.2e}), max rank {rank}")
        assert err_rrt < 1e-9 and err_rrt < 1e-8 * smax
        assert err_y < max(20 * floor, 1e-7)
        assert err_y < 1e-4 * float(yo.abs().max())                     # north-star tolerance on the sampled labels
        # continue both agents from the same labels
        agent.Hallcinated_X_train = oagent.Hallcinated_X_train.to(agent.torch_device)
        agent.Hallcinated_Y_train = oagent.Hallcinated_Y_train.to(agent.torch_device)


def test_eigh_whole_batch_fallback_and_root_modes(sg):
    """gpytorch falls back for the WHOLE batch when any element fails: ROOT_EIGH (what a rank without a failing chain
    runs when another rank had one) redraws chains whose Cholesky succeeded; ROOT_CHOLESKY never falls back."""
    Ns, H = 4, 10
    p = _car(Ns, H, 1, jitter=1e-9)
    agent, oagent = make_agents(sg, p)
    g = torch.Generator().manual_seed(2)
    x_h = np.tile(np.array(p["env"]["start"]), (H, Ns)) + 0.05 * torch.randn(H, Ns * 4, generator=g, dtype=F64).numpy()
    u_h = 0.3 * torch.randn(H, Ns, 2, generator=g, dtype=F64).numpy()
    agent.train_hallucinated_dynGP(0)
    oagent.train_hallucinated_dynGP(0)
    gx = agent.env_model.get_g_xu_hat(agent.get_batch_x_hat_u_diff(x_h, u_h)).contiguous()
    z = agent.epistimic_random_vector[0][0]
    post = agent.model_i(gx)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y_chol, bits = post._run(z, clip=False)
        assert not bits & sg._lib.INFO_ROOT_EIGH          # jitter 1e-9: the Cholesky chain succeeds
        y_eigh, bits = post._run(z, clip=False, want_root=True, root_mode=sg._lib.ROOT_EIGH)
        assert bits & sg._lib.INFO_ROOT_EIGH
        R = post.root.cpu()
        S = post.covariance_matrix.cpu()
    ev, U = torch.linalg.eigh(S)
    Splus = (U * ev.clamp_min(0).unsqueeze(-2)) @ U.transpose(-1, -2)
    assert float((R @ R.transpose(-1, -2) - Splus).abs().max()) < 1e-8 * float(S.abs().max())
    # same distribution, different root: the two draws differ but both reproduce their own root
    zz = z.reshape(Ns, 3, -1, 1).cpu()
    np.testing.assert_allclose((y_eigh.cpu() - post.mean.cpu()).reshape(Ns, 3, -1), (R @ zz).squeeze(-1), rtol=0, atol=1e-12)
    assert float((y_eigh - y_chol).abs().max()) > 1e-6
    # shipped jitter + ROOT_CHOLESKY: failing chains return NaN and say so
    p2 = _car(Ns, H, 1)
    agent2, _ = make_agents(sg, p2)
    agent2.train_hallucinated_dynGP(0)
    post2 = agent2.model_i(gx)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y_nan, bits = post2._run(z, clip=False, root_mode=sg._lib.ROOT_CHOLESKY)
    assert bits & sg._lib.INFO_ROOT_FAIL and not bits & sg._lib.INFO_ROOT_EIGH
    assert torch.isnan(y_nan).any()


def test_config5_shard_as_shipped(sg):
    """BASELINE configs[4] on its per-GPU shard, AS SHIPPED (params_car_residual.yaml, Dyn_gp_jitter 1e-20): Ns = 8192 / 8
    = 1024 samples, H = 40, SQP iterations k = 0..3 with the hallucinated set growing by H points per iteration
    (reference src/solver.py:84-94; linearisation points from the deterministic surrogate of SURVEY.md 8d).
    Finite outputs, info bits, sample-subset invariance (bit exact), an 8-sample subset against the oracle."""
    Ns, H, iters, sub = 1024, 40, 4, 8
    p = _car(Ns, H, iters)
    p["agent"]["base_sample_generator"] = "vectorized"
    torch.manual_seed(11)
    agent, _ = make_agents(sg, p)
    erv = agent.epistimic_random_vector.cpu()
    ps = _car(sub, H, iters)
    small, osmall = make_agents(sg, ps, erv=erv[:, :, :sub])
    small.debug_keep_root = True
    m0, msub = Ns // 2 - 8, 16                                      # a 16-sample window from the MIDDLE of the batch as well
    mid, _ = make_agents(sg, _car(msub, H, iters), erv=erv[:, :, m0:m0 + msub])
    raw = sg._lib.load()
    beta = p["agent"]["Dyn_gp_beta"]
    x0 = np.array(p["env"]["start"], dtype=np.float64)
    u_h = np.zeros((H, 2))
    u_h[:, 0] = 0.05 * np.sin(2 * np.pi * np.arange(H) / H)
    x_h = np.tile(x0, (H, Ns))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k in range(iters):
            for a in (agent, small, osmall, mid):
                a.train_hallucinated_dynGP(k)
            xs = x_h.reshape(H, Ns, 4)[:, :sub].reshape(H, sub * 4)
            xm = x_h.reshape(H, Ns, 4)[:, m0:m0 + msub].reshape(H, msub * 4)
            gp_val, y_grad, u_grad = agent.dyn_fg_jacobians(agent.get_batch_x_hat(x_h, u_h), k)
            # the kernel path the dispatcher takes at the shipped shape: the matrix pipe from the second SQP iteration on
            # (120 hallucinated slots >= GPMPC_JOINT_MFMA_FROM); a dispatcher change that silently drops it fails here
            assert raw.gpmpc_joint_last_path() == sg._lib.JOINT_MFMA, k     # (k = 0: no hallucinated slot, joint_real_mfma_kernel)
            sv, sy, su = small.dyn_fg_jacobians(small.get_batch_x_hat(xs, u_h), k)
            mv, my, _ = mid.dyn_fg_jacobians(mid.get_batch_x_hat(xm, u_h), k)
            np.testing.assert_array_equal(gp_val[m0:m0 + msub], mv)
            np.testing.assert_array_equal(y_grad[m0:m0 + msub], my)
            osmall.dyn_fg_jacobians(osmall.get_batch_x_hat(xs, u_h), k)
            info = agent.model_i_call.last_info.cpu().numpy()
            assert np.isfinite(gp_val).all() and np.isfinite(y_grad).all() and np.isfinite(u_grad).all()
            assert (info & sg._lib.INFO_ROOT_EIGH).all() and not (info & sg._lib.INFO_EIGH_NOCONV).any()
            assert not (info & sg._lib.INFO_TRAIN_CHOL_FAIL).any()
            # after a redraw every chain reports the BATCH's outcome (all retries exhausted, root failed, eigh root): chains
            # that see another chain's failure abandon their own Cholesky attempts, and which ones do is a matter of timing
            assert (info & sg._lib.INFO_ROOT_FAIL).all() and (((info >> 1) & 7) == 3).all()
            # a chain's draw does not depend on which other chains are in the launch
            np.testing.assert_array_equal(gp_val[:sub], sv)
            np.testing.assert_array_equal(y_grad[:sub], sy)
            post, opost = small.model_i_call, osmall.model_i_call
            assert opost.root_info.used_eigh
            np.testing.assert_allclose(post.mean.cpu().numpy(), opost.mean.numpy(), rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(post.variance.cpu().numpy(), opost.variance.numpy(), rtol=1e-4, atol=1e-12)
            z = osmall.epistimic_random_vector[0][k]
            yo, _ = _sign_aligned_oracle_sample(opost, post.root.cpu(), z, beta)
            err = float((small.model_i_samples.cpu() - yo).abs().max())
            floor = _one_ulp_floor(opost, z, beta)
            n_o = agent.model_i.plan.n_r + agent.model_i.n_h * 3
            print(f"config 5 shard k={k}: n_o={n_o}, 8-sample subset vs oracle: sample err {err:.2e} "
                  f"(1-ulp floor {floor:.2e}, max|y| {float(yo.abs().max()):.2e})")
            assert err < max(20 * floor, 1e-7) and err < 1e-4 * float(yo.abs().max())
            # the oracle continues from the kernel's labels (its own eigh signs differ), all three from the same points
            osmall.Hallcinated_X_train = small.Hallcinated_X_train.cpu()
            osmall.Hallcinated_Y_train = small.Hallcinated_Y_train.cpu()
            mean_next = gp_val[:, :, :, 0].mean(axis=0).T
            x_h = np.tile(np.vstack([x0[None, :], mean_next[:-1]]), (1, Ns))


def test_eigh_root_value_only_model_rank_one(sg):
    """T = 1 (env.use_model_without_derivatives) joint draw whose test points all coincide - what iteration 0 of the closed
    loop looks like: the posterior covariance has rank ONE, every Cholesky attempt fails at the second pivot, the eigh root
    is a single column.  Exercises joint_eigh_kernel<1, *> and the smallest Jacobi (r = 1, padded to 2)."""
    from tests.helpers import fs_params
    Ns, H = 5, 20
    p = fs_params("params_car_residual_fs", Ns, 1, nograd=True)
    p["optimizer"]["H"] = H
    p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = 1, 1
    p["agent"]["Dyn_gp_jitter"] = 1e-20
    agent, oagent = make_agents(sg, p)
    agent.debug_keep_root = True
    x_h = np.tile(np.array(p["env"]["start"], dtype=np.float64), (H, Ns))
    u_h = np.zeros((H, 2))
    for a in (agent, oagent):
        a.train_hallucinated_dynGP(0, use_model_without_derivatives=True)
    # whether the second pivot of the rank-one matrix comes out as +-1 ulp is a round-off coin flip (LAPACK: eigh branch;
    # the kernel's Cholesky may pass its first retry): the eigh root is requested explicitly on the HIP side
    post = agent.model_i(agent.env_model.get_g_xu_hat(agent.get_batch_x_hat(x_h, u_h)).contiguous())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y, _ = post._run(agent.epistimic_random_vector[0][0], clip=True, beta=p["agent"]["Dyn_gp_beta"],
                         want_root=True, root_mode=sg._lib.ROOT_EIGH)
    yo = oagent.sample_gp(oagent.env_model.get_g_xu_hat(oagent.get_batch_x_hat(x_h, u_h)),
                          base_samples=oagent.epistimic_random_vector[0][0])
    opost = oagent.model_i_call
    assert opost.root_info.used_eigh and (post.last_info.cpu().numpy() & sg._lib.INFO_ROOT_EIGH).all()
    R = post.root.cpu()
    rank = int((R.abs().amax(dim=-2) > 0).sum(dim=-1).max())
    ysa, Ro = _sign_aligned_oracle_sample(opost, R, oagent.epistimic_random_vector[0][0], p["agent"]["Dyn_gp_beta"])
    err = float((y.cpu() - ysa).abs().max())
    print(f"value-only rank-one joint draw: rank {rank}, sample err {err:.2e} (max|y| {float(ysa.abs().max()):.2e})")
    assert rank <= 2
    np.testing.assert_allclose(post.mean.cpu().numpy(), opost.mean.numpy(), rtol=1e-6, atol=1e-10)
    assert err < 1e-6 * float(ysa.abs().max()) + 1e-9


def test_eigh_root_against_50_digit_eigensolver(sg):
    """the kernel's eigendecomposition root against 50-digit arithmetic on a <= 8-slot singular covariance.
    Two stages of the car horizon at the SAME linearisation point give a 6 x 6 posterior covariance of rank 3; the kernel's
    own Sigma (covariance_matrix) is decomposed with mpmath's symmetric eigensolver at 50 digits, and the root the kernel
    drew with must satisfy R R^T == max(Sigma, 0) to FP64 round-off of Sigma's scale (no LAPACK in the comparison)."""
    from tests.test_oracle_independent import _psd_part_mp
    Ns, H = 3, 2
    p = _car(Ns, H, 1)
    agent, _ = make_agents(sg, p)
    agent.debug_keep_root = True
    x_h = np.tile(np.array(p["env"]["start"], dtype=np.float64)[:4], (H, Ns))
    u_h = np.array([[0.03, 0.0], [0.03, 0.0]])
    agent.train_hallucinated_dynGP(0)
    g_in = agent.env_model.get_g_xu_hat(agent.get_batch_x_hat(x_h, u_h)).contiguous()
    post = agent.model_i(g_in)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        post._run(agent.epistimic_random_vector[0][0], clip=True, beta=p["agent"]["Dyn_gp_beta"], want_root=True,
                  want_covar=True, root_mode=sg._lib.ROOT_EIGH)
    S_all, R_all = post.covariance_matrix.cpu().numpy(), post.root.cpu().numpy()
    worst = 0.0
    for s in range(Ns):
        for o in range(3):
            S = 0.5 * (S_all[s, o] + S_all[s, o].T)
            P, lam = _psd_part_mp(S)
            R = R_all[s, o]
            scale = np.abs(S).max()
            err = np.abs(R @ R.T - P).max() / scale
            worst = max(worst, err)
            assert int((lam > 1e-12 * lam.max()).sum()) <= 3          # rank 3: the second stage repeats the first
    print(f"kernel eigh root vs 50-digit max(Sigma, 0) on 6 x 6 rank-3 covariances: max |R R^T - P| / max|Sigma| = {worst:.2e}")
    assert worst < 1e-12


def test_eigh_narrow_and_deferred_launches_give_the_same_bits(sg):
    """Round 5: batches of low rank run a NARROW instantiation of the eigh kernel first (LDS for ranks <= 32, twice the chains per
    CU); a chain beyond that rank is deferred to the full instantiation in a second launch.  Which launch handles a chain must not
    show in its result: the same draws with the two-launch form forced off and on - on smooth points (ranks ~10: nothing
    deferred), on scattered points (ranks ~50: everything deferred) and on a batch that mixes both."""
    import ctypes as C
    raw = sg._lib.load()
    raw.gpmpc_debug_eigh_deferred.restype = C.c_longlong
    Ns, H, iters = 48, 40, 2
    p = _car(Ns, H, iters)
    p["agent"]["base_sample_generator"] = "vectorized"
    g = torch.Generator().manual_seed(17)
    x0 = np.array(p["env"]["start"], dtype=np.float64)
    smooth = np.tile(x0, (H, Ns)) + 0.02 * np.arange(H)[:, None]
    rough = smooth + 0.05 * torch.randn(H, Ns * 4, generator=g, dtype=F64).numpy()
    mixed = smooth.copy()
    mixed[:, : (Ns // 2) * 4] = rough[:, : (Ns // 2) * 4]
    u_s = np.zeros((H, Ns, 2))
    u_r = 0.3 * torch.randn(H, Ns, 2, generator=g, dtype=F64).numpy()
    u_m = u_s.copy()
    u_m[:, : Ns // 2] = u_r[:, : Ns // 2]
    try:
        # (chains the narrow launch must defer: none on the smooth points, most on the scattered ones - a few of those stay below
        # rank 32 -, some but not all of the mixed batch)
        for name, x_h, u_h, lo, hi in (("smooth", smooth, u_s, 0, 0), ("rough", rough, u_r, 2 * Ns, 3 * Ns),
                                       ("mixed", mixed, u_m, Ns, 3 * Ns - Ns)):
            torch.manual_seed(5)
            agent, _ = make_agents(sg, p)
            agent.mpc_iteration(0)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for k in range(iters):
                    agent.train_hallucinated_dynGP(k)
                    bx = agent.get_batch_x_hat_u_diff(x_h, u_h)
                    g_xu = agent.env_model.get_g_xu_hat(bx).contiguous()
                    z = agent.epistimic_random_vector[agent.mpc_iter][k]
                    cache = agent._ws_cache.get("joint_factor_cache")
                    held = cache.n_valid if cache is not None else 0
                    out = {}
                    for mode in (0, 1):
                        raw.gpmpc_debug_eigh_narrow(mode)
                        raw.gpmpc_debug_eigh_deferred(1)
                        if agent._ws_cache.get("joint_factor_cache") is not None:
                            agent._ws_cache["joint_factor_cache"].rewind(held)
                        y = agent.sample_gp(g_xu, base_samples=z).clone()
                        torch.cuda.synchronize()
                        out[mode] = (y, agent.model_i_call.last_info.clone(), int(raw.gpmpc_debug_eigh_deferred(0)))
                    assert (out[0][1] & sg._lib.INFO_ROOT_EIGH).all()
                    assert torch.equal(out[0][0], out[1][0]), f"{name} k={k}: the two-launch form changed a sample"
                    assert torch.equal(out[0][1], out[1][1])
                    assert out[0][2] == 0 and lo <= out[1][2] <= hi, (name, k, out[0][2], out[1][2])
                    raw.gpmpc_debug_eigh_narrow(-1)
                    if agent._ws_cache.get("joint_factor_cache") is not None:
                        agent._ws_cache["joint_factor_cache"].rewind(held)
                    agent.get_batch_gp_sensitivities(bx, k)
    finally:
        raw.gpmpc_debug_eigh_narrow(-1)


@pytest.mark.parametrize("suffix", ["", "_gpytorch"])
def test_joint_draws_of_the_closed_loop_against_reference_run(sg, suffix):
    """agent_e2e_J_car_split.npz (committed): the reference's own Agent through these draws with the import stub's algebra.
    The day gpytorch exists, ``tests/golden/make_goldens.py --real-gpytorch`` writes the closed-loop draws of the car as shipped
    (k = 0..3 and the 45 + 480-slot k = 0 of the next MPC step: the matrix-pipe path's TOP + BOTTOM launches) from the REAL
    ``model_i(x)``; this test then pins the HIP path - dispatcher's choice, path asserted - against them.  Skipped until then."""
    path = os.path.join(os.path.dirname(__file__), "golden", f"agent_e2e_J_car_split{suffix}.npz")
    if not os.path.exists(path):
        pytest.skip("no real-gpytorch golden (run tests/golden/make_goldens.py --real-gpytorch where gpytorch is installed)")
    from tests.test_oracle_golden import replay_joint_car_split
    d = np.load(path)
    p = _car(int(d["Ns"]), int(d["H"]), int(d["iters"]))
    p["common"]["num_MPC_itrs"] = 2
    agent, _ = make_agents(sg, p, erv=d["epistimic_random_vector"])
    raw = sg._lib.load()

    def check(step, k, post):
        assert raw.gpmpc_joint_last_path() == sg._lib.JOINT_MFMA
        assert (post.last_info & sg._lib.INFO_ROOT_EIGH).all()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        replay_joint_car_split(agent, d, check=check)
