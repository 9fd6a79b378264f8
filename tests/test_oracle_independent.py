"""Independent checks of the part of the oracle that no reference fixture can pin (the GP algebra is
gpytorch's, which is neither under /root/reference nor installed).

* the RBF(+gradient) blocks are the partial derivatives of the value kernel   (mpmath automatic differentiation)
* posterior mean / covariance / sample of App. A re-evaluated in 40-digit arithmetic from those derivatives
* analytic identities: posterior at the training inputs (mean = y - S alpha, cov = S - S (K+S)^-1 S); sequential conditioning on exact draws equals the
  joint draw with the same base samples (mode R == mode J in the noise-free limit; also pins the slot interleaving)

These are CPU tests of test infrastructure (``oracle/``); the HIP path is compared with the oracle in
``test_hip_parity.py``.
"""
import os
import sys

import mpmath as mp
import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gp_oracle as go  # noqa: E402

F64 = torch.float64
mp.mp.dps = 40


def _k_mp(ell, os_):
    """value kernel k(x, x') as an mpmath function of the 2*D scalars (x_0..x_{D-1}, x'_0..x'_{D-1})"""
    D = len(ell)

    def k(*a):
        s = mp.mpf(0)
        for d in range(D):
            r = a[d] - a[D + d]
            s += r * r / (mp.mpf(ell[d]) ** 2)
        return mp.mpf(os_) * mp.exp(-s / 2)
    return k


def _block_mp(x, xp, ell, os_):
    """(T x T) covariance block between the label slots (f, d_0 f, ..) at x and at xp by differentiating k"""
    D = len(ell)
    k = _k_mp(ell, os_)
    args = [mp.mpf(float(v)) for v in x] + [mp.mpf(float(v)) for v in xp]
    T = 1 + D
    B = mp.zeros(T, T)
    for a in range(T):
        for b in range(T):
            order = [0] * (2 * D)
            if a > 0:
                order[a - 1] += 1            # d / d x_{a-1}
            if b > 0:
                order[D + b - 1] += 1        # d / d x'_{b-1}
            B[a, b] = k(*args) if not any(order) else mp.diff(k, tuple(args), tuple(order))
    return B


def _gram_mp(X1, X2, ell, os_):
    T = 1 + len(ell)
    G = mp.zeros(len(X1) * T, len(X2) * T)
    for i, x in enumerate(X1):
        for j, xp in enumerate(X2):
            B = _block_mp(x, xp, ell, os_)
            for a in range(T):
                for b in range(T):
                    G[i * T + a, j * T + b] = B[a, b]        # point-major, task-minor slots
    return G


def _to_np(M):
    return np.array([[float(M[i, j]) for j in range(M.cols)] for i in range(M.rows)])


def test_kernel_blocks_are_derivatives_of_the_value_kernel():
    rng = np.random.default_rng(5)
    ell, os_ = [0.7, 1.9], 1.3
    X1, X2 = rng.normal(size=(3, 2)), rng.normal(size=(4, 2))
    K = go.scaled_rbf_kernel(torch.tensor(X1), torch.tensor(X2), torch.tensor(ell, dtype=F64),
                             torch.tensor(os_, dtype=F64), True).numpy()
    G = _to_np(_gram_mp(X1, X2, ell, os_))
    assert K.shape == G.shape == (9, 12)
    np.testing.assert_allclose(K, G, rtol=1e-12, atol=1e-14)
    # value-only kernel = the (0, 0) entries
    K0 = go.scaled_rbf_kernel(torch.tensor(X1), torch.tensor(X2), torch.tensor(ell, dtype=F64),
                              torch.tensor(os_, dtype=F64), False).numpy()
    np.testing.assert_allclose(K0, G[0::3, 0::3], rtol=1e-13)


def _small_problem(seed=2):
    rng = np.random.default_rng(seed)
    ell, os_ = [0.9, 2.2], 0.8
    noise = [3e-4, 2e-3, 5e-4]
    Xr = rng.uniform(-1, 1, size=(5, 2))
    Yr = np.full((5, 3), np.nan)
    Yr[:, 0] = np.sin(Xr[:, 0]) + 0.3 * Xr[:, 1]                       # real data: value-only labels
    Xh = rng.uniform(-1, 1, size=(2, 2))
    Yh = rng.normal(size=(2, 3)) * 0.2                                  # appended points: value + gradient labels
    Xs = rng.uniform(-1, 1, size=(2, 2))
    return ell, os_, noise, Xr, Yr, Xh, Yh, Xs


def test_posterior_and_sample_against_40_digit_arithmetic():
    ell, os_, noise, Xr, Yr, Xh, Yh, Xs = _small_problem()
    X = np.concatenate([Xr, Xh]); Y = np.concatenate([Yr, Yh])
    hyper = go.GPHyper(torch.tensor([ell], dtype=F64), torch.tensor([os_], dtype=F64), torch.tensor(noise, dtype=F64),
                       1e-8, True)
    gp = go.OracleGP(torch.tensor(X).reshape(1, 1, -1, 2), torch.tensor(Y).reshape(1, 1, -1, 3), hyper)
    post = gp(torch.tensor(Xs).reshape(1, 1, -1, 2))
    z = np.random.default_rng(0).normal(size=(2, 3))
    y = post.sample(torch.tensor(z).reshape(1, 1, 2, 3))
    assert float(post.root_info.jitter_added.abs().max()) == 0.0 and not post.root_info.used_eigh

    # the same in 40 digits: Gram from derivatives, noise on every slot, NaN slots dropped, zero prior mean
    T, N = 3, len(X)
    obs = [i for i in range(N * T) if not np.isnan(Y.reshape(-1)[i])]
    Kxx = _gram_mp(X, X, ell, os_)
    for i in range(N * T):
        Kxx[i, i] += mp.mpf(noise[i % T])
    Koo = mp.matrix(len(obs), len(obs))
    for a, i in enumerate(obs):
        for b, j in enumerate(obs):
            Koo[a, b] = Kxx[i, j]
    yo = mp.matrix([mp.mpf(float(Y.reshape(-1)[i])) for i in obs])
    Kso_full = _gram_mp(Xs, X, ell, os_)
    Kso = mp.matrix(Kso_full.rows, len(obs))
    for a in range(Kso_full.rows):
        for b, j in enumerate(obs):
            Kso[a, b] = Kso_full[a, j]
    Kss = _gram_mp(Xs, Xs, ell, os_)
    Kinv = mp.inverse(Koo)                                              # 40 digits: conditioning is irrelevant
    mean = Kso * (Kinv * yo)
    cov = Kss - Kso * Kinv * Kso.T
    R = mp.cholesky(cov)                                                # lower
    ys = mean + R * mp.matrix([mp.mpf(float(v)) for v in z.reshape(-1)])

    vec = lambda M: [float(M[i, 0]) for i in range(M.rows)]
    np.testing.assert_allclose(post.mean.reshape(-1).numpy(), vec(mean), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(post.covariance_matrix[0, 0].numpy(), _to_np(cov), rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(y.reshape(-1).numpy(), vec(ys), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(post.variance.reshape(-1).numpy(), np.diag(_to_np(cov)), rtol=1e-7, atol=1e-11)


def test_posterior_at_the_training_inputs_identities():
    """Exact identities at the training inputs (any noise level), observed slots o, S = diag(noise):
    mean_o = y_o - S alpha   and   cov_oo = S - S (K_oo + S)^-1 S.   Unobserved (NaN) slots are not pinned by them."""
    ell, os_, noise, Xr, Yr, Xh, Yh, _ = _small_problem(seed=4)
    hyper = go.GPHyper(torch.tensor([ell], dtype=F64), torch.tensor([os_], dtype=F64), torch.tensor(noise, dtype=F64),
                       1e-8, True)
    X = np.concatenate([Xr, Xh]); Y = np.concatenate([Yr, Yh])
    gp = go.OracleGP(torch.tensor(X).reshape(1, 1, -1, 2), torch.tensor(Y).reshape(1, 1, -1, 3), hyper)
    post = gp(torch.tensor(X).reshape(1, 1, -1, 2))
    obs, L, alpha = gp._train_cache()
    S = torch.tensor(noise, dtype=F64).repeat(len(X))[obs]
    y_o = torch.tensor(Y).reshape(-1)[obs]
    np.testing.assert_allclose(post.mean.reshape(-1)[obs].numpy(), (y_o - S * alpha[0, 0, :, 0]).numpy(), rtol=1e-9, atol=1e-11)
    Kinv = torch.cholesky_inverse(L[0, 0])
    cov_oo = post.covariance_matrix[0, 0][obs][:, obs]
    np.testing.assert_allclose(cov_oo.numpy(), (torch.diag(S) - S[:, None] * Kinv * S[None, :]).numpy(), rtol=1e-6, atol=1e-11)
    # value-only real points: the value is pinned to within its noise, the (unobserved) gradient is not
    var = post.variance[0, 0]
    assert float(var[:5, 0].max()) <= noise[0] and float(var[:5, 1:].min()) > float(var[:5, 0].max())


def test_sequential_conditioning_on_exact_draws_equals_the_joint_draw():
    """y1 ~ p(.|D), y2 ~ p(.|D, y1) with base samples z1, z2  ==  [y1, y2] = mu + chol(Sigma) [z1, z2]  when the
    appended labels carry (almost) no noise: the block-Cholesky identity behind the reference's two modes."""
    ell, os_, _, Xr, Yr, _, _, Xs = _small_problem(seed=7)
    tiny = 1e-13
    hyper = go.GPHyper(torch.tensor([ell], dtype=F64), torch.tensor([os_], dtype=F64),
                       torch.tensor([2e-3, tiny, tiny], dtype=F64), 1e-8, True)
    # real rows: value-only with noise 2e-3; appended rows: the value slot shares that noise in the reference's model,
    # so make the comparison on a model whose first task noise is tiny as well
    hyper_exact = go.GPHyper(hyper.ell, hyper.outputscale, torch.full((3,), tiny, dtype=F64), 1e-8, True)
    Xr_t, Yr_t = torch.tensor(Xr).reshape(1, 1, -1, 2), torch.tensor(Yr).reshape(1, 1, -1, 3)
    z = torch.tensor(np.random.default_rng(1).normal(size=(1, 1, 2, 3)))
    xs = torch.tensor(Xs).reshape(1, 1, 2, 2)
    joint = go.OracleGP(Xr_t, Yr_t, hyper_exact)(xs).sample(z)
    y1 = go.OracleGP(Xr_t, Yr_t, hyper_exact)(xs[:, :, :1]).sample(z[:, :, :1])
    gp2 = go.OracleGP(torch.cat([Xr_t, xs[:, :, :1]], 2), torch.cat([Yr_t, y1], 2), hyper_exact)
    y2 = gp2(xs[:, :, 1:]).sample(z[:, :, 1:])
    np.testing.assert_allclose(y1.numpy(), joint[:, :, :1].numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(y2.numpy(), joint[:, :, 1:].numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("pname,lo,hi", [("params_car_residual_fs", (-1.2, -0.7), (1.2, 0.7)),
                                         ("params_pendulum1D_samples", (2.0, -5.5), (3.7, 5.5))])
def test_value_only_posterior_against_scikit_learn(pname, lo, hi):
    """The value-only (T = 1) GP algebra of the oracle - ARD RBF convention exp(-r^2 / (2 l^2)), outputscale as a
    multiplier, noise on the diagonal, zero mean, posterior mean and covariance, Cholesky-root sampling - against
    scikit-learn's GaussianProcessRegressor (an independent third-party exact-GP implementation; hyper-parameters
    fixed, optimizer off) on the reference's training grids (car 5 x 9, pendulum 4 x 9) with the shipped
    hyper-parameters of every output."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel
    from tests.helpers import load_params
    from oracle import agent_oracle as ao
    p = load_params(pname)
    p["common"]["use_cuda"] = False
    env = ao.make_oracle_env(p)
    X, Y = env.initial_training_data()                       # (N_r, 2), (g_ny, N_r, 3); value-only: column 0
    hy = go.GPHyper.from_params(p, use_grad=False)
    rs = np.random.RandomState(3)
    Xs = np.stack([rs.uniform(lo[0], hi[0], 7), rs.uniform(lo[1], hi[1], 7)], axis=1)
    for o in range(hy.ell.shape[0]):
        hyp = go.GPHyper(hy.ell[[o]], hy.outputscale[[o]], hy.noise_diag, hy.jitter, False)
        gp = go.OracleGP(X.reshape(1, 1, -1, 2).to(F64), Y[o, :, [0]].reshape(1, 1, -1, 1).to(F64), hyp)
        post = gp(torch.tensor(Xs).reshape(1, 1, -1, 2))
        kern = ConstantKernel(float(hy.outputscale[o]), "fixed") * RBF(hy.ell[o].numpy(), "fixed")
        sk = GaussianProcessRegressor(kernel=kern, alpha=float(hy.noise_diag[0]), optimizer=None, normalize_y=False)
        sk.fit(X.numpy(), Y[o, :, 0].numpy())
        mu, cov = sk.predict(Xs, return_cov=True)
        np.testing.assert_allclose(post.mean.reshape(-1).numpy(), mu, rtol=1e-8, atol=1e-10 * np.abs(mu).max())
        np.testing.assert_allclose(post.covariance_matrix[0, 0].numpy(), cov, rtol=1e-6, atol=1e-9 * np.abs(cov).max())
        # Cholesky-root sampling with given base samples == mean + chol(cov) z on scikit-learn's covariance
        z = torch.tensor(rs.randn(7)).reshape(1, 1, 7, 1)
        Lc = np.linalg.cholesky(cov + 0.0 * np.eye(7))
        y_sk = mu + Lc @ z.reshape(-1).numpy()
        info = go.FactorInfo()
        R = go.root_decomposition(post.covariance_matrix, hyp.jitter, info)
        if float(info.jitter_added.max()) == 0.0:               # un-jittered branch: same root
            np.testing.assert_allclose(post.sample(z).reshape(-1).numpy(), y_sk, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("pname", ["params_pendulum1D_samples", "params_car_residual_fs"])
def test_jitter_retry_branch_against_scikit_learn(pname):
    """A.7 step 3 (the branch every pendulum mode-J draw ends in): a posterior covariance that is singular - two of the
    test points coincide - fails the plain Cholesky; the oracle's root is then chol(Sigma + j I) for the first j in
    jitter x {1, 10, 100} that succeeds.  Pinned against scikit-learn's covariance (third party): the retry level the
    oracle reports, the root it returns and the sample y = mu + R z."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel
    from tests.helpers import load_params
    from oracle import agent_oracle as ao
    p = load_params(pname)
    p["common"]["use_cuda"] = False
    env = ao.make_oracle_env(p)
    X, Y = env.initial_training_data()
    hy = go.GPHyper.from_params(p, use_grad=False)
    jitter = 1e-6                                            # the pendulum's shipped value; the car's 1e-20 never succeeds
    rs = np.random.RandomState(11)
    lo, hi = X.min(0).values.numpy(), X.max(0).values.numpy()
    Xs = rs.uniform(lo, hi, size=(6, 2))
    Xs[4] = Xs[1]                                            # coincident test points: Sigma is exactly singular
    Xs[5] = Xs[2]
    for o in range(hy.ell.shape[0]):
        hyp = go.GPHyper(hy.ell[[o]], hy.outputscale[[o]], hy.noise_diag, jitter, False)
        gp = go.OracleGP(X.reshape(1, 1, -1, 2).to(F64), Y[o, :, [0]].reshape(1, 1, -1, 1).to(F64), hyp)
        post = gp(torch.tensor(Xs).reshape(1, 1, -1, 2))
        kern = ConstantKernel(float(hy.outputscale[o]), "fixed") * RBF(hy.ell[o].numpy(), "fixed")
        sk = GaussianProcessRegressor(kernel=kern, alpha=float(hy.noise_diag[0]), optimizer=None, normalize_y=False)
        sk.fit(X.numpy(), Y[o, :, 0].numpy())
        mu, cov = sk.predict(Xs, return_cov=True)
        cov = 0.5 * (cov + cov.T)
        with pytest.raises(np.linalg.LinAlgError):
            np.linalg.cholesky(cov)                          # singular to round-off: the un-jittered attempt fails
        info = go.FactorInfo()
        R = go.root_decomposition(post.covariance_matrix, hyp.jitter, info)[0, 0].numpy()
        level = float(info.jitter_added.max())
        assert level in (jitter, 10 * jitter, 100 * jitter) and not info.used_eigh
        # the first level at which scikit-learn's covariance factorises is the level the oracle reports
        want = next(j for j in (jitter, 10 * jitter, 100 * jitter) if np.all(np.linalg.eigvalsh(cov + j * np.eye(6)) > 1e-3 * j))
        assert level == want
        Lsk = np.linalg.cholesky(cov + level * np.eye(6))
        np.testing.assert_allclose(R, Lsk, rtol=1e-5, atol=1e-7 * np.abs(Lsk).max())
        z = rs.randn(6)
        y = post.sample(torch.tensor(z).reshape(1, 1, 6, 1)).reshape(-1).numpy()
        np.testing.assert_allclose(y, mu + Lsk @ z, rtol=1e-6, atol=1e-7 * np.abs(mu).max() + 1e-9)


def _singular_joint_problem():
    """Two test points 2e-4 apart: with value AND gradient slots at both, f(x2) ~ f(x1) + dx . grad f(x1) makes the 6 x 6
    posterior covariance numerically singular (eigenvalues span 1e-1 .. 1e-17) - the structure of the reference's mode-J
    draws, where neighbouring stages of the horizon are almost the same GP input."""
    ell, os_ = [0.9, 2.2], 0.8
    noise = [1e-6, 1e-6, 1e-6]
    rng = np.random.default_rng(4)
    Xr = rng.uniform(-1, 1, size=(5, 2))
    Yr = np.full((5, 3), np.nan)
    Yr[:, 0] = np.sin(Xr[:, 0]) + 0.3 * Xr[:, 1]            # value-only real labels
    Xs = np.array([[0.31, -0.42], [0.3102, -0.4199]])
    return ell, os_, noise, Xr, Yr, Xs


def _posterior_mp(ell, os_, noise, Xr, Yr, Xs, dps=50):
    old = mp.mp.dps
    mp.mp.dps = dps
    try:
        T = 3
        obs = [i for i in range(len(Xr) * T) if not np.isnan(Yr.reshape(-1)[i])]
        Kfull = _gram_mp(Xr, Xr, ell, os_)
        Kxx = mp.matrix(len(obs), len(obs))
        for a, i in enumerate(obs):
            for b, j in enumerate(obs):
                Kxx[a, b] = Kfull[i, j] + (mp.mpf(noise[i % T]) if i == j else 0)
        Kfs = _gram_mp(Xs, Xr, ell, os_)
        Ksx = mp.matrix(Kfs.rows, len(obs))
        for a in range(Kfs.rows):
            for b, j in enumerate(obs):
                Ksx[a, b] = Kfs[a, j]
        Kss = _gram_mp(Xs, Xs, ell, os_)
        y = mp.matrix([mp.mpf(float(Yr.reshape(-1)[i])) for i in obs])
        sol = mp.lu_solve(Kxx, y)
        mu = Ksx * sol
        Sig = Kss - Ksx * mp.inverse(Kxx) * Ksx.T
        Sig = (Sig + Sig.T) / 2
        lam, U = mp.eigsy(Sig)
        n = Sig.rows
        R = mp.zeros(n, n)
        for j in range(n):
            sj = mp.sqrt(lam[j]) if lam[j] > 0 else mp.mpf(0)
            for i in range(n):
                R[i, j] = U[i, j] * sj
        Spos = R * R.T                                       # max(Sigma, 0) in 50 digits
        return _to_np(mu).reshape(-1), _to_np(Sig), _to_np(Spos), np.array([float(v) for v in lam])
    finally:
        mp.mp.dps = old


def _psd_part_mp(S, dps=50):
    """max(S, 0) = U max(Lambda, 0) U^T of a symmetric FP64 matrix, its eigenvalues, in `dps`-digit arithmetic"""
    old = mp.mp.dps
    mp.mp.dps = dps
    try:
        n = S.shape[0]
        M = mp.matrix(n, n)
        for i in range(n):
            for j in range(n):
                M[i, j] = mp.mpf(float(S[i, j]))
        lam, U = mp.eigsy(M)
        P = mp.zeros(n, n)
        for k in range(n):
            if lam[k] > 0:
                for i in range(n):
                    for j in range(n):
                        P[i, j] += lam[k] * U[i, k] * U[j, k]
        return _to_np(P), np.array([float(v) for v in lam])
    finally:
        mp.mp.dps = old


def test_eigendecomposition_root_against_50_digit_arithmetic():
    """A.7 step 4 (the branch every car mode-J draw ends in): R = U sqrt(max(lambda, 0)).
    (1) The posterior covariance of a numerically singular 6-slot problem in FP64 against 50-digit arithmetic: the two agree
        to the round-off of the cancellation K** - K*o (Koo + S)^-1 Ko*.
    (2) That FP64 matrix, shifted so that its smallest eigenvalues are negative (as round-off makes them in the reference's
        car runs), fails the plain Cholesky and the retries at jitter 1e-20 .. 1e-18; the oracle's eigh root R then satisfies
        R R^T == max(Sigma, 0) of the SAME matrix evaluated with a 50-digit symmetric eigensolver, the squared column norms
        are the clamped eigenvalues in ascending order, and y = mu + R z has the distribution's Mahalanobis footprint."""
    ell, os_, noise, Xr, Yr, Xs = _singular_joint_problem()
    mu_mp, Sig_mp, _, lam = _posterior_mp(ell, os_, noise, Xr, Yr, Xs)
    assert lam.max() > 1e-3 and lam.min() < 3e-16 * lam.max()          # numerically singular in FP64
    hyper = go.GPHyper(torch.tensor([ell], dtype=F64), torch.tensor([os_], dtype=F64), torch.tensor(noise, dtype=F64), 1e-20, True)
    gp = go.OracleGP(torch.tensor(Xr).reshape(1, 1, -1, 2), torch.tensor(Yr).reshape(1, 1, -1, 3), hyper)
    post = gp(torch.tensor(Xs).reshape(1, 1, -1, 2))
    Sig = post.covariance_matrix[0, 0].numpy()
    scale = np.abs(Sig_mp).max()
    round_off = np.abs(Sig - Sig_mp).max()
    print(f"FP64 posterior covariance vs 50 digits: max abs difference {round_off:.2e} (scale {scale:.2e}); 50-digit eigenvalues {lam}")
    assert round_off < 1e-9 * scale
    np.testing.assert_allclose(post.mean.reshape(-1).numpy(), mu_mp, rtol=1e-8, atol=1e-9)
    # (2) an indefinite matrix of the same structure
    shift = 4.0 * max(round_off, 1e-17 * scale)
    S2 = Sig - shift * np.eye(6)
    S2 = 0.5 * (S2 + S2.T)
    P_mp, lam2 = _psd_part_mp(S2)
    assert lam2.min() < -1e-18 * scale
    info = go.FactorInfo()
    R = go.root_decomposition(torch.tensor(S2).reshape(1, 1, 6, 6), hyper.jitter, info)[0, 0].numpy()
    assert info.used_eigh and info.tries == 3, "the jitter chain 1e-20 .. 1e-18 must fail on this matrix"
    tol = 64 * np.finfo(np.float64).eps * scale
    np.testing.assert_allclose(R @ R.T, P_mp, rtol=0, atol=tol)
    np.testing.assert_allclose(np.sort((R * R).sum(0)), np.sort(np.clip(lam2, 0, None)), rtol=0, atol=tol)
    assert np.all((R * R).sum(0)[: int((lam2 <= 0).sum())] <= tol)      # the clamped directions come first (ascending order)
    np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eigh_root_case.npy"),
            np.concatenate([S2.reshape(-1), P_mp.reshape(-1)])) if os.environ.get("GPMPC_WRITE_GOLDENS") else None
