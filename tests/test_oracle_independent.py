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


def test_value_only_posterior_against_scikit_learn():
    """The value-only (T = 1) GP algebra of the oracle - ARD RBF convention exp(-r^2 / (2 l^2)), outputscale as a
    multiplier, noise on the diagonal, zero mean, posterior mean and covariance, Cholesky-root sampling - against
    scikit-learn's GaussianProcessRegressor (an independent third-party exact-GP implementation; hyper-parameters
    fixed, optimizer off) on the car's 5 x 9 training grid with the shipped hyper-parameters of every output."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel
    from tests.helpers import load_params
    from oracle import agent_oracle as ao
    p = load_params("params_car_residual_fs")
    p["common"]["use_cuda"] = False
    env = ao.make_oracle_env(p)
    X, Y = env.initial_training_data()                       # (45, 2), (3, 45, 3); value-only: column 0
    hy = go.GPHyper.from_params(p, use_grad=False)
    rs = np.random.RandomState(3)
    Xs = np.stack([rs.uniform(-1.2, 1.2, 7), rs.uniform(-0.7, 0.7, 7)], axis=1)
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
