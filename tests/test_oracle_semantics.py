"""The gpytorch behaviours the oracle cannot verify here are SWITCHABLE (hard part 1): each switch
of ``oracle.gp_oracle.OracleSemantics`` changes exactly what its docstring says, and the defaults are what the HIP kernels
implement.  ``tests/golden/make_goldens.py --real-gpytorch`` reports which settings the genuine library matches."""
import numpy as np
import torch

from oracle.gp_oracle import (F64, FactorInfo, GPHyper, OracleGP, OracleSemantics, psd_safe_cholesky,
                              root_decomposition)


def _batch_pd_and_singular():
    g = torch.Generator().manual_seed(0)
    B = torch.randn(5, 5, generator=g, dtype=F64)
    pd = B @ B.T + 0.5 * torch.eye(5, dtype=F64)
    v = torch.randn(5, 2, generator=g, dtype=F64)
    sing = v @ v.T                                               # rank 2: the plain Cholesky fails
    return torch.stack([pd, sing])


def test_jitter_policy_switch():
    A = _batch_pd_and_singular()
    info = FactorInfo()
    L = psd_safe_cholesky(A, 1e-6, info, OracleSemantics(jitter_policy="failed_elements"))
    assert info.jitter_added.tolist()[0] == 0.0 and info.jitter_added.tolist()[1] > 0.0
    np.testing.assert_allclose((L[0] @ L[0].T).numpy(), A[0].numpy(), rtol=1e-13)
    info2 = FactorInfo()
    L2 = psd_safe_cholesky(A, 1e-6, info2, OracleSemantics(jitter_policy="whole_batch"))
    assert info2.jitter_added[0] == info2.jitter_added[1] > 0
    assert float((L2[0] @ L2[0].T - A[0]).diagonal().min()) > 0       # the p.d. element got the jitter too
    info3 = FactorInfo()
    psd_safe_cholesky(A[:1], 1e-6, info3, OracleSemantics(jitter_policy="always"))
    assert info3.tries == 0 and float(info3.jitter_added[0]) == 1e-6


def test_eigh_fallback_switch():
    A = _batch_pd_and_singular()
    A[1] -= 1e-3 * torch.eye(5, dtype=F64)                         # indefinite: no retry at jitter 1e-12 can succeed
    info = FactorInfo()
    R = root_decomposition(A, 1e-12, info, OracleSemantics(eigh_fallback="whole_batch"))
    assert info.used_eigh
    assert float(torch.triu(R[0], 1).abs().max()) > 1e-3            # element 0 carries an eigh root (not triangular)
    Rp = root_decomposition(A, 1e-12, FactorInfo(), OracleSemantics(eigh_fallback="failed_elements"))
    assert float(torch.triu(Rp[0], 1).abs().max()) == 0.0           # element 0 keeps its Cholesky root
    np.testing.assert_allclose(Rp[1].numpy(), R[1].numpy())
    for r in (R, Rp):
        np.testing.assert_allclose((r[0] @ r[0].T).numpy(), A[0].numpy(), rtol=1e-10, atol=1e-12)


def _small_gp(sem, nan_in_first_sample=True):
    g = torch.Generator().manual_seed(1)
    Ns, n = 2, 4
    X = torch.rand(Ns, 1, n, 2, generator=g, dtype=F64)
    Y = torch.randn(Ns, 1, n, 3, generator=g, dtype=F64)
    if nan_in_first_sample:
        Y[0, 0, 1, 2] = float("nan")
    h = GPHyper(torch.tensor([[0.7, 0.9]], dtype=F64), torch.tensor([0.8], dtype=F64),
                torch.tensor([1e-4, 1e-4, 1e-4], dtype=F64), 1e-8, True, sem)
    xs = torch.rand(Ns, 1, 3, 2, generator=g, dtype=F64)
    return OracleGP(X, Y, h), xs, X, Y, h


def test_nan_mask_batch_collapse_switch():
    gp, xs, X, Y, h = _small_gp(OracleSemantics(nan_mask_batch_collapse=True))
    post = gp(xs)
    gp2, _, _, _, _ = _small_gp(OracleSemantics(nan_mask_batch_collapse=False))
    post2 = gp2(xs)
    # sample 1 has no NaN: without the collapse it conditions on ALL its labels = the model built on sample 1 alone
    alone = OracleGP(X[1:], Y[1:], h)(xs[1:])
    np.testing.assert_allclose(post2.mean[1].numpy(), alone.mean[0].numpy(), rtol=1e-12)
    assert float((post.mean[1] - alone.mean[0]).abs().max()) > 1e-6   # with the collapse it lost the slot sample 0 masked
    # sample 0 masks its own slot either way
    np.testing.assert_allclose(post2.mean[0].numpy(), post.mean[0].numpy(), rtol=1e-10)


def test_variance_floor_switch():
    gp, xs, X, Y, h = _small_gp(OracleSemantics(variance_floor=None), nan_in_first_sample=False)
    post = gp(X[:, :, :2])                                            # at training inputs the variance is ~noise (1e-4)
    gpf, _, _, _, _ = _small_gp(OracleSemantics(variance_floor=1e-2), nan_in_first_sample=False)
    postf = gpf(X[:, :, :2])
    assert float(post.variance.min()) < 1e-2 and float(postf.variance.min()) == 1e-2
