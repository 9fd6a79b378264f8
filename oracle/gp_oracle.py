"""CPU oracle for the GP algebra of the sampling-gpmpc rollout hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module;
the product package (``sampling_gpmpc_amd``) never does and fails loudly when its HIP library is missing.

What it restates
----------------
The arithmetic the reference obtains from its third-party dependency ``gpytorch==1.13``
(reference ``requirements.txt:6``; pulls ``linear_operator`` >= 0.5.3), which is NOT under ``/root/reference``
and not installed in this image.  The restatement follows the published gpytorch 1.13 algorithm at the
reference's call sites:

* model definition ........ reference ``src/GP_model.py:50-91`` (ConstantMean[Grad] == 0, Scale(RBFKernel[Grad]))
* hyper-parameter injection reference ``src/GP_model.py:121-143``
* likelihood noise ........ reference ``src/agent.py:235-240`` (MultitaskGaussianLikelihood, rank 0, global noise)
* settings ................ reference ``src/agent.py:630-638`` (nan policy "mask", fast_computations all off,
                            cholesky_jitter = Dyn_gp_jitter)
* posterior call .......... reference ``src/agent.py:640`` (``model_i(x)``: ExactGP eval-mode prediction)
* sampling ................ reference ``src/agent.py:641`` (``.sample(base_samples)``), ``benchmarking/
                            simulate_true_reachable_set.py:208-211`` (``.sample()``)
* variance ................ reference ``src/agent.py:648,703,706`` (``.variance`` with the 1e-10 FP64 floor)

Every tensor carries the reference's leading batch shape ``(Ns, g_ny)``, the real data are tiled ``Ns`` times,
the kernel matrix is rebuilt densely and factorised from scratch on every call - deliberately the reference's
op sequence, because this file is also the "reference on host cores" CPU baseline of ``bench.py``.

PARITY UNPINNED: the reference holds no assertion-bearing test, golden vector or fixture for this arithmetic
(``test/partial_gp_updates.py`` only plots) and gpytorch cannot be run here, so the formulas below are pinned
only by (a) independent high-precision (mpmath) re-evaluation, (b) analytic identities, and (c) the in-reference
numpy sampler ``extra/conditioning_gp.py`` for the value-only case - see ``tests/test_oracle_*.py``.  Everything
around the GP algebra (data generation, batching, post-processing, Jacobian assembly, rollout loop) IS pinned
against the reference's own code, see ``tests/golden/make_goldens.py``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import torch

F64 = torch.float64
MIN_VARIANCE_F64 = 1e-10        # gpytorch.settings.min_variance, double value
CHOLESKY_MAX_TRIES = 3          # gpytorch.settings.cholesky_max_tries default


class NotPSDError(RuntimeError):
    """All jitter retries failed (linear_operator.utils.errors.NotPSDError is a RuntimeError too)."""


@dataclass
class OracleSemantics:
    """The gpytorch / linear_operator behaviours this restatement cannot verify here (hard part 1),
    each one switchable.  The DEFAULTS are the reading of gpytorch 1.13 / linear_operator 0.5.x the HIP kernels
    implement; ``tests/golden/make_goldens.py --real-gpytorch`` reports which setting the genuine library matches when
    it is run on a machine that has it.

    jitter_policy ............ "failed_elements": retry i adds ``jitter * 10**i`` (total) to the diagonal of the batch
                               elements whose previous attempt failed (psd_safe_cholesky since linear_operator 0.1);
                               "whole_batch": ... of EVERY batch element as soon as one fails (gpytorch <= 1.5);
                               "always": the first attempt already carries ``jitter`` (no un-jittered try).
    eigh_fallback ............ "whole_batch": when any element is still not p.d. after the retries, the root of every
                               element is the eigendecomposition root (NotPSDError is raised for the whole operator);
                               "failed_elements": only the failing elements take it, the others keep their Cholesky root.
    variance_floor ........... lower clamp of ``.variance`` (gpytorch.settings.min_variance, 1e-10 for float64); None = off.
    nan_mask_batch_collapse .. True: a label slot that is NaN in ANY batch element is dropped for the WHOLE batch (the
                               mask is reduced over the batch dimensions); False: every batch element drops its own slots.
    """
    jitter_policy: str = "failed_elements"
    eigh_fallback: str = "whole_batch"
    variance_floor: Optional[float] = MIN_VARIANCE_F64
    nan_mask_batch_collapse: bool = True

    def __post_init__(self):
        assert self.jitter_policy in ("failed_elements", "whole_batch", "always")
        assert self.eigh_fallback in ("whole_batch", "failed_elements")


DEFAULT_SEMANTICS = OracleSemantics()


# ----------------------------------------------------------------------------------------------------------------
# A.2  kernel:  sigma^2 * RBF (value-only)  or  sigma^2 * RBFKernelGrad (value + gradient), interleaved ordering
# ----------------------------------------------------------------------------------------------------------------
def scaled_rbf_kernel(x1: torch.Tensor, x2: torch.Tensor, ell: torch.Tensor, outputscale: torch.Tensor,
                      use_grad: bool) -> torch.Tensor:
    """Dense cross-covariance between the label slots of ``x1`` and ``x2``.

    x1 ``(..., n1, D)``, x2 ``(..., n2, D)``; ``ell`` broadcastable to ``(..., 1, 1, D)``; ``outputscale``
    broadcastable to ``(..., 1, 1)``.  Returns ``(..., n1*T, n2*T)`` with T = 1 (value-only) or 1+D, slot order
    point-major / task-minor (gpytorch's RBFKernelGrad applies a perfect shuffle to reach that order; reference
    label tensors ``(n, T)`` flatten row-major into it).

    With r = x - x' and k = exp(-1/2 sum_d r_d^2 / l_d^2) the T x T block is
        cov(f,      f')       =  k
        cov(f,      d_j f')   = +k r_j / l_j^2
        cov(d_i f,  f')       = -k r_i / l_i^2
        cov(d_i f,  d_j f')   =  k (delta_ij / l_i^2 - r_i r_j / (l_i^2 l_j^2))
    """
    D = x1.shape[-1]
    r = x1.unsqueeze(-2) - x2.unsqueeze(-3)                       # (..., n1, n2, D)
    ell = ell.reshape(ell.shape[:-1] + (1, 1, D)) if ell.dim() >= 1 else ell
    inv_l2 = 1.0 / (ell * ell)                                     # (..., 1, 1, D)
    k = torch.exp(-0.5 * (r * r * inv_l2).sum(-1))                 # (..., n1, n2)
    os_ = outputscale.reshape(outputscale.shape + (1, 1)) if outputscale.dim() >= 1 else outputscale
    if not use_grad:
        return os_ * k
    T = 1 + D
    n1, n2 = x1.shape[-2], x2.shape[-2]
    q = r * inv_l2                                                 # (..., n1, n2, D) = r_d / l_d^2
    blk = torch.empty(k.shape + (T, T), dtype=k.dtype)
    blk[..., 0, 0] = k
    blk[..., 0, 1:] = k.unsqueeze(-1) * q
    blk[..., 1:, 0] = -k.unsqueeze(-1) * q
    eye = torch.diag_embed(inv_l2.expand(r.shape[:-3] + (1, 1, D)))   # (..., 1, 1, D, D) = delta_ij / l_i^2
    blk[..., 1:, 1:] = k.unsqueeze(-1).unsqueeze(-1) * (eye - q.unsqueeze(-1) * q.unsqueeze(-2))
    K = blk.permute(*range(k.dim() - 2), -4, -2, -3, -1).reshape(k.shape[:-2] + (n1 * T, n2 * T))
    return os_ * K


# ----------------------------------------------------------------------------------------------------------------
# A.7  safe Cholesky with jitter-on-failure, root decomposition with eigh fallback
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class FactorInfo:
    """Diagnostics of one (batched) safe Cholesky / root decomposition."""
    jitter_added: Optional[torch.Tensor] = None   # per batch element: total jitter on the diagonal (0 if none)
    first_info: Optional[torch.Tensor] = None     # per batch element: cholesky_ex info of the un-jittered attempt
    used_eigh: bool = False
    tries: int = 0


def psd_safe_cholesky(A: torch.Tensor, jitter: float, info_out: Optional[FactorInfo] = None,
                      sem: OracleSemantics = DEFAULT_SEMANTICS) -> torch.Tensor:
    """linear_operator.utils.cholesky.psd_safe_cholesky restated (A.7 steps 2-3).

    Plain ``cholesky_ex``; if any batch element fails, up to three retries adding ``jitter * 10**i`` (total) to the
    diagonal of the batch elements that failed the *previous* attempt (``sem.jitter_policy``), re-factorising the whole
    batch each time.  On NotPSDError the exception carries the last factor and info (``.L``, ``.info``) for the
    per-element eigh fallback.
    """
    A0 = A
    if sem.jitter_policy == "always":
        A0 = A.clone()
        A0.diagonal(dim1=-1, dim2=-2).add_(jitter)
    L, info = torch.linalg.cholesky_ex(A0)
    if info_out is not None:
        info_out.first_info = info.clone()
        info_out.jitter_added = torch.full(A.shape[:-2], jitter if sem.jitter_policy == "always" else 0.0, dtype=A.dtype)
        info_out.tries = 0
    if not torch.any(info):
        return L
    if torch.isnan(A).any():
        raise ValueError("cholesky_cpu: NaN in input")          # NanError in the library
    Aprime = A0.clone()
    jitter_prev = jitter if sem.jitter_policy == "always" else 0.0
    for i in range(CHOLESKY_MAX_TRIES):
        jitter_new = jitter * (10 ** (i + 1 if sem.jitter_policy == "always" else i))
        who = (info > 0) if sem.jitter_policy != "whole_batch" else torch.ones_like(info, dtype=torch.bool)
        add = who.to(A.dtype) * (jitter_new - jitter_prev)
        Aprime.diagonal(dim1=-1, dim2=-2).add_(add.unsqueeze(-1))
        if info_out is not None:
            info_out.jitter_added = info_out.jitter_added + add
            info_out.tries = i + 1
        jitter_prev = jitter_new
        L, info = torch.linalg.cholesky_ex(Aprime)
        if not torch.any(info):
            return L
    err = NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {jitter_new:.1e}.")
    err.L, err.info = L, info
    raise err


def root_decomposition(S: torch.Tensor, jitter: float, info_out: Optional[FactorInfo] = None,
                       sem: OracleSemantics = DEFAULT_SEMANTICS) -> torch.Tensor:
    """LinearOperator.root_decomposition(method="cholesky") restated (A.7 steps 1-4)."""
    if S.shape[-1] == 1:
        return S.sqrt()                                          # 1x1: plain sqrt, no jitter, NaN if negative
    try:
        return psd_safe_cholesky(S, jitter, info_out, sem)
    except NotPSDError as e:
        evals, evecs = torch.linalg.eigh(S)
        if info_out is not None:
            info_out.used_eigh = True
        R = evecs * evals.clamp_min(0.0).sqrt().unsqueeze(-2)
        if sem.eigh_fallback == "whole_batch":
            return R                                             # whole batch falls back
        failed = (e.info > 0).reshape(e.info.shape + (1, 1))
        return torch.where(failed, R, e.L)


# ----------------------------------------------------------------------------------------------------------------
# A.1, A.3-A.6, A.8  the batched exact GP ("model_i") and its posterior ("model_i_call")
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class GPHyper:
    """Hyper-parameters as injected by reference ``src/GP_model.py:121-143``; they depend on the output only."""
    ell: torch.Tensor           # (g_ny, D)   Dyn_gp_lengthscale.both[o]
    outputscale: torch.Tensor   # (g_ny,)     Dyn_gp_outputscale.both[o]
    noise_diag: torch.Tensor    # (T,)        task_noises.val[t]*multiplier + Dyn_gp_noise
    jitter: float               # Dyn_gp_jitter
    use_grad: bool = True
    semantics: OracleSemantics = field(default_factory=OracleSemantics)

    @property
    def T(self) -> int:
        return int(self.noise_diag.numel())

    @staticmethod
    def from_params(params: dict, use_grad: bool) -> "GPHyper":
        ag = params["agent"]
        ell = torch.tensor(ag["Dyn_gp_lengthscale"]["both"], dtype=F64)
        g_ny = ag["g_dim"]["ny"]
        D = ag["g_dim"]["nx"] + ag["g_dim"]["nu"]
        ell = ell.reshape(g_ny, D)
        osc = torch.tensor(ag["Dyn_gp_outputscale"]["both"], dtype=F64).reshape(g_ny)
        tn = ag["Dyn_gp_task_noises"]["val"] if use_grad else [ag["Dyn_gp_task_noises"]["val"][0]]
        noise = torch.tensor(tn, dtype=F64) * ag["Dyn_gp_task_noises"]["multiplier"] + ag["Dyn_gp_noise"]
        return GPHyper(ell, osc, noise, float(ag["Dyn_gp_jitter"]), use_grad)


class OraclePosterior:
    """Stand-in for the MultitaskMultivariateNormal returned by ``model_i(x)``."""

    def __init__(self, mean: torch.Tensor, covar: torch.Tensor, jitter: float,
                 sem: OracleSemantics = DEFAULT_SEMANTICS):
        self.mean = mean                    # (Ns, g_ny, m, T)
        self.covariance_matrix = covar      # (Ns, g_ny, m*T, m*T)
        self._jitter = jitter
        self._sem = sem
        self.root_info = FactorInfo()

    @property
    def variance(self) -> torch.Tensor:     # A.8: diag, no jitter, floor 1e-10
        v = self.covariance_matrix.diagonal(dim1=-1, dim2=-2).reshape(self.mean.shape)
        return v if self._sem.variance_floor is None else v.clamp_min(self._sem.variance_floor)

    @property
    def stddev(self) -> torch.Tensor:
        return self.variance.sqrt()

    def confidence_region(self):
        s2 = self.stddev * 2
        return self.mean - s2, self.mean + s2

    def root(self) -> torch.Tensor:
        return root_decomposition(self.covariance_matrix, self._jitter, self.root_info, self._sem)

    def sample(self, base_samples: Optional[torch.Tensor] = None) -> torch.Tensor:
        """A.7: y = mu + R z with z the given base samples flattened interleaved, or internal randn."""
        R = self.root()
        bshape = self.mean.shape[:-2]
        if base_samples is None:
            z = torch.randn(*bshape, R.shape[-1], 1, dtype=self.mean.dtype)
        else:
            if base_samples.shape != self.mean.shape:
                raise RuntimeError("base_samples shape must match the mean shape")
            z = base_samples.reshape(*bshape, -1, 1)
        y = (R @ z).squeeze(-1) + self.mean.reshape(*bshape, -1)
        return y.reshape(self.mean.shape)


class OracleGP:
    """Stand-in for ``BatchMultitaskGPModelWithDerivatives_fromParams`` in eval mode (from-scratch algebra)."""

    def __init__(self, train_x: torch.Tensor, train_y: torch.Tensor, hyper: GPHyper):
        assert train_x.dim() == 4 and train_y.dim() == 4
        self.train_inputs = (train_x,)
        self.train_targets = train_y
        self.batch_shape = torch.Size(train_x.shape[:2])
        self.hyper = hyper
        self._cache = None
        self.train_info = FactorInfo()

    def eval(self):
        return self

    # A.4: a label slot that is NaN in ANY batch element is dropped for the whole batch
    def observed_mask(self) -> torch.Tensor:
        y = self.train_targets
        return ~torch.any(torch.isnan(y.reshape(-1, y.shape[-2] * y.shape[-1])), dim=0)

    def _per_element(self, x: torch.Tensor) -> "OraclePosterior":
        """``nan_mask_batch_collapse = False``: every batch element conditions on its own observed slots (a loop of
        single-element models; only for the small comparison runs of the semantics switches)."""
        import copy
        sem = copy.copy(self.hyper.semantics)
        sem.nan_mask_batch_collapse = True
        Ns, g_ny = self.batch_shape
        means, covs = [], []
        for s_ in range(Ns):
            mo, co = [], []
            for o in range(g_ny):
                h = GPHyper(self.hyper.ell[o:o + 1], self.hyper.outputscale[o:o + 1], self.hyper.noise_diag,
                            self.hyper.jitter, self.hyper.use_grad, sem)
                one = OracleGP(self.train_inputs[0][s_:s_ + 1, o:o + 1], self.train_targets[s_:s_ + 1, o:o + 1], h)
                post = one(x[s_:s_ + 1, o:o + 1])
                mo.append(post.mean)
                co.append(post.covariance_matrix)
            means.append(torch.cat(mo, dim=1))
            covs.append(torch.cat(co, dim=1))
        return OraclePosterior(torch.cat(means, dim=0), torch.cat(covs, dim=0), self.hyper.jitter, self.hyper.semantics)

    def _ell_os(self):
        h = self.hyper
        ell = h.ell.reshape(1, *h.ell.shape)                      # (1, g_ny, D) -> broadcast over Ns
        osc = h.outputscale.reshape(1, -1)
        return ell, osc

    def _train_cache(self):
        if self._cache is None:
            h = self.hyper
            X, Y = self.train_inputs[0], self.train_targets
            N, T = Y.shape[-2], Y.shape[-1]
            obs = self.observed_mask()
            ell, osc = self._ell_os()
            K = scaled_rbf_kernel(X, X, ell, osc, h.use_grad)     # (Ns, g_ny, N*T, N*T), dense, every call
            K = K + torch.diag(h.noise_diag.repeat(N))            # kron(I_N, diag(noise))  (A.3)
            Koo = K[..., obs, :][..., :, obs]
            L = psd_safe_cholesky(Koo, h.jitter, self.train_info, h.semantics)  # A.5
            yo = Y.reshape(*Y.shape[:-2], N * T)[..., obs].unsqueeze(-1)
            alpha = torch.cholesky_solve(yo, L)                   # mean is zero
            self._cache = (obs, L, alpha)
        return self._cache

    def __call__(self, x: torch.Tensor) -> OraclePosterior:
        h = self.hyper
        if not h.semantics.nan_mask_batch_collapse and bool(torch.isnan(self.train_targets).any()):
            return self._per_element(x)
        obs, L, alpha = self._train_cache()
        X = self.train_inputs[0]
        T = self.train_targets.shape[-1]
        m = x.shape[-2]
        ell, osc = self._ell_os()
        K_so = scaled_rbf_kernel(x, X, ell, osc, h.use_grad)[..., :, obs]   # (Ns, g_ny, m*T, n_o)
        K_ss = scaled_rbf_kernel(x, x, ell, osc, h.use_grad)
        mean = (K_so @ alpha).squeeze(-1).reshape(*x.shape[:-2], m, T)
        corr = torch.cholesky_solve(K_so.transpose(-1, -2), L)               # (K_oo+S)^-1 K_o*
        covar = K_ss + K_so @ corr.mul(-1)                                   # A.6
        return OraclePosterior(mean, covar, h.jitter, h.semantics)
