"""``model_i``: the batched exact GP of the reference, backed by the HIP kernels.

Mirrors the surface reference callers use on ``BatchMultitaskGPModelWithDerivatives_fromParams``
(reference ``src/GP_model.py:94-143``) and on the object ``model_i(x)`` returns:

* ``model_i.train_inputs[0]`` / ``model_i.train_targets`` / ``model_i.batch_shape`` / ``model_i.eval()``
  (reference ``src/visu.py:483-484``, ``benchmarking/simulate_true_reachable_set.py:183``,
  ``benchmarking/robust_tube_based_GPMPC_koller.py:184-191``)
* ``model_i(x)`` -> ``.mean``, ``.variance``, ``.sample(base_samples)``, ``.confidence_region()``
  (reference ``src/agent.py:640-641,648,701-706``, ``src/solver.py:254-259``)

Differences by design: the real data are NOT tiled ``Ns`` times (they are shared by every sample; the tiled view
is materialised only if somebody reads ``train_inputs``), nothing is autograd-capable, and the arithmetic runs in
``libgpmpc_hip.so`` (``gpmpc_plan_build`` + ``gpmpc_joint_sample``).  No gpytorch, no CPU fallback.
"""
from __future__ import annotations

import itertools
import warnings
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib

F64 = torch.float64
import os

# GPMPC_VERIFY_FACTOR_CACHE=1: the joint factor cache compares the slot list and the points with its snapshots (two
# torch.equal + host syncs per draw) even when the Agent's lineage counter vouches for them (tests; debugging foreign code that
# edits Hallcinated_* in place without Agent.invalidate_factor_cache())
_VERIFY_CACHE = os.environ.get("GPMPC_VERIFY_FACTOR_CACHE") == "1"

JOINT_WS_HINT_MAX_BYTES = 32 << 30   # largest workspace allocated ahead of need (Agent's bound on the conditioning set)
MAX_JOINT_ROWS = 2048            # gpmpc_joint_sample: n_ho + 1 + m*T label rows per chain (include/gpmpc_hip.h)
MAX_JOINT_TEST_SLOTS = 256       # m*T


class NumericalWarning(RuntimeWarning):
    """Same role as gpytorch.utils.warnings.NumericalWarning (jitter added / variance clamped)."""


class NotPSDError(RuntimeError):
    """Factorisation of the training covariance failed (gpytorch raises NotPSDError after its jitter retries)."""


@dataclass
class GPHyperParams:
    """Hyper-parameters injected from the YAML exactly as reference ``src/GP_model.py:121-143`` does."""
    g_ny: int
    D: int
    T: int
    ell: list            # [g_ny][D]   agent.Dyn_gp_lengthscale.both
    outputscale: list    # [g_ny]      agent.Dyn_gp_outputscale.both
    noise: list          # [T]         task_noises.val * multiplier + Dyn_gp_noise
    jitter: float        # agent.Dyn_gp_jitter
    use_grad: bool

    @staticmethod
    def from_params(params: dict, use_grad: bool) -> "GPHyperParams":
        ag = params["agent"]
        g_ny = ag["g_dim"]["ny"]
        D = ag["g_dim"]["nx"] + ag["g_dim"]["nu"]
        ell = torch.tensor(ag["Dyn_gp_lengthscale"]["both"], dtype=F64).reshape(g_ny, D).tolist()
        osc = torch.tensor(ag["Dyn_gp_outputscale"]["both"], dtype=F64).reshape(g_ny).tolist()
        vals = ag["Dyn_gp_task_noises"]["val"] if use_grad else [ag["Dyn_gp_task_noises"]["val"][0]]
        noise = [v * ag["Dyn_gp_task_noises"]["multiplier"] + ag["Dyn_gp_noise"] for v in vals]
        T = 1 + D if use_grad else 1
        return GPHyperParams(g_ny, D, T, ell, osc, noise, float(ag["Dyn_gp_jitter"]), use_grad)


class RealDataPlan:
    """The shared real-data block, factorised once on the device (``gpmpc_plan_build``)."""

    _versions = itertools.count(1)

    def __init__(self, X_r: torch.Tensor, Y_r: torch.Tensor, hyper: GPHyperParams):
        dev = _lib.require_hip_device(X_r.device)
        self.version = next(RealDataPlan._versions)               # identifies the plan for caches (id() can be reused)
        lib = _lib.load()
        self.hyper = hyper
        self.X_r = X_r.to(dtype=F64).contiguous()                 # (N_r, D)
        self.Y_r = Y_r.to(dtype=F64).contiguous()                 # (g_ny, N_r, T)
        N_r = self.X_r.shape[0]
        nan = torch.isnan(self.Y_r)
        if hyper.T == 1:
            has_grad = False
            if bool(nan.any()):
                raise NotImplementedError("NaN value labels in the real data are not supported")
        else:
            grad_nan = nan[:, :, 1:]
            if bool(nan[:, :, 0].any()) or not (bool(grad_nan.all()) or not bool(grad_nan.any())):
                raise NotImplementedError("real-data label mask must be 'value only' or 'all tasks' (uniform)")
            has_grad = not bool(grad_nan.any())
        self.real_has_grad = has_grad
        self.grid = _detect_tensor_grid(self.X_r)
        self.desc = _lib.make_gp_desc(hyper.g_ny, hyper.D, hyper.T, N_r, has_grad, hyper.ell, hyper.outputscale,
                                      hyper.noise, hyper.jitter, grid=self.grid)
        nbytes = lib.gpmpc_plan_bytes(self.desc)
        if nbytes == 0:
            _lib.check(-1, "gpmpc_plan_bytes")
        self.buf = torch.empty(nbytes // 8, dtype=F64, device=dev)
        info = torch.zeros(hyper.g_ny, dtype=torch.int32, device=dev)
        _lib.check(lib.gpmpc_plan_build(self.desc, _lib.dptr(self.X_r), _lib.dptr(self.Y_r), _lib.dptr(self.buf),
                                        _lib.dptr(info), _lib.current_stream_ptr()), "gpmpc_plan_build")
        _lib.host_wait(info)
        if int(info.max().item()) != 0:
            raise NotPSDError("Cholesky of the real-data covariance K_rr + Sigma failed")
        self.n_r = N_r * hyper.T if has_grad else N_r


def _detect_tensor_grid(X: torch.Tensor):
    """(n0, n1) if X (N, 2) is exactly meshgrid(axis0, axis1, indexing="ij") flattened row-major, else (0, 0)."""
    if X.dim() != 2 or X.shape[1] != 2:
        return (0, 0)
    Xc = X.detach().cpu()
    N = Xc.shape[0]
    first = Xc[0, 0]
    n1 = int((Xc[:, 0] == first).sum().item()) if N > 0 else 0
    if n1 == 0 or N % n1 != 0:
        return (0, 0)
    n0 = N // n1
    G = Xc.reshape(n0, n1, 2)
    ok = bool((G[:, :, 0] == G[:, :1, 0]).all()) and bool((G[:, :, 1] == G[:1, :, 1]).all())
    return (n0, n1) if ok else (0, 0)


class JointFactorCache:
    """Caller-owned factor cache of ``gpmpc_joint_sample`` (include/gpmpc_hip.h): per chain the hallucinated rows of the
    train-side factor.  The reference re-factorises the whole conditioning set on every ``model_i(x)`` call; between two
    resets the hallucinated set only grows (``src/agent.py:164-202, 261-272``), so the rows of the slots that were there
    at the previous call are reused.  The façade vouches for validity by comparing the current slot list and the points of
    the cached rows with a snapshot taken when they were written (the factor does not depend on the labels)."""

    MAX_BYTES = 64 << 30          # cache budget (the per-GPU shard of BASELINE configs[4] needs ~7 GB, configs[4] whole on one GPU -
                                  # Ns = 8192 - 52 GB of the 288 GB); a batch that needs more caches the factor rows of a PREFIX of
                                  # its samples (``n_samples``), the others recompute (and the draw leaves no pending rows)

    def __init__(self):
        self.buf = None
        self.pending = None           # the last call's pending rows (note_pending)
        self.rows = 0                 # capacity (label rows per chain)
        self.key = None               # (Ns, g_ny, n_r, T, plan version)
        self.slots = None             # int32 device tensor: the slots whose rows are valid
        self.Xbuf = None              # (Ns, g_ny, rows, D) snapshot buffer of the points behind them, n_pts of it in use
        self.n_pts = 0
        self.T = 1
        self.lineage = None
        self.n_samples = 0            # samples whose chains are cached (== Ns unless the budget is too small for all)
        self.enabled = True

    def invalidate(self):
        self.slots = None
        self.n_pts = 0
        self.lineage = None
        self.pending = None

    @property
    def X(self):
        return None if self.slots is None else self.Xbuf[:, :, :self.n_pts]

    def prepare(self, mdl: "HipGPModel", Ns: int, n_ho: int):
        """-> (buffer or None, capacity, n_cached) for a call with ``n_ho`` observed hallucinated slots."""
        if not self.enabled or n_ho < 16:
            return None, 0, 0
        lib = _lib.load()
        hy = mdl.hyper
        key = (Ns, hy.g_ny, mdl.plan.n_r, hy.T, mdl.plan.version)
        if self.buf is None or key != self.key or n_ho > self.rows:
            # the set grows by the same number of slots every SQP iteration: room for the Agent's bound on it (max_sqp_iter * H
            # points) where there is one, else for four of them where that fits
            n_samp = Ns
            # The hint is honoured only when the set it describes is REACHABLE (it fits gpmpc_joint_sample's row limit: the
            # shipped car's max_sqp_iter * H * T = 22500 does not - the size then follows the 4x rule, which is monotonic in Ns;
            # same rule as HipPosterior's workspace hint below).  A reachable hint is honoured in full: when the rows do not fit the
            # byte budget for every sample, a PREFIX of the samples is cached at that row count (a smaller row count would have to
            # regrow - and start from zero cached rows - in the middle of the SQP loop)
            hint = getattr(mdl, "_ws_cache", {}).get("joint_points_hint")
            n_hint = int(hint) * hy.T if hint else 0
            hint_ok = n_hint >= n_ho and n_hint + 1 <= MAX_JOINT_ROWS
            if hint_ok:
                rows = min(MAX_JOINT_ROWS, max(256, -(-n_hint // 128) * 128))
                nbytes = lib.gpmpc_joint_cache_bytes(mdl.plan.desc, Ns, rows)
            else:
                for mult in (4.0, 2.0, 1.25):
                    rows = min(MAX_JOINT_ROWS, max(256, -(-int(mult * n_ho) // 128) * 128))
                    nbytes = lib.gpmpc_joint_cache_bytes(mdl.plan.desc, Ns, rows)
                    if 0 < nbytes <= self.MAX_BYTES:
                        break
            if nbytes > self.MAX_BYTES:                                        # not for every sample: a prefix of them
                per_sample = lib.gpmpc_joint_cache_bytes(mdl.plan.desc, 1, rows)
                n_samp = int(self.MAX_BYTES // per_sample) if per_sample > 0 else 0
                nbytes = lib.gpmpc_joint_cache_bytes(mdl.plan.desc, n_samp, rows) if n_samp > 0 else 0
            if nbytes == 0:
                self.buf, self.rows, self.key, self.n_samples = None, 0, None, 0
                self.invalidate()
                return None, 0, 0
            had = self.buf is not None
            self.buf = None                                                    # release the old one first, back to the driver
            if had and mdl.plan.X_r.is_cuda:
                torch.cuda.empty_cache()
            self.buf = torch.empty(nbytes // 8, dtype=F64, device=mdl.plan.X_r.device)
            self.Xbuf = torch.empty(n_samp, hy.g_ny, rows, hy.D, dtype=F64, device=mdl.plan.X_r.device)
            self.n_samples = n_samp
            self.rows, self.key = rows, key
            self.invalidate()
        n_c = 0
        if self.slots is not None:
            n_old, n_pts = int(self.slots.numel()), self.n_pts
            # append-only growth: the old slot list is a prefix of the new one and the points it was built on are unchanged
            if n_old <= n_ho and n_pts <= mdl.n_h:
                lin = getattr(mdl, "lineage", None)
                if lin is not None and self.lineage is not None and not _VERIFY_CACHE:
                    # the Agent counts the history of its hallucinated set: same generation = the first n_pts points and
                    # their (all observed) slots are the ones the rows were computed from - no tensor comparison, no sync
                    same = lin[0] == self.lineage[0] and n_pts <= lin[1] and n_old == n_pts * mdl.hyper.T
                else:
                    same = bool(torch.equal(mdl.h_slots[:n_old], self.slots)) \
                        and bool(torch.equal(mdl.hall_X[:self.n_samples, :, :n_pts], self.X))
                if same:
                    n_c = n_old
        return self.buf, self.rows, n_c

    # -- pending rows (include/gpmpc_hip.h, gpmpc_joint_sample_pending) ------------------------------------------------------------
    def pending_ok(self, mdl: "HipGPModel", n_ho: int, n_c: int) -> bool:
        """May the call about to be made treat the cache rows ``n_c .. n_ho - 1`` as the previous call's pending rows?  Yes when
        that call wrote them (``note_pending``) behind exactly the ``n_c`` slots that are cached now, and the hallucinated set has
        since grown by exactly that call's test points with all their tasks observed - vouched for by the Agent (it appends the
        points it drew at and says so: ``_ws_cache['joint_pending_points']``; same hallucinated-set generation), or, with
        GPMPC_VERIFY_FACTOR_CACHE=1, by comparing the points themselves."""
        p = self.pending
        if p is None or n_c == 0 or n_c != p["n_ho"] or n_ho != n_c + p["m"] * p["T"] or mdl.n_h != p["n_pts"] + p["m"]:
            return False
        lin = getattr(mdl, "lineage", None)
        tok = getattr(mdl, "_ws_cache", {}).get("joint_pending_points")
        if lin is None or tok is None or tok[0] != lin[0] or tok[1] != mdl.n_h or tok[2] is not p["post"]:
            return False
        if p["gen"] != lin[0] or n_ho != mdl.n_h * mdl.hyper.T:
            return False
        if _VERIFY_CACHE and not bool(torch.equal(mdl.hall_X[:, :, p["n_pts"]:], p["post"]._x)):
            return False
        return True

    def note_pending(self, mdl: "HipGPModel", post: "HipPosterior", n_ho: int, m: int, written: bool):
        lin = getattr(mdl, "lineage", None)
        self.pending = None
        if written and lin is not None and self.slots is not None and int(self.slots.numel()) == n_ho:
            self.pending = {"n_ho": n_ho, "m": m, "T": int(mdl.hyper.T), "n_pts": mdl.n_h, "gen": lin[0], "post": post}

    def rewind(self, n_slots: int):
        """Forget the rows beyond the first ``n_slots`` (benchmarks: put the cache back into the state it had before a
        draw, so that repeated timed draws do the work of the first one).  Pending rows (``pending_ok``) are forgotten too: a
        draw that used them has factorised their diagonal block in place."""
        self.pending = None
        if self.slots is not None:
            n = max(0, min(int(n_slots), int(self.slots.numel())))
            self.slots = self.slots[:n]
            if self.lineage is not None:                           # rows vouched for by lineage: whole points only
                self.n_pts = min(self.n_pts, n // self.T)
                self.slots = self.slots[:self.n_pts * self.T]

    @property
    def n_valid(self) -> int:
        return 0 if self.slots is None else int(self.slots.numel())

    def commit(self, mdl: "HipGPModel", n_ho: int, ok: bool, n_cached: int = 0):
        """After a call that filled the cache for all ``n_ho`` slots (``ok``: no factorisation failure)."""
        if self.buf is None or not ok or n_ho > self.rows or mdl.n_h > self.rows:
            self.invalidate()
            return
        if n_cached == n_ho and self.slots is not None and self.n_pts == mdl.n_h:
            return                                                 # nothing new was written (mean- / covariance-only repeats)
        # `prepare` has just verified the first n_pts points against the snapshot: only the appended ones are copied
        keep = self.n_pts if (self.slots is not None and n_cached > 0) else 0
        self.Xbuf[:, :, keep:mdl.n_h] = mdl.hall_X[:self.n_samples, :, keep:]
        self.n_pts = mdl.n_h
        self.slots = mdl.h_slots[:n_ho].clone()
        self.T = int(mdl.hyper.T)
        lin = getattr(mdl, "lineage", None)
        # rows vouched for by lineage need every slot of every point observed (slot list == 0 .. T n_pts - 1)
        self.lineage = lin if (lin is not None and n_ho == mdl.n_h * mdl.hyper.T) else None


class HipPosterior:
    """What ``model_i(x)`` returns: a lazily evaluated joint posterior at the ``m`` test points of every chain."""

    def __init__(self, model: "HipGPModel", x: torch.Tensor):
        self._model = model
        self._x = x.to(dtype=F64).contiguous()
        assert self._x.dim() == 4 and tuple(self._x.shape[:2]) == tuple(model.batch_shape)
        self._mean = self._var = self._covar = self._root = None
        self.last_info = None
        self.used_eigh = False

    # -- one kernel launch ----------------------------------------------------------------------------------
    def _run(self, z: Optional[torch.Tensor], clip: bool, beta: float = 0.0, var_zero_thr: float = -1.0,
             want_covar: bool = False, want_root: bool = False, root_mode: int = _lib.ROOT_AUTO,
             raise_chol_fail: bool = True):
        mdl = self._model
        lib = _lib.load()
        hy = mdl.hyper
        Ns, g_ny, m, _ = self._x.shape
        dev = self._x.device
        shape = (Ns, g_ny, m, hy.T)
        if z is None:
            z = torch.zeros(shape, dtype=F64, device=dev)
        z = z.to(device=dev, dtype=F64).contiguous()
        if tuple(z.shape) != shape:
            raise RuntimeError(f"base_samples shape {tuple(z.shape)} does not match the mean shape {shape}")
        mean, var, y = (torch.empty(shape, dtype=F64, device=dev) for _ in range(3))
        covar = torch.empty((Ns, g_ny, m * hy.T, m * hy.T), dtype=F64, device=dev) if want_covar else None
        root = torch.zeros((Ns, g_ny, m * hy.T, m * hy.T), dtype=F64, device=dev) if want_root else None
        info = torch.zeros((Ns, g_ny), dtype=torch.int32, device=dev)
        n_ho = int(mdl.h_slots.numel())
        if n_ho + 1 + m * hy.T > MAX_JOINT_ROWS or m * hy.T > MAX_JOINT_TEST_SLOTS:
            raise _lib.GpmpcError(
                f"joint draw needs {n_ho} hallucinated + 1 + {m * hy.T} test label rows per chain; the kernels are "
                f"instantiated for <= {MAX_JOINT_ROWS} rows and <= {MAX_JOINT_TEST_SLOTS} test slots (m*T).  In the SQP "
                f"loop the hallucinated set grows by H*T rows per iteration and is reset at sqp_iter == 0: with H*T = "
                f"{m * hy.T} that is at most {(MAX_JOINT_ROWS - 1 - m * hy.T) // max(m * hy.T, 1)} iterations after the "
                f"reset (the reference's max_sqp_iter of 150 is not reachable: its cost grows with the cube of the rows).")
        ws_bytes = lib.gpmpc_joint_workspace_bytes(mdl.plan.desc, Ns, n_ho, m)
        # (the Agent's bound on the conditioning set: one allocation for the closed loop's largest draw instead of one per SQP
        # iteration of the first MPC step; ignored where it would not fit the entry point's limits or a 32 GiB budget)
        hint = mdl._ws_cache.get("joint_points_hint")
        if hint:
            n_hint = int(hint) * hy.T
            if n_ho < n_hint and n_hint + 1 + m * hy.T <= MAX_JOINT_ROWS:
                hb = lib.gpmpc_joint_workspace_bytes(mdl.plan.desc, Ns, n_hint, m)
                if ws_bytes < hb <= JOINT_WS_HINT_MAX_BYTES:
                    ws_bytes = hb
        ws = mdl._workspace(ws_bytes)
        fcache = mdl._ws_cache.setdefault("joint_factor_cache", JointFactorCache())
        fbuf, frows, n_c = fcache.prepare(mdl, Ns, n_ho)
        n_cs = fcache.n_samples if fbuf is not None else Ns       # samples [0, n_cs) use the cache, the rest recompute

        # pending rows: this call's test points become the next call's new slots in the SQP loop (the Agent appends them); the
        # whole batch has to be in the cache, and a draw that only repeats (mean / covariance of the same call) changes nothing
        whole = fbuf is not None and n_cs >= Ns
        pend = 0
        if whole and os.environ.get("GPMPC_JOINT_PENDING", "1") != "0":
            pend = _lib.PENDING_WRITE | (_lib.PENDING_USE if fcache.pending_ok(mdl, n_ho, n_c) else 0)
        self.used_pending = bool(pend & _lib.PENDING_USE)
        fcache.pending = None                                      # consumed (or stale) either way

        def call(lo, hi, cache, mode):
            sl = lambda t: None if t is None else t[lo:hi]
            _lib.check(lib.gpmpc_joint_sample_pending(
                mdl.plan.desc, _lib.dptr(mdl.plan.buf), _lib.dptr(mdl.plan.X_r), hi - lo, mdl.n_h,
                _lib.dptr(sl(mdl.hall_X)) if mdl.n_h else None, _lib.dptr(sl(mdl.hall_Y)) if mdl.n_h else None,
                _lib.dptr(mdl.h_slots) if n_ho else None, n_ho, m, _lib.dptr(sl(self._x)), _lib.dptr(sl(z)),
                float(var_zero_thr), float(beta), int(bool(clip)), _lib.dptr(sl(mean)), _lib.dptr(sl(var)), _lib.dptr(sl(y)),
                _lib.dptr(sl(covar)), _lib.dptr(sl(root)), int(mode), _lib.dptr(sl(info)), _lib.dptr(ws), ws.numel() * 8,
                _lib.current_stream_ptr(), _lib.dptr(fbuf) if cache else None, int(frows) if cache else 0,
                int(n_c) if cache else 0, int(pend) if cache else 0), "gpmpc_joint_sample")

        self._pending_written = False
        ev = mdl._ws_cache.get("joint_time_events")               # benchmarks: a pair of timing events around the entry point's launches
        if ev is not None:
            ev[0].record()
        if n_cs >= Ns:
            call(0, Ns, fbuf is not None, root_mode)
            self._pending_written = bool(pend) and bool(lib.gpmpc_joint_pending_written())
        else:
            # The cache holds a prefix of the samples: two launches.  "One chain failed every retry => the WHOLE batch takes
            # the eigendecomposition root" (A.7 step 4) spans both: the part that did not fall back by itself is redrawn.
            call(0, n_cs, True, root_mode)
            call(n_cs, Ns, False, root_mode)
            if root_mode == _lib.ROOT_AUTO and m * hy.T > 1:
                e = [bool(_or_reduce(info[a:b]) & _lib.INFO_ROOT_EIGH) for a, b in ((0, n_cs), (n_cs, Ns))]
                if e[0] != e[1]:
                    if e[0]:
                        call(n_cs, Ns, False, _lib.ROOT_EIGH)
                    else:
                        info[:n_cs] = 0
                        call(0, n_cs, True, _lib.ROOT_EIGH)
        if ev is not None:
            ev[1].record()
        self.n_cached_rows = n_c
        self._mean, self._var = mean, var
        if want_covar:
            self._covar = covar
        if want_root:
            self._root = root
        bits = _or_reduce(info)
        self.last_info = info
        self.last_bits = int(bits)
        self.used_eigh = bool(bits & _lib.INFO_ROOT_EIGH)
        if fbuf is not None:
            fcache.commit(mdl, n_ho, ok=not (bits & _lib.INFO_TRAIN_CHOL_FAIL), n_cached=n_c)
            fcache.note_pending(mdl, self, n_ho, m, self._pending_written and not (bits & _lib.INFO_TRAIN_CHOL_FAIL))
        if (bits & _lib.INFO_TRAIN_CHOL_FAIL) and raise_chol_fail:
            raise NotPSDError("Cholesky of the training covariance (real + hallucinated data) failed")
        if bits & _lib.INFO_VAR_CLAMPED:
            warnings.warn("Negative variance values detected; rounding them up to 1e-10.", NumericalWarning)
        if bits & _lib.INFO_ROOT_JITTER_MASK:
            warnings.warn(f"posterior covariance not p.d. - added jitter of up to "
                          f"{hy.jitter * 10 ** (((bits & _lib.INFO_ROOT_JITTER_MASK) >> 1) - 1):.1e} to the diagonal",
                          NumericalWarning)
        if bits & _lib.INFO_ROOT_EIGH:
            # gpytorch: NotPSDError inside root_decomposition -> eigh root for the WHOLE batch (A.7 step 4); drawn by
            # the second kernel of the same gpmpc_joint_sample call
            warnings.warn("Cholesky of the posterior covariance failed after 3 jitter retries; using the "
                          "eigendecomposition root for the whole batch", NumericalWarning)
        if bits & _lib.INFO_EIGH_NOCONV:
            warnings.warn("Jacobi eigensolver of the posterior covariance hit its sweep limit", NumericalWarning)
        return y, bits

    # -- distribution surface -----------------------------------------------------------------------------------
    @property
    def mean(self) -> torch.Tensor:
        if self._mean is None:
            self._run(None, clip=False)
        return self._mean

    @property
    def variance(self) -> torch.Tensor:
        if self._var is None:
            self._run(None, clip=False)
        return self._var

    @property
    def stddev(self) -> torch.Tensor:
        return self.variance.sqrt()

    @property
    def covariance_matrix(self) -> torch.Tensor:
        if self._covar is None:
            self._run(None, clip=False, want_covar=True)
        return self._covar

    def confidence_region(self):
        s2 = self.stddev * 2
        return self.mean - s2, self.mean + s2

    def sample(self, base_samples: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``y = mean + R z``.  ``base_samples`` has the mean's shape ``(Ns, g_ny, m, T)``;
        None draws z internally like gpytorch does (``randn``, device generator)."""
        return self._sample(base_samples, clip=False)

    def _sample(self, base_samples, clip, beta=0.0, var_zero_thr=-1.0, want_root=False):
        Ns, g_ny, m, _ = self._x.shape
        T = self._model.hyper.T
        if base_samples is None:
            base_samples = torch.randn(Ns, g_ny, m, T, dtype=F64, device=self._x.device)
        group = self._model.dist_group
        if group is None:
            y, bits = self._run(base_samples, clip, beta, var_zero_thr, want_root=want_root)
            return y
        # Samples sharded over ranks: two properties of the draw belong to the WHOLE batch (all ranks' chains) - a failed
        # training factorisation raises on every rank (a rank raising alone would leave the others in the collective), and
        # the eigh fallback: a rank none of whose chains failed redraws with the eigendecomposition root when a chain on
        # another rank did.  One all-reduce of both flags; every rank takes the same branch.
        import torch.distributed as dist
        y, bits = self._run(base_samples, clip, beta, var_zero_thr, want_root=want_root, raise_chol_fail=False)
        flags = torch.tensor([1 if (bits & _lib.INFO_TRAIN_CHOL_FAIL) else 0, 1 if self.used_eigh else 0],
                             dtype=torch.int32, device=self._x.device)
        dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
        _lib.host_wait(flags)
        chol_fail, any_eigh = (int(v) for v in flags.tolist())
        if chol_fail:
            raise NotPSDError("Cholesky of the training covariance (real + hallucinated data) failed"
                              + ("" if bits & _lib.INFO_TRAIN_CHOL_FAIL else " on another rank's samples"))
        if any_eigh and not self.used_eigh and m * T > 1:
            y, bits = self._run(base_samples, clip, beta, var_zero_thr, want_root=want_root, root_mode=_lib.ROOT_EIGH)
        return y

    @property
    def root(self) -> Optional[torch.Tensor]:
        """The eigendecomposition root of the last draw (only kept when the draw asked for it; tests)."""
        return self._root


_INFO_BITS = (0x1, 0x2, 0x4, 0x8, 0x10, 0x20, 0x40, 0x80, 0x100, 0x200)


def _or_reduce(info: torch.Tensor, group=None) -> int:
    """bitwise OR over an int32 tensor and, with ``group``, over its ranks: ONE launch (``gpmpc_or_reduce_words``) and one word
    read back (it used to be a reduction per flag bit: 21 launches per joint draw)."""
    if info.numel() == 0 and group is None:
        return 0
    if info.is_cuda:
        word = torch.zeros(1, dtype=torch.int32, device=info.device)
        v = info.contiguous()
        _lib.check(_lib.load().gpmpc_or_reduce_words(_lib.dptr(v), v.numel(), _lib.dptr(word), _lib.current_stream_ptr()),
                   "gpmpc_or_reduce_words")
        if group is not None:
            import torch.distributed as dist
            packed = torch.stack([(word[0] & b) for b in _INFO_BITS])          # all_reduce has MAX, not OR
            dist.all_reduce(packed, op=dist.ReduceOp.MAX, group=group)
            _lib.host_wait(packed)
            bits = 0
            for x in packed.tolist():
                bits |= int(x)
            return bits
        return int(_lib.to_host(word)[0])
    v = info.flatten()                                                         # CPU tensors (the gloo tests)
    packed = torch.stack([(v & b).max() for b in _INFO_BITS]) if v.numel() else torch.zeros(len(_INFO_BITS), dtype=info.dtype)
    if group is not None:
        import torch.distributed as dist
        dist.all_reduce(packed, op=dist.ReduceOp.MAX, group=group)
    bits = 0
    for x in packed.tolist():
        bits |= int(x)
    return bits


_ALL_SLOTS: dict = {}


def _all_slots(n: int, device) -> torch.Tensor:
    """0 .. n-1 as int32 on ``device`` (cached: the slot list of a hallucinated set without NaN labels)."""
    key = (str(device), n)
    t = _ALL_SLOTS.get(key)
    if t is None:
        if len(_ALL_SLOTS) > 64:
            _ALL_SLOTS.clear()
        t = torch.arange(n, dtype=torch.int32, device=device)
        _ALL_SLOTS[key] = t
    return t


def _observed_slots(hall_Y: torch.Tensor, dist_group=None) -> torch.Tensor:
    """Observed hallucinated label slots (ascending, slot = point * T + task).  gpytorch's "mask" policy drops a slot that
    is NaN in ANY batch element; with the samples sharded over ranks the batch is the whole sample
    set, so the NaN flags are OR-ed over the process group."""
    Ns, g_ny, n_h, T = hall_Y.shape
    nan_any = torch.isnan(hall_Y).reshape(Ns * g_ny, n_h * T).any(dim=0)
    if dist_group is not None:
        import torch.distributed as dist
        flag = nan_any.to(torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=dist_group)
        nan_any = flag.bool()
    return torch.nonzero(~nan_any).flatten().to(torch.int32).contiguous()


class HipGPModel:
    """Conditioning set = shared real data (factorised plan) + per-sample hallucinated data."""

    def __init__(self, plan: RealDataPlan, hall_X: torch.Tensor, hall_Y: torch.Tensor, batch_shape,
                 ws_cache: Optional[dict] = None, dist_group=None, all_observed: bool = False, lineage=None):
        """``all_observed``: the caller vouches that ``hall_Y`` holds no NaN (every appended label came out of a draw and no
        min-distance filter ran): the observed-slot scan (isnan / any / nonzero: seven launches and a host sync) is skipped.
        ``lineage = (generation, n_points)``: identifies an append-only history of the hallucinated set - two models with
        the same generation hold the same first ``min(n_points)`` points (``Agent`` counts it) - which is what lets the joint
        factor cache vouch for its rows without comparing tensors."""
        self.lineage = lineage
        self.plan = plan
        self.dist_group = dist_group
        self.hyper = plan.hyper
        self.batch_shape = torch.Size(batch_shape)
        Ns, g_ny = self.batch_shape
        dev = plan.X_r.device
        self.hall_X = hall_X.to(device=dev, dtype=F64).contiguous()      # (Ns, g_ny, n_h, D)
        self.hall_Y = hall_Y.to(device=dev, dtype=F64).contiguous()      # (Ns, g_ny, n_h, T)
        self.n_h = int(self.hall_X.shape[2])
        assert tuple(self.hall_X.shape[:2]) == (Ns, g_ny) and self.hall_Y.shape[-1] == self.hyper.T
        if self.n_h and all_observed:
            self.h_slots = _all_slots(self.n_h * self.hyper.T, dev)
        elif self.n_h:
            self.h_slots = _observed_slots(self.hall_Y, dist_group)
        else:
            self.h_slots = torch.empty(0, dtype=torch.int32, device=dev)
        self._ws_cache = ws_cache if ws_cache is not None else {}
        self.use_grad = self.hyper.use_grad

    def _workspace(self, nbytes: int) -> torch.Tensor:
        buf = self._ws_cache.get("joint")
        if buf is None or buf.numel() * 8 < nbytes:
            # grow-only, with headroom (the conditioning set grows every SQP iteration); the outgrown buffer goes back to
            # the driver instead of staying in torch's cache, where nothing of that size would ever reuse it
            had = buf is not None
            buf = None
            self._ws_cache.pop("joint", None)
            if had and self.plan.X_r.is_cuda:
                torch.cuda.empty_cache()
            buf = torch.empty(((nbytes + nbytes // 4) + 7) // 8, dtype=F64, device=self.plan.X_r.device)
            self._ws_cache["joint"] = buf
        return buf

    # -- reference attribute surface ---------------------------------------------------------------------------
    @property
    def train_inputs(self):
        Ns, g_ny = self.batch_shape
        Xr = self.plan.X_r.reshape(1, 1, *self.plan.X_r.shape).expand(Ns, g_ny, -1, -1)
        return (torch.cat([Xr, self.hall_X], dim=2),)

    @property
    def train_targets(self):
        Ns, _ = self.batch_shape
        Yr = self.plan.Y_r.unsqueeze(0).expand(Ns, -1, -1, -1)
        return torch.cat([Yr, self.hall_Y], dim=2)

    def eval(self):
        return self

    def cuda(self):
        return self

    def __call__(self, x: torch.Tensor) -> HipPosterior:
        return HipPosterior(self, x.to(self.plan.X_r.device))
