"""``Agent``: the reference's hot-path API (reference ``src/agent.py``), backed by libgpmpc_hip.so.

Same constructor, attribute names, method names, argument meaning, shapes, dtypes and return TYPES as the reference
class, so that reference ``src/DEMPC.py`` / ``src/solver.py`` / the benchmarking scripts can consume it unchanged
(SURVEY.md section 8b lists every use).  What differs is underneath:

* the GP is never rebuilt: ``train_hallucinated_dynGP`` only re-points ``model_i`` at (shared real-data plan,
  current hallucinated tensors); the real data are factorised once (``gpmpc_plan_build``) and never tiled;
* ``sample_gp`` is ONE kernel launch (``gpmpc_joint_sample``: posterior, root with gpytorch's jitter chain, sample,
  variance-is-zero replacement, beta clip);
* ``dyn_fg_jacobians`` assembles value and Jacobians on the device (``gpmpc_assemble_jacobians``) and returns the
  three float64 numpy arrays the solver indexes;
* whole forward-sampling rollouts run in one launch through ``sampling_gpmpc_amd.rollout``.

Host-only logic (base-sample generation, tensor reshapes, tightenings, dataset bookkeeping) works without a GPU;
every method that needs GP arithmetic raises ``GpmpcError`` when no HIP device / library is present.
"""
from __future__ import annotations

import warnings
from typing import Optional

import itertools

import numpy as np
import torch

from . import _lib
from .gp_model import GPHyperParams, HipGPModel, RealDataPlan
from .reachable_set import get_reachable_set_ball

F64 = torch.float64


def random_vector_within_bounds(params, g_ny: int, T: int, device="cpu", mode: Optional[str] = None) -> torch.Tensor:
    """Base samples ``(n_mpc, n_itrs, Ns, g_ny, H, T)``: i.i.d. N(0,1) vectors of shape (g_ny, H, T), the WHOLE vector
    redrawn until every entry lies in [-beta, beta] (reference ``src/agent.py:76-104``).

    mode "reference": one ``torch.normal`` call per candidate on the global CPU generator, call for call what the
    reference does, so a seeded run reproduces its stream exactly (the reference draws on its compute device; CUDA
    generator streams are not reproducible on ROCm, so the draw is pinned to the CPU generator here).
    mode "vectorized": same distribution, candidates drawn in blocks (for Ns in the 1e5 range).
    mode "counter": same distribution from a counter-based stream keyed by (seed, mpc step, SQP iteration, GLOBAL sample
    id, attempt): ``agent.base_sample_offset`` (default 0) is the global id of this Agent's first sample, so a rank of a
    sample-sharded run generates exactly its own shard - on its own device, nothing of the other shards - and the
    assembled run is the same for every GPU count (``sampling_gpmpc_amd.distributed``).
    """
    H = params["optimizer"]["H"]
    n_dyn = params["agent"]["num_dyn_samples"]
    beta = params["agent"]["Dyn_gp_beta"]
    n_mpc = params["common"]["num_MPC_itrs"]
    n_itrs = params["optimizer"]["SEMPC"]["max_sqp_iter"]
    total = n_mpc * n_itrs * n_dyn
    if mode is None:
        mode = params["agent"].get("base_sample_generator", "reference" if total <= 200_000 else "vectorized")
    out = torch.empty(n_mpc, n_itrs, n_dyn, g_ny, H, T, dtype=F64)
    if mode == "reference":
        for j in range(n_mpc):
            for i in range(n_itrs):
                k = 0
                while k < n_dyn:
                    w = torch.normal(0, 1, size=(1, g_ny, H, T), dtype=F64)
                    if torch.all(w >= -beta) and torch.all(w <= beta):
                        out[j, i, k] = w[0]
                        k += 1
    elif mode == "vectorized":
        flat = out.view(total, g_ny * H * T)
        filled = 0
        while filled < total:
            cand = torch.randn(max(1024, int(1.5 * (total - filled))), g_ny * H * T, dtype=F64)
            cand = cand[(cand.abs() <= beta).all(dim=1)]
            n = min(cand.shape[0], total - filled)
            flat[filled:filled + n] = cand[:n]
            filled += n
    elif mode == "counter":
        return counter_base_samples(n_mpc, n_itrs, n_dyn, g_ny, H, T, beta,
                                    seed=int(params["agent"].get("base_sample_seed", 123456)),
                                    offset=int(params["agent"].get("base_sample_offset", 0)), device=device)
    else:
        raise ValueError(f"unknown base_sample_generator {mode!r}")
    return out.to(device)


_M64 = (1 << 64) - 1


def _s64(v: int) -> int:
    """Python int -> the int64 with the same 64 bits."""
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(x: torch.Tensor, k: int) -> torch.Tensor:
    """Logical right shift of an int64 tensor (torch's >> is arithmetic)."""
    return (x >> k) & ((1 << (64 - k)) - 1)


def _mix64(x: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser; int64 arithmetic wraps, which is the mod 2^64 arithmetic the mixer is defined in."""
    x = (x ^ _lsr(x, 30)) * _s64(0xBF58476D1CE4E5B9)
    x = (x ^ _lsr(x, 27)) * _s64(0x94D049BB133111EB)
    return x ^ _lsr(x, 31)


def counter_base_samples(n_mpc, n_itrs, n_dyn, g_ny, H, T, beta, seed=123456, offset=0, device="cpu", _force_torch=False) -> torch.Tensor:
    """``(n_mpc, n_itrs, n_dyn, g_ny, H, T)`` truncated-normal vectors with the whole-vector rejection rule of reference
    ``src/agent.py:84-100``, each vector a pure function of (seed, j, i, offset + s): attempt ``a`` of a vector draws its
    ``V = g_ny H T`` entries from the hashed counters ``2 (a V + e)``, ``2 (a V + e) + 1`` (Box-Muller), and the first
    attempt with every entry in [-beta, beta] is kept.  Generated on ``device`` one (j, i) slab at a time."""
    V = g_ny * H * T
    dev = torch.device(device)
    if dev.type == "cuda" and not _force_torch:
        # one launch of gpmpc_base_samples (one wave per vector, the rejection loop inside it): bit-identical to the torch form
        # below evaluated on the same device (tests/test_hip_parity.py), which costs ~15 launches per attempt and (j, i) slab
        out = torch.empty(n_mpc, n_itrs, n_dyn, V, dtype=F64, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().gpmpc_base_samples(int(seed) & _M64, n_mpc, n_itrs, int(offset), n_dyn, V, float(beta), _lib.dptr(out), None,
                                                      _lib.current_stream_ptr()), "gpmpc_base_samples")
        return out.reshape(n_mpc, n_itrs, n_dyn, g_ny, H, T)
    out = torch.empty(n_mpc, n_itrs, n_dyn, V, dtype=F64, device=dev)
    sid = torch.arange(offset, offset + n_dyn, dtype=torch.int64, device=dev)
    e2 = 2 * torch.arange(V, dtype=torch.int64, device=dev)
    two_pi = 2.0 * 3.141592653589793
    for j in range(n_mpc):
        for i in range(n_itrs):
            key = _mix64(sid * _s64(0x9E3779B97F4A7C15) + _s64(seed * 0xD1B54A32D192ED03 + (j * n_itrs + i) * 0x8CB92BA72F3D8DD7))
            slab = out[j, i]
            pending = torch.arange(n_dyn, device=dev)
            attempt = 0
            while pending.numel():
                c = key[pending].unsqueeze(1) + (e2 + 2 * attempt * V).unsqueeze(0) * _s64(0xDA942042E4DD58B5)
                u1 = (_lsr(_mix64(c), 11).to(F64) + 0.5) * (1.0 / 9007199254740992.0)
                u2 = (_lsr(_mix64(c + _s64(0xDA942042E4DD58B5)), 11).to(F64) + 0.5) * (1.0 / 9007199254740992.0)
                w = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(two_pi * u2)
                ok = (w.abs() <= beta).all(dim=1)
                slab[pending[ok]] = w[ok]
                pending = pending[~ok]
                attempt += 1
    return out.reshape(n_mpc, n_itrs, n_dyn, g_ny, H, T)


_HALL_GENERATION = itertools.count(1)


class Agent(object):
    def __init__(self, params, env_model) -> None:
        self.my_key = 0
        self.params = params
        self.env_model = env_model
        ag = params["agent"]
        self.g_nx, self.g_nu, self.g_ny = ag["g_dim"]["nx"], ag["g_dim"]["nu"], ag["g_dim"]["ny"]
        self.ns = ag["num_dyn_samples"]
        self.nx, self.nu = ag["dim"]["nx"], ag["dim"]["nu"]
        self.in_dim_x = self.g_nx + self.g_nu
        self.in_dim_y = 1 if params["env"]["use_model_without_derivatives"] else 1 + self.in_dim_x
        self.batch_shape = torch.Size([self.ns, self.g_ny])
        self.mean_shift_val = ag.get("mean_shift_val")
        self.converged = False

        if params["common"]["use_cuda"] and torch.cuda.is_available():
            self.use_cuda, self.torch_device = True, torch.device("cuda")
        else:
            self.use_cuda, self.torch_device = False, torch.device("cpu")
        # unlike the reference no global torch.set_default_device side effect is installed

        self.model_i = None
        self.model_i_call = None
        self.model_i_samples = None
        self.likelihood = None
        self.mpc_iter = 0
        # samples sharded over ranks (sampling_gpmpc_amd.distributed.make_sharded_agent): the process group through which
        # the reference's cross-sample couplings are reduced; None = this Agent holds every sample
        self.dist_group, self.shard, self.ns_global = None, None, self.ns
        self.debug_keep_root = False   # tests: keep the eigendecomposition root of the last joint draw (model_i_call.root)
        self._plans = {}          # T -> RealDataPlan (real block factorised once per label layout)
        self._ws_cache = {}
        # The hallucinated set grows by H points per SQP iteration and is reset once per MPC step - AFTER the model of iteration 0 has
        # been built (reference src/agent.py:261-272): a joint draw conditions on at most max_sqp_iter * H points.  The joint
        # workspace and the factor cache are sized for that once (multi-GiB device allocations cost 0.3-0.9 s each on this
        # driver: growing them at every iteration of the first MPC step made it four of them, `tools/debug/first_call_costs.py`).
        try:
            self._ws_cache["joint_points_hint"] = int(params["optimizer"]["SEMPC"]["max_sqp_iter"]) * int(params["optimizer"]["H"])
        except (KeyError, TypeError, ValueError):
            pass
        self._reset_hallucinated()

        X, Y = env_model.initial_training_data()
        self.Dyn_gp_X_train = X.to(self.torch_device)
        self.Dyn_gp_Y_train = Y.to(self.torch_device)
        if self.in_dim_y == 1:
            self.Dyn_gp_Y_train = self.Dyn_gp_Y_train[:, :, [0]]
        self.real_data_batch()
        self.planned_measure_loc = np.array([2])
        self.epistimic_random_vector = self.random_vector_within_bounds()
        if "terminal_tightening" in params["optimizer"]:
            self.tilde_eps_list, self.ci_list = get_reachable_set_ball(params, np.ones(params["optimizer"]["H"] + 1))

    # ---------------------------------------------------------------------------------------------------------
    # host-side bookkeeping
    # ---------------------------------------------------------------------------------------------------------
    # The hallucinated set ``Hallcinated_{X,Y}_train`` (reference attribute names, src/agent.py:52-62) with a LINEAGE: a
    # generation number that changes whenever the tensors are replaced other than by appending points, and a flag "no NaN
    # label".  ``train_hallucinated_dynGP`` hands both to the model: the joint factor cache then vouches for its rows by
    # comparing two integers (no ``torch.equal`` + host sync per draw) and the observed-slot scan is skipped.  Assigning the
    # attributes from outside starts a new generation; an edit IN PLACE is noticed through the tensors' version counters
    # (``_check_in_place_edits``; ``invalidate_factor_cache()`` announces one explicitly).
    @property
    def Hallcinated_X_train(self):
        return self._hall_X

    @Hallcinated_X_train.setter
    def Hallcinated_X_train(self, t):
        self._hall_X = t
        self._hall_gen = next(_HALL_GENERATION)
        self._hall_seen = None

    @property
    def Hallcinated_Y_train(self):
        return self._hall_Y

    @Hallcinated_Y_train.setter
    def Hallcinated_Y_train(self, t):
        self._hall_Y = t
        self._hall_gen = next(_HALL_GENERATION)
        self._hall_all_observed = False                   # unknown labels: scan them
        self._hall_seen = None

    def _hall_versions(self):
        """torch's in-place version counters of the hallucinated tensors (a view shares its base's counter)."""
        return (self._hall_X._version, self._hall_Y._version)

    def _check_in_place_edits(self):
        """An in-place edit of ``Hallcinated_{X,Y}_train`` from outside - the reference's own idiom ``Hallcinated_X_train[rejected]
        = ...`` - that nobody announced (``invalidate_factor_cache``): the tensors' version counters have moved since the Agent last
        assigned them.  Then the factor rows of the cache and the "no NaN label" flag are no longer vouched for: new generation.
        (No host sync: the counters are Python integers.)"""
        seen = getattr(self, "_hall_seen", None)
        now = self._hall_versions()
        if seen is not None and seen != now:
            self.invalidate_factor_cache()
        self._hall_seen = now

    def invalidate_factor_cache(self):
        """after an in-place edit of the hallucinated tensors (``prepare_dynamics_set``'s survivor replacement does one)"""
        self._hall_gen = next(_HALL_GENERATION)
        self._hall_all_observed = False
        self._hall_seen = self._hall_versions()

    def _reset_hallucinated(self):
        self._hall_X = torch.empty(self.ns, self.g_ny, 0, self.in_dim_x, dtype=F64, device=self.torch_device)
        self._hall_Y = torch.empty(self.ns, self.g_ny, 0, self.in_dim_y, dtype=F64, device=self.torch_device)
        self._hall_gen = next(_HALL_GENERATION)
        self._hall_all_observed = True
        self._hall_seen = self._hall_versions()

    def random_vector_within_bounds(self):
        return random_vector_within_bounds(self.params, self.g_ny, self.in_dim_y, device=self.torch_device)

    def real_data_batch(self):
        """reference ``src/agent.py:204-214`` tiles the real data Ns times; here the batch tensors are stride-0
        expanded VIEWS of the shared data (same shapes and values, no memory)."""
        self.Dyn_gp_X_train_batch = self.Dyn_gp_X_train.reshape(1, 1, *self.Dyn_gp_X_train.shape).expand(
            self.ns, self.g_ny, -1, -1)
        self.Dyn_gp_Y_train_batch = self.Dyn_gp_Y_train.unsqueeze(0).expand(self.ns, -1, -1, -1)

    def update_current_location(self, loc):
        self.current_location = loc

    def update_current_state(self, state):
        self.current_state = state
        self.update_current_location(state[: self.nx])

    def mpc_iteration(self, i):
        self.mpc_iter = i

    def get_next_to_go_loc(self):
        return self.planned_measure_loc

    def concatenate_real_hallucinated_data(self):
        return (torch.concat([self.Dyn_gp_X_train_batch, self.Hallcinated_X_train], dim=2),
                torch.concat([self.Dyn_gp_Y_train_batch, self.Hallcinated_Y_train], dim=2))

    def update_hallucinated_Dyn_dataset(self, newX, newY):
        """Append sampled points; optional min-distance filter (reference ``src/agent.py:164-202``)."""
        min_distance = self.params["agent"]["Dyn_gp_min_data_dist"]
        newX = newX.to(self.torch_device)
        newY = newY.to(self.torch_device)
        if min_distance >= 0.0:
            X_cond, _ = self.concatenate_real_hallucinated_data()
            dist_norm = torch.linalg.vector_norm(newX[:, :, None, :, :] - X_cond[:, :, :, None, :], dim=-1)
            filt = torch.any(dist_norm <= min_distance, dim=2)                     # (Ns, g_ny, m)
            newY = torch.where(filt.unsqueeze(-1), torch.full_like(newY, float("nan")), newY)
            if self.dist_group is not None:                                        # samples sharded over ranks
                from .distributed import filtered_in_all_samples
                all_s = filtered_in_all_samples(filt, self.dist_group)
            else:
                all_s = torch.all(filt, dim=0)
            keep = ~torch.any(all_s, dim=0)                                        # filtered in ALL samples -> drop
            newX, newY = newX[:, :, keep, :], newY[:, :, keep, :]
        # appended behind what is there: same generation (the factor rows of the earlier points stay valid); the labels are
        # draws (no NaN) unless the min-distance filter ran
        self._check_in_place_edits()                      # (an unannounced edit of the old points must not ride along into the new tensors)
        self._hall_X = torch.cat([self._hall_X, newX], 2)
        self._hall_Y = torch.cat([self._hall_Y, newY], 2)
        self._hall_seen = self._hall_versions()
        if min_distance >= 0.0:
            self._hall_all_observed = False
        # a draw that failed leaves NaN labels behind (negative 1 x 1 variance, an eigensolver that did not converge, a root that
        # failed with the eigh root switched off): the observed-slot scan has to see them - gpytorch's mask policy drops such slots
        call = self.model_i_call
        bits = int(getattr(call, "last_bits", 0)) if call is not None else 0
        root_nan = (bits & _lib.INFO_ROOT_FAIL) and not (bits & _lib.INFO_ROOT_EIGH)   # ROOT_FAIL | ROOT_EIGH = the batch was redrawn
        if root_nan or (bits & (_lib.INFO_NEG_1x1 | _lib.INFO_EIGH_NOCONV)):            # with the eigh root: finite labels (the car as shipped)
            self._hall_all_observed = False

    def get_batch_x_hat_u_diff(self, x_h, u_h):
        """x_h (H, Ns*nx), u_h (H, Ns, nu) -> (Ns, nx, H, nx+nu), state row replicated nx times (:480-501)."""
        H = self.params["optimizer"]["H"]
        # the (H, Ns*nx) / (H, Ns, nu) iterates go to the device as they are; the nx-fold replication happens there (the
        # replicated tensor is nx times the bytes: 8 MB over PCIe per SQP iteration at the configs[4] shard)
        x_h = torch.as_tensor(x_h, dtype=F64).to(self.torch_device)
        u_h = torch.as_tensor(u_h, dtype=F64).to(self.torch_device)
        xb = x_h.transpose(0, 1).reshape(self.ns, self.nx, H).transpose(1, 2)
        ub = u_h.transpose(0, 1).reshape(self.ns, H, self.nu)
        ret = torch.cat([xb, ub], 2)
        return ret.unsqueeze(1).expand(-1, self.nx, -1, -1).contiguous()

    def get_batch_x_hat(self, x_h, u_h):
        """x_h (H, Ns*nx), u_h (H, nu) shared by all samples -> (Ns, nx, H, nx+nu)   (:503-527)."""
        H = self.params["optimizer"]["H"]
        x_h = torch.as_tensor(x_h, dtype=F64).to(self.torch_device)
        u_h = torch.as_tensor(u_h, dtype=F64).to(self.torch_device)
        xb = x_h.transpose(0, 1).reshape(self.ns, self.nx, H).transpose(1, 2)
        ub = torch.ones(self.ns, H, 1, dtype=F64, device=self.torch_device) * u_h
        ret = torch.cat([xb, ub], 2)
        return ret.unsqueeze(1).expand(-1, self.nx, -1, -1).contiguous()

    # ---------------------------------------------------------------------------------------------------------
    # GP (device)
    # ---------------------------------------------------------------------------------------------------------
    def _plan(self, use_grad: bool) -> RealDataPlan:
        T = 1 + self.in_dim_x if use_grad else 1
        plan = self._plans.get(T)
        if plan is None:
            _lib.require_hip_device(self.torch_device)
            hyper = GPHyperParams.from_params(self.params, use_grad)
            Y = self.Dyn_gp_Y_train if use_grad else self.Dyn_gp_Y_train[:, :, [0]]
            if use_grad and Y.shape[-1] != T:
                raise RuntimeError("value+gradient model requested but the agent was built value-only "
                                   "(env.use_model_without_derivatives)")
            plan = RealDataPlan(self.Dyn_gp_X_train, Y, hyper)
            self._plans[T] = plan
        return plan

    def env_desc(self, use_feedback: Optional[bool] = None):
        p = self.params
        fb = p["agent"]["feedback"]["use"] if use_feedback is None else use_feedback
        tt = p["optimizer"].get("terminal_tightening", {})
        K = tt.get("K") if fb else None
        p0, p1 = self.env_model.env_params
        return _lib.make_env_desc(self.env_model.env_id, self.nx, self.nu, fb, p["optimizer"]["dt"], p0, p1, K,
                                  p["env"]["goal_state"])

    def train_hallucinated_dynGP(self, sqp_iter, use_model_without_derivatives=False):
        """(Re)define ``model_i`` on real + hallucinated data (reference ``src/agent.py:216-272``), including the
        order quirk: the model sees the PRE-reset hallucinated set; the set is emptied afterwards when sqp_iter==0."""
        plan = self._plan(use_grad=not use_model_without_derivatives)
        if use_model_without_derivatives:      # real data only (reference :221-226)
            hx = torch.empty(self.ns, self.g_ny, 0, self.in_dim_x, dtype=F64, device=self.torch_device)
            hy = torch.empty(self.ns, self.g_ny, 0, 1, dtype=F64, device=self.torch_device)
        else:
            self._check_in_place_edits()
            hx, hy = self.Hallcinated_X_train, self.Hallcinated_Y_train
        own = not use_model_without_derivatives
        self.model_i = HipGPModel(plan, hx, hy, self.batch_shape, self._ws_cache, dist_group=self.dist_group,
                                  all_observed=(own and self._hall_all_observed and self.dist_group is None),
                                  lineage=((self._hall_gen, int(hx.shape[2])) if own else None))
        self.likelihood = plan.hyper
        if sqp_iter == 0:
            self._reset_hallucinated()

    def train_forward_sampling_dynGP(self):
        """real + forward-sampling + hallucinated data (reference ``src/agent.py:283-329``)."""
        plan = self._plan(use_grad=True)
        hx = torch.concat([self.FS_X_train_batch, self.Hallcinated_X_train], dim=2)
        hy = torch.concat([self.FS_Y_train_batch, self.Hallcinated_Y_train], dim=2)
        self.model_i = HipGPModel(plan, hx, hy, self.batch_shape, self._ws_cache, dist_group=self.dist_group)
        self.likelihood = plan.hyper

    def sample_gp(self, x_input, base_samples=None):
        """Joint posterior draw + post-processing (reference ``src/agent.py:629-730``) in one ``gpmpc_joint_sample`` call
        (posterior + Cholesky root with the jitter chain; the eigendecomposition root for the whole batch when a chain
        fails every retry, as with the shipped car_residual jitter of 1e-20)."""
        ag = self.params["agent"]
        self.model_i_call = self.model_i(x_input)
        y = self.model_i_call._sample(base_samples, clip=(ag["Dyn_gp_min_data_dist"] < 0.0),
                                      beta=ag["Dyn_gp_beta"], var_zero_thr=ag["Dyn_gp_variance_is_zero"],
                                      want_root=self.debug_keep_root)
        if ag["Dyn_gp_min_data_dist"] >= 0.0:
            # overwrite by the closest observed training label when a test input is too close to it (:666-698),
            # then clip (:701-708).  Off in every shipped config; plain device tensor ops.
            y_train, x_train = self.model_i.train_targets, self.model_i.train_inputs[0]
            m = x_input.shape[2]
            dn = torch.linalg.vector_norm(x_input[:, :, None, :, :] - x_train[:, :, :, None, :], dim=-1)
            isnan = torch.any(torch.isnan(y_train), dim=3).unsqueeze(-1).expand(-1, -1, -1, m)
            dn = torch.where(isnan, torch.full_like(dn, float("inf")), dn)
            too_small = torch.any(dn <= ag["Dyn_gp_min_data_dist"], dim=2).unsqueeze(-1)
            idx = torch.argmin(dn, dim=2)                                          # (Ns, g_ny, m)
            closest = torch.gather(y_train, 2, idx.unsqueeze(-1).expand(-1, -1, -1, y_train.shape[-1]))
            y = torch.where(too_small, closest, y)
            assert not torch.any(torch.isnan(y))
            mean, sd = self.model_i_call.mean, ag["Dyn_gp_beta"] * torch.sqrt(self.model_i_call.variance)
            y = torch.min(torch.max(y, mean - sd), mean + sd)
        self.model_i_samples = y
        return y

    def get_batch_gp_sensitivities(self, xu_hat, sqp_iter):
        """GP value+gradient sample at the linearisation points (reference ``src/agent.py:566-627``): a joint draw for
        every dynamics sample, except that the leading samples can be pinned - first to the true dynamics
        (``true_dyn_as_sample``), then to the posterior mean (``mean_as_dyn_sample``).  When the pinned samples are ALL the
        samples nothing is drawn and the hallucinated set is left alone."""
        cfg = self.params["agent"]
        g_in = self.env_model.get_g_xu_hat(xu_hat).contiguous()
        pinned = [kind for kind, on in (("true_dyn", cfg["true_dyn_as_sample"]), ("mean", cfg["mean_as_dyn_sample"])) if on]
        nothing_drawn = (len(pinned) >= 1 and self.ns == 1) or (len(pinned) == 2 and self.ns == 2)
        if nothing_drawn:
            y = torch.zeros((self.ns, self.g_ny, self.params["optimizer"]["H"], self.in_dim_y), dtype=F64,
                            device=self.torch_device)
            self.model_i_call = self.model_i(g_in)
        else:
            y = self.sample_gp(g_in, base_samples=self.epistimic_random_vector[self.mpc_iter][sqp_iter])
        for slot, kind in enumerate(pinned):
            if kind == "true_dyn":
                truth = self.env_model.get_prior_data(g_in[slot, 0, :, :])
                y[slot, :, :, :] = truth[:, :, [0]] if self.in_dim_y == 1 else truth
            else:
                y[[slot], :, :, :] = self.model_i_call.mean[[slot], :, :, :]
        if not nothing_drawn:
            n_before = int(self._hall_X.shape[2])
            self.update_hallucinated_Dyn_dataset(g_in, y)
            # the points appended are the points the draw was made at (no min-distance filter dropped or masked any): the draw's
            # own X / S are the new rows of the factor (gpmpc_joint_sample_pending) - said here, checked by JointFactorCache.pending_ok
            if cfg["Dyn_gp_min_data_dist"] < 0.0 and int(self._hall_X.shape[2]) == n_before + int(g_in.shape[2]):
                self._ws_cache["joint_pending_points"] = (self._hall_gen, int(self._hall_X.shape[2]), self.model_i_call)
            else:
                self._ws_cache.pop("joint_pending_points", None)
        return y

    def dyn_fg_jacobians_device(self, xu_hat, sqp_iter):
        """``dyn_fg_jacobians`` without the device-to-host copies: the three float64 DEVICE tensors ``gp_val (Ns,nx,H,1)``,
        ``y_grad (Ns,nx,H,nx)``, ``u_grad (Ns,nx,H,nu)`` (what ``pack_p_lin`` and ``distributed.gather_jacobians`` consume)."""
        lib = _lib.load()
        xu_hat = xu_hat.to(device=self.torch_device, dtype=F64).contiguous()
        ns, nH = xu_hat.shape[0], xu_hat.shape[2]
        y = self.get_batch_gp_sensitivities(xu_hat, sqp_iter).contiguous()
        dev = _lib.require_hip_device(self.torch_device)
        # the three arrays are consecutive pieces of ONE buffer: a single device-to-host copy moves them all
        n1, n2, n3 = ns * self.nx * nH, ns * self.nx * nH * self.nx, ns * self.nx * nH * self.nu
        flat = torch.empty(n1 + n2 + n3, dtype=F64, device=dev)
        gp_val = flat[:n1].view(ns, self.nx, nH, 1)
        y_grad = flat[n1:n1 + n2].view(ns, self.nx, nH, self.nx)
        u_grad = flat[n1 + n2:].view(ns, self.nx, nH, self.nu)
        self._last_device_jacobians_flat = flat
        plan = self.model_i.plan
        _lib.check(lib.gpmpc_assemble_jacobians(plan.desc, self.env_desc(), ns, nH, _lib.dptr(xu_hat), _lib.dptr(y),
                                                _lib.dptr(gp_val), _lib.dptr(y_grad), _lib.dptr(u_grad),
                                                _lib.current_stream_ptr()), "gpmpc_assemble_jacobians")
        self._last_device_jacobians = (gp_val, y_grad, u_grad)
        return self._last_device_jacobians

    def dyn_fg_jacobians(self, xu_hat, sqp_iter):
        """Full-state value and Jacobians ``f + B_d g`` at the linearisation points (reference ``src/agent.py:532-564``).
        Returns numpy float64 ``gp_val (Ns,nx,H,1)``, ``y_grad (Ns,nx,H,nx)``, ``u_grad (Ns,nx,H,nu)``."""
        gp_val, y_grad, u_grad = self.dyn_fg_jacobians_device(xu_hat, sqp_iter)
        h = _lib.to_host(self._last_device_jacobians_flat)        # one pinned copy; the arrays are views of it
        n1, n2 = gp_val.numel(), y_grad.numel()
        out = (h[:n1].reshape(gp_val.shape), h[n1:n1 + n2].reshape(y_grad.shape), h[n1 + n2:].reshape(u_grad.shape))
        if not (np.isfinite(out[0]).all() and np.isfinite(out[1]).all() and np.isfinite(out[2]).all()):
            print("Nan/inf in y_sample")
        return out

    def pack_p_lin(self, x_h, u_h, xg, w, K=None):
        """Stage parameter vectors for the acados OCP in the layout of reference ``src/solver.py:98-131``, packed on
        the device from the Jacobians of the last ``dyn_fg_jacobians`` call.  Returns numpy (H, len)."""
        lib = _lib.load()
        gp_val, y_grad, u_grad = self._last_device_jacobians
        dev = gp_val.device
        H = gp_val.shape[2]
        # everything that comes from the host goes up in ONE copy: [x_h | u_h | xg | w]; the tightenings and the feedback gain
        # are constants of the Agent (uploaded once); the feedback product y_grad + u_grad K (reference src/solver.py:90)
        # happens inside the packing kernel
        x_np, u_np = np.asarray(x_h, dtype=np.float64).reshape(-1), np.asarray(u_h, dtype=np.float64).reshape(-1)
        xg_np, w_np = np.asarray(xg, dtype=np.float64).reshape(-1)[:H], np.asarray(w, dtype=np.float64).reshape(-1)[:H]
        up = torch.from_numpy(np.concatenate([x_np, u_np, xg_np, w_np])).to(dev)
        o1, o2, o3 = x_np.size, x_np.size + u_np.size, x_np.size + u_np.size + H
        const = self._ws_cache.get("plin_const")
        if const is None or const[0] != H:
            te = torch.as_tensor(np.stack(self.tilde_eps_list)[:H], dtype=F64).to(dev).contiguous()
            const = (H, te, {})
            self._ws_cache["plin_const"] = const
        te, Kc = const[1], const[2]
        K_d = None
        if K is not None:
            key = np.asarray(K, dtype=np.float64).tobytes()
            K_d = Kc.get(key)
            if K_d is None:
                K_d = torch.as_tensor(np.asarray(K), dtype=F64).to(dev).contiguous()
                Kc.clear()
                Kc[key] = K_d
        n = lib.gpmpc_plin_len(self.nx, self.nu, self.ns)
        p_lin = torch.empty(H, n, dtype=F64, device=dev)
        _lib.check(lib.gpmpc_pack_plin_fb(self.nx, self.nu, self.ns, H, _lib.dptr(y_grad), _lib.dptr(u_grad),
                                          _lib.dptr(gp_val), _lib.dptr(up[:o1]), _lib.dptr(up[o1:o2]), _lib.dptr(up[o2:o3]),
                                          _lib.dptr(up[o3:]), _lib.dptr(te), _lib.dptr(K_d), _lib.dptr(p_lin),
                                          _lib.current_stream_ptr()), "gpmpc_pack_plin_fb")
        return _lib.to_host(p_lin)

    def sqp_linearisation(self, x_h, u_h, sqp_iter, xg, w, K=None, u_nominal=None, train=True):
        """One SQP iteration's GP side as the reference's solver consumes it (``src/solver.py:84-131``): ``train_hallucinated_dynGP``
        -> batch_x_hat -> joint draw -> Jacobians -> stage parameter vectors ``p_lin`` (numpy ``(H, len)``, acados layout) with
        the host <-> device traffic the loop actually needs: ONE upload of ``[x_h | u_h | u_nominal | xg | w]`` (pinned staging),
        one launch for ``batch_x_hat`` (``gpmpc_build_x_hat``), the draw, ONE launch for the Jacobians AND ``p_lin``
        (``gpmpc_assemble_jacobians_plin``), ONE download of ``p_lin``.  The three Jacobian arrays stay on the device
        (``self._last_device_jacobians``; ``dyn_fg_jacobians`` remains the reference-shaped call that brings them to the host).

        ``u_h``: ``(H, nu)`` shared by the samples (reference ``get_batch_x_hat``) or ``(H, Ns, nu)`` per sample (the feedback
        form, ``get_batch_x_hat_u_diff``); ``u_nominal`` ``(H, nu)``: the inputs of the stage tail (default: ``u_h`` when it is
        shared).  ``K``: feedback gain folded into ``A_i = y_grad + u_grad K`` (``src/solver.py:90``)."""
        lib = _lib.load()
        dev = _lib.require_hip_device(self.torch_device)
        if train:
            self.train_hallucinated_dynGP(sqp_iter)
        H, ns, nx, nu = int(self.params["optimizer"]["H"]), self.ns, self.nx, self.nu
        x_np = np.asarray(x_h, dtype=np.float64).reshape(-1)
        u_np = np.asarray(u_h, dtype=np.float64)
        per_sample = u_np.ndim == 3
        un_np = (u_np if not per_sample else np.zeros((H, nu))) if u_nominal is None else np.asarray(u_nominal, dtype=np.float64)
        u_np, un_np = u_np.reshape(-1), un_np.reshape(-1)[: H * nu]
        xg_np, w_np = np.asarray(xg, dtype=np.float64).reshape(-1)[:H], np.asarray(w, dtype=np.float64).reshape(-1)[:H]
        sizes = (x_np.size, u_np.size, un_np.size, H, H)
        offs = np.concatenate([[0], np.cumsum(sizes)])
        st = self._ws_cache.get("sqp_staging")
        if st is None or st[0].numel() != offs[-1]:
            st = (torch.empty(int(offs[-1]), dtype=F64, pin_memory=True), torch.empty(int(offs[-1]), dtype=F64, device=dev),
                  torch.empty(ns, nx, H, nx + nu, dtype=F64, device=dev))
            self._ws_cache["sqp_staging"] = st
        hst, up, xu = st
        hn = hst.numpy()
        for a, o in zip((x_np, u_np, un_np, xg_np, w_np), offs[:-1]):
            hn[o:o + a.size] = a
        up.copy_(hst, non_blocking=True)
        seg = [up[offs[i]:offs[i + 1]] for i in range(5)]
        stream = _lib.current_stream_ptr()
        _lib.check(lib.gpmpc_build_x_hat(nx, nu, ns, H, _lib.dptr(seg[0]), _lib.dptr(seg[1]), int(per_sample), _lib.dptr(xu), stream),
                   "gpmpc_build_x_hat")
        y = self.get_batch_gp_sensitivities(xu, sqp_iter).contiguous()
        const = self._ws_cache.get("plin_const")
        if const is None or const[0] != H:
            te = torch.as_tensor(np.stack(self.tilde_eps_list)[:H], dtype=F64).to(dev).contiguous()
            const = (H, te, {})
            self._ws_cache["plin_const"] = const
        te, Kc = const[1], const[2]
        K_d = None
        if K is not None:
            key = np.asarray(K, dtype=np.float64).tobytes()
            K_d = Kc.get(key)
            if K_d is None:
                K_d = torch.as_tensor(np.asarray(K), dtype=F64).to(dev).contiguous()
                Kc.clear()
                Kc[key] = K_d
        n1, n2, n3 = ns * nx * H, ns * nx * H * nx, ns * nx * H * nu
        n = lib.gpmpc_plin_len(nx, nu, ns)
        out = self._ws_cache.get("sqp_out")
        if out is None or out[0].numel() != n1 + n2 + n3 or out[1].shape != (H, n):
            out = (torch.empty(n1 + n2 + n3, dtype=F64, device=dev), torch.empty(H, n, dtype=F64, device=dev))
            self._ws_cache["sqp_out"] = out
        flat, p_lin = out
        gp_val = flat[:n1].view(ns, nx, H, 1)
        y_grad = flat[n1:n1 + n2].view(ns, nx, H, nx)
        u_grad = flat[n1 + n2:].view(ns, nx, H, nu)
        plan = self.model_i.plan
        _lib.check(lib.gpmpc_assemble_jacobians_plin(plan.desc, self.env_desc(), ns, H, _lib.dptr(xu), _lib.dptr(y), _lib.dptr(gp_val),
                                                     _lib.dptr(y_grad), _lib.dptr(u_grad), _lib.dptr(seg[2]), _lib.dptr(seg[3]),
                                                     _lib.dptr(seg[4]), _lib.dptr(te), _lib.dptr(K_d), _lib.dptr(p_lin), stream),
                   "gpmpc_assemble_jacobians_plin")
        self._last_device_jacobians_flat = flat
        self._last_device_jacobians = (gp_val, y_grad, u_grad)
        return _lib.to_host(p_lin)

    # ---------------------------------------------------------------------------------------------------------
    # forward sampling with rejection (reference src/agent.py:331-443)
    # ---------------------------------------------------------------------------------------------------------
    def prepare_dynamics_set(self, X_soln, U_soln, X_kp1, *, base_samples=None, rng=None, fused=None):
        """Propagate every sampled dynamics along the shifted solution, reject the samples that leave the
        ``ci_list`` tube, replace their hallucinated data by randomly chosen survivors' (reference ``src/agent.py:331-443``).

        The first propagation step draws from the model the SQP loop left behind (``self.model_i``, as the reference
        does); the remaining ``H-2`` steps - each conditioning on real + hallucinated + the value-only draws so far,
        ``train_forward_sampling_dynGP`` in the reference - run as ONE ``gpmpc_rollout_seeded`` launch when the
        conditioning set fits the rollout kernel (``fused=None``: decide by size; the per-step path through
        ``gpmpc_joint_sample`` otherwise, or with ``fused=False``).  The tube test and the survivor bookkeeping are
        device tensor operations on the rollout's ``X_traj``.

        Keyword-only reproducibility hooks (the reference has neither): ``base_samples`` - one ``(Ns, g_ny, 1, T)``
        tensor per propagation step instead of the internal ``randn`` draw; ``rng`` - a ``numpy.random.RandomState``
        for the survivor choice instead of the global ``np.random``.  ``self.rejection_trace`` keeps the survivor mask
        after every step."""
        Ns, nx = self.ns, self.nx
        dev = _lib.require_hip_device(self.torch_device)
        cfg_t = self.params["agent"]["tight"]
        first_bound = (cfg_t["dyn_eps"] + cfg_t["w_bound"]) * np.sqrt(self.params["optimizer"]["terminal_tightening"]["P"][1][1])
        rng = np.random if rng is None else rng
        T = self.in_dim_y

        def inside(err, bound):                      # 1 where every state component is strictly inside the tube, else 0
            return (torch.abs(err) - bound < 0).all(dim=1).to(torch.int64)

        def gp_input_batch(states, u_row):           # (Ns or 1, nx) states + one input row -> (Ns, nx, 1, nx + nu)
            xu = torch.cat([states.expand(Ns, nx), u_row.reshape(1, -1).expand(Ns, -1)], dim=-1)
            return xu[:, None, None, :].expand(Ns, nx, 1, nx + self.nu).contiguous()

        plan_x = torch.as_tensor(X_soln, dtype=F64).reshape(X_soln.shape[0], Ns, nx).to(dev)     # the solver's trajectories
        plan_u = torch.as_tensor(U_soln, dtype=F64).to(dev)
        x_real = torch.as_tensor(X_kp1, dtype=F64).transpose(0, 1).to(dev)                          # (1, nx): the state reached
        steps = plan_x.shape[0] - 2                                                                # propagation steps 1 .. steps
        self.FS_X_train_batch = torch.empty(Ns, self.g_ny, 0, self.in_dim_x, dtype=F64, device=dev)
        self.FS_Y_train_batch = torch.empty(Ns, self.g_ny, 0, T, dtype=F64, device=dev)

        alive = inside(plan_x[1] - x_real, first_bound)
        self.rejection_trace = [alive.clone()]

        n_hall = int(self.Hallcinated_X_train.shape[2])
        from .rollout import seeds_fit
        has_nan = torch.isnan(self.Hallcinated_Y_train).any().to(torch.int32).reshape(1)
        if self.dist_group is not None:            # every rank must take the same path (the per-step one has collectives)
            import torch.distributed as dist
            dist.all_reduce(has_nan, op=dist.ReduceOp.MAX, group=self.dist_group)
        _lib.host_wait(has_nan)
        fusable = (steps >= 3 and T == 1 + self.in_dim_x and seeds_fit(self, n_hall, 1, steps - 1, T, 1)
                   and not bool(has_nan.item()))
        if fused is None:
            fused = fusable
        elif fused and not fusable:
            raise _lib.GpmpcError("prepare_dynamics_set: the conditioning set does not fit the fused rollout "
                                  f"({n_hall} hallucinated points x {T} tasks + {steps - 1} value-only draws > 256 label slots)")

        states = x_real
        for step in range(1, steps + 1):
            xu = gp_input_batch(states, plan_u[step])
            g_in = self.env_model.get_g_xu_hat(xu).contiguous()
            draw = self.model_i(g_in).sample(None if base_samples is None else base_samples[step - 1].to(dev))
            residual = draw[:, :, 0, 0]                                                   # value component of every output
            states = self.env_model.known_dyn(xu).reshape(Ns, nx) + residual @ self.env_model.B_d.t()
            alive = alive * inside(plan_x[step + 1] - states, self.ci_list[step])
            self.rejection_trace.append(alive.clone())
            if step == steps:
                break
            # the draw becomes a value-only training point of the forward-sampling GP (gradient labels unobserved)
            label = draw.clone()
            label[:, :, :, 1:] = float("nan")
            self.FS_X_train_batch = torch.cat([self.FS_X_train_batch, g_in], dim=2)
            self.FS_Y_train_batch = torch.cat([self.FS_Y_train_batch, label], dim=2)
            if fused:
                done = self._propagate_fused(plan_x, plan_u, states, alive, base_samples, steps, dev)
                if done is not None:
                    alive = done
                    break
                # a chain's T x T posterior root failed every jitter retry: the fused kernel has no eigendecomposition
                # root (the failing chains carry NaN), the per-step path (gpmpc_joint_sample) does - finish there
                fused = False
            self.train_forward_sampling_dynGP()

        # rejected samples inherit the hallucinated data of randomly chosen survivors (X and Y drawn separately, as the
        # reference does: two calls of the generator)
        if self.dist_group is not None:                                            # samples sharded over ranks
            from .distributed import replace_rejected_samples
            self.Hallcinated_X_train, self.Hallcinated_Y_train = replace_rejected_samples(
                self.Hallcinated_X_train, self.Hallcinated_Y_train, alive, self.ns_global, rng, self.dist_group)
        elif (_lib.host_wait(alive), int(alive.sum().item()))[1] > 0:
            rejected = alive == 0
            n_rejected = int(rejected.sum().item())
            survivors = torch.nonzero(alive > 0).reshape(-1).cpu().numpy()
            self.Hallcinated_X_train[rejected] = self.Hallcinated_X_train[rng.choice(survivors, n_rejected).tolist()]
            self.Hallcinated_Y_train[rejected] = self.Hallcinated_Y_train[rng.choice(survivors, n_rejected).tolist()]
            self.invalidate_factor_cache()                        # edited in place
        self.train_hallucinated_dynGP(sqp_iter=self.params["optimizer"]["SEMPC"]["max_sqp_iter"])

    def _propagate_fused(self, X_soln, U_soln, x_next1, samples_left, base_samples, n_last, dev):
        """Propagation steps i = 2 .. n_last of ``prepare_dynamics_set`` in one ``gpmpc_rollout_seeded`` launch: the
        chains condition on the hallucinated points (all tasks), on step 1's value-only draw, and append their own
        value-only draws (``hall_tasks = 1``, reference ``src/agent.py:399-405``); inputs ``U_soln[i]`` without feedback
        (``:409-415``), no clip (the reference calls ``.sample()`` directly)."""
        from .rollout import rollout_device
        Hf, T = n_last - 1, self.in_dim_y
        if base_samples is None:
            z = torch.randn(Hf, self.ns, self.g_ny, 1, T, dtype=F64, device=dev)
        else:
            z = torch.stack([base_samples[i - 1].to(dev) for i in range(2, n_last + 1)]).to(F64)
        z = z.reshape(Hf, -1).contiguous()
        seeds = (self.Hallcinated_X_train, self.Hallcinated_Y_train) if self.Hallcinated_X_train.shape[2] else None
        vseeds = (self.FS_X_train_batch, torch.nan_to_num(self.FS_Y_train_batch, nan=0.0))
        res = rollout_device(self, U_soln[2:2 + Hf].cpu().numpy(), z.reshape(-1), z.shape[1], H=Hf, mode=_lib.MODE_RECONDITIONED,
                             use_model_without_derivatives=False, use_feedback=False, x0=x_next1, hall_tasks=1,
                             var_zero_thr=-1.0, beta=float("inf"), seeds=seeds, value_seeds=vseeds)
        from .gp_model import NotPSDError, _or_reduce
        bits = _or_reduce(res.info, self.dist_group)                # OR over samples (and ranks): not the max of the words
        if bits & _lib.INFO_TRAIN_CHOL_FAIL:
            raise NotPSDError("Cholesky of the training covariance failed during forward sampling")
        if bits & (_lib.INFO_ROOT_FAIL | _lib.INFO_NEG_1x1):
            return None                                             # the caller continues on the per-step path
        for t in range(Hf):
            i = t + 2
            diff = X_soln[i + 1, :, :] - res.X_traj[:, :, t + 1]
            samples_left = samples_left * torch.prod(torch.abs(diff) - self.ci_list[i] < 0, dim=1)
            self.rejection_trace.append(samples_left.clone())
        if Hf > 1:                                                  # the draws of steps 2 .. n_last-1 are training points too
            Xi = res.Xi[:, None, :Hf - 1, :].expand(-1, self.g_ny, -1, -1)
            Yv = res.Y[:, :, :Hf - 1, :].clone()
            Yv[..., 1:] = float("nan")
            self.FS_X_train_batch = torch.cat([self.FS_X_train_batch, Xi], dim=2)
            self.FS_Y_train_batch = torch.cat([self.FS_Y_train_batch, Yv], dim=2)
        return samples_left
