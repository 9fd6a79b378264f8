"""Whole-horizon rollouts of the sampled dynamics in one kernel launch (``gpmpc_rollout``).

Drivers with the loop structure of the reference's harnesses:

* ``forward_sampling_rollout``  - reference ``benchmarking/simulate_forward_sampling_car.py:108-138``
  (mode "I" when ``env.use_model_without_derivatives`` is True as shipped, mode "R" otherwise);
* ``true_reachable_set_rollout`` - the sampling loop of reference ``benchmarking/simulate_true_reachable_set.py:
  179-258`` for the residual environments (open-loop inputs, internally drawn base samples, variance-is-zero
  replacement), expressed on the same kernel.

Both return ``X_traj (Ns, nx, H+1)`` float64 - the array the reference pickles (``data_X_traj_<idx>.pkl``).

``forward_sampling_stepwise`` is the same harness loop driven through the ``Agent`` methods one step at a time (one
``gpmpc_joint_sample`` + ``gpmpc_assemble_jacobians`` launch per step).  ``forward_sampling_rollout`` routes to it when
a feature the fused kernel does not implement is switched on (``Dyn_gp_min_data_dist >= 0``: label overwrite and
dataset filtering, ``true_dyn_as_sample`` / ``mean_as_dyn_sample`` short cuts); every shipped YAML takes the fused path.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _lib
from .agent import Agent

F64 = torch.float64


class RolloutResult:
    def __init__(self, X_traj, Y, Xi, info):
        self.X_traj, self.Y, self.Xi, self.info = X_traj, Y, Xi, info


def pin_rollout_kernel_like(agent: Agent, *, H: int, Ns_launch: int, mode: int = _lib.MODE_RECONDITIONED,
                            use_model_without_derivatives: bool = False, hall_tasks: Optional[int] = None) -> int:
    """Pin ``gpmpc_rollout``'s kernel to the one a launch of ``Ns_launch`` samples of this shape would take, and return it.

    The dispatcher picks by launch size, and the kernels sum in different orders: a sample's trajectory is bit-identical
    between two launches only if both ran the same kernel.  A sample-sharded run that must reproduce the single-GPU run of the
    same samples bit for bit calls this with the GLOBAL sample count before its shard-sized launches (at the price of the
    kernel that is best for the shard size).  ``hall_tasks``: the appended labels' task count of the launches to follow (a
    value-only rollout, ``hall_tasks = 1``, may take another kernel than a ``T``-task one).  The pin is process-global and stays
    until it is released: prefer ``with pinned_rollout_kernel_like(...)``, which restores what was pinned before."""
    lib = _lib.load()
    lib.gpmpc_rollout_pin_kernel(-1)
    plan = agent._plan(use_grad=not use_model_without_derivatives)
    k = lib.gpmpc_rollout_kernel_for(plan.desc, agent.env_desc(), mode, plan.hyper.T if hall_tasks is None else int(hall_tasks),
                                     int(Ns_launch), int(H))
    lib.gpmpc_rollout_pin_kernel(k)
    global _PINNED_KERNEL
    _PINNED_KERNEL = k
    return k


_PINNED_KERNEL = -1          # what this module last pinned (gpmpc_rollout_pin_kernel has no getter for the pin itself)


class pinned_rollout_kernel_like:
    """``with pinned_rollout_kernel_like(agent, H=.., Ns_launch=..) as kernel: ...`` - the pin of ``pin_rollout_kernel_like`` for
    the body only; the pin that was in force before (usually none) is restored on exit, also when the body raises."""

    def __init__(self, agent: Agent, **kw):
        self.agent, self.kw = agent, kw

    def __enter__(self):
        global _PINNED_KERNEL
        self.prev = _PINNED_KERNEL
        return pin_rollout_kernel_like(self.agent, **self.kw)

    def __exit__(self, *exc):
        global _PINNED_KERNEL
        _lib.load().gpmpc_rollout_pin_kernel(self.prev)
        _PINNED_KERNEL = self.prev
        return False


def rollout_device(agent: Agent, u_ff, z: torch.Tensor, z_step_stride: int, *, H: int, mode: int,
                   use_model_without_derivatives: bool, use_feedback: Optional[bool] = None,
                   x0=None, hall_tasks: Optional[int] = None, var_zero_thr: Optional[float] = None,
                   beta: Optional[float] = None, want_samples: bool = True, sample_slice=None,
                   seeds=None, value_seeds=None, state: Optional["RolloutState"] = None,
                   resume: bool = False) -> RolloutResult:
    """Launch ``gpmpc_rollout`` for the samples of ``agent`` (or a contiguous ``sample_slice`` of them) and return
    device tensors.  ``z`` element (t, s, o, b) is read at ``z[t*z_step_stride + ((s*g_ny)+o)*T + b]``.

    ``seeds = (X (Ns, g_ny, n, D), Y (Ns, g_ny, n, T))``: points every chain conditions on before step 0 (all tasks
    observed); ``value_seeds``: the same with only the first ``hall_tasks`` label entries observed; ``state``: a
    ``RolloutState`` that receives the chains' factor (and, with ``resume``, provides it) - these go through
    ``gpmpc_rollout_seeded``."""
    lib = _lib.load()
    dev = _lib.require_hip_device(agent.torch_device)
    p = agent.params
    plan = agent._plan(use_grad=not use_model_without_derivatives)
    T = plan.hyper.T
    lo, hi = (0, agent.ns) if sample_slice is None else sample_slice
    Ns = hi - lo
    nx, g_ny, D = agent.nx, agent.g_ny, agent.in_dim_x
    u_ff_d = torch.as_tensor(np.asarray(u_ff), dtype=F64).reshape(H, agent.nu).to(dev).contiguous()
    if x0 is None:
        x0_d = torch.as_tensor(np.asarray(p["env"]["start"], dtype=np.float64)[:nx]).to(dev).contiguous()
        per_sample = 0
    else:
        x0_d = torch.as_tensor(x0, dtype=F64).to(dev).contiguous()
        per_sample = int(x0_d.dim() == 2)
        if per_sample:
            x0_d = x0_d[lo:hi].contiguous()
    if hall_tasks is None:
        hall_tasks = T
    if var_zero_thr is None:
        var_zero_thr = p["agent"]["Dyn_gp_variance_is_zero"]
    if beta is None:
        beta = p["agent"]["Dyn_gp_beta"]
    X_traj = torch.empty(Ns, nx, H + 1, dtype=F64, device=dev)
    Y = torch.empty(Ns, g_ny, H, T, dtype=F64, device=dev) if want_samples else None
    Xi = torch.empty(Ns, H, D, dtype=F64, device=dev) if want_samples else None
    info = torch.zeros(Ns, dtype=torch.int32, device=dev)
    # seed points lengthen the chains' factor: n_h0*T + hall_tasks*(n_v0 + H-1) label slots, which leaves LDS sooner
    n_seed_pts = int(seeds[0].shape[2]) if seeds is not None else 0
    n_vseed_pts = int(value_seeds[0].shape[2]) if value_seeds is not None else 0
    if state is None and (n_seed_pts or n_vseed_pts):
        ws_bytes = lib.gpmpc_rollout_seeded_workspace_bytes(plan.desc, mode, hall_tasks, Ns, H, n_seed_pts, n_vseed_pts)
    else:
        ws_bytes = lib.gpmpc_rollout_workspace_bytes(plan.desc, mode, hall_tasks, Ns, H)
    if ws_bytes == 0:
        _lib.check(-4, "gpmpc_rollout_workspace_bytes")
    ws = agent._ws_cache.get("rollout")
    if ws is None or ws.numel() * 8 < ws_bytes:
        ws = torch.empty((ws_bytes + 7) // 8, dtype=F64, device=dev)
        agent._ws_cache["rollout"] = ws
    assert z.is_cuda and z.dtype == F64
    z_ptr = z.data_ptr() + 8 * lo * g_ny * T
    common = (plan.desc, agent.env_desc(use_feedback), _lib.dptr(plan.buf), _lib.dptr(plan.X_r),
              mode, hall_tasks, float(var_zero_thr), float(beta), Ns, H,
              _lib.dptr(x0_d), per_sample, _lib.dptr(u_ff_d), z_ptr, int(z_step_stride),
              _lib.dptr(X_traj), _lib.dptr(Y), _lib.dptr(Xi), _lib.dptr(info),
              _lib.dptr(ws), ws.numel() * 8, _lib.current_stream_ptr())
    if seeds is None and value_seeds is None and state is None:
        _lib.check(lib.gpmpc_rollout(*common), "gpmpc_rollout")
        return RolloutResult(X_traj, Y, Xi, info)

    def prep(pair, last):
        if pair is None:
            return None, None, 0
        X_, Y_ = (t.to(device=dev, dtype=F64)[lo:hi].contiguous() for t in pair)
        if X_.shape[:2] != (Ns, g_ny) or Y_.shape[:3] != X_.shape[:3] or X_.shape[3] != D or Y_.shape[3] != last:
            raise ValueError("seed points must be (Ns, g_ny, n, D) / (Ns, g_ny, n, T)")
        if bool(torch.isnan(Y_[..., :last if pair is seeds else hall_tasks]).any()):
            raise ValueError("seed labels must be observed (no NaN): NaN-masked points need the joint kernel")
        return X_, Y_, int(X_.shape[2])
    Xs, Ys, n0 = prep(seeds, T)
    Xv, Yv, nv = prep(value_seeds, T)
    if sample_slice is not None and state is not None:
        raise ValueError("a factor state covers all samples of the agent")
    rc = lib.gpmpc_rollout_seeded(*common, _lib.dptr(Xs), _lib.dptr(Ys), n0, _lib.dptr(Xv), _lib.dptr(Yv), nv,
                                  _lib.dptr(state.buf) if state is not None else None,
                                  state.slots if state is not None else 0, state.points if state is not None else 0,
                                  int(bool(resume)))
    _lib.check(rc, "gpmpc_rollout_seeded")
    return RolloutResult(X_traj, Y, Xi, info)


class RolloutState:
    """The chains' factor after a rollout (``gpmpc_rollout_seeded``: per sample the point list and per chain
    ``L_hr^T``, ``L_hh``, ``w``, ``1/diag``), so that a later rollout continues without re-factorising.
    ``slots`` / ``points``: label-slot and point capacity per chain (slots <= 256)."""

    def __init__(self, agent: Agent, slots: int, points: int, use_model_without_derivatives: bool = False):
        lib = _lib.load()
        dev = _lib.require_hip_device(agent.torch_device)
        plan = agent._plan(use_grad=not use_model_without_derivatives)
        self.slots, self.points, self.Ns = int(slots), int(points), agent.ns
        nbytes = lib.gpmpc_rollout_state_bytes(plan.desc, agent.ns, self.slots, self.points)
        if nbytes == 0:
            raise _lib.GpmpcError("gpmpc_rollout_state_bytes: unsupported size (slots must be 1..256)")
        self.buf = torch.zeros(nbytes // 8, dtype=F64, device=dev)
        self._stride = (nbytes // 8) // agent.ns if agent.ns else 0

    def counts(self):
        """(seed points, appended points) per sample, read back from the device (uniform over samples)."""
        _lib.host_wait(self.buf)
        return int(self.buf[0].item()), int(self.buf[1].item())


MAX_ROLLOUT_SLOTS = 256          # generic kernel: label slots per chain (include/gpmpc_hip.h)


def seeds_fit(agent: Agent, n_seed_points: int, n_value_points: int, H: int, T: int, hall_tasks: int) -> bool:
    return n_seed_points * T + hall_tasks * (n_value_points + H - 1) <= MAX_ROLLOUT_SLOTS


def fused_rollout_supported(agent: Agent) -> bool:
    """True when ``gpmpc_rollout`` implements everything the configuration asks of ``sample_gp`` /
    ``update_hallucinated_Dyn_dataset`` (reference ``src/agent.py:164-202, 566-730``)."""
    ag = agent.params["agent"]
    return (ag["Dyn_gp_min_data_dist"] < 0.0 and not ag.get("true_dyn_as_sample", False)
            and not ag.get("mean_as_dyn_sample", False))


def forward_sampling_stepwise(agent: Agent, u_ff, x0=None, return_samples: bool = False):
    """The forward-sampling harness (reference ``benchmarking/simulate_forward_sampling_car.py:108-138``) through the
    ``Agent`` call surface, one step per iteration: re-train on the points appended so far, draw at the current
    state/input, hand the sampled next state over.  Slower than the fused kernel (per-step launches and two host
    round trips, like the reference) but covers every ``sample_gp`` option."""
    p = agent.params
    u_ff = np.asarray(u_ff, dtype=np.float64)
    H_traj = u_ff.shape[0]
    ns, nx = agent.ns, agent.nx
    nograd = bool(p["env"]["use_model_without_derivatives"])
    start = np.asarray(p["env"]["start"] if x0 is None else x0, dtype=np.float64)
    states = np.tile(start[:nx].reshape(1, nx), (1, ns))                         # (1, Ns*nx): the H=1 "x_h" row
    K = np.asarray(p["optimizer"]["terminal_tightening"]["K"], dtype=np.float64)
    x_goal = np.asarray(p["env"]["goal_state"], dtype=np.float64)
    X_traj = np.empty((ns, nx, H_traj + 1))
    draws = []
    for t in range(H_traj):
        agent.train_hallucinated_dynGP(1, use_model_without_derivatives=nograd)
        agent.mpc_iteration(t)
        u_t = u_ff[t].reshape(1, -1)
        if p["agent"]["feedback"]["use"]:
            u_fb = (states.reshape(1, ns, nx) - x_goal) @ K.T + u_t[:, None, :]     # u_ff + K (x - x_goal)
            xu = agent.get_batch_x_hat_u_diff(states, u_fb)
        else:
            xu = agent.get_batch_x_hat(states, u_t)
        gp_val, _, _ = agent.dyn_fg_jacobians(xu, 1)
        if return_samples:
            draws.append(agent.model_i_samples.detach().cpu().numpy())
        X_traj[:, :, t] = states.reshape(ns, nx)
        states = gp_val[:, :, 0, 0].reshape(1, -1)
    X_traj[:, :, H_traj] = states.reshape(ns, nx)
    if return_samples:
        return X_traj, np.concatenate(draws, axis=2)
    return X_traj


def forward_sampling_rollout(agent: Agent, u_ff, x0=None, return_samples: bool = False, check: bool = True):
    """The reference forward-sampling loop, one launch.

    ``agent.epistimic_random_vector`` must have the layout the reference script relies on: ``optimizer.H == 1``,
    ``num_MPC_itrs >= H_traj``, ``max_sqp_iter >= 2``; step ``t`` uses the slab ``[t][1]``.
    Side effect (as in the reference): ``agent.Hallcinated_{X,Y}_train`` end up holding the H appended points.
    """
    p = agent.params
    u_ff = np.asarray(u_ff, dtype=np.float64)
    H = u_ff.shape[0]
    erv = agent.epistimic_random_vector
    if not fused_rollout_supported(agent):
        return forward_sampling_stepwise(agent, u_ff, x0=x0, return_samples=return_samples)
    nograd_ = bool(p["env"]["use_model_without_derivatives"])
    seeds = None
    if agent.Hallcinated_X_train.shape[2] != 0 and not nograd_:
        # The reference loop's train_hallucinated_dynGP(1) never resets: a second call on the same agent (or a call after
        # closed-loop iterations) conditions on the points already there.  They seed the chains' factor
        # (gpmpc_rollout_seeded) when they fit and are fully observed; otherwise the per-step harness conditions on them.
        n0 = agent.Hallcinated_X_train.shape[2]
        if (not seeds_fit(agent, n0, 0, H, 1 + agent.in_dim_x, 1 + agent.in_dim_x)
                or bool(torch.isnan(agent.Hallcinated_Y_train).any())):
            return forward_sampling_stepwise(agent, u_ff, x0=x0, return_samples=return_samples)
        seeds = (agent.Hallcinated_X_train, agent.Hallcinated_Y_train)
    if p["optimizer"]["H"] != 1 or erv.shape[0] < H or erv.shape[1] < 2:
        raise ValueError("forward sampling needs optimizer.H == 1, num_MPC_itrs >= H_traj and max_sqp_iter >= 2 "
                         "(reference simulate_forward_sampling_car.py indexes epistimic_random_vector[H_idx][1])")
    nograd = bool(p["env"]["use_model_without_derivatives"])
    mode = _lib.MODE_INDEPENDENT if nograd else _lib.MODE_RECONDITIONED
    T = 1 if nograd else 1 + agent.in_dim_x
    erv = erv.to(device=agent.torch_device, dtype=F64).contiguous()
    n_itrs = erv.shape[1]
    per_slab = agent.ns * agent.g_ny * 1 * T
    z = erv.reshape(-1)[per_slab:]                       # starts at [0][1]
    res = rollout_device(agent, u_ff, z, n_itrs * per_slab, H=H, mode=mode,
                         use_model_without_derivatives=nograd, x0=x0, seeds=seeds)
    if check:
        _raise_on_info(res.info)
    # dataset side effect of the reference loop (appended at every step, also in mode I where it is never used)
    agent.Hallcinated_X_train = torch.cat(
        [agent.Hallcinated_X_train, res.Xi.unsqueeze(1).expand(-1, agent.g_ny, -1, -1)], dim=2)
    agent.Hallcinated_Y_train = torch.cat([agent.Hallcinated_Y_train, res.Y], dim=2)
    agent.model_i_samples = res.Y[:, :, [H - 1], :]
    _lib.host_wait(res.X_traj)
    X = res.X_traj.cpu().numpy()
    if return_samples:
        return X, res.Y.cpu().numpy()
    return X


def true_reachable_set_rollout(agent: Agent, u_seq, z: Optional[torch.Tensor] = None, x0=None,
                               return_samples: bool = False):
    """Sequential re-conditioned rollout with open-loop inputs and internally drawn base samples
    (reference ``simulate_true_reachable_set.py:179-258``: ``.sample()`` without base samples, variance-is-zero
    replacement, beta clip, conditioning on the sampled values, state hand-over)."""
    u_seq = np.asarray(u_seq, dtype=np.float64)
    H = u_seq.shape[0]
    T = 1 + agent.in_dim_x
    dev = _lib.require_hip_device(agent.torch_device)
    if z is None:
        z = torch.randn(H, agent.ns, agent.g_ny, T, dtype=F64, device=dev)
    z = z.to(device=dev, dtype=F64).contiguous()
    res = rollout_device(agent, u_seq, z.reshape(-1), agent.ns * agent.g_ny * T, H=H,
                         mode=_lib.MODE_RECONDITIONED, use_model_without_derivatives=False, use_feedback=False, x0=x0)
    _raise_on_info(res.info)
    X = res.X_traj.cpu().numpy()
    if return_samples:
        return X, res.Y.cpu().numpy()
    return X


def _raise_on_info(info: torch.Tensor):
    from .gp_model import NotPSDError, _or_reduce
    bits = _or_reduce(info)
    if bits & _lib.INFO_TRAIN_CHOL_FAIL:
        raise NotPSDError("rollout: Cholesky of a chain's training covariance failed")
    if bits & _lib.INFO_ROOT_FAIL:
        raise NotPSDError("rollout: posterior root failed after 3 jitter retries (eigh fallback is only "
                          "available through Agent.sample_gp)")


class RolloutRunner:
    """Preallocated, allocation-free repeated launches of one rollout configuration (used by bench.py and by
    ``sampling_gpmpc_amd.distributed``): every buffer lives in HBM before the first launch."""

    def __init__(self, agent: Agent, u_ff, z: torch.Tensor, z_step_stride: int, H: int, mode: int,
                 use_model_without_derivatives: bool, x0=None, want_samples: bool = False,
                 hall_tasks: Optional[int] = None):
        self.lib = _lib.load()
        dev = _lib.require_hip_device(agent.torch_device)
        p = agent.params
        self.agent, self.H, self.mode = agent, H, mode
        self.plan = agent._plan(use_grad=not use_model_without_derivatives)
        T = self.plan.hyper.T
        # labels observed at an appended point: all T (the sample_gp path) or the value only (src/agent.py:402)
        ht = T if hall_tasks is None else int(hall_tasks)
        self.Ns = agent.ns
        self.u_ff = torch.as_tensor(np.asarray(u_ff), dtype=F64).reshape(H, agent.nu).to(dev).contiguous()
        x0 = p["env"]["start"] if x0 is None else x0
        self.x0 = torch.as_tensor(np.asarray(x0, dtype=np.float64)[: agent.nx]).to(dev).contiguous()
        self.z = z.to(device=dev, dtype=F64).contiguous()
        self.z_step_stride = int(z_step_stride)
        self.X_traj = torch.empty(self.Ns, agent.nx, H + 1, dtype=F64, device=dev)
        self.Y = torch.empty(self.Ns, agent.g_ny, H, T, dtype=F64, device=dev) if want_samples else None
        self.Xi = torch.empty(self.Ns, H, agent.in_dim_x, dtype=F64, device=dev) if want_samples else None
        self.info = torch.zeros(self.Ns, dtype=torch.int32, device=dev)
        nbytes = self.lib.gpmpc_rollout_workspace_bytes(self.plan.desc, mode, ht, self.Ns, H)
        self.ws = torch.empty((nbytes + 7) // 8, dtype=F64, device=dev)
        self.env = agent.env_desc()
        self.beta = float(p["agent"]["Dyn_gp_beta"])
        self.var_zero = float(p["agent"]["Dyn_gp_variance_is_zero"])
        self._args = (self.plan.desc, self.env, _lib.dptr(self.plan.buf), _lib.dptr(self.plan.X_r), mode, ht,
                      self.var_zero, self.beta, self.Ns, H, _lib.dptr(self.x0), 0, _lib.dptr(self.u_ff),
                      self.z.data_ptr(), self.z_step_stride, _lib.dptr(self.X_traj), _lib.dptr(self.Y),
                      _lib.dptr(self.Xi), _lib.dptr(self.info), _lib.dptr(self.ws), self.ws.numel() * 8)

    def launch(self, stream_ptr: Optional[int] = None, out: Optional[torch.Tensor] = None):
        """One rollout launch; ``out`` (same shape / dtype / device as ``X_traj``) redirects the trajectory output, e.g.
        into the alternating buffers of ``distributed.OverlappedTubeGather``."""
        st = _lib.current_stream_ptr() if stream_ptr is None else stream_ptr
        args = self._args
        if out is not None:
            if out.shape != self.X_traj.shape or out.dtype != F64 or not out.is_contiguous():
                raise ValueError("out must be a contiguous float64 tensor shaped like X_traj")
            args = args[:15] + (_lib.dptr(out),) + args[16:]
        rc = self.lib.gpmpc_rollout(*args, st)
        if rc != 0:
            _lib.check(rc, "gpmpc_rollout")
        return self.X_traj if out is None else out
