"""Sample-sharded rollouts over the GPUs of one node (one process per GPU, ``torch.distributed``; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" is used by the CPU tests).

The path shards trivially: samples are independent given the shared real data, the shared input
sequence and their own base samples.  Rank r owns the contiguous global samples ``[r*Ns/G, (r+1)*Ns/G)``; the plan
(real-data factor, ~20 KB) is replicated; base samples are addressed by GLOBAL sample index, so the assembled tube is
identical for every GPU count.  The only collective is ONE all-gather of the ``(Ns/G, nx, H+1)`` float64 trajectory
shards per rollout (the reference has no collective at all: it scales out with a SLURM job array and merges pickles
offline, ``benchmarking/euler_job.sh:5-11``, ``generate_convex_hull.py:76-83``).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(Ns: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, as-even-as-possible split of ``Ns`` global sample ids."""
    base, rem = divmod(Ns, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_tube(X_local: torch.Tensor, Ns: int, group=None) -> torch.Tensor:
    """Assemble the reachable tube ``(Ns, nx, H+1)`` on every rank from the per-rank shards.

    Equal shards use a single ``all_gather_into_tensor`` (one ring/direct collective, no host staging); ragged shards
    (Ns not divisible by the world size) are padded to the largest shard and trimmed after the gather."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nloc = X_local.shape[0]
    sizes = [shard_range(Ns, r, world) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    if nloc != sizes[rank][1] - sizes[rank][0]:
        raise ValueError("local shard size does not match shard_range()")
    if nmax * world == Ns:
        out = torch.empty((Ns,) + tuple(X_local.shape[1:]), dtype=X_local.dtype, device=X_local.device)
        dist.all_gather_into_tensor(out, X_local.contiguous(), group=group)
        return out
    pad = torch.zeros((nmax,) + tuple(X_local.shape[1:]), dtype=X_local.dtype, device=X_local.device)
    pad[:nloc] = X_local
    buf = torch.empty((world * nmax,) + tuple(X_local.shape[1:]), dtype=X_local.dtype, device=X_local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * nmax: r * nmax + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


def sharded_rollout(local_rollout: Callable[[int, int], torch.Tensor], Ns: int, group=None) -> torch.Tensor:
    """Run ``local_rollout(lo, hi) -> X_traj[lo:hi]`` for this rank's shard of global samples and all-gather.

    ``local_rollout`` is the HIP rollout on the GPU path (see ``sharded_forward_sampling_rollout``); the CPU tests pass
    an oracle-backed callable to exercise the sharding + collective logic under gloo."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(Ns, rank, world)
    X_local = local_rollout(lo, hi)
    return all_gather_tube(X_local, Ns, group)


def sharded_forward_sampling_rollout(agent, u_ff, x0=None, group=None) -> torch.Tensor:
    """Forward-sampling rollout of ``agent.ns`` GLOBAL samples, sharded over the process group.

    Every rank holds the same ``agent`` configuration and the same base-sample tensor (seeded identically, or
    broadcast); each launches ``gpmpc_rollout`` on its own contiguous slice and the shards are all-gathered over
    RCCL.  Returns the full tube on every rank (device tensor)."""
    import numpy as np
    from . import _lib
    from .rollout import rollout_device
    p = agent.params
    u_ff = np.asarray(u_ff, dtype=np.float64)
    H = u_ff.shape[0]
    nograd = bool(p["env"]["use_model_without_derivatives"])
    mode = _lib.MODE_INDEPENDENT if nograd else _lib.MODE_RECONDITIONED
    T = 1 if nograd else 1 + agent.in_dim_x
    erv = agent.epistimic_random_vector.to(device=agent.torch_device, dtype=torch.float64).contiguous()
    per_slab = agent.ns * agent.g_ny * T
    z = erv.reshape(-1)[per_slab:]

    def local(lo, hi):
        res = rollout_device(agent, u_ff, z, erv.shape[1] * per_slab, H=H, mode=mode,
                             use_model_without_derivatives=nograd, x0=x0, want_samples=False, sample_slice=(lo, hi))
        return res.X_traj

    return sharded_rollout(local, agent.ns, group)
