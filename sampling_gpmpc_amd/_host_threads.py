"""Host thread pools sized to the CPU time the process actually has.

The GP side of an SQP iteration is a handful of kernel launches and small host arrays, but torch (OpenMP) and numpy
(BLAS) size their worker pools by the number of VISIBLE cores.  In a container whose cgroup grants fewer CPUs than it shows
(the MI355X boxes of this project: 256 visible, ``cpu.max`` = 16 CPUs per 100 ms period) every small host op wakes
hundreds of spinning workers, the period's quota is gone within milliseconds and the whole process - the thread that
launches kernels included - is frozen until the next 100 ms boundary: 8 ms of GPU work per SQP iteration took 100 ms of
wall clock (profiles/r3_closed_loop_trace.md: ``nr_throttled`` counts it; one worker thread: 39 ms per MPC step instead
of 200-600).  ``limit_host_threads()`` runs when the package is imported and only ever LOWERS the pools, only when they
exceed the quota, and never when the user has sized them (OMP_NUM_THREADS / GPMPC_HOST_THREADS=keep)."""
from __future__ import annotations

import os
from typing import Optional

_CGROUP = "/sys/fs/cgroup"


def parse_cpu_max(text: str) -> Optional[float]:
    """cgroup v2 ``cpu.max`` ("<quota> <period>" in microseconds, or "max <period>") -> CPUs, None when unlimited."""
    parts = text.split()
    if len(parts) != 2 or parts[0] == "max":
        return None
    quota, period = float(parts[0]), float(parts[1])
    return quota / period if quota > 0 and period > 0 else None


def cgroup_cpu_quota(root: str = _CGROUP) -> Optional[float]:
    """CPUs granted to this process by its cgroup (v2 ``cpu.max``, v1 ``cpu.cfs_quota_us``); None: no limit / unknown."""
    try:
        with open(os.path.join(root, "cpu.max")) as f:
            return parse_cpu_max(f.read())
    except OSError:
        pass
    try:
        with open(os.path.join(root, "cpu", "cpu.cfs_quota_us")) as f:
            q = float(f.read())
        with open(os.path.join(root, "cpu", "cpu.cfs_period_us")) as f:
            per = float(f.read())
        return q / per if q > 0 and per > 0 else None
    except (OSError, ValueError):
        return None


def host_thread_budget(visible: int, quota: Optional[float], current: int) -> Optional[int]:
    """Pool size to set, or None to leave the pools alone.  Workers spin between parallel regions, so a pool as large as
    the quota still exhausts it: half the quota, at most 8 (nothing on this path has more host parallelism than that)."""
    cpus = min(float(visible), quota) if quota is not None else float(visible)
    if current <= cpus:
        return None
    return max(1, min(8, int(cpus // 2)))


def limit_host_threads() -> Optional[int]:
    mode = os.environ.get("GPMPC_HOST_THREADS", "")
    if mode == "keep" or (not mode and os.environ.get("OMP_NUM_THREADS")):
        return None
    import torch
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    if mode.isdigit() and int(mode) < 1:                  # torch.set_num_threads(0) raises: an import must not
        import logging
        logging.getLogger("sampling_gpmpc_amd").warning("GPMPC_HOST_THREADS=%s ignored (a positive thread count or 'keep')", mode)
        mode = ""
    want = int(mode) if mode.isdigit() else host_thread_budget(visible, cgroup_cpu_quota(), torch.get_num_threads())
    if want is None:
        return None
    if want < torch.get_num_threads():
        import logging
        logging.getLogger("sampling_gpmpc_amd").warning(
            "host thread pools lowered from %d to %d (the container's CPU quota; GPMPC_HOST_THREADS=keep leaves them alone): "
            "this caps every CPU pool of the process (torch, BLAS, OpenMP), including a host-side QP solver's",
            torch.get_num_threads(), want)
    torch.set_num_threads(want)
    try:                                                  # numpy / scipy BLAS and OpenMP pools that are already loaded
        import threadpoolctl
        threadpoolctl.threadpool_limits(limits=want)
    except Exception:                                     # not installed, or a pool that cannot be resized: torch's is the large one
        pass
    return want
