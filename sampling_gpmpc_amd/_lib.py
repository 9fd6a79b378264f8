"""ctypes binding of libgpmpc_hip.so (C-ABI declared in include/gpmpc_hip.h).

The library is the product: there is NO CPU fallback.  ``load()`` raises if the shared object is missing, and every
compute wrapper raises if its tensors are not on a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgpmpc_hip.so")

MAX_NY, MAX_D, MAX_T, MAX_NX, MAX_NU = 4, 4, 5, 8, 4
ABI_VERSION = 9

ENV_PENDULUM1D, ENV_CAR_RESIDUAL = 0, 1
MODE_INDEPENDENT, MODE_RECONDITIONED = 0, 1

INFO_TRAIN_CHOL_FAIL = 0x0001
INFO_ROOT_JITTER_MASK = 0x000E
INFO_ROOT_FAIL = 0x0010
INFO_VAR_CLAMPED = 0x0020
INFO_NEG_1x1 = 0x0040
INFO_ROOT_EIGH = 0x0080
INFO_EIGH_NOCONV = 0x0100
INFO_STATE_FULL = 0x0200

ROOT_AUTO, ROOT_EIGH, ROOT_CHOLESKY = 0, 1, 2
# gpmpc_rollout_pin_kernel / gpmpc_rollout_last_kernel (include/gpmpc_hip.h)
PENDING_USE, PENDING_WRITE = 1, 2                      # gpmpc_joint_sample_pending
JOINT_AUTO, JOINT_VALU, JOINT_MFMA = 0, 1, 2          # gpmpc_joint_pin_path / gpmpc_joint_last_path
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_FAST, KERNEL_INDEP, KERNEL_TILES, KERNEL_ONE = -1, 0, 1, 2, 3, 4


class GpDesc(C.Structure):
    _fields_ = [("g_ny", C.c_int32), ("D", C.c_int32), ("T", C.c_int32), ("N_r", C.c_int32),
                ("real_has_grad", C.c_int32), ("grid_n0", C.c_int32), ("grid_n1", C.c_int32), ("_pad", C.c_int32),
                ("ell", (C.c_double * MAX_D) * MAX_NY), ("outputscale", C.c_double * MAX_NY),
                ("noise", C.c_double * MAX_T), ("jitter", C.c_double), ("var_floor", C.c_double)]


class EnvDesc(C.Structure):
    _fields_ = [("env_id", C.c_int32), ("nx", C.c_int32), ("nu", C.c_int32), ("use_feedback", C.c_int32),
                ("dt", C.c_double), ("p0", C.c_double), ("p1", C.c_double),
                ("K", (C.c_double * MAX_NX) * MAX_NU), ("x_goal", C.c_double * MAX_NX)]


# every symbol include/gpmpc_hip.h declares: name -> (restype, argtypes)
_P, _I32, _I64, _D, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_size_t
SYMBOLS = {
    "gpmpc_abi_version": (C.c_int, []),
    "gpmpc_last_error_string": (C.c_char_p, []),
    "gpmpc_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gpmpc_selftest": (C.c_int, [_P]),
    "gpmpc_plan_bytes": (_SZ, [C.POINTER(GpDesc)]),
    "gpmpc_plan_build": (C.c_int, [C.POINTER(GpDesc), _P, _P, _P, _P, _P]),
    "gpmpc_rollout_workspace_bytes": (_SZ, [C.POINTER(GpDesc), _I32, _I32, _I64, _I32]),
    "gpmpc_rollout": (C.c_int, [C.POINTER(GpDesc), C.POINTER(EnvDesc), _P, _P, _I32, _I32, _D, _D, _I64, _I32,
                                _P, _I32, _P, _P, _I64, _P, _P, _P, _P, _P, _SZ, _P]),
    "gpmpc_rollout_state_bytes": (_SZ, [C.POINTER(GpDesc), _I64, _I32, _I32]),
    "gpmpc_rollout_pin_kernel": (C.c_int, [_I32]),
    "gpmpc_rollout_last_kernel": (C.c_int, []),
    "gpmpc_rollout_kernel_for": (C.c_int, [C.POINTER(GpDesc), C.POINTER(EnvDesc), _I32, _I32, _I64, _I32]),
    "gpmpc_rollout_seeded_workspace_bytes": (_SZ, [C.POINTER(GpDesc), _I32, _I32, _I64, _I32, _I32, _I32]),
    "gpmpc_rollout_seeded": (C.c_int, [C.POINTER(GpDesc), C.POINTER(EnvDesc), _P, _P, _I32, _I32, _D, _D, _I64, _I32,
                                       _P, _I32, _P, _P, _I64, _P, _P, _P, _P, _P, _SZ, _P, _P, _P, _I32, _P, _P, _I32, _P, _I32, _I32, _I32]),
    "gpmpc_joint_workspace_bytes": (_SZ, [C.POINTER(GpDesc), _I64, _I32, _I32]),
    "gpmpc_joint_cache_bytes": (_SZ, [C.POINTER(GpDesc), _I64, _I32]),
    "gpmpc_joint_sample": (C.c_int, [C.POINTER(GpDesc), _P, _P, _I64, _I32, _P, _P, _P, _I32, _I32, _P, _P,
                                     _D, _D, _I32, _P, _P, _P, _P, _P, _I32, _P, _P, _SZ, _P, _P, _I32, _I32]),
    "gpmpc_joint_sample_pending": (C.c_int, [C.POINTER(GpDesc), _P, _P, _I64, _I32, _P, _P, _P, _I32, _I32, _P, _P,
                                             _D, _D, _I32, _P, _P, _P, _P, _P, _I32, _P, _P, _SZ, _P, _P, _I32, _I32, _I32]),
    "gpmpc_joint_pending_written": (C.c_int, []),
    "gpmpc_joint_pin_path": (C.c_int, [_I32]),
    "gpmpc_joint_last_path": (C.c_int, []),
    "gpmpc_assemble_jacobians": (C.c_int, [C.POINTER(GpDesc), C.POINTER(EnvDesc), _I64, _I32, _P, _P, _P, _P, _P, _P]),
    "gpmpc_plin_len": (_I64, [_I32, _I32, _I64]),
    "gpmpc_pack_plin": (C.c_int, [_I32, _I32, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gpmpc_pack_plin_fb": (C.c_int, [_I32, _I32, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gpmpc_build_x_hat": (C.c_int, [_I32, _I32, _I64, _I32, _P, _P, _I32, _P, _P]),
    "gpmpc_assemble_jacobians_plin": (C.c_int, [C.POINTER(GpDesc), C.POINTER(EnvDesc), _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                                _P, _P, _P]),
    "gpmpc_or_reduce_words": (C.c_int, [_P, _I64, _P, _P]),
    "gpmpc_base_samples": (C.c_int, [C.c_uint64, _I32, _I32, _I64, _I64, _I32, _D, _P, _P, _P]),
}

_lib: Optional[C.CDLL] = None


class GpmpcError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libgpmpc_hip.so and bind every declared symbol.  Raises loudly if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpmpcError(
            f"{LIB_PATH} not found: build it with `python sampling_gpmpc_amd/csrc/build.py` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the GP rollout path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = res, args
    v = lib.gpmpc_abi_version()
    if v != ABI_VERSION:
        raise GpmpcError(f"libgpmpc_hip.so ABI version {v}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().gpmpc_last_error_string()
        raise GpmpcError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def require_hip_device(dev) -> torch.device:
    dev = torch.device(dev)
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise GpmpcError("the GP rollout path needs a HIP device (torch 'cuda' on ROCm); no CPU fallback exists")
    return dev


def dptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous float64/int32 HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise GpmpcError("tensor passed to libgpmpc_hip.so is not on a HIP device")
    if not t.is_contiguous():
        raise GpmpcError("tensor passed to libgpmpc_hip.so is not contiguous")
    return t.data_ptr()


def current_stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def to_host(t: torch.Tensor):
    """Device tensor -> numpy array through a PINNED staging tensor of torch's caching host allocator: the copy runs at the
    DMA rate instead of through the runtime's pageable bounce buffer (19 MB of Jacobians and stage parameters per SQP
    iteration at the configs[4] shard: 1.5 -> 0.8 ms), and the array owns its buffer (the allocator hands the block out again
    only after the array has been collected), so callers may keep results across calls as with ``.cpu().numpy()``."""
    if not t.is_cuda:
        return t.detach().numpy()
    out = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    out.copy_(t.detach(), non_blocking=True)
    # the copy is asynchronous and nothing blocking follows it here: wait for ITS completion in either wait mode (polling, or
    # the runtime's blocking wait under GPMPC_HOST_WAIT=block) before the host memory is handed out
    ev = torch.cuda.Event()
    ev.record()
    if os.environ.get("GPMPC_HOST_WAIT") == "block":
        ev.synchronize()
    else:
        while not ev.query():
            pass
    return out.numpy()


def host_wait(t: Optional[torch.Tensor] = None) -> None:
    """Wait on the host, by POLLING, for the work queued on the current stream; call it in front of every host read of a
    device result (``.item()``, ``.cpu()``, ``.tolist()``) that may follow a kernel of more than a millisecond.

    Why: the runtime's own wait spins briefly and then sleeps until the completion interrupt.  Where that interrupt is not
    delivered promptly (profiles/r3_closed_loop_trace.md: on this pool every such wait ends on a 100 ms timer tick, which
    made an SQP iteration of 8 ms of GPU work take 100 ms), polling ``hipEventQuery`` costs one busy host thread for the
    length of the kernel and returns when the kernel does.  ``GPMPC_HOST_WAIT=block`` restores the runtime's wait.
    CPU tensors (the gloo tests) need nothing."""
    if t is not None and not t.is_cuda:
        return
    if not torch.cuda.is_available() or os.environ.get("GPMPC_HOST_WAIT") == "block":
        return
    ev = torch.cuda.Event()
    ev.record()
    while not ev.query():
        pass


def make_gp_desc(g_ny, D, T, N_r, real_has_grad, ell, outputscale, noise, jitter, var_floor=1e-10,
                 grid=(0, 0)) -> GpDesc:
    d = GpDesc()
    d.g_ny, d.D, d.T, d.N_r, d.real_has_grad = int(g_ny), int(D), int(T), int(N_r), int(bool(real_has_grad))
    d.grid_n0, d.grid_n1 = int(grid[0]), int(grid[1])
    for o in range(g_ny):
        for k in range(D):
            d.ell[o][k] = float(ell[o][k])
        d.outputscale[o] = float(outputscale[o])
    for t in range(T):
        d.noise[t] = float(noise[t])
    d.jitter, d.var_floor = float(jitter), float(var_floor)
    return d


def make_env_desc(env_id, nx, nu, use_feedback, dt, p0, p1, K, x_goal) -> EnvDesc:
    e = EnvDesc()
    e.env_id, e.nx, e.nu, e.use_feedback = int(env_id), int(nx), int(nu), int(bool(use_feedback))
    e.dt, e.p0, e.p1 = float(dt), float(p0), float(p1)
    if K is not None:
        for i in range(nu):
            for j in range(nx):
                e.K[i][j] = float(K[i][j])
    for j in range(nx):
        e.x_goal[j] = float(x_goal[j])
    return e


def device_info(dev: int = 0):
    lib = load()
    name = C.create_string_buffer(256)
    cu, lds = C.c_int(0), C.c_int(0)
    check(lib.gpmpc_device_info(dev, name, 256, C.byref(cu), C.byref(lds)), "gpmpc_device_info")
    return name.value.decode(), cu.value, lds.value
