"""Environment plug-ins used by the BASELINE configs (host-side API; their per-step maps are fused into the HIP
rollout kernel, see csrc/gpmpc_device.hpp)."""
from .pendulum1D import Pendulum
from .car_model_residual import CarKinematicsModel

# the names reference main.py:11-14 binds, as looked up through params["env"]["dynamics"]
REGISTRY = {"Pendulum1D": Pendulum, "bicycle_Bdx": CarKinematicsModel}


def make_env(params, device=None):
    return REGISTRY[params["env"]["dynamics"]](params, device=device)
