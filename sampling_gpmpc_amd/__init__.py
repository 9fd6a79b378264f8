"""MI355X-native GP-posterior-sample rollout for sampling-based GP-MPC (drop-in for the hot path of
manish-pra/sampling-gpmpc: ``src/agent.py`` + ``src/GP_model.py`` + the forward-sampling harnesses)."""
from ._host_threads import limit_host_threads
limit_host_threads()        # pools larger than the cgroup's CPU quota freeze the launching thread (see _host_threads.py)
from .agent import Agent, random_vector_within_bounds                   # noqa: F401
from .environments import make_env, Pendulum, CarKinematicsModel        # noqa: F401
from .reachable_set import get_reachable_set_ball                       # noqa: F401
from . import _lib                                                      # noqa: F401

__all__ = ["Agent", "make_env", "Pendulum", "CarKinematicsModel", "get_reachable_set_ball",
           "random_vector_within_bounds"]
