"""gym-rs_amd — MI355X-native batched stepper for the gym-rs classic-control hot path.

Host-side mirror of the reference interface for ONE path: ``Env::step`` / ``Env::reset`` /
``action_space`` / ``observation_space`` (``/root/reference/src/core.rs:25-90``) of CartPole and
MountainCar (``src/envs/classical_control``), plus a spec-derived Pendulum.  All compute happens in
hand-written gfx950 HIP kernels behind the C ABI in ``include/gymrs_amd.h``; this package is the thin
ctypes binding over that ABI.  There is no CPU fallback: importing works anywhere, but creating an
engine without the built library or without a HIP device raises.

The directory name contains a hyphen, so import it with::

    import importlib; gymrs = importlib.import_module("gym-rs_amd")

or through the alias module ``gymrs_amd`` at the repository root.
"""
from .core import ActionReward, RewardRange
from .spaces import BoxR, Discrete
from .engine import (
    AUTO_RESET,
    TIME_LIMIT,
    TRACK_STATS,
    CARTPOLE,
    MOUNTAIN_CAR,
    PENDULUM,
    BatchedEngine,
    CartPoleParams,
    GymrsError,
    InvalidActionError,
    MountainCarParams,
    PendulumParams,
    ShardedEngine,
    shard_range,
)
from .envs import (
    CartPoleEnv,
    CartPoleObservation,
    MountainCarEnv,
    MountainCarObservation,
    PendulumEnv,
    PendulumObservation,
    RenderMode,
)
from ._lib import library_path, load_library
from . import sharded
from .engine import params_from_json

__all__ = [
    "ActionReward", "RewardRange", "BoxR", "Discrete", "BatchedEngine", "GymrsError", "InvalidActionError",
    "CartPoleParams", "MountainCarParams", "PendulumParams", "CartPoleEnv", "MountainCarEnv", "PendulumEnv",
    "CartPoleObservation", "MountainCarObservation", "PendulumObservation", "RenderMode",
    "AUTO_RESET", "TRACK_STATS", "TIME_LIMIT", "CARTPOLE", "MOUNTAIN_CAR", "PENDULUM",
    "library_path", "load_library", "shard_range", "ShardedEngine", "sharded", "params_from_json",
]
