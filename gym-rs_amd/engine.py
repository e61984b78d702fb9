"""BatchedEngine — N independent gym-rs envs stepped by one HIP kernel launch.

Thin ctypes binding over the C ABI (``include/gymrs_amd.h``).  The batched form of the reference
API: ``step(actions)`` fills device arrays ``obs`` / ``reward`` / ``done`` / ``truncated``
(``ActionReward``, core.rs:94-106) for every lane; ``reset(seed, options)`` is ``Env::reset``
(core.rs:45-50) for every lane.  numpy is used only to hand host buffers across the boundary.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from ._lib import load_library

CARTPOLE, MOUNTAIN_CAR, PENDULUM = 0, 1, 2
AUTO_RESET, TRACK_STATS, TIME_LIMIT = 1, 2, 4

_OK, _EINVAL, _EHIP, _ENCCL, _ENOMEM, _EACTION = range(6)


class GymrsError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"gymrs status {status}: {message}")
        self.status = status


class InvalidActionError(GymrsError, AssertionError):
    """An action outside the action space: the reference's ``assert!`` panic (cartpole.rs:402-406)."""


class CartPoleParams(C.Structure):
    """The ``pub`` physics fields of ``CartPoleEnv`` (cartpole.rs:53-82)."""

    _fields_ = [
        ("gravity", C.c_double), ("masscart", C.c_double), ("masspole", C.c_double), ("length", C.c_double),
        ("force_mag", C.c_double), ("tau", C.c_double), ("theta_threshold_radians", C.c_double),
        ("x_threshold", C.c_double), ("kinematics_integrator", C.c_int32), ("max_episode_steps", C.c_uint32),
    ]


class MountainCarParams(C.Structure):
    """The ``pub`` physics fields of ``MountainCarEnv`` (mountain_car.rs:49-77)."""

    _fields_ = [
        ("min_position", C.c_double), ("max_position", C.c_double), ("max_speed", C.c_double),
        ("goal_position", C.c_double), ("goal_velocity", C.c_double), ("force", C.c_double),
        ("gravity", C.c_double), ("max_episode_steps", C.c_uint32), ("_pad", C.c_uint32),
    ]


class PendulumParams(C.Structure):
    """Gym Pendulum-v1 constants (spec-derived; not in the reference)."""

    _fields_ = [
        ("max_speed", C.c_double), ("max_torque", C.c_double), ("dt", C.c_double), ("g", C.c_double),
        ("m", C.c_double), ("l", C.c_double), ("max_episode_steps", C.c_uint32), ("_pad", C.c_uint32),
    ]


class Trajectory(C.Structure):
    """``gymrs_trajectory`` (include/gymrs_amd.h)."""
    _fields_ = [("obs", C.c_void_p), ("actions", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p),
                ("truncated", C.c_void_p), ("lane_stride", C.c_uint64)]


_PARAMS = {CARTPOLE: CartPoleParams, MOUNTAIN_CAR: MountainCarParams, PENDULUM: PendulumParams}
_STATE_DIM = {CARTPOLE: 4, MOUNTAIN_CAR: 2, PENDULUM: 2}
_OBS_DIM = {CARTPOLE: 4, MOUNTAIN_CAR: 2, PENDULUM: 3}
_ACTION_DTYPE = {CARTPOLE: np.uint8, MOUNTAIN_CAR: np.uint8, PENDULUM: np.float32}


def default_params(kind: int):
    lib = load_library()
    p = _PARAMS[kind]()
    _check(lib, lib.gymrs_default_params(kind, C.byref(p)))
    return p


def params_from_json(kind: int, text: str, params=None):
    """Inverse of ``BatchedEngine.env_json`` for the physics fields: (params, state or None).  Missing keys keep the
    value in ``params`` (default: the env's defaults)."""
    lib = load_library()
    p = params if params is not None else default_params(kind)
    state = (C.c_double * 4)()
    dim = C.c_int()
    _check(lib, lib.gymrs_params_from_json(int(kind), text.encode("utf-8"), C.byref(p), state, C.byref(dim)))
    return p, ([state[j] for j in range(dim.value)] if dim.value else None)


def _check(lib, status: int) -> None:
    if status == _OK:
        return
    msg = lib.gymrs_last_error().decode("utf-8", "replace")
    if status == _EACTION:
        raise InvalidActionError(status, msg)
    raise GymrsError(status, msg)


def shard_range(n_total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous lane shard of rank ``rank``: (global offset, count).  Lanes are independent, so
    the batch shards trivially; the global id keeps Philox draws identical for any world size."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world_size)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


class BatchedEngine:
    """One engine = one shard of lanes on one GPU, driven by one host thread."""

    def __init__(self, kind: int, n_envs: int, *, global_env_offset: int = 0, device: int = 0, params=None,
                 flags: int = 0, lanes_per_thread: Optional[int] = None):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.kind = int(kind)
        self.n_envs = int(n_envs)
        self.global_env_offset = int(global_env_offset)
        self.flags = int(flags)
        self.state_dim = _STATE_DIM[self.kind]
        self.obs_dim = _OBS_DIM[self.kind]
        self.action_dtype = _ACTION_DTYPE[self.kind]
        self.params = params if params is not None else default_params(self.kind)
        _check(self._lib, self._lib.gymrs_engine_create(self.kind, self.n_envs, self.global_env_offset, int(device),
                                                        C.byref(self.params), self.flags, C.byref(self._h)))
        if lanes_per_thread is not None:
            self.set_tuning(lanes_per_thread)

    # -- lifetime ---------------------------------------------------------------------------
    def close(self) -> None:
        """``Env::close`` (core.rs:56)."""
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.gymrs_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- Clone + Serialize (core.rs:25) ---------------------------------------------------------
    def clone(self) -> "BatchedEngine":
        """Deep copy (``Env: Clone``): lane state, episode bookkeeping, statistics, RNG position."""
        h = C.c_void_p()
        _check(self._lib, self._lib.gymrs_engine_clone(self._h, C.byref(h)))
        other = BatchedEngine.__new__(BatchedEngine)
        other.__dict__.update({k: v for k, v in self.__dict__.items() if k != "_h"})
        other.params = type(self.params).from_buffer_copy(self.params)
        other._h = h
        return other

    def snapshot(self) -> bytes:
        """Everything a step can observe as an opaque blob (``Env: Serialize``); see ``restore``."""
        size = C.c_uint64()
        _check(self._lib, self._lib.gymrs_snapshot_size(self._h, C.byref(size)))
        buf = (C.c_char * size.value)()
        _check(self._lib, self._lib.gymrs_snapshot_save(self._h, buf, size.value))
        return bytes(buf)

    def restore(self, blob: bytes) -> None:
        """Load a ``snapshot()`` taken from an engine of the same kind, n_envs and flags; stepping then
        continues bit-identically to the engine the snapshot came from."""
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        _check(self._lib, self._lib.gymrs_snapshot_load(self._h, buf, len(blob)))

    # -- stream / tuning ----------------------------------------------------------------------
    @property
    def stream(self) -> int:
        p = C.c_void_p()
        _check(self._lib, self._lib.gymrs_get_stream(self._h, C.byref(p)))
        return p.value or 0

    def set_stream(self, hip_stream: int) -> None:
        _check(self._lib, self._lib.gymrs_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_tuning(self, lanes_per_thread: int = 4, memory_hint: int = 0) -> None:
        """lanes_per_thread: 4 or 8.  memory_hint: 0 automatic, 1 every access non-temporal, 2 none, 3 only the stores nobody reads again."""
        _check(self._lib, self._lib.gymrs_set_tuning(self._h, int(lanes_per_thread), int(memory_hint)))

    # -- the pub physics fields after construction; the Serialize view -------------------------------
    def set_params(self, params) -> None:
        """Assign the pub physics fields (cartpole.rs:53-82) of every lane: only the launch constants change --
        state, steps_beyond_terminated, episode clocks, statistics, seed and tick carry on."""
        if not isinstance(params, _PARAMS[self.kind]):
            raise TypeError(f"expected {_PARAMS[self.kind].__name__}")
        _check(self._lib, self._lib.gymrs_set_params(self._h, C.byref(params)))
        self.params = type(params).from_buffer_copy(params)

    def get_params(self):
        p = _PARAMS[self.kind]()
        _check(self._lib, self._lib.gymrs_get_params(self._h, C.byref(p)))
        return p

    def env_json(self, lane: int = 0) -> str:
        """What ``serde_json::to_string(&env)`` prints for the reference env lane ``lane`` stands for (core.rs:25)."""
        need = C.c_uint64()
        buf = C.create_string_buffer(2048)
        st = self._lib.gymrs_env_json(self._h, int(lane), buf, len(buf), C.byref(need))
        if st != _OK and need.value > len(buf):
            buf = C.create_string_buffer(need.value)
            st = self._lib.gymrs_env_json(self._h, int(lane), buf, len(buf), C.byref(need))
        _check(self._lib, st)
        return buf.value.decode("utf-8")

    # -- Env::reset ------------------------------------------------------------------------------
    def reset(self, seed: Optional[int] = None, options: Optional[Sequence[float]] = None) -> int:
        """``reset(seed, _, options)`` for every lane.  ``options`` = state_dim lows then state_dim
        highs.  Returns the seed number used (seeding.rs:21-26)."""
        bounds = None
        if options is not None:
            arr = np.ascontiguousarray(options, dtype=np.float32)
            if arr.size != 2 * self.state_dim:
                raise ValueError(f"options needs {2 * self.state_dim} floats (lows then highs)")
            bounds = arr.ctypes.data_as(C.POINTER(C.c_float))
        used = C.c_uint64()
        _check(self._lib, self._lib.gymrs_reset(self._h, 0 if seed is None else 1, 0 if seed is None else int(seed),
                                                bounds, C.byref(used)))
        return used.value

    def reset_pcg64(self, seed: Optional[int] = None, seeds_dev: Optional[int] = None,
                    options: Optional[Sequence[float]] = None) -> int:
        """``reset`` with the reference's own generator chain (``gymrs_reset_pcg64``): lane i gets, rounded once to f32,
        the state the reference's ``reset(Some(s), _, options)`` returns for ``s = seed + global_env_offset + i`` or, if
        ``seeds_dev`` (device address of n_envs u64) is given, ``s = seeds[i]``.  ``options`` = lows then highs, f64.
        CartPole / MountainCar only.  Returns the seed number used."""
        bounds = None
        if options is not None:
            arr = np.ascontiguousarray(options, dtype=np.float64)
            if arr.size != 2 * self.state_dim:
                raise ValueError(f"options needs {2 * self.state_dim} floats (lows then highs)")
            bounds = arr.ctypes.data_as(C.POINTER(C.c_double))
        used = C.c_uint64()
        _check(self._lib, self._lib.gymrs_reset_pcg64(self._h, 0 if seed is None else 1, 0 if seed is None else int(seed) & ((1 << 64) - 1),
                                                      C.c_void_p(seeds_dev) if seeds_dev else None, bounds, C.byref(used)))
        return used.value

    # -- Env::step ---------------------------------------------------------------------------------
    def step(self, actions_dev: int) -> None:
        """Asynchronous; ``actions_dev`` is a device address of n_envs actions."""
        _check(self._lib, self._lib.gymrs_step(self._h, C.c_void_p(actions_dev)))

    def step_host(self, actions) -> None:
        a = np.ascontiguousarray(actions, dtype=self.action_dtype)
        if a.size != self.n_envs:
            raise ValueError("one action per lane expected")
        _check(self._lib, self._lib.gymrs_step_host(self._h, a.ctypes.data_as(C.c_void_p)))
        self.sync()  # the host buffer must outlive the copy

    def step_many(self, actions_dev: int, stride_bytes: int, n_buffers: int, n_steps: int, use_graph: bool = False) -> None:
        _check(self._lib, self._lib.gymrs_step_many(self._h, C.c_void_p(actions_dev), int(stride_bytes), int(n_buffers),
                                                    int(n_steps), 1 if use_graph else 0))

    def sync(self) -> None:
        _check(self._lib, self._lib.gymrs_sync(self._h))

    # -- device views -----------------------------------------------------------------------------
    def obs_ptrs(self):
        ptrs = (C.c_void_p * 4)()
        dim = C.c_int()
        _check(self._lib, self._lib.gymrs_obs_ptrs(self._h, ptrs, C.byref(dim)))
        return [ptrs[j] for j in range(dim.value)]

    def state_ptrs(self):
        ptrs = (C.c_void_p * 4)()
        dim = C.c_int()
        _check(self._lib, self._lib.gymrs_state_ptrs(self._h, ptrs, C.byref(dim)))
        return [ptrs[j] for j in range(dim.value)]

    def _ptr(self, fn) -> int:
        p = C.c_void_p()
        _check(self._lib, fn(self._h, C.byref(p)))
        return p.value or 0

    @property
    def reward_ptr(self) -> int:
        return self._ptr(self._lib.gymrs_reward_ptr)

    @property
    def done_ptr(self) -> int:
        return self._ptr(self._lib.gymrs_done_ptr)

    @property
    def truncated_ptr(self) -> int:
        return self._ptr(self._lib.gymrs_truncated_ptr)

    # -- host copies ---------------------------------------------------------------------------------
    def get_obs(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        count = self.n_envs - first if count is None else count
        out = np.empty((self.obs_dim, count), dtype=np.float32)
        _check(self._lib, self._lib.gymrs_get_obs(self._h, first, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def get_state(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        count = self.n_envs - first if count is None else count
        out = np.empty((self.state_dim, count), dtype=np.float32)
        _check(self._lib, self._lib.gymrs_get_state(self._h, first, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_state(self, state, first: int = 0) -> None:
        arr = np.ascontiguousarray(state, dtype=np.float32)
        if arr.ndim != 2 or arr.shape[0] != self.state_dim:
            raise ValueError(f"state must have shape ({self.state_dim}, count)")
        _check(self._lib, self._lib.gymrs_set_state(self._h, first, arr.shape[1], arr.ctypes.data_as(C.c_void_p)))

    def get_step_result(self, first: int = 0, count: Optional[int] = None):
        count = self.n_envs - first if count is None else count
        reward = np.empty(count, dtype=np.float32)
        done = np.empty(count, dtype=np.uint8)
        trunc = np.empty(count, dtype=np.uint8)
        _check(self._lib, self._lib.gymrs_get_step_result(self._h, first, count, reward.ctypes.data_as(C.c_void_p),
                                                          done.ctypes.data_as(C.c_void_p), trunc.ctypes.data_as(C.c_void_p)))
        return reward, done, trunc

    # -- statistics ----------------------------------------------------------------------------------
    def stats(self) -> np.ndarray:
        """{sum_return, sum_length, n_episodes, n_steps} of this shard."""
        out = (C.c_double * 4)()
        _check(self._lib, self._lib.gymrs_stats(self._h, out))
        return np.array(out[:], dtype=np.float64)

    def stats_clear(self) -> None:
        _check(self._lib, self._lib.gymrs_stats_clear(self._h))

    def stats_device(self) -> int:
        """Device address of 4 doubles, valid after the engine stream reaches this point."""
        p = C.c_void_p()
        _check(self._lib, self._lib.gymrs_stats_device(self._h, C.byref(p)))
        return p.value or 0

    # RCCL, one process per GPU
    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * 128)()
        _check(self._lib, self._lib.gymrs_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, n_ranks: int, rank: int, unique_id: bytes) -> None:
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _check(self._lib, self._lib.gymrs_comm_init(self._h, n_ranks, rank, buf))

    def allreduce_stats(self) -> np.ndarray:
        out = (C.c_double * 4)()
        _check(self._lib, self._lib.gymrs_allreduce_stats(self._h, out))
        return np.array(out[:], dtype=np.float64)

    # -- utilities -----------------------------------------------------------------------------------
    def rollout(self, n_steps: int, action_seed: int, action_t0: int = 0) -> None:
        """``n_steps`` random-policy steps of every lane in one kernel launch: the effect of
        ``fill_actions(buf, action_seed, action_t0 + k); step(buf)`` for k in range(n_steps), bit for bit
        (the caller loop of examples/cartpole.rs:15-30)."""
        _check(self._lib, self._lib.gymrs_rollout(self._h, int(n_steps), int(action_seed), int(action_t0)))

    def rollout_record(self, n_steps: int, action_seed: int, action_t0: int, *, obs: int, actions: int, reward: int,
                       done: int, truncated: int = 0, lane_stride: Optional[int] = None) -> None:
        """``rollout`` that also keeps the trajectory.  The arguments are device addresses of buffers laid out
        ``obs[n_steps][obs_dim][lane_stride]`` (f32), ``actions/reward/done/truncated[n_steps][lane_stride]``;
        ``lane_stride`` defaults to n_envs rounded up to a multiple of 16."""
        stride = int(lane_stride) if lane_stride is not None else (self.n_envs + 15) // 16 * 16
        traj = Trajectory(C.c_void_p(obs), C.c_void_p(actions), C.c_void_p(reward), C.c_void_p(done),
                          C.c_void_p(truncated or None), stride)
        _check(self._lib, self._lib.gymrs_rollout_record(self._h, int(n_steps), int(action_seed), int(action_t0), C.byref(traj)))

    def fill_actions(self, actions_dev: int, seed: int, t: int) -> None:
        _check(self._lib, self._lib.gymrs_fill_actions(self._h, C.c_void_p(actions_dev), int(seed), int(t)))

    def tick(self) -> Tuple[int, int]:
        t, s = C.c_uint64(), C.c_uint64()
        _check(self._lib, self._lib.gymrs_get_tick(self._h, C.byref(t), C.byref(s)))
        return t.value, s.value


class _ShardView(BatchedEngine):
    """Block r of a ShardedEngine: the engine's views (obs / reward / done pointers, stream, env_json).  Borrowed -- the
    sharder owns it, ``close`` is a no-op, and it must not be stepped directly while the sharder is in use."""

    def __init__(self, lib, handle, kind, n_envs, offset, flags, device):
        self._lib = lib
        self._h = C.c_void_p(handle)
        self.kind, self.n_envs, self.global_env_offset, self.flags, self.device = kind, n_envs, offset, flags, device
        self.state_dim, self.obs_dim, self.action_dtype = _STATE_DIM[kind], _OBS_DIM[kind], _ACTION_DTYPE[kind]
        self.params = None

    def close(self) -> None:
        self._h = C.c_void_p()


class ShardedEngine:
    """One batch of ``n_total`` lanes over several GPUs in ONE process (``gymrs_sharded_*``, include/gymrs_amd.h): contiguous
    blocks, one engine and one native host thread per block, global lane ids -- every result is bit-identical to one
    ``BatchedEngine`` of ``n_total`` lanes.  ``devices[r]`` is block r's device (a device may repeat)."""

    def __init__(self, kind: int, n_total: int, devices: Sequence[int], *, global_env_offset: int = 0, params=None, flags: int = 0):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.kind, self.n_total, self.flags = int(kind), int(n_total), int(flags)
        self.state_dim, self.obs_dim, self.action_dtype = _STATE_DIM[self.kind], _OBS_DIM[self.kind], _ACTION_DTYPE[self.kind]
        self.params = params if params is not None else default_params(self.kind)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        _check(self._lib, self._lib.gymrs_sharded_create(self.kind, self.n_total, int(global_env_offset), len(devices), devs,
                                                         C.byref(self.params), self.flags, C.byref(self._h)))
        self.shards = []
        for r in range(len(devices)):
            eng, first, count, dev = C.c_void_p(), C.c_uint64(), C.c_uint64(), C.c_int()
            _check(self._lib, self._lib.gymrs_sharded_shard(self._h, r, C.byref(eng), C.byref(first), C.byref(count), C.byref(dev)))
            view = _ShardView(self._lib, eng.value, self.kind, count.value, int(global_env_offset) + first.value, self.flags, dev.value)
            view.first_lane = first.value
            self.shards.append(view)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            for s in self.shards:
                s.close()
            self._lib.gymrs_sharded_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ptrs(self, addresses):
        if len(addresses) != len(self.shards):
            raise ValueError("one device address per shard expected")
        return (C.c_void_p * len(addresses))(*[C.c_void_p(int(a)) for a in addresses])

    def reset(self, seed: Optional[int] = None, options: Optional[Sequence[float]] = None) -> int:
        bounds = None
        if options is not None:
            arr = np.ascontiguousarray(options, dtype=np.float32)
            if arr.size != 2 * self.state_dim:
                raise ValueError(f"options needs {2 * self.state_dim} floats (lows then highs)")
            bounds = arr.ctypes.data_as(C.POINTER(C.c_float))
        used = C.c_uint64()
        _check(self._lib, self._lib.gymrs_sharded_reset(self._h, 0 if seed is None else 1, 0 if seed is None else int(seed), bounds, C.byref(used)))
        return used.value

    def step(self, actions_dev: Sequence[int]) -> None:
        """``actions_dev[r]``: device address (on block r's device) of block r's actions.  Asynchronous."""
        _check(self._lib, self._lib.gymrs_sharded_step(self._h, self._ptrs(actions_dev)))

    def step_many(self, actions_dev: Sequence[int], stride_bytes: int, n_buffers: int, n_steps: int, use_graph: bool = False) -> None:
        _check(self._lib, self._lib.gymrs_sharded_step_many(self._h, self._ptrs(actions_dev), int(stride_bytes), int(n_buffers), int(n_steps),
                                                            1 if use_graph else 0))

    def fill_actions(self, actions_dev: Sequence[int], seed: int, t: int) -> None:
        _check(self._lib, self._lib.gymrs_sharded_fill_actions(self._h, self._ptrs(actions_dev), int(seed), int(t)))

    def rollout(self, n_steps: int, action_seed: int, action_t0: int = 0) -> None:
        """``BatchedEngine.rollout`` on every block: ``n_steps`` random-policy steps fused into one launch per block."""
        _check(self._lib, self._lib.gymrs_sharded_rollout(self._h, int(n_steps), int(action_seed), int(action_t0)))

    def set_params(self, params) -> None:
        """Assign the pub physics fields of every lane of the batch (``gymrs_set_params`` on every block)."""
        if not isinstance(params, _PARAMS[self.kind]):
            raise TypeError(f"expected {_PARAMS[self.kind].__name__}")
        _check(self._lib, self._lib.gymrs_sharded_set_params(self._h, C.byref(params)))
        self.params = type(params).from_buffer_copy(params)

    def sync(self) -> None:
        _check(self._lib, self._lib.gymrs_sharded_sync(self._h))

    def stats(self) -> np.ndarray:
        """{sum_return, sum_length, n_episodes, n_steps} of the WHOLE batch (RCCL all-reduce on distinct devices, host sum otherwise)."""
        out = (C.c_double * 4)()
        _check(self._lib, self._lib.gymrs_sharded_stats(self._h, out))
        return np.array(out[:], dtype=np.float64)

    def stats_clear(self) -> None:
        _check(self._lib, self._lib.gymrs_sharded_stats_clear(self._h))

    @property
    def reduce_path(self) -> str:
        return self._lib.gymrs_sharded_reduce_path(self._h).decode()

    def get_state(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        count = self.n_total - first if count is None else count
        out = np.empty((self.state_dim, count), dtype=np.float32)
        _check(self._lib, self._lib.gymrs_sharded_get_state(self._h, first, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def get_step_result(self, first: int = 0, count: Optional[int] = None):
        count = self.n_total - first if count is None else count
        reward, done, trunc = np.empty(count, np.float32), np.empty(count, np.uint8), np.empty(count, np.uint8)
        _check(self._lib, self._lib.gymrs_sharded_get_step_result(self._h, first, count, reward.ctypes.data_as(C.c_void_p),
                                                                  done.ctypes.data_as(C.c_void_p), trunc.ctypes.data_as(C.c_void_p)))
        return reward, done, trunc
