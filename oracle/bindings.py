"""ctypes bindings of the CPU oracle (oracle/gymrs_oracle.c, f64) and the CPU f32 twin
(oracle/f32_twin.cpp).  TEST INFRASTRUCTURE: used by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only — as the checker / the timed CPU baseline, never as the product path."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_BUILD = _DIR / "build"


def build() -> None:
    subprocess.run(["make", "-C", str(_DIR), "all"], check=True, capture_output=True)


def _load(name: str) -> C.CDLL:
    path = _BUILD / name
    if not path.exists():
        build()
    return C.CDLL(str(path))


class CartPoleParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("gravity", "masscart", "masspole", "length", "force_mag", "tau",
                                          "theta_threshold_radians", "x_threshold")] + [("kinematics_integrator", C.c_int)]


class MountainCarParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("min_position", "max_position", "max_speed", "goal_position",
                                          "goal_velocity", "force", "gravity")]


class PendulumParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("max_speed", "max_torque", "dt", "g", "m", "l")]


class CartPoleEnv(C.Structure):
    _fields_ = [("x", C.c_double), ("x_dot", C.c_double), ("theta", C.c_double), ("theta_dot", C.c_double),
                ("has_steps_beyond", C.c_int), ("steps_beyond", C.c_long)]


class MountainCarEnv(C.Structure):
    _fields_ = [("position", C.c_double), ("velocity", C.c_double)]


class PendulumEnv(C.Structure):
    _fields_ = [("theta", C.c_double), ("theta_dot", C.c_double)]


class Pcg64(C.Structure):
    """rand_pcg::Pcg64 (Lcg128Xsl64): 128-bit state and increment as two u64 each."""
    _fields_ = [("state_lo", C.c_uint64), ("state_hi", C.c_uint64), ("incr_lo", C.c_uint64), ("incr_hi", C.c_uint64)]


class StepResult(C.Structure):
    _fields_ = [("obs", C.c_double * 4), ("reward", C.c_double), ("done", C.c_int), ("truncated", C.c_int)]


_f64 = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


class Oracle:
    """f64 restatement of the reference (see oracle/gymrs_oracle.h for the citations)."""

    def __init__(self):
        L = self.lib = _load("libgymrs_oracle.so")
        L.orc_clip.restype = C.c_double
        L.orc_clip.argtypes = [C.c_double] * 3
        L.orc_clip_i64.restype = C.c_long
        L.orc_clip_i64.argtypes = [C.c_long] * 3
        L.orc_discrete_contains.restype = C.c_int
        L.orc_discrete_contains.argtypes = [C.c_size_t, C.c_size_t]
        L.orc_rand_random_seed.restype = C.c_uint64
        L.orc_rand_random_seed.argtypes = [C.c_int, C.c_uint64, C.c_uint64]
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.orc_u01.restype = C.c_double
        L.orc_u01.argtypes = [C.c_uint32]
        L.orc_cartpole_step.argtypes = [C.POINTER(CartPoleEnv), C.POINTER(CartPoleParams), C.c_size_t, C.POINTER(StepResult)]
        L.orc_cartpole_reset.argtypes = [C.POINTER(CartPoleEnv), C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_mountain_car_step.argtypes = [C.POINTER(MountainCarEnv), C.POINTER(MountainCarParams), C.c_size_t, C.POINTER(StepResult)]
        L.orc_mountain_car_reset.argtypes = [C.POINTER(MountainCarEnv), C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_pendulum_step.argtypes = [C.POINTER(PendulumEnv), C.POINTER(PendulumParams), C.c_double, C.POINTER(StepResult)]
        L.orc_pendulum_reset.argtypes = [C.POINTER(PendulumEnv), C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_cartpole_step_batch.restype = C.c_long
        L.orc_cartpole_step_batch.argtypes = [C.c_size_t, _f64, _f64, _f64, _f64, _u8, _u8, C.POINTER(CartPoleParams), _f64, _u8]
        L.orc_mountain_car_step_batch.restype = C.c_long
        L.orc_mountain_car_step_batch.argtypes = [C.c_size_t, _f64, _f64, _u8, C.POINTER(MountainCarParams), _f64, _u8]
        L.orc_pendulum_step_batch.restype = C.c_long
        L.orc_pendulum_step_batch.argtypes = [C.c_size_t, _f64, _f64, _f64, C.POINTER(PendulumParams), _f64, _f64, _f64]
        L.orc_cartpole_reset_batch.argtypes = [C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, _f64, _f64, _f64, _f64]
        L.orc_mountain_car_reset_batch.argtypes = [C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, _f64, _f64]
        L.orc_pendulum_reset_batch.argtypes = [C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, _f64, _f64]
        L.orc_pcg64_new.argtypes = [C.POINTER(Pcg64)] + [C.c_uint64] * 4
        L.orc_pcg64_from_seed.argtypes = [C.POINTER(Pcg64), C.c_char_p]
        L.orc_pcg64_seed_from_u64.argtypes = [C.POINTER(Pcg64), C.c_uint64]
        L.orc_pcg64_next_u64.restype = C.c_uint64
        L.orc_pcg64_next_u64.argtypes = [C.POINTER(Pcg64)]
        L.orc_uniform_f64_new.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.orc_uniform_f64_sample.restype = C.c_double
        L.orc_uniform_f64_sample.argtypes = [C.POINTER(Pcg64), C.c_double, C.c_double]
        L.orc_cartpole_reset_pcg64.argtypes = [C.POINTER(CartPoleEnv), C.c_uint64, C.c_void_p]
        L.orc_mountain_car_reset_pcg64.argtypes = [C.POINTER(MountainCarEnv), C.c_uint64, C.c_void_p]
        L.orc_baseline_loop.restype = C.c_double
        L.orc_baseline_loop.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_double)]

    # -- params --
    def cartpole_params(self) -> CartPoleParams:
        p = CartPoleParams()
        self.lib.orc_cartpole_default_params(C.byref(p))
        return p

    def mountain_car_params(self) -> MountainCarParams:
        p = MountainCarParams()
        self.lib.orc_mountain_car_default_params(C.byref(p))
        return p

    def pendulum_params(self) -> PendulumParams:
        p = PendulumParams()
        self.lib.orc_pendulum_default_params(C.byref(p))
        return p

    # -- helpers --
    def philox(self, ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        self.lib.orc_philox4x32_10(c, k, o)
        return list(o)

    @staticmethod
    def _bounds(b):
        if b is None:
            return None, None
        arr = np.ascontiguousarray(b, dtype=np.float64)
        return arr, arr.ctypes.data_as(C.c_void_p)

    # -- the reference's own reset stream (Pcg64::seed_from_u64 + Uniform<f64>) --
    def pcg64(self, *, new=None, from_seed=None, seed_from_u64=None) -> Pcg64:
        g = Pcg64()
        if new is not None:
            state, stream = new
            m = (1 << 64) - 1
            self.lib.orc_pcg64_new(C.byref(g), state & m, state >> 64, stream & m, stream >> 64)
        elif from_seed is not None:
            assert len(from_seed) == 32
            self.lib.orc_pcg64_from_seed(C.byref(g), bytes(from_seed))
        else:
            self.lib.orc_pcg64_seed_from_u64(C.byref(g), int(seed_from_u64))
        return g

    def pcg64_next(self, g: Pcg64) -> int:
        return int(self.lib.orc_pcg64_next_u64(C.byref(g)))

    def uniform_f64_scale(self, low: float, high: float):
        """UniformFloat<f64>::new(low, high).scale, or None where the reference panics."""
        out = C.c_double()
        return None if self.lib.orc_uniform_f64_new(low, high, C.byref(out)) else out.value

    def reset_pcg64(self, kind: int, seed: int, bounds=None):
        """The f64 state ``reset(Some(seed), _, bounds)`` of the reference returns (kind 0 CartPole, 1 MountainCar);
        None where it would panic."""
        keep, ptr = self._bounds(bounds)
        if kind == 0:
            e = CartPoleEnv()
            rc = self.lib.orc_cartpole_reset_pcg64(C.byref(e), seed & ((1 << 64) - 1), ptr)
            return None if rc else [e.x, e.x_dot, e.theta, e.theta_dot]
        e = MountainCarEnv()
        rc = self.lib.orc_mountain_car_reset_pcg64(C.byref(e), seed & ((1 << 64) - 1), ptr)
        return None if rc else [e.position, e.velocity]

    # -- scalar envs --
    def cartpole_step(self, env: CartPoleEnv, action: int, params=None):
        r = StepResult()
        p = params or self.cartpole_params()
        rc = self.lib.orc_cartpole_step(C.byref(env), C.byref(p), action, C.byref(r))
        return rc, r

    def mountain_car_step(self, env: MountainCarEnv, action: int, params=None):
        r = StepResult()
        p = params or self.mountain_car_params()
        rc = self.lib.orc_mountain_car_step(C.byref(env), C.byref(p), action, C.byref(r))
        return rc, r

    def pendulum_step(self, env: PendulumEnv, action: float, params=None):
        r = StepResult()
        p = params or self.pendulum_params()
        rc = self.lib.orc_pendulum_step(C.byref(env), C.byref(p), action, C.byref(r))
        return rc, r

    # -- batch (SoA f64 numpy arrays, updated in place) --
    def cartpole_step_batch(self, state, beyond, action, params=None):
        """state: (4, n) f64, beyond: (n,) u8 (in/out), action: (n,) u8 -> reward f64, done u8, n_invalid"""
        n = state.shape[1]
        reward = np.empty(n, np.float64)
        done = np.empty(n, np.uint8)
        p = params or self.cartpole_params()
        bad = self.lib.orc_cartpole_step_batch(n, state[0], state[1], state[2], state[3], beyond,
                                               np.ascontiguousarray(action, np.uint8), C.byref(p), reward, done)
        return reward, done, bad

    def mountain_car_step_batch(self, state, action, params=None):
        n = state.shape[1]
        reward = np.empty(n, np.float64)
        done = np.empty(n, np.uint8)
        p = params or self.mountain_car_params()
        bad = self.lib.orc_mountain_car_step_batch(n, state[0], state[1], np.ascontiguousarray(action, np.uint8),
                                                   C.byref(p), reward, done)
        return reward, done, bad

    def pendulum_step_batch(self, state, action, params=None):
        n = state.shape[1]
        reward = np.empty(n, np.float64)
        oc = np.empty(n, np.float64)
        os_ = np.empty(n, np.float64)
        p = params or self.pendulum_params()
        self.lib.orc_pendulum_step_batch(n, state[0], state[1], np.ascontiguousarray(action, np.float64),
                                         C.byref(p), oc, os_, reward)
        return reward, oc, os_

    def reset_batch(self, kind: int, n: int, gid0: int, seed: int, tick: int, bounds=None) -> np.ndarray:
        keep, bp = self._bounds(bounds)
        if kind == 0:
            st = np.zeros((4, n), np.float64)
            self.lib.orc_cartpole_reset_batch(n, gid0, seed, tick, bp, st[0], st[1], st[2], st[3])
        elif kind == 1:
            st = np.zeros((2, n), np.float64)
            self.lib.orc_mountain_car_reset_batch(n, gid0, seed, tick, bp, st[0], st[1])
        else:
            st = np.zeros((2, n), np.float64)
            self.lib.orc_pendulum_reset_batch(n, gid0, seed, tick, bp, st[0], st[1])
        del keep
        return st

    def baseline_loop(self, kind: int, n_steps: int, max_episode_steps: int = 0, seed: int = 0):
        """Times the reference's caller loop (examples/cartpole.rs:15-30) on one thread."""
        out = (C.c_double * 4)()
        secs = self.lib.orc_baseline_loop(kind, n_steps, max_episode_steps, seed, out)
        return secs, list(out)


class Twin:
    """CPU f32 twin of the batched engine (bit-exact checker)."""

    def __init__(self):
        L = self.lib = _load("libgymrs_f32twin.so")
        L.twin_create.restype = C.c_void_p
        L.twin_create.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32]
        L.twin_destroy.argtypes = [C.c_void_p]
        L.twin_set_params.argtypes = [C.c_void_p, C.c_void_p]
        L.twin_set_params.restype = None
        L.twin_reset.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.twin_step.argtypes = [C.c_void_p, C.c_void_p]
        L.twin_get_state.argtypes = [C.c_void_p, C.c_void_p]
        L.twin_set_state.argtypes = [C.c_void_p, C.c_void_p]
        L.twin_get_obs.argtypes = [C.c_void_p, C.c_void_p]
        L.twin_get_result.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.twin_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.twin_stats_clear.argtypes = [C.c_void_p]
        L.twin_invalid_count.restype = C.c_uint64
        L.twin_invalid_count.argtypes = [C.c_void_p]
        L.twin_fill_actions.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        L.twin_sincosf.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.twin_philox.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.twin_uniform_between.restype = C.c_float
        L.twin_uniform_between.argtypes = [C.c_uint32, C.c_float, C.c_float]
        L.twin_uniform_in_box.restype = C.c_float
        L.twin_uniform_in_box.argtypes = [C.c_uint32, C.c_float, C.c_float]
        L.twin_clipf.restype = C.c_float
        L.twin_clipf.argtypes = [C.c_float] * 3
        L.twin_angle_normalize.restype = C.c_float
        L.twin_angle_normalize.argtypes = [C.c_float]

    def sincosf(self, x):
        x = np.ascontiguousarray(x, np.float32)
        s = np.empty_like(x)
        c = np.empty_like(x)
        self.lib.twin_sincosf(x.size, x.ctypes.data, s.ctypes.data, c.ctypes.data)
        return s, c

    def philox(self, ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        self.lib.twin_philox(c, k, o)
        return list(o)


class TwinEngine:
    _STATE_DIM = {0: 4, 1: 2, 2: 2}
    _OBS_DIM = {0: 4, 1: 2, 2: 3}

    def __init__(self, twin: Twin, kind: int, n: int, params, flags: int = 0, gid0: int = 0):
        """`params` is the PRODUCT's ctypes params struct (gym-rs_amd.engine.*Params layout)."""
        self.lib = twin.lib
        self.kind, self.n = kind, n
        self.h = C.c_void_p(self.lib.twin_create(kind, n, gid0, C.byref(params), flags))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.twin_destroy(self.h)
            self.h = None

    def set_params(self, params):
        self.lib.twin_set_params(self.h, C.byref(params))

    def reset(self, seed: int, bounds=None):
        arr = None if bounds is None else np.ascontiguousarray(bounds, np.float32)
        self.lib.twin_reset(self.h, seed, None if arr is None else arr.ctypes.data)

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.float32 if self.kind == 2 else np.uint8)
        assert a.size == self.n
        self.lib.twin_step(self.h, a.ctypes.data)

    def get_state(self):
        out = np.empty((self._STATE_DIM[self.kind], self.n), np.float32)
        self.lib.twin_get_state(self.h, out.ctypes.data)
        return out

    def set_state(self, st):
        st = np.ascontiguousarray(st, np.float32)
        assert st.shape == (self._STATE_DIM[self.kind], self.n)
        self.lib.twin_set_state(self.h, st.ctypes.data)

    def get_obs(self):
        out = np.empty((self._OBS_DIM[self.kind], self.n), np.float32)
        self.lib.twin_get_obs(self.h, out.ctypes.data)
        return out

    def get_result(self):
        r = np.empty(self.n, np.float32)
        d = np.empty(self.n, np.uint8)
        t = np.empty(self.n, np.uint8)
        self.lib.twin_get_result(self.h, r.ctypes.data, d.ctypes.data, t.ctypes.data)
        return r, d, t

    def stats(self):
        out = (C.c_double * 4)()
        self.lib.twin_stats(self.h, out)
        return np.array(out[:])

    def stats_clear(self):
        self.lib.twin_stats_clear(self.h)

    def invalid_count(self) -> int:
        return self.lib.twin_invalid_count(self.h)

    def fill_actions(self, seed: int, t: int):
        a = np.empty(self.n, np.float32 if self.kind == 2 else np.uint8)
        self.lib.twin_fill_actions(self.h, a.ctypes.data, seed, t)
        return a
