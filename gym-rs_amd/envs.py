"""Single-env compatibility layer: the reference's per-env structs and ``Env`` trait surface
(``/root/reference/src/core.rs:25-90``) over a ONE-lane GPU engine.

``CartPoleEnv`` / ``MountainCarEnv`` keep the reference's names, constructor argument, public physics
fields, method names, argument meaning and error behaviour (an invalid action raises, like the
reference's ``assert!``), so a test written against gym-rs reads the same here.  Every ``step`` is one
kernel launch on one lane plus a small device->host copy: this is the plumbing configuration
(BASELINE.json configs[0]), not the fast path — use ``BatchedEngine`` for throughput.
Rendering (``src/utils/renderer.rs``, ``screen.rs``) is out of scope: only ``RenderMode.NONE`` exists.
"""
from __future__ import annotations

import enum
import math
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from .core import ActionReward, RewardRange
from .engine import CARTPOLE, MOUNTAIN_CAR, PENDULUM, BatchedEngine, InvalidActionError, default_params
from .spaces import BoxR, Discrete


class RenderMode(enum.Enum):
    """``RenderMode`` (utils/renderer.rs:83-114).  Only ``NONE`` is supported (GUI is out of scope)."""

    HUMAN = "human"
    SINGLE_RGB_ARRAY = "single_rgb_array"
    RGB_ARRAY = "rgb_array"
    ANSI = "ansi"
    NONE = "none"


@dataclass(frozen=True)
class CartPoleObservation:
    """cartpole.rs:328-334; ``Into<Vec<f64>>`` order x, x_dot, theta, theta_dot (cartpole.rs:336-349)."""

    x: float
    x_dot: float
    theta: float
    theta_dot: float

    def to_vec(self):
        return [self.x, self.x_dot, self.theta, self.theta_dot]

    def __neg__(self):  # cartpole.rs:367-378
        return CartPoleObservation(-self.x, -self.x_dot, -self.theta, -self.theta_dot)


@dataclass(frozen=True)
class MountainCarObservation:
    """mountain_car.rs:122-128; ``Into<Vec<f64>>`` order position, velocity (mountain_car.rs:193-197)."""

    position: float
    velocity: float

    def to_vec(self):
        return [self.position, self.velocity]


@dataclass(frozen=True)
class PendulumObservation:
    """Spec-derived (Gym Pendulum-v1): (cos theta, sin theta, theta_dot)."""

    cos_theta: float
    sin_theta: float
    theta_dot: float

    def to_vec(self):
        return [self.cos_theta, self.sin_theta, self.theta_dot]


@dataclass(frozen=True)
class Metadata:
    """``Metadata<T>{render_modes, render_fps}`` (utils/custom/structs.rs:12-19)."""

    render_modes: tuple
    render_fps: int


class _SingleEnv:
    """Shared plumbing: a one-lane engine with the reference's semantics (no auto-reset, no
    truncation: SURVEY Q2, Q3)."""

    _KIND = -1
    _FIELDS: tuple = ()

    def __init__(self, render_mode: RenderMode = RenderMode.NONE, *, device: int = 0, reset_rng: str = "philox"):
        """``reset_rng``: "philox" (the build's counter-based stream, north_star) or "pcg64": ``reset(seed)`` then
        returns, rounded to f32, the state the reference's ``reset(Some(seed))`` returns (its own
        ``Pcg64::seed_from_u64`` + ``Uniform`` chain, ``BatchedEngine.reset_pcg64``)."""
        if render_mode is not RenderMode.NONE:
            raise NotImplementedError("rendering is out of scope for the MI355X hot path; use RenderMode.NONE")
        if reset_rng not in ("philox", "pcg64") or (reset_rng == "pcg64" and self._KIND == PENDULUM):
            raise ValueError("reset_rng is 'philox' or, for CartPole / MountainCar, 'pcg64'")
        object.__setattr__(self, "_reset_rng", reset_rng)
        object.__setattr__(self, "_params", default_params(self._KIND))
        object.__setattr__(self, "_engine", BatchedEngine(self._KIND, 1, params=self._params, flags=0, device=device))
        object.__setattr__(self, "_render_mode", render_mode)

    # the reference's physics constants are `pub` fields (cartpole.rs:53-82, mountain_car.rs:49-77)
    def __getattr__(self, name):
        if name in type(self)._FIELDS:
            return getattr(self._params, name)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in type(self)._FIELDS:
            # Assigning a pub field of the reference struct touches nothing else (cartpole.rs:455-464 reads the fields
            # afresh on every step): gymrs_set_params swaps only the launch constants -- the engine, its device, the
            # state, steps_beyond_terminated, seed/tick and the last reward/done all carry on.
            setattr(self._params, name, value)
            self._engine.set_params(self._params)
        else:
            object.__setattr__(self, name, value)

    def to_json(self) -> str:
        """``serde_json::to_string(&env)``: the serde-visible fields under the reference's names (core.rs:25)."""
        return self._engine.env_json(0)

    def _options(self, options: Optional[BoxR]):
        if options is None:
            return None
        return list(options.low.to_vec()) + list(options.high.to_vec())

    def _reset_lane(self, seed: Optional[int], options: Optional[BoxR]) -> None:
        if self._reset_rng == "pcg64":
            self._engine.reset_pcg64(seed, options=self._options(options))
        else:
            self._engine.reset(seed, self._options(options))

    def render(self, mode: RenderMode = RenderMode.NONE):
        """``Env::render`` (core.rs:53): with RenderMode::None nothing is drawn (renderer.rs:52-62)."""
        return None

    def close(self) -> None:
        """``Env::close`` (core.rs:56; cartpole.rs:534-536, mountain_car.rs:503-505): the reference only drops its GUI
        handle (screen.rs:80-82) and the env stays steppable -- examples/mountain_car.rs:30-37 steps 200 more times
        after ``close()``.  There is no GUI here, so this is a no-op; ``release()`` (or dropping the object) frees the
        GPU lane."""

    def release(self) -> None:
        """Free the device memory of this env's one-lane engine (not part of the reference's surface)."""
        self._engine.close()

    def render_mode(self) -> RenderMode:
        return self._render_mode

    def reward_range(self) -> RewardRange:
        """core.rs:80-82: the default (-inf, inf)."""
        return RewardRange()

    def rand_random(self) -> Tuple[int, int]:
        """core.rs:73 returns ``&Pcg64``; this build's generator is counter-based Philox4x32-10, whose
        whole state is (seed, tick) — the one unavoidable deviation of the API."""
        tick, seed = self._engine.tick()
        return seed, tick


class CartPoleEnv(_SingleEnv):
    """``CartPoleEnv`` (cartpole.rs:51-87) on one GPU lane."""

    _KIND = CARTPOLE
    _FIELDS = ("gravity", "masscart", "masspole", "length", "force_mag", "tau", "theta_threshold_radians",
               "x_threshold", "kinematics_integrator")

    def __init__(self, render_mode: RenderMode = RenderMode.NONE, **kw):
        super().__init__(render_mode, **kw)
        self.metadata = Metadata((RenderMode.HUMAN, RenderMode.RGB_ARRAY), 50)  # cartpole.rs:265-271

    @property
    def state(self) -> CartPoleObservation:
        return CartPoleObservation(*[float(v) for v in self._engine.get_state()[:, 0]])

    @state.setter
    def state(self, obs: CartPoleObservation) -> None:
        self._engine.set_state(np.array(obs.to_vec(), dtype=np.float32).reshape(4, 1))

    def action_space(self) -> Discrete:
        return Discrete(2)  # cartpole.rs:114

    def observation_space(self) -> BoxR:
        high = CartPoleObservation(self.x_threshold * 2.0, math.inf, self.theta_threshold_radians * 2.0, math.inf)
        return BoxR(-high, high)  # cartpole.rs:105-115

    def step(self, action: int) -> ActionReward:
        """``Env::step`` (cartpole.rs:398-483)."""
        if not self.action_space().contains(action):  # cartpole.rs:402-406
            raise InvalidActionError(5, f"{action} usize invalid")
        self._engine.step_host([action])
        reward, done, _ = self._engine.get_step_result()
        if done[0]:  # cartpole.rs:455-464: None -> Some(0) on the terminating step, then counted up (with a warning)
            beyond = self.steps_beyond_terminated
            object.__setattr__(self, "_steps_beyond", 0 if beyond is None else beyond + 1)
        # truncated is hard-coded false, info is Some(()) (cartpole.rs:480-481)
        return ActionReward(self.state, float(reward[0]), bool(done[0]), False, ())

    @property
    def steps_beyond_terminated(self) -> Optional[int]:
        """The pub field of cartpole.rs:83 (``Option<usize>``): None until the episode terminated, then the number of
        steps taken after that.  The device keeps ``is_some()`` (that is all ``step`` reads); the count is kept here."""
        return getattr(self, "_steps_beyond", None)

    def reset(self, seed: Optional[int] = None, return_info: bool = False, options: Optional[BoxR] = None):
        """``Env::reset`` (cartpole.rs:485-516)."""
        self._reset_lane(seed, options)
        object.__setattr__(self, "_steps_beyond", None)  # cartpole.rs:504
        return self.state, (() if return_info else None)


class MountainCarEnv(_SingleEnv):
    """``MountainCarEnv`` (mountain_car.rs:46-84) on one GPU lane."""

    _KIND = MOUNTAIN_CAR
    _FIELDS = ("min_position", "max_position", "max_speed", "goal_position", "goal_velocity", "force", "gravity")

    def __init__(self, render_mode: RenderMode = RenderMode.NONE, **kw):
        super().__init__(render_mode, **kw)
        self.metadata = Metadata((RenderMode.HUMAN, RenderMode.RGB_ARRAY, RenderMode.SINGLE_RGB_ARRAY, RenderMode.NONE), 30)

    @property
    def state(self) -> MountainCarObservation:
        return MountainCarObservation(*[float(v) for v in self._engine.get_state()[:, 0]])

    @state.setter
    def state(self, obs: MountainCarObservation) -> None:
        self._engine.set_state(np.array(obs.to_vec(), dtype=np.float32).reshape(2, 1))

    def action_space(self) -> Discrete:
        return Discrete(3)  # mountain_car.rs:362

    def observation_space(self) -> BoxR:
        return BoxR(MountainCarObservation(self.min_position, -self.max_speed),
                    MountainCarObservation(self.max_position, self.max_speed))  # mountain_car.rs:353-364

    def step(self, action: int) -> ActionReward:
        """``Env::step`` (mountain_car.rs:398-435)."""
        if not self.action_space().contains(action):  # mountain_car.rs:402-406
            raise InvalidActionError(5, f"{action} (usize) invalid")
        self._engine.step_host([action])
        reward, done, _ = self._engine.get_step_result()
        return ActionReward(self.state, float(reward[0]), bool(done[0]), False, None)  # info: None (mountain_car.rs:433)

    def reset(self, seed: Optional[int] = None, return_info: bool = False, options: Optional[BoxR] = None):
        """``Env::reset`` (mountain_car.rs:464-501)."""
        self._reset_lane(seed, options)
        return self.state, (() if return_info else None)


class PendulumEnv(_SingleEnv):
    """Gym Pendulum-v1 in gym-rs style (spec-derived; NOT in the reference)."""

    _KIND = PENDULUM
    _FIELDS = ("max_speed", "max_torque", "dt", "g", "m", "l")

    def __init__(self, render_mode: RenderMode = RenderMode.NONE, **kw):
        super().__init__(render_mode, **kw)
        self.metadata = Metadata((RenderMode.HUMAN, RenderMode.RGB_ARRAY), 30)

    @property
    def state(self) -> Tuple[float, float]:
        th, thd = self._engine.get_state()[:, 0]
        return float(th), float(thd)

    @state.setter
    def state(self, value) -> None:
        self._engine.set_state(np.array(value, dtype=np.float32).reshape(2, 1))

    def action_space(self) -> BoxR:
        return BoxR(-self.max_torque, self.max_torque)

    def observation_space(self) -> BoxR:
        return BoxR(PendulumObservation(-1.0, -1.0, -self.max_speed), PendulumObservation(1.0, 1.0, self.max_speed))

    def _obs(self) -> PendulumObservation:
        return PendulumObservation(*[float(v) for v in self._engine.get_obs()[:, 0]])

    def step(self, action: float) -> ActionReward:
        self._engine.step_host([action])
        reward, done, _ = self._engine.get_step_result()
        return ActionReward(self._obs(), float(reward[0]), bool(done[0]), False, None)

    def _options(self, options):
        """``options``: a BoxR over the (theta, theta_dot) state -- a pair of 2-sequences -- or a flat
        [theta_low, theta_dot_low, theta_high, theta_dot_high] list."""
        if options is None:
            return None
        if isinstance(options, BoxR):
            return [float(v) for v in options.low] + [float(v) for v in options.high]
        return [float(v) for v in options]

    def reset(self, seed: Optional[int] = None, return_info: bool = False, options=None):
        self._engine.reset(seed, self._options(options))
        return self._obs(), (() if return_info else None)
