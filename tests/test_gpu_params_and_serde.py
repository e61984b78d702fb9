"""ABI 2 on the GPU: the pub physics fields after construction (gymrs_set_params), the #[derive(Serialize)] view
(gymrs_env_json / gymrs_params_from_json), action buffers of any alignment, the copy probe."""
import ctypes as C
import json
import math

import numpy as np
import pytest
import torch

from oracle.bindings import TwinEngine

pytestmark = pytest.mark.gpu


def test_assigning_a_pub_field_keeps_the_episode_single_env(gymrs):
    """`env.gravity = g` between two step() calls touches nothing but the constant -- in the
    reference steps_beyond_terminated, the state and the PRNG carry on (cartpole.rs:455-464)."""
    env = gymrs.CartPoleEnv()
    env.reset(seed=3)
    env.state = gymrs.CartPoleObservation(2.39, 3.0, 0.0, 0.0)
    assert env.steps_beyond_terminated is None
    r1 = env.step(1)
    assert r1.done and r1.reward == 1.0  # the terminating step pays 1.0 and sets steps_beyond_terminated = Some(0)
    assert env.steps_beyond_terminated == 0
    tick_before = env.rand_random()
    state_before = env.state
    env.gravity = 19.6  # a pub field of the reference struct
    assert env.gravity == 19.6 and env.state == state_before and env.rand_random() == tick_before
    r2 = env.step(1)
    assert r2.done and r2.reward == 0.0, "steps_beyond_terminated was lost by the assignment"
    assert env.steps_beyond_terminated == 1
    env.reset(seed=4)
    assert env.steps_beyond_terminated is None
    # and the new constant is the one the next step used: compare with a fresh env given the same state
    ref = gymrs.CartPoleEnv()
    ref.gravity = 19.6
    ref.reset(seed=3)
    ref.state = state_before
    assert ref.step(1).observation == r2.observation
    env.close()
    ref.close()


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_set_params_mid_run_is_bit_exact_with_the_twin(gymrs, twin, kind):
    n = 3000
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS | gymrs.TIME_LIMIT
    p = gymrs.engine.default_params(kind)
    p.max_episode_steps = 23
    eng = gymrs.BatchedEngine(kind, n, flags=flags, params=p, global_env_offset=999)
    tw = TwinEngine(twin, kind, n, p, flags=flags, gid0=999)
    eng.reset(seed=8)
    tw.reset(8)
    buf = torch.empty(n, dtype=torch.float32 if kind == 2 else torch.uint8, device="cuda:0")

    def run(k, t0):
        for t in range(t0, t0 + k):
            eng.fill_actions(buf.data_ptr(), seed=2, t=t)
            eng.step(buf.data_ptr())
            tw.step(tw.fill_actions(2, t))

    run(17, 0)
    q = type(p).from_buffer_copy(p)
    if kind == 0:
        q.gravity, q.force_mag, q.kinematics_integrator = 4.9, 14.0, 1
    elif kind == 1:
        q.force, q.gravity = 0.0015, 0.002
    else:
        q.g, q.max_torque = 6.0, 1.5
    q.max_episode_steps = 31
    stats_before, tick_before = eng.stats(), eng.tick()
    eng.set_params(q)
    tw.set_params(q)
    assert np.array_equal(eng.stats(), stats_before) and eng.tick() == tick_before  # nothing but the constants moved
    got = eng.get_params()
    assert bytes(got) == bytes(q)
    run(40, 17)
    eng.sync()
    assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32))
    gs, ts = eng.stats(), tw.stats()
    assert np.array_equal(gs[1:], ts[1:]) and gs[0] == pytest.approx(ts[0], rel=1e-6)
    eng.close()


def test_env_json_has_the_reference_fields_and_round_trips(gymrs):
    p = gymrs.engine.default_params(0)
    p.gravity, p.kinematics_integrator, p.max_episode_steps = 9.81, 1, 123
    eng = gymrs.BatchedEngine(0, 10, flags=0, params=p, global_env_offset=40)
    eng.reset(seed=77)
    st = np.zeros((4, 10), np.float32)
    st[:, 3] = [0.25, -1.5, 0.03125, 2.0]
    eng.set_state(st)
    d = json.loads(eng.env_json(3))
    # text level: serde_json prints 10.0 / 0.02 / 9.8, never "1e1"
    text3 = eng.env_json(3)
    assert '"force_mag":10.0' in text3 and '"tau":0.02' in text3 and '"masscart":1.0' in text3 and '"x_threshold":2.4' in text3
    # serde field order of cartpole.rs:52-87 (renderer/screen are GUI-only, rand_random is #[serde(skip_serializing)])
    assert list(d) == ["action_space", "observation_space", "render_mode", "state", "metadata", "gravity", "masscart", "masspole",
                       "length", "force_mag", "tau", "kinematics_integrator", "theta_threshold_radians", "x_threshold",
                       "steps_beyond_terminated", "gymrs"]
    assert d["action_space"] == 2 and d["render_mode"] == "None" and d["kinematics_integrator"] == "Other"
    assert d["state"] == {"x": 0.25, "x_dot": -1.5, "theta": 0.03125, "theta_dot": 2.0}
    assert d["gravity"] == 9.81 and d["theta_threshold_radians"] == 12.0 * 2.0 * math.pi / 360.0 and d["x_threshold"] == 2.4
    hi = d["observation_space"]["high"]
    assert hi["x"] == 4.8 and hi["x_dot"] is None and hi["theta_dot"] is None  # serde_json prints +-inf as null
    assert d["observation_space"]["low"]["theta"] == -hi["theta"]
    assert d["metadata"] == {"render_modes": ["Human", "RgbArray"], "render_fps": 50, "marker": None}
    assert d["steps_beyond_terminated"] is None
    assert d["gymrs"]["global_env_id"] == 43 and d["gymrs"]["seed"] == 77 and d["gymrs"]["max_episode_steps"] == 123
    # steps_beyond_terminated becomes Some(..) after a terminating step (no auto-reset)
    st[:, 3] = [2.39, 3.0, 0.0, 0.0]
    eng.set_state(st)
    eng.step_host(np.ones(10, np.uint8))
    assert json.loads(eng.env_json(3))["steps_beyond_terminated"] == 0 and json.loads(eng.env_json(2))["steps_beyond_terminated"] is None
    # round trip: JSON -> params (+ state) -> a second engine prints the same physics fields
    text = eng.env_json(3)
    q, state = gymrs.params_from_json(0, text)
    assert bytes(q) == bytes(eng.get_params()) and state == [float(v) for v in eng.get_state()[:, 3]]
    eng2 = gymrs.BatchedEngine(0, 1, flags=0, params=q)
    d1, d2 = json.loads(text), json.loads(eng2.env_json(0))
    for key in ("gravity", "masscart", "masspole", "length", "force_mag", "tau", "kinematics_integrator",
                "theta_threshold_radians", "x_threshold", "observation_space", "action_space", "metadata"):
        assert d1[key] == d2[key], key
    # the buffer-size protocol of the C entry point
    lib = gymrs.load_library()
    need = C.c_uint64()
    small = C.create_string_buffer(8)
    assert lib.gymrs_env_json(eng._h, 3, small, 8, C.byref(need)) == 1 and need.value == len(text) + 1
    assert lib.gymrs_env_json(eng._h, 10, small, 8, None) == 1  # lane out of range
    eng.close()
    eng2.close()
    # MountainCar: declaration order of mountain_car.rs:48-80
    with gymrs.BatchedEngine(1, 4, flags=0) as mc:
        d = json.loads(mc.env_json(0))
        assert list(d) == ["min_position", "max_position", "max_speed", "goal_position", "goal_velocity", "force", "gravity",
                           "render_mode", "action_space", "observation_space", "state", "metadata", "gymrs"]
        assert d["action_space"] == 3 and d["metadata"]["render_fps"] == 30 and d["force"] == 0.001 and d["state"]["velocity"] == 0.0
        assert d["observation_space"] == {"low": {"position": -1.2, "velocity": -0.07}, "high": {"position": 0.6, "velocity": 0.07}}
        q, state = gymrs.params_from_json(1, mc.env_json(0))
        assert bytes(q) == bytes(mc.get_params()) and len(state) == 2
    env = gymrs.CartPoleEnv()
    assert json.loads(env.to_json())["gravity"] == 9.8
    env.close()


@pytest.mark.parametrize("kind", [0, 2])
def test_action_buffers_of_any_alignment(gymrs, twin, kind):
    """the ABI states no alignment for actions_dev; a ring with stride = n (n odd) puts every other slot on
    an odd address.  Misaligned buffers are read lane by lane and give the same bits."""
    n, steps, nbuf = 4099, 24, 5
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
    eng = gymrs.BatchedEngine(kind, n, flags=flags)
    tw = TwinEngine(twin, kind, n, gymrs.engine.default_params(kind), flags=flags)
    eng.reset(seed=6)
    tw.reset(6)
    dtype = torch.float32 if kind == 2 else torch.uint8
    esz = 4 if kind == 2 else 1
    pool = torch.zeros(nbuf * n + 8, dtype=dtype, device="cuda:0")
    off = 1  # elements: the ring starts one action past an aligned address
    acts = [tw.fill_actions(3, b) for b in range(nbuf)]
    for b in range(nbuf):
        pool[off + b * n: off + (b + 1) * n] = torch.from_numpy(np.ascontiguousarray(acts[b])).to("cuda:0")
    torch.cuda.synchronize()  # (the slice assignments are kernels on torch's stream; the engine reads on its own non-blocking stream)
    base = pool.data_ptr() + off * esz
    assert base % (4 * esz) != 0
    eng.step_many(base, n * esz, nbuf, steps)
    for t in range(steps):
        tw.step(acts[t % nbuf])
    eng.sync()
    assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32))
    assert np.array_equal(eng.stats(), tw.stats()) or kind == 2
    # single misaligned step through gymrs_step as well
    eng.step(base)
    tw.step(acts[0])
    eng.sync()
    assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32))
    eng.close()


def test_copy_probe_reports_a_plausible_floor(gymrs):
    """tools/copy_probe (a measurement tool since ABI 3; bench.py's copy floor): HIP launches and launches of a chain of the dispatcher."""
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("copy_probe_build", Path(__file__).resolve().parent.parent / "tools" / "copy_probe" / "build.py")
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    lib = tool.load()
    us = C.c_double()
    n = 1 << 20
    assert lib.gymrs_tool_copy_probe(0, 17 * n, 21 * n, 300, 1, C.byref(us)) == 0, lib.gymrs_tool_copy_probe_error()
    assert 2.0 < us.value < 12.0  # round 1 measured 5.0-5.5 us for this footprint
    assert lib.gymrs_tool_copy_probe(0, 1 << 29, 1 << 29, 10, 0, C.byref(us)) == 0
    gbps = 2 * (1 << 29) / (us.value * 1e-6) / 1e9
    assert 3000.0 < gbps < 8000.0  # HBM3E: 8 TB/s peak, ~6.3 TB/s for a float4 copy
    assert lib.gymrs_tool_copy_probe(99, 16, 16, 1, 0, C.byref(us)) == 1 and b"device index" in lib.gymrs_tool_copy_probe_error()
    # mode | 16: a source of zeros (what the probe copied before round 4's last evidence set); no other bit above 8 exists
    assert lib.gymrs_tool_copy_probe(0, 17 * n, 21 * n, 50, 16 | 1, C.byref(us)) == 0 and 2.0 < us.value < 12.0
    assert lib.gymrs_tool_copy_probe(0, 17 * n, 21 * n, 50, 32, C.byref(us)) == 1
    # through a chain of the dispatcher (the tool's own build of it, against the tool's own code object): faster than with a release per launch
    hip_us = tool.copy_probe(0, 17 * n, 21 * n, 300, 0)
    chain_us = tool.copy_probe(0, 17 * n, 21 * n, 300, 2)
    assert chain_us is None or 1.5 < chain_us < hip_us * 1.05, (chain_us, hip_us, lib.gymrs_tool_copy_probe_error())
