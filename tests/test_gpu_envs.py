"""The single-env compatibility layer on a real GPU: reads like a test of the reference crate —
`CartPoleEnv::new(RenderMode::None)`, `env.step(action)`, `env.reset(seed, return_info, options)`,
`env.action_space()`, `env.observation_space()` (core.rs:25-90) — with one lane of the batched engine
behind it.  Expected values: tests/golden."""
import math

import numpy as np

import pytest

pytestmark = pytest.mark.gpu


def test_cartpole_env_surface_and_kat(gymrs, golden):
    env = gymrs.CartPoleEnv(gymrs.RenderMode.NONE)
    assert env.action_space() == gymrs.Discrete(2)
    space = env.observation_space()
    assert space.high.to_vec() == [4.8, math.inf, 0.41887902047863906, math.inf] and space.low == -space.high
    assert env.gravity == 9.8 and env.masspole == 0.1 and env.length == 0.5 and env.tau == 0.02  # pub fields
    assert env.reward_range() == gymrs.RewardRange() and env.render_mode() is gymrs.RenderMode.NONE
    obs, info = env.reset(seed=64, return_info=True)
    assert info == () and all(abs(v) < 0.05 for v in obs.to_vec())
    obs2, info2 = env.reset(seed=64)
    assert info2 is None and obs2 == obs  # same seed, same state
    for tr in golden("cartpole")["trajectories"]:
        policy = {"always_1": lambda t: 1, "always_0": lambda t: 0, "alternate_1_0": lambda t: (t + 1) % 2}[tr["policy"]]
        env.reset(seed=0)  # clears steps_beyond_terminated (cartpole.rs:504); assigning `state` does not
        env.state = gymrs.CartPoleObservation(*tr["start"])
        t, total = 0, 0.0
        while True:
            ar = env.step(policy(t))
            assert ar.truncated is False and ar.info == ()  # cartpole.rs:480-481
            total += ar.reward
            t += 1
            if ar.done:
                break
        assert t == tr["steps"] and total == float(tr["steps"])  # 10 / 9 / 60
        assert ar.observation.to_vec() == pytest.approx(tr["final"], rel=2e-4)
    # stepping past termination pays 1.0 once, then 0.0 (cartpole.rs:455-464)
    bt = golden("cartpole")["beyond_terminated"]
    env.reset(seed=0)
    env.state = gymrs.CartPoleObservation(*bt["start"])
    rewards = [env.step(bt["action"]).reward for _ in bt["rewards"]]
    assert rewards == bt["rewards"]
    with pytest.raises(AssertionError):  # assert!(self.action_space.contains(action)) cartpole.rs:402-406
        env.step(2)
    env.close()


def test_cartpole_first_step_matches_known_answer(gymrs):
    env = gymrs.CartPoleEnv()
    env.state = gymrs.CartPoleObservation(0.01, 0.02, 0.03, 0.04)
    ar = env.step(1)
    # SURVEY Appendix C; 0.356 (reference, polemass_length = 0.6), not Gym's 0.215 (Q1)
    assert ar.observation.to_vec() == pytest.approx([0.0104, 0.35615076996399875, 0.0308, -0.2430694901285738], rel=1e-6)
    assert ar.reward == 1.0 and ar.done is False
    env.close()


def test_mountain_car_env_surface_and_kat(gymrs, golden):
    env = gymrs.MountainCarEnv(gymrs.RenderMode.NONE)
    assert env.action_space() == gymrs.Discrete(3)
    space = env.observation_space()
    assert space.low.to_vec() == [-1.2, -0.07] and space.high.to_vec() == [0.6, 0.07]
    obs, _ = env.reset(seed=1)
    assert -0.6 <= obs.position < -0.4 and obs.velocity == 0.0  # mountain_car.rs:162-167
    tr = golden("mountain_car")["trajectories"][0]
    env.state = gymrs.MountainCarObservation(*tr["start"])
    t, total = 0, 0.0
    while True:
        ar = env.step(2 if env.state.velocity >= 0 else 0)
        assert ar.info is None and ar.truncated is False  # mountain_car.rs:432-433
        total += ar.reward
        t += 1
        if ar.done:
            break
    assert t == 124 and total == -124.0
    assert ar.observation.to_vec() == pytest.approx(tr["final"], rel=1e-4)
    env.state = gymrs.MountainCarObservation(-1.19, -0.07)
    ar = env.step(0)
    assert ar.observation.to_vec() == [pytest.approx(-1.2), 0.0] and not ar.done  # wall rule
    with pytest.raises(AssertionError):
        env.step(3)
    env.close()


def test_pendulum_env_spec(gymrs, golden):
    env = gymrs.PendulumEnv()
    tr = golden("pendulum")["trajectory"]
    env.state = tr["start"]
    ret = 0.0
    for t in range(tr["steps"]):
        ar = env.step(2.0 if (t // 10) % 2 == 0 else -2.0)
        ret += ar.reward
        assert not ar.done
    assert ret == pytest.approx(tr["total_reward"], rel=1e-4)
    assert list(env.state) == pytest.approx(tr["final"], rel=1e-3)
    env.close()


def test_mutable_physics_fields(gymrs):
    env = gymrs.CartPoleEnv()
    env.state = gymrs.CartPoleObservation(0.0, 0.0, 0.05, 0.0)
    a = env.step(1).observation
    env.state = gymrs.CartPoleObservation(0.0, 0.0, 0.05, 0.0)
    env.force_mag = 20.0  # pub field (cartpole.rs:61)
    b = env.step(1).observation
    assert b.x_dot > a.x_dot * 1.5
    env.close()


def test_cpp_trait_mirror(tmp_path):
    """include/gymrs_env.hpp (the C++ host side above the C ABI) on the GPU: tests/cpp/test_env_mirror.cpp."""
    import spawn_server
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "test_env_mirror"
    lib_dir = root / "gym-rs_amd"
    spawn_server.run(["g++", "-std=c++17", "-O1", f"-I{root / 'include'}", str(root / "tests" / "cpp" / "test_env_mirror.cpp"),
                    "-o", str(exe), f"-L{lib_dir}", "-lgymrs_amd", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"],
                   check=True, capture_output=True, text=True)
    res = spawn_server.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "CPP_MIRROR_OK" in res.stdout, res.stdout + res.stderr


# ---- Clone + Serialize (core.rs:25) ------------------------------------------------------------
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_clone_and_snapshot_continue_bit_identically(gymrs, kind):
    """A clone, and an engine restored from a snapshot, continue exactly like the original: same states,
    same auto-reset draws (RNG position = seed + tick), same statistics."""
    n = 10_007
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS | gymrs.TIME_LIMIT
    p = gymrs.engine.default_params(kind)
    p.max_episode_steps = 25
    if kind == 0:
        p.gravity = 9.9  # non-default constants travel with the snapshot
    a = gymrs.BatchedEngine(kind, n, global_env_offset=77, flags=flags, params=p)
    a.reset(seed=31, options=None)
    a.rollout(30, action_seed=5, action_t0=0)
    blob = a.snapshot()
    b = a.clone()
    c = gymrs.BatchedEngine(kind, n, flags=flags)  # default constants, other shard offset: all overwritten
    c.restore(blob)
    for eng in (a, b, c):
        eng.rollout(40, action_seed=5, action_t0=30)
    ref_state, ref_stats, ref_res = a.get_state(), a.stats(), a.get_step_result()
    for eng in (b, c):
        assert np.array_equal(eng.get_state().view(np.uint32), ref_state.view(np.uint32))
        assert np.array_equal(eng.stats(), ref_stats)
        for x, y in zip(eng.get_step_result(), ref_res):
            assert np.array_equal(x, y)
        assert eng.tick() == a.tick()
    assert ref_stats[2] > 0
    for eng in (a, b, c):
        eng.close()


def test_snapshot_rejects_a_mismatched_engine(gymrs):
    with gymrs.BatchedEngine(0, 100, flags=gymrs.AUTO_RESET) as a, gymrs.BatchedEngine(0, 101, flags=gymrs.AUTO_RESET) as b, \
            gymrs.BatchedEngine(1, 100, flags=gymrs.AUTO_RESET) as c:
        a.reset(seed=1)
        blob = a.snapshot()
        for other in (b, c):
            with pytest.raises(gymrs.GymrsError):
                other.restore(blob)
        with pytest.raises(gymrs.GymrsError):
            a.restore(blob[:64])
        with pytest.raises(gymrs.GymrsError):
            a.restore(b"NOTASNAP" + blob[8:])


def test_closed_loop_example_runs_and_balances(tmp_path):
    """examples/closed_loop_policy.py: zero-copy observation columns in torch, a linear policy on the engine's
    stream.  The controller keeps the pole up far longer than the ~22 steps of a random policy."""
    import re
    import spawn_server
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    out = spawn_server.run([sys.executable, str(root / "examples" / "closed_loop_policy.py"), "--n-envs", "8192", "--steps", "600"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1500:]
    m = re.search(r"finished episodes: (\d+), mean return ([0-9.]+)", out.stdout)
    assert m, out.stdout
    episodes, mean_return = int(m.group(1)), float(m.group(2))
    assert episodes == 0 or mean_return > 100.0, out.stdout


def test_reference_example_loops_run_through_the_mirrors():
    """examples/cartpole.py and examples/mountain_car.py are the caller loops of the reference's examples/cartpole.rs:7-33
    and examples/mountain_car.rs:8-40 (incl. stepping after close(), which only drops the GUI in the reference)."""
    import importlib.util
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent

    def load(name):
        spec = importlib.util.spec_from_file_location(f"example_{name}", root / "examples" / f"{name}.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    cp = load("cartpole")
    rewards = cp.single_env(n_episodes=4, seed=1)
    assert len(rewards) == 4 and all(8.0 <= r <= 200.0 and r == int(r) for r in rewards)  # a random policy lasts ~22 steps
    b = cp.batched(n_envs=1 << 16, steps=100)
    assert b["steps"] == (1 << 16) * 100 and b["finished_episodes"] > (1 << 16) * 2 and 15.0 < b["mean_return"] < 30.0
    mc = load("mountain_car")
    total = mc.main(seed=3, verbose=False)
    assert 201 <= total <= 401  # the episode loop stops at done or after 201 steps; 200 more steps after close()
    # the same loop for ONE batch cut into 3 blocks through the C ABI's native sharder (examples/sharded_cartpole.py): the batch's statistics = one engine's
    sc = load("sharded_cartpole")
    got = sc.main(lanes_per_gpu=1 << 15, blocks=3, steps=100, ring=4)
    import importlib

    import torch

    gymrs = importlib.import_module("gym-rs_amd")
    one = gymrs.BatchedEngine(gymrs.CARTPOLE, 3 << 15, flags=gymrs.AUTO_RESET | gymrs.TRACK_STATS)
    one.reset(seed=0)
    ring = torch.empty((4, 3 << 15), dtype=torch.uint8, device="cuda:0")
    for b_ in range(4):
        one.fill_actions(ring[b_].data_ptr(), seed=1, t=b_)
    one.step_many(ring.data_ptr(), 3 << 15, 4, 100)
    one.sync()
    assert list(got) == list(one.stats()) and got[3] == (3 << 15) * 100 and got[2] > 0
    one.close()
