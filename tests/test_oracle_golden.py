"""The C f64 oracle against the golden vectors (tests/golden/*.json, produced by the independent
Python evaluation in tests/golden/make_golden.py; values equal SURVEY.md Appendix C).

Parity status: the reference has no golden vectors for step()/reset(), so these pin the oracle to
the SOURCE TEXT of the reference, restated twice independently (Python and C)."""
import math

import numpy as np
import pytest

from oracle.bindings import CartPoleEnv, MountainCarEnv, PendulumEnv


def ulps(a, b):
    if a == b:
        return 0.0
    return abs(a - b) / np.spacing(max(abs(a), abs(b)))


def test_cartpole_constants(golden, oracle):
    c = golden("cartpole")["constants"]
    p = oracle.cartpole_params()
    for k, v in c.items():
        assert getattr(p, k) == v, k
    assert p.kinematics_integrator == 0
    assert p.theta_threshold_radians == 0.20943951023931953  # cartpole.rs:102


def test_cartpole_single_steps(golden, oracle):
    cp = golden("cartpole")
    assert len(cp["single_steps_f32_inputs"]) == 512  # (round 5: states exact in f32, for the GPU's 1e-6 test; same evaluation, same bound here)
    for case in cp["single_steps"] + cp["single_steps_f32_inputs"]:
        e = CartPoleEnv(*case["state"], 0, 0)
        rc, r = oracle.cartpole_step(e, case["action"])
        assert rc == 0
        for got, want in zip(r.obs, case["next"]):
            assert ulps(got, want) <= 2, (case, list(r.obs))  # libm pow(x,2) vs x*x: <= 1 ulp each
        assert r.reward == case["reward"] and bool(r.done) is case["done"] and r.truncated == 0
        assert [e.x, e.x_dot, e.theta, e.theta_dot] == list(r.obs)  # state is updated in place


def test_cartpole_known_answers_from_survey(oracle):
    # SURVEY.md Appendix C, single step from (0.01, 0.02, 0.03, 0.04)
    e = CartPoleEnv(0.01, 0.02, 0.03, 0.04, 0, 0)
    _, r = oracle.cartpole_step(e, 1)
    assert list(r.obs) == pytest.approx([0.0104, 0.35615076996399875, 0.030799999999999998, -0.2430694901285738], rel=1e-15)
    # the Gym formula (polemass_length = masspole*length) would give x_dot = 0.2146...: Q1 reproduced, not fixed
    assert abs(r.obs[1] - 0.21467919574755523) > 0.1


def test_cartpole_semi_implicit(golden, oracle):
    p = oracle.cartpole_params()
    p.kinematics_integrator = 1
    for case in golden("cartpole")["semi_implicit"]:
        e = CartPoleEnv(*case["state"], 0, 0)
        _, r = oracle.cartpole_step(e, case["action"], p)
        for got, want in zip(r.obs, case["next"]):
            assert ulps(got, want) <= 2


def test_cartpole_trajectories(golden, oracle):
    policies = {"always_1": lambda t: 1, "always_0": lambda t: 0, "alternate_1_0": lambda t: (t + 1) % 2}
    for tr in golden("cartpole")["trajectories"]:
        e = CartPoleEnv(*tr["start"], 0, 0)
        t, total = 0, 0.0
        while True:
            _, r = oracle.cartpole_step(e, policies[tr["policy"]](t))
            total += r.reward
            t += 1
            if r.done:
                break
        assert t == tr["steps"]  # 10 / 9 / 60
        assert total == tr["total_reward"] == float(t)  # the terminal step pays 1.0 (Q3)
        assert list(r.obs) == pytest.approx(tr["final"], rel=1e-12)


def test_cartpole_steps_beyond_terminated(golden, oracle):
    bt = golden("cartpole")["beyond_terminated"]
    e = CartPoleEnv(*bt["start"], 0, 0)
    rewards, dones = [], []
    for _ in bt["rewards"]:
        _, r = oracle.cartpole_step(e, bt["action"])
        rewards.append(r.reward)
        dones.append(bool(r.done))
    assert rewards == bt["rewards"] and dones == bt["dones"]  # 1,...,1 (terminal),0,0,...
    assert e.has_steps_beyond == 1 and e.steps_beyond == sum(1 for x in rewards if x == 0.0)


def test_cartpole_invalid_action_panics_before_touching_state(oracle):
    e = CartPoleEnv(0.01, 0.02, 0.03, 0.04, 0, 0)
    rc, _ = oracle.cartpole_step(e, 2)
    assert rc == -1 and [e.x, e.x_dot, e.theta, e.theta_dot] == [0.01, 0.02, 0.03, 0.04]


def test_cartpole_nan_counts_as_done(oracle):
    e = CartPoleEnv(math.nan, 0.0, 0.0, 0.0, 0, 0)
    _, r = oracle.cartpole_step(e, 0)
    assert r.done == 1  # OrderedFloat: NaN > x_threshold


def test_mountain_car_single_steps_and_wall(golden, oracle):
    p = oracle.mountain_car_params()
    for k, v in golden("mountain_car")["constants"].items():
        assert getattr(p, k) == v
    for case in golden("mountain_car")["single_steps"] + golden("mountain_car")["single_steps_f32_inputs"]:
        e = MountainCarEnv(*case["state"])
        rc, r = oracle.mountain_car_step(e, case["action"])
        assert rc == 0
        assert [r.obs[0], r.obs[1]] == case["next"], case  # bit-exact: same libm cos, same op order
        assert r.reward == -1.0 and bool(r.done) is case["done"] and r.truncated == 0
    e = MountainCarEnv(-1.19, -0.07)
    _, r = oracle.mountain_car_step(e, 0)
    assert (r.obs[0], r.obs[1], r.done) == (-1.2, 0.0, 0)  # wall rule (mountain_car.rs:418-420)
    assert oracle.mountain_car_step(MountainCarEnv(-0.5, 0.0), 3)[0] == -1  # Discrete(3)


def test_mountain_car_trajectory(golden, oracle):
    tr = golden("mountain_car")["trajectories"][0]
    e = MountainCarEnv(*tr["start"])
    t, total = 0, 0.0
    while True:
        _, r = oracle.mountain_car_step(e, 2 if e.velocity >= 0 else 0)
        total += r.reward
        t += 1
        if r.done:
            break
    assert t == tr["steps"] == 124 and total == -124.0
    assert [r.obs[0], r.obs[1]] == tr["final"]


def test_pendulum_spec_vectors(golden, oracle):
    """Spec-derived (Gym Pendulum-v1), not reference data: parity unpinned."""
    pd = golden("pendulum")
    for case in pd["single_steps"] + pd["single_steps_f32_inputs"]:
        e = PendulumEnv(*case["state"])
        _, r = oracle.pendulum_step(e, case["action"])
        assert [e.theta, e.theta_dot] == pytest.approx(case["next"], rel=1e-15, abs=1e-15)
        assert list(r.obs)[:3] == pytest.approx(case["obs"], rel=1e-14, abs=1e-15)
        assert r.reward == pytest.approx(case["reward"], rel=1e-14)
        assert r.done == 0
    tr = pd["trajectory"]
    e = PendulumEnv(*tr["start"])
    ret = 0.0
    for t in range(tr["steps"]):
        _, r = oracle.pendulum_step(e, 2.0 if (t // 10) % 2 == 0 else -2.0)
        ret += r.reward
    assert [e.theta, e.theta_dot] == pytest.approx(tr["final"], rel=1e-10)
    assert ret == pytest.approx(tr["total_reward"], rel=1e-12)


def test_philox_known_answers(golden, oracle, twin):
    ph = golden("philox")
    for k in ph["random123_kat"] + ph["model"]:
        assert oracle.philox(k["ctr"], k["key"]) == k["out"]
        assert twin.philox(k["ctr"], k["key"]) == k["out"]  # the product header's Philox, host build


def test_reset_sampling_vectors(golden, oracle):
    ph = golden("philox")["resets"]
    for kind, name in ((0, "cartpole"), (1, "mountain_car"), (2, "pendulum")):
        for case in ph[name]:
            st = oracle.reset_batch(kind, 1, case["gid"], case["seed"], case["tick"])
            assert list(st[:, 0]) == case["state"], (name, case)
    # order x, x_dot, theta, theta_dot, each on [-0.05, 0.05) (cartpole.rs:317-324, 353-361)
    st = oracle.reset_batch(0, 4096, 0, 0, 0)
    assert st.min() >= -0.05 and st.max() < 0.05
    mc = oracle.reset_batch(1, 4096, 0, 0, 0)
    assert mc[0].min() >= -0.6 and mc[0].max() < -0.4 and np.all(mc[1] == 0.0)  # mountain_car.rs:162-167
    # options override the box (cartpole.rs:352-364)
    st = oracle.reset_batch(0, 1024, 0, 3, 0, bounds=[-1, 0, 0.1, 5, 1, 0.5, 0.2, 6])
    assert st[0].min() >= -1 and st[0].max() < 1 and st[3].min() >= 5 and st[3].max() < 6


def test_baseline_loop_counts(oracle):
    secs, out = oracle.baseline_loop(0, 100000, 475, 0)
    sum_return, sum_length, n_episodes, n_steps = out
    assert n_steps == 100000 and sum_length == 100000 and sum_return == sum_length and n_episodes > 1000 and secs > 0
    secs, out = oracle.baseline_loop(1, 50000, 200, 0)
    assert out[0] == -out[1] == -50000.0 and out[2] >= 250
