#!/usr/bin/env python
"""Generates the golden fixtures in this directory.

The reference (MathisWellmann/gym-rs) is a Rust crate; this image has no Rust toolchain, so it cannot
be run here, and it holds NO golden vectors or tests for step()/reset().  These fixtures
are therefore produced by an INDEPENDENT literal evaluation of the reference's formulas in Python
(`math.sin/cos` = the same glibc libm that Rust's f64 methods call), written from the source text
  /root/reference/src/envs/classical_control/cartpole.rs:398-483
  /root/reference/src/envs/classical_control/mountain_car.rs:398-435
  /root/reference/src/utils/custom/util_fns.rs:2-10
and cross-checked against SURVEY.md Appendix C.  They pin the C oracle (oracle/gymrs_oracle.c), which
is a second, separate restatement; the two must agree to the last bit (or 1 ulp where libm `pow`
is involved).  Philox known answers are the Random123 KATs plus outputs of the
10-line Python model below.  Pendulum vectors are spec-derived (Gym Pendulum-v1): not reference data.

Run:  python tests/golden/make_golden.py     (writes *.json next to this file; deterministic)
"""
from __future__ import annotations

import json
import math
import random
from pathlib import Path

HERE = Path(__file__).resolve().parent

# ------------------------------------------------------------------------------------------
# CartPole (cartpole.rs:94-103 constants; :146-152 helpers; :408-464 step)
CP = dict(gravity=9.8, masscart=1.0, masspole=0.1, length=0.5, force_mag=10.0, tau=0.02,
          theta_threshold_radians=12. * 2. * math.pi / 360., x_threshold=2.4)


def cartpole_step(state, action, beyond, integrator="euler", p=CP):
    assert action < 2  # Discrete(2).contains
    x, x_dot, theta, theta_dot = state
    total_mass = p["masspole"] + p["masscart"]
    polemass_length = p["masspole"] + p["length"]  # (sic) the reference adds
    force = p["force_mag"] if action == 1 else -p["force_mag"]
    costheta = math.cos(theta)
    sintheta = math.sin(theta)
    temp = (force + polemass_length * (theta_dot * theta_dot) * sintheta) / total_mass
    thetaacc = (p["gravity"] * sintheta - costheta * temp) / (
        p["length"] * (4.0 / 3.0 - p["masspole"] * (costheta * costheta) / total_mass))
    xacc = temp - polemass_length * thetaacc * costheta / total_mass
    if integrator == "euler":
        x = x + p["tau"] * x_dot
        x_dot = x_dot + p["tau"] * xacc
        theta = theta + p["tau"] * theta_dot
        theta_dot = theta_dot + p["tau"] * thetaacc
    else:
        x_dot = x_dot + p["tau"] * xacc
        x = x + p["tau"] * x_dot
        theta_dot = theta_dot + p["tau"] * thetaacc
        theta = theta + p["tau"] * theta_dot
    done = (x < -p["x_threshold"] or x > p["x_threshold"] or theta < -p["theta_threshold_radians"]
            or theta > p["theta_threshold_radians"])
    if not done:
        reward = 1.0
    elif not beyond:
        beyond = True
        reward = 1.0
    else:
        reward = 0.0
    return (x, x_dot, theta, theta_dot), reward, done, beyond


# ------------------------------------------------------------------------------------------
# MountainCar (mountain_car.rs:344-351 constants; :408-423 step; util_fns.rs:2-10 clip)
MC = dict(min_position=-1.2, max_position=0.6, max_speed=0.07, goal_position=0.5, goal_velocity=0.0,
          force=0.001, gravity=0.0025)


def clip(value, left, right):
    if left <= value and value <= right:
        return value
    elif value > right:
        return right
    else:
        return left


def mountain_car_step(state, action, p=MC):
    assert action < 3
    position, velocity = state
    velocity += (float(action) - 1.0) * p["force"] + math.cos(3.0 * position) * (-p["gravity"])
    velocity = clip(velocity, -p["max_speed"], p["max_speed"])
    position += velocity
    position = clip(position, p["min_position"], p["max_position"])
    if position == p["min_position"] and velocity < 0.0:
        velocity = 0.0
    done = position >= p["goal_position"] and velocity >= p["goal_velocity"]
    return (position, velocity), -1.0, done


# ------------------------------------------------------------------------------------------
# Pendulum — Gym Pendulum-v1 (spec-derived; NOT in the reference)
PD = dict(max_speed=8.0, max_torque=2.0, dt=0.05, g=10.0, m=1.0, l=1.0)


def pendulum_step(state, action, p=PD):
    th, thdot = state
    u = clip(action, -p["max_torque"], p["max_torque"])
    y = th + math.pi
    an = (y - (2 * math.pi) * math.floor(y / (2 * math.pi))) - math.pi
    costs = an * an + 0.1 * (thdot * thdot) + 0.001 * (u * u)
    newthdot = thdot + (3.0 * p["g"] / (2.0 * p["l"]) * math.sin(th) + 3.0 / (p["m"] * (p["l"] * p["l"])) * u) * p["dt"]
    newthdot = clip(newthdot, -p["max_speed"], p["max_speed"])
    newth = th + newthdot * p["dt"]
    return (newth, newthdot), (math.cos(newth), math.sin(newth), newthdot), -costs


# ------------------------------------------------------------------------------------------
# Philox4x32-10 (Random123)
def philox4x32_10(ctr, key):
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c = list(ctr)
    k = list(key)
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xffffffff, p1 & 0xffffffff,
             ((p0 >> 32) ^ c[3] ^ k[1]) & 0xffffffff, p0 & 0xffffffff]
        k = [(k[0] + W0) & 0xffffffff, (k[1] + W1) & 0xffffffff]
    return c


def draw4(seed, gid, tick, stream):
    ctr = [gid & 0xffffffff, (gid >> 32) & 0xffffffff, tick & 0xffffffff, ((tick >> 32) & 0xffff) | (stream << 16)]
    return philox4x32_10(ctr, [seed & 0xffffffff, (seed >> 32) & 0xffffffff])


def uniform_between(r, low, high):
    return (r >> 8) * (1.0 / 16777216.0) * (high - low) + low


# ------------------------------------------------------------------------------------------
def main():
    rnd = random.Random(20260927)

    # ---- CartPole ----
    kat_state = (0.01, 0.02, 0.03, 0.04)
    cp = {"constants": CP, "single_steps": [], "trajectories": [], "beyond_terminated": None, "semi_implicit": []}
    for a in (1, 0):
        s, r, d, _ = cartpole_step(kat_state, a, False)
        cp["single_steps"].append({"state": kat_state, "action": a, "next": s, "reward": r, "done": d})
    for _ in range(256):
        st = (rnd.uniform(-2.4, 2.4), rnd.uniform(-3, 3), rnd.uniform(-0.21, 0.21), rnd.uniform(-3, 3))
        a = rnd.randrange(2)
        s, r, d, _ = cartpole_step(st, a, False)
        cp["single_steps"].append({"state": st, "action": a, "next": s, "reward": r, "done": d})
    for _ in range(16):
        st = (rnd.uniform(-0.05, 0.05),) * 1 + (rnd.uniform(-0.05, 0.05), rnd.uniform(-0.05, 0.05), rnd.uniform(-0.05, 0.05))
        a = rnd.randrange(2)
        s, r, d, _ = cartpole_step(st, a, False, integrator="other")
        cp["semi_implicit"].append({"state": st, "action": a, "next": s, "reward": r, "done": d})
    for name, policy in (("always_1", lambda t: 1), ("always_0", lambda t: 0), ("alternate_1_0", lambda t: (t + 1) % 2)):
        s, t, total, beyond = kat_state, 0, 0.0, False
        while True:
            s, r, d, beyond = cartpole_step(s, policy(t), beyond)
            total += r
            t += 1
            if d:
                break
        cp["trajectories"].append({"policy": name, "start": kat_state, "steps": t, "final": s, "total_reward": total})
    # reward 1 -> (terminal) 1 -> 0, 0 after termination (cartpole.rs:455-464)
    s, beyond, rewards, dones = kat_state, False, [], []
    for t in range(14):
        s, r, d, beyond = cartpole_step(s, 1, beyond)
        rewards.append(r)
        dones.append(d)
    cp["beyond_terminated"] = {"start": kat_state, "action": 1, "rewards": rewards, "dones": dones, "final": s}
    # observation space (cartpole.rs:105-113)
    cp["observation_space_high"] = [CP["x_threshold"] * 2.0, "inf", CP["theta_threshold_radians"] * 2.0, "inf"]

    # ---- MountainCar ----
    mc = {"constants": MC, "single_steps": [], "trajectories": []}
    for a in (0, 1, 2):
        s, r, d = mountain_car_step((-0.5, 0.0), a)
        mc["single_steps"].append({"state": (-0.5, 0.0), "action": a, "next": s, "reward": r, "done": d})
    s, r, d = mountain_car_step((-1.19, -0.07), 0)  # wall rule
    mc["single_steps"].append({"state": (-1.19, -0.07), "action": 0, "next": s, "reward": r, "done": d})
    s, r, d = mountain_car_step((0.49, 0.07), 2)  # goal
    mc["single_steps"].append({"state": (0.49, 0.07), "action": 2, "next": s, "reward": r, "done": d})
    for _ in range(256):
        st = (rnd.uniform(-1.2, 0.6), rnd.uniform(-0.07, 0.07))
        a = rnd.randrange(3)
        s, r, d = mountain_car_step(st, a)
        mc["single_steps"].append({"state": st, "action": a, "next": s, "reward": r, "done": d})
    s, t, total = (-0.5, 0.0), 0, 0.0
    while True:
        a = 2 if s[1] >= 0 else 0
        s, r, d = mountain_car_step(s, a)
        total += r
        t += 1
        if d:
            break
    mc["trajectories"].append({"policy": "bang_bang", "start": (-0.5, 0.0), "steps": t, "final": s, "total_reward": total})

    # ---- Pendulum (spec-derived) ----
    pd = {"constants": PD, "note": "spec-derived from Gym Pendulum-v1; NOT reference data", "single_steps": [], "trajectory": None}
    for _ in range(256):
        st = (rnd.uniform(-math.pi, math.pi) + rnd.choice([0, 0, 10 * math.pi, -37.0]), rnd.uniform(-8, 8))
        a = rnd.uniform(-2.5, 2.5)
        ns, obs, rew = pendulum_step(st, a)
        pd["single_steps"].append({"state": st, "action": a, "next": ns, "obs": obs, "reward": rew})
    s, ret = (math.pi - 0.1, 0.5), 0.0
    for t in range(200):
        s, obs, rew = pendulum_step(s, 2.0 if (t // 10) % 2 == 0 else -2.0)
        ret += rew
    pd["trajectory"] = {"start": (math.pi - 0.1, 0.5), "policy": "+2 for 10 steps, -2 for 10 steps", "steps": 200,
                        "final": s, "total_reward": ret}

    # ---- Philox + reset sampling ----
    ph = {
        "random123_kat": [
            {"ctr": [0, 0, 0, 0], "key": [0, 0], "out": [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]},
            {"ctr": [0xffffffff] * 4, "key": [0xffffffff] * 2, "out": [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]},
            {"ctr": [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], "key": [0xa4093822, 0x299f31d0],
             "out": [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]},
        ],
        "model": [],
        "resets": {"cartpole": [], "mountain_car": [], "pendulum": []},
    }
    for k in ph["random123_kat"]:
        assert philox4x32_10(k["ctr"], k["key"]) == k["out"], "python Philox model disagrees with Random123 KAT"
    for _ in range(16):
        ctr = [rnd.getrandbits(32) for _ in range(4)]
        key = [rnd.getrandbits(32) for _ in range(2)]
        ph["model"].append({"ctr": ctr, "key": key, "out": philox4x32_10(ctr, key)})
    for seed, gid, tick in [(0, 0, 0), (0, 1, 0), (0, (1 << 20) - 1, 0), (42, 5, 17), (0xDEADBEEFCAFEF00D, (1 << 33) + 3, (1 << 32) + 9)]:
        r = draw4(seed, gid, tick, 0)
        ph["resets"]["cartpole"].append({"seed": seed, "gid": gid, "tick": tick,
                                         "state": [uniform_between(r[j], -0.05, 0.05) for j in range(4)]})
        ph["resets"]["mountain_car"].append({"seed": seed, "gid": gid, "tick": tick,
                                             "state": [uniform_between(r[0], -0.6, -0.4), 0.0]})
        ph["resets"]["pendulum"].append({"seed": seed, "gid": gid, "tick": tick,
                                         "state": [uniform_between(r[0], -math.pi, math.pi), uniform_between(r[1], -1.0, 1.0)]})

    # ---- single steps from f32-REPRESENTABLE states (round 5): the GPU holds its state in f32, so a fixture whose inputs are exact
    #      in f32 lets the GPU test hold north_star's 1e-6 itself instead of 5e-6 (input rounding amplified by |d next / d state| ~ 20).  A second
    #      generator: the vectors above stay what they were.  Evaluated in f64 like everything else here.
    import struct

    def f32(x):
        return struct.unpack("<f", struct.pack("<f", x))[0]

    r32 = random.Random(20260929)
    cp["single_steps_f32_inputs"] = []
    for _ in range(512):
        st = tuple(f32(v) for v in (r32.uniform(-2.4, 2.4), r32.uniform(-3, 3), r32.uniform(-0.21, 0.21), r32.uniform(-3, 3)))
        a = r32.randrange(2)
        s, r, d, _ = cartpole_step(st, a, False)
        cp["single_steps_f32_inputs"].append({"state": st, "action": a, "next": s, "reward": r, "done": d})
    mc["single_steps_f32_inputs"] = []
    for _ in range(512):
        st = (f32(r32.uniform(-1.2, 0.6)), f32(r32.uniform(-0.07, 0.07)))
        a = r32.randrange(3)
        s, r, d = mountain_car_step(st, a)
        mc["single_steps_f32_inputs"].append({"state": st, "action": a, "next": s, "reward": r, "done": d})
    pd["single_steps_f32_inputs"] = []
    for _ in range(512):
        st = (f32(r32.uniform(-math.pi, math.pi) + r32.choice([0, 0, 10 * math.pi, -37.0])), f32(r32.uniform(-8, 8)))
        a = f32(r32.uniform(-2.5, 2.5))
        ns, obs, rew = pendulum_step(st, a)
        pd["single_steps_f32_inputs"].append({"state": st, "action": a, "next": ns, "obs": obs, "reward": rew})

    # ---- the reference's own unit tests, as data (spaces/discrete.rs:27-41, util_fns.rs:16-32,
    #      seeding.rs:33-39 + doctest seeding.rs:11-20) ----
    pins = {
        "discrete_contains": [{"n": 3, "value": 3, "expect": False}, {"n": 3, "value": 4, "expect": False},
                              {"n": 3, "value": 1, "expect": True}, {"n": 3, "value": 2, "expect": True}],
        "clip": [{"value": 2, "left": 0, "right": 1, "expect": 1}, {"value": -1, "left": 0, "right": 1, "expect": 0},
                 {"value": 1, "left": -1, "right": 2, "expect": 1}],
        "seed_echo": [{"seed": 42, "expect": 42}, {"seed": 64, "expect": 64}],
    }

    for name, obj in (("cartpole", cp), ("mountain_car", mc), ("pendulum", pd), ("philox", ph), ("reference_unit_tests", pins)):
        (HERE / f"{name}.json").write_text(json.dumps(obj, indent=1) + "\n")
        print("wrote", name)


if __name__ == "__main__":
    main()
