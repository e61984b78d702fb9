"""GYMRS_TIME_LIMIT elision (docs/history/DESIGN_rounds_1-4.md §3.1 item 6b): with all three flags a CartPole launch runs WITHOUT the time limit
-- the reset-logged headline kernel -- whenever the host can prove that no lane reaches the limit
in that step (every open episode started at or after `start_bound`, refreshed asynchronously from the age of the oldest
episode).  Nothing observable may change: states, rewards, done / truncated flags and statistics stay bit-identical to the
CPU f32 twin, which checks the limit on every lane in every step."""
import json

import numpy as np
import pytest
import torch

from oracle.bindings import TwinEngine

pytestmark = pytest.mark.gpu
ALL = 1 | 2 | 4


class Pair:
    def __init__(self, gymrs, twin, kind, n, limit=None, gid0=0, seed=5):
        self.kind, self.n = kind, n
        p = gymrs.engine.default_params(kind)
        if limit is not None:
            p.max_episode_steps = limit
        self.p = p
        self.eng = gymrs.BatchedEngine(kind, n, flags=ALL, params=p, global_env_offset=gid0)
        self.tw = TwinEngine(twin, kind, n, p, flags=ALL, gid0=gid0)
        self.eng.reset(seed=seed)
        self.tw.reset(seed)
        self.buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
        self.t = 0

    def step(self, k=1, check_flags=False):
        for _ in range(k):
            self.eng.fill_actions(self.buf.data_ptr(), seed=3, t=self.t)
            self.eng.step(self.buf.data_ptr())
            self.tw.step(self.tw.fill_actions(3, self.t))
            self.t += 1
            if check_flags:
                self.check_flags()

    def check_flags(self):
        r, d, tr = self.eng.get_step_result()
        wr, wd, wtr = self.tw.get_result()
        assert np.array_equal(d, wd) and np.array_equal(tr, wtr) and np.array_equal(r, wr), f"step {self.t}"

    def check(self, what=""):
        assert np.array_equal(self.eng.get_state().view(np.uint32), self.tw.get_state().view(np.uint32)), what
        assert np.array_equal(self.eng.stats(), self.tw.stats()), (what, self.eng.stats(), self.tw.stats())
        self.check_flags()

    def elided(self):
        return json.loads(self.eng.env_json(0))["gymrs"]["time_limit_elided_launches"]


def test_cartpole_v1_limit_never_reached_runs_the_headline_kernel(gymrs, twin):
    """Random policy, limit 500: episodes last ~22 steps, the oldest a few hundred at most.  Nearly every launch runs
    without the limit; the bound is refreshed before it expires."""
    p = Pair(gymrs, twin, 0, 6000, gid0=123)
    for block in range(14):
        p.step(50, check_flags=block in (0, 9, 10))
        p.check(f"after {p.t} steps")
    assert p.elided() >= p.t - 20, (p.elided(), p.t)


@pytest.mark.parametrize("limit", [1, 2, 9, 23, 64])
def test_cartpole_short_limits_switch_back_and_forth(gymrs, twin, limit):
    """Limits that random-policy episodes do reach: launches with and without the limit alternate, refreshes with
    back-off in between; flags compared after every single step."""
    p = Pair(gymrs, twin, 0, 3000, limit=limit, seed=limit)
    p.step(260, check_flags=True)
    p.check()
    if limit >= 9:
        assert 0 < p.elided() < p.t


def test_mountain_car_v0_truncates_everybody_at_200(gymrs, twin):
    """Random policy never reaches the flag: every lane is truncated at step 200, 400, ... (all re-armed at once).  The
    elision is CartPole's (for MountainCar it measured no gain); this pins the plain time-limit path next to it."""
    p = Pair(gymrs, twin, 1, 4000)
    p.step(190)
    p.check()
    p.step(25, check_flags=True)  # across the first mass truncation
    _, _, tr = p.eng.get_step_result()
    p.step(175)
    p.step(30, check_flags=True)  # and the second
    p.check()
    s = p.eng.stats()
    assert s[2] >= 2 * 4000 * 0.99
    assert "time_limit_elided_launches" not in json.loads(p.eng.env_json(0))["gymrs"]


def test_limit_changed_in_mid_run(gymrs, twin):
    """set_params lowers max_episode_steps below the age of open episodes: the very next launch must truncate them."""
    p = Pair(gymrs, twin, 0, 4000)
    p.step(60)
    q = type(p.p).from_buffer_copy(p.p)
    q.max_episode_steps = 12
    p.eng.set_params(q)
    p.tw.set_params(q)
    p.step(40, check_flags=True)
    q.max_episode_steps = 500
    p.eng.set_params(q)
    p.tw.set_params(q)
    p.step(80, check_flags=True)
    p.check()


def test_clone_snapshot_reset_rollout_and_graphs(gymrs, twin):
    p = Pair(gymrs, twin, 0, 5000, limit=40, gid0=9)
    p.step(13)  # launches without the limit, reset-log rows pending
    c = p.eng.clone()
    blob = p.eng.snapshot()
    p.step(70, check_flags=True)
    for t in range(p.t - 70, p.t):
        c.fill_actions(p.buf.data_ptr(), seed=3, t=t)
        c.step(p.buf.data_ptr())
    assert np.array_equal(c.get_state().view(np.uint32), p.tw.get_state().view(np.uint32))
    assert np.array_equal(c.stats(), p.tw.stats())
    c.close()
    # restore into an engine in another phase: nothing is known about the loaded clocks until a refresh has looked
    r = gymrs.BatchedEngine(0, 5000, flags=ALL, params=p.p, global_env_offset=500)
    r.reset(seed=1)
    r.restore(blob)
    for t in range(p.t - 70, p.t):
        r.fill_actions(p.buf.data_ptr(), seed=3, t=t)
        r.step(p.buf.data_ptr())
    assert np.array_equal(r.get_state().view(np.uint32), p.tw.get_state().view(np.uint32))
    assert np.array_equal(r.stats(), p.tw.stats())
    _, _, tr = r.get_step_result()
    assert np.array_equal(tr, p.tw.get_result()[2])
    r.close()
    # fused rollout in between (it writes the flags itself), then steps again
    p.eng.rollout(25, action_seed=3, action_t0=p.t)
    for _ in range(25):
        p.tw.step(p.tw.fill_actions(3, p.t))
        p.t += 1
    p.step(30, check_flags=True)
    p.check("after the rollout")
    # graph replays keep the limit in every captured launch; eager steps after them elide again
    nbuf = 8
    bufs = torch.empty((nbuf, p.n), dtype=torch.uint8, device="cuda:0")
    acts = []
    for b in range(nbuf):
        p.eng.fill_actions(bufs[b].data_ptr(), seed=11, t=b)
        acts.append(p.tw.fill_actions(11, b))
    p.eng.step_many(bufs.data_ptr(), p.n, nbuf, 70, use_graph=True)
    for t in range(70):
        p.tw.step(acts[t % nbuf])
    p.t += 70
    p.check("after graph replays")
    p.step(50, check_flags=True)
    # reset in the middle of everything
    p.eng.reset(seed=77)
    p.tw.reset(77)
    p.t = 0
    p.step(90, check_flags=True)
    p.check("after the reset")


def test_full_size_long_run_statistics(gymrs):
    """2^20 lanes, limit 500, 1200 steps: nobody is truncated under a random policy; the statistics equal those of an
    engine without the limit flag (same seed, same actions), and the limit was checked in next to no launch."""
    n, steps, nbuf = 1 << 20, 1200, 8
    a = gymrs.BatchedEngine(0, n, flags=ALL)
    b = gymrs.BatchedEngine(0, n, flags=1 | 2)
    bufs = torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
    for e in (a, b):
        e.reset(seed=4)
    for k in range(nbuf):
        a.fill_actions(bufs[k].data_ptr(), seed=2, t=k)
    a.step_many(bufs.data_ptr(), n, nbuf, steps)
    a.sync()  # (b's chain would otherwise compete with a's launches and its bound refreshes: this test counts elided launches)
    b.step_many(bufs.data_ptr(), n, nbuf, steps)
    assert np.array_equal(a.stats(), b.stats())
    assert np.array_equal(a.get_state().view(np.uint32), b.get_state().view(np.uint32))
    _, _, tr = a.get_step_result()
    assert not tr.any()
    # (a launch checks the limit itself when the refresh it would need has not arrived after a bounded wait -- inside a chain the
    # host runs far ahead of the device, so each approach to the bound costs a few such launches: a few per cent of a run)
    x = json.loads(a.env_json(0))["gymrs"]
    assert x["time_limit_elided_launches"] >= 0.9 * steps, x
    a.close()
    b.close()


def test_elision_on_a_callers_stream_and_on_one_lane(gymrs, twin):
    """The refresh kernels follow the engine onto an external stream (gymrs_set_stream); a one-lane engine (arrays in
    mapped host memory) elides and truncates like any other."""
    import ctypes as C

    p = Pair(gymrs, twin, 0, 3000, limit=30, seed=2)
    s = torch.cuda.Stream()
    lib = gymrs.load_library()
    assert lib.gymrs_set_stream(p.eng._h, C.c_void_p(s.cuda_stream)) == 0
    with torch.cuda.stream(s):
        p.step(150, check_flags=True)
    p.check("on the caller's stream")
    assert 0 < p.elided() < p.t
    one = Pair(gymrs, twin, 0, 1, limit=12, seed=4)
    one.step(120, check_flags=True)
    one.check("one lane")
