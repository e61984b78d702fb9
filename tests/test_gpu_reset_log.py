"""The reset log (docs/history/DESIGN_rounds_1-4.md §3.1 item 6): CartPole engines with GYMRS_TRACK_STATS and without GYMRS_TIME_LIMIT keep their
episode bookkeeping as done-mask rows in an 8-row ring that every 8th launch folds inside the kernel, and that a
stand-alone kernel folds on demand.  Whatever the host does between two steps -- reading statistics at any phase of the
ring, changing the launch shape, replaying captured graphs, running the fused rollout, cloning, snapshotting,
resetting -- statistics, ep_start-derived lengths and state must stay bit-identical to the CPU f32 twin, which knows
nothing about rings."""
import numpy as np
import pytest
import torch

from oracle.bindings import TwinEngine

pytestmark = pytest.mark.gpu
A, S = 1, 2


class Pair:
    """A GPU engine and its twin, stepped in lockstep with the shared random-policy action stream."""

    def __init__(self, gymrs, twin, n, gid0=0, vec=None, seed=5):
        self.n = n
        self.eng = gymrs.BatchedEngine(gymrs.CARTPOLE, n, flags=A | S, global_env_offset=gid0, lanes_per_thread=vec)
        self.tw = TwinEngine(twin, 0, n, gymrs.engine.default_params(0), flags=A | S, gid0=gid0)
        self.eng.reset(seed=seed)
        self.tw.reset(seed)
        self.buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
        self.t = 0

    def step(self, k=1):
        for _ in range(k):
            self.eng.fill_actions(self.buf.data_ptr(), seed=3, t=self.t)
            self.eng.step(self.buf.data_ptr())
            self.tw.step(self.tw.fill_actions(3, self.t))
            self.t += 1

    def rollout(self, k):
        self.eng.rollout(k, action_seed=3, action_t0=self.t)
        for _ in range(k):
            self.tw.step(self.tw.fill_actions(3, self.t))
            self.t += 1

    def check(self, what=""):
        assert np.array_equal(self.eng.stats(), self.tw.stats()), (what, self.eng.stats(), self.tw.stats())
        assert np.array_equal(self.eng.get_state().view(np.uint32), self.tw.get_state().view(np.uint32)), what


def test_statistics_read_at_every_phase_of_the_ring(gymrs, twin):
    """stats() after 1, 2, ... 40 steps: 0 .. 7 rows pending, in-kernel folds in between, on-demand folds every time."""
    p = Pair(gymrs, twin, 6000, gid0=1000)
    for k in range(1, 41):
        p.step()
        p.check(f"after {k} steps")
    # and without any read in between: 3 in-kernel folds, 5 rows pending at the read
    q = Pair(gymrs, twin, 6000, gid0=1000)
    q.step(29)
    q.check("29 steps, one read")


@pytest.mark.parametrize("first", [1, 3, 7, 8, 9])
def test_launch_shape_change_with_rows_pending(gymrs, twin, first):
    """Rows are laid out per wavefront of ONE launch shape: changing lanes per work-item folds the pending rows first."""
    p = Pair(gymrs, twin, 9001)
    p.step(first)
    p.eng.set_tuning(8)
    p.step(13)
    p.eng.set_tuning(4)
    p.step(6)
    p.check()
    p.eng.set_tuning(8)
    p.step(8)  # exactly one ring period in the other shape
    p.check()


@pytest.mark.parametrize("nbuf,steps", [(4, 100), (5, 93), (8, 64), (3, 150), (32, 70)])
def test_graph_replay_on_a_reset_logged_engine(gymrs, twin, nbuf, steps):
    """A captured graph holds a whole number of ring periods and carries its folding launches at fixed positions; replays,
    the eager remainder and eager steps around them leave the same bits as eager stepping."""
    n = 5000
    eng = gymrs.BatchedEngine(gymrs.CARTPOLE, n, flags=A | S)
    tw = TwinEngine(twin, 0, n, gymrs.engine.default_params(0), flags=A | S)
    eng.reset(seed=21)
    tw.reset(21)
    bufs = torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
    acts = [tw.fill_actions(8, b) for b in range(nbuf)]
    for b in range(nbuf):
        eng.fill_actions(bufs[b].data_ptr(), seed=8, t=b)
    eng.step_many(bufs.data_ptr(), n, nbuf, 3)  # 3 eager steps first: the replay must start on a folded ring
    for t in range(3):
        tw.step(acts[t % nbuf])
    for call in range(2):
        eng.step_many(bufs.data_ptr(), n, nbuf, steps, use_graph=True)
        for t in range(steps):
            tw.step(acts[t % nbuf])
        assert np.array_equal(eng.stats(), tw.stats()), call
        assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32)), call
    eng.step_many(bufs.data_ptr(), n, nbuf, 5)
    for t in range(5):
        tw.step(acts[t % nbuf])
    assert np.array_equal(eng.stats(), tw.stats())
    eng.close()


def test_rollout_clone_snapshot_and_reset_with_rows_pending(gymrs, twin):
    p = Pair(gymrs, twin, 7000, gid0=12)
    p.step(5)
    p.rollout(11)  # the rollout kernel carries ep_start itself: the 5 pending rows are folded first
    p.step(3)
    p.check("steps, rollout, steps")
    # clone with rows pending: the copy continues like the original
    p.step(4)
    c = p.eng.clone()
    blob = p.eng.snapshot()
    p.step(10)
    for t in range(p.t - 10, p.t):
        c.fill_actions(p.buf.data_ptr(), seed=3, t=t)
        c.step(p.buf.data_ptr())
    assert np.array_equal(c.stats(), p.tw.stats())
    assert np.array_equal(c.get_state().view(np.uint32), p.tw.get_state().view(np.uint32))
    c.close()
    # snapshot taken with rows pending, restored into an engine that has rows pending itself
    r = gymrs.BatchedEngine(gymrs.CARTPOLE, 7000, flags=A | S, global_env_offset=999)
    r.reset(seed=77)
    for t in range(6):
        r.fill_actions(p.buf.data_ptr(), seed=9, t=t)
        r.step(p.buf.data_ptr())
    r.restore(blob)
    for t in range(p.t - 10, p.t):
        r.fill_actions(p.buf.data_ptr(), seed=3, t=t)
        r.step(p.buf.data_ptr())
    assert np.array_equal(r.stats(), p.tw.stats())
    r.close()
    # reset with rows pending discards them
    p.step(3)
    p.eng.reset(seed=8)
    p.tw.reset(8)
    p.t = 0
    p.step(20)
    p.check("after a reset in mid-ring")
    # stats_clear in mid-ring
    p.step(3)
    p.eng.stats_clear()
    p.tw.stats_clear()
    p.step(12)
    p.check("after stats_clear in mid-ring")


def test_full_size_statistics_are_consistent(gymrs):
    """2^20 lanes, 200 steps (25 in-kernel folds): every finished episode is counted once and the lengths tile the steps."""
    n, steps, nbuf = 1 << 20, 200, 8
    eng = gymrs.BatchedEngine(gymrs.CARTPOLE, n, flags=A | S)
    eng.reset(seed=0)
    bufs = torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
    for b in range(nbuf):
        eng.fill_actions(bufs[b].data_ptr(), seed=1, t=b)
    for t in range(steps):
        eng.step(bufs[t % nbuf].data_ptr())
        if t % 37 == 0 or t > steps - 10:  # reads at irregular phases of the ring
            s = eng.stats()
            assert s[3] == n * (t + 1) and s[0] == s[1] and s[1] <= s[3]
    s = eng.stats()
    # a second engine stepping the same actions through ONE graph-replayed call gives the same statistics
    eng2 = gymrs.BatchedEngine(gymrs.CARTPOLE, n, flags=A | S)
    eng2.reset(seed=0)
    eng2.step_many(bufs.data_ptr(), n, nbuf, steps, use_graph=True)
    s2 = eng2.stats()
    assert np.array_equal(s, s2)
    # open episodes: total steps - finished lengths = sum of the ages of the open episodes, each at most the longest
    # episode a random policy produces here (well under 200)
    open_steps = s[3] - s[1]
    assert 0 < open_steps <= n * 200 and s[2] > n * 4
    assert np.array_equal(eng.get_state().view(np.uint32), eng2.get_state().view(np.uint32))
    eng.close()
    eng2.close()
