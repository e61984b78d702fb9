"""gymrs_rollout (the fused multi-step kernel) against the CPU f32 twin driven step by step
through fill_actions + step, and against the per-step GPU kernel: bit-exact state, per-step outputs of the
last step, statistics and tick."""
import numpy as np
import pytest
import torch

from oracle.bindings import TwinEngine

pytestmark = pytest.mark.gpu

A, S, T = 1, 2, 4


def assert_same(eng, tw, kind, flags):
    assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32))
    assert np.array_equal(eng.get_obs().view(np.uint32), tw.get_obs().view(np.uint32))
    gr, gd, gt = eng.get_step_result()
    tr, td, tt = tw.get_result()
    assert np.array_equal(gr.view(np.uint32), tr.view(np.uint32))
    assert np.array_equal(gd, td)
    if flags & T:
        assert np.array_equal(gt, tt)
    gs, ts = eng.stats(), tw.stats()
    assert np.array_equal(gs[1:], ts[1:])
    if kind == 2:
        assert gs[0] == pytest.approx(ts[0], rel=1e-5)  # f32 partial sums are grouped per wave on the GPU
    else:
        assert gs[0] == ts[0]


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("flags", [A | S, A | S | T, A, 0, T, A | T])
@pytest.mark.parametrize("n,vec", [(5000, 4), (777, 8)])
def test_rollout_equals_fill_actions_plus_step(gymrs, twin, kind, flags, n, vec):
    p = gymrs.engine.default_params(kind)
    p.max_episode_steps = 17
    eng = gymrs.BatchedEngine(kind, n, global_env_offset=12345, flags=flags, params=p, lanes_per_thread=vec)
    tw = TwinEngine(twin, kind, n, p, flags=flags, gid0=12345)
    eng.reset(seed=3)
    tw.reset(3)
    t = 0
    for steps, t0 in ((1, 0), (7, 1), (40, 8), (3, 1001)):  # unaligned starts, chunk remainders, two time limits
        eng.rollout(steps, action_seed=6, action_t0=t0)
        for k in range(steps):
            tw.step(tw.fill_actions(6, t0 + k))
        t += steps
        assert_same(eng, tw, kind, flags)
        assert eng.tick()[0] == t + 1
    eng.close()


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_rollout_interleaves_with_per_step_kernel(gymrs, twin, kind):
    """rollout -> step -> rollout on the same engine: both kernels keep the shared bookkeeping (ep_start,
    statistics slots, Pendulum's uniform episode clock) in the same state."""
    n, flags = 6001, A | S | T
    p = gymrs.engine.default_params(kind)
    p.max_episode_steps = 11
    eng = gymrs.BatchedEngine(kind, n, flags=flags, params=p)
    tw = TwinEngine(twin, kind, n, p, flags=flags)
    eng.reset(seed=8)
    tw.reset(8)
    buf = torch.empty(n, dtype=torch.float32 if kind == 2 else torch.uint8, device="cuda:0")
    t = 0
    for phase in range(3):
        eng.rollout(9, action_seed=2, action_t0=t)
        for k in range(9):
            tw.step(tw.fill_actions(2, t + k))
        t += 9
        for k in range(5):
            eng.fill_actions(buf.data_ptr(), seed=2, t=t + k)
            eng.step(buf.data_ptr())
            tw.step(tw.fill_actions(2, t + k))
        t += 5
        assert_same(eng, tw, kind, flags)
    eng.close()


def test_rollout_matches_per_step_gpu_at_full_size(gymrs):
    """BASELINE configs[1] size: 2^20 CartPole lanes, 64 steps, fused vs per-step kernel on the GPU."""
    n, steps = 1 << 20, 64
    flags = A | S
    a = gymrs.BatchedEngine(0, n, flags=flags)
    b = gymrs.BatchedEngine(0, n, flags=flags)
    a.reset(seed=0)
    b.reset(seed=0)
    a.rollout(steps, action_seed=1, action_t0=0)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    for k in range(steps):
        b.fill_actions(buf.data_ptr(), seed=1, t=k)
        b.step(buf.data_ptr())
    assert np.array_equal(a.get_state().view(np.uint32), b.get_state().view(np.uint32))
    assert np.array_equal(a.stats(), b.stats())
    sa = a.stats()
    assert sa[3] == n * steps and 15 < sa[1] / sa[2] < 30  # random-policy CartPole episodes last ~22 steps
    a.close()
    b.close()


def test_rollout_zero_steps_is_a_no_op(gymrs):
    with gymrs.BatchedEngine(0, 100, flags=A | S) as eng:
        eng.reset(seed=1)
        before = eng.get_state()
        eng.rollout(0, action_seed=1)
        assert np.array_equal(before, eng.get_state()) and eng.tick()[0] == 1


def test_pendulum_returns_survive_a_change_of_launch_shape(gymrs, twin):
    """Pendulum's open-episode reward sums live in per-wavefront slots; changing the lanes per work-item
    (4 -> 16 -> rollout at 4 -> 8) in the middle of episodes must not lose any of them."""
    n, flags = 20_011, A | S | T
    p = gymrs.engine.default_params(2)
    p.max_episode_steps = 30
    eng = gymrs.BatchedEngine(2, n, flags=flags, params=p)
    tw = TwinEngine(twin, 2, n, p, flags=flags)
    eng.reset(seed=4)
    tw.reset(4)
    buf = torch.empty(n, dtype=torch.float32, device="cuda:0")
    t = 0

    def eager(k):
        nonlocal t
        for _ in range(k):
            eng.fill_actions(buf.data_ptr(), seed=9, t=t)
            eng.step(buf.data_ptr())
            tw.step(tw.fill_actions(9, t))
            t += 1

    def fused(k):
        nonlocal t
        eng.rollout(k, action_seed=9, action_t0=t)
        for _ in range(k):
            tw.step(tw.fill_actions(9, t))
            t += 1

    eager(7)
    eng.set_tuning(8)
    eager(11)
    eng.set_tuning(4)
    fused(9)          # the launch shape changes in mid-episode: the open sums are gathered first
    eng.set_tuning(8)
    eager(13)         # crosses the 30-step limit: every lane's episode closes here
    fused(25)
    eager(10)         # and a second time
    gs, ts = eng.stats(), tw.stats()
    assert gs[2] == ts[2] == 2 * n and gs[1] == ts[1] == 60 * n
    assert gs[0] == pytest.approx(ts[0], rel=1e-6)
    assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32))
    eng.close()


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("flags", [A | S, A | S | T, T])
def test_rollout_record_keeps_what_per_step_stepping_shows(gymrs, kind, flags):
    """gymrs_rollout_record: row k of the trajectory = what the per-step API shows after step k (observation incl.
    fresh states of re-armed lanes, action, reward, done, truncated), and the engines end up identical."""
    n, steps, t0 = 5001, 37, 5
    stride = 5008
    p = gymrs.engine.default_params(kind)
    p.max_episode_steps = 9
    rec = gymrs.BatchedEngine(kind, n, flags=flags, params=p, global_env_offset=64)
    ref = gymrs.BatchedEngine(kind, n, flags=flags, params=p, global_env_offset=64)
    rec.reset(seed=12)
    ref.reset(seed=12)
    obs_dim = rec.obs_dim
    act_dtype = torch.float32 if kind == 2 else torch.uint8
    dev = "cuda:0"
    obs = torch.full((steps, obs_dim, stride), float("nan"), dtype=torch.float32, device=dev)
    act = torch.zeros((steps, stride), dtype=act_dtype, device=dev)
    rew = torch.full((steps, stride), float("nan"), dtype=torch.float32, device=dev)
    done = torch.full((steps, stride), 9, dtype=torch.uint8, device=dev)
    trunc = torch.full((steps, stride), 9, dtype=torch.uint8, device=dev)
    # torch filled these on ITS stream; the engine writes them on its own non-blocking stream: without this wait a fill can land after the rollout's first rows
    # (found by round 5's suite soak, profiles/r05_suite_soak.log: 2 runs in 40 saw the fill value in the trailing lanes of row 0 / 1)
    torch.cuda.synchronize()
    rec.rollout_record(steps, 3, t0, obs=obs.data_ptr(), actions=act.data_ptr(), reward=rew.data_ptr(), done=done.data_ptr(),
                       truncated=trunc.data_ptr(), lane_stride=stride)
    rec.sync()
    obs_h, act_h, rew_h, done_h, trunc_h = (x.cpu().numpy() for x in (obs, act, rew, done, trunc))
    buf = torch.empty(n, dtype=act_dtype, device=dev)
    for k in range(steps):
        ref.fill_actions(buf.data_ptr(), seed=3, t=t0 + k)
        ref.step(buf.data_ptr())
        ref.sync()
        assert np.array_equal(act_h[k, :n], buf.cpu().numpy()), k
        assert np.array_equal(obs_h[k, :, :n].view(np.uint32), ref.get_obs().view(np.uint32)), k
        r, d, tr = ref.get_step_result()
        assert np.array_equal(rew_h[k, :n].view(np.uint32), r.view(np.uint32)), k
        assert np.array_equal(done_h[k, :n], d), k
        if flags & T:
            assert np.array_equal(trunc_h[k, :n], tr), k
    assert np.isnan(obs_h[:, :, n:]).all() and (done_h[:, n:] == 9).all()  # the padding of a row is never written
    assert np.array_equal(rec.get_state().view(np.uint32), ref.get_state().view(np.uint32))
    assert np.array_equal(rec.stats(), ref.stats()) and rec.tick() == ref.tick()
    rec.close()
    ref.close()


def test_rollout_record_rejects_bad_buffers(gymrs):
    with gymrs.BatchedEngine(0, 100, flags=A) as eng:
        eng.reset(seed=1)
        good = torch.zeros(4 * 112 * 4, dtype=torch.float32, device="cuda:0").data_ptr()
        with pytest.raises(gymrs.GymrsError):
            eng.rollout_record(1, 1, 0, obs=good, actions=good, reward=good, done=good, lane_stride=96)   # < n
        with pytest.raises(gymrs.GymrsError):
            eng.rollout_record(1, 1, 0, obs=good, actions=good, reward=good, done=good, lane_stride=104)  # not a multiple of 16
        with pytest.raises(gymrs.GymrsError):
            eng.rollout_record(1, 1, 0, obs=good + 4, actions=good, reward=good, done=good, lane_stride=112)  # misaligned
        with pytest.raises(gymrs.GymrsError):
            eng.rollout_record(1, 1, 0, obs=good, actions=0, reward=good, done=good, lane_stride=112)  # missing buffer
