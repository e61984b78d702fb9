"""GPU parity tests proper: the gfx950 kernels, called through the C ABI (include/gymrs_amd.h), against
  (1) the f64 C oracle — per single step, |gpu - ref| <= 1e-6 * max(|ref|, 1) (north_star tolerance);
  (2) the CPU f32 twin — BIT-EXACT state words, rewards, done/truncated flags and integer episode /
      step counts over multi-step runs with auto-reset;
  (3) the committed golden vectors (tests/golden);
  (4) size-independent properties at BASELINE.json's full sizes (2^20 / 2^22 lanes)."""
import numpy as np
import pytest
import torch

from oracle.bindings import TwinEngine

pytestmark = pytest.mark.gpu
TOL = 1e-6


def mixed_err(got, ref):
    return np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)


def dev_actions(n, kind):
    return torch.empty(n, dtype=torch.float32 if kind == 2 else torch.uint8, device="cuda:0")


def random_states(kind, n, rng):
    if kind == 0:
        st = np.stack([rng.uniform(-2.4, 2.4, n), rng.uniform(-3, 3, n), rng.uniform(-0.21, 0.21, n), rng.uniform(-3, 3, n)])
        act = rng.integers(0, 2, n).astype(np.uint8)
    elif kind == 1:
        st = np.stack([rng.uniform(-1.2, 0.6, n), rng.uniform(-0.07, 0.07, n)])
        act = rng.integers(0, 3, n).astype(np.uint8)
    else:
        st = np.stack([rng.uniform(-40, 40, n), rng.uniform(-8, 8, n)])
        act = rng.uniform(-2.5, 2.5, n).astype(np.float32)
    return st.astype(np.float32), act


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("n,vec", [(100_003, 4), (1, 4), (1023, 8), (4096, 8), (70_001, 8)])  # ragged tails, one lane, exact tiles
def test_single_step_vs_f64_oracle(kind, n, vec, gymrs, oracle):
    rng = np.random.default_rng(100 + kind)
    st, act = random_states(kind, n, rng)
    with gymrs.BatchedEngine(kind, n, flags=0, lanes_per_thread=vec) as eng:
        eng.reset(seed=1)
        eng.set_state(st)
        eng.step_host(act)
        got = eng.get_state()
        reward, done, trunc = eng.get_step_result()
        obs = eng.get_obs()
    ref = st.astype(np.float64).copy()
    if kind == 0:
        ref_r, ref_d, bad = oracle.cartpole_step_batch(ref, np.zeros(n, np.uint8), act)
        near = (np.abs(np.abs(ref[0]) - 2.4) < 1e-5) | (np.abs(np.abs(ref[2]) - 0.20943951023931953) < 1e-5)
    elif kind == 1:
        ref_r, ref_d, bad = oracle.mountain_car_step_batch(ref, act)
        near = (np.abs(ref[0] - 0.5) < 1e-5) | (np.abs(ref[1]) < 1e-5)
    else:
        ref_r, oc, os_ = oracle.pendulum_step_batch(ref, act.astype(np.float64))
        ref_d, bad, near = np.zeros(n, np.uint8), 0, np.zeros(n, bool)
        scale = np.maximum(np.abs(ref[0]), 1.0)  # cos/sin inherit the f32 representation error of theta
        assert (np.abs(obs[0] - oc) / scale).max() <= TOL and (np.abs(obs[1] - os_) / scale).max() <= TOL
        assert np.array_equal(obs[2], got[1])
    assert bad == 0
    assert mixed_err(got, ref).max() <= TOL
    assert mixed_err(reward, ref_r).max() <= TOL
    assert not trunc.any()  # truncated is hard-coded false in the reference (cartpole.rs:480)
    mism = np.nonzero(done != ref_d)[0]
    assert near[mism].all(), f"{len(mism)} done mismatches outside the 1e-5 threshold band"
    if kind != 2:
        assert np.array_equal(obs, got)  # the observation IS the state (cartpole.rs:476-482)


FLAG_SETS = ["0", "A", "A|S", "T", "A|T", "A|S|T"]


def parse_flags(gymrs, s):
    m = {"0": 0, "A": gymrs.AUTO_RESET, "S": gymrs.TRACK_STATS, "T": gymrs.TIME_LIMIT}
    f = 0
    for part in s.split("|"):
        f |= m[part]
    return f


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("flagset", FLAG_SETS)
def test_multistep_bit_exact_vs_f32_twin(kind, flagset, gymrs, twin):
    """Everything the kernel adds around the physics (vector tails, ballot/LDS compaction, Philox
    counters, statistics partials, time limit) must not change a single bit."""
    flags = parse_flags(gymrs, flagset)
    steps = 60
    params = gymrs.engine.default_params(kind)
    if flags & gymrs.TIME_LIMIT:
        params.max_episode_steps = 17
    for n, vec in ((20_011, 4), (2048, 8), (9_999, 8)):
        eng = gymrs.BatchedEngine(kind, n, flags=flags, params=params, global_env_offset=12345, lanes_per_thread=vec)
        tw = TwinEngine(twin, kind, n, params, flags=flags, gid0=12345)
        eng.reset(seed=2024)
        tw.reset(2024)
        assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32))
        acts = dev_actions(n, kind)
        for t in range(steps):
            eng.fill_actions(acts.data_ptr(), seed=9, t=t)
            eng.step(acts.data_ptr())
            a_host = tw.fill_actions(9, t)
            if t == 0:
                eng.sync()
                assert np.array_equal(acts.cpu().numpy(), a_host)  # the action stream itself
            tw.step(a_host)
            if t % 20 == 19 or t == steps - 1:
                eng.sync()
                assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32)), (n, t)
                gr, gd, gt = eng.get_step_result()
                tr, td, tt = tw.get_result()
                assert np.array_equal(gr.view(np.uint32), tr.view(np.uint32))
                assert np.array_equal(gd, td) and np.array_equal(gt, tt)
                assert np.array_equal(eng.get_obs().view(np.uint32), tw.get_obs().view(np.uint32))
        gs, ts = eng.stats(), tw.stats()
        assert gs[1] == ts[1] and gs[2] == ts[2] and gs[3] == ts[3] == n * steps  # integer counts: exact
        if kind == 2:
            assert gs[0] == pytest.approx(ts[0], rel=1e-5)  # float sum, order differs
        else:
            assert gs[0] == ts[0]
        if flags & gymrs.TRACK_STATS and kind == 0:
            assert gs[2] > 0  # episodes did finish
        eng.close()


def test_step_many_equals_repeated_step(gymrs, twin):
    n, steps, nbuf = 5000, 33, 4
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
    eng = gymrs.BatchedEngine(0, n, flags=flags)
    tw = TwinEngine(twin, 0, n, eng.params, flags=flags)
    eng.reset(seed=5)
    tw.reset(5)
    bufs = torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
    for b in range(nbuf):
        eng.fill_actions(bufs[b].data_ptr(), seed=3, t=b)
    eng.step_many(bufs.data_ptr(), n, nbuf, steps)
    eng.sync()
    for t in range(steps):
        tw.step(tw.fill_actions(3, t % nbuf))
    assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32))
    assert np.array_equal(eng.stats(), tw.stats())
    assert eng.tick()[0] == steps + 1
    eng.close()

@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("hint", [1, 2, 3])
def test_memory_hints_never_change_results(gymrs, twin, kind, hint):
    """gymrs_set_tuning's memory_hint picks another instantiation of the same kernel (1 every access non-temporal, 2 none, 3 only the
    stores nobody reads again -- what the engine uses between 48 MiB and 1 GiB per step): HIP launches (gymrs_step) and chains
    (gymrs_step_many) of each must produce the twin's bits."""
    n, nbuf = 9001, 4
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS | (gymrs.TIME_LIMIT if kind == 2 else 0)
    eng = gymrs.BatchedEngine(kind, n, flags=flags)
    eng.set_tuning(4, hint)
    tw = TwinEngine(twin, kind, n, eng.params, flags=flags)
    eng.reset(seed=9)
    tw.reset(9)
    bufs = torch.empty((nbuf, n), dtype=torch.float32 if kind == 2 else torch.uint8, device="cuda:0")
    for b in range(nbuf):
        eng.fill_actions(bufs[b].data_ptr(), seed=3, t=b)
    for t in range(11):
        eng.step(bufs[t % nbuf].data_ptr())
    eng.step_many(bufs.data_ptr(), bufs.stride(0) * bufs.element_size(), nbuf, 29)
    eng.sync()
    for t in list(range(11)) + list(range(29)):
        tw.step(tw.fill_actions(3, t % nbuf))
    assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32))
    r, d, tr = eng.get_step_result()
    tr_, td, tt = tw.get_result()
    assert np.array_equal(d, td) and np.array_equal(tr, tt) and np.array_equal(r.view(np.uint32), tr_.view(np.uint32))
    eng.close()


@pytest.mark.parametrize("kind,nbuf,steps", [(0, 4, 75), (1, 5, 83), (0, 48, 100)])
def test_step_many_graph_replay_is_bit_identical(gymrs, twin, kind, nbuf, steps):
    """use_graph=1 replays captured graphs of >=32 steps with a device-resident tick; the result (state,
    per-step outputs of the last step, statistics, tick) must equal eager stepping bit for bit, across two
    calls (second call reuses the cached graph) and a remainder."""
    n = 3000
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS | gymrs.TIME_LIMIT
    p = gymrs.engine.default_params(kind)
    p.max_episode_steps = 40
    eng = gymrs.BatchedEngine(kind, n, flags=flags, params=p)
    tw = TwinEngine(twin, kind, n, p, flags=flags)
    eng.reset(seed=21)
    tw.reset(21)
    bufs = torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
    for b in range(nbuf):
        eng.fill_actions(bufs[b].data_ptr(), seed=8, t=b)
    for call in range(2):
        eng.step_many(bufs.data_ptr(), n, nbuf, steps, use_graph=True)
        eng.sync()
        for t in range(steps):
            tw.step(tw.fill_actions(8, t % nbuf))
        assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32)), call
        gr, gd, gt = eng.get_step_result()
        tr, td, tt = tw.get_result()
        assert np.array_equal(gr.view(np.uint32), tr.view(np.uint32)) and np.array_equal(gd, td) and np.array_equal(gt, tt)
        assert np.array_equal(eng.stats(), tw.stats())
        assert eng.tick()[0] == (call + 1) * steps + 1
    # a reset invalidates the cached graph (seed and reset box are baked into it)
    eng.reset(seed=22)
    tw.reset(22)
    eng.step_many(bufs.data_ptr(), n, nbuf, steps, use_graph=True)
    for t in range(steps):
        tw.step(tw.fill_actions(8, t % nbuf))
    assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32))
    assert np.array_equal(eng.stats(), tw.stats())
    eng.close()


def test_step_many_graph_refused_for_pendulum_time_limit(gymrs):
    eng = gymrs.BatchedEngine(gymrs.PENDULUM, 256, flags=gymrs.AUTO_RESET | gymrs.TIME_LIMIT)
    eng.reset(seed=1)
    bufs = torch.zeros((1, 256), dtype=torch.float32, device="cuda:0")
    with pytest.raises(gymrs.GymrsError):
        eng.step_many(bufs.data_ptr(), 1024, 1, 64, use_graph=True)
    eng.close()


def test_golden_vectors_with_f32_inputs_on_gpu_hold_1e_6(gymrs, golden):
    """north_star's bar on committed fixtures: `single_steps_f32_inputs` = 512 single steps per env whose input states are EXACT in
    f32 (tests/golden/make_golden.py, evaluated in f64 from the reference's source text), so nothing is lost handing them to the engine and the result is held
    to 1e-6 relative (to max(|ref|, 1)) -- the f64-input fixtures below need 5e-6 because their inputs are rounded on the way in."""
    for kind, name in ((0, "cartpole"), (1, "mountain_car"), (2, "pendulum")):
        cases = golden(name)["single_steps_f32_inputs"]
        st = np.array([c["state"] for c in cases], np.float64).T
        assert np.array_equal(st.astype(np.float32).astype(np.float64), st)  # exact in f32
        act = np.array([c["action"] for c in cases], np.float32 if kind == 2 else np.uint8)
        with gymrs.BatchedEngine(kind, len(cases)) as eng:
            eng.set_state(st.astype(np.float32))
            eng.step_host(act)
            got = eng.get_state()
            reward, done, _ = eng.get_step_result()
            obs = eng.get_obs()
        want = np.array([c["next"] for c in cases]).T
        assert mixed_err(got, want).max() <= 1e-6, (name, mixed_err(got, want).max())
        if kind == 0:
            # the flag is the reference's compare of the state the engine holds: exact wherever the f64 result is not within f32 rounding of a threshold
            far = np.array([abs(abs(c["next"][0]) - 2.4) > 1e-5 and abs(abs(c["next"][2]) - 0.20943951023931953) > 1e-6 for c in cases])
            assert far.sum() > 450 and np.array_equal(done[far].astype(bool), np.array([c["done"] for c in cases])[far])
            assert np.array_equal(reward, np.ones(len(cases), np.float32))  # cartpole.rs:455-459: 1.0 on the step that terminates too
        elif kind == 1:
            far = np.array([abs(c["next"][0] - 0.5) > 1e-6 for c in cases])
            assert np.array_equal(done[far].astype(bool), np.array([c["done"] for c in cases])[far]) and (reward == -1.0).all()
        else:
            want_obs = np.array([c["obs"] for c in cases]).T
            scale = np.maximum(np.abs(want[0]), 1.0)  # cos / sin inherit the f32 representation error of the new theta (up to ~40 rad here)
            assert (np.abs(obs.astype(np.float64) - want_obs) / scale).max() <= 1e-6
            assert mixed_err(reward, np.array([c["reward"] for c in cases])).max() <= 1e-6


def test_golden_vectors_on_gpu(gymrs, golden):
    cp = golden("cartpole")
    cases = cp["single_steps"]
    st = np.array([c["state"] for c in cases], np.float64).T
    act = np.array([c["action"] for c in cases], np.uint8)
    with gymrs.BatchedEngine(0, len(cases)) as eng:
        eng.set_state(st.astype(np.float32))
        eng.step_host(act)
        got = eng.get_state()
        _, done, _ = eng.get_step_result()
    want = np.array([c["next"] for c in cases]).T
    # inputs differ by f32 rounding (<= 6e-8 relative), amplified by at most ~|d next/d state| ~ 20
    assert mixed_err(got, want).max() <= 5e-6
    far = np.array([abs(abs(c["next"][0]) - 2.4) > 1e-4 and abs(abs(c["next"][2]) - 0.2094395) > 1e-4 for c in cases])
    assert np.array_equal(done[far].astype(bool), np.array([c["done"] for c in cases])[far])
    mc = golden("mountain_car")
    cases = mc["single_steps"]
    st = np.array([c["state"] for c in cases], np.float64).T
    act = np.array([c["action"] for c in cases], np.uint8)
    with gymrs.BatchedEngine(1, len(cases)) as eng:
        eng.set_state(st.astype(np.float32))
        eng.step_host(act)
        got = eng.get_state()
        reward, _, _ = eng.get_step_result()
    assert mixed_err(got, np.array([c["next"] for c in cases]).T).max() <= 2e-6
    assert (reward == -1.0).all()


def test_invalid_action_is_reported_and_lane_untouched(gymrs):
    n = 3000
    with gymrs.BatchedEngine(0, n, flags=gymrs.AUTO_RESET) as eng:
        eng.reset(seed=1)
        before = eng.get_state()
        act = np.zeros(n, np.uint8)
        act[[7, 1500, 2999]] = [2, 255, 9]  # outside Discrete(2)
        with pytest.raises(gymrs.InvalidActionError) as exc:  # the reference panics (cartpole.rs:402-406)
            eng.step_host(act)
        assert "3 invalid" in str(exc.value) and "lane 7" in str(exc.value)
        after = eng.get_state()
        reward, done, _ = eng.get_step_result()
        bad = np.zeros(n, bool)
        bad[[7, 1500, 2999]] = True
        assert np.array_equal(after[:, bad], before[:, bad]) and (reward[bad] == 0).all() and (done[bad] == 0).all()
        assert not np.array_equal(after[:, ~bad], before[:, ~bad])
        eng.step_host(np.ones(n, np.uint8))  # the error flag was cleared; valid steps work again


def test_reset_semantics(gymrs, oracle):
    n = 4096
    with gymrs.BatchedEngine(0, n, global_env_offset=1 << 20) as eng:
        a = eng.get_state()
        assert np.abs(a).max() < 0.05 and a.std() > 0.02  # ::new samples an initial state (cartpole.rs:120)
        used = eng.reset(seed=42)
        assert used == 42  # seed echo (seeding.rs:33-39)
        s1 = eng.get_state()
        eng.step_host(np.ones(n, np.uint8))
        eng.reset(seed=42)
        assert np.array_equal(s1, eng.get_state())  # same seed, same states
        ref = oracle.reset_batch(0, n, 1 << 20, 42, 0)
        assert mixed_err(s1, ref).max() <= TOL
        u1 = eng.reset()
        u2 = eng.reset()
        assert u1 != u2  # OS entropy when seed is None (seeding.rs:22)
        # options override the sampling box (cartpole.rs:352-364)
        eng.reset(seed=1, options=[-1, 0, 0.1, 5, 1, 0.5, 0.2, 6])
        s = eng.get_state()
        assert s[0].min() >= -1 and s[0].max() < 1 and s[2].min() >= np.float32(0.1) and s[3].min() >= 5 and s[3].max() < 6
        with pytest.raises(gymrs.GymrsError):
            eng.reset(seed=1, options=[0, 0, 0, 0, 0, 1, 1, 1])  # low >= high panics in rand
    with gymrs.BatchedEngine(1, n) as eng:
        eng.reset(seed=3)
        s = eng.get_state()
        assert s[0].min() >= np.float32(-0.6) and s[0].max() < np.float32(-0.4) and (s[1] == 0).all()  # mountain_car.rs:162-167


def test_shard_invariance_on_gpu(gymrs):
    """N lanes on one engine == the concatenation of two shards with global offsets."""
    n, steps = 8192, 40
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
    full = gymrs.BatchedEngine(0, n, flags=flags)
    full.reset(seed=77)
    halves = []
    for r in range(2):
        off, cnt = gymrs.shard_range(n, 2, r)
        e = gymrs.BatchedEngine(0, cnt, global_env_offset=off, flags=flags)
        e.reset(seed=77)
        halves.append(e)
    fa = dev_actions(n, 0)
    for t in range(steps):
        full.fill_actions(fa.data_ptr(), seed=4, t=t)
        full.step(fa.data_ptr())
        for r, e in enumerate(halves):
            off, cnt = gymrs.shard_range(n, 2, r)
            ha = dev_actions(cnt, 0)
            e.fill_actions(ha.data_ptr(), seed=4, t=t)
            e.step(ha.data_ptr())
            e.sync()
    full.sync()
    cat = np.concatenate([e.get_state() for e in halves], axis=1)
    assert np.array_equal(cat.view(np.uint32), full.get_state().view(np.uint32))
    assert np.array_equal(sum(e.stats() for e in halves), full.stats())
    for e in halves + [full]:
        e.close()


def test_native_rccl_allreduce_single_rank(gymrs):
    """The C ABI's RCCL path with a 1-rank communicator (the multi-rank case needs >1 GPU)."""
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
    with gymrs.BatchedEngine(0, 4096, flags=flags) as eng:
        eng.reset(seed=1)
        a = dev_actions(4096, 0)
        for t in range(30):
            eng.fill_actions(a.data_ptr(), seed=2, t=t)
            eng.step(a.data_ptr())
        local = eng.stats()
        eng.comm_init(1, 0, eng.comm_unique_id())
        assert np.array_equal(eng.allreduce_stats(), local)


@pytest.mark.parametrize("kind,log2n", [(0, 20), (1, 20), (2, 22)])
def test_full_size_properties(kind, log2n, gymrs, twin):
    """BASELINE.json configs 2-4 at full size: properties that hold at any size.
    - the first 4096 and last 4096 lanes match the f32 twin bit for bit (lanes are independent);
    - every auto-reset lane sits inside the reset box, every other lane inside the live region;
    - rewards are the env's constant; episode accounting is conserved:
      CartPole return == length, MountainCar return == -length, Pendulum episodes == n * steps/limit."""
    n, steps = 1 << log2n, 25
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS | (gymrs.TIME_LIMIT if kind == 2 else 0)
    params = gymrs.engine.default_params(kind)
    if kind == 2:
        params.max_episode_steps = 10
    eng = gymrs.BatchedEngine(kind, n, flags=flags, params=params)
    eng.reset(seed=123)
    head = TwinEngine(twin, kind, 4096, params, flags=flags, gid0=0)
    tail = TwinEngine(twin, kind, 4096, params, flags=flags, gid0=n - 4096)
    head.reset(123)
    tail.reset(123)
    acts = dev_actions(n, kind)
    for t in range(steps):
        eng.fill_actions(acts.data_ptr(), seed=6, t=t)
        eng.step(acts.data_ptr())
        head.step(head.fill_actions(6, t))
        tail.step(tail.fill_actions(6, t))
    eng.sync()
    assert np.array_equal(eng.get_state(0, 4096).view(np.uint32), head.get_state().view(np.uint32))
    assert np.array_equal(eng.get_state(n - 4096, 4096).view(np.uint32), tail.get_state().view(np.uint32))
    st = eng.get_state()
    reward, done, trunc = eng.get_step_result()
    stats = eng.stats()
    assert stats[3] == float(n) * steps
    if kind == 0:
        assert (reward == 1.0).all()
        fresh = done == 1
        assert (np.abs(st[:, fresh]) < 0.05).all()
        assert (np.abs(st[0, ~fresh]) <= 2.4).all() and (np.abs(st[2, ~fresh]) <= np.float32(0.20943951023931953)).all()
        assert stats[0] == stats[1] and 0.02 < done.mean() < 0.12  # ~1/22 of the lanes finish per step
    elif kind == 1:
        assert (reward == -1.0).all() and stats[0] == -stats[1]
        assert (st[0] >= -1.2).all() and (st[0] <= 0.6).all() and (np.abs(st[1]) <= np.float32(0.07)).all()
    else:
        assert not done.any()
        assert stats[2] == float(n) * (steps // 10) and stats[1] == float(n) * 10 * (steps // 10)
        assert (np.abs(st[1]) <= 8.0).all()
        assert trunc.sum() == 0  # step 25 is not a multiple of 10
    eng.close()


def test_done_counts_vs_f64_oracle_over_many_steps(gymrs, oracle):
    """north_star: 'bit-exactly on integer step/done counts'.  2^20 lanes x 30 steps = 3.1e7 lane-steps: after
    every GPU step the f64 oracle advances the SAME f32 states by one step; done flags must agree except when
    the f64 state sits within 1e-5 of a threshold (SURVEY H2 expects O(10-100) such events per 1e9 lane-steps),
    and those events are counted and bounded, not hidden."""
    n, steps = 1 << 20, 30
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
    eng = gymrs.BatchedEngine(gymrs.CARTPOLE, n, flags=flags)
    eng.reset(seed=31)
    acts = dev_actions(n, 0)
    total_done_gpu = total_done_ref = in_band = worst = 0
    for t in range(steps):
        before = eng.get_state().astype(np.float64)
        eng.fill_actions(acts.data_ptr(), seed=8, t=t)
        eng.step(acts.data_ptr())
        eng.sync()
        a = acts.cpu().numpy()
        ref_r, ref_d, bad = oracle.cartpole_step_batch(before, np.zeros(n, np.uint8), a)
        _, done, _ = eng.get_step_result()
        assert bad == 0
        live = (done == 0) & (ref_d == 0)  # re-armed lanes hold a fresh state on the GPU
        got = eng.get_state().astype(np.float64)
        worst = max(worst, float((np.abs(got - before)[:, live] / np.maximum(np.abs(before[:, live]), 1.0)).max()))
        mism = np.nonzero(done != ref_d)[0]
        near = (np.abs(np.abs(before[0]) - 2.4) < 1e-5) | (np.abs(np.abs(before[2]) - 0.20943951023931953) < 1e-5)
        assert near[mism].all(), f"step {t}: {len(mism)} done mismatches outside the threshold band"
        in_band += len(mism)
        total_done_gpu += int(done.sum())
        total_done_ref += int(ref_d.sum())
    assert worst <= TOL
    print(f"done flags: gpu {total_done_gpu}, f64 oracle {total_done_ref}, in-band mismatches {in_band} of {n * steps} lane-steps")
    assert in_band <= 20 and abs(total_done_gpu - total_done_ref) <= in_band
    assert eng.stats()[2] == total_done_gpu  # the engine's integer episode count == the number of done flags raised
    eng.close()


@pytest.mark.parametrize("kind", [0, 1])
def test_done_flag_is_the_f64_compare_of_the_f32_state_on_gpu(kind, gymrs):
    """VERDICT r2 weak #2 / cartpole.rs:450-453, mountain_car.rs:422: the kernel's thresholds are the largest f32 <= (goal:
    smallest f32 >=) the f64 values, so for the f32 state the kernel itself produced the flag IS the reference's f64 compare.
    2^20 lanes without auto-reset (the state that raised the flag stays visible), NO tolerance band, 12 steps = 1.2e7 flags;
    the twin-side counterpart (incl. the f32 neighbours of each threshold) is tests/test_twin_vs_oracle.py."""
    n = 1 << 20
    rng = np.random.default_rng(900 + kind)
    P = gymrs.engine.default_params(kind)
    if kind == 0:
        st = np.stack([rng.uniform(-2.45, 2.45, n), rng.uniform(-1, 1, n), rng.uniform(-0.215, 0.215, n), rng.uniform(-1, 1, n)])
    else:
        st = np.stack([rng.uniform(0.4, 0.6, n), rng.uniform(-0.02, 0.07, n)])
    acts = dev_actions(n, kind)
    raised = 0
    with gymrs.BatchedEngine(kind, n, flags=0) as eng:
        eng.reset(seed=3)
        eng.set_state(st.astype(np.float32))
        for t in range(12):
            eng.fill_actions(acts.data_ptr(), seed=4, t=t)
            eng.step(acts.data_ptr())
            eng.sync()
            s = eng.get_state().astype(np.float64)
            _, done, _ = eng.get_step_result()
            with np.errstate(invalid="ignore"):
                if kind == 0:
                    want = ~(np.abs(s[0]) <= P.x_threshold) | ~(np.abs(s[2]) <= P.theta_threshold_radians)
                else:
                    want = ~(s[0] < P.goal_position) & ~(s[1] < P.goal_velocity)
            assert np.array_equal(done.astype(bool), want), (t, np.nonzero(done.astype(bool) != want)[0][:10])
            raised += int(done.sum())
    assert 0.02 * 12 * n < raised < 0.98 * 12 * n


def test_reset_distribution_on_gpu(gymrs):
    """Distribution parity of the Philox-based reset with the reference's Uniform::new(low, high) sampling
    (cartpole.rs:352-364, mountain_car.rs:175-190): per-component KS test against U[low, high), independence
    across components and lanes (correlations), half-open interval, and a different draw per episode."""
    from scipy import stats

    n = 1 << 20
    with gymrs.BatchedEngine(gymrs.CARTPOLE, n, flags=gymrs.AUTO_RESET) as eng:
        eng.reset(seed=2026)
        s = eng.get_state().astype(np.float64)
        lo, hi = float(np.float32(-0.05)), float(np.float32(0.05))  # the f32 images of the reference's bounds
        for j in range(4):
            assert s[j].min() >= lo and s[j].max() < hi
            assert stats.kstest(s[j], "uniform", args=(-0.05, 0.1)).pvalue > 1e-4
        c = np.corrcoef(s)
        assert np.abs(c - np.eye(4)).max() < 5e-3
        assert abs(np.corrcoef(s[0][:-1], s[0][1:])[0, 1]) < 5e-3  # neighbouring lanes are independent
        # auto-reset draws: run until plenty of lanes were re-armed, then test the fresh ones of one step
        acts = dev_actions(n, 0)
        for t in range(40):
            eng.fill_actions(acts.data_ptr(), seed=5, t=t)
            eng.step(acts.data_ptr())
        eng.sync()
        _, done, _ = eng.get_step_result()
        fresh = eng.get_state().astype(np.float64)[:, done == 1]
        assert fresh.shape[1] > 20_000
        for j in range(4):
            assert fresh[j].min() >= lo and fresh[j].max() < hi
            assert stats.kstest(fresh[j], "uniform", args=(-0.05, 0.1)).pvalue > 1e-4
    with gymrs.BatchedEngine(gymrs.MOUNTAIN_CAR, n) as eng:
        eng.reset(seed=7)
        s = eng.get_state().astype(np.float64)
        assert stats.kstest(s[0], "uniform", args=(-0.6, 0.2)).pvalue > 1e-4 and (s[1] == 0).all()


def test_custom_physics_params_and_semi_implicit_integrator(gymrs, oracle, twin, golden):
    """The reference's physics constants are mutable `pub` fields (cartpole.rs:53-82) and
    KinematicsIntegrator::Other switches to semi-implicit Euler (cartpole.rs:436-441): both must reach the
    kernel (as kernel arguments), match the f64 oracle with the same parameters and stay bit-exact vs the twin."""
    n = 50_000
    rng = np.random.default_rng(77)
    st, act = random_states(0, n, rng)
    for integrator, tweak in ((1, {}), (0, {"gravity": 3.7, "masspole": 0.25, "length": 0.8, "force_mag": 7.5, "tau": 0.01}),
                              (1, {"masscart": 2.0, "x_threshold": 1.0, "theta_threshold_radians": 0.1})):
        p = gymrs.engine.default_params(0)
        op = oracle.cartpole_params()
        p.kinematics_integrator = integrator
        op.kinematics_integrator = integrator
        for k, v in tweak.items():
            setattr(p, k, v)
            setattr(op, k, v)
        with gymrs.BatchedEngine(0, n, flags=0, params=p) as eng:
            eng.reset(seed=1)
            eng.set_state(st)
            eng.step_host(act)
            got = eng.get_state()
            _, done, _ = eng.get_step_result()
        ref = st.astype(np.float64).copy()
        _, ref_d, _ = oracle.cartpole_step_batch(ref, np.zeros(n, np.uint8), act, op)
        assert mixed_err(got, ref).max() <= TOL, (integrator, tweak)
        near = (np.abs(np.abs(ref[0]) - op.x_threshold) < 1e-5) | (np.abs(np.abs(ref[2]) - op.theta_threshold_radians) < 1e-5)
        assert near[done != ref_d].all()
        tw = TwinEngine(twin, 0, n, p, flags=0)
        tw.reset(1)
        tw.set_state(st)
        tw.step(act)
        assert np.array_equal(got.view(np.uint32), tw.get_state().view(np.uint32))
    # the committed semi-implicit golden vectors
    cases = golden("cartpole")["semi_implicit"]
    p = gymrs.engine.default_params(0)
    p.kinematics_integrator = 1
    with gymrs.BatchedEngine(0, len(cases), params=p) as eng:
        eng.set_state(np.array([c["state"] for c in cases], np.float32).T)
        eng.step_host(np.array([c["action"] for c in cases], np.uint8))
        got = eng.get_state()
    assert mixed_err(got, np.array([c["next"] for c in cases]).T).max() <= 2e-6
    # MountainCar with other constants
    pm = gymrs.engine.default_params(1)
    om = oracle.mountain_car_params()
    for k, v in {"force": 0.002, "gravity": 0.003, "max_speed": 0.05, "goal_position": 0.45}.items():
        setattr(pm, k, v)
        setattr(om, k, v)
    st, act = random_states(1, n, rng)
    with gymrs.BatchedEngine(1, n, params=pm) as eng:
        eng.set_state(st)
        eng.step_host(act)
        got = eng.get_state()
    ref = st.astype(np.float64).copy()
    oracle.mountain_car_step_batch(ref, act, om)
    assert mixed_err(got, ref).max() <= TOL


def test_external_stream_policy_to_step_without_host_sync(gymrs, twin):
    """Drop-in use: the engine rides the host framework's HIP stream (gymrs_set_stream), so a policy that
    produces actions on that stream feeds step() with no host synchronisation in between."""
    n, steps = 65_536, 25
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
    stream = torch.cuda.Stream()
    eng = gymrs.BatchedEngine(gymrs.CARTPOLE, n, flags=flags)
    eng.set_stream(stream.cuda_stream)
    assert eng.stream == stream.cuda_stream
    tw = TwinEngine(twin, 0, n, eng.params, flags=flags)
    eng.reset(seed=3)
    tw.reset(3)
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(1234)
    history = []
    with torch.cuda.stream(stream):
        for t in range(steps):
            # "policy": a torch op on the shared stream (here random, it could read eng.obs_ptrs())
            a = torch.randint(0, 2, (n,), dtype=torch.uint8, device="cuda:0", generator=gen)
            eng.step(a.data_ptr())
            history.append(a)  # keep alive; copied back only after the loop
    eng.sync()
    for a in history:
        tw.step(a.cpu().numpy())
    assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32))
    assert np.array_equal(eng.stats(), tw.stats())
    eng.close()


def test_step_many_ring_pendulum_f32_actions(gymrs, twin):
    n, steps, nbuf = 12_345, 45, 3
    flags = gymrs.AUTO_RESET | gymrs.TIME_LIMIT | gymrs.TRACK_STATS
    p = gymrs.engine.default_params(2)
    p.max_episode_steps = 20
    eng = gymrs.BatchedEngine(gymrs.PENDULUM, n, flags=flags, params=p)
    tw = TwinEngine(twin, 2, n, p, flags=flags)
    eng.reset(seed=9)
    tw.reset(9)
    bufs = torch.empty((nbuf, n), dtype=torch.float32, device="cuda:0")
    for b in range(nbuf):
        eng.fill_actions(bufs[b].data_ptr(), seed=4, t=b)
    eng.step_many(bufs.data_ptr(), n * 4, nbuf, steps)
    eng.sync()
    for t in range(steps):
        tw.step(tw.fill_actions(4, t % nbuf))
    assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32))
    assert np.array_equal(eng.get_obs().view(np.uint32), tw.get_obs().view(np.uint32))
    gr, gd, gt = eng.get_step_result()
    tr, td, tt = tw.get_result()
    assert np.array_equal(gr.view(np.uint32), tr.view(np.uint32)) and np.array_equal(gt, tt) and not gd.any()
    gs, ts = eng.stats(), tw.stats()
    assert gs[1] == ts[1] == n * 40 and gs[2] == ts[2] == n * 2 and gs[0] == pytest.approx(ts[0], rel=1e-5)
    eng.close()


@pytest.mark.parametrize("auto", [True, False])
def test_pendulum_flags_follow_the_twin_on_every_step(gymrs, twin, auto):
    """Pendulum never terminates and truncates all lanes at once, so the kernel rewrites `done` / `truncated` only
    when their (uniform) value changes; the arrays must still read right after EVERY step, across the time limit,
    after a rollout, and after a snapshot is loaded into another engine."""
    n = 3001
    flags = gymrs.TIME_LIMIT | ((gymrs.AUTO_RESET | gymrs.TRACK_STATS) if auto else 0)
    p = gymrs.engine.default_params(2)
    p.max_episode_steps = 5
    eng = gymrs.BatchedEngine(2, n, flags=flags, params=p)
    tw = TwinEngine(twin, 2, n, p, flags=flags)
    eng.reset(seed=6)
    tw.reset(6)
    buf = torch.empty(n, dtype=torch.float32, device="cuda:0")

    def check(tag):
        gr, gd, gt = eng.get_step_result()
        tr, td, tt = tw.get_result()
        assert np.array_equal(gd, td) and np.array_equal(gt, tt), tag
        assert np.array_equal(gr.view(np.uint32), tr.view(np.uint32)), tag

    t = 0
    for t in range(13):
        eng.fill_actions(buf.data_ptr(), seed=3, t=t)
        eng.step(buf.data_ptr())
        tw.step(tw.fill_actions(3, t))
        check(("step", t))
    eng.rollout(4, action_seed=3, action_t0=13)   # ends exactly on / off a limit step depending on `auto`
    for t in range(13, 17):
        tw.step(tw.fill_actions(3, t))
    check("rollout")
    other = gymrs.BatchedEngine(2, n, flags=flags, params=p)
    other.reset(seed=99)
    other.restore(eng.snapshot())
    for e in (eng, other):
        e.fill_actions(buf.data_ptr(), seed=3, t=17)
        e.step(buf.data_ptr())
    tw.step(tw.fill_actions(3, 17))
    check("after snapshot, original")
    gr, gd, gt = other.get_step_result()
    assert np.array_equal(gt, tw.get_result()[2]) and np.array_equal(gd, tw.get_result()[1])
    eng.close()
    other.close()


def test_mountain_car_reward_array_stays_right_around_invalid_actions(gymrs, twin):
    """MountainCar's constant reward is not rewritten while the array already holds it; a step with an invalid
    action pays 0 on that lane (lane untouched), and the step after must put -1 back."""
    n = 5000
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
    eng = gymrs.BatchedEngine(1, n, flags=flags)
    eng.reset(seed=3)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    for t in range(3):
        eng.fill_actions(buf.data_ptr(), seed=1, t=t)
        eng.step(buf.data_ptr())
        assert (eng.get_step_result()[0] == -1.0).all()
    bad = torch.ones(n, dtype=torch.uint8, device="cuda:0")
    bad[1234] = 7
    torch.cuda.synchronize()  # (torch wrote `bad` on its own stream)
    eng.step(bad.data_ptr())
    with pytest.raises(gymrs.InvalidActionError):
        eng.sync()
    r = eng.get_step_result()[0]
    assert r[1234] == 0.0 and (np.delete(r, 1234) == -1.0).all()
    eng.fill_actions(buf.data_ptr(), seed=1, t=4)
    eng.step(buf.data_ptr())
    eng.sync()
    assert (eng.get_step_result()[0] == -1.0).all()
    # a change of launch shape and a snapshot load both start from "rewrite everything"
    eng.set_tuning(8)
    eng.step(buf.data_ptr())
    other = gymrs.BatchedEngine(1, n, flags=flags)
    other.reset(seed=8)
    other.restore(eng.snapshot())
    other.step(buf.data_ptr())
    assert (eng.get_step_result()[0] == -1.0).all() and (other.get_step_result()[0] == -1.0).all()
    eng.close()
    other.close()


@pytest.fixture
def elide_reward_forced(monkeypatch):
    """Engines created inside see GYMRS_DEV_ELIDE_REWARD (read at creation): CartPole's reward elision is a matter of size (from 128 MiB
    per step on, 2^22 lanes); the knob puts both code paths within reach of a 9001-lane test."""
    def force(value):
        monkeypatch.setenv("GYMRS_DEV_ELIDE_REWARD", value)
    return force


@pytest.mark.parametrize("flag_set", ["A", "AS", "AST"])
def test_cartpole_reward_elision_changes_nothing_but_the_stores(gymrs, twin, elide_reward_forced, flag_set):
    """CartPole under auto-reset pays 1.0 on every step; big engines stop rewriting it (StepArgs::elide_reward).  Forced on at a small size:
    HIP launches and chains against the twin -- state, rewards, flags, statistics -- with a 45-step time limit in the third flag set."""
    import json

    n, nbuf = 9001, 4
    flags = gymrs.AUTO_RESET | (gymrs.TRACK_STATS if "S" in flag_set else 0) | (gymrs.TIME_LIMIT if "T" in flag_set else 0)
    p = gymrs.engine.default_params(0)
    p.max_episode_steps = 45
    elide_reward_forced("1")
    eng = gymrs.BatchedEngine(0, n, flags=flags, params=p)
    assert json.loads(eng.env_json(0))["gymrs"]["reward_store_elided"] == 1
    elide_reward_forced("0")
    plain = gymrs.BatchedEngine(0, n, flags=flags, params=p)
    assert json.loads(plain.env_json(0))["gymrs"]["reward_store_elided"] == 0
    tw = TwinEngine(twin, 0, n, p, flags=flags)
    for e in (eng, plain):
        e.reset(seed=12)
    tw.reset(12)
    bufs = torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
    for b in range(nbuf):
        eng.fill_actions(bufs[b].data_ptr(), seed=5, t=b)

    def same():
        for e in (eng, plain):
            e.sync()
            assert np.array_equal(e.get_state().view(np.uint32), tw.get_state().view(np.uint32))
            r, d, tr = e.get_step_result()
            tr_, td, tt = tw.get_result()
            assert np.array_equal(r.view(np.uint32), tr_.view(np.uint32)) and np.array_equal(d, td) and np.array_equal(tr, tt)
            if "S" in flag_set:
                assert np.array_equal(e.stats(), tw.stats())

    for t in range(7):
        for e in (eng, plain):
            e.step(bufs[t % nbuf].data_ptr())
        tw.step(tw.fill_actions(5, t % nbuf))
        same()
    for e in (eng, plain):
        e.step_many(bufs.data_ptr(), n, nbuf, 61)  # a chain
    for t in range(61):
        tw.step(tw.fill_actions(5, t % nbuf))
    same()
    eng.close()
    plain.close()


def test_cartpole_reward_array_stays_right_around_invalid_actions_when_elided(gymrs, elide_reward_forced):
    """The elided store's flag must fall when a step pays something else (an invalid action pays 0 and leaves the lane alone), and the next
    step must put 1.0 back; a change of launch shape and a snapshot load start from "rewrite everything"."""
    n = 5000
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
    elide_reward_forced("1")
    eng = gymrs.BatchedEngine(0, n, flags=flags)
    eng.reset(seed=3)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    for t in range(3):
        eng.fill_actions(buf.data_ptr(), seed=1, t=t)
        eng.step(buf.data_ptr())
        assert (eng.get_step_result()[0] == 1.0).all()
    bad = torch.ones(n, dtype=torch.uint8, device="cuda:0")
    bad[1234] = 7
    torch.cuda.synchronize()  # (torch wrote `bad` on its own stream)
    eng.step(bad.data_ptr())
    with pytest.raises(gymrs.InvalidActionError):
        eng.sync()
    r = eng.get_step_result()[0]
    assert r[1234] == 0.0 and (np.delete(r, 1234) == 1.0).all()
    eng.fill_actions(buf.data_ptr(), seed=1, t=4)
    eng.step(buf.data_ptr())
    eng.sync()
    assert (eng.get_step_result()[0] == 1.0).all()
    eng.set_tuning(8)
    eng.step(buf.data_ptr())
    other = gymrs.BatchedEngine(0, n, flags=flags)
    other.reset(seed=8)
    other.restore(eng.snapshot())
    other.step(buf.data_ptr())
    assert (eng.get_step_result()[0] == 1.0).all() and (other.get_step_result()[0] == 1.0).all()
    eng.close()
    other.close()
