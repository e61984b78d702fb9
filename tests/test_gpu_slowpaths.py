"""The device-side SLOW paths on the device: states that leave the straight-line code --
angles beyond pi/4 (f32 Cody-Waite), beyond 200 rad (f64 Cody-Waite), beyond 2^28*pi/2 (Payne-Hanek), NaN / +-inf
states (OrderedFloat's total order, SURVEY Q10: cartpole.rs:450-453, mountain_car.rs:416-422), MountainCar exactly on
its clip bounds with +-0 velocity (the `==` wall rule, Q11: mountain_car.rs:418-420), Pendulum beyond 200 rad
(angle_normalize's f64 branch) with torques beyond +-2 -- run on gfx950 through the C ABI and are compared with
  * the f64 oracle: flags exactly, finite values within 1e-6 * max(|ref|, 1);
  * the CPU f32 twin: bit for bit (any NaN equals any NaN: the sign/payload of a generated NaN is not specified).
Plus a direct Philox known-answer test on the GPU (Random123's vector, tests/golden/philox.json)."""
import numpy as np
import pytest
import torch

from oracle.bindings import TwinEngine

pytestmark = pytest.mark.gpu
TOL = 1e-6
THETAS = [1.0, 3.0, 199.0, 201.0, 1e4, 4e8, 1e30]


def same_bits_or_both_nan(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    both_nan = np.isnan(a) & np.isnan(b)
    return bool(np.all(both_nan | (a.view(np.uint32) == b.view(np.uint32))))


def close_or_same_special(got, ref):
    """finite reference: mixed 1e-6; NaN reference: NaN; infinite reference: the same infinity."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    fin = np.isfinite(ref)
    ok = np.empty(ref.shape, bool)
    ok[fin] = np.abs(got[fin] - ref[fin]) <= TOL * np.maximum(np.abs(ref[fin]), 1.0)
    ok[~fin] = (np.isnan(ref[~fin]) & np.isnan(got[~fin])) | (got[~fin] == ref[~fin])
    return ok


def step_all(gymrs, twin, kind, st, act, flags=0, vec=4):
    n = st.shape[1]
    with gymrs.BatchedEngine(kind, n, flags=flags, lanes_per_thread=vec) as eng:
        eng.reset(seed=5)
        eng.set_state(st)
        eng.step_host(act)
        got = eng.get_state()
        reward, done, trunc = eng.get_step_result()
        obs = eng.get_obs()
    tw = TwinEngine(twin, kind, n, gymrs.engine.default_params(kind), flags=flags)
    tw.reset(5)
    tw.set_state(st)
    tw.step(act)
    t_r, t_d, t_t = tw.get_result()
    return (got, reward, done, trunc, obs), (tw.get_state(), t_r, t_d, t_t, tw.get_obs())


@pytest.mark.parametrize("vec", [4, 8])
def test_cartpole_large_angles_and_non_finite_states(gymrs, twin, oracle, vec):
    inf, nan = np.inf, np.nan
    rows = []
    for th in THETAS:
        for sign in (1.0, -1.0):
            for a in (0, 1):
                rows.append(((0.3, -0.7, sign * th, 0.9), a))
    rows += [((nan, 0.1, 0.01, 0.2), 0), ((0.1, 0.1, nan, 0.2), 1), ((inf, 0.1, 0.01, 0.2), 0), ((-inf, 0.1, 0.01, 0.2), 1),
             ((0.1, 0.1, inf, 0.2), 0), ((0.1, 0.1, -inf, 0.2), 1), ((0.1, nan, 0.01, 0.2), 0), ((0.1, 0.1, 0.01, inf), 1),
             # around the thresholds (the velocities are 0, so the step leaves x / theta where they are): fl32(2.4) = 2.4000001 and
             # fl32(theta_thr) = 0.20943952 lie ABOVE the f64 thresholds -> the reference ends these lanes (cartpole.rs:450-453);
             # the largest f32 values below them (2.3999999, 0.2094395) stay alive: the compares are strict
             ((2.4, 0.0, 0.0, 0.0), 1), ((-2.4, 0.0, 0.0, 0.0), 0), ((2.3999998569488525, 0.0, 0.0, 0.0), 1), ((-2.3999998569488525, 0.0, 0.0, 0.0), 0),
             ((0.0, 0.0, 0.20943951606750488, 0.0), 1), ((0.0, 0.0, -0.20943951606750488, 0.0), 0),
             ((0.0, 0.0, 0.20943950116634369, 0.0), 1), ((0.0, 0.0, -0.20943950116634369, 0.0), 0),
             ((0.0, 0.0, 0.7853981, 3.0), 0), ((0.0, 0.0, 0.7853982, 3.0), 1)]
    # pad with ordinary lanes so that the special ones share wavefronts with common-path lanes (and with each other)
    rng = np.random.default_rng(1)
    while len(rows) < 300:
        rows.append(((rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-0.1, 0.1), rng.uniform(-1, 1)), int(rng.integers(0, 2))))
    st = np.array([r[0] for r in rows], np.float32).T.copy()
    act = np.array([r[1] for r in rows], np.uint8)
    (got, reward, done, _, _), (t_st, t_r, t_d, _, _) = step_all(gymrs, twin, 0, st, act, vec=vec)
    # bit-exact with the twin
    assert same_bits_or_both_nan(got, t_st) and np.array_equal(done, t_d) and np.array_equal(reward, t_r)
    # against the f64 oracle
    ref = st.astype(np.float64).copy()
    ref_r, ref_d, bad = oracle.cartpole_step_batch(ref, np.zeros(len(rows), np.uint8), act)
    assert bad == 0
    # The kernel compares against the largest f32 <= each f64 threshold (gymrs_physics.h f32_not_above), which for an f32
    # state IS the reference's f64 compare: the flags agree on every lane, the ones placed one f32 step either side of the
    # thresholds included (round 2 rounded the thresholds to nearest and tolerated 3 mismatches here).
    assert np.array_equal(done, ref_d), np.nonzero(done != ref_d)
    on = {2.4: 1, -2.4: 1, 2.3999998569488525: 0, -2.3999998569488525: 0}
    for i, r in enumerate(rows):
        if r[0][0] in on and r[0][1] == 0.0:
            assert done[i] == on[r[0][0]], (i, r)
        if r[0][3] == 0.0 and abs(r[0][2]) == 0.20943951606750488:
            assert done[i] == 1, (i, r)
        if r[0][3] == 0.0 and abs(r[0][2]) == 0.20943950116634369:
            assert done[i] == 0, (i, r)
    # and the flag equals the reference's f64 compares applied to the f32 state the kernel produced (no tolerance)
    with np.errstate(invalid="ignore"):
        g64 = got.astype(np.float64)
        want = ~(np.abs(g64[0]) <= 2.4) | ~(np.abs(g64[2]) <= 0.20943951023931953)
    assert np.array_equal(done.astype(bool), want)
    assert close_or_same_special(got, ref).all(), np.nonzero(~close_or_same_special(got, ref))
    assert np.array_equal(reward, ref_r.astype(np.float32))
    # Q10: a NaN anywhere in x or theta terminates; every large angle terminates
    nan_rows = [i for i, r in enumerate(rows) if np.isnan(r[0][0]) or np.isnan(r[0][2])]
    assert len(nan_rows) == 2 and done[nan_rows].all()
    assert done[: 4 * len(THETAS)].all()


def test_mountain_car_bounds_zero_velocity_and_non_finite(gymrs, twin, oracle):
    inf, nan = np.inf, np.nan
    pos = [-1.2, 0.6, 0.5, -1.2000000476837158, 0.5000000001, 0.49999997, -0.5]
    vel = [0.0, -0.0, 0.07, -0.07, 1e-9, -1e-9]
    rows = [((p, v), a) for p in pos for v in vel for a in (0, 1, 2)]
    rows += [((nan, 0.0), 1), ((0.0, nan), 2), ((inf, 0.0), 0), ((-inf, 0.0), 2), ((0.0, inf), 1), ((0.0, -inf), 1),
             ((70.0, 0.01), 2), ((-1e6, 0.0), 0), ((4e8, 0.0), 1)]  # cos(3 * position) on the long reductions
    st = np.array([r[0] for r in rows], np.float32).T.copy()
    act = np.array([r[1] for r in rows], np.uint8)
    (got, reward, done, _, _), (t_st, t_r, t_d, _, _) = step_all(gymrs, twin, 1, st, act)
    assert same_bits_or_both_nan(got, t_st) and np.array_equal(done, t_d) and np.array_equal(reward, t_r)
    ref = st.astype(np.float64).copy()
    ref_r, ref_d, bad = oracle.mountain_car_step_batch(ref, act)
    assert bad == 0
    # f32 rounds 0.5000000001 to exactly 0.5 and the f64 oracle sees the f32 value, so the flags must agree everywhere
    # except within 1e-5 of a threshold after a non-trivial update
    near = (np.abs(ref[0] - 0.5) < 1e-5) | (np.abs(ref[1]) < 1e-5)
    mism = done != ref_d
    assert near[mism].all(), np.nonzero(mism & ~near)
    assert close_or_same_special(got, ref).all(), np.nonzero(~close_or_same_special(got, ref))
    assert (reward == -1.0).all()
    # wall rule (Q11): a lane that ends on the lower bound never keeps a negative velocity
    on_wall = got[0] == np.float32(-1.2)
    assert on_wall.any() and (got[1][on_wall] >= 0).all()
    # NaN: clip() returns its upper bound for NaN (OrderedFloat: NaN is the maximum).  A NaN position makes cos() NaN,
    # so velocity -> max_speed and position -> max_position: done.  A NaN velocity alone becomes max_speed: a plain step.
    i_nan = [i for i, r in enumerate(rows) if np.isnan(r[0][0]) or np.isnan(r[0][1])]
    assert np.array_equal(done[i_nan], ref_d[i_nan]) and list(done[i_nan]) == [1, 0]
    assert np.isfinite(got[:, i_nan]).all() and got[0, i_nan[0]] == np.float32(0.6) and got[1, i_nan[1]] == np.float32(0.07)


def test_pendulum_beyond_200_rad_and_clipped_torque(gymrs, twin, oracle):
    inf, nan = np.inf, np.nan
    rows = []
    # (beyond ~1e15 rad an f64 floor-modulo has no correct digits left in the oracle either; the spec-derived env reaches
    # |theta| <= pi + 200 * 8 * 0.05 = 83 rad inside an episode)
    for th in [0.5, 3.0, 3.1415927, 199.0, 200.0, 200.00002, 201.0, 1e4, 4e8]:
        for sign in (1.0, -1.0):
            for thd in (0.0, -8.0, 7.9):
                for u in (-5.0, -2.0, 0.3, 2.0, 1e9):
                    rows.append(((sign * th, thd), u))
    rows += [((nan, 0.0), 0.0), ((0.0, nan), 1.0), ((inf, 0.0), 0.0), ((1.0, 0.5), nan), ((1.0, 0.5), inf), ((1.0, 0.5), -inf)]
    st = np.array([r[0] for r in rows], np.float32).T.copy()
    act = np.array([r[1] for r in rows], np.float32)
    (got, reward, done, trunc, obs), (t_st, t_r, t_d, t_t, t_obs) = step_all(gymrs, twin, 2, st, act)
    assert same_bits_or_both_nan(got, t_st) and same_bits_or_both_nan(reward, t_r) and same_bits_or_both_nan(obs, t_obs)
    assert not done.any() and not trunc.any()
    ref = st.astype(np.float64).copy()
    ref_r, oc, os_ = oracle.pendulum_step_batch(ref, act.astype(np.float64))
    fin = np.isfinite(ref).all(axis=0) & np.isfinite(ref_r)
    assert fin.sum() >= len(rows) - 8
    assert close_or_same_special(got[:, fin], ref[:, fin]).all()
    # the cost's angle term: angle_normalize in f32 carries |theta| * 2^-24 of representation error into a value <= pi,
    # which the square amplifies by <= 2 pi: tolerance 1e-6 * max(|reward|, 1) holds for |theta| <= 200; beyond that the
    # f64 branch is exact for the f32 input, so the same bound holds
    assert (np.abs(reward[fin].astype(np.float64) - ref_r[fin]) <= 2e-6 * np.maximum(np.abs(ref_r[fin]), 1.0)).all()
    # observation columns of the NEW state (cos, sin, theta_dot)
    th_new = got[0, fin].astype(np.float64)
    assert np.abs(obs[0, fin] - np.cos(th_new)).max() <= TOL and np.abs(obs[1, fin] - np.sin(th_new)).max() <= TOL
    assert np.array_equal(obs[2].view(np.uint32), got[1].view(np.uint32))
    # NaN torque: clip() returns the upper bound for NaN (OrderedFloat) -> same as +max_torque
    i_nan_u = [i for i, r in enumerate(rows) if np.isnan(r[1])][0]
    i_inf_u = [i for i, r in enumerate(rows) if r[1] == inf][0]
    assert np.isfinite(got[:, i_nan_u]).all() and np.array_equal(got[:, i_nan_u], got[:, i_inf_u])


def test_slow_paths_inside_auto_reset_waves(gymrs, twin):
    """A wave that holds run-away lanes takes the general per-lane code AND the compaction path in the same step."""
    n, flags = 5000, None
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
    rng = np.random.default_rng(7)
    st = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(-0.1, 0.1, n), rng.uniform(-1, 1, n)]).astype(np.float32)
    special = rng.choice(n, 400, replace=False)
    st[2, special[:100]] = 1e30
    st[2, special[100:200]] = 250.0
    st[0, special[200:300]] = np.nan
    st[3, special[300:]] = np.inf
    eng = gymrs.BatchedEngine(0, n, flags=flags, global_env_offset=77)
    tw = TwinEngine(twin, 0, n, gymrs.engine.default_params(0), flags=flags, gid0=77)
    eng.reset(seed=3)
    tw.reset(3)
    eng.set_state(st)
    tw.set_state(st)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    for t in range(12):
        eng.fill_actions(buf.data_ptr(), seed=4, t=t)
        eng.step(buf.data_ptr())
        tw.step(tw.fill_actions(4, t))
        eng.sync()
        assert same_bits_or_both_nan(eng.get_state(), tw.get_state()), t
        assert np.array_equal(eng.get_step_result()[1], tw.get_result()[1]), t
    assert np.array_equal(eng.stats(), tw.stats())
    assert np.isfinite(eng.get_state()).all()  # every poisoned lane terminated on its first step and was re-armed
    eng.close()


def test_philox_known_answer_on_the_gpu(gymrs, twin, golden):
    """Direct: a reset box [0, 2^24) makes uniform_in_box the identity on the top 24 bits of each Philox word, so lane
    `gid` after reset(seed) shows philox4x32_10(counter = (gid lo, gid hi, 0, 0), key = (seed lo, seed hi)) >> 8."""
    kat = golden("philox")["random123_kat"][0]
    assert kat["ctr"] == [0, 0, 0, 0] and kat["key"] == [0, 0]
    box = [0.0] * 4 + [16777216.0] * 4
    with gymrs.BatchedEngine(0, 8, flags=0, global_env_offset=0) as eng:
        eng.reset(seed=0, options=box)
        lane0 = eng.get_state()[:, 0]
    assert [int(v) for v in lane0] == [w >> 8 for w in kat["out"]]  # Random123 kat_vectors, philox4x32 10 rounds, zeros
    # counters and keys the zero vector does not reach: 64-bit global ids and seeds, against the KAT-pinned CPU Philox
    rng = np.random.default_rng(11)
    for _ in range(6):
        gid0 = int(rng.integers(0, 2**63)) & ~0xFF
        seed = int(rng.integers(0, 2**63)) * 2 + 1
        n = 300
        with gymrs.BatchedEngine(0, n, flags=0, global_env_offset=gid0) as eng:
            eng.reset(seed=seed, options=box)
            got = eng.get_state().astype(np.uint32)
        for lane in (0, 1, 63, 64, 255, 256, 299):
            gid = gid0 + lane
            want = twin.philox([gid & 0xFFFFFFFF, gid >> 32, 0, 0], [seed & 0xFFFFFFFF, seed >> 32])
            assert [int(v) for v in got[:, lane]] == [int(w) >> 8 for w in want], (gid0, seed, lane)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_small_engines_in_mapped_host_memory_match_large_ones(gymrs, kind):
    """Engines of up to 64 lanes keep their arrays in mapped host memory (results are read with plain loads after a stream
    synchronisation: the single-env mirrors step 3.5x faster); lane i of a 64-lane engine must do exactly what lane i of a
    65-lane engine (device memory) does -- through steps, set_state, clone, snapshot, the fused rollout and an invalid action."""
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS | gymrs.TIME_LIMIT
    p = gymrs.engine.default_params(kind)
    p.max_episode_steps = 17
    small = gymrs.BatchedEngine(kind, 64, flags=flags, params=p, global_env_offset=5)
    large = gymrs.BatchedEngine(kind, 65, flags=flags, params=p, global_env_offset=5)
    dtype = np.float32 if kind == 2 else np.uint8
    rs = np.random.default_rng(kind)

    def acts():
        return rs.uniform(-2, 2, 65).astype(np.float32) if kind == 2 else rs.integers(0, 2, 65).astype(np.uint8)

    def same(a, b):
        assert np.array_equal(a.get_state().view(np.uint32), b.get_state()[:, :64].view(np.uint32))
        ra, rb = a.get_step_result(), b.get_step_result()
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y[:64])

    for e in (small, large):
        e.reset(seed=9)
    same(small, large)
    for t in range(40):
        a = acts()
        small.step_host(a[:64])
        large.step_host(a)
        same(small, large)
    st = large.get_state().copy()
    st[0, :] += 0.01
    small.set_state(st[:, :64])
    large.set_state(st)
    c_small, c_large = small.clone(), large.clone()
    blob = small.snapshot()
    for e in (small, large, c_small, c_large):
        e.rollout(23, action_seed=4, action_t0=7)
    same(small, large)
    same(c_small, c_large)
    r = gymrs.BatchedEngine(kind, 64, flags=flags, params=p)
    r.restore(blob)
    r.rollout(23, action_seed=4, action_t0=7)
    same(r, large)
    assert np.array_equal(small.stats()[1:], c_small.stats()[1:])
    if not (kind == 2):  # captured graphs on the mapped-memory engine (Pendulum with the time limit has no graph mode)
        ring = torch.zeros((8, 65), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()  # torch's stream and the engines' streams are not ordered against each other: wait on both sides of the hand-over
        for b in range(8):
            large.fill_actions(ring[b].data_ptr(), seed=6, t=b)
        large.sync()
        small_ring = ring[:, :64].contiguous()
        torch.cuda.synchronize()
        small.step_many(small_ring.data_ptr(), 64, 8, 80, use_graph=True)
        large.step_many(ring.data_ptr(), 65, 8, 80, use_graph=True)
        same(small, large)
    if kind != 2:  # an invalid action is still reported (through the mapped error flag)
        bad = np.zeros(64, dtype)
        bad[3] = 7
        with pytest.raises(gymrs.InvalidActionError):
            small.step_host(bad)
        small.step_host(np.zeros(64, dtype))  # and the engine goes on
    for e in (small, large, c_small, c_large, r):
        e.close()
