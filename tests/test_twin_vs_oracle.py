"""CPU tests of the f32 arithmetic the kernels use (host build of the product's shared header)
against the f64 oracle.  Tolerance (BASELINE.json north_star): per single step from
identical f32-representable inputs, |f32 - f64| <= 1e-6 * max(|f64|, 1)."""
import numpy as np
import pytest

from oracle.bindings import TwinEngine

TOL = 1e-6


def mixed_err(got, ref):
    return np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)


def test_sincos_accuracy_full_range(twin):
    rng = np.random.default_rng(1)
    xs = np.concatenate([
        rng.uniform(-0.8, 0.8, 200_000), rng.uniform(-100, 100, 200_000), rng.uniform(-1e6, 1e6, 100_000),
        rng.integers(0, 2**32, 400_000, dtype=np.uint64).astype(np.uint32).view(np.float32).astype(np.float64),
        np.array([0.0, -0.0, 0.785398185253143, 0.7853982, 1.5707964, 3.1415927, 6.2831855, 4.2e8, 4.3e8, 1e30, 3.4e38]),
    ]).astype(np.float32)
    xs = xs[np.isfinite(xs)]
    s, c = twin.sincosf(xs)
    xd = xs.astype(np.float64)
    short = (np.abs(xd) > np.pi / 4) & (np.abs(xd) <= 200.0)  # f32 Cody-Waite range: absolute accuracy
    for got, ref in ((s, np.sin(xd)), (c, np.cos(xd))):
        ulp = np.spacing(np.abs(ref.astype(np.float32))).astype(np.float64)
        err = np.abs(got.astype(np.float64) - ref)
        assert (err[~short] / ulp[~short]).max() <= 2.0
        assert err[short].max() <= 1.5e-7
    s, c = twin.sincosf(np.array([np.inf, -np.inf, np.nan], np.float32))
    assert np.isnan(s).all() and np.isnan(c).all()
    s, c = twin.sincosf(np.array([0.0], np.float32))
    assert s[0] == 0.0 and c[0] == 1.0


def test_uniform_between_half_open(twin):
    f = twin.lib.twin_uniform_between
    assert f(0, -0.05, 0.05) == np.float32(-0.05)
    top = f(0xFFFFFFFF, -0.05, 0.05)
    assert top < np.float32(0.05) and top > 0.0499
    assert f(0xFFFFFFFF, -0.6, -0.4) < np.float32(-0.4)
    assert f(0xFFFFFFFF, 0.0, 1e-38) < np.float32(1e-38)  # the result always stays below `high`
    assert f(0x80000000, 0.0, 2.0) == 1.0
    # the host-prepared form the kernels use (3 instructions per draw) is bit-identical
    g = twin.lib.twin_uniform_in_box
    rng = np.random.default_rng(3)
    for lo, hi in ((-0.05, 0.05), (-0.6, -0.4), (-3.1415927, 3.1415927), (0.0, 1e-38), (-1.0, 0.0), (5.0, 6.0), (-2.0, 2.0)):
        for r in [0, 1, 0xFF, 0x100, 0x7FFFFFFF, 0x80000000, 0xFFFFFF00, 0xFFFFFFFF] + [int(v) for v in rng.integers(0, 2**32, 200)]:
            assert g(r, lo, hi) == f(r, lo, hi), (r, lo, hi)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_single_step_within_tolerance_of_f64_oracle(kind, twin, oracle, gymrs):
    n = 200_000
    rng = np.random.default_rng(10 + kind)
    P = gymrs.engine.default_params(kind)
    te = TwinEngine(twin, kind, n, P, flags=0)
    te.reset(1)
    if kind == 0:
        st = np.stack([rng.uniform(-2.4, 2.4, n), rng.uniform(-3, 3, n), rng.uniform(-0.21, 0.21, n), rng.uniform(-3, 3, n)])
        act = rng.integers(0, 2, n).astype(np.uint8)
    elif kind == 1:
        st = np.stack([rng.uniform(-1.2, 0.6, n), rng.uniform(-0.07, 0.07, n)])
        act = rng.integers(0, 3, n).astype(np.uint8)
    else:
        st = np.stack([rng.uniform(-40, 40, n), rng.uniform(-8, 8, n)])
        act = rng.uniform(-2.5, 2.5, n).astype(np.float32)
    st = st.astype(np.float32)
    te.set_state(st)
    te.step(act)
    ref = st.astype(np.float64).copy()
    got_r, got_d, _ = te.get_result()
    if kind == 0:
        ref_r, ref_d, bad = oracle.cartpole_step_batch(ref, np.zeros(n, np.uint8), act)
    elif kind == 1:
        ref_r, ref_d, bad = oracle.mountain_car_step_batch(ref, act)
    else:
        ref_r, oc, os_ = oracle.pendulum_step_batch(ref, act.astype(np.float64))
        ref_d, bad = np.zeros(n, np.uint8), 0
        obs = te.get_obs()
        # obs = (cos, sin) of the NEW theta.  theta is an f32 state that may sit tens of radians from
        # zero, so cos/sin inherit its representation error (|theta| * 2^-24): compare against the
        # f64 cos/sin of the f32 theta the engine holds, and against the oracle at 1e-6 * max(|theta|, 1).
        th32 = te.get_state()[0].astype(np.float64)
        assert np.abs(obs[0] - np.cos(th32)).max() <= TOL and np.abs(obs[1] - np.sin(th32)).max() <= TOL
        scale = np.maximum(np.abs(ref[0]), 1.0)
        assert (np.abs(obs[0] - oc) / scale).max() <= TOL and (np.abs(obs[1] - os_) / scale).max() <= TOL
    assert bad == 0
    assert mixed_err(te.get_state(), ref).max() <= TOL
    assert mixed_err(got_r, ref_r).max() <= TOL
    # done may legitimately differ only when the f64 state is within 1e-5 of a threshold
    mism = np.nonzero(got_d != ref_d)[0]
    if kind == 0:
        near = (np.abs(np.abs(ref[0]) - 2.4) < 1e-5) | (np.abs(np.abs(ref[2]) - 0.20943951023931953) < 1e-5)
    elif kind == 1:
        near = (np.abs(ref[0] - 0.5) < 1e-5) | (np.abs(ref[1]) < 1e-5)
    else:
        near = np.zeros(n, bool)
    assert near[mism].all(), f"{len(mism)} done mismatches outside the threshold band"


def test_twin_kat_trajectories_match_reference_counts(twin, golden, gymrs):
    """Integer step counts of the KAT trajectories (10 / 9 / 60) in f32."""
    P = gymrs.engine.default_params(0)
    policies = {"always_1": lambda t: 1, "always_0": lambda t: 0, "alternate_1_0": lambda t: (t + 1) % 2}
    for tr in golden("cartpole")["trajectories"]:
        te = TwinEngine(twin, 0, 1, P, flags=0)
        te.reset(0)
        te.set_state(np.array(tr["start"], np.float32).reshape(4, 1))
        t, total = 0, 0.0
        while True:
            te.step([policies[tr["policy"]](t)])
            r, d, _ = te.get_result()
            total += float(r[0])
            t += 1
            if d[0]:
                break
        assert t == tr["steps"] and total == tr["total_reward"]
        assert te.get_state()[:, 0] == pytest.approx(tr["final"], rel=2e-4)  # 60 unstable f32 steps
    Pm = gymrs.engine.default_params(1)
    tr = golden("mountain_car")["trajectories"][0]
    te = TwinEngine(twin, 1, 1, Pm, flags=0)
    te.reset(0)
    te.set_state(np.array(tr["start"], np.float32).reshape(2, 1))
    t = 0
    while True:
        te.step([2 if te.get_state()[1, 0] >= 0 else 0])
        t += 1
        if te.get_result()[1][0]:
            break
    assert t == tr["steps"]


def test_twin_beyond_terminated_and_autoreset(twin, golden, gymrs):
    P = gymrs.engine.default_params(0)
    bt = golden("cartpole")["beyond_terminated"]
    te = TwinEngine(twin, 0, 1, P, flags=0)
    te.reset(0)
    te.set_state(np.array(bt["start"], np.float32).reshape(4, 1))
    rewards, dones = [], []
    for _ in bt["rewards"]:
        te.step([bt["action"]])
        r, d, _ = te.get_result()
        rewards.append(float(r[0]))
        dones.append(bool(d[0]))
    assert rewards == bt["rewards"] and dones == bt["dones"]
    # auto-reset: a finished lane is re-armed inside the step, so the reward never drops to 0
    flags = gymrs.AUTO_RESET | gymrs.TRACK_STATS
    te = TwinEngine(twin, 0, 512, P, flags=flags)
    te.reset(5)
    n_done = 0
    for t in range(100):
        te.step(te.fill_actions(1, t))
        r, d, _ = te.get_result()
        assert (r == 1.0).all()
        st = te.get_state()
        assert (np.abs(st[:, d == 1]) < 0.05 + 1e-9).all()  # fresh states come from the reset box
        n_done += int(d.sum())
    s = te.stats()
    assert s[2] == n_done and s[0] == s[1] and s[3] == 512 * 100


def test_reset_f32_close_to_f64_oracle(twin, oracle, gymrs):
    for kind in (0, 1, 2):
        te = TwinEngine(twin, kind, 4096, gymrs.engine.default_params(kind), gid0=1 << 20)
        te.reset(99)
        ref = oracle.reset_batch(kind, 4096, 1 << 20, 99, 0)
        assert mixed_err(te.get_state(), ref).max() <= TOL


def test_time_limit_truncation_twin(twin, gymrs):
    P = gymrs.engine.default_params(1)
    P.max_episode_steps = 7
    te = TwinEngine(twin, 1, 64, P, flags=gymrs.TIME_LIMIT | gymrs.AUTO_RESET | gymrs.TRACK_STATS)
    te.reset(3)
    for t in range(21):
        te.step(np.ones(64, np.uint8))
        _, d, tr = te.get_result()
        assert not d.any()
        assert tr.all() == ((t + 1) % 7 == 0) and tr.any() == ((t + 1) % 7 == 0)
    s = te.stats()
    assert s[2] == 64 * 3 and s[1] == 64 * 21 and s[0] == -64 * 21


def _f64_flags_of_f32_state(kind, st32, P):
    """The reference's own termination compares (f64) applied to an f32 state: cartpole.rs:450-453, mountain_car.rs:422."""
    s = st32.astype(np.float64)
    with np.errstate(invalid="ignore"):
        if kind == 0:  # strict; a NaN is the maximum in OrderedFloat's order, hence "not <="
            return ~(np.abs(s[0]) <= P.x_threshold) | ~(np.abs(s[2]) <= P.theta_threshold_radians)
        return ~(s[0] < P.goal_position) & ~(s[1] < P.goal_velocity)  # inclusive


@pytest.mark.parametrize("kind,tweak", [(0, {}), (0, {"x_threshold": 1.0 / 3.0, "theta_threshold_radians": 0.1}),
                                        (1, {}), (1, {"goal_position": 0.1, "goal_velocity": 1e-3})])
def test_done_flag_is_the_f64_compare_of_the_f32_state(kind, tweak, twin, gymrs):
    """the kernels keep their thresholds as the largest f32 <= (smallest f32 >=) the f64 value, so that
    for the f32 state they hold the flag IS the reference's f64 compare -- checked with NO tolerance band, on states spread
    over the thresholds and on the f32 neighbours of every threshold."""
    n = 400_000
    rng = np.random.default_rng(50 + kind)
    P = gymrs.engine.default_params(kind)
    for k, v in tweak.items():
        setattr(P, k, v)
    te = TwinEngine(twin, kind, n, P, flags=0)
    te.reset(1)
    if kind == 0:
        xt, tt = P.x_threshold, P.theta_threshold_radians
        st = np.stack([rng.uniform(-1.02 * xt, 1.02 * xt, n), rng.uniform(-0.5, 0.5, n), rng.uniform(-1.02 * tt, 1.02 * tt, n),
                       rng.uniform(-0.5, 0.5, n)]).astype(np.float32)
        # lanes that sit still on the f32 neighbours of +-threshold (zero velocities keep x and theta where they are)
        edge = []
        for thr, row in ((xt, 0), (tt, 2)):
            f = np.float32(thr)
            for v in (f, np.nextafter(f, np.float32(0)), np.nextafter(f, np.float32(9))):
                for sgn in (1, -1):
                    e = np.zeros(4, np.float32)
                    e[row] = sgn * v
                    edge.append(e)
        st[:, : len(edge)] = np.array(edge).T
        act = rng.integers(0, 2, n).astype(np.uint8)
    else:
        gp, gv = P.goal_position, P.goal_velocity
        st = np.stack([rng.uniform(gp - 0.05, gp + 0.05, n), rng.uniform(gv - 0.01, gv + 0.01, n)]).astype(np.float32)
        act = rng.integers(0, 3, n).astype(np.uint8)
    te.set_state(st)
    te.step(act)
    _, done, _ = te.get_result()
    after = te.get_state()
    want = _f64_flags_of_f32_state(kind, after, P)
    assert 0.02 < want.mean() < 0.98  # the sample straddles the thresholds
    assert np.array_equal(done.astype(bool), want), np.nonzero(done.astype(bool) != want)[0][:10]
