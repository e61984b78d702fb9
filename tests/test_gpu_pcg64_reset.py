"""gymrs_reset_pcg64 on the GPU: every lane re-armed by the reference's own generator chain (Pcg64::seed_from_u64 +
Uniform over f64, seeding.rs:21-26, cartpole.rs:485-516, mountain_car.rs:464-501) must hold, bit for bit, the oracle's
f64 state rounded once to f32.  The oracle's chain is pinned in tests/test_pcg64_reset.py."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle.bindings import TwinEngine

pytestmark = pytest.mark.gpu
M64 = (1 << 64) - 1


def want_states(oracle, kind, seeds, bounds=None):
    return np.array([oracle.reset_pcg64(kind, int(s), bounds) for s in seeds], dtype=np.float64).astype(np.float32).T


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("seed,gid0", [(0, 0), (42, 1000), (M64 - 700, 500), (2**63 + 12345, 2**40 + 3)])
def test_lane_i_holds_the_references_reset_of_seed_plus_global_id(gymrs, oracle, kind, seed, gid0):
    n = 3001
    with gymrs.BatchedEngine(kind, n, flags=0, global_env_offset=gid0) as eng:
        assert eng.reset_pcg64(seed) == seed  # the seed echo of rand_random (seeding.rs:21-26)
        got = eng.get_state()
        seeds = [(seed + gid0 + i) & M64 for i in range(n)]  # wraps around 2^64 like u64 arithmetic
        assert np.array_equal(got.view(np.uint32), want_states(oracle, kind, seeds).view(np.uint32))
        if kind == 1:
            assert not got[1].any()  # velocity: OrderedFloat(0.), mountain_car.rs:162-167
        # the same seed gives the same states every time; another seed does not
        eng.reset_pcg64(seed)
        assert np.array_equal(eng.get_state(), got)
        eng.reset_pcg64(seed + 1)
        assert np.array_equal(eng.get_state()[0, :-1], got[0, 1:])  # lane i of seed + 1 = lane i + 1 of seed


@pytest.mark.parametrize("kind", [0, 1])
def test_explicit_per_lane_seeds_and_options(gymrs, oracle, kind):
    n = 2000
    rs = np.random.default_rng(kind)
    seeds = rs.integers(0, 2**64, n, dtype=np.uint64)
    seeds[:4] = [0, 1, M64, 2**63]
    dev = torch.from_numpy(seeds.view(np.int64)).to("cuda:0")
    d = 4 if kind == 0 else 2
    bounds = [-1.0, 0.5, -1e-3, 100.0, 2.0, 0.75, 1e-3, 100.5] if kind == 0 else [-1.0, 5.0, 0.25, -5.0]
    with gymrs.BatchedEngine(kind, n, flags=0, global_env_offset=77) as eng:
        eng.reset_pcg64(5, seeds_dev=dev.data_ptr())
        assert np.array_equal(eng.get_state().view(np.uint32), want_states(oracle, kind, seeds).view(np.uint32))
        eng.reset_pcg64(5, seeds_dev=dev.data_ptr(), options=bounds)
        got = eng.get_state()
        assert np.array_equal(got.view(np.uint32), want_states(oracle, kind, seeds, bounds).view(np.uint32))
        for j in range(d if kind == 0 else 1):
            assert got[j].min() >= np.float32(bounds[j]) and got[j].max() <= np.float32(bounds[d + j])
        # where the reference's Uniform::new panics (or would spin for 1e15 rounds) the call is refused, nothing changes
        bad = list(bounds)
        bad[d] = bad[0]
        for b in (bad, [1e5, 0, 0, 0, 1e5 + 1e-10, 1, 1, 1][: 2 * d] if kind == 0 else [1e5, 0, 1e5 + 1e-10, 0]):
            with pytest.raises(gymrs.GymrsError):
                eng.reset_pcg64(5, options=b)
        assert np.array_equal(eng.get_state(), got)
    with gymrs.BatchedEngine(gymrs.PENDULUM, 8, flags=0) as pend:
        with pytest.raises(gymrs.GymrsError, match="no Pendulum"):
            pend.reset_pcg64(1)


def test_everything_else_is_as_after_a_philox_reset(gymrs, twin):
    """Flags cleared, statistics restarted, tick = 1, and the later GYMRS_AUTO_RESET re-arms come from the Philox stream
    keyed by the same seed: the twin, given the PCG64 start states, stays bit-identical through 60 steps."""
    n, seed, flags = 5000, 99, 1 | 2 | 4
    p = gymrs.engine.default_params(0)
    p.max_episode_steps = 30
    eng = gymrs.BatchedEngine(0, n, flags=flags, params=p, global_env_offset=10)
    tw = TwinEngine(twin, 0, n, p, flags=flags, gid0=10)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    for t in range(13):  # leave episodes open, reset-log rows pending, statistics non-zero
        eng.fill_actions(buf.data_ptr(), seed=1, t=t)
        eng.step(buf.data_ptr())
    eng.reset_pcg64(seed)
    assert eng.tick() == (1, seed) and not eng.stats().any()
    _, done, trunc = eng.get_step_result()
    assert not done.any() and not trunc.any()
    tw.reset(seed)
    tw.set_state(eng.get_state())
    for t in range(60):
        eng.fill_actions(buf.data_ptr(), seed=1, t=t)
        eng.step(buf.data_ptr())
        tw.step(tw.fill_actions(1, t))
    assert np.array_equal(eng.get_state().view(np.uint32), tw.get_state().view(np.uint32))
    assert np.array_equal(eng.stats(), tw.stats())
    eng.close()


def test_single_env_mirrors_return_the_references_reset_state(gymrs, oracle):
    env = gymrs.CartPoleEnv(reset_rng="pcg64")
    for seed in (0, 42, M64):
        obs, info = env.reset(seed=seed, return_info=True)
        want = [float(np.float32(v)) for v in oracle.reset_pcg64(0, seed)]
        assert obs.to_vec() == want and info == () and env.state == obs and env.steps_beyond_terminated is None
    lo = gymrs.CartPoleObservation(-1.0, 0.5, -1e-3, 100.0)
    hi = gymrs.CartPoleObservation(2.0, 0.75, 1e-3, 100.5)
    obs, _ = env.reset(seed=7, options=gymrs.BoxR(lo, hi))
    assert obs.to_vec() == [float(np.float32(v)) for v in oracle.reset_pcg64(0, 7, lo.to_vec() + hi.to_vec())]
    used = env.reset()  # seed = None: OS entropy, still inside the default box
    assert all(-0.05 <= v <= 0.05 for v in used[0].to_vec())
    r = env.step(1)  # and the env steps on from there
    assert not r.done
    env.release()
    car = gymrs.MountainCarEnv(reset_rng="pcg64")
    obs, _ = car.reset(seed=2024)
    assert obs.to_vec() == [float(np.float32(v)) for v in oracle.reset_pcg64(1, 2024)]
    car.release()
    with pytest.raises(ValueError):
        gymrs.PendulumEnv(reset_rng="pcg64")
    # the default stays the Philox stream (north_star): a different state for the same seed
    plain = gymrs.CartPoleEnv()
    assert plain.reset(seed=42)[0].to_vec() != [float(np.float32(v)) for v in oracle.reset_pcg64(0, 42)]
    plain.release()


def test_c_abi_argument_checks(gymrs):
    lib = gymrs.load_library()
    assert lib.gymrs_reset_pcg64(None, 1, 0, None, None, None) == 1
    with gymrs.BatchedEngine(0, 4, flags=0) as eng:
        used = C.c_uint64()
        assert lib.gymrs_reset_pcg64(eng._h, 0, 0, None, None, C.byref(used)) == 0  # OS entropy
        a = eng.get_state().copy()
        assert lib.gymrs_reset_pcg64(eng._h, 1, used.value, None, None, None) == 0
        assert np.array_equal(eng.get_state(), a)  # the echoed seed reproduces the draw


def test_full_size_pcg64_reset(gymrs, oracle):
    """2^20 lanes (BASELINE configs[1]): the shift property lane i of seed s + 1 = lane i + 1 of seed s over the whole batch,
    2000 sampled lanes against the oracle, range and moments of the draw."""
    n, seed, gid0 = 1 << 20, 123456789, 1 << 35
    with gymrs.BatchedEngine(0, n, flags=0, global_env_offset=gid0) as eng:
        eng.reset_pcg64(seed)
        a = eng.get_state().copy()
        eng.reset_pcg64(seed + 1)
        b = eng.get_state()
        assert np.array_equal(a[:, 1:].view(np.uint32), b[:, :-1].view(np.uint32))
        idx = np.random.default_rng(0).integers(0, n, 2000)
        want = want_states(oracle, 0, [(seed + gid0 + int(i)) & M64 for i in idx])
        assert np.array_equal(a[:, idx].view(np.uint32), want.view(np.uint32))
        assert a.min() >= np.float32(-0.05) and a.max() <= np.float32(0.05)
        assert abs(float(a.mean())) < 2e-4 and abs(float(a.std()) - 0.1 / np.sqrt(12)) < 2e-4
        assert abs(np.corrcoef(a)[0, 1]) < 5e-3  # x and x_dot of one lane come from consecutive outputs of one generator


@pytest.mark.parametrize("n", [1, 7, 64, 65, 1000])
def test_pcg64_reset_ragged_sizes(gymrs, oracle, n):
    """Lane counts that are no multiple of anything, incl. engines whose arrays live in mapped host memory (n <= 64)."""
    for kind in (0, 1):
        with gymrs.BatchedEngine(kind, n, flags=0, global_env_offset=3) as eng:
            eng.reset_pcg64(11)
            want = want_states(oracle, kind, [11 + 3 + i for i in range(n)])
            assert np.array_equal(eng.get_state().view(np.uint32), want.view(np.uint32))
