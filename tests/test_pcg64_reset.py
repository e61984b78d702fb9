"""The oracle's restatement of the reference's own reset stream (seeding.rs:21-26 -> rand_pcg::Pcg64::seed_from_u64,
rand::distributions::Uniform over f64; cartpole.rs:293-297,317-324,352-364; mountain_car.rs:145,162-190).

The crates are third-party and not on this box; what pins the restatement:
  * rand_pcg's own known answers (tests/golden/pcg64.json) -- they cover seed_from_u64, from_seed, new and next_u64;
  * numpy.random.PCG64 (same 128-bit LCG multiplier, same XSL-RR output) on arbitrary states;
  * a second, pure-Python restatement written from the same published algorithms (arbitrary-precision integers, no
    shared code with the C oracle);
  * tests/golden/from_reference/reset_kat.json once bindings/rust/src/bin/make_golden.rs has run (skipped here: no
    cargo in this image) -- the only thing that pins the Uniform stage by the reference itself.
"""
import json
import math
import struct
from pathlib import Path

import numpy as np
import pytest

GOLDEN = Path(__file__).resolve().parent / "golden"
M64, M128 = (1 << 64) - 1, (1 << 128) - 1
PCG_MUL = 0x2360ED051FC65DA44385DF649FCCF645


class PyPcg64:
    """Pure-Python Lcg128Xsl64."""

    def __init__(self, state, incr):
        self.inc = incr & M128
        self.state = (state + self.inc) & M128
        self._step()

    def _step(self):
        self.state = (self.state * PCG_MUL + self.inc) & M128

    def next_u64(self):
        self._step()
        rot = self.state >> 122
        x = ((self.state >> 64) ^ self.state) & M64
        return ((x >> rot) | (x << ((64 - rot) & 63))) & M64

    @classmethod
    def seed_from_u64(cls, s):
        raw = b""
        for _ in range(8):
            s = (s * 6364136223846793005 + 11634580027462260723) & M64
            xs = (((s >> 18) ^ s) >> 27) & 0xFFFFFFFF
            rot = s >> 59
            raw += (((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xFFFFFFFF).to_bytes(4, "little")
        w = [int.from_bytes(raw[8 * i: 8 * i + 8], "little") for i in range(4)]
        return cls(w[0] | (w[1] << 64), (w[2] | (w[3] << 64)) | 1)


def py_uniform_scale(low, high):
    assert (high - low) >= 1e-3 * max(abs(low), abs(high))  # see test_uniform_constructor
    scale = high - low
    max_rand = 1.0 - 2.0 ** -52
    while scale * max_rand + low >= high:
        scale = struct.unpack("<d", struct.pack("<Q", struct.unpack("<Q", struct.pack("<d", scale))[0] - 1))[0]
    return scale


def py_uniform(g, low, scale):
    v12 = struct.unpack("<d", struct.pack("<Q", (g.next_u64() >> 12) | 0x3FF0000000000000))[0]
    return (v12 - 1.0) * scale + low  # python floats: the product and the sum round separately


def test_known_answers_of_rand_pcg(oracle):
    kat = json.loads((GOLDEN / "pcg64.json").read_text())
    g = oracle.pcg64(new=(kat["new_42_54"]["state"], kat["new_42_54"]["stream"]))
    assert [oracle.pcg64_next(g) for _ in range(6)] == [int(v, 16) for v in kat["new_42_54"]["next_u64"]]
    g = oracle.pcg64(from_seed=bytes(kat["from_seed_1_to_32"]["seed_bytes"]))
    assert oracle.pcg64_next(g) == kat["from_seed_1_to_32"]["first_next_u64"]
    g = oracle.pcg64(seed_from_u64=kat["seed_from_u64_0"]["seed"])
    assert oracle.pcg64_next(g) == kat["seed_from_u64_0"]["first_next_u64"]
    # the pure-Python restatement agrees with the same vectors (it is the second witness below)
    assert PyPcg64.seed_from_u64(0).next_u64() == kat["seed_from_u64_0"]["first_next_u64"]


def test_lcg_and_output_stage_against_numpy_pcg64(oracle):
    """numpy's PCG64 is the same generator family: load the oracle's state/increment into it and compare raw output."""
    rs = np.random.default_rng(5)
    for seed in [0, 1, 42, 2**63, M64] + [int(v) for v in rs.integers(0, 2**63, 20)]:
        g = oracle.pcg64(seed_from_u64=seed)
        bg = np.random.PCG64()
        bg.state = {"bit_generator": "PCG64", "has_uint32": 0, "uinteger": 0,
                    "state": {"state": g.state_lo | (g.state_hi << 64), "inc": g.incr_lo | (g.incr_hi << 64)}}
        want = [int(v) for v in bg.random_raw(16)]
        assert [oracle.pcg64_next(g) for _ in range(16)] == want
        p = PyPcg64.seed_from_u64(seed)
        assert [p.next_u64() for _ in range(16)] == want


def test_uniform_constructor(oracle):
    assert oracle.uniform_f64_scale(-0.05, 0.05) == 0.1 and oracle.uniform_f64_scale(-0.6, -0.4) == -0.4 - -0.6
    # where Uniform::new panics (cartpole.rs:363 would abort): low >= high, non-finite bounds, overflowing range
    for low, high in [(0.0, 0.0), (1.0, -1.0), (0.0, math.inf), (-math.inf, 0.0), (math.nan, 1.0), (-1.7e308, 1.7e308)]:
        assert oracle.uniform_f64_scale(low, high) is None
    # the scale is shrunk until the largest draw stays below high
    rs = np.random.default_rng(11)
    shrunk = 0
    for _ in range(3000):
        low = float(rs.normal()) * 10.0 ** int(rs.integers(-8, 8))
        high = low + abs(float(rs.normal())) * 10.0 ** int(rs.integers(-12, 6))
        if not low < high or (high - low) < 1e-3 * max(abs(low), abs(high)):
            continue  # the constructor's loop runs ~ulp(high) / ulp(high - low) times: narrow boxes far from 0 take forever
        scale = oracle.uniform_f64_scale(low, high)
        assert scale == py_uniform_scale(low, high)
        assert scale * (1.0 - 2.0 ** -52) + low < high
        shrunk += scale != high - low
    assert shrunk > 0  # the loop body is exercised
    assert oracle.uniform_f64_scale(1e5, 1e5 + 1e-10) is None  # 1e15 rounds in the reference: refused after 2^22


def test_reset_states_against_the_python_restatement(oracle):
    rs = np.random.default_rng(3)
    for seed in [0, 1, 42, 2024, M64] + [int(v) for v in rs.integers(0, 2**63, 200)]:
        g = PyPcg64.seed_from_u64(seed)
        want = [py_uniform(g, -0.05, 0.1) for _ in range(4)]
        assert oracle.reset_pcg64(0, seed) == want
        assert all(-0.05 <= v < 0.05 for v in want)
        g = PyPcg64.seed_from_u64(seed)
        assert oracle.reset_pcg64(1, seed) == [py_uniform(g, -0.6, -0.4 - -0.6), 0.0]
    # `options`: four samplers built first, then four draws in field order (cartpole.rs:293-297,317-324)
    b = [-1.0, 0.5, -1e-3, 100.0, 2.0, 0.75, 1e-3, 100.5]
    g = PyPcg64.seed_from_u64(7)
    assert oracle.reset_pcg64(0, 7, b) == [py_uniform(g, b[j], py_uniform_scale(b[j], b[4 + j])) for j in range(4)]
    assert oracle.reset_pcg64(0, 7, [0, 0, 0, 0, 1, 1, 0, 1]) is None  # theta: low == high -> the reference panics
    # MountainCar ignores the velocity bounds (mountain_car.rs:145)
    g = PyPcg64.seed_from_u64(9)
    assert oracle.reset_pcg64(1, 9, [-1.0, 5.0, 0.25, -5.0]) == [py_uniform(g, -1.0, 1.25), 0.0]
    # the same seed always gives the same state, different seeds differ
    assert oracle.reset_pcg64(0, 5) == oracle.reset_pcg64(0, 5) != oracle.reset_pcg64(0, 6)


def test_reset_distribution(oracle):
    xs = np.array([oracle.reset_pcg64(0, s) for s in range(4000)])
    assert xs.min() >= -0.05 and xs.max() < 0.05
    assert abs(xs.mean()) < 2e-3 and abs(xs.std() - 0.1 / math.sqrt(12)) < 1e-3
    assert abs(np.corrcoef(xs.T)[0, 1]) < 0.06


def test_reset_states_written_by_the_reference(oracle):
    """tests/golden/from_reference/reset_kat.json (bindings/rust/src/bin/make_golden.rs: the reference's own
    `reset(Some(seed), false, None)`).  Turns "restated from the crates' publications" into "pinned by the reference"."""
    f = GOLDEN / "from_reference" / "reset_kat.json"
    if not f.exists():
        pytest.skip("tests/golden/from_reference/reset_kat.json not generated: this image has no cargo")
    kat = json.loads(f.read_text())
    for case in kat["cartpole"]:
        assert oracle.reset_pcg64(0, case["seed"]) == case["state"], case
    for case in kat["mountain_car"]:
        assert oracle.reset_pcg64(1, case["seed"]) == case["state"], case
