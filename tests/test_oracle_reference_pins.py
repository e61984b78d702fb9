"""Pins the oracle (and the product's host-side mirrors) against the reference's OWN unit tests —
the only tests the reference has for this path:
  /root/reference/src/spaces/discrete.rs:22-42        Discrete::contains
  /root/reference/src/utils/custom/util_fns.rs:12-33  clip
  /root/reference/src/utils/seeding.rs:28-40 + doctest :11-20   seed echo
The expected values live in tests/golden/reference_unit_tests.json (data copied from those tests)."""
import math


def test_discrete_contains_like_reference(golden, oracle, gymrs):
    for case in golden("reference_unit_tests")["discrete_contains"]:
        # given_value_greater_or_eq_than_upper_bound... / given_value_less_than_upper_bound...
        assert bool(oracle.lib.orc_discrete_contains(case["n"], case["value"])) is case["expect"]
        assert gymrs.Discrete(case["n"]).contains(case["value"]) is case["expect"]
        assert bool(gymrs.load_library().gymrs_discrete_contains(case["n"], case["value"])) is case["expect"]


def test_clip_like_reference(golden, oracle, twin):
    for case in golden("reference_unit_tests")["clip"]:
        v, l, r, want = case["value"], case["left"], case["right"], case["expect"]
        assert oracle.lib.orc_clip_i64(v, l, r) == want  # the reference tests clip on integers
        assert oracle.lib.orc_clip(float(v), float(l), float(r)) == float(want)
        assert twin.lib.twin_clipf(float(v), float(l), float(r)) == float(want)


def test_clip_total_order_nan(oracle, twin):
    # O64 = OrderedFloat<f64>: NaN is greater than every number, so clip(NaN, l, r) = r
    assert oracle.lib.orc_clip(math.nan, -1.0, 2.0) == 2.0
    assert twin.lib.twin_clipf(math.nan, -1.0, 2.0) == 2.0
    assert oracle.lib.orc_clip(-math.inf, -1.0, 2.0) == -1.0
    assert twin.lib.twin_clipf(math.inf, -1.0, 2.0) == 2.0


def test_seed_echo_like_reference(golden, oracle):
    # given_seed_when_rand_random_then_generator_is_created_using_seed (seeding.rs:33-39)
    for case in golden("reference_unit_tests")["seed_echo"]:
        assert oracle.lib.orc_rand_random_seed(1, case["seed"], 0xABCDEF) == case["expect"]
    assert oracle.lib.orc_rand_random_seed(0, 42, 0xABCDEF) == 0xABCDEF
