// gymrs_pcg64.h -- the reference's OWN reset stream, as an optional mode of Env::reset (gymrs_reset_pcg64).
//
// `reset(Some(seed), _, options)` of the reference (cartpole.rs:485-516, mountain_car.rs:464-501) builds
// `rand_pcg::Pcg64::seed_from_u64(seed)` (seeding.rs:21-26) and draws the state with `rand::distributions::Uniform`
// over `UniformOrdered<f64>` (cartpole.rs:293-297,317-324; mountain_car.rs:145,162-167).  The crates are third-party
// and absent from the reference tree (rand 0.8, rand_pcg 0.3, rand_core 0.6, ordered-float >= 3.9.1; Cargo.toml:24-33,
// SURVEY App. B.2), so their published algorithms are restated here:
//   * rand_core 0.6 `SeedableRng::seed_from_u64`: the 32 seed bytes come 4 at a time from a PCG32 (XSH-RR 64/32) whose
//     state starts at the seed number;
//   * rand_pcg 0.3 `Lcg128Xsl64::from_seed`: state = bytes 0..15, increment = bytes 16..31 | 1 (little endian),
//     `from_state_incr`: state += increment, one LCG step; `next_u64`: one LCG step, then XSL-RR 128/64;
//   * rand 0.8 `UniformFloat<f64>`: 52 random bits as the mantissa of a double in [1, 2), minus 1, times `scale`, plus
//     `low` (two roundings, no fma), with `scale = high - low` shrunk ulp by ulp at construction until the largest
//     draw stays below `high`.
// Pinned by rand_pcg's own known answers (tests/golden/pcg64.json: `new(42, 54)`, `from_seed([1..=32])`,
// `seed_from_u64(0)`) through the CPU oracle's independent restatement (oracle/gymrs_oracle.c), which the GPU must match
// bit for bit after the one f64 -> f32 rounding of the state.  One generator per lane, 128-bit arithmetic on 64-bit
// halves: a lane's reset is ~12 wide multiplies, off the per-step path.
#pragma once
#include <cstdint>
#include "gymrs_philox.h" // GYMRS_HD

namespace gymrs {

struct U128 {
    uint64_t lo, hi;
};

GYMRS_HD uint64_t mul_hi_u64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

struct Pcg64Lane {
    U128 state, incr;

    // state = state * 0x2360ED051FC65DA4_4385DF649FCCF645 + incr  (mod 2^128)
    GYMRS_HD void advance()
    {
        constexpr uint64_t kMulHi = 0x2360ED051FC65DA4ull, kMulLo = 0x4385DF649FCCF645ull;
        const uint64_t lo = state.lo * kMulLo;
        const uint64_t hi = mul_hi_u64(state.lo, kMulLo) + state.lo * kMulHi + state.hi * kMulLo;
        const uint64_t sum_lo = lo + incr.lo;
        state.hi = hi + incr.hi + (sum_lo < lo ? 1u : 0u);
        state.lo = sum_lo;
    }

    GYMRS_HD uint64_t next_u64()
    {
        advance();
        const uint32_t rot = (uint32_t)(state.hi >> 58); // state >> 122
        const uint64_t x = state.hi ^ state.lo;
        return (x >> rot) | (x << ((64u - rot) & 63u));
    }

    // Pcg64::seed_from_u64 (what seeding.rs:23 calls).
    GYMRS_HD static Pcg64Lane seed_from_u64(uint64_t s)
    {
        uint64_t w[4];
        for (int i = 0; i < 4; ++i) {
            uint64_t word = 0;
            for (int half = 0; half < 2; ++half) {
                s = s * 6364136223846793005ull + 11634580027462260723ull;
                const uint32_t xs = (uint32_t)(((s >> 18) ^ s) >> 27);
                const uint32_t rot = (uint32_t)(s >> 59);
                const uint32_t out = (xs >> rot) | (xs << ((32u - rot) & 31u));
                word |= (uint64_t)out << (32 * half);
            }
            w[i] = word;
        }
        Pcg64Lane g;
        g.incr = U128{w[2] | 1ull, w[3]};
        const uint64_t lo = w[0] + g.incr.lo;
        g.state = U128{lo, w[1] + g.incr.hi + (lo < w[0] ? 1u : 0u)};
        g.advance();
        return g;
    }
};

// One UniformFloat<f64> draw: the f64 the reference would store, before this build rounds it to f32.
GYMRS_HD double pcg64_uniform(Pcg64Lane& g, double low, double scale)
{
    const uint64_t bits = (g.next_u64() >> 12) | 0x3FF0000000000000ull;
    double one_two;
    __builtin_memcpy(&one_two, &bits, sizeof(one_two));
    const double zero_one = one_two - 1.0;
#if defined(__HIP_DEVICE_COMPILE__)
    return __dadd_rn(__dmul_rn(zero_one, scale), low); // the reference rounds the product, then the sum
#else
    volatile double product = zero_one * scale;
    return product + low;
#endif
}

} // namespace gymrs
