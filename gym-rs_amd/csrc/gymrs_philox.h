// gymrs_philox.h — hand-written Philox4x32-10 (Salmon et al., SC'11) for reset sampling.
//
// Replaces the reference's reset RNG chain
//   seeding::rand_random -> Pcg64::seed_from_u64      /root/reference/src/utils/seeding.rs:21-26
//   Uniform::new(low, high).sample(rng)               cartpole.rs:363, mountain_car.rs:189
// with a counter-based generator (north_star): a lane's draw is a pure function of
// (seed, global env id, tick), so results do not depend on how lanes are sharded over GPUs,
// and there is no per-lane generator state to keep in HBM.  The reference re-creates its PRNG on
// every reset(), so no reference behaviour depends on stream continuity.
//
// Counter layout (same in oracle/gymrs_oracle.c, which restates it independently):
//   key     = (seed lo, seed hi)
//   counter = (gid lo, gid hi, tick lo, (tick hi & 0xffff) | stream << 16)
//   stream 0 = reset sampling, stream 1 = synthetic action generation (bench / tests)
#pragma once
#include "gymrs_math.h"

namespace gymrs {

struct u32x4 {
    uint32_t v[4];
};

constexpr uint32_t kPhiloxM0 = 0xD2511F53u, kPhiloxM1 = 0xCD9E8D57u;
constexpr uint32_t kPhiloxW0 = 0x9E3779B9u, kPhiloxW1 = 0xBB67AE85u;

// a ^ b ^ c: one v_bitop3_b32 on gfx950 (the compiler leaves two v_xor_b32 here on its own)
GYMRS_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

GYMRS_HD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)kPhiloxM0 * c0;
        uint64_t p1 = (uint64_t)kPhiloxM1 * c2;
        uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += kPhiloxW0;
        k1 += kPhiloxW1;
    }
    return u32x4{{c0, c1, c2, c3}};
}

constexpr uint32_t kStreamReset = 0, kStreamAction = 1;

GYMRS_HD u32x4 draw4(uint64_t seed, uint64_t gid, uint64_t tick, uint32_t stream)
{
    return philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)tick,
                         ((uint32_t)(tick >> 32) & 0xffffu) | (stream << 16), (uint32_t)seed,
                         (uint32_t)(seed >> 32));
}

// Random-policy action stream (stream 1; examples/cartpole.rs:19 `rng.gen_range(0..=1)`).  Four neighbouring
// lanes share one Philox block per time slot -- lane gid takes word (gid & 3) of block (gid >> 2, slot) -- so
// a work-item that owns 4 aligned lanes evaluates ONE block for all of them:
//   Box actions (Pendulum):   slot = t,       action = uniform_between(word, low, high)   (24 random bits)
//   Discrete(n) actions:      slot = t >> 1,  the 16-bit half (t & 1) of the word,  action = (half * n) >> 16
// (16 bits: exact for n = 2; for n = 3 the three actions have probabilities 21846, 21845, 21845 / 65536).
GYMRS_HD u32x4 action_block(uint64_t seed, uint64_t gid, uint64_t slot) { return draw4(seed, gid >> 2, slot, kStreamAction); }
GYMRS_HD uint32_t pick_word(const u32x4& b, uint32_t j) { return j == 0 ? b.v[0] : (j == 1 ? b.v[1] : (j == 2 ? b.v[2] : b.v[3])); }
GYMRS_HD uint32_t action_word(uint64_t seed, uint64_t gid, uint64_t t) { return pick_word(action_block(seed, gid, t), (uint32_t)gid & 3u); }
GYMRS_HD uint8_t discrete_from_word(uint32_t word, uint64_t t, uint32_t n_actions)
{
    const uint32_t half = (word >> (((uint32_t)t & 1u) * 16u)) & 0xffffu;
    return (uint8_t)((half * n_actions) >> 16);
}
GYMRS_HD uint8_t action_discrete(uint64_t seed, uint64_t gid, uint64_t t, uint32_t n_actions)
{
    return discrete_from_word(action_word(seed, gid, t >> 1), t, n_actions);
}

// u32 -> uniform f32 on [low, high): 24 random bits, u * scale + low, kept below `high` (half-open like
// rand's Uniform::new, cartpole.rs:363).  Everything that does not depend on the random word is prepared
// once on the host: v = min(fma(float(r >> 8), scale24, low), high_prev), scale24 = (high - low) * 2^-24,
// high_prev = the largest float below high.  Three instructions per draw on the GPU.
struct SampleBox {
    float lo[4], scale24[4], hi_prev[4];
};

inline SampleBox make_sample_box(const float* lo, const float* hi, int dims)
{
    SampleBox b;
    for (int j = 0; j < 4; ++j) {
        b.lo[j] = 0.0f;
        b.scale24[j] = 0.0f;
        b.hi_prev[j] = 0.0f;
    }
    for (int j = 0; j < dims; ++j) {
        b.lo[j] = lo[j];
        b.scale24[j] = (hi[j] - lo[j]) * 0x1p-24f;
        uint32_t h = f2u(hi[j]);
        h = (hi[j] > 0.0f) ? h - 1u : ((h & 0x7fffffffu) == 0u ? 0x80000001u : h + 1u);
        b.hi_prev[j] = u2f(h);
    }
    return b;
}

GYMRS_HD float uniform_in_box(uint32_t r, const SampleBox& b, int j)
{
    const float v = fmaf_((float)(r >> 8), b.scale24[j], b.lo[j]);
    return v < b.hi_prev[j] ? v : b.hi_prev[j]; // v is never NaN: both operands of the min are finite
}

GYMRS_HD float uniform_between(uint32_t r, float low, float high)
{
    SampleBox b;
    b.lo[0] = low;
    b.scale24[0] = (high - low) * 0x1p-24f;
    uint32_t h = f2u(high);
    h = (high > 0.0f) ? h - 1u : ((h & 0x7fffffffu) == 0u ? 0x80000001u : h + 1u);
    b.hi_prev[0] = u2f(h);
    return uniform_in_box(r, b, 0);
}

} // namespace gymrs
