// f32_twin.cpp — CPU f32 TWIN of the batched engine.  TEST INFRASTRUCTURE ONLY (same rules as
// gymrs_oracle.h: nothing in the product path may use it).
//
// It compiles the PRODUCT's shared host/device physics header (gym-rs_amd/csrc/gymrs_physics.h) for
// x86 and replays the engine's step()/reset() semantics lane by lane in a plain loop, so the tests
// can demand that the gfx950 kernels produce the SAME BITS (state words, rewards, done / truncated
// flags, integer episode and step counts) — BASELINE.json north_star: "bit-exactly on integer
// step/done counts".  The f64 oracle (gymrs_oracle.c) is the independent check of the physics; the
// twin is the check of everything the kernels add around it (vector tails, auto-reset compaction,
// Philox counters, statistics), which must not change any value.
#include <cstdint>
#include <cstring>
#include <vector>

#include "gymrs_physics.h"

using namespace gymrs;

struct twin_engine {
    int kind;
    uint64_t n, gid0;
    uint32_t flags;
    CartPoleConsts cp;
    MountainCarConsts mc;
    PendulumConsts pd;
    float max_torque;
    float lo[4], hi[4], dlo[4], dhi[4];
    std::vector<float> s[4], obs_cos, obs_sin, reward, ep_ret;
    std::vector<uint8_t> done, truncated, beyond;
    std::vector<uint32_t> ep_start;
    uint64_t seed, tick;
    uint64_t n_invalid;
    double sum_return, n_steps;
    uint64_t sum_length, n_episodes;
};

extern "C" {

twin_engine* twin_create(int kind, uint64_t n, uint64_t gid0, const void* params, uint32_t flags)
{
    twin_engine* e = new twin_engine();
    e->kind = kind;
    e->n = n;
    e->gid0 = gid0;
    e->flags = flags;
    std::memset(e->lo, 0, sizeof(e->lo));
    std::memset(e->hi, 0, sizeof(e->hi));
    std::memset(e->dlo, 0, sizeof(e->dlo));
    std::memset(e->dhi, 0, sizeof(e->dhi));
    e->max_torque = 2.0f;
    if (kind == GYMRS_CARTPOLE) {
        e->cp = make_consts(*static_cast<const gymrs_cartpole_params*>(params));
        for (int j = 0; j < 4; ++j) {
            e->dlo[j] = -0.05f;
            e->dhi[j] = 0.05f;
        }
    } else if (kind == GYMRS_MOUNTAIN_CAR) {
        e->mc = make_consts(*static_cast<const gymrs_mountain_car_params*>(params));
        e->dlo[0] = -0.6f;
        e->dhi[0] = -0.4f;
    } else {
        const auto& p = *static_cast<const gymrs_pendulum_params*>(params);
        e->pd = make_consts(p);
        e->max_torque = (float)p.max_torque;
        e->dlo[0] = -kPiF;
        e->dhi[0] = kPiF;
        e->dlo[1] = -1.0f;
        e->dhi[1] = 1.0f;
    }
    std::memcpy(e->lo, e->dlo, sizeof(e->lo));
    std::memcpy(e->hi, e->dhi, sizeof(e->hi));
    for (auto& v : e->s) v.assign(n, 0.0f);
    e->obs_cos.assign(n, 0.0f);
    e->obs_sin.assign(n, 0.0f);
    e->reward.assign(n, 0.0f);
    e->ep_ret.assign(n, 0.0f);
    e->done.assign(n, 0);
    e->truncated.assign(n, 0);
    e->beyond.assign(n, 0);
    e->ep_start.assign(n, 0);
    e->seed = 0;
    e->tick = 0;
    e->n_invalid = 0;
    e->sum_return = 0;
    e->n_steps = 0;
    e->sum_length = 0;
    e->n_episodes = 0;
    return e;
}

void twin_destroy(twin_engine* e) { delete e; }

// gymrs_set_params: assigning the pub physics fields between two steps changes the constants and nothing else
void twin_set_params(twin_engine* e, const void* params)
{
    if (e->kind == GYMRS_CARTPOLE) {
        e->cp = make_consts(*static_cast<const gymrs_cartpole_params*>(params));
    } else if (e->kind == GYMRS_MOUNTAIN_CAR) {
        e->mc = make_consts(*static_cast<const gymrs_mountain_car_params*>(params));
    } else {
        const auto& p = *static_cast<const gymrs_pendulum_params*>(params);
        e->pd = make_consts(p);
        e->max_torque = (float)p.max_torque;
    }
}

static int state_dim(const twin_engine* e) { return e->kind == GYMRS_CARTPOLE ? 4 : 2; }

static void sample_lane(twin_engine* e, uint64_t i, uint64_t tick)
{
    const u32x4 r = draw4(e->seed, e->gid0 + i, tick, kStreamReset);
    const SampleBox b = make_sample_box(e->lo, e->hi, state_dim(e));
    if (e->kind == GYMRS_CARTPOLE) {
        cartpole_sample(r, b, e->s[0][i], e->s[1][i], e->s[2][i], e->s[3][i]);
    } else if (e->kind == GYMRS_MOUNTAIN_CAR) {
        mountain_car_sample(r, b, e->s[0][i], e->s[1][i]);
    } else {
        pendulum_sample(r, b, e->s[0][i], e->s[1][i]);
    }
}

void twin_reset(twin_engine* e, uint64_t seed, const float* bounds)
{
    std::memcpy(e->lo, e->dlo, sizeof(e->lo));
    std::memcpy(e->hi, e->dhi, sizeof(e->hi));
    if (bounds) {
        const int d = state_dim(e);
        const int sampled = e->kind == GYMRS_MOUNTAIN_CAR ? 1 : d;
        for (int j = 0; j < sampled; ++j) {
            e->lo[j] = bounds[j];
            e->hi[j] = bounds[d + j];
        }
    }
    e->seed = seed;
    e->tick = 0;
    for (uint64_t i = 0; i < e->n; ++i) {
        sample_lane(e, i, e->tick);
        if (e->kind == GYMRS_PENDULUM) sincosf_(e->s[0][i], &e->obs_sin[i], &e->obs_cos[i]);
        e->reward[i] = 0.0f;
        e->done[i] = 0;
        e->truncated[i] = 0;
        e->beyond[i] = 0;
        e->ep_start[i] = (uint32_t)(e->tick + 1);
        e->ep_ret[i] = 0.0f;
    }
    e->tick += 1;
    // a reset discards the open episodes and starts the statistics afresh
    e->sum_return = 0;
    e->n_steps = 0;
    e->sum_length = 0;
    e->n_episodes = 0;
}

void twin_stats_clear(twin_engine* e)
{
    e->sum_return = 0;
    e->n_steps = 0;
    e->sum_length = 0;
    e->n_episodes = 0;
}

// One engine step over host action arrays (u8, or f32 for Pendulum).
void twin_step(twin_engine* e, const void* actions)
{
    const bool AUTO = e->flags & GYMRS_AUTO_RESET;
    const bool STATS = AUTO && (e->flags & GYMRS_TRACK_STATS);
    const bool TLIM = e->flags & GYMRS_TIME_LIMIT;
    const uint32_t tick_next = (uint32_t)(e->tick + 1);
    for (uint64_t i = 0; i < e->n; ++i) {
        float r = 0.0f;
        bool d = false, t = false, ok = true;
        uint32_t max_steps = 0;
        if (e->kind == GYMRS_CARTPOLE) {
            const uint8_t a = static_cast<const uint8_t*>(actions)[i];
            ok = a < 2;
            max_steps = e->cp.max_steps;
            if (ok) {
                d = cartpole_advance(e->cp, e->s[0][i], e->s[1][i], e->s[2][i], e->s[3][i], a);
                r = 1.0f;
                if (!AUTO) {
                    bool b = e->beyond[i] != 0;
                    r = cartpole_reward(d, b);
                    e->beyond[i] = b ? 1 : 0;
                }
            }
        } else if (e->kind == GYMRS_MOUNTAIN_CAR) {
            const uint8_t a = static_cast<const uint8_t*>(actions)[i];
            ok = a < 3;
            max_steps = e->mc.max_steps;
            if (ok) {
                d = mountain_car_advance(e->mc, e->s[0][i], e->s[1][i], a);
                r = -1.0f;
            }
        } else {
            const float a = static_cast<const float*>(actions)[i];
            max_steps = e->pd.max_steps;
            r = pendulum_advance(e->pd, e->s[0][i], e->s[1][i], a);
        }
        if (ok) {
            if (TLIM) t = (tick_next - e->ep_start[i]) >= max_steps;
            if (STATS && e->kind == GYMRS_PENDULUM) e->ep_ret[i] += r;
        } else {
            e->n_invalid += 1;
        }
        e->reward[i] = r;
        e->done[i] = d ? 1 : 0;
        if (TLIM) e->truncated[i] = t ? 1 : 0;
        if (AUTO && (d || t)) {
            sample_lane(e, i, e->tick);
            if (STATS || TLIM) {
                const uint32_t len = tick_next - e->ep_start[i];
                e->ep_start[i] = tick_next;
                if (STATS) {
                    e->sum_length += len;
                    e->n_episodes += 1;
                    if (e->kind == GYMRS_CARTPOLE) e->sum_return += (double)len;
                    else if (e->kind == GYMRS_MOUNTAIN_CAR) e->sum_return -= (double)len;
                    else {
                        e->sum_return += (double)e->ep_ret[i];
                        e->ep_ret[i] = 0.0f;
                    }
                }
            }
        }
        if (e->kind == GYMRS_PENDULUM) sincosf_(e->s[0][i], &e->obs_sin[i], &e->obs_cos[i]);
    }
    e->tick += 1;
    e->n_steps += (double)e->n;
}

void twin_get_state(const twin_engine* e, float* out)
{
    const int d = state_dim(e);
    for (int j = 0; j < d; ++j) std::memcpy(out + (size_t)j * e->n, e->s[j].data(), e->n * sizeof(float));
}

void twin_set_state(twin_engine* e, const float* in)
{
    const int d = state_dim(e);
    for (int j = 0; j < d; ++j) std::memcpy(e->s[j].data(), in + (size_t)j * e->n, e->n * sizeof(float));
    for (uint64_t i = 0; i < e->n; ++i) {
        if (e->kind == GYMRS_PENDULUM) sincosf_(e->s[0][i], &e->obs_sin[i], &e->obs_cos[i]);
    }
}

void twin_get_obs(const twin_engine* e, float* out)
{
    if (e->kind == GYMRS_PENDULUM) {
        std::memcpy(out, e->obs_cos.data(), e->n * sizeof(float));
        std::memcpy(out + e->n, e->obs_sin.data(), e->n * sizeof(float));
        std::memcpy(out + 2 * e->n, e->s[1].data(), e->n * sizeof(float));
    } else {
        twin_get_state(e, out);
    }
}

void twin_get_result(const twin_engine* e, float* reward, uint8_t* done, uint8_t* truncated)
{
    std::memcpy(reward, e->reward.data(), e->n * sizeof(float));
    std::memcpy(done, e->done.data(), e->n);
    std::memcpy(truncated, e->truncated.data(), e->n);
}

void twin_stats(const twin_engine* e, double out[4])
{
    out[0] = e->sum_return;
    out[1] = (double)e->sum_length;
    out[2] = (double)e->n_episodes;
    out[3] = e->n_steps;
}

uint64_t twin_invalid_count(const twin_engine* e) { return e->n_invalid; }

void twin_fill_actions(const twin_engine* e, void* actions, uint64_t seed, uint64_t t)
{
    for (uint64_t i = 0; i < e->n; ++i) {
        if (e->kind == GYMRS_PENDULUM) {
            static_cast<float*>(actions)[i] = uniform_between(action_word(seed, e->gid0 + i, t), -e->max_torque, e->max_torque);
        } else {
            static_cast<uint8_t*>(actions)[i] = action_discrete(seed, e->gid0 + i, t, e->kind == GYMRS_CARTPOLE ? 2u : 3u);
        }
    }
}

// ---- direct access to the shared math for accuracy tests ----
void twin_sincosf(uint64_t n, const float* x, float* s, float* c)
{
    for (uint64_t i = 0; i < n; ++i) sincosf_(x[i], &s[i], &c[i]);
}

void twin_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    const u32x4 r = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    for (int j = 0; j < 4; ++j) out[j] = r.v[j];
}

float twin_uniform_between(uint32_t r, float lo, float hi) { return uniform_between(r, lo, hi); }
float twin_uniform_in_box(uint32_t r, float lo, float hi)
{
    const SampleBox b = make_sample_box(&lo, &hi, 1);
    return uniform_in_box(r, b, 0);
}
float twin_clipf(float v, float l, float r) { return clipf(v, l, r); }
float twin_angle_normalize(float x) { return angle_normalize(x); }

} // extern "C"
