// gymrs_env.hpp — header-only C++ mirror of the reference's trait surface over the C ABI (gymrs_amd.h).
//
// The reference is a Rust crate; this image has no Rust toolchain, so the host side above the C ABI is
// C++ (the reference is compiled code).  Names, argument meaning and error behaviour follow
//   trait Env / EnvProperties           /root/reference/src/core.rs:25-90
//   ActionReward, RewardRange           /root/reference/src/core.rs:94-122
//   Discrete, BoxR                      /root/reference/src/spaces/discrete.rs:12-20, box_r.rs:5-13
//   CartPoleEnv, CartPoleObservation    /root/reference/src/envs/classical_control/cartpole.rs:51-87,328-349
//   MountainCarEnv, ...Observation      /root/reference/src/envs/classical_control/mountain_car.rs:46-84,122-128
// so a test written against gym-rs reads the same here.  A single env is ONE lane of the batched GPU
// engine (plumbing configuration); `VecEnv` is the batched form the hot path is built for.
// Errors: the reference panics (assert!/unwrap); here they are C++ exceptions (gymrs::Panic).
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <array>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "gymrs_amd.h"

namespace gymrs {

struct Panic : std::runtime_error {
    gymrs_status status;
    Panic(gymrs_status s, const std::string& m) : std::runtime_error(m), status(s) {}
};
inline void check(gymrs_status s)
{
    if (s != GYMRS_OK) throw Panic(s, gymrs_last_error());
}

enum class RenderMode { Human, SingleRgbArray, RgbArray, Ansi, None }; // utils/renderer.rs:83-114 (only None is supported)

struct Discrete { // spaces/discrete.rs:12-20
    std::size_t n;
    bool contains(std::size_t value) const { return value < n; }
    bool operator==(const Discrete& o) const { return n == o.n; }
};
template <class T>
struct BoxR { // spaces/box_r.rs:5-13
    T low, high;
};
struct RewardRange { // core.rs:109-122, default (-inf, inf) core.rs:16-19
    double lower_bound = -std::numeric_limits<double>::infinity();
    double upper_bound = std::numeric_limits<double>::infinity();
};
template <class Obs, class Info>
struct ActionReward { // core.rs:94-106
    Obs observation;
    double reward;
    bool done;
    bool truncated;
    std::optional<Info> info;
};
struct Unit {}; // Rust's ()

// Which generator reset() draws the start state from: the build's counter-based Philox stream (default), or the
// reference's own Pcg64::seed_from_u64 + Uniform chain, so that reset(Some(s)) returns the reference's state for s.
enum class ResetRng { Philox, Pcg64 };

struct CartPoleObservation { // cartpole.rs:328-334; Into<Vec<f64>> order :336-349
    double x, x_dot, theta, theta_dot;
    std::vector<double> to_vec() const { return {x, x_dot, theta, theta_dot}; }
    CartPoleObservation operator-() const { return {-x, -x_dot, -theta, -theta_dot}; } // :367-378
};
struct MountainCarObservation { // mountain_car.rs:122-128
    double position, velocity;
    std::vector<double> to_vec() const { return {position, velocity}; }
};

// ---- batched form: the shape the hot path is built for ---------------------------------------------
class VecEnv {
public:
    VecEnv(gymrs_env_kind kind, std::uint64_t n_envs, std::uint32_t flags = 0, const void* params = nullptr,
           std::uint64_t global_env_offset = 0, int device = 0)
        : kind_(kind), n_(n_envs)
    {
        check(gymrs_engine_create(kind, n_envs, global_env_offset, device, params, flags, &e_));
        int d = 0;
        float* p[4];
        check(gymrs_state_ptrs(e_, p, &d));
        state_dim_ = d;
    }
    ~VecEnv() { gymrs_engine_destroy(e_); }
    // `Env: Clone` (core.rs:25): a deep copy on the device, including the RNG position (seed, tick)
    VecEnv(const VecEnv& other) : kind_(other.kind_), n_(other.n_), state_dim_(other.state_dim_)
    {
        check(gymrs_engine_clone(other.e_, &e_));
    }
    VecEnv& operator=(const VecEnv&) = delete;

    std::uint64_t reset(std::optional<std::uint64_t> seed, const float* bounds_low_high = nullptr)
    {
        std::uint64_t used = 0;
        check(gymrs_reset(e_, seed.has_value(), seed.value_or(0), bounds_low_high, &used));
        return used;
    }
    // The same reset drawn from the reference's own generator chain (Pcg64::seed_from_u64 + Uniform over f64,
    // seeding.rs:21-26): lane i = the reference's reset(Some(seed + offset + i)) rounded to f32 (gymrs_reset_pcg64).
    std::uint64_t reset_pcg64(std::optional<std::uint64_t> seed, const double* bounds_low_high = nullptr,
                              const std::uint64_t* seeds_dev = nullptr)
    {
        std::uint64_t used = 0;
        check(gymrs_reset_pcg64(e_, seed.has_value(), seed.value_or(0), seeds_dev, bounds_low_high, &used));
        return used;
    }
    void step_device(const void* actions_dev) { check(gymrs_step(e_, actions_dev)); } // async
    void step_host(const void* actions_host)
    {
        check(gymrs_step_host(e_, actions_host));
        check(gymrs_sync(e_));
    }
    // n_steps consecutive steps, step t with the actions of buffer t % n_buffers of a ring of device buffers stride_bytes apart; one
    // asynchronous call (8 steps and more: one chain through the engine's own HSA queue)
    void step_many(const void* actions_dev, std::uint64_t stride_bytes, std::uint32_t n_buffers, std::uint32_t n_steps, bool use_graph = false)
    {
        check(gymrs_step_many(e_, actions_dev, stride_bytes, n_buffers, n_steps, use_graph ? 1 : 0));
    }
    void sync() { check(gymrs_sync(e_)); }
    // the loop of examples/cartpole.rs:15-30 (random action, step, reset on done) for every lane, fused into one launch
    void rollout(std::uint32_t n_steps, std::uint64_t action_seed, std::uint64_t action_t0 = 0)
    {
        check(gymrs_rollout(e_, n_steps, action_seed, action_t0));
    }
    // the same loop keeping every step's (observation, action, reward, done) in device buffers
    void rollout_record(std::uint32_t n_steps, std::uint64_t action_seed, std::uint64_t action_t0, const gymrs_trajectory& out)
    {
        check(gymrs_rollout_record(e_, n_steps, action_seed, action_t0, &out));
    }
    std::array<double, 4> stats() // {sum_return, sum_length, n_episodes, n_steps}
    {
        std::array<double, 4> out{};
        check(gymrs_stats(e_, out.data()));
        return out;
    }
    // `Env: Serialize` (core.rs:25): everything a step can observe, as an opaque blob
    std::vector<unsigned char> snapshot()
    {
        std::uint64_t bytes = 0;
        check(gymrs_snapshot_size(e_, &bytes));
        std::vector<unsigned char> blob(bytes);
        check(gymrs_snapshot_save(e_, blob.data(), bytes));
        return blob;
    }
    void restore(const std::vector<unsigned char>& blob) { check(gymrs_snapshot_load(e_, blob.data(), blob.size())); }
    std::vector<float> state(std::uint64_t first, std::uint64_t count)
    {
        std::vector<float> out(count * state_dim_);
        check(gymrs_get_state(e_, first, count, out.data()));
        return out;
    }
    void set_state(std::uint64_t first, std::uint64_t count, const float* soa) { check(gymrs_set_state(e_, first, count, soa)); }
    void result(std::uint64_t first, std::uint64_t count, float* reward, std::uint8_t* done, std::uint8_t* truncated)
    {
        check(gymrs_get_step_result(e_, first, count, reward, done, truncated));
    }
    // the pub physics fields after construction (cartpole.rs:53-82): only the constants change, the episode carries on
    void set_params(const void* params) { check(gymrs_set_params(e_, params)); }
    void get_params(void* params_out) { check(gymrs_get_params(e_, params_out)); }
    // `#[derive(Serialize)]` view of the reference env lane `lane` stands for (core.rs:25)
    std::string to_json(std::uint64_t lane = 0)
    {
        std::uint64_t need = 0;
        std::string buf(2048, '\0');
        gymrs_status st = gymrs_env_json(e_, lane, buf.data(), buf.size(), &need);
        if (st != GYMRS_OK && need > buf.size()) {
            buf.assign(need, '\0');
            st = gymrs_env_json(e_, lane, buf.data(), buf.size(), &need);
        }
        check(st);
        buf.resize(need ? need - 1 : 0);
        return buf;
    }
    gymrs_engine* handle() { return e_; }
    std::uint64_t size() const { return n_; }
    int state_dim() const { return state_dim_; }

private:
    gymrs_env_kind kind_;
    std::uint64_t n_;
    gymrs_engine* e_ = nullptr;
    int state_dim_ = 0;
};

// ---- one batch over several GPUs, in one process (gymrs_sharded_*: one engine + one native host thread per block) ---------
// contiguous blocks of lanes, global lane ids, no exchange except the four statistics doubles (RCCL over xGMI on
// distinct devices, a host-side sum where blocks share a device).  Every result is bit-identical to ONE VecEnv of n_total lanes.
class ShardedVecEnv {
public:
    ShardedVecEnv(gymrs_env_kind kind, std::uint64_t n_total, const std::vector<int>& devices, std::uint32_t flags = 0, const void* params = nullptr,
                  std::uint64_t global_env_offset = 0)
        : n_(n_total), state_dim_(kind == GYMRS_CARTPOLE ? 4 : 2)
    {
        check(gymrs_sharded_create(kind, n_total, global_env_offset, (int)devices.size(), devices.data(), params, flags, &h_));
        int k = 0;
        check(gymrs_sharded_count(h_, &k));
        n_blocks_ = (std::size_t)k;
    }
    ~ShardedVecEnv() { gymrs_sharded_destroy(h_); }
    ShardedVecEnv(const ShardedVecEnv&) = delete;
    ShardedVecEnv& operator=(const ShardedVecEnv&) = delete;

    struct Block {
        gymrs_engine* engine; // for the zero-copy views (gymrs_obs_ptrs, gymrs_reward_ptr, ...); not to be stepped directly
        std::uint64_t first_lane, n_lanes;
        int device;
    };
    int n_blocks()
    {
        int k = 0;
        check(gymrs_sharded_count(h_, &k));
        return k;
    }
    Block block(int r)
    {
        Block b{};
        check(gymrs_sharded_shard(h_, r, &b.engine, &b.first_lane, &b.n_lanes, &b.device));
        return b;
    }
    std::uint64_t reset(std::optional<std::uint64_t> seed, const float* bounds_low_high = nullptr)
    {
        std::uint64_t used = 0;
        check(gymrs_sharded_reset(h_, seed.has_value(), seed.value_or(0), bounds_low_high, &used));
        return used;
    }
    // actions_dev[r]: block r's actions (ring) on ITS device
    // (the native side reads actions_dev[r] for EVERY block: a shorter vector would be an out-of-bounds read that ends as a wild device pointer)
    void step_device(const std::vector<const void*>& actions_dev) // async
    {
        one_pointer_per_block(actions_dev.size(), "step_device");
        check(gymrs_sharded_step(h_, actions_dev.data()));
    }
    void step_many(const std::vector<const void*>& actions_dev, std::uint64_t stride_bytes, std::uint32_t n_buffers, std::uint32_t n_steps)
    {
        one_pointer_per_block(actions_dev.size(), "step_many");
        check(gymrs_sharded_step_many(h_, actions_dev.data(), stride_bytes, n_buffers, n_steps, 0));
    }
    void fill_actions(const std::vector<void*>& actions_dev, std::uint64_t seed, std::uint64_t t)
    {
        one_pointer_per_block(actions_dev.size(), "fill_actions");
        check(gymrs_sharded_fill_actions(h_, actions_dev.data(), seed, t));
    }
    void rollout(std::uint32_t n_steps, std::uint64_t action_seed, std::uint64_t action_t0 = 0) { check(gymrs_sharded_rollout(h_, n_steps, action_seed, action_t0)); }
    void set_params(const void* params) { check(gymrs_sharded_set_params(h_, params)); }
    void sync() { check(gymrs_sharded_sync(h_)); }
    std::array<double, 4> stats() // {sum_return, sum_length, n_episodes, n_steps} of the whole batch
    {
        std::array<double, 4> out{};
        check(gymrs_sharded_stats(h_, out.data()));
        return out;
    }
    void stats_clear() { check(gymrs_sharded_stats_clear(h_)); }
    std::string reduce_path() { return gymrs_sharded_reduce_path(h_); } // "rccl" | "host" | "none"
    std::vector<float> state(std::uint64_t first, std::uint64_t count)
    {
        std::vector<float> out(count * state_dim_);
        check(gymrs_sharded_get_state(h_, first, count, out.data()));
        return out;
    }
    void result(std::uint64_t first, std::uint64_t count, float* reward, std::uint8_t* done, std::uint8_t* truncated)
    {
        check(gymrs_sharded_get_step_result(h_, first, count, reward, done, truncated));
    }
    std::uint64_t size() const { return n_; }

private:
    void one_pointer_per_block(std::size_t given, const char* who) const
    {
        if (given != n_blocks_)
            throw std::invalid_argument(std::string("ShardedVecEnv::") + who + ": " + std::to_string(given) + " action pointers for " + std::to_string(n_blocks_) + " blocks");
    }
    std::uint64_t n_;
    int state_dim_;
    std::size_t n_blocks_ = 0;
    gymrs_sharded* h_ = nullptr;
};

// ---- single envs with the reference's surface ------------------------------------------------------
class CartPoleEnv {
public:
    using Action = std::size_t;
    using Observation = CartPoleObservation;
    explicit CartPoleEnv(RenderMode mode = RenderMode::None) : env_(make(mode)) {}
    ResetRng reset_rng = ResetRng::Philox;

    ActionReward<Observation, Unit> step(Action action) // Env::step, cartpole.rs:398-483
    {
        if (!action_space().contains(action)) throw Panic(GYMRS_EACTION, std::to_string(action) + " usize invalid"); // :402-406
        const std::uint8_t a = static_cast<std::uint8_t>(action);
        env_.step_host(&a);
        float r;
        std::uint8_t d;
        env_.result(0, 1, &r, &d, nullptr);
        return {state(), r, d != 0, false, Unit{}}; // truncated: false, info: Some(()) (:480-481)
    }
    std::pair<Observation, std::optional<Unit>> reset(std::optional<std::uint64_t> seed, bool return_info,
                                                       std::optional<BoxR<Observation>> options) // :485-516
    {
        float b[8];
        double b64[8];
        if (options) {
            const auto lo = options->low.to_vec(), hi = options->high.to_vec();
            for (int j = 0; j < 4; ++j) {
                b[j] = static_cast<float>(b64[j] = lo[j]);
                b[4 + j] = static_cast<float>(b64[4 + j] = hi[j]);
            }
        }
        if (reset_rng == ResetRng::Pcg64)
            env_.reset_pcg64(seed, options ? b64 : nullptr);
        else
            env_.reset(seed, options ? b : nullptr);
        return {state(), return_info ? std::optional<Unit>(Unit{}) : std::nullopt};
    }
    void render(RenderMode) {} // renderer.rs:52-62: nothing is drawn under RenderMode::None
    void close() {}
    Observation state()
    {
        const auto s = env_.state(0, 1);
        return {s[0], s[1], s[2], s[3]};
    }
    void set_state(const Observation& o)
    {
        const float s[4] = {(float)o.x, (float)o.x_dot, (float)o.theta, (float)o.theta_dot};
        env_.set_state(0, 1, s);
    }
    // The reference's constants are pub fields (cartpole.rs:53-82): read them all, edit, assign them back.  Like the
    // assignment in Rust this changes nothing but the constants (state, steps_beyond_terminated, seed/tick carry on).
    gymrs_cartpole_params params()
    {
        gymrs_cartpole_params p;
        env_.get_params(&p);
        return p;
    }
    void set_params(const gymrs_cartpole_params& p) { env_.set_params(&p); }
    std::string to_json() { return env_.to_json(0); } // serde_json::to_string(&env)
    Discrete action_space() const { return Discrete{2}; } // cartpole.rs:114
    BoxR<Observation> observation_space()               // cartpole.rs:105-115
    {
        const double inf = std::numeric_limits<double>::infinity();
        const gymrs_cartpole_params p = params();
        const Observation high{p.x_threshold * 2., inf, p.theta_threshold_radians * 2., inf};
        return {-high, high};
    }
    RewardRange reward_range() const { return {}; }
    RenderMode render_mode() const { return RenderMode::None; }

private:
    static VecEnv make(RenderMode mode)
    {
        if (mode != RenderMode::None) throw Panic(GYMRS_EINVAL, "rendering is out of scope: use RenderMode::None");
        return VecEnv(GYMRS_CARTPOLE, 1);
    }
    VecEnv env_;
};

class MountainCarEnv {
public:
    using Action = std::size_t;
    using Observation = MountainCarObservation;
    explicit MountainCarEnv(RenderMode mode = RenderMode::None) : env_(make(mode)) {}
    ResetRng reset_rng = ResetRng::Philox;

    ActionReward<Observation, Unit> step(Action action) // mountain_car.rs:398-435
    {
        if (!action_space().contains(action)) throw Panic(GYMRS_EACTION, std::to_string(action) + " (usize) invalid");
        const std::uint8_t a = static_cast<std::uint8_t>(action);
        env_.step_host(&a);
        float r;
        std::uint8_t d;
        env_.result(0, 1, &r, &d, nullptr);
        return {state(), r, d != 0, false, std::nullopt}; // info: None (:433)
    }
    std::pair<Observation, std::optional<Unit>> reset(std::optional<std::uint64_t> seed, bool return_info,
                                                       std::optional<BoxR<Observation>> options) // :464-501
    {
        float b[4];
        double b64[4];
        if (options) {
            b[0] = (float)(b64[0] = options->low.position);
            b[1] = (float)(b64[1] = options->low.velocity);
            b[2] = (float)(b64[2] = options->high.position);
            b[3] = (float)(b64[3] = options->high.velocity);
        }
        if (reset_rng == ResetRng::Pcg64)
            env_.reset_pcg64(seed, options ? b64 : nullptr);
        else
            env_.reset(seed, options ? b : nullptr);
        return {state(), return_info ? std::optional<Unit>(Unit{}) : std::nullopt};
    }
    void render(RenderMode) {}
    void close() {}
    Observation state()
    {
        const auto s = env_.state(0, 1);
        return {s[0], s[1]};
    }
    void set_state(const Observation& o)
    {
        const float s[2] = {(float)o.position, (float)o.velocity};
        env_.set_state(0, 1, s);
    }
    gymrs_mountain_car_params params()
    {
        gymrs_mountain_car_params p;
        env_.get_params(&p);
        return p;
    }
    void set_params(const gymrs_mountain_car_params& p) { env_.set_params(&p); }
    std::string to_json() { return env_.to_json(0); }
    Discrete action_space() const { return Discrete{3}; }                                    // mountain_car.rs:362
    BoxR<Observation> observation_space() const { return {{-1.2, -0.07}, {0.6, 0.07}}; }      // :353-364

private:
    static VecEnv make(RenderMode mode)
    {
        if (mode != RenderMode::None) throw Panic(GYMRS_EINVAL, "rendering is out of scope: use RenderMode::None");
        return VecEnv(GYMRS_MOUNTAIN_CAR, 1);
    }
    VecEnv env_;
};

} // namespace gymrs
