// C++ test of the trait mirror (include/gymrs_env.hpp) on a real GPU: reads like a test of the reference
// crate.  Expected values: SURVEY.md Appendix C (tests/golden/*.json hold the same numbers).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "gymrs_env.hpp"

#define REQUIRE(cond)                                                              \
    do {                                                                           \
        if (!(cond)) {                                                             \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);          \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

// the sharded block below needs device buffers for its action rings; this test is built with plain g++ (no HIP headers), so the two HIP entry
// points it uses are declared by hand (hipError_t is an int-sized enum, hipSuccess = 0) and libamdhip64 is named on the link line
extern "C" int hipSetDevice(int device);
extern "C" int hipMalloc(void** ptr, std::size_t bytes);
extern "C" int hipFree(void* ptr);
extern "C" int hipGetDeviceCount(int* count);

static bool close_to(double a, double b, double rel) { return std::fabs(a - b) <= rel * std::fmax(std::fabs(b), 1.0); }

int main()
{
    using namespace gymrs;
    {
        CartPoleEnv env(RenderMode::None);
        REQUIRE(env.action_space() == Discrete{2});
        REQUIRE(env.action_space().contains(1) && !env.action_space().contains(2));
        const auto space = env.observation_space();
        REQUIRE(space.high.x == 4.8 && std::isinf(space.high.x_dot) && close_to(space.high.theta, 0.41887902047863906, 1e-15));
        auto [obs, info] = env.reset(64, true, std::nullopt);
        REQUIRE(info.has_value() && std::fabs(obs.x) < 0.05 && std::fabs(obs.theta_dot) < 0.05);
        auto [obs2, info2] = env.reset(64, false, std::nullopt);
        REQUIRE(!info2.has_value() && obs2.x == obs.x && obs2.theta == obs.theta); // same seed, same state
        // single step from (0.01, 0.02, 0.03, 0.04), action 1: 0.356 (reference), not Gym's 0.215 (Q1)
        env.set_state({0.01, 0.02, 0.03, 0.04});
        auto ar = env.step(1);
        REQUIRE(close_to(ar.observation.x_dot, 0.35615076996399875, 1e-6) && close_to(ar.observation.theta_dot, -0.2430694901285738, 1e-6));
        REQUIRE(ar.reward == 1.0 && !ar.done && !ar.truncated && ar.info.has_value());
        // assigning a pub physics field keeps the episode: the terminating step pays 1.0, the next one 0.0 even though
        // gravity was changed in between (cartpole.rs:455-464); the serde view carries the reference's field names
        env.set_state({2.39, 3.0, 0.0, 0.0});
        auto term = env.step(1);
        REQUIRE(term.done && term.reward == 1.0);
        auto p = env.params();
        REQUIRE(p.gravity == 9.8);
        p.gravity = 19.6;
        env.set_params(p);
        REQUIRE(env.params().gravity == 19.6);
        auto beyond = env.step(1);
        REQUIRE(beyond.done && beyond.reward == 0.0);
        const std::string js = env.to_json();
        REQUIRE(js.find("\"gravity\":19.6") != std::string::npos && js.find("\"steps_beyond_terminated\":0") != std::string::npos);
        REQUIRE(js.find("\"kinematics_integrator\":\"Euler\"") != std::string::npos);
        p.gravity = 9.8;
        env.set_params(p);
        // trajectories: always 1 -> 10 steps, always 0 -> 9, alternate 1,0,... -> 60
        const int expect[3] = {10, 9, 60};
        for (int p = 0; p < 3; ++p) {
            env.reset(0, false, std::nullopt);
            env.set_state({0.01, 0.02, 0.03, 0.04});
            int t = 0;
            double total = 0;
            for (;;) {
                const std::size_t a = p == 0 ? 1 : p == 1 ? 0 : (t + 1) % 2;
                auto r = env.step(a);
                total += r.reward;
                ++t;
                if (r.done) break;
            }
            REQUIRE(t == expect[p] && total == (double)t);
        }
        // stepping past termination: 1.0 once more, then 0.0 (cartpole.rs:455-464)
        REQUIRE(env.step(1).reward == 0.0);
        bool panicked = false;
        try {
            env.step(2);
        } catch (const Panic& e) {
            panicked = e.status == GYMRS_EACTION;
        }
        REQUIRE(panicked);
        // options override the sampling box
        auto [o3, i3] = env.reset(1, false, BoxR<CartPoleObservation>{{1., 2., 0.1, 5.}, {1.5, 3., 0.2, 6.}});
        REQUIRE(o3.x >= 1. && o3.x < 1.5 && o3.x_dot >= 2. && o3.theta >= 0.0999 && o3.theta_dot >= 5. && o3.theta_dot < 6.);
        // the reference's own reset stream: Pcg64::seed_from_u64(42) + four Uniform draws (oracle/gymrs_oracle.c,
        // orc_cartpole_reset_pcg64; tests/test_pcg64_reset.py pins the chain), rounded once to f32
        env.reset_rng = ResetRng::Pcg64;
        auto [o4, i4] = env.reset(42, false, std::nullopt);
        REQUIRE(o4.x == (double)(float)-0.027348748207168663 && o4.x_dot == (double)(float)-0.02608933096814006);
        REQUIRE(o4.theta == (double)(float)0.02608334106867409 && o4.theta_dot == (double)(float)0.002783965735815165);
        REQUIRE(!env.step(0).done);
    }
    {
        MountainCarEnv env(RenderMode::None);
        REQUIRE(env.action_space() == Discrete{3});
        auto [obs, info] = env.reset(1, false, std::nullopt);
        REQUIRE(obs.position >= -0.6 - 1e-7 && obs.position < -0.4 && obs.velocity == 0.0);
        env.reset_rng = ResetRng::Pcg64; // mountain_car.rs:464-501 with the reference's generator: one draw, velocity 0
        auto [pc, pi] = env.reset(42, false, std::nullopt);
        REQUIRE(pc.position == (double)(float)-0.5546974964143373 && pc.velocity == 0.0);
        env.reset_rng = ResetRng::Philox;
        env.set_state({-0.5, 0.0});
        int t = 0;
        double total = 0;
        for (;;) {
            auto r = env.step(env.state().velocity >= 0 ? 2 : 0);
            total += r.reward;
            ++t;
            REQUIRE(!r.info.has_value() && !r.truncated);
            if (r.done) break;
        }
        REQUIRE(t == 124 && total == -124.0);
        env.set_state({-1.19, -0.07});
        auto r = env.step(0);
        REQUIRE(close_to(r.observation.position, -1.2, 1e-6) && r.observation.velocity == 0.0 && !r.done); // wall rule
    }
    {
        // batched form: 4096 lanes, one launch per step, host actions
        VecEnv env(GYMRS_CARTPOLE, 4096, GYMRS_AUTO_RESET | GYMRS_TRACK_STATS);
        env.reset(7);
        std::vector<std::uint8_t> act(4096, 1);
        for (int t = 0; t < 50; ++t) env.step_host(act.data());
        double st[4];
        check(gymrs_stats(env.handle(), st));
        REQUIRE(st[3] == 4096.0 * 50 && st[2] > 0 && st[0] == st[1]);
    }
    {
        // the examples' loop fused into one launch; Clone and snapshot/restore continue identically
        VecEnv a(GYMRS_MOUNTAIN_CAR, 10000, GYMRS_AUTO_RESET | GYMRS_TRACK_STATS | GYMRS_TIME_LIMIT);
        a.reset(3);
        a.rollout(150, 11, 0);
        VecEnv b(a); // Env: Clone
        const std::vector<unsigned char> blob = a.snapshot();
        VecEnv c(GYMRS_MOUNTAIN_CAR, 10000, GYMRS_AUTO_RESET | GYMRS_TRACK_STATS | GYMRS_TIME_LIMIT);
        c.restore(blob);
        a.rollout(150, 11, 150);
        b.rollout(150, 11, 150);
        c.rollout(150, 11, 150);
        const auto sa = a.stats(), sb = b.stats(), sc = c.stats();
        REQUIRE(sa[3] == 10000.0 * 300 && sa[2] >= 10000.0 && sa[0] == -sa[1]); // 200-step limit: every lane finished once
        REQUIRE(sa == sb && sa == sc);
        REQUIRE(a.state(0, 10000) == b.state(0, 10000) && a.state(0, 10000) == c.state(0, 10000));
    }
    {
        // one batch over several GPUs in ONE process (gymrs_sharded_*): 3 blocks == 1 engine, state bits and statistics.  The blocks take the box's GPUs
        // round-robin (one GPU: they share it and the statistics are summed on the host; distinct GPUs: the grouped RCCL all-reduce)
        int n_dev = 0;
        REQUIRE(hipGetDeviceCount(&n_dev) == 0 && n_dev >= 1);
        const std::uint64_t n = 3 * 4096 + 100;
        const std::uint32_t flags = GYMRS_AUTO_RESET | GYMRS_TRACK_STATS, nbuf = 4;
        ShardedVecEnv sh(GYMRS_CARTPOLE, n, {0, 1 % n_dev, 2 % n_dev}, flags);
        VecEnv one(GYMRS_CARTPOLE, n, flags);
        REQUIRE(sh.n_blocks() == 3 && sh.block(0).first_lane == 0 && sh.block(1).first_lane == sh.block(0).n_lanes);
        REQUIRE(sh.block(0).n_lanes + sh.block(1).n_lanes + sh.block(2).n_lanes == n);
        sh.reset(21);
        one.reset(21);
        const std::uint64_t pitch = 8192; // >= every block's lanes: one ring stride for all blocks
        std::vector<void*> rings(3);
        std::vector<const void*> crings(3);
        for (int r = 0; r < 3; ++r) {
            REQUIRE(hipSetDevice(sh.block(r).device) == 0 && hipMalloc(&rings[r], nbuf * pitch) == 0);
            crings[r] = rings[r];
        }
        void* ring1 = nullptr;
        REQUIRE(hipSetDevice(0) == 0 && hipMalloc(&ring1, nbuf * n) == 0);
        for (std::uint32_t b = 0; b < nbuf; ++b) {
            std::vector<void*> row(3);
            for (int r = 0; r < 3; ++r) row[r] = static_cast<char*>(rings[r]) + b * pitch;
            sh.fill_actions(row, 1, b);
            check(gymrs_fill_actions(one.handle(), static_cast<char*>(ring1) + b * n, 1, b));
        }
        sh.step_device(crings); // one step with buffer 0, then 99 more through step_many
        one.step_device(ring1);
        sh.step_many(crings, pitch, nbuf, 99);
        one.step_many(ring1, n, nbuf, 99);
        sh.sync();
        one.sync();
        REQUIRE(sh.state(0, n) == one.state(0, n));
        const auto a = sh.stats(), b1 = one.stats();
        REQUIRE(a == b1 && a[3] == 100.0 * n && a[2] > 0);
        REQUIRE(sh.reduce_path() == (n_dev >= 3 ? "rccl" : "host"));
        for (int r = 0; r < 3; ++r) hipFree(rings[r]);
        hipFree(ring1);
    }
    std::printf("CPP_MIRROR_OK\n");
    return 0;
}
