/*
 * gymrs_oracle.h — CPU f64 ORACLE for the gym-rs classic-control hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (gym-rs_amd/, include/)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker / the timed CPU baseline.
 *
 * It is a plain-C restatement (NOT a copy: the reference is Rust) of
 *   /root/reference/src/envs/classical_control/cartpole.rs:146-152,398-483,485-516
 *   /root/reference/src/envs/classical_control/mountain_car.rs:398-435,464-501
 *   /root/reference/src/utils/custom/util_fns.rs:2-10      (clip)
 *   /root/reference/src/spaces/discrete.rs:14-19           (Discrete::contains)
 *   /root/reference/src/utils/seeding.rs:21-26             (seed echo)
 * in the reference's own arithmetic type (f64) and operation order.
 *
 * PARITY PINNING STATUS
 *   - clip, Discrete::contains, seed echo: pinned against the reference's own unit
 *     tests (util_fns.rs:16-32, discrete.rs:27-41, seeding.rs:33-39) in
 *     tests/test_oracle_reference_pins.py.
 *   - step()/reset() physics: **parity unpinned** by the reference — the reference
 *     holds no golden vectors or tests for step/reset, and no Rust toolchain exists
 *     in this image to run it (cargo/rustc missing), so oracle/_ref cannot be built.
 *     The pin is the source text; tests/golden/ holds vectors produced by an
 *     independent literal evaluation of that text (tests/golden/make_golden.py).
 *   - Pendulum: NOT in the reference at all (classical_control/mod.rs:1-4 exports
 *     only cartpole and mountain_car).  Spec-derived from the public Gym definition;
 *     **parity unpinned**.
 *   - reset sampling: the reference uses rand_pcg::Pcg64 + rand 0.8 Uniform (third
 *     party, sources absent).  north_star mandates counter-based Philox4x32-10
 *     instead; the oracle restates Philox (Random123 KATs in tests) and the
 *     reference's sampling *order and ranges* (cartpole.rs:317-324,353-361,
 *     mountain_car.rs:162-167,176-187).
 */
#ifndef GYMRS_ORACLE_H
#define GYMRS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference helper functions --------------------------------------------------- */
/* util_fns.rs:2-10 */
double orc_clip(double value, double left_bound, double right_bound);
long orc_clip_i64(long value, long left_bound, long right_bound);
/* discrete.rs:14-19 : value < n */
int orc_discrete_contains(size_t n, size_t value);
/* seeding.rs:21-26 : returns the seed number that would be used (echo when has_seed) */
uint64_t orc_rand_random_seed(int has_seed, uint64_t seed, uint64_t os_entropy);

/* ---- Philox4x32-10 (Random123), independent restatement --------------------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* u32 -> [0,1) with 24 random bits, in f64 */
double orc_u01(uint32_t r);

/* ---- CartPole --------------------------------------------------------------------- */
typedef struct {
    double gravity, masscart, masspole, length, force_mag, tau; /* cartpole.rs:94-99 */
    double theta_threshold_radians, x_threshold;                /* cartpole.rs:102-103 */
    int kinematics_integrator; /* 0 = Euler (default, cartpole.rs:100), 1 = Other */
} orc_cartpole_params;

typedef struct {
    double x, x_dot, theta, theta_dot; /* cartpole.rs:328-334 */
    int has_steps_beyond;              /* Option<usize> cartpole.rs:82 */
    long steps_beyond;
} orc_cartpole_env;

typedef struct {
    double obs[4]; /* order x, x_dot, theta, theta_dot: cartpole.rs:336-349 */
    double reward;
    int done;
    int truncated; /* always 0: cartpole.rs:480 */
} orc_step_result;

void orc_cartpole_default_params(orc_cartpole_params *p);
/* returns 0, or -1 when the action is invalid (reference panics: cartpole.rs:402-406;
 * state is left untouched in that case, as the assert precedes every state access). */
int orc_cartpole_step(orc_cartpole_env *e, const orc_cartpole_params *p, size_t action,
                      orc_step_result *out);
/* reset(seed, options): Philox key=(seed lo,hi), counter=(gid lo, gid hi, tick lo, tick hi);
 * bounds = {low[4], high[4]} or NULL for +-0.05 (cartpole.rs:352-364). */
void orc_cartpole_reset(orc_cartpole_env *e, uint64_t seed, uint64_t gid, uint64_t tick,
                        const double *bounds_low_high);

/* ---- the reference's own reset stream: Pcg64::seed_from_u64 + Uniform<f64> (seeding.rs:21-26) ---- */
typedef struct {
    uint64_t state_lo, state_hi, incr_lo, incr_hi;
} orc_pcg64;
void orc_pcg64_new(orc_pcg64 *g, uint64_t state_lo, uint64_t state_hi, uint64_t stream_lo, uint64_t stream_hi);
void orc_pcg64_from_seed(orc_pcg64 *g, const uint8_t seed[32]);
void orc_pcg64_seed_from_u64(orc_pcg64 *g, uint64_t seed);
uint64_t orc_pcg64_next_u64(orc_pcg64 *g);
int orc_uniform_f64_new(double low, double high, double *scale_out); /* -1 where Uniform::new panics */
double orc_uniform_f64_sample(orc_pcg64 *g, double low, double scale);
/* reset(Some(seed), _, bounds) of the reference in f64; bounds = lows then highs, or NULL; -1 where it would panic */
int orc_cartpole_reset_pcg64(orc_cartpole_env *e, uint64_t seed, const double *bounds_low_high);

/* ---- MountainCar ------------------------------------------------------------------ */
typedef struct {
    double min_position, max_position, max_speed, goal_position, goal_velocity; /* mountain_car.rs:344-348 */
    double force, gravity;                                                      /* mountain_car.rs:350-351 */
} orc_mountain_car_params;

typedef struct {
    double position, velocity; /* mountain_car.rs:122-128 */
} orc_mountain_car_env;

void orc_mountain_car_default_params(orc_mountain_car_params *p);
int orc_mountain_car_step(orc_mountain_car_env *e, const orc_mountain_car_params *p, size_t action,
                          orc_step_result *out);
void orc_mountain_car_reset(orc_mountain_car_env *e, uint64_t seed, uint64_t gid, uint64_t tick,
                            const double *bounds_low_high /* {low_pos, high_pos} or NULL */);
int orc_mountain_car_reset_pcg64(orc_mountain_car_env *e, uint64_t seed, const double *bounds_low_high);

/* ---- Pendulum (spec-derived; NOT in the reference; parity unpinned) ---------------- */
typedef struct {
    double max_speed, max_torque, dt, g, m, l;
} orc_pendulum_params;

typedef struct {
    double theta, theta_dot;
} orc_pendulum_env;

void orc_pendulum_default_params(orc_pendulum_params *p);
/* obs = (cos, sin, theta_dot); never terminates. */
int orc_pendulum_step(orc_pendulum_env *e, const orc_pendulum_params *p, double action,
                      orc_step_result *out);
void orc_pendulum_reset(orc_pendulum_env *e, uint64_t seed, uint64_t gid, uint64_t tick,
                        const double *bounds_low_high /* {low_th, low_thd, high_th, high_thd} or NULL */);

/* ---- batch drivers over SoA f64 arrays (for numpy/ctypes parity tests) ------------- */
/* One step() per lane.  beyond[] is the per-lane steps_beyond_terminated.is_some() flag
 * (in/out).  Returns the number of lanes whose action was invalid (those lanes are left
 * untouched and get reward 0 / done 0). */
long orc_cartpole_step_batch(size_t n, double *x, double *x_dot, double *theta, double *theta_dot,
                             uint8_t *beyond, const uint8_t *action, const orc_cartpole_params *p,
                             double *reward, uint8_t *done);
long orc_mountain_car_step_batch(size_t n, double *position, double *velocity, const uint8_t *action,
                                 const orc_mountain_car_params *p, double *reward, uint8_t *done);
long orc_pendulum_step_batch(size_t n, double *theta, double *theta_dot, const double *action,
                             const orc_pendulum_params *p, double *obs_cos, double *obs_sin,
                             double *reward);
void orc_cartpole_reset_batch(size_t n, uint64_t gid0, uint64_t seed, uint64_t tick,
                              const double *bounds_low_high, double *x, double *x_dot,
                              double *theta, double *theta_dot);
void orc_mountain_car_reset_batch(size_t n, uint64_t gid0, uint64_t seed, uint64_t tick,
                                  const double *bounds_low_high, double *position, double *velocity);
void orc_pendulum_reset_batch(size_t n, uint64_t gid0, uint64_t seed, uint64_t tick,
                              const double *bounds_low_high, double *theta, double *theta_dot);

/* ---- the caller loop the CPU baseline times ---------------------------------------- */
/* Shape of /root/reference/examples/cartpole.rs:15-30 with RenderMode::None semantics:
 * one env, one thread; step with a random action until done (or max_episode_steps,
 * examples/cartpole.rs:18 uses 475; 0 = unlimited), then reset; repeat until n_steps
 * env.step() calls were made.  Returns elapsed seconds; fills out[4] =
 * {sum_return, sum_length, n_episodes, n_steps}.  kind: 0 CartPole, 1 MountainCar,
 * 2 Pendulum. */
double orc_baseline_loop(int kind, uint64_t n_steps, uint64_t max_episode_steps, uint64_t seed,
                         double out[4]);

#ifdef __cplusplus
}
#endif
#endif /* GYMRS_ORACLE_H */
