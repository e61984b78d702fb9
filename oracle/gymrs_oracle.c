/*
 * gymrs_oracle.c — CPU f64 ORACLE (test infrastructure; see gymrs_oracle.h header block).
 *
 * Compile with: gcc -O2 -fno-fast-math -ffp-contract=off  (so that no a*b+c is fused and the
 * operation order below is exactly the reference's, evaluated in IEEE f64 like Rust does).
 * Every function cites the reference lines it restates.  Parity status: step/reset physics
 * "parity unpinned" by the reference's own tests (it has none for them); clip / contains /
 * seed echo are pinned.  Pendulum is spec-derived (not in the reference).
 * The one-command pin by the reference itself (needs a Rust toolchain, absent here): bindings/rust/src/bin/make_golden.rs
 * runs gym-rs' own step()/reset() on the fixture inputs; tests/test_oracle_reference_pins.py then holds this file to it.
 */
#include "gymrs_oracle.h"

#include <math.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------ */
/* The reference's scalar type is O64 = OrderedFloat<f64> (utils/custom/types.rs:4).  Its
 * comparisons are a TOTAL order in which NaN equals NaN and is greater than every number
 * (ordered-float crate, [RECALLED]).  These helpers restate that order so that
 * every comparison below behaves like the reference's even for NaN states.              */
static int of_lt(double a, double b)
{
    if (isnan(a)) return 0;          /* NaN is the maximum: never less than anything */
    if (isnan(b)) return 1;          /* every number is less than NaN */
    return a < b;
}
static int of_gt(double a, double b) { return of_lt(b, a); }
static int of_le(double a, double b) { return !of_gt(a, b); }
static int of_ge(double a, double b) { return !of_lt(a, b); }
static int of_eq(double a, double b) { return (isnan(a) && isnan(b)) || a == b; }

/* util_fns.rs:2-10                                                                      */
/*   if left <= value && value <= right { value } else if value > right { right } else { left } */
double orc_clip(double value, double left_bound, double right_bound)
{
    if (of_le(left_bound, value) && of_le(value, right_bound)) {
        return value;
    } else if (of_gt(value, right_bound)) {
        return right_bound;
    } else {
        return left_bound;
    }
}

long orc_clip_i64(long value, long left_bound, long right_bound)
{
    if (left_bound <= value && value <= right_bound) {
        return value;
    } else if (value > right_bound) {
        return right_bound;
    } else {
        return left_bound;
    }
}

/* discrete.rs:14-19 : Discrete(upper_bound) => value < upper_bound */
int orc_discrete_contains(size_t n, size_t value) { return value < n; }

/* seeding.rs:21-26 : seed_no = seed.unwrap_or(thread_rng().gen()) ; echo of the seed */
uint64_t orc_rand_random_seed(int has_seed, uint64_t seed, uint64_t os_entropy)
{
    return has_seed ? seed : os_entropy;
}

/* ------------------------------------------------------------------------------------ */
/* Philox4x32-10, Salmon et al. SC'11 (Random123).  Constants cross-checked against
 * /opt/rocm/include/rocrand/rocrand_philox4x32_10.h:62-65 (constants only).            */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

double orc_u01(uint32_t r) { return (double)(r >> 8) * (1.0 / 16777216.0); }

/* Counter layout shared with the engine (DESIGN.md "RNG"): key = seed (lo, hi);
 * counter = (gid lo, gid hi, tick lo, (tick hi & 0xffff) | stream << 16).
 * stream 0 = reset sampling, 1 = synthetic action generation. */
static void draw4(uint64_t seed, uint64_t gid, uint64_t tick, uint32_t stream, uint32_t out[4])
{
    uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)tick,
                       ((uint32_t)(tick >> 32) & 0xffffu) | (stream << 16)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    orc_philox4x32_10(ctr, key, out);
}

/* Uniform on [low, high): u*scale + low, scale = high-low (rand 0.8 UniformFloat shape,
 * SURVEY Appendix B.2), guarded so the result stays < high. */
static double uniform_between(uint32_t r, double low, double high)
{
    double v = orc_u01(r) * (high - low) + low;
    if (!(v < high)) v = nextafter(high, low);
    return v;
}

/* ------------------------------------------------------------------------------------ */
/* CartPole                                                                              */

void orc_cartpole_default_params(orc_cartpole_params *p)
{
    /* cartpole.rs:94-103 */
    p->gravity = 9.8;
    p->masscart = 1.0;
    p->masspole = 0.1;
    p->length = 0.5;
    p->force_mag = 10.0;
    p->tau = 0.02;
    p->kinematics_integrator = 0;                                   /* Euler, cartpole.rs:100 */
    p->theta_threshold_radians = 12. * 2. * 3.14159265358979323846264338327950288 / 360.; /* cartpole.rs:102 */
    p->x_threshold = 2.4;                                           /* cartpole.rs:103 */
}

/* cartpole.rs:146-148 */
static double cp_total_mass(const orc_cartpole_params *p) { return p->masspole + p->masscart; }
/* cartpole.rs:150-152 — the reference ADDS (masspole + length = 0.6); Gym multiplies.
 * reproduce, do not fix. */
static double cp_polemass_length(const orc_cartpole_params *p) { return p->masspole + p->length; }

int orc_cartpole_step(orc_cartpole_env *e, const orc_cartpole_params *p, size_t action,
                      orc_step_result *out)
{
    /* cartpole.rs:402-406 : assert!(self.action_space.contains(action)) with Discrete(2) */
    if (!orc_discrete_contains(2, action)) return -1;

    /* cartpole.rs:408-413 */
    double x = e->x, x_dot = e->x_dot, theta = e->theta, theta_dot = e->theta_dot;
    /* cartpole.rs:414-418 */
    double force = (action == 1) ? p->force_mag : -p->force_mag;
    /* cartpole.rs:420-421 */
    double costheta = cos(theta);
    double sintheta = sin(theta);
    /* cartpole.rs:423-424 : (force + polemass_length * theta_dot^2 * sintheta) / total_mass
     * Rust evaluates a * b * c left to right: (pl * td^2) * sin */
    double temp = (force + (cp_polemass_length(p) * pow(theta_dot, 2.)) * sintheta) / cp_total_mass(p);
    /* cartpole.rs:425-428 */
    double thetaacc = (p->gravity * sintheta - costheta * temp) /
                      (p->length * ((4.0 / 3.0) - (p->masspole * pow(costheta, 2.)) / cp_total_mass(p)));
    /* cartpole.rs:429 : temp - polemass_length * thetaacc * costheta / total_mass */
    double xacc = temp - ((cp_polemass_length(p) * thetaacc) * costheta) / cp_total_mass(p);

    if (p->kinematics_integrator == 0) {
        /* cartpole.rs:431-435 (Euler) */
        x += p->tau * x_dot;
        x_dot += p->tau * xacc;
        theta += p->tau * theta_dot;
        theta_dot += p->tau * thetaacc;
    } else {
        /* cartpole.rs:436-441 (semi-implicit) */
        x_dot += p->tau * xacc;
        x += p->tau * x_dot;
        theta_dot += p->tau * thetaacc;
        theta += p->tau * theta_dot;
    }
    /* cartpole.rs:443-448 */
    e->x = x; e->x_dot = x_dot; e->theta = theta; e->theta_dot = theta_dot;

    /* cartpole.rs:450-453 — OrderedFloat total order: NaN is greater than every number, so a
     * NaN x satisfies `x > x_threshold`. */
    int done = of_lt(x, -p->x_threshold) || of_gt(x, p->x_threshold) ||
               of_lt(theta, -p->theta_threshold_radians) || of_gt(theta, p->theta_threshold_radians);

    /* cartpole.rs:455-464 */
    double reward;
    if (!done) {
        reward = 1.0;
    } else if (!e->has_steps_beyond) {
        e->has_steps_beyond = 1;
        e->steps_beyond = 0;
        reward = 1.0;
    } else {
        e->steps_beyond += 1; /* + warn!() */
        reward = 0.0;
    }
    /* cartpole.rs:476-482 */
    out->obs[0] = x; out->obs[1] = x_dot; out->obs[2] = theta; out->obs[3] = theta_dot;
    out->reward = reward;
    out->done = done;
    out->truncated = 0;
    return 0;
}

void orc_cartpole_reset(orc_cartpole_env *e, uint64_t seed, uint64_t gid, uint64_t tick,
                        const double *b)
{
    /* cartpole.rs:352-361 default bounds +-0.05 on all four fields */
    static const double dflt[8] = {-0.05, -0.05, -0.05, -0.05, 0.05, 0.05, 0.05, 0.05};
    if (!b) b = dflt;
    uint32_t r[4];
    draw4(seed, gid, tick, 0, r);
    /* cartpole.rs:317-324 : sampled in the order x, x_dot, theta, theta_dot */
    e->x = uniform_between(r[0], b[0], b[4]);
    e->x_dot = uniform_between(r[1], b[1], b[5]);
    e->theta = uniform_between(r[2], b[2], b[6]);
    e->theta_dot = uniform_between(r[3], b[3], b[7]);
    /* cartpole.rs:504 */
    e->has_steps_beyond = 0;
    e->steps_beyond = 0;
}

/* ------------------------------------------------------------------------------------ */
/* The reference's OWN reset stream (seeding.rs:21-26 -> rand_pcg::Pcg64, rand::distributions::Uniform).
 * Third-party crates absent from /root/reference (Cargo.toml:24-33: rand 0.8, rand_pcg 0.3; rand_core 0.6 through
 * them; Cargo.lock is git-ignored, so the patch versions are unpinned).  Their published algorithms, restated
 * [RECALLED, SURVEY App. B.2], pinned by rand_pcg's own test values (tests/golden/pcg64.json) and cross-checked
 * against numpy.random.PCG64 (the same LCG multiplier and XSL-RR output) in tests/test_pcg64_reset.py; the uniform
 * stage is pinned only once bindings/rust/src/bin/make_golden.rs has written from_reference/reset_kat.json.  */
typedef unsigned __int128 u128;
#define PCG_MUL ((((u128)0x2360ED051FC65DA4ull) << 64) | (u128)0x4385DF649FCCF645ull)

static void pcg64_step(orc_pcg64 *g)
{
    u128 st = ((u128)g->state_hi << 64) | g->state_lo;
    const u128 inc = ((u128)g->incr_hi << 64) | g->incr_lo;
    st = st * PCG_MUL + inc;
    g->state_lo = (uint64_t)st;
    g->state_hi = (uint64_t)(st >> 64);
}

/* rand_pcg 0.3 Lcg128Xsl64::from_state_incr: "move away from the initial value" */
static void pcg64_from_state_incr(orc_pcg64 *g, u128 state, u128 incr)
{
    state += incr;
    g->state_lo = (uint64_t)state;
    g->state_hi = (uint64_t)(state >> 64);
    g->incr_lo = (uint64_t)incr;
    g->incr_hi = (uint64_t)(incr >> 64);
    pcg64_step(g);
}

/* Lcg128Xsl64::new(state, stream): increment = (stream << 1) | 1 */
void orc_pcg64_new(orc_pcg64 *g, uint64_t state_lo, uint64_t state_hi, uint64_t stream_lo, uint64_t stream_hi)
{
    const u128 stream = ((u128)stream_hi << 64) | stream_lo;
    pcg64_from_state_incr(g, ((u128)state_hi << 64) | state_lo, (stream << 1) | 1);
}

/* Lcg128Xsl64::from_seed: four little-endian u64; state = words 0,1; increment = words 2,3 with bit 0 forced */
void orc_pcg64_from_seed(orc_pcg64 *g, const uint8_t seed[32])
{
    uint64_t w[4];
    for (int i = 0; i < 4; ++i) {
        w[i] = 0;
        for (int b = 7; b >= 0; --b) w[i] = (w[i] << 8) | seed[8 * i + b];
    }
    pcg64_from_state_incr(g, ((u128)w[1] << 64) | w[0], (((u128)w[3] << 64) | w[2]) | 1);
}

/* rand_core 0.6 SeedableRng::seed_from_u64: a PCG32 (XSH-RR) fills the seed 4 bytes at a time */
void orc_pcg64_seed_from_u64(orc_pcg64 *g, uint64_t state)
{
    uint8_t seed[32];
    for (int chunk = 0; chunk < 8; ++chunk) {
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        const uint32_t rot = (uint32_t)(state >> 59);
        const uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        for (int b = 0; b < 4; ++b) seed[4 * chunk + b] = (uint8_t)(x >> (8 * b));
    }
    orc_pcg64_from_seed(g, seed);
}

/* next_u64: step, then XSL-RR 128/64 */
uint64_t orc_pcg64_next_u64(orc_pcg64 *g)
{
    pcg64_step(g);
    const uint32_t rot = (uint32_t)(g->state_hi >> 58);
    const uint64_t xsl = g->state_hi ^ g->state_lo;
    return (xsl >> rot) | (xsl << ((64 - rot) & 63));
}

/* rand 0.8 UniformFloat<f64>::new(low, high) -> scale; returns -1 where the reference panics */
int orc_uniform_f64_new(double low, double high, double *scale_out)
{
    if (!(low < high)) return -1;                  /* "Uniform::new called with `low >= high`" */
    if (!isfinite(low) || !isfinite(high)) return -1; /* "... non-finite boundaries" */
    const double max_rand = 1.0 - 0x1p-52;          /* (u64::MAX >> 12) as a double in [1,2), minus 1 */
    double scale = high - low;
    if (!isfinite(scale)) return -1;                /* "Uniform::new: range overflow" */
    /* NOTE: the loop takes ~ulp(high) / (2 ulp(scale)) rounds: bounds like [1e5, 1e5 + 1e-10) keep the reference
     * busy for 1e15 iterations.  The oracle gives up after 2^22 rounds (-2: "the reference would hang"). */
    for (long rounds = 0; scale * max_rand + low >= high; ++rounds) {
        if (rounds == (1L << 22)) return -2;
        uint64_t bits;
        memcpy(&bits, &scale, 8);
        bits -= 1;                                  /* decrease_masked: one ulp down */
        memcpy(&scale, &bits, 8);
    }
    *scale_out = scale;
    return 0;
}

/* UniformFloat<f64>::sample: 52 bits into [1,2), minus 1, times scale, plus low */
double orc_uniform_f64_sample(orc_pcg64 *g, double low, double scale)
{
    const uint64_t bits = (orc_pcg64_next_u64(g) >> 12) | 0x3FF0000000000000ull;
    double value1_2;
    memcpy(&value1_2, &bits, 8);
    const double value0_1 = value1_2 - 1.0;
    return value0_1 * scale + low;
}

/* cartpole.rs:485-516 with seed = Some(seed): Pcg64::seed_from_u64, then x, x_dot, theta, theta_dot in that order
 * (cartpole.rs:317-324) from four samplers built BEFORE the first draw (cartpole.rs:293-297). */
int orc_cartpole_reset_pcg64(orc_cartpole_env *e, uint64_t seed, const double *b)
{
    static const double dflt[8] = {-0.05, -0.05, -0.05, -0.05, 0.05, 0.05, 0.05, 0.05};
    if (!b) b = dflt;
    double scale[4], v[4];
    for (int j = 0; j < 4; ++j)
        if (orc_uniform_f64_new(b[j], b[4 + j], &scale[j])) return -1;
    orc_pcg64 g;
    orc_pcg64_seed_from_u64(&g, seed);
    for (int j = 0; j < 4; ++j) v[j] = orc_uniform_f64_sample(&g, b[j], scale[j]);
    e->x = v[0];
    e->x_dot = v[1];
    e->theta = v[2];
    e->theta_dot = v[3];
    e->has_steps_beyond = 0; /* cartpole.rs:504 */
    e->steps_beyond = 0;
    return 0;
}

/* mountain_car.rs:464-501: one draw for the position (:145), velocity = 0 (:162-167); bounds {low pos, low vel, high pos, high vel} */
int orc_mountain_car_reset_pcg64(orc_mountain_car_env *e, uint64_t seed, const double *b)
{
    static const double dflt[4] = {-0.6, 0.0, -0.4, 0.0};
    if (!b) b = dflt;
    double scale;
    if (orc_uniform_f64_new(b[0], b[2], &scale)) return -1;
    orc_pcg64 g;
    orc_pcg64_seed_from_u64(&g, seed);
    e->position = orc_uniform_f64_sample(&g, b[0], scale);
    e->velocity = 0.0;
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* MountainCar                                                                           */

void orc_mountain_car_default_params(orc_mountain_car_params *p)
{
    /* mountain_car.rs:344-351 */
    p->min_position = -1.2;
    p->max_position = 0.6;
    p->max_speed = 0.07;
    p->goal_position = 0.5;
    p->goal_velocity = 0.;
    p->force = 0.001;
    p->gravity = 0.0025;
}

int orc_mountain_car_step(orc_mountain_car_env *e, const orc_mountain_car_params *p, size_t action,
                          orc_step_result *out)
{
    /* mountain_car.rs:402-406 with Discrete(3) */
    if (!orc_discrete_contains(3, action)) return -1;
    /* mountain_car.rs:408-409 */
    double position = e->position;
    double velocity = e->velocity;
    /* mountain_car.rs:411-412 */
    velocity += ((double)action - 1.) * p->force + cos(3. * position) * (-p->gravity);
    /* mountain_car.rs:413 */
    velocity = orc_clip(velocity, -p->max_speed, p->max_speed);
    /* mountain_car.rs:415-416 */
    position += velocity;
    position = orc_clip(position, p->min_position, p->max_position);
    /* mountain_car.rs:418-420 — exact float equality against the clip bound */
    if (of_eq(position, p->min_position) && of_lt(velocity, 0.)) {
        velocity = 0.;
    }
    /* mountain_car.rs:422-423 */
    int done = of_ge(position, p->goal_position) && of_ge(velocity, p->goal_velocity);
    double reward = -1.0;
    /* mountain_car.rs:425 */
    e->position = position;
    e->velocity = velocity;
    /* mountain_car.rs:428-434 */
    out->obs[0] = position; out->obs[1] = velocity; out->obs[2] = 0.; out->obs[3] = 0.;
    out->reward = reward;
    out->done = done;
    out->truncated = 0;
    return 0;
}

void orc_mountain_car_reset(orc_mountain_car_env *e, uint64_t seed, uint64_t gid, uint64_t tick,
                            const double *b)
{
    /* mountain_car.rs:175-187 : position ~ U[-0.6, -0.4) */
    static const double dflt[2] = {-0.6, -0.4};
    if (!b) b = dflt;
    uint32_t r[4];
    draw4(seed, gid, tick, 0, r);
    /* mountain_car.rs:162-167 : one draw for position; velocity is exactly 0 */
    e->position = uniform_between(r[0], b[0], b[1]);
    e->velocity = 0.;
}

/* ------------------------------------------------------------------------------------ */
/* Pendulum — spec-derived from the public Gym Pendulum-v1 definition. */
/* NOT in the reference.  Parity unpinned.                                                */

void orc_pendulum_default_params(orc_pendulum_params *p)
{
    p->max_speed = 8.;
    p->max_torque = 2.;
    p->dt = 0.05;
    p->g = 10.0;
    p->m = 1.;
    p->l = 1.;
}

static double angle_normalize(double x)
{
    const double pi = 3.14159265358979323846264338327950288;
    double y = x + pi;
    double m = y - (2. * pi) * floor(y / (2. * pi)); /* floored modulo */
    return m - pi;
}

int orc_pendulum_step(orc_pendulum_env *e, const orc_pendulum_params *p, double action,
                      orc_step_result *out)
{
    double th = e->theta, thdot = e->theta_dot;
    double u = orc_clip(action, -p->max_torque, p->max_torque);
    double an = angle_normalize(th);
    double costs = an * an + 0.1 * (thdot * thdot) + 0.001 * (u * u);
    double newthdot = thdot + (3. * p->g / (2. * p->l) * sin(th) + 3. / (p->m * (p->l * p->l)) * u) * p->dt;
    newthdot = orc_clip(newthdot, -p->max_speed, p->max_speed);
    double newth = th + newthdot * p->dt;
    e->theta = newth;
    e->theta_dot = newthdot;
    out->obs[0] = cos(newth); out->obs[1] = sin(newth); out->obs[2] = newthdot; out->obs[3] = 0.;
    out->reward = -costs;
    out->done = 0;
    out->truncated = 0;
    return 0;
}

void orc_pendulum_reset(orc_pendulum_env *e, uint64_t seed, uint64_t gid, uint64_t tick,
                        const double *b)
{
    const double pi = 3.14159265358979323846264338327950288;
    double dflt[4] = {-pi, -1., pi, 1.};
    if (!b) b = dflt;
    uint32_t r[4];
    draw4(seed, gid, tick, 0, r);
    e->theta = uniform_between(r[0], b[0], b[2]);
    e->theta_dot = uniform_between(r[1], b[1], b[3]);
}

/* ------------------------------------------------------------------------------------ */
/* batch drivers                                                                         */

long orc_cartpole_step_batch(size_t n, double *x, double *x_dot, double *theta, double *theta_dot,
                             uint8_t *beyond, const uint8_t *action, const orc_cartpole_params *p,
                             double *reward, uint8_t *done)
{
    long bad = 0;
    for (size_t i = 0; i < n; ++i) {
        orc_cartpole_env e = {x[i], x_dot[i], theta[i], theta_dot[i], beyond ? beyond[i] : 0, 0};
        orc_step_result r;
        if (orc_cartpole_step(&e, p, action[i], &r) != 0) {
            ++bad;
            reward[i] = 0.;
            done[i] = 0;
            continue;
        }
        x[i] = e.x; x_dot[i] = e.x_dot; theta[i] = e.theta; theta_dot[i] = e.theta_dot;
        if (beyond) beyond[i] = (uint8_t)e.has_steps_beyond;
        reward[i] = r.reward;
        done[i] = (uint8_t)r.done;
    }
    return bad;
}

long orc_mountain_car_step_batch(size_t n, double *position, double *velocity, const uint8_t *action,
                                 const orc_mountain_car_params *p, double *reward, uint8_t *done)
{
    long bad = 0;
    for (size_t i = 0; i < n; ++i) {
        orc_mountain_car_env e = {position[i], velocity[i]};
        orc_step_result r;
        if (orc_mountain_car_step(&e, p, action[i], &r) != 0) {
            ++bad;
            reward[i] = 0.;
            done[i] = 0;
            continue;
        }
        position[i] = e.position; velocity[i] = e.velocity;
        reward[i] = r.reward;
        done[i] = (uint8_t)r.done;
    }
    return bad;
}

long orc_pendulum_step_batch(size_t n, double *theta, double *theta_dot, const double *action,
                             const orc_pendulum_params *p, double *obs_cos, double *obs_sin,
                             double *reward)
{
    for (size_t i = 0; i < n; ++i) {
        orc_pendulum_env e = {theta[i], theta_dot[i]};
        orc_step_result r;
        orc_pendulum_step(&e, p, action[i], &r);
        theta[i] = e.theta; theta_dot[i] = e.theta_dot;
        obs_cos[i] = r.obs[0]; obs_sin[i] = r.obs[1];
        reward[i] = r.reward;
    }
    return 0;
}

void orc_cartpole_reset_batch(size_t n, uint64_t gid0, uint64_t seed, uint64_t tick,
                              const double *b, double *x, double *x_dot, double *theta,
                              double *theta_dot)
{
    for (size_t i = 0; i < n; ++i) {
        orc_cartpole_env e;
        orc_cartpole_reset(&e, seed, gid0 + i, tick, b);
        x[i] = e.x; x_dot[i] = e.x_dot; theta[i] = e.theta; theta_dot[i] = e.theta_dot;
    }
}

void orc_mountain_car_reset_batch(size_t n, uint64_t gid0, uint64_t seed, uint64_t tick,
                                  const double *b, double *position, double *velocity)
{
    for (size_t i = 0; i < n; ++i) {
        orc_mountain_car_env e;
        orc_mountain_car_reset(&e, seed, gid0 + i, tick, b);
        position[i] = e.position; velocity[i] = e.velocity;
    }
}

void orc_pendulum_reset_batch(size_t n, uint64_t gid0, uint64_t seed, uint64_t tick,
                              const double *b, double *theta, double *theta_dot)
{
    for (size_t i = 0; i < n; ++i) {
        orc_pendulum_env e;
        orc_pendulum_reset(&e, seed, gid0 + i, tick, b);
        theta[i] = e.theta; theta_dot[i] = e.theta_dot;
    }
}

/* ------------------------------------------------------------------------------------ */
/* The reference's caller loop (examples/cartpole.rs:15-30, examples/mountain_car.rs:15-24),
 * RenderMode::None semantics (renderer.rs:40-49: render_step is a no-op), one env, one
 * thread.  The action source is a cheap xorshift so the RNG does not dominate the timing
 * (the example uses rand::thread_rng, which is also a fast generator).                  */

static inline uint64_t xorshift64(uint64_t *s)
{
    uint64_t x = *s;
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return *s = x;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double orc_baseline_loop(int kind, uint64_t n_steps, uint64_t max_episode_steps, uint64_t seed,
                         double out[4])
{
    uint64_t rng = 0x9E3779B97F4A7C15ull ^ (seed * 0xD1342543DE82EF95ull + 1);
    double sum_return = 0., sum_length = 0., n_episodes = 0.;
    uint64_t done_steps = 0, episode = 0;
    double t0 = now_s();
    if (kind == 0) {
        orc_cartpole_params p;
        orc_cartpole_default_params(&p);
        orc_cartpole_env e;
        orc_cartpole_reset(&e, seed, 0, episode++, NULL);
        while (done_steps < n_steps) {
            double ret = 0.;
            uint64_t t = 0;
            for (;;) {
                size_t a = (size_t)(xorshift64(&rng) >> 63);
                orc_step_result r;
                orc_cartpole_step(&e, &p, a, &r);
                ret += r.reward;
                ++t;
                ++done_steps;
                if (r.done || (max_episode_steps && t >= max_episode_steps) || done_steps >= n_steps) break;
            }
            orc_cartpole_reset(&e, seed, 0, episode++, NULL);
            sum_return += ret; sum_length += (double)t; n_episodes += 1.;
        }
    } else if (kind == 1) {
        orc_mountain_car_params p;
        orc_mountain_car_default_params(&p);
        orc_mountain_car_env e;
        orc_mountain_car_reset(&e, seed, 0, episode++, NULL);
        while (done_steps < n_steps) {
            double ret = 0.;
            uint64_t t = 0;
            for (;;) {
                size_t a = (size_t)((xorshift64(&rng) >> 33) % 3u);
                orc_step_result r;
                orc_mountain_car_step(&e, &p, a, &r);
                ret += r.reward;
                ++t;
                ++done_steps;
                if (r.done || (max_episode_steps && t >= max_episode_steps) || done_steps >= n_steps) break;
            }
            orc_mountain_car_reset(&e, seed, 0, episode++, NULL);
            sum_return += ret; sum_length += (double)t; n_episodes += 1.;
        }
    } else {
        orc_pendulum_params p;
        orc_pendulum_default_params(&p);
        orc_pendulum_env e;
        orc_pendulum_reset(&e, seed, 0, episode++, NULL);
        if (!max_episode_steps) max_episode_steps = 200;
        while (done_steps < n_steps) {
            double ret = 0.;
            uint64_t t = 0;
            for (;;) {
                double a = ((double)(xorshift64(&rng) >> 11) * (1.0 / 9007199254740992.0)) * 4. - 2.;
                orc_step_result r;
                orc_pendulum_step(&e, &p, a, &r);
                ret += r.reward;
                ++t;
                ++done_steps;
                if (t >= max_episode_steps || done_steps >= n_steps) break;
            }
            orc_pendulum_reset(&e, seed, 0, episode++, NULL);
            sum_return += ret; sum_length += (double)t; n_episodes += 1.;
        }
    }
    double t1 = now_s();
    out[0] = sum_return; out[1] = sum_length; out[2] = n_episodes; out[3] = (double)done_steps;
    return t1 - t0;
}
