// gymrs_physics.h — per-lane f32 physics of the gym-rs classic-control envs, shared by the HIP
// kernels (device) and the host (CPU f32 twin used by the tests for bit-exact comparison).
//
// Restates, in f32 and per lane:
//   CartPoleEnv::step      /root/reference/src/envs/classical_control/cartpole.rs:398-483
//   MountainCarEnv::step   /root/reference/src/envs/classical_control/mountain_car.rs:398-435
//   clip                   /root/reference/src/utils/custom/util_fns.rs:2-10
//   reset sampling order   cartpole.rs:317-324,352-364 ; mountain_car.rs:162-167,175-190
//   Pendulum: spec-derived (Gym Pendulum-v1); NOT in the reference.
//
// Arithmetic contract (what "the f32 twin" means): IEEE f32, round-to-nearest, denormals kept,
// -ffp-contract=off on both compilers; a*b+c is fused ONLY where fmaf_ is written.  Divisions are
// IEEE-correct on both sides (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt).
// sin/cos come from gymrs_math.h.  With that, gfx950 and x86 produce identical bits.
#pragma once
#include "gymrs_amd.h"
#include "gymrs_math.h"
#include "gymrs_philox.h"

namespace gymrs {

// ---------------------------------------------------------------------------------------------
// clip (util_fns.rs:2-10) with OrderedFloat's total order (NaN is the maximum):
//   l <= v && v <= r -> v ; v > r -> r ; else l.   NaN fails `v <= r` and is `> r`, so -> r.
GYMRS_HD float clipf(float v, float l, float r)
{
    if (!(v <= r)) return r;
    if (!(v >= l)) return l;
    return v;
}

// Lane-uniform sub-expressions of the CartPole update are folded on the host, in f64, and rounded to f32 ONCE
// (make_consts): the three divisions by total_mass, polemass_length / total_mass, length * 4/3 and
// length * masspole / total_mass.  The step kernel is VALU-issue-bound once its loads have landed, so every
// instruction removed from the per-lane chain is time: 14 f32 operations + one IEEE division per lane-step instead of 19 +
// one (and of the reference's 3 divisions and 2 powf).  Each folded constant is within half an ulp of the f64 value, each
// remaining operation rounds once: the state stays <= 3e-7 relative from the f64 oracle's (budget 1e-6, measured in
// tests/test_twin_vs_oracle.py and on the GPU), and gfx950 and x86 execute the identical operations, so the GPU stays
// bit-identical to its CPU twin.  The f64 oracle (oracle/gymrs_oracle.c) keeps the reference's exact operation order.

// Thresholds the reference compares an f64 state against (cartpole.rs:450-453 `x > x_threshold`, mountain_car.rs:422
// `position >= goal_position`), restated for an f32 state so that the FLAG is exact: for every f32 v,
//   v > t  (f64)  <=>  v > f32_not_above(t)        v >= t  (f64)  <=>  v >= f32_not_below(t)
// (round-to-nearest would put fl32(2.4) = 2.4000001 above 2.4 and keep a lane sitting on 2.4000001 alive where the
// reference ends it).  With these, done/terminated on identical f32 inputs equals the reference's f64 compare bit for bit;
// only the f32-vs-f64 error of the STATE remains.
inline float f32_not_above(double t)
{
    float f = (float)t;
    if ((double)f > t) f = __builtin_nextafterf(f, -__builtin_inff());
    return f;
}
inline float f32_not_below(double t)
{
    float f = (float)t;
    if ((double)f < t) f = __builtin_nextafterf(f, __builtin_inff());
    return f;
}

// ---------------------------------------------------------------------------------------------
// CartPole
struct CartPoleConsts {
    float gravity, tau;
    float force_over_tm;   // force_mag / total_mass                      cartpole.rs:414-418, 423-424
    float pml_over_tm;     // polemass_length / total_mass, with polemass_length = masspole + length (sic, Q1: cartpole.rs:150-152)
                           // and total_mass = masspole + masscart (cartpole.rs:146-148)
    float len_43;          // length * (4.0/3.0)                          cartpole.rs:427
    float len_mp_over_tm;  // length * masspole / total_mass              cartpole.rs:427-428
    float theta_thr, x_thr; // largest f32 <= the f64 thresholds (f32_not_above): `|v| > thr` is then the reference's compare
    int32_t integrator;    // 0 Euler, 1 Other             cartpole.rs:380-387
    uint32_t max_steps;
};

inline CartPoleConsts make_consts(const gymrs_cartpole_params& p)
{
    CartPoleConsts c;
    const double total_mass = p.masspole + p.masscart;
    const double polemass_length = p.masspole + p.length; // the reference ADDS; do not "fix"
    c.gravity = (float)p.gravity;
    c.tau = (float)p.tau;
    c.force_over_tm = (float)(p.force_mag / total_mass);
    c.pml_over_tm = (float)(polemass_length / total_mass);
    c.len_43 = (float)(p.length * (4.0 / 3.0));
    c.len_mp_over_tm = (float)(p.length * p.masspole / total_mass);
    c.theta_thr = f32_not_above(p.theta_threshold_radians); // cartpole.rs:450-453: strict f64 compares of an f32 state
    c.x_thr = f32_not_above(p.x_threshold);
    c.integrator = p.kinematics_integrator;
    c.max_steps = p.max_episode_steps ? p.max_episode_steps : 500u;
    return c;
}

// One step of the dynamics + termination test.  Returns done.  (cartpole.rs:408-453)
// SC selects how sin/cos are evaluated (gymrs_math.h); every choice gives the same bits on its domain.
// INTEG: -1 = read c.integrator at run time, 0 = Euler, 1 = semi-implicit (lets a kernel hoist the choice).
template <class SC = SinCosGeneral, int INTEG = -1>
GYMRS_HD bool cartpole_advance(const CartPoleConsts& c, float& x, float& x_dot, float& theta, float& theta_dot,
                               uint32_t action)
{
    const float force = (action == 1u) ? c.force_over_tm : -c.force_over_tm; // :414-418, already / total_mass
    float sintheta, costheta;
    SC::eval(theta, &sintheta, &costheta); // :420-421
    // :423-424  temp = (force + polemass_length * theta_dot^2 * sintheta) / total_mass
    const float temp = fmaf_(c.pml_over_tm, (theta_dot * theta_dot) * sintheta, force);
    // :425-428  thetaacc = (g*sin - cos*temp) / (length * (4/3 - masspole*cos^2/total_mass))
    const float num = fmaf_(-costheta, temp, c.gravity * sintheta);
    const float den = fmaf_(-c.len_mp_over_tm, costheta * costheta, c.len_43);
    const float thetaacc = num / den;
    // :429  xacc = temp - polemass_length * thetaacc * costheta / total_mass
    const float xacc = fmaf_(-c.pml_over_tm, thetaacc * costheta, temp);
    if (INTEG == 0 || (INTEG < 0 && c.integrator == 0)) { // :431-435 Euler: x and theta advance with the OLD velocities
        x = fmaf_(c.tau, x_dot, x);
        x_dot = fmaf_(c.tau, xacc, x_dot);
        theta = fmaf_(c.tau, theta_dot, theta);
        theta_dot = fmaf_(c.tau, thetaacc, theta_dot);
    } else { // :436-441 semi-implicit
        x_dot = fmaf_(c.tau, xacc, x_dot);
        x = fmaf_(c.tau, x_dot, x);
        theta_dot = fmaf_(c.tau, thetaacc, theta_dot);
        theta = fmaf_(c.tau, theta_dot, theta);
    }
    // :450-453 strict compares; a NaN counts as "> threshold" in OrderedFloat's order (Q10)
    return !(fabsf_(x) <= c.x_thr) || !(fabsf_(theta) <= c.theta_thr);
}

// :455-464 — reward with steps_beyond_terminated (`beyond` = is_some()).
GYMRS_HD float cartpole_reward(bool done, bool& beyond)
{
    if (!done) return 1.0f;
    if (!beyond) {
        beyond = true;
        return 1.0f;
    }
    return 0.0f;
}

// cartpole.rs:317-324: x, x_dot, theta, theta_dot in this order, each on [low, high).
GYMRS_HD void cartpole_sample(const u32x4& r, const SampleBox& b, float& x, float& x_dot, float& theta, float& theta_dot)
{
    x = uniform_in_box(r.v[0], b, 0);
    x_dot = uniform_in_box(r.v[1], b, 1);
    theta = uniform_in_box(r.v[2], b, 2);
    theta_dot = uniform_in_box(r.v[3], b, 3);
}

// ---------------------------------------------------------------------------------------------
// MountainCar
struct MountainCarConsts {
    float min_position, max_position, max_speed, goal_position, goal_velocity, force, gravity;
    uint32_t max_steps;
};

inline MountainCarConsts make_consts(const gymrs_mountain_car_params& p)
{
    MountainCarConsts c;
    c.min_position = (float)p.min_position;
    c.max_position = (float)p.max_position;
    c.max_speed = (float)p.max_speed;
    c.goal_position = f32_not_below(p.goal_position); // mountain_car.rs:422: inclusive f64 compares of an f32 state
    c.goal_velocity = f32_not_below(p.goal_velocity);
    c.force = (float)p.force;
    c.gravity = (float)p.gravity;
    c.max_steps = p.max_episode_steps ? p.max_episode_steps : 200u;
    return c;
}

// mountain_car.rs:408-423.  Returns done.
template <class SC = SinCosGeneral>
GYMRS_HD bool mountain_car_advance(const MountainCarConsts& c, float& position, float& velocity, uint32_t action)
{
    // :411-412  velocity += (action - 1) * force + cos(3 * position) * (-gravity)
    const float push = ((float)action - 1.0f) * c.force;
    float sin3p, cos3p;
    SC::eval(3.0f * position, &sin3p, &cos3p);
    velocity = velocity + fmaf_(cos3p, -c.gravity, push);
    velocity = clipf(velocity, -c.max_speed, c.max_speed); // :413
    position = position + velocity;                        // :415
    position = clipf(position, c.min_position, c.max_position); // :416
    // :418-420 exact equality with the clip bound (Q11)
    if (position == c.min_position && velocity < 0.0f) velocity = 0.0f;
    // :422 inclusive; NaN >= anything in OrderedFloat's order
    return !(position < c.goal_position) && !(velocity < c.goal_velocity);
}

// mountain_car.rs:162-167: one draw for position, velocity exactly 0.
GYMRS_HD void mountain_car_sample(const u32x4& r, const SampleBox& b, float& position, float& velocity)
{
    position = uniform_in_box(r.v[0], b, 0);
    velocity = 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Pendulum (spec-derived; not in the reference)
struct PendulumConsts {
    float max_speed, max_torque, dt;
    float c_sin; // 3*g/(2*l)
    float c_u;   // 3/(m*l^2)
    uint32_t max_steps;
};

inline PendulumConsts make_consts(const gymrs_pendulum_params& p)
{
    PendulumConsts c;
    c.max_speed = (float)p.max_speed;
    c.max_torque = (float)p.max_torque;
    c.dt = (float)p.dt;
    c.c_sin = (float)(3. * p.g / (2. * p.l));
    c.c_u = (float)(3. / (p.m * (p.l * p.l)));
    c.max_steps = p.max_episode_steps ? p.max_episode_steps : 200u;
    return c;
}

// ((x + pi) mod 2pi) - pi with a floored modulo.  |x| <= 200: f32, x - q*(T1 + T2) in two fma (the first
// forms x - q*T1 exactly and rounds once: absolute error <= 2.5e-7 on a value up to pi, i.e. <= 1.6e-6 on a
// cost of ~10).  Which side of the wrap point a boundary value lands on does not matter: the cost uses the
// square, which is continuous there.  Beyond 200 rad the same in f64.
GYMRS_HD float angle_normalize(float x)
{
    if (in_short_range(x)) {
        const float pi = 0x1.921fb6p+1f, inv_two_pi = 0x1.45f306p-3f, T1 = 0x1.921fb6p+2f, T2 = -0x1.777a5cp-23f;
        const float q = __builtin_floorf((x + pi) * inv_two_pi);
        float m = fmaf_(-q, T1, x);
        m = fmaf_(-q, T2, m);
        return m;
    }
    const double pi = 0x1.921fb54442d18p+1, two_pi = 0x1.921fb54442d18p+2, inv_two_pi = 0x1.45f306dc9c883p-3;
    double y = (double)x + pi;
    double q = __builtin_floor(y * inv_two_pi);
    double m = fma_(-q, two_pi, y);
    return (float)(m - pi);
}

// Returns the reward (-cost, from the OLD state); never terminates.
template <class SC = SinCosGeneral>
GYMRS_HD float pendulum_advance(const PendulumConsts& c, float& theta, float& theta_dot, float action)
{
    const float u = clipf(action, -c.max_torque, c.max_torque);
    const float an = angle_normalize(theta);
    const float cost = fmaf_(0.001f, u * u, fmaf_(0.1f, theta_dot * theta_dot, an * an));
    float sin_th, cos_th;
    SC::eval(theta, &sin_th, &cos_th);
    const float acc = fmaf_(c.c_sin, sin_th, c.c_u * u);
    float nthd = fmaf_(acc, c.dt, theta_dot);
    nthd = clipf(nthd, -c.max_speed, c.max_speed);
    theta = fmaf_(nthd, c.dt, theta); // semi-implicit: uses the NEW theta_dot
    theta_dot = nthd;
    return -cost;
}

GYMRS_HD void pendulum_sample(const u32x4& r, const SampleBox& b, float& theta, float& theta_dot)
{
    theta = uniform_in_box(r.v[0], b, 0);
    theta_dot = uniform_in_box(r.v[1], b, 1);
}

// Default reset boxes (obs_dim lows then highs in the API; here split).
constexpr float kPiF = 3.14159274101257324f; // fl32(pi); the box is [-fl32(pi), fl32(pi))

} // namespace gymrs
