// gymrs_kernels.hip — hand-written CDNA4 (gfx950) kernels of the batched classic-control stepper.
//
// One kernel template, instantiated per env type: each lane is one independent gym-rs env
//   CartPoleEnv::step     /root/reference/src/envs/classical_control/cartpole.rs:398-483
//   MountainCarEnv::step  /root/reference/src/envs/classical_control/mountain_car.rs:398-435
//   (+ reset, cartpole.rs:485-516 / mountain_car.rs:464-501, fused in for finished lanes)
//
// Shape of the work: pure streaming, HBM/L2-bound, no contraction -> no MFMA.
//   * SoA f32 state in HBM; a work-item owns VEC consecutive lanes and moves them with one
//     dwordx4/x2 load + store per array (16 B/lane-group, 1 KiB per wave instruction), actions and
//     done flags as packed bytes.
//   * 256-thread workgroups (4 wave64); workgroup b always owns lanes [b*256*VEC, (b+1)*256*VEC), so
//     the same lanes come back to the same XCD's L2 every step (blocks are dealt round-robin to
//     XCDs): the state a step wrote is what the next step reads from L2, not from HBM.
//   * auto-reset: per wave a __ballot done-mask; quiet waves skip everything.  Finished lanes are
//     compacted through the wave's own LDS slice (ballot + mbcnt ranks) so that ONE Philox4x32-10
//     evaluation is spent per finished lane instead of one per wave-lane under divergence; the fresh
//     states come back through LDS to the owning work-item, which stores them with its vector store.
//     Everything stays inside one wavefront: no s_barrier on the path.
//   * episode statistics cost the step nothing it can wait on: a finished lane only WRITES the tick
//     its next episode starts at (episodes tile a lane's time axis, so the sum of finished lengths is
//     sum(ep_start) - n*epoch, evaluated when statistics are read), plus one fire-and-forget atomic
//     per wave on a per-workgroup counter.
#include <type_traits>

#include "gymrs_kernels.h"

namespace gymrs {

// ---------------------------------------------------------------------------------------------
// vector access helpers
template <class T, int V>
struct alignas(sizeof(T) * V) Vec {
    T v[V];
};

template <class T, int V>
__device__ __forceinline__ Vec<T, V> load_vec(const T* __restrict__ p, uint64_t base, uint64_t n, bool full, T fill)
{
    Vec<T, V> r;
    if (full) {
        r = *reinterpret_cast<const Vec<T, V>*>(p + base);
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k) r.v[k] = (base + k < n) ? p[base + k] : fill;
    }
    return r;
}

template <class T, int V>
__device__ __forceinline__ void store_vec(T* __restrict__ p, uint64_t base, uint64_t n, bool full, const Vec<T, V>& r)
{
    if (full) {
        *reinterpret_cast<Vec<T, V>*>(p + base) = r;
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k)
            if (base + k < n) p[base + k] = r.v[k];
    }
}

// ---------------------------------------------------------------------------------------------
// Packed-f32 form of cartpole_advance<SinCosSmall, INTEG> (gymrs_physics.h) for TWO lanes at once.
// CDNA's VALU issues a wave64 f32 op in 4 cycles whether it carries one value per lane or a packed
// pair (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32), so pairing the lanes of a work-item halves the
// arithmetic issue time.  Every packed op is the same IEEE operation per component, in the same
// order, as the scalar code: results are bit-identical (the GPU parity tests compare against the
// scalar CPU twin).
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_splat(float v) { return f32x2{v, v}; }

__device__ __forceinline__ f32x2 pk_div_by_uniform(f32x2 a, float b, float rb)
{
    const f32x2 q = a * pk_splat(rb);
    const f32x2 e = pk_fma(pk_splat(-b), q, a);
    return pk_fma(e, pk_splat(rb), q);
}

__device__ __forceinline__ void pk_sincos_poly(f32x2 r, f32x2* s, f32x2* c)
{
    const float S1 = -0.166666641831398f, S2 = 0.008332724682986736f, S3 = -0.0001958291686605662f;
    const float C1 = 0.04166654869914055f, C2 = -0.0013880875194445252f, C3 = 2.360151120228693e-05f;
    const f32x2 z = r * r;
    f32x2 ps = pk_fma(z, pk_splat(S3), pk_splat(S2));
    ps = pk_fma(z, ps, pk_splat(S1));
    f32x2 pc = pk_fma(z, pk_splat(C3), pk_splat(C2));
    pc = pk_fma(z, pc, pk_splat(C1));
    const f32x2 rz = r * z;
    const f32x2 w = z * z;
    *s = pk_fma(rz, ps, r);
    *c = pk_fma(w, pc, pk_fma(z, pk_splat(-0.5f), pk_splat(1.0f)));
}

template <int INTEG>
__device__ __forceinline__ void cartpole_advance_x2(const CartPoleConsts& c, f32x2& x, f32x2& x_dot, f32x2& theta,
                                                    f32x2& theta_dot, uint32_t a0, uint32_t a1, bool& d0, bool& d1)
{
    const f32x2 force = {a0 == 1u ? c.force_mag : -c.force_mag, a1 == 1u ? c.force_mag : -c.force_mag};
    f32x2 sintheta, costheta;
    pk_sincos_poly(theta, &sintheta, &costheta);
    const f32x2 temp = pk_div_by_uniform(pk_fma(pk_splat(c.polemass_length) * (theta_dot * theta_dot), sintheta, force),
                                         c.total_mass, c.inv_total_mass);
    const f32x2 num = pk_fma(-costheta, temp, pk_splat(c.gravity) * sintheta);
    const f32x2 den = pk_splat(c.length) *
                      (pk_splat(c.four_thirds) - pk_div_by_uniform(pk_splat(c.masspole) * (costheta * costheta), c.total_mass, c.inv_total_mass));
    const f32x2 thetaacc = {num.x / den.x, num.y / den.y};
    const f32x2 xacc = temp - pk_div_by_uniform((pk_splat(c.polemass_length) * thetaacc) * costheta, c.total_mass, c.inv_total_mass);
    const f32x2 tau = pk_splat(c.tau);
    if (INTEG == 0) {
        x = pk_fma(tau, x_dot, x);
        x_dot = pk_fma(tau, xacc, x_dot);
        theta = pk_fma(tau, theta_dot, theta);
        theta_dot = pk_fma(tau, thetaacc, theta_dot);
    } else {
        x_dot = pk_fma(tau, xacc, x_dot);
        x = pk_fma(tau, x_dot, x);
        theta_dot = pk_fma(tau, thetaacc, theta_dot);
        theta = pk_fma(tau, theta_dot, theta);
    }
    d0 = !(fabsf_(x.x) <= c.x_thr) || !(fabsf_(theta.x) <= c.theta_thr);
    d1 = !(fabsf_(x.y) <= c.x_thr) || !(fabsf_(theta.y) <= c.theta_thr);
}

// ---------------------------------------------------------------------------------------------
// Env policies: what differs between the three env types.
struct CartPoleT {
    using Consts = CartPoleConsts;
    using Action = uint8_t;
    static constexpr int kState = 4;
    static constexpr bool kConstReward = true; // under auto-reset every step pays 1.0 (cartpole.rs:455-459)
    static constexpr float kRewardValue = 1.0f;
    static constexpr bool kHasBeyond = true;
    static constexpr bool kHasObsExtra = false;
    __device__ static bool valid(Action a) { return a < 2; } // Discrete(2).contains, discrete.rs:14-19
    __device__ static void advance(const Consts& c, float* st, Action a, float& reward, bool& done)
    {
        done = cartpole_advance(c, st[0], st[1], st[2], st[3], a);
        reward = 1.0f; // the beyond-terminated case is applied by the caller when auto-reset is off
    }
    // branch-free variant, legal when fast_ok holds (|theta| <= pi/4: the polynomial needs no reduction)
    __device__ static bool fast_ok(const float* st, Action a) { return a < 2 && in_small_range(st[2]); }
    template <int INTEG>
    __device__ static void advance_fast(const Consts& c, float* st, Action a, float& reward, bool& done)
    {
        done = cartpole_advance<SinCosSmall, INTEG>(c, st[0], st[1], st[2], st[3], a);
        reward = 1.0f;
    }
    static constexpr int kVariants = 2; // the integrator choice is hoisted out of the per-lane code
    __device__ static int variant(const Consts& c) { return c.integrator == 0 ? 0 : 1; }
    static constexpr bool kHasPacked = true; // two lanes per packed-f32 op
    template <int INTEG>
    __device__ static void advance_fast_x2(const Consts& c, f32x2* st, Action a0, Action a1, f32x2& reward, bool& d0, bool& d1)
    {
        cartpole_advance_x2<INTEG>(c, st[0], st[1], st[2], st[3], a0, a1, d0, d1);
        reward = pk_splat(1.0f);
    }
    __device__ static void obs_extra(float, float*, float*) {}
    __device__ static void sample(const u32x4& r, const float* lo, const float* hi, float* st)
    {
        cartpole_sample(r, lo, hi, st[0], st[1], st[2], st[3]);
    }
};

struct MountainCarT {
    using Consts = MountainCarConsts;
    using Action = uint8_t;
    static constexpr int kState = 2;
    static constexpr bool kConstReward = true; // -1.0 on every step (mountain_car.rs:423)
    static constexpr float kRewardValue = -1.0f;
    static constexpr bool kHasBeyond = false;
    static constexpr bool kHasObsExtra = false;
    __device__ static bool valid(Action a) { return a < 3; } // Discrete(3)
    __device__ static void advance(const Consts& c, float* st, Action a, float& reward, bool& done)
    {
        done = mountain_car_advance(c, st[0], st[1], a);
        reward = -1.0f;
    }
    __device__ static bool fast_ok(const float* st, Action a) { return a < 3 && in_medium_range(3.0f * st[0]); }
    template <int>
    __device__ static void advance_fast(const Consts& c, float* st, Action a, float& reward, bool& done)
    {
        done = mountain_car_advance<SinCosMedium>(c, st[0], st[1], a);
        reward = -1.0f;
    }
    static constexpr int kVariants = 1;
    __device__ static int variant(const Consts&) { return 0; }
    static constexpr bool kHasPacked = false;
    template <int>
    __device__ static void advance_fast_x2(const Consts&, f32x2*, Action, Action, f32x2&, bool&, bool&) {}
    __device__ static void obs_extra(float, float*, float*) {}
    __device__ static void sample(const u32x4& r, const float* lo, const float* hi, float* st)
    {
        mountain_car_sample(r, lo, hi, st[0], st[1]);
    }
};

struct PendulumT { // spec-derived, not in the reference
    using Consts = PendulumConsts;
    using Action = float;
    static constexpr int kState = 2;
    static constexpr bool kConstReward = false;
    static constexpr float kRewardValue = 0.0f;
    static constexpr bool kHasBeyond = false;
    static constexpr bool kHasObsExtra = true;
    __device__ static bool valid(Action) { return true; } // a Box action is clipped, never rejected
    __device__ static void advance(const Consts& c, float* st, Action a, float& reward, bool& done)
    {
        reward = pendulum_advance(c, st[0], st[1], a);
        done = false;
    }
    // one step moves theta by at most max_speed*dt, so the new theta (observation) stays in range too
    __device__ static bool fast_ok(const float* st, Action) { return (f2u(st[0]) & 0x7fffffffu) < 0x4d000000u; } // |theta| < 2^27
    template <int>
    __device__ static void advance_fast(const Consts& c, float* st, Action a, float& reward, bool& done)
    {
        reward = pendulum_advance<SinCosMedium>(c, st[0], st[1], a);
        done = false;
    }
    static constexpr int kVariants = 1;
    __device__ static int variant(const Consts&) { return 0; }
    static constexpr bool kHasPacked = false;
    template <int>
    __device__ static void advance_fast_x2(const Consts&, f32x2*, Action, Action, f32x2&, bool&, bool&) {}
    __device__ static void sample(const u32x4& r, const float* lo, const float* hi, float* st)
    {
        pendulum_sample(r, lo, hi, st[0], st[1]);
    }
};

// ---------------------------------------------------------------------------------------------
// THE hot kernel: one Env::step() per lane, VEC lanes per work-item, 64*VEC lanes per wavefront,
// TILES tiles of 256*VEC lanes per workgroup.
//
// There is no workgroup barrier anywhere: a wavefront owns its lanes from load to store, so the
// 4 waves of a workgroup (and the workgroups of a CU) drift apart and one wave's arithmetic hides
// under another wave's memory traffic.  With TILES > 1 a wave issues the loads of all its tiles up
// front and works through them in arrival order (vmcnt is in-order), which spreads the arithmetic
// over the read phase instead of leaving it all behind the last load.
template <class Env, int VEC, uint32_t FLAGS>
struct TileRegs {
    static constexpr bool AUTO = (FLAGS & GYMRS_AUTO_RESET) != 0;
    static constexpr bool STATS = AUTO && (FLAGS & GYMRS_TRACK_STATS) != 0;
    static constexpr bool TLIM = (FLAGS & GYMRS_TIME_LIMIT) != 0;
    Vec<float, VEC> st[Env::kState];
    Vec<typename Env::Action, VEC> act;
    Vec<uint8_t, VEC> beyond;
    Vec<uint32_t, VEC> ep_start;
    Vec<float, VEC> ep_ret;
};

template <class Env, int VEC, uint32_t FLAGS, bool FULL>
__device__ __forceinline__ void load_tile(const StepArgs& a, uint64_t base, TileRegs<Env, VEC, FLAGS>& d)
{
    using R = TileRegs<Env, VEC, FLAGS>;
    using Action = typename Env::Action;
#pragma unroll
    for (int j = 0; j < Env::kState; ++j) d.st[j] = load_vec<float, VEC>(a.s[j], base, a.n, FULL, 0.0f);
    d.act = load_vec<Action, VEC>(static_cast<const Action*>(a.action), base, a.n, FULL, Action(0));
    if (Env::kHasBeyond && !R::AUTO) d.beyond = load_vec<uint8_t, VEC>(a.beyond, base, a.n, FULL, uint8_t(0));
    if (R::TLIM) d.ep_start = load_vec<uint32_t, VEC>(a.ep_start, base, a.n, FULL, 0u);
    if (R::STATS && !Env::kConstReward) d.ep_ret = load_vec<float, VEC>(a.ep_ret, base, a.n, FULL, 0.0f);
}

// The branch-free physics of all VEC lanes of a work-item in one basic block (V = Env variant).
template <class Env, int VEC, int V>
__device__ __forceinline__ void advance_fast_all(const typename Env::Consts& c, float (&ls)[Env::kState][VEC],
                                                 const typename Env::Action (&la)[VEC], float (&rw)[VEC], bool (&dn)[VEC])
{
    constexpr int NS = Env::kState;
    if constexpr (Env::kHasPacked && VEC % 2 == 0) {
#pragma unroll
        for (int k = 0; k < VEC; k += 2) {
            f32x2 pst[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) pst[j] = f32x2{ls[j][k], ls[j][k + 1]};
            f32x2 r;
            Env::template advance_fast_x2<V>(c, pst, la[k], la[k + 1], r, dn[k], dn[k + 1]);
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                ls[j][k] = pst[j].x;
                ls[j][k + 1] = pst[j].y;
            }
            rw[k] = r.x;
            rw[k + 1] = r.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float lane_st[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) lane_st[j] = ls[j][k];
            Env::template advance_fast<V>(c, lane_st, la[k], rw[k], dn[k]);
#pragma unroll
            for (int j = 0; j < NS; ++j) ls[j][k] = lane_st[j];
        }
    }
}

// physics + auto-reset + stores of one tile whose loads were issued by load_tile
template <class Env, int VEC, uint32_t FLAGS, bool FULL>
__device__ __forceinline__ void finish_tile(const StepArgs& a, const typename Env::Consts& c, uint64_t tile_base,
                                            TileRegs<Env, VEC, FLAGS>& d, uint16_t* s_list, float* s_new,
                                            uint32_t& wave_resets, float& wave_ret)
{
    using R = TileRegs<Env, VEC, FLAGS>;
    constexpr bool AUTO = R::AUTO, STATS = R::STATS, TLIM = R::TLIM;
    constexpr int NS = Env::kState;
    constexpr int LPW = 64 * VEC; // lanes per wavefront
    using Action = typename Env::Action;

    const uint32_t tid = threadIdx.x;
    const uint64_t base = tile_base + (uint64_t)tid * VEC;
    const uint32_t tick_next = (uint32_t)(a.tick + 1);

    // ---- physics ----
    // State is unpacked into plain per-lane scalars (registers) for the arithmetic and re-packed
    // into vectors only for the stores.
    float ls[NS][VEC];
    Action la[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        la[k] = d.act.v[k];
#pragma unroll
        for (int j = 0; j < NS; ++j) ls[j][k] = d.st[j].v[k];
    }
    float rw[VEC];
    bool need_reset[VEC];
    bool dn[VEC], tr[VEC];
    // Wave-uniform choice: when every lane of the wavefront holds a valid action and an angle the
    // branch-free sin/cos covers (always, in practice), the VEC lanes of a work-item are advanced in
    // ONE basic block, so their independent dependency chains interleave and pack into v_pk_* ops.
    // Otherwise the general per-lane code runs.  Both produce the same bits.
    bool fast = FULL;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        float lane_st[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) lane_st[j] = ls[j][k];
        fast = fast && Env::fast_ok(lane_st, la[k]);
    }
    if (__all(fast)) {
        if (Env::kVariants == 1 || Env::variant(c) == 0)
            advance_fast_all<Env, VEC, 0>(c, ls, la, rw, dn);
        else
            advance_fast_all<Env, VEC, 1>(c, ls, la, rw, dn);
    } else {
        uint32_t n_bad = 0, first_bad = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const bool live = FULL || (base + k < a.n);
            const bool ok = Env::valid(la[k]);
            float lane_st[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) lane_st[j] = ls[j][k];
            float r = 0.0f;
            bool dk = false;
            if (live && ok) {
                Env::advance(c, lane_st, la[k], r, dk);
            } else if (live) { // invalid action: the reference panics before touching the env
                n_bad += 1;
                first_bad = min(first_bad, (uint32_t)(base + k));
            }
#pragma unroll
            for (int j = 0; j < NS; ++j) ls[j][k] = lane_st[j];
            rw[k] = r;
            dn[k] = dk;
        }
        if (n_bad) {
            atomicAdd(&a.err[0], n_bad);
            atomicMin(&a.err[1], first_bad);
        }
    }
    Vec<uint8_t, VEC> done, trunc;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const bool stepped = (FULL || (base + k < a.n)) && Env::valid(la[k]);
        if (Env::kHasBeyond && !AUTO) { // cartpole.rs:455-464
            bool b = d.beyond.v[k] != 0;
            const float r = cartpole_reward(dn[k], b);
            if (stepped) {
                rw[k] = r;
                d.beyond.v[k] = b ? 1 : 0;
            }
        }
        tr[k] = TLIM && stepped && (tick_next - d.ep_start.v[k]) >= c.max_steps;
        if (STATS && !Env::kConstReward) d.ep_ret.v[k] += rw[k];
        done.v[k] = dn[k] ? 1 : 0;
        trunc.v[k] = tr[k] ? 1 : 0;
        need_reset[k] = AUTO && (dn[k] || tr[k]);
    }

#ifdef GYMRS_EXP_CHEAP_RESET
    if (AUTO) {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            if (need_reset[k]) {
#pragma unroll
                for (int j = 0; j < NS; ++j) ls[j][k] = 0.01f * (j + 1);
            }
    }
    if (false) {
#else
    // ---- auto-reset: wave __ballot done-mask -> LDS-staged Philox, all inside one wavefront ----
    if (AUTO) {
#endif
        const uint32_t wave = tid >> 6, lane = tid & 63u;
        uint16_t* list = s_list + wave * LPW;  // this wave's compacted list of finished lanes
        float* fresh = s_new + wave * NS * LPW; // their new states, [NS][LPW] by compact slot
        uint32_t slot[VEC];
        uint32_t total = 0; // wave-uniform
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            slot[k] = 0;
            const unsigned long long m = __ballot(need_reset[k]);
            if (m != 0ull) { // wave-uniform
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (need_reset[k]) {
                    slot[k] = total + rank;
                    list[slot[k]] = (uint16_t)(lane * VEC + k);
                }
                total += (uint32_t)__popcll(m);
            }
        }
        if (total != 0) { // quiet waves skip everything below
            // DS operations of one wavefront execute in order: no barrier, only keep the compiler
            // from moving LDS accesses across the hand-over points.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint64_t wave_base = tile_base + (uint64_t)wave * LPW;
            for (uint32_t i = lane; i < total; i += 64u) { // one Philox4x32-10 per finished lane
                const uint64_t gl = wave_base + list[i];
#ifdef GYMRS_EXP_NO_PHILOX
                const u32x4 r = u32x4{{(uint32_t)gl * 2654435761u, (uint32_t)gl * 40503u, (uint32_t)a.tick * 2246822519u, (uint32_t)gl}};
#else
                const u32x4 r = draw4(a.seed, a.gid0 + gl, a.tick, kStreamReset);
#endif
                float ns[NS];
                Env::sample(r, a.lo, a.hi, ns);
#pragma unroll
                for (int j = 0; j < NS; ++j) fresh[j * LPW + i] = ns[j];
#ifndef GYMRS_EXP_NO_EPSTORE
                if (STATS || TLIM) a.ep_start[gl] = tick_next; // the new episode starts at the next tick
#endif
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float ret_sum = 0.0f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                if (need_reset[k]) {
#pragma unroll
                    for (int j = 0; j < NS; ++j) ls[j][k] = fresh[j * LPW + slot[k]];
                    if (STATS && !Env::kConstReward) {
                        ret_sum += d.ep_ret.v[k];
                        d.ep_ret.v[k] = 0.0f;
                    }
                }
            }
            if (STATS) {
                wave_resets += total; // folded into this wave's private statistics slot by step_block
                if (!Env::kConstReward) {
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) ret_sum += __shfl_xor(ret_sum, off);
                    wave_ret += ret_sum;
                }
            }
        }
    }

    // ---- stores ----
    Vec<float, VEC> reward;
#pragma unroll
    for (int k = 0; k < VEC; ++k) reward.v[k] = rw[k];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        Vec<float, VEC> out;
#pragma unroll
        for (int k = 0; k < VEC; ++k) out.v[k] = ls[j][k];
        store_vec<float, VEC>(a.s[j], base, a.n, FULL, out);
    }
    store_vec<float, VEC>(a.reward, base, a.n, FULL, reward);
    store_vec<uint8_t, VEC>(a.done, base, a.n, FULL, done);
    if (TLIM) store_vec<uint8_t, VEC>(a.truncated, base, a.n, FULL, trunc);
    if (Env::kHasBeyond && !AUTO) store_vec<uint8_t, VEC>(a.beyond, base, a.n, FULL, d.beyond);
    if (STATS && !Env::kConstReward) store_vec<float, VEC>(a.ep_ret, base, a.n, FULL, d.ep_ret);
    if (Env::kHasObsExtra) {
        Vec<float, VEC> oc, os;
        bool med = true;
#pragma unroll
        for (int k = 0; k < VEC; ++k) med = med && in_medium_range(ls[0][k]);
        if (__all(med)) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) sincos_medium(ls[0][k], &os.v[k], &oc.v[k]);
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) sincosf_(ls[0][k], &os.v[k], &oc.v[k]);
        }
        store_vec<float, VEC>(a.obs_cos, base, a.n, FULL, oc);
        store_vec<float, VEC>(a.obs_sin, base, a.n, FULL, os);
    }
}

template <class Env, int VEC, uint32_t FLAGS, int TILES, bool FULL>
__device__ __forceinline__ void step_block(const StepArgs& a, const typename Env::Consts& c, uint16_t* s_list, float* s_new)
{
    constexpr int LPB = kBlock * VEC;
    constexpr bool STATS = (FLAGS & GYMRS_AUTO_RESET) && (FLAGS & GYMRS_TRACK_STATS);
    const uint64_t block_base = (uint64_t)blockIdx.x * (LPB * TILES);
    // Episode statistics: every wavefront owns one {finished episodes, sum of returns} slot.  Its old
    // value is fetched with the first loads and the updated value leaves with the last stores, so the
    // hot path holds no atomic and nothing waits on the statistics.  (Launches are stream-ordered and
    // a slot has exactly one writer per launch.)
    unsigned long long* slot = a.block_stats + ((size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 2;
    unsigned long long old_resets = 0;
    double old_ret = 0.0;
    if (STATS) {
        old_resets = slot[0];
        if (!Env::kConstReward) old_ret = reinterpret_cast<const double*>(slot)[1];
    }
    TileRegs<Env, VEC, FLAGS> d[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
        load_tile<Env, VEC, FLAGS, FULL>(a, block_base + (uint64_t)t * LPB + (uint64_t)threadIdx.x * VEC, d[t]);
    uint32_t wave_resets = 0;
    float wave_ret = 0.0f;
#pragma unroll
    for (int t = 0; t < TILES; ++t)
        finish_tile<Env, VEC, FLAGS, FULL>(a, c, block_base + (uint64_t)t * LPB, d[t], s_list, s_new, wave_resets, wave_ret);
    if (STATS && wave_resets != 0 && (threadIdx.x & 63u) == 0) {
        slot[0] = old_resets + wave_resets;
        if (!Env::kConstReward) reinterpret_cast<double*>(slot)[1] = old_ret + (double)wave_ret;
    }
}

// waves_per_eu(4, 8): 4 workgroups of 4 waves per CU is all a 2^20-lane launch needs at VEC = 4, so the
// register allocator may use up to 128 VGPRs instead of spilling to reach 8 waves per SIMD.
template <class Env, int VEC, uint32_t FLAGS, int TILES>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 8))) void step_kernel(const StepArgs a, const typename Env::Consts c)
{
    constexpr bool AUTO = (FLAGS & GYMRS_AUTO_RESET) != 0;
    constexpr int LPB = kBlock * VEC;
    __shared__ uint16_t s_list[AUTO ? LPB : 1];
    __shared__ float s_new[AUTO ? Env::kState * LPB : 1];
    if (a.prio_div != 0) { // experiment: stagger the waves that share a SIMD (they come from different workgroups)
        switch ((blockIdx.x / a.prio_div) & 3u) {
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        case 2: __builtin_amdgcn_s_setprio(1); break;
        default: break;
        }
    }
    // workgroup-uniform: every workgroup but the last runs the unguarded body
    if ((uint64_t)(blockIdx.x + 1) * (LPB * TILES) <= a.n)
        step_block<Env, VEC, FLAGS, TILES, true>(a, c, s_list, s_new);
    else
        step_block<Env, VEC, FLAGS, TILES, false>(a, c, s_list, s_new);
}

// ---------------------------------------------------------------------------------------------
// Env::reset for every lane (cartpole.rs:485-516, mountain_car.rs:464-501): off the per-step path.
template <class Env>
__global__ __launch_bounds__(kBlock) void reset_kernel(const ResetArgs a)
{
    const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (lane >= a.n) return;
    const u32x4 r = draw4(a.seed, a.gid0 + lane, a.tick, kStreamReset);
    float ns[Env::kState];
    Env::sample(r, a.lo, a.hi, ns);
#pragma unroll
    for (int j = 0; j < Env::kState; ++j) a.s[j][lane] = ns[j];
    if (Env::kHasObsExtra) {
        float sn, cs;
        sincosf_(ns[0], &sn, &cs);
        a.obs_cos[lane] = cs;
        a.obs_sin[lane] = sn;
    }
    a.reward[lane] = 0.0f;
    a.done[lane] = 0;
    a.truncated[lane] = 0;
    if (Env::kHasBeyond) a.beyond[lane] = 0; // steps_beyond_terminated = None, cartpole.rs:504
    a.ep_start[lane] = (uint32_t)(a.tick + 1); // = the epoch the statistics are measured from
    if (a.ep_ret) a.ep_ret[lane] = 0.0f;
}

// Random-policy actions (examples/cartpole.rs:19 `rng.gen_range(0..=1)`), Philox stream 1.
template <class Env>
__global__ __launch_bounds__(kBlock) void fill_actions_kernel(typename Env::Action* out, uint64_t n, uint64_t gid0,
                                                              uint64_t seed, uint64_t t, uint32_t n_actions,
                                                              float max_torque)
{
    const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (lane >= n) return;
    const u32x4 r = draw4(seed, gid0 + lane, t, kStreamAction);
    if constexpr (sizeof(typename Env::Action) == 1) {
        out[lane] = (uint8_t)(((uint64_t)r.v[0] * n_actions) >> 32);
    } else {
        out[lane] = uniform_between(r.v[0], -max_torque, max_torque);
    }
}

// Statistics read-out (off the hot path).  acc = {L, E, R}: L = sum over lanes of
// (ep_start - epoch) = total length of the episodes finished since the last reset() (episodes tile a
// lane's time axis), E = finished episodes, R = sum of returns (Pendulum only; for the const-reward
// envs return = +-length).
__global__ __launch_bounds__(kBlock) void stats_accumulate_kernel(const uint32_t* __restrict__ ep_start, uint64_t n,
                                                                  uint32_t epoch,
                                                                  const unsigned long long* __restrict__ bs,
                                                                  uint32_t n_blocks, unsigned long long* __restrict__ acc)
{
    __shared__ unsigned long long s_len[kBlock], s_ep[kBlock];
    __shared__ double s_ret[kBlock];
    unsigned long long len = 0, ep = 0;
    double ret = 0.0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) len += (uint32_t)(ep_start[i] - epoch);
    for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < n_blocks; b += stride) {
        ep += bs[b * 2];
        ret += reinterpret_cast<const double*>(bs)[b * 2 + 1];
    }
    s_len[threadIdx.x] = len;
    s_ep[threadIdx.x] = ep;
    s_ret[threadIdx.x] = ret;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s_len[threadIdx.x] += s_len[threadIdx.x + off];
            s_ep[threadIdx.x] += s_ep[threadIdx.x + off];
            s_ret[threadIdx.x] += s_ret[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(acc + 0, s_len[0]);
        atomicAdd(acc + 1, s_ep[0]);
        atomicAdd(reinterpret_cast<double*>(acc + 2), s_ret[0]);
    }
}

// mode 0: out4 = {sum_return, sum_length, n_episodes, n_steps}; mode 1: remember L as the new base
// (statistics cleared); mode 2: base = 0 (after reset()).
__global__ void stats_finalize_kernel(const unsigned long long* acc, unsigned long long* base, int mode, int reward_sign,
                                      double n_steps, double* out4)
{
    if (mode == 1) {
        base[0] = acc[0];
        return;
    }
    if (mode == 2) {
        base[0] = 0;
        return;
    }
    const double len = (double)(acc[0] - base[0]);
    out4[0] = reward_sign != 0 ? reward_sign * len : *reinterpret_cast<const double*>(acc + 2);
    out4[1] = len;
    out4[2] = (double)acc[1];
    out4[3] = n_steps;
}

// ---------------------------------------------------------------------------------------------
// launch tables
template <class Env, int VEC, uint32_t FLAGS, int TILES>
static hipError_t launch_one(const StepArgs& a, const void* consts, hipStream_t stream)
{
    const uint32_t grid = step_grid(a.n, VEC * TILES);
    hipLaunchKernelGGL((step_kernel<Env, VEC, FLAGS, TILES>), dim3(grid), dim3(kBlock), 0, stream, a,
                       *static_cast<const typename Env::Consts*>(consts));
    return hipGetLastError();
}

template <class Env, int VEC, uint32_t FLAGS>
static hipError_t launch_tiles(int tiles, const StepArgs& a, const void* consts, hipStream_t stream)
{
    switch (tiles) {
    case 1: return launch_one<Env, VEC, FLAGS, 1>(a, consts, stream);
    case 2: return launch_one<Env, VEC, FLAGS, 2>(a, consts, stream);
#ifndef GYMRS_PROBE_BUILD
    case 4: return launch_one<Env, VEC, FLAGS, 4>(a, consts, stream);
#endif
    default: return hipErrorInvalidValue;
    }
}

template <class Env, int VEC>
static hipError_t launch_flags(uint32_t flags, int tiles, const StepArgs& a, const void* consts, hipStream_t stream)
{
    constexpr uint32_t A = GYMRS_AUTO_RESET, S = GYMRS_TRACK_STATS, T = GYMRS_TIME_LIMIT;
    if (!(flags & A)) flags &= ~S; // statistics need auto-reset
    switch (flags & (A | S | T)) {
    case 0: return launch_tiles<Env, VEC, 0>(tiles, a, consts, stream);
    case A: return launch_tiles<Env, VEC, A>(tiles, a, consts, stream);
    case A | S: return launch_tiles<Env, VEC, A | S>(tiles, a, consts, stream);
#ifndef GYMRS_PROBE_BUILD
    case T: return launch_tiles<Env, VEC, T>(tiles, a, consts, stream);
    case A | T: return launch_tiles<Env, VEC, A | T>(tiles, a, consts, stream);
    case A | S | T: return launch_tiles<Env, VEC, A | S | T>(tiles, a, consts, stream);
#endif
    default: return hipErrorInvalidValue;
    }
}

template <class Env>
static hipError_t launch_vec(int vec, uint32_t flags, int tiles, const StepArgs& a, const void* consts, hipStream_t stream)
{
    switch (vec) {
#ifndef GYMRS_PROBE_BUILD
    case 1: return launch_flags<Env, 1>(flags, tiles, a, consts, stream);
#endif
    case 2: return launch_flags<Env, 2>(flags, tiles, a, consts, stream);
    case 4: return launch_flags<Env, 4>(flags, tiles, a, consts, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_step(gymrs_env_kind kind, int vec, int tiles, uint32_t flags, const StepArgs& a, const void* consts,
                       hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    switch (kind) {
    case GYMRS_CARTPOLE: return launch_vec<CartPoleT>(vec, flags, tiles, a, consts, stream);
    case GYMRS_MOUNTAIN_CAR: return launch_vec<MountainCarT>(vec, flags, tiles, a, consts, stream);
#ifndef GYMRS_PROBE_BUILD
    case GYMRS_PENDULUM: return launch_vec<PendulumT>(vec, flags, tiles, a, consts, stream);
#endif
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_reset(gymrs_env_kind kind, const ResetArgs& a, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    const uint32_t grid = (uint32_t)((a.n + kBlock - 1) / kBlock);
    switch (kind) {
    case GYMRS_CARTPOLE: hipLaunchKernelGGL((reset_kernel<CartPoleT>), dim3(grid), dim3(kBlock), 0, stream, a); break;
    case GYMRS_MOUNTAIN_CAR: hipLaunchKernelGGL((reset_kernel<MountainCarT>), dim3(grid), dim3(kBlock), 0, stream, a); break;
    case GYMRS_PENDULUM: hipLaunchKernelGGL((reset_kernel<PendulumT>), dim3(grid), dim3(kBlock), 0, stream, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_fill_actions(gymrs_env_kind kind, void* actions, uint64_t n, uint64_t gid0, uint64_t seed, uint64_t t,
                               float max_torque, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint32_t grid = (uint32_t)((n + kBlock - 1) / kBlock);
    switch (kind) {
    case GYMRS_CARTPOLE:
        hipLaunchKernelGGL((fill_actions_kernel<CartPoleT>), dim3(grid), dim3(kBlock), 0, stream,
                           static_cast<uint8_t*>(actions), n, gid0, seed, t, 2u, 0.0f);
        break;
    case GYMRS_MOUNTAIN_CAR:
        hipLaunchKernelGGL((fill_actions_kernel<MountainCarT>), dim3(grid), dim3(kBlock), 0, stream,
                           static_cast<uint8_t*>(actions), n, gid0, seed, t, 3u, 0.0f);
        break;
    case GYMRS_PENDULUM:
        hipLaunchKernelGGL((fill_actions_kernel<PendulumT>), dim3(grid), dim3(kBlock), 0, stream,
                           static_cast<float*>(actions), n, gid0, seed, t, 0u, max_torque);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_stats(const StatsArgs& a, int mode, hipStream_t stream)
{
    hipError_t err = hipMemsetAsync(a.acc, 0, 3 * sizeof(unsigned long long), stream);
    if (err != hipSuccess) return err;
    if (mode != 2) {
        const uint64_t work = a.track ? a.n : 0; // without GYMRS_TRACK_STATS ep_start carries no statistics
        uint32_t grid = (uint32_t)((work + kBlock - 1) / kBlock);
        grid = grid < 1 ? 1 : (grid > 1024 ? 1024 : grid);
        hipLaunchKernelGGL(stats_accumulate_kernel, dim3(grid), dim3(kBlock), 0, stream, a.ep_start, work, a.epoch,
                           a.block_stats, a.track ? a.n_blocks : 0u, a.acc);
    }
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(1), dim3(1), 0, stream, a.acc, a.base, mode, a.reward_sign, a.n_steps, a.out4);
    return hipGetLastError();
}

} // namespace gymrs
