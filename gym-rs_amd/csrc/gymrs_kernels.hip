// gymrs_kernels.hip — hand-written CDNA4 (gfx950) kernels of the batched classic-control stepper.
//
// One kernel template, instantiated per env type: each lane is one independent gym-rs env
//   CartPoleEnv::step     /root/reference/src/envs/classical_control/cartpole.rs:398-483
//   MountainCarEnv::step  /root/reference/src/envs/classical_control/mountain_car.rs:398-435
//   (+ reset, cartpole.rs:485-516 / mountain_car.rs:464-501, fused in for finished lanes)
//
// Shape of the work: pure streaming, memory-latency/bandwidth-bound, no contraction ->
// no MFMA.  What the measurements on MI355X dictated (profiles/, DESIGN.md "what was measured"):
//   * SoA f32 state in HBM; a work-item owns 4 consecutive lanes and moves them with one dwordx4
//     load + store per array (16 B per work-item, 1 KiB per wave instruction); actions and done flags
//     travel as one packed dword.
//   * A 2^20-lane launch is ONE generation of waves (1024 workgroups x 4 wave64, all resident): every
//     wave loads, computes, stores once.  The kernel time is launch + memory round trip + whatever
//     VALU work and latency sit between a wave's last load and its stores.
//   * The physics of the 4 lanes of a work-item runs as ONE basic block whenever the whole wave is on
//     the common path (valid actions, angle in the polynomial's range), so the 4 independent dependency
//     chains interleave; the general per-lane code is the wave-uniform fallback.  Both give the same
//     bits.  (Packed v_pk_*_f32 math was measured to run at half rate on this part: no gain, not used.)
//   * auto-reset: about 1 lane in 22 finishes per step, so ~95 % of the 256-lane waves hold a finished
//     lane.  A wave __ballot done-mask + mbcnt ranks compact them into the wave's own LDS segment and the
//     wave evaluates ONE Philox4x32-10 block per finished lane (one pass of ~11 active lanes) instead of
//     one per lane under divergence; the fresh states return through LDS to the owning work-item, which
//     stores them with its vector store.  Everything stays inside one wavefront: no s_barrier (a
//     workgroup-level variant with one Philox pass per workgroup measured the same on CartPole and 10 %
//     slower on MountainCar: its barrier re-couples the waves), no speculative evaluation (Philox in the
//     load shadow measured no better: v_mad_u64_u32 is quarter rate and the shadow is not free).
//     Quiet waves (MountainCar, Pendulum: nearly all) skip the whole path.
//   * Episode statistics cost the step nothing it can wait on: a re-armed lane's worker only WRITES the
//     tick its next episode starts at (episodes tile a lane's time axis, so the sum of finished lengths
//     is sum(ep_start) - n*epoch, evaluated when statistics are read) and every wave keeps a private
//     episode counter slot (plain load at start, plain store at end; no atomics).
#include <type_traits>

#include "gymrs_kernels.h"

namespace gymrs {


// Developer instrumentation (tools/probe --trace): per-wave s_memtime stamps at the phase boundaries.
#ifdef GYMRS_TRACE_TIMES
#define GYMRS_STAMP(slot_)                                                                                   \
    do {                                                                                                     \
        if (a.trace && (threadIdx.x & 63u) == 0)                                                             \
            a.trace[((size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 8 + (slot_)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define GYMRS_STAMP(slot_) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// vector access helpers
template <class T, int V>
struct alignas(sizeof(T) * V) Vec {
    T v[V];
};

// NT = the accesses carry the non-temporal hint.  Every array is read once and written once per step and
// the next reader is the NEXT kernel (a kernel boundary flushes/invalidates the per-XCD L2s anyway), so
// nothing is gained by keeping the lines in L2: measured -0.4 us per 2^20-lane launch.  The hint also keeps
// the lines out of the 256 MB Infinity Cache, which is what serves the next step's reads while the working
// set fits there, so above ~2^20 CartPole lanes plain accesses win again; the engine picks per launch.
template <class T, int V, bool NT>
__device__ __forceinline__ Vec<T, V> load_vec(const T* __restrict__ p, uint64_t base, uint64_t n, bool full, T fill)
{
    typedef T vt __attribute__((ext_vector_type(V)));
    Vec<T, V> r;
    if (full) {
        const vt x = NT ? __builtin_nontemporal_load(reinterpret_cast<const vt*>(p + base)) : *reinterpret_cast<const vt*>(p + base);
#pragma unroll
        for (int k = 0; k < V; ++k) r.v[k] = x[k];
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k) r.v[k] = (base + k < n) ? p[base + k] : fill;
    }
    return r;
}

template <class T, int V, bool NT>
__device__ __forceinline__ void store_vec(T* __restrict__ p, uint64_t base, uint64_t n, bool full, const Vec<T, V>& r)
{
    typedef T vt __attribute__((ext_vector_type(V)));
    if (full) {
        vt x;
#pragma unroll
        for (int k = 0; k < V; ++k) x[k] = r.v[k];
        if (NT)
            __builtin_nontemporal_store(x, reinterpret_cast<vt*>(p + base));
        else
            *reinterpret_cast<vt*>(p + base) = x;
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k)
            if (base + k < n) p[base + k] = r.v[k];
    }
}

// ---------------------------------------------------------------------------------------------
// Env policies: what differs between the three env types.
struct CartPoleT {
    using Consts = CartPoleConsts;
    using Action = uint8_t;
    static constexpr int kState = 4;
    static constexpr bool kConstReward = true;    // under auto-reset every step pays 1.0 (cartpole.rs:455-459)
    static constexpr bool kHasBeyond = true;
    static constexpr bool kHasObsExtra = false;
    static constexpr bool kNeverTerminates = false;
    __device__ static bool valid(Action a) { return a < 2; } // Discrete(2).contains, discrete.rs:14-19
    __device__ static void advance(const Consts& c, float* st, Action a, float& reward, bool& done)
    {
        done = cartpole_advance(c, st[0], st[1], st[2], st[3], a);
        reward = 1.0f; // the beyond-terminated case is applied by the caller when auto-reset is off
    }
    // branch-free variant, legal when fast_ok holds (|theta| <= pi/4: the polynomial needs no reduction)
    __device__ static bool fast_ok(const float* st, Action a) { return a < 2 && in_small_range(st[2]); }
    static constexpr int kVariants = 2; // the integrator choice is hoisted out of the per-lane code
    __device__ static int variant(const Consts& c) { return c.integrator == 0 ? 0 : 1; }
    template <int INTEG>
    __device__ static void advance_fast(const Consts& c, float* st, Action a, float& reward, bool& done)
    {
        done = cartpole_advance<SinCosSmall, INTEG>(c, st[0], st[1], st[2], st[3], a);
        reward = 1.0f;
    }
    __device__ static void sample(const u32x4& r, const SampleBox& b, float* st) { cartpole_sample(r, b, st[0], st[1], st[2], st[3]); }
};

struct MountainCarT {
    using Consts = MountainCarConsts;
    using Action = uint8_t;
    static constexpr int kState = 2;
    static constexpr bool kConstReward = true; // -1.0 on every step (mountain_car.rs:423)
    static constexpr bool kHasBeyond = false;
    static constexpr bool kHasObsExtra = false;
    static constexpr bool kNeverTerminates = false;
    __device__ static bool valid(Action a) { return a < 3; } // Discrete(3)
    __device__ static void advance(const Consts& c, float* st, Action a, float& reward, bool& done)
    {
        done = mountain_car_advance(c, st[0], st[1], a);
        reward = -1.0f;
    }
    __device__ static bool fast_ok(const float* st, Action a) { return a < 3 && in_short_range(3.0f * st[0]); }
    static constexpr int kVariants = 1;
    __device__ static int variant(const Consts&) { return 0; }
    template <int>
    __device__ static void advance_fast(const Consts& c, float* st, Action a, float& reward, bool& done)
    {
        done = mountain_car_advance<SinCosShort>(c, st[0], st[1], a);
        reward = -1.0f;
    }
    __device__ static void sample(const u32x4& r, const SampleBox& b, float* st) { mountain_car_sample(r, b, st[0], st[1]); }
};

struct PendulumT { // spec-derived, not in the reference
    using Consts = PendulumConsts;
    using Action = float;
    static constexpr int kState = 2;
    static constexpr bool kConstReward = false;
    static constexpr bool kHasBeyond = false;
    static constexpr bool kHasObsExtra = true;
    // No termination and no invalid actions: every lane's episode clock is the same, so the time limit is a
    // kernel argument (StepArgs::truncate_all) instead of a per-lane compare against a dense ep_start read.
    static constexpr bool kNeverTerminates = true;
    __device__ static bool valid(Action) { return true; } // a Box action is clipped, never rejected
    __device__ static void advance(const Consts& c, float* st, Action a, float& reward, bool& done)
    {
        reward = pendulum_advance(c, st[0], st[1], a);
        done = false;
    }
    __device__ static bool fast_ok(const float* st, Action) { return in_short_range(st[0]); } // |theta| <= 200
    static constexpr int kVariants = 1;
    __device__ static int variant(const Consts&) { return 0; }
    template <int>
    __device__ static void advance_fast(const Consts& c, float* st, Action a, float& reward, bool& done)
    {
        reward = pendulum_advance<SinCosShort>(c, st[0], st[1], a);
        done = false;
    }
    __device__ static void sample(const u32x4& r, const SampleBox& b, float* st) { pendulum_sample(r, b, st[0], st[1]); }
};

// ---------------------------------------------------------------------------------------------
// THE hot kernel: one Env::step() per lane; a work-item owns VEC lanes, a wavefront 64*VEC, a workgroup 256*VEC.
template <class Env, int VEC, uint32_t FLAGS>
struct TileRegs {
    static constexpr bool AUTO = (FLAGS & GYMRS_AUTO_RESET) != 0;
    static constexpr bool STATS = AUTO && (FLAGS & GYMRS_TRACK_STATS) != 0;
    static constexpr bool TLIM = (FLAGS & GYMRS_TIME_LIMIT) != 0;
    static constexpr bool NT = (FLAGS & kFlagNonTemporal) != 0;
    Vec<float, VEC> st[Env::kState];
    Vec<typename Env::Action, VEC> act;
    Vec<uint8_t, VEC> beyond;
    Vec<uint32_t, VEC> ep_start;
    Vec<float, VEC> ep_ret;
};

template <class Env, int VEC, uint32_t FLAGS, bool FULL>
__device__ __forceinline__ void load_tile(const StepArgs& a, uint64_t base, TileRegs<Env, VEC, FLAGS>& d)
{
    constexpr int kVec = VEC;
    using R = TileRegs<Env, VEC, FLAGS>;
    using Action = typename Env::Action;
#pragma unroll
    for (int j = 0; j < Env::kState; ++j) d.st[j] = load_vec<float, kVec, R::NT>(a.s[j], base, a.n, FULL, 0.0f);
    d.act = load_vec<Action, kVec, R::NT>(static_cast<const Action*>(a.action), base, a.n, FULL, Action(0));
    if (Env::kHasBeyond && !R::AUTO) d.beyond = load_vec<uint8_t, kVec, R::NT>(a.beyond, base, a.n, FULL, uint8_t(0));
    if (R::TLIM && !Env::kNeverTerminates) d.ep_start = load_vec<uint32_t, kVec, false>(a.ep_start, base, a.n, FULL, 0u);
    if (R::STATS && !Env::kConstReward) d.ep_ret = load_vec<float, kVec, R::NT>(a.ep_ret, base, a.n, FULL, 0.0f);
}

// LDS of one workgroup for the auto-reset hand-off: every wavefront uses its own 256-entry segment
// (list of finished lanes, their fresh states, their finished returns); waves never touch each other's.
template <class Env, int VEC>
struct ResetLds {
    static constexpr int kLanes = kBlock * VEC;
    uint16_t list[kLanes];
    struct alignas(Env::kState * 4) State {
        float v[Env::kState];
    };
    State fresh[kLanes]; // one ds_write/ds_read of 8 or 16 bytes per finished lane
    float ret[Env::kConstReward ? 1 : kLanes];
};

// The branch-free physics of the 4 lanes of a work-item in one basic block (V = Env variant).
template <class Env, int VEC, int V>
__device__ __forceinline__ void advance_fast_all(const typename Env::Consts& c, float (&ls)[Env::kState][VEC],
                                                 const typename Env::Action (&la)[VEC], float (&rw)[VEC], bool (&dn)[VEC])
{
    constexpr int kVec = VEC;
    constexpr int NS = Env::kState;
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
        float lane_st[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) lane_st[j] = ls[j][k];
        Env::template advance_fast<V>(c, lane_st, la[k], rw[k], dn[k]);
#pragma unroll
        for (int j = 0; j < NS; ++j) ls[j][k] = lane_st[j];
    }
}

// physics + auto-reset selection + stores of one tile whose loads were issued by load_tile
template <class Env, int VEC, uint32_t FLAGS, bool FULL>
__device__ __forceinline__ void finish_tile(const StepArgs& a, const typename Env::Consts& c, uint64_t base,
                                            TileRegs<Env, VEC, FLAGS>& d, ResetLds<Env, VEC>& lds, unsigned long long old_resets,
                                            double old_ret)
{
    constexpr int kVec = VEC;
    using R = TileRegs<Env, VEC, FLAGS>;
    constexpr bool AUTO = R::AUTO, STATS = R::STATS, TLIM = R::TLIM;
    constexpr int NS = Env::kState;
    using Action = typename Env::Action;
    const uint32_t tick_next = (uint32_t)(a.tick + 1);

    // ---- physics ----
    // State is unpacked into plain per-lane scalars (registers) for the arithmetic and re-packed
    // into vectors only for the stores.
    float ls[NS][kVec];
    Action la[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
        la[k] = d.act.v[k];
#pragma unroll
        for (int j = 0; j < NS; ++j) ls[j][k] = d.st[j].v[k];
    }
    float rw[kVec];
    bool dn[kVec], tr[kVec], need_reset[kVec];
#ifdef GYMRS_TRACE_TIMES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GYMRS_STAMP(2); // all loads have landed
#endif
    bool fast = FULL;
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
        float lane_st[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) lane_st[j] = ls[j][k];
        fast = fast && Env::fast_ok(lane_st, la[k]);
    }
    if (__all(fast)) { // wave-uniform: the common path
        if (Env::kVariants == 1 || Env::variant(c) == 0)
            advance_fast_all<Env, VEC, 0>(c, ls, la, rw, dn);
        else
            advance_fast_all<Env, VEC, 1>(c, ls, la, rw, dn);
    } else { // general per-lane code: ragged tail, invalid actions, angles outside the fast range
        uint32_t n_bad = 0, first_bad = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < kVec; ++k) {
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
    GYMRS_STAMP(3); // physics done
    Vec<uint8_t, kVec> done, trunc;
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
        const bool stepped = (FULL || (base + k < a.n)) && Env::valid(la[k]);
        if (Env::kHasBeyond && !AUTO) { // cartpole.rs:455-464
            bool b = d.beyond.v[k] != 0;
            const float r = cartpole_reward(dn[k], b);
            if (stepped) {
                rw[k] = r;
                d.beyond.v[k] = b ? 1 : 0;
            }
        }
        if (Env::kNeverTerminates)
            tr[k] = TLIM && stepped && a.truncate_all != 0;
        else
            tr[k] = TLIM && stepped && (tick_next - d.ep_start.v[k]) >= c.max_steps;
        if (STATS && !Env::kConstReward) d.ep_ret.v[k] += rw[k];
        done.v[k] = dn[k] ? 1 : 0;
        trunc.v[k] = tr[k] ? 1 : 0;
        need_reset[k] = AUTO && (dn[k] || tr[k]);
    }

    // ---- auto-reset: wave __ballot done-mask -> LDS-staged Philox, all inside one wavefront ----
    if (AUTO) {
        constexpr int LPW = 64 * kVec; // lanes per wavefront = capacity of a wave's LDS segment
        const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
        uint16_t* list = lds.list + wave * LPW; // this wave's compacted list of finished lanes
        uint32_t slot[kVec];
        uint32_t total = 0; // finished lanes of this wave (wave-uniform)
#pragma unroll
        for (int k = 0; k < kVec; ++k) {
            const unsigned long long m = __ballot(need_reset[k]);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            slot[k] = total + rank;
            if (need_reset[k]) {
                list[slot[k]] = (uint16_t)(lane * kVec + k); // wave-local lane
                if (STATS && !Env::kConstReward) lds.ret[wave * LPW + slot[k]] = d.ep_ret.v[k];
            }
            total += (uint32_t)__popcll(m);
        }
        if (total != 0) { // quiet waves (MountainCar / Pendulum: nearly all) skip everything below
            // DS operations of one wavefront execute in order: no barrier is needed, only the compiler
            // must not move LDS accesses across the hand-over points.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint64_t wave_base = (uint64_t)blockIdx.x * (kBlock * kVec) + (uint64_t)wave * LPW;
            float ret_sum = 0.0f;
            for (uint32_t i = lane; i < total; i += 64u) { // one Philox4x32-10 block per finished lane
                const uint64_t gl = wave_base + list[i];
                const u32x4 r = draw4(a.seed, a.gid0 + gl, a.tick, kStreamReset);
                float ns[NS];
                Env::sample(r, a.box, ns);
                typename ResetLds<Env, VEC>::State fs;
#pragma unroll
                for (int j = 0; j < NS; ++j) fs.v[j] = ns[j];
                lds.fresh[wave * LPW + i] = fs;
                if (STATS || TLIM) a.ep_start[gl] = tick_next; // the new episode starts at the next tick (plain store:
                                                               // a non-temporal scattered dword store measured slower)
                if (STATS && !Env::kConstReward) ret_sum += lds.ret[wave * LPW + i]; // return of the finished episode
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < kVec; ++k) {
                if (need_reset[k]) {
                    const typename ResetLds<Env, VEC>::State fs = lds.fresh[wave * LPW + slot[k]];
#pragma unroll
                    for (int j = 0; j < NS; ++j) ls[j][k] = fs.v[j];
                    if (STATS && !Env::kConstReward) d.ep_ret.v[k] = 0.0f;
                }
            }
            if (STATS) { // the wave's private statistics slot: plain read-modify-write, no atomics
                unsigned long long* bs = a.block_stats + ((size_t)blockIdx.x * (kBlock / 64) + wave) * 2;
                if (!Env::kConstReward) {
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) ret_sum += __shfl_xor(ret_sum, off);
                    if (lane == 0) reinterpret_cast<double*>(bs)[1] = old_ret + (double)ret_sum;
                }
                if (lane == 0) bs[0] = old_resets + total;
            }
        }
    }

    GYMRS_STAMP(4); // auto-reset done
    // ---- stores ----
    Vec<float, kVec> reward;
#pragma unroll
    for (int k = 0; k < kVec; ++k) reward.v[k] = rw[k];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        Vec<float, kVec> out;
#pragma unroll
        for (int k = 0; k < kVec; ++k) out.v[k] = ls[j][k];
        store_vec<float, kVec, R::NT>(a.s[j], base, a.n, FULL, out);
    }
    store_vec<float, kVec, R::NT>(a.reward, base, a.n, FULL, reward);
    store_vec<uint8_t, kVec, R::NT>(a.done, base, a.n, FULL, done);
    if (TLIM) store_vec<uint8_t, kVec, R::NT>(a.truncated, base, a.n, FULL, trunc);
    if (Env::kHasBeyond && !AUTO) store_vec<uint8_t, kVec, R::NT>(a.beyond, base, a.n, FULL, d.beyond);
    if (STATS && !Env::kConstReward) store_vec<float, kVec, R::NT>(a.ep_ret, base, a.n, FULL, d.ep_ret);
    if (Env::kHasObsExtra) {
        Vec<float, kVec> oc, os;
        bool med = true;
#pragma unroll
        for (int k = 0; k < kVec; ++k) med = med && in_short_range(ls[0][k]);
        if (__all(med)) {
#pragma unroll
            for (int k = 0; k < kVec; ++k) sincos_short(ls[0][k], &os.v[k], &oc.v[k]);
        } else {
#pragma unroll
            for (int k = 0; k < kVec; ++k) sincosf_(ls[0][k], &os.v[k], &oc.v[k]);
        }
        store_vec<float, kVec, R::NT>(a.obs_cos, base, a.n, FULL, oc);
        store_vec<float, kVec, R::NT>(a.obs_sin, base, a.n, FULL, os);
    }
    GYMRS_STAMP(5); // stores issued
#ifdef GYMRS_TRACE_TIMES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

template <class Env, int VEC, uint32_t FLAGS, bool FULL>
__device__ __forceinline__ void step_block(const StepArgs& a, const typename Env::Consts& c, ResetLds<Env, VEC>& lds)
{
    constexpr int kVec = VEC;
    constexpr int LPB = kBlock * kVec;
    constexpr bool AUTO = (FLAGS & GYMRS_AUTO_RESET) != 0;
    constexpr bool STATS = AUTO && (FLAGS & GYMRS_TRACK_STATS);
    const uint64_t base = (uint64_t)blockIdx.x * LPB + (uint64_t)threadIdx.x * kVec;
    GYMRS_STAMP(0);
    TileRegs<Env, VEC, FLAGS> d;
    load_tile<Env, VEC, FLAGS, FULL>(a, base, d);
    // Episode statistics: every wavefront owns one {finished episodes, sum of returns} slot.  The old value
    // is fetched right behind the state loads (after them, so that the slot pointer does not split the
    // kernel-argument fetch in two) and the updated value leaves with the wave's last stores: the hot
    // path holds no atomic and nothing waits on the statistics.  (Launches are stream-ordered and a slot
    // has exactly one writer per launch.)
    unsigned long long old_resets = 0;
    double old_ret = 0.0;
    if (STATS) {
        const unsigned long long* bs = a.block_stats + ((size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 2;
        old_resets = bs[0];
        if (!Env::kConstReward) old_ret = reinterpret_cast<const double*>(bs)[1];
    }
    GYMRS_STAMP(1);
    finish_tile<Env, VEC, FLAGS, FULL>(a, c, base, d, lds, old_resets, old_ret);
    GYMRS_STAMP(6);
}

// VEC lanes per work-item: 4 (1024 workgroups for 2^20 lanes, 4 waves per SIMD), 8 (2 waves per SIMD) or 16
// (1 wave per SIMD).  More lanes per wave = fewer waves = fewer per-wave fixed costs (address set-up, the
// auto-reset Philox pass, which costs the same whether 11 or 45 of its 64 lanes are active) at the price of
// registers; waves_per_eu lets the allocator use them instead of spilling to chase occupancy.
// The pointers and the lane count the first instructions of a wave need are separate scalar kernel parameters
// placed first, so that kernel-argument preloading (-mllvm -amdgpu-kernarg-preload-count, see build.py) can
// deliver them in SGPRs at wave launch instead of behind an s_load round trip; the rest travels in StepArgs.
template <class Env, int VEC, uint32_t FLAGS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, 16 / VEC < 1 ? 1 : 16 / VEC))) void step_kernel(
    float* s0, float* s1, float* s2, float* s3, const void* action, uint64_t n, const StepArgs rest,
    const typename Env::Consts c)
{
    constexpr int LPB = kBlock * VEC;
    __shared__ ResetLds<Env, VEC> lds;
    StepArgs a = rest;
    a.s[0] = s0;
    a.s[1] = s1;
    a.s[2] = s2;
    a.s[3] = s3;
    a.action = action;
    a.n = n;
    if (rest.tick_base) a.tick = rest.tick + *rest.tick_base; // replayed HIP graph: the tick lives on the device
    // workgroup-uniform: every workgroup but the last runs the unguarded body
    if ((uint64_t)(blockIdx.x + 1) * LPB <= a.n)
        step_block<Env, VEC, FLAGS, true>(a, c, lds);
    else
        step_block<Env, VEC, FLAGS, false>(a, c, lds);
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
    Env::sample(r, a.box, ns);
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
                                                                  uint32_t n_slots, unsigned long long* __restrict__ acc)
{
    __shared__ unsigned long long s_len[kBlock], s_ep[kBlock];
    __shared__ double s_ret[kBlock];
    unsigned long long len = 0, ep = 0;
    double ret = 0.0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) len += (uint32_t)(ep_start[i] - epoch);
    for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < n_slots; b += stride) {
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
template <class Env, int VEC, uint32_t FLAGS>
static hipError_t launch_one(const StepArgs& a, const void* consts, hipStream_t stream)
{
    hipLaunchKernelGGL((step_kernel<Env, VEC, FLAGS>), dim3(step_grid(a.n, VEC)), dim3(kBlock), 0, stream, a.s[0], a.s[1],
                       a.s[2], a.s[3], a.action, a.n, a, *static_cast<const typename Env::Consts*>(consts));
    return hipGetLastError();
}

template <class Env, int VEC, uint32_t NTBIT>
static hipError_t launch_flags_nt(uint32_t flags, const StepArgs& a, const void* consts, hipStream_t stream)
{
    constexpr uint32_t A = GYMRS_AUTO_RESET, S = GYMRS_TRACK_STATS, T = GYMRS_TIME_LIMIT;
    if (!(flags & A)) flags &= ~S; // statistics need auto-reset
    switch (flags & (A | S | T)) {
    case 0: return launch_one<Env, VEC, 0 | NTBIT>(a, consts, stream);
    case A: return launch_one<Env, VEC, A | NTBIT>(a, consts, stream);
    case A | S: return launch_one<Env, VEC, A | S | NTBIT>(a, consts, stream);
    case T: return launch_one<Env, VEC, T | NTBIT>(a, consts, stream);
    case A | T: return launch_one<Env, VEC, A | T | NTBIT>(a, consts, stream);
    case A | S | T: return launch_one<Env, VEC, A | S | T | NTBIT>(a, consts, stream);
    default: return hipErrorInvalidValue;
    }
}

template <class Env, int VEC>
static hipError_t launch_flags(uint32_t flags, const StepArgs& a, const void* consts, hipStream_t stream)
{
    return (flags & kFlagNonTemporal) ? launch_flags_nt<Env, VEC, kFlagNonTemporal>(flags, a, consts, stream)
                                      : launch_flags_nt<Env, VEC, 0u>(flags, a, consts, stream);
}

template <class Env>
static hipError_t launch_vec(int vec, uint32_t flags, const StepArgs& a, const void* consts, hipStream_t stream)
{
    switch (vec) {
    case 4: return launch_flags<Env, 4>(flags, a, consts, stream);
    case 8: return launch_flags<Env, 8>(flags, a, consts, stream);
    case 16: return launch_flags<Env, 16>(flags, a, consts, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_step(gymrs_env_kind kind, int vec, uint32_t flags, const StepArgs& a, const void* consts,
                       hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    switch (kind) {
    case GYMRS_CARTPOLE: return launch_vec<CartPoleT>(vec, flags, a, consts, stream);
    case GYMRS_MOUNTAIN_CAR: return launch_vec<MountainCarT>(vec, flags, a, consts, stream);
    case GYMRS_PENDULUM: return launch_vec<PendulumT>(vec, flags, a, consts, stream);
    default: return hipErrorInvalidValue;
    }
}

__global__ void tick_advance_kernel(unsigned long long* tick_dev, unsigned long long by) { *tick_dev += by; }

hipError_t launch_tick_advance(unsigned long long* tick_dev, unsigned long long by, hipStream_t stream)
{
    hipLaunchKernelGGL(tick_advance_kernel, dim3(1), dim3(1), 0, stream, tick_dev, by);
    return hipGetLastError();
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
