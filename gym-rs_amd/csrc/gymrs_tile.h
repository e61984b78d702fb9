// gymrs_tile.h -- device code shared by the per-step kernel (gymrs_step_impl.h) and the fused rollout kernel
// (gymrs_rollout.hip): the env policies, the register tile of a work-item, and one Env::step() of that tile
// (physics + wave-level auto-reset).  gfx950 device code only.
#pragma once
#include <type_traits>

#include "gymrs_kernels.h"

namespace gymrs {


// Developer instrumentation (tools/probe --trace): per-wave s_memtime stamps at the phase boundaries.
#ifdef GYMRS_TRACE_TIMES
#define GYMRS_STAMP(slot_)                                                                                   \
    do {                                                                                                     \
        if (a.trace && (threadIdx.x & 63u) == 0)                                                             \
            a.trace[((size_t)blockIdx.x * a.trace_wpb + (threadIdx.x >> 6)) * 8 + (slot_)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define GYMRS_STAMP(slot_) do { } while (0)
#endif

// Sum of `v` over the 64 lanes of the wavefront, returned wave-uniform.  Six DPP adds (quad swaps, row
// mirrors, row broadcasts -- the cross-lane paths of the VALU itself) instead of six ds_bpermute round trips.
__device__ __forceinline__ float wave_sum(float v)
{
    auto dpp_add = [](float x, auto ctrl, auto row_mask) {
        const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(row_mask)::value, 0xf, true);
        return x + __builtin_bit_cast(float, moved);
    };
    using std::integral_constant;
    v = dpp_add(v, integral_constant<int, 0xb1>{}, integral_constant<int, 0xf>{});  // quad_perm [1,0,3,2]
    v = dpp_add(v, integral_constant<int, 0x4e>{}, integral_constant<int, 0xf>{});  // quad_perm [2,3,0,1]
    v = dpp_add(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{}); // row_half_mirror
    v = dpp_add(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{}); // row_mirror: every lane holds its row's sum
    v = dpp_add(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}); // row_bcast:15 into rows 1 and 3
    v = dpp_add(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}); // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// The same for an integer (wave-uniform result).
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    auto dpp_add = [](uint32_t x, auto ctrl, auto row_mask) {
        return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, decltype(ctrl)::value, decltype(row_mask)::value, 0xf, true);
    };
    using std::integral_constant;
    v = dpp_add(v, integral_constant<int, 0xb1>{}, integral_constant<int, 0xf>{});
    v = dpp_add(v, integral_constant<int, 0x4e>{}, integral_constant<int, 0xf>{});
    v = dpp_add(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{});
    v = dpp_add(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{});
    v = dpp_add(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});
    v = dpp_add(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

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

// POL: 0 plain, 1 non-temporal, 2 (developer builds, GYMRS_EXP_SC1) written through at agent scope (`sc1`)
template <class T, int V, int POL>
__device__ __forceinline__ void store_vec(T* __restrict__ p, uint64_t base, uint64_t n, bool full, const Vec<T, V>& r)
{
    typedef T vt __attribute__((ext_vector_type(V)));
    if (full) {
        vt x;
#pragma unroll
        for (int k = 0; k < V; ++k) x[k] = r.v[k];
        if constexpr (POL == 2) {
            static_assert(sizeof(vt) == 4 || sizeof(vt) == 16, "sc1 stores: 4 lanes per work-item only");
            if constexpr (sizeof(vt) == 16) {
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(reinterpret_cast<vt*>(p + base)), "v"(__builtin_bit_cast(u4, x)) : "memory");
            }
            else
                asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" ::"v"(reinterpret_cast<vt*>(p + base)), "v"(__builtin_bit_cast(uint32_t, x)) : "memory");
        } else if constexpr (POL == 1)
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
    static constexpr int kSampled = 4; // state words reset() draws (cartpole.rs:317-324)
    static constexpr bool kConstReward = true;    // under auto-reset every step pays 1.0 (cartpole.rs:455-459)
    static constexpr float kReward = 1.0f;
    static constexpr bool kElideConstReward = false; // not unconditionally: per engine, StepArgs::elide_reward (from 128 MiB per step on; at 2^20 lanes the flag load and the 4 B per lane are a wash)
#ifdef GYMRS_EXP_NO_RESET_LOG // (developer builds: ablation)
    static constexpr bool kUseResetLog = false;
#else
    static constexpr bool kUseResetLog = true;       // measured: 6.67 -> 6.44 us per 2^20-lane step (1 lane in 22 re-arms per step)
#endif
    static constexpr bool kHasBeyond = true;
    static constexpr bool kHasObsExtra = false;
    static constexpr bool kNeverTerminates = false;
    static constexpr int kThreads = 512; // work-items per workgroup of the per-step kernel at >= 2^20 lanes: 8 waves measured 2 % faster than 4
    static constexpr bool kLoadsAheadOfArguments = true; // step_kernel_body (round 6): 2^20 lanes 6.39-6.47 -> 6.14-6.26 us, five box / run pairs out of five
    __device__ static bool valid(Action a) { return a < 2; } // Discrete(2).contains, discrete.rs:14-19
    __device__ static void advance(const Consts& c, float* st, Action a, float& reward, bool& done)
    {
        done = cartpole_advance(c, st[0], st[1], st[2], st[3], a);
        reward = 1.0f; // the beyond-terminated case is applied by the caller when auto-reset is off
    }
    // branch-free variant, legal when fast_ok holds (|theta| <= pi/4: the polynomial needs no reduction)
    // A tile is on the fast path when every action is < kActions and the largest range_key of its lanes is <= kRangeMax.
    static constexpr uint32_t kActions = 2;
    static constexpr uint32_t kRangeMax = 0x3f490fdbu; // |theta| <= fl32(pi/4); a NaN's bits lie above every bound
    __device__ static uint32_t range_key(const float* st) { return f2u(st[2]) & 0x7fffffffu; }
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
    static constexpr int kSampled = 1; // state words reset() draws (velocity is set to 0, mountain_car.rs:162-167)
    static constexpr bool kConstReward = true; // -1.0 on every step (mountain_car.rs:423)
    static constexpr float kReward = -1.0f;
    static constexpr bool kElideConstReward = true;  // measured: 4.16 -> 3.95 us per 2^20-lane step
    static constexpr bool kUseResetLog = false;      // measured: 4.16 -> 4.24 .. 4.8 us (re-arms are rare: nearly every wave is
                                                     // quiet and pays only for the folding launches)
    static constexpr bool kHasBeyond = false;
    static constexpr bool kHasObsExtra = false;
    static constexpr bool kNeverTerminates = false;
    static constexpr int kThreads = 256; // (512 measured 1-2 % slower for this short kernel)
    static constexpr bool kLoadsAheadOfArguments = false; // (round 6: -4 % / level at 2^20 lanes, but +3 .. +4 % from 2^22 on, in either order of the new structure)
    __device__ static bool valid(Action a) { return a < 3; } // Discrete(3)
    __device__ static void advance(const Consts& c, float* st, Action a, float& reward, bool& done)
    {
        done = mountain_car_advance(c, st[0], st[1], a);
        reward = -1.0f;
    }
    static constexpr uint32_t kActions = 3;
    static constexpr uint32_t kRangeMax = 0x43480000u; // |3 * position| <= 200: f32 Cody-Waite range
    __device__ static uint32_t range_key(const float* st) { return f2u(3.0f * st[0]) & 0x7fffffffu; }
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
    static constexpr int kSampled = 2; // state words reset() draws (theta, theta_dot)
    static constexpr bool kConstReward = false;
    static constexpr float kReward = 0.0f; // unused
    static constexpr bool kElideConstReward = false;
    static constexpr bool kUseResetLog = false;
    static constexpr bool kHasBeyond = false;
    static constexpr bool kHasObsExtra = true;
    // No termination and no invalid actions: every lane's episode clock is the same, so the time limit is a
    // kernel argument (StepArgs::truncate_all) instead of a per-lane compare against a dense ep_start read.
    static constexpr bool kNeverTerminates = true;
    static constexpr int kThreads = 256;
    static constexpr bool kLoadsAheadOfArguments = true; // step_kernel_body (round 6): 2^20 lanes -9 % on four boxes; its BASELINE size 2^22: -2 .. -5 % on three boxes (both ring sizes), +6 % once
    __device__ static bool valid(Action) { return true; } // a Box action is clipped, never rejected
    __device__ static void advance(const Consts& c, float* st, Action a, float& reward, bool& done)
    {
        reward = pendulum_advance(c, st[0], st[1], a);
        done = false;
    }
    static constexpr uint32_t kActions = 0;            // a Box action is clipped, never rejected
    static constexpr uint32_t kRangeMax = 0x43480000u; // |theta| <= 200
    __device__ static uint32_t range_key(const float* st) { return f2u(st[0]) & 0x7fffffffu; }
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
    // Which accesses carry the hint when the launch asks for it.  The state arrays are the only ones the NEXT step reads
    // again; actions are read once, rewards / flags / Pendulum's (cos, sin) are written and never read by a step.
    // GYMRS_EXP_HINTS (developer builds, tools/devbuild.py) overrides the launch flag per class of access:
    // bit 0 state loads, bit 1 state stores, bit 2 action loads, bit 3 reward / done / truncated / obs stores.
#ifdef GYMRS_EXP_HINTS
    static constexpr bool NT_SL = (GYMRS_EXP_HINTS & 1) != 0, NT_SS = (GYMRS_EXP_HINTS & 2) != 0, NT_A = (GYMRS_EXP_HINTS & 4) != 0,
                          NT_O = (GYMRS_EXP_HINTS & 8) != 0;
#else
    static constexpr bool NT_SL = NT || (FLAGS & kFlagNtStateLoads) != 0, NT_SS = NT, NT_A = NT, NT_O = NT || (FLAGS & kFlagNtOut) != 0;
#endif
    // Store policies (store_vec): 0 plain, 1 non-temporal, 2 written through (`sc1`; developer builds: GYMRS_EXP_SC1 bit 1 = state
    // stores, bit 3 = reward / done / truncated / obs stores -- profiles/r04_visible_step_probe.log says why no product launch uses it)
#ifdef GYMRS_EXP_SC1
    static constexpr int POL_SS = (VEC == 4 && (GYMRS_EXP_SC1 & 2)) ? 2 : (NT_SS ? 1 : 0), POL_O = (VEC == 4 && (GYMRS_EXP_SC1 & 8)) ? 2 : (NT_O ? 1 : 0);
#else
    static constexpr int POL_SS = NT_SS ? 1 : 0, POL_O = NT_O ? 1 : 0;
#endif
    // Episode bookkeeping of the per-step kernel goes through the reset log (StepArgs::reset_log) when nothing in the
    // step needs ep_start itself: statistics on, no time limit, constant reward (return = +-length).
    static constexpr bool LOGGED = STATS && !TLIM && Env::kConstReward && Env::kUseResetLog;
    Vec<float, VEC> st[Env::kState];
    Vec<typename Env::Action, VEC> act;
    Vec<uint8_t, VEC> beyond;
    Vec<uint32_t, VEC> ep_start;
};

// ROLL = the fused multi-step kernel: actions are generated in registers and ep_start travels densely
// (loaded once, updated in registers, stored once) instead of by sparse stores from the reset workers.
template <class Env, int VEC, uint32_t FLAGS, bool FULL, bool ROLL = false>
__device__ __forceinline__ void load_tile(const StepArgs& a, uint64_t base, TileRegs<Env, VEC, FLAGS>& d)
{
    constexpr int kVec = VEC;
    using R = TileRegs<Env, VEC, FLAGS>;
    using Action = typename Env::Action;
#pragma unroll
    for (int j = 0; j < Env::kState; ++j) d.st[j] = load_vec<float, kVec, R::NT_SL>(a.s[j], base, a.n, FULL, 0.0f);
    if (!ROLL) d.act = load_vec<Action, kVec, R::NT_A>(static_cast<const Action*>(a.action), base, a.n, FULL, Action(0));
    if (Env::kHasBeyond && !R::AUTO) d.beyond = load_vec<uint8_t, kVec, R::NT_SL>(a.beyond, base, a.n, FULL, uint8_t(0));
    if (ROLL ? (R::STATS || R::TLIM) : (R::TLIM && !Env::kNeverTerminates))
        d.ep_start = load_vec<uint32_t, kVec, false>(a.ep_start, base, a.n, FULL, 0u);
}

// LDS of one workgroup for the auto-reset hand-off: every wavefront uses its own 256-entry segment
// (list of finished lanes, their fresh states); waves never touch each other's.
template <class Env, int VEC, int THREADS = Env::kThreads>
struct ResetLds {
    static constexpr int kLanes = THREADS * VEC;
    uint16_t list[kLanes];
    struct alignas(Env::kState * 4) State {
        float v[Env::kState];
    };
    State fresh[kLanes]; // one ds_write/ds_read of 8 or 16 bytes per finished lane
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

// Every action of a work-item's packed u8 actions is < N (Discrete(N).contains, discrete.rs:14-19), tested on the dwords:
// N = 2: no byte has a bit above bit 0; N = 3: no byte has a bit above bit 1 and none is 0b11.  N = 0: nothing to test.
template <uint32_t N, class A, int VEC>
__device__ __forceinline__ bool actions_all_below(const Vec<A, VEC>& act)
{
    if constexpr (N == 0 || sizeof(A) != 1) {
        return true;
    } else {
        static_assert(N == 2 || N == 3, "only Discrete(2) and Discrete(3) are packed here");
        struct Words {
            uint32_t w[VEC / 4];
        };
        const Words words = __builtin_bit_cast(Words, act);
        uint32_t bad = 0;
#pragma unroll
        for (int i = 0; i < VEC / 4; ++i) {
            const uint32_t x = words.w[i];
            bad |= N == 2 ? (x & 0xfefefefeu) : ((x & 0xfcfcfcfcu) | (x & (x >> 1) & 0x01010101u));
        }
        return bad == 0;
    }
}

// What one step produces besides the new state in TileRegs.
template <int VEC>
struct StepOut {
    Vec<float, VEC> reward;
    Vec<uint8_t, VEC> done, trunc;
    bool reward_is_const; // wave-uniform: every stepped lane of the wave earned Env::kReward (constant-reward envs)
    unsigned long long masks[VEC]; // wave-uniform: bit i of masks[k] = work-item i re-armed its lane k in this step
};

// One Env::step() of a tile held in registers: physics + auto-reset of the finished lanes.  The per-step
// kernel calls it once between load_tile and store_tile; the fused rollout kernel calls it in a loop.
// `resets`/`ret` are the wave's statistics slot values: the per-step kernel stores the updated slot from
// here, the rollout kernel (ROLL) keeps accumulating in registers and stores once at the end.
// `open` (envs with a non-constant reward, i.e. Pendulum) is the wave's sum of the rewards of its OPEN episodes:
// such an env never terminates, so all lanes share one episode clock, finish together, and the return of the
// finished episodes of a wave is just that sum -- no per-lane return accumulator in HBM (which cost 8 B per
// lane-step: 29.6 vs 23.4 us per 2^22-lane step).  The caller loads/stores `open` (StepArgs::wave_open).
// a.fold_step: a folding launch of a reset-logged per-step kernel (step_block): the step's done-masks stay in registers.
// vblock: the index of the workgroup-sized tile (THREADS * VEC lanes) this is -- blockIdx.x, unless a workgroup steps several tiles.
template <class Env, int VEC, uint32_t FLAGS, bool FULL, bool ROLL = false, int THREADS = Env::kThreads>
__device__ __forceinline__ void advance_tile(const StepArgs& a, const typename Env::Consts& c, uint64_t base,
                                             TileRegs<Env, VEC, FLAGS>& d, ResetLds<Env, VEC, THREADS>& lds, unsigned long long& resets,
                                             double& ret, double& open, StepOut<VEC>& out, uint32_t vblock)
{
    static_assert(Env::kConstReward || Env::kNeverTerminates,
                  "return tracking assumes a constant reward (return = +-length) or one shared episode clock");
    constexpr int kVec = VEC;
    using R = TileRegs<Env, VEC, FLAGS>;
    constexpr bool AUTO = R::AUTO, STATS = R::STATS, TLIM = R::TLIM;
    constexpr bool LOGGED = R::LOGGED && !ROLL; // the rollout kernel keeps ep_start and its counters in registers
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
    // One compare for the tile's range (max of the lanes' |angle| bit patterns) and one for its packed actions, instead of
    // two compares per lane.
    uint32_t key = 0;
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
        float lane_st[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) lane_st[j] = ls[j][k];
        const uint32_t kk = Env::range_key(lane_st);
        key = key > kk ? key : kk;
    }
    const bool fast = FULL && key <= Env::kRangeMax && actions_all_below<Env::kActions>(d.act);
    out.reward_is_const = true; // the fast path pays the constant on every lane
#pragma unroll
    for (int k = 0; k < kVec; ++k) out.masks[k] = 0;
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
            *a.err_seen = 1u;
        }
        out.reward_is_const = __all(n_bad == 0); // an invalid action leaves the lane untouched and pays 0
    }
    GYMRS_STAMP(3); // physics done
    Vec<uint8_t, kVec>&done = out.done, &trunc = out.trunc;
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
        done.v[k] = dn[k] ? 1 : 0;
        trunc.v[k] = tr[k] ? 1 : 0;
        need_reset[k] = AUTO && (dn[k] || tr[k]);
    }
    if (STATS && !Env::kConstReward) { // this step's rewards join the wave's open-episode sum (lanes not stepped have 0)
        float rs = 0.0f;
#pragma unroll
        for (int k = 0; k < kVec; ++k) rs += rw[k];
        open += (double)wave_sum(rs);
    }

#ifdef GYMRS_EXP_EARLY_OUT // (developer builds: the step's reward / done / truncated leave BEFORE the re-arm pass; profiles/r04_wave_variants.log)
    if (!ROLL && !Env::kElideConstReward) {
#pragma unroll
        for (int k = 0; k < kVec; ++k) out.reward.v[k] = rw[k];
        store_vec<float, kVec, R::POL_O>(a.reward, base, a.n, FULL, out.reward);
        if (!Env::kNeverTerminates) store_vec<uint8_t, kVec, R::POL_O>(a.done, base, a.n, FULL, out.done);
        if (TLIM && !(Env::kNeverTerminates && a.skip_trunc_store)) store_vec<uint8_t, kVec, R::POL_O>(a.truncated, base, a.n, FULL, out.trunc);
    }
#endif
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
            out.masks[k] = m;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            slot[k] = total + rank;
            if (need_reset[k]) {
                list[slot[k]] = (uint16_t)(lane * kVec + k); // wave-local lane
            }
            total += (uint32_t)__popcll(m);
        }
        if (total != 0) { // quiet waves (MountainCar / Pendulum: nearly all) skip everything below
            if (LOGGED && a.fold_step == 0 && lane == 0) {
                // The whole episode bookkeeping of this step for the wave's 64 * VEC lanes: 8 * VEC contiguous bytes.  (Measured at
                // 2^20 CartPole lanes: the scattered ep_start stores cost 0.45 us per launch -- ~47k partial cache lines -- and the
                // counter's load + store 0.07; this row entry costs 0.02.)  A folding launch consumes its masks from registers.
                typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
                const size_t wave_slot = (size_t)vblock * (THREADS / 64) + wave;
                ull2* row = reinterpret_cast<ull2*>(a.reset_log + (size_t)((uint32_t)a.tick & (kResetLogRows - 1u)) * a.reset_log_row_words +
                                                    wave_slot * kVec);
#pragma unroll
                for (int k = 0; k < kVec; k += 2) row[k / 2] = ull2{out.masks[k], out.masks[k + 1]};
            }
            // DS operations of one wavefront execute in order: no barrier is needed, only the compiler
            // must not move LDS accesses across the hand-over points.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint64_t wave_base = base - (uint64_t)lane * kVec; // global lane of this wave's first lane
            for (uint32_t i = lane; i < total; i += 64u) { // one Philox4x32-10 block per finished lane
                const uint64_t gl = wave_base + list[i];
                const u32x4 r = draw4(a.seed, a.gid0 + gl, a.tick, kStreamReset);
                float ns[NS];
                Env::sample(r, a.box, ns);
                typename ResetLds<Env, VEC, THREADS>::State fs;
#pragma unroll
                for (int j = 0; j < NS; ++j) fs.v[j] = ns[j];
                lds.fresh[wave * LPW + i] = fs;
                if (!ROLL && !LOGGED && (STATS || TLIM)) a.ep_start[gl] = tick_next; // the new episode starts at the next tick (plain
                                                                          // store: a non-temporal scattered dword store measured slower)
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < kVec; ++k) {
                if (need_reset[k]) {
                    const typename ResetLds<Env, VEC, THREADS>::State fs = lds.fresh[wave * LPW + slot[k]];
#pragma unroll
                    for (int j = 0; j < NS; ++j) ls[j][k] = fs.v[j];
                    if (ROLL && (STATS || TLIM)) d.ep_start.v[k] = tick_next;
                }
            }
            if (STATS && !LOGGED) { // the wave's private statistics slot: plain read-modify-write, no atomics
                unsigned long long* bs = a.block_stats + ((size_t)vblock * (THREADS / 64) + wave) * 2;
                if (!Env::kConstReward) { // every lane of the wave finished (shared clock): the open sum is their return
                    ret += open;
                    open = 0.0;
                    if (!ROLL && lane == 0) reinterpret_cast<double*>(bs)[1] = ret;
                }
                resets += total;
                if (!ROLL && lane == 0) bs[0] = resets;
            }
        }
    }
    GYMRS_STAMP(4); // auto-reset done
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
        out.reward.v[k] = rw[k];
#pragma unroll
        for (int j = 0; j < NS; ++j) d.st[j].v[k] = ls[j][k];
    }
}

// The stores of one tile: new state (= observation), the step's reward/done/truncated, Pendulum's (cos, sin).
template <class Env, int VEC, uint32_t FLAGS, bool FULL, bool ROLL = false>
__device__ __forceinline__ void store_tile(const StepArgs& a, uint64_t base, const TileRegs<Env, VEC, FLAGS>& d,
                                           const StepOut<VEC>& out, bool skip_reward = false)
{
    constexpr int kVec = VEC;
    using R = TileRegs<Env, VEC, FLAGS>;
    constexpr bool AUTO = R::AUTO, STATS = R::STATS, TLIM = R::TLIM;
#pragma unroll
    for (int j = 0; j < Env::kState; ++j) store_vec<float, kVec, R::POL_SS>(a.s[j], base, a.n, FULL, d.st[j]);
#ifdef GYMRS_EXP_EARLY_OUT
    constexpr bool kOutStored = !ROLL && !Env::kElideConstReward; // (advance_tile stored them already)
#else
    constexpr bool kOutStored = false;
#endif
    if (!skip_reward && !kOutStored) store_vec<float, kVec, R::POL_O>(a.reward, base, a.n, FULL, out.reward);
    // An env that never terminates never changes `done` (reset() zeroed it), and its `truncated` flag is the same
    // for every lane: neither is rewritten while it already holds the right value (2 of Pendulum's 34 real bytes).
    if (!Env::kNeverTerminates && !kOutStored) store_vec<uint8_t, kVec, R::POL_O>(a.done, base, a.n, FULL, out.done);
    if (TLIM && !kOutStored && !(Env::kNeverTerminates && a.skip_trunc_store)) store_vec<uint8_t, kVec, R::POL_O>(a.truncated, base, a.n, FULL, out.trunc);
    if (Env::kHasBeyond && !AUTO) store_vec<uint8_t, kVec, R::POL_SS>(a.beyond, base, a.n, FULL, d.beyond);
    if (ROLL && (STATS || TLIM)) store_vec<uint32_t, kVec, 0>(a.ep_start, base, a.n, FULL, d.ep_start);
    if (Env::kHasObsExtra) {
        Vec<float, kVec> oc, os;
        bool med = true;
#pragma unroll
        for (int k = 0; k < kVec; ++k) med = med && in_short_range(d.st[0].v[k]);
        if (__all(med)) {
#pragma unroll
            for (int k = 0; k < kVec; ++k) sincos_short(d.st[0].v[k], &os.v[k], &oc.v[k]);
        } else {
#pragma unroll
            for (int k = 0; k < kVec; ++k) sincosf_(d.st[0].v[k], &os.v[k], &oc.v[k]);
        }
        store_vec<float, kVec, R::POL_O>(a.obs_cos, base, a.n, FULL, oc);
        store_vec<float, kVec, R::POL_O>(a.obs_sin, base, a.n, FULL, os);
    }
    GYMRS_STAMP(5); // stores issued
#ifdef GYMRS_TRACE_TIMES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}


} // namespace gymrs
