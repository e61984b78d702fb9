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
    __device__ static void sample(const u32x4& r, const float* lo, const float* hi, float* st)
    {
        pendulum_sample(r, lo, hi, st[0], st[1]);
    }
};

// ---------------------------------------------------------------------------------------------
// THE hot kernel: one Env::step() per lane, VEC lanes per work-item, 64*VEC lanes per wavefront.
//
// There is no workgroup barrier anywhere: a wavefront owns its lanes from load to store, so the
// 4 waves of a workgroup (and the workgroups of a CU) drift apart and one wave's arithmetic hides
// under another wave's memory traffic.
template <class Env, int VEC, uint32_t FLAGS, bool FULL>
__device__ __forceinline__ void step_tile(const StepArgs& a, const typename Env::Consts& c, uint16_t* s_list, float* s_new)
{
    constexpr bool AUTO = (FLAGS & GYMRS_AUTO_RESET) != 0;
    constexpr bool STATS = AUTO && (FLAGS & GYMRS_TRACK_STATS) != 0;
    constexpr bool TLIM = (FLAGS & GYMRS_TIME_LIMIT) != 0;
    constexpr int NS = Env::kState;
    constexpr int LPB = kBlock * VEC; // lanes per workgroup
    constexpr int LPW = 64 * VEC;     // lanes per wavefront
    using Action = typename Env::Action;

    const uint32_t tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * LPB + (uint64_t)tid * VEC;
    const uint32_t tick_next = (uint32_t)(a.tick + 1);

    // ---- loads (issued together; the first use waits) ----
    Vec<float, VEC> st[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) st[j] = load_vec<float, VEC>(a.s[j], base, a.n, FULL, 0.0f);
    const Vec<Action, VEC> act = load_vec<Action, VEC>(static_cast<const Action*>(a.action), base, a.n, FULL, Action(0));
    Vec<uint8_t, VEC> beyond;
    if (Env::kHasBeyond && !AUTO) beyond = load_vec<uint8_t, VEC>(a.beyond, base, a.n, FULL, uint8_t(0));
    Vec<uint32_t, VEC> ep_start;
    if (TLIM) ep_start = load_vec<uint32_t, VEC>(a.ep_start, base, a.n, FULL, 0u);
    Vec<float, VEC> ep_ret;
    if (STATS && !Env::kConstReward) ep_ret = load_vec<float, VEC>(a.ep_ret, base, a.n, FULL, 0.0f);

    // ---- physics ----
    Vec<float, VEC> reward;
    Vec<uint8_t, VEC> done, trunc;
    bool need_reset[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const bool live = FULL || (base + k < a.n);
        const Action ak = act.v[k];
        const bool ok = Env::valid(ak);
        float lane_st[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) lane_st[j] = st[j].v[k];
        float r = 0.0f;
        bool d = false, t = false;
        if (live && ok) {
            Env::advance(c, lane_st, ak, r, d);
            if (Env::kHasBeyond && !AUTO) {
                bool b = beyond.v[k] != 0;
                r = cartpole_reward(d, b);
                beyond.v[k] = b ? 1 : 0;
            }
            if (TLIM) t = (tick_next - ep_start.v[k]) >= c.max_steps;
            if (STATS && !Env::kConstReward) ep_ret.v[k] += r;
        } else if (live) { // invalid action: the reference panics before touching the env
            atomicAdd(&a.err[0], 1u);
            atomicMin(&a.err[1], (uint32_t)(base + k));
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) st[j].v[k] = lane_st[j];
        reward.v[k] = r;
        done.v[k] = d ? 1 : 0;
        trunc.v[k] = t ? 1 : 0;
        need_reset[k] = AUTO && (d || t);
    }

    // ---- auto-reset: wave __ballot done-mask -> LDS-staged Philox, all inside one wavefront ----
    if (AUTO) {
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
            const uint64_t wave_base = (uint64_t)blockIdx.x * LPB + (uint64_t)wave * LPW;
            for (uint32_t i = lane; i < total; i += 64u) { // one Philox4x32-10 per finished lane
                const uint64_t gl = wave_base + list[i];
                const u32x4 r = draw4(a.seed, a.gid0 + gl, a.tick, kStreamReset);
                float ns[NS];
                Env::sample(r, a.lo, a.hi, ns);
#pragma unroll
                for (int j = 0; j < NS; ++j) fresh[j * LPW + i] = ns[j];
                if (STATS || TLIM) a.ep_start[gl] = tick_next; // the new episode starts at the next tick
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float ret_sum = 0.0f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                if (need_reset[k]) {
#pragma unroll
                    for (int j = 0; j < NS; ++j) st[j].v[k] = fresh[j * LPW + slot[k]];
                    if (STATS && !Env::kConstReward) {
                        ret_sum += ep_ret.v[k];
                        ep_ret.v[k] = 0.0f;
                    }
                }
            }
            if (STATS) {
                // per-workgroup partials, fire-and-forget (no returned value, no same-address contention)
                unsigned long long* bs = a.block_stats + (size_t)blockIdx.x * 2;
                if (!Env::kConstReward) {
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) ret_sum += __shfl_xor(ret_sum, off);
                    if (lane == 0) atomicAdd(reinterpret_cast<double*>(bs + 1), (double)ret_sum);
                }
                if (lane == 0) atomicAdd(bs, (unsigned long long)total);
            }
        }
    }

    // ---- stores ----
#pragma unroll
    for (int j = 0; j < NS; ++j) store_vec<float, VEC>(a.s[j], base, a.n, FULL, st[j]);
    store_vec<float, VEC>(a.reward, base, a.n, FULL, reward);
    store_vec<uint8_t, VEC>(a.done, base, a.n, FULL, done);
    if (TLIM) store_vec<uint8_t, VEC>(a.truncated, base, a.n, FULL, trunc);
    if (Env::kHasBeyond && !AUTO) store_vec<uint8_t, VEC>(a.beyond, base, a.n, FULL, beyond);
    if (STATS && !Env::kConstReward) store_vec<float, VEC>(a.ep_ret, base, a.n, FULL, ep_ret);
    if (Env::kHasObsExtra) {
        Vec<float, VEC> oc, os;
#pragma unroll
        for (int k = 0; k < VEC; ++k) sincosf_(st[0].v[k], &os.v[k], &oc.v[k]);
        store_vec<float, VEC>(a.obs_cos, base, a.n, FULL, oc);
        store_vec<float, VEC>(a.obs_sin, base, a.n, FULL, os);
    }
}

template <class Env, int VEC, uint32_t FLAGS>
__global__ __launch_bounds__(kBlock) void step_kernel(const StepArgs a, const typename Env::Consts c)
{
    constexpr bool AUTO = (FLAGS & GYMRS_AUTO_RESET) != 0;
    constexpr int LPB = kBlock * VEC;
    __shared__ uint16_t s_list[AUTO ? LPB : 1];
    __shared__ float s_new[AUTO ? Env::kState * LPB : 1];
    // workgroup-uniform: every workgroup but the last runs the unguarded body
    if ((uint64_t)(blockIdx.x + 1) * LPB <= a.n)
        step_tile<Env, VEC, FLAGS, true>(a, c, s_list, s_new);
    else
        step_tile<Env, VEC, FLAGS, false>(a, c, s_list, s_new);
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
template <class Env, int VEC, uint32_t FLAGS>
static hipError_t launch_one(const StepArgs& a, const void* consts, hipStream_t stream)
{
    const uint32_t grid = step_grid(a.n, VEC);
    hipLaunchKernelGGL((step_kernel<Env, VEC, FLAGS>), dim3(grid), dim3(kBlock), 0, stream, a,
                       *static_cast<const typename Env::Consts*>(consts));
    return hipGetLastError();
}

template <class Env, int VEC>
static hipError_t launch_flags(uint32_t flags, const StepArgs& a, const void* consts, hipStream_t stream)
{
    constexpr uint32_t A = GYMRS_AUTO_RESET, S = GYMRS_TRACK_STATS, T = GYMRS_TIME_LIMIT;
    if (!(flags & A)) flags &= ~S; // statistics need auto-reset
    switch (flags & (A | S | T)) {
    case 0: return launch_one<Env, VEC, 0>(a, consts, stream);
    case A: return launch_one<Env, VEC, A>(a, consts, stream);
    case A | S: return launch_one<Env, VEC, A | S>(a, consts, stream);
    case T: return launch_one<Env, VEC, T>(a, consts, stream);
    case A | T: return launch_one<Env, VEC, A | T>(a, consts, stream);
    case A | S | T: return launch_one<Env, VEC, A | S | T>(a, consts, stream);
    default: return hipErrorInvalidValue;
    }
}

template <class Env>
static hipError_t launch_vec(int vec, uint32_t flags, const StepArgs& a, const void* consts, hipStream_t stream)
{
    switch (vec) {
    case 1: return launch_flags<Env, 1>(flags, a, consts, stream);
    case 2: return launch_flags<Env, 2>(flags, a, consts, stream);
    case 4: return launch_flags<Env, 4>(flags, a, consts, stream);
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
