// gymrs_aux.hip -- the kernels OFF the per-step path: Env::reset for every lane, the synthetic random-policy action
// stream, the statistics read-out, and two one-thread helpers.  Kept in their own translation unit so that the
// step-kernel tables (gymrs_step_<env>.hip) are not rebuilt when one of these changes.
#include "gymrs_tile.h"
#include "gymrs_pcg64.h"

namespace gymrs {

// ---------------------------------------------------------------------------------------------
// Env::reset for every lane (cartpole.rs:485-516, mountain_car.rs:464-501): off the per-step path.
template <class Env>
__global__ __launch_bounds__(kBlock) void reset_kernel(const ResetArgs a)
{
    const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (lane >= a.n) return;
    float ns[Env::kState];
    if (a.pcg64) { // wave-uniform: the reference's own generator, one per lane (gymrs_pcg64.h)
        Pcg64Lane g = Pcg64Lane::seed_from_u64(a.pcg_seeds ? a.pcg_seeds[lane] : a.seed + a.gid0 + lane);
#pragma unroll
        for (int j = 0; j < Env::kState; ++j)
            ns[j] = j < Env::kSampled ? (float)pcg64_uniform(g, a.pcg_low[j], a.pcg_scale[j]) : 0.0f;
    } else {
        const u32x4 r = draw4(a.seed, a.gid0 + lane, a.tick, kStreamReset);
        Env::sample(r, a.box, ns);
    }
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
}

// Random-policy actions (examples/cartpole.rs:19 `rng.gen_range(0..=1)`), Philox stream 1.  A work-item serves one
// aligned group of four global env ids: they share one Philox block (gymrs_philox.h).
template <class Env>
__global__ __launch_bounds__(kBlock) void fill_actions_kernel(typename Env::Action* out, uint64_t n, uint64_t gid0,
                                                              uint64_t seed, uint64_t t, uint32_t n_actions,
                                                              float max_torque)
{
    constexpr bool kDiscrete = sizeof(typename Env::Action) == 1;
    const uint64_t group = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t first_gid = (gid0 & ~3ull) + 4 * group;
    if (first_gid >= gid0 + n) return;
    const u32x4 blk = action_block(seed, first_gid, kDiscrete ? (t >> 1) : t);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint64_t gid = first_gid + j;
        if (gid < gid0 || gid >= gid0 + n) continue; // the shard need not start or end on a multiple of 4
        if constexpr (kDiscrete)
            out[gid - gid0] = discrete_from_word(blk.v[j], t, n_actions);
        else
            out[gid - gid0] = uniform_between(blk.v[j], -max_torque, max_torque);
    }
}

// ---------------------------------------------------------------------------------------------
// Statistics read-out (off the hot path; one call per gymrs_stats / gymrs_stats_clear / all-reduce).
//   L = sum over lanes of (ep_start - epoch) = total length of the episodes finished since the last reset()
//       (episodes tile a lane's time axis, so the start tick of the open episode is all a lane has to keep),
//   E = finished episodes, R = sum of returns (Pendulum only; for the constant-reward envs return = +-length).
// Read-only towards everything a step kernel touches: rows of the reset log that are not folded yet are counted where they lie (see_through_log).
// Two launches, no atomics, no memset: stats_partial_kernel streams ep_start (one dwordx4 per work-item per pass) and
// the per-wavefront slots, reduces inside the wavefront with shuffles, across the 16 wavefronts of a workgroup
// through 384 bytes of LDS, and writes ONE {L, E, R} triple per workgroup; stats_finalize_kernel (one workgroup)
// folds the <= 256 triples in a fixed order -- so R, a double, is reproducible from call to call -- and applies the
// read / clear / after-reset logic.  2^20 lanes: 256 workgroups x 1024 work-items, one pass.
// The loads are PLAIN on purpose.  Measured: a non-temporal sweep leaves ep_start outside the Infinity Cache, and
// every following step launch then pays 0.36 us more (6.79 -> 7.15 us at 2^20 CartPole lanes) for its ~47k scattered
// 4-byte ep_start stores until something brings the array back.
// (Round 1 used <= 1024 workgroups, three LDS tree reductions and three same-address atomics per workgroup: 40 us.)
constexpr int kStatsThreads = 1024;
constexpr int kStatsMaxBlocks = kStatsPartials;

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Workgroup-wide {len, ep, ret} -> valid in work-item 0.  WAVES = wavefronts per workgroup (<= 64).
template <int WAVES>
__device__ __forceinline__ void block_sum3(unsigned long long& len, unsigned long long& ep, double& ret)
{
    __shared__ unsigned long long s_len[WAVES], s_ep[WAVES];
    __shared__ double s_ret[WAVES];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    len = wave_sum_u64(len);
    ep = wave_sum_u64(ep);
    ret = wave_sum_f64(ret);
    if (lane == 0) {
        s_len[wave] = len;
        s_ep[wave] = ep;
        s_ret[wave] = ret;
    }
    __syncthreads();
    if (wave == 0) {
        len = lane < WAVES ? s_len[lane] : 0ull;
        ep = lane < WAVES ? s_ep[lane] : 0ull;
        ret = lane < WAVES ? s_ret[lane] : 0.0;
        len = wave_sum_u64(len);
        ep = wave_sum_u64(ep);
        ret = wave_sum_f64(ret);
    }
}

// The reset log's pending rows as the read-out sees them (StatsArgs::log): passed by value, wave-uniform.
struct PendingLog {
    const unsigned long long* log;
    uint32_t row_words, pending;
    uint64_t first_tick;
    uint32_t vec;
};

// Lanes l0 .. l0 + count - 1 (l0 a multiple of 4, count <= 4; all in one work-item's group of the launches that wrote the rows): start ticks as a fold
// of the pending rows WOULD leave them -- the most recent re-arm of a lane wins -- and, for the group that holds bit 0 of its words, the number of bits
// (= finished episodes) of those words over all pending rows.  Read-only.
__device__ __forceinline__ void see_through_log(const PendingLog& lg, uint64_t l0, uint32_t count, uint32_t* ep4, unsigned long long& episodes)
{
    const uint64_t lanes_per_wave = 64ull * lg.vec;
    const uint64_t wave = l0 / lanes_per_wave;
    const uint32_t within = (uint32_t)(l0 % lanes_per_wave), bit = within / lg.vec, k0 = within % lg.vec;
    const uint64_t word0 = wave * lg.vec + k0;
    if (word0 + count > lg.row_words) return; // (cannot happen: the ring has a word for every group of the batch)
    uint32_t seen = 0;
    for (uint32_t i = 0; i < lg.pending; ++i) { // newest first
        const uint64_t t = lg.first_tick + lg.pending - 1 - i;
        const unsigned long long* row = lg.log + (size_t)((uint32_t)t & (kResetLogRows - 1u)) * lg.row_words + word0;
        for (uint32_t j = 0; j < count; ++j) {
            const unsigned long long w = row[j];
            if (bit == 0) episodes += (unsigned long long)__popcll(w);
            if (((w >> bit) & 1ull) && !((seen >> j) & 1u)) {
                seen |= 1u << j;
                ep4[j] = (uint32_t)(t + 1); // the next episode started at the next tick
            }
        }
    }
}

__global__ __launch_bounds__(kStatsThreads) void stats_partial_kernel(const uint32_t* __restrict__ ep_start, uint64_t n, uint32_t epoch,
                                                                      const unsigned long long* __restrict__ bs, uint32_t n_slots,
                                                                      unsigned long long* __restrict__ partials, const PendingLog lg)
{
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    unsigned long long len = 0, ep = 0;
    double ret = 0.0;
    const uint64_t tid = (uint64_t)blockIdx.x * kStatsThreads + threadIdx.x, stride = (uint64_t)gridDim.x * kStatsThreads;
    const uint64_t n4 = n >> 2; // ep_start is 256-byte aligned (engine_create)
    const u4* v = reinterpret_cast<const u4*>(ep_start);
    const bool through = lg.pending != 0; // wave-uniform
    for (uint64_t i = tid; i < n4; i += stride) {
        const u4 x = v[i]; // plain, not non-temporal: see above
        uint32_t e4[4] = {x.x, x.y, x.z, x.w};
        if (through) see_through_log(lg, i << 2, 4u, e4, ep);
        // each difference is taken in 32 bits (ticks wrap), the sum in 64
        len += (unsigned long long)(uint32_t)(e4[0] - epoch) + (uint32_t)(e4[1] - epoch) + (uint32_t)(e4[2] - epoch) + (uint32_t)(e4[3] - epoch);
    }
    if ((n & 3u) != 0 && tid == 0) { // the ragged last group of up to three lanes
        uint32_t e4[4] = {0, 0, 0, 0};
        const uint32_t count = (uint32_t)(n & 3u);
        for (uint32_t j = 0; j < count; ++j) e4[j] = ep_start[(n4 << 2) + j];
        if (through) see_through_log(lg, n4 << 2, count, e4, ep);
        for (uint32_t j = 0; j < count; ++j) len += (uint32_t)(e4[j] - epoch);
    }
    // The per-wavefront slots are read AROUND the caches (system-scope loads): each slot has one writer -- its wavefront, in the stream's launches or in a chain
    // on the engine's own queue -- and this kernel may run on any XCD.
    for (uint64_t b = tid; b < n_slots; b += stride) {
        ep += __hip_atomic_load(bs + b * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ret += __builtin_bit_cast(double, __hip_atomic_load(bs + b * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    }
    block_sum3<kStatsThreads / 64>(len, ep, ret);
    if (threadIdx.x == 0) {
        unsigned long long* out = partials + (size_t)blockIdx.x * 3;
        out[0] = len;
        out[1] = ep;
        out[2] = __builtin_bit_cast(unsigned long long, ret);
    }
}

// Age of the oldest open episode (see launch_max_age).  Like the statistics read-out: two launches, no atomics (4096
// same-address atomics, one per wavefront, made a first version take 30-80 us).  max_age_partial_kernel streams ep_start
// with plain loads (the step kernels' scattered stores want the array in the Infinity Cache) and leaves one maximum per
// workgroup; max_age_finalize_kernel (one workgroup) takes the maximum of those and hands it to the host through mapped
// host memory -- no copy engine in the stream: a 4-byte device-to-host copy costs the compute queue tens of
// microseconds of cross-engine synchronisation.  ~4 + 3 us at 2^20 lanes, a few times per time limit.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(kStatsThreads) void max_age_partial_kernel(const uint32_t* __restrict__ ep_start, uint64_t n, uint32_t tick_ref,
                                                                        uint32_t* __restrict__ partials)
{
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    __shared__ uint32_t s_age[kStatsThreads / 64];
    const uint64_t tid = (uint64_t)blockIdx.x * kStatsThreads + threadIdx.x, stride = (uint64_t)gridDim.x * kStatsThreads;
    const uint64_t n4 = n >> 2;
    const u4* v = reinterpret_cast<const u4*>(ep_start);
    uint32_t age = 0;
    auto upd = [&](uint32_t start) {
        const uint32_t a = tick_ref - start; // 32-bit ticks wrap; ages do not reach 2^32
        age = a > age ? a : age;
    };
    for (uint64_t i = tid; i < n4; i += stride) {
        const u4 x = v[i];
        upd(x.x);
        upd(x.y);
        upd(x.z);
        upd(x.w);
    }
    for (uint64_t i = (n4 << 2) + tid; i < n; i += stride) upd(ep_start[i]);
    age = wave_max_u32(age);
    if ((threadIdx.x & 63u) == 0) s_age[threadIdx.x >> 6] = age;
    __syncthreads();
    if (threadIdx.x < 64) {
        age = wave_max_u32(threadIdx.x < kStatsThreads / 64 ? s_age[threadIdx.x] : 0u);
        if (threadIdx.x == 0) partials[blockIdx.x] = age;
    }
}

__global__ __launch_bounds__(kStatsMaxBlocks) void max_age_finalize_kernel(const uint32_t* __restrict__ partials, uint32_t n_partials,
                                                                           volatile uint32_t* host_out2, uint32_t seq)
{
    __shared__ uint32_t s_age[kStatsMaxBlocks / 64];
    uint32_t age = threadIdx.x < n_partials ? partials[threadIdx.x] : 0u;
    age = wave_max_u32(age);
    if ((threadIdx.x & 63u) == 0) s_age[threadIdx.x >> 6] = age;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kStatsMaxBlocks / 64; ++w) age = s_age[w] > age ? s_age[w] : age;
        host_out2[0] = age;
        __threadfence_system();
        host_out2[1] = seq; // the host reads the age only after it has seen the sequence number
        __threadfence_system();
    }
}

hipError_t launch_max_age(const uint32_t* ep_start, uint64_t n, uint32_t tick_ref, uint32_t* partials, uint32_t* host_out2, uint32_t seq,
                          hipStream_t stream)
{
    launch_begin();
    const uint64_t want = (n / 4 + kStatsThreads - 1) / kStatsThreads;
    const uint32_t grid = (uint32_t)(want < 1 ? 1 : (want > (uint64_t)kStatsMaxBlocks ? (uint64_t)kStatsMaxBlocks : want));
    hipLaunchKernelGGL(max_age_partial_kernel, dim3(grid), dim3(kStatsThreads), 0, stream, ep_start, n, tick_ref, partials);
    hipLaunchKernelGGL(max_age_finalize_kernel, dim3(1), dim3(kStatsMaxBlocks), 0, stream, partials, grid, host_out2, seq);
    return hipGetLastError();
}

// mode 0: out4 = {sum_return, sum_length, n_episodes, n_steps} since the baseline; mode 1: the totals now become the baseline
// (statistics cleared); mode 2: baseline = 0 (after reset(), which zeroed the counters itself; n_partials = 0).
__global__ __launch_bounds__(kStatsMaxBlocks) void stats_finalize_kernel(const unsigned long long* __restrict__ partials, uint32_t n_partials,
                                                                         unsigned long long* base, int mode, int reward_sign, double n_steps,
                                                                         double* out4, volatile double* host_out4)
{
    unsigned long long len = 0, ep = 0;
    double ret = 0.0;
    if (threadIdx.x < n_partials) {
        const unsigned long long* p = partials + (size_t)threadIdx.x * 3;
        len = p[0];
        ep = p[1];
        ret = __builtin_bit_cast(double, p[2]);
    }
    block_sum3<kStatsMaxBlocks / 64>(len, ep, ret);
    if (threadIdx.x != 0) return;
    if (mode == 1) {
        base[0] = len;
        base[1] = ep;
        base[2] = __builtin_bit_cast(unsigned long long, ret);
        return;
    }
    if (mode == 2) {
        base[0] = base[1] = 0;
        base[2] = __builtin_bit_cast(unsigned long long, 0.0);
        return;
    }
    const double flen = (double)(len - base[0]);
    const double r[4] = {reward_sign != 0 ? reward_sign * flen : ret - __builtin_bit_cast(double, base[2]), flen, (double)(ep - base[1]), n_steps};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        out4[j] = r[j];
        if (host_out4) host_out4[j] = r[j]; // mapped host memory: gymrs_stats reads it after the stream sync, no copy
    }
    if (host_out4) __threadfence_system();
}

hipError_t launch_stats(const StatsArgs& a, int mode, hipStream_t stream)
{
    launch_begin();
    uint32_t grid = 0;
    if (mode != 2 && a.track) { // without GYMRS_TRACK_STATS ep_start and the slots carry no statistics
        const uint64_t items = (a.n >> 2) > a.n_blocks ? (a.n >> 2) : a.n_blocks;
        const uint64_t want = (items + kStatsThreads - 1) / kStatsThreads;
        grid = (uint32_t)(want < 1 ? 1 : (want > (uint64_t)kStatsMaxBlocks ? (uint64_t)kStatsMaxBlocks : want));
        const PendingLog lg{a.log, a.log_row_words, a.log ? a.log_pending : 0u, a.log_first_tick, (uint32_t)(a.log_vec > 0 ? a.log_vec : 4)};
        hipLaunchKernelGGL(stats_partial_kernel, dim3(grid), dim3(kStatsThreads), 0, stream, a.ep_start, a.n, a.epoch, a.block_stats,
                           a.n_blocks, a.partials, lg);
    }
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(1), dim3(kStatsMaxBlocks), 0, stream, a.partials, grid, a.base, mode, a.reward_sign,
                       a.n_steps, a.out4, a.host_out4);
    return hipGetLastError();
}

__global__ void fold_open_kernel(double* wave_open, uint32_t n_slots)
{
    double sum = 0.0;
    for (uint32_t i = 0; i < n_slots; ++i) {
        sum += wave_open[i];
        wave_open[i] = 0.0;
    }
    wave_open[0] = sum;
}

hipError_t launch_fold_open(double* wave_open, uint32_t n_slots, hipStream_t stream)
{
    launch_begin();
    hipLaunchKernelGGL(fold_open_kernel, dim3(1), dim3(1), 0, stream, wave_open, n_slots);
    return hipGetLastError();
}

__global__ void tick_advance_kernel(unsigned long long* tick_dev, unsigned long long by) { *tick_dev += by; }

hipError_t launch_tick_advance(unsigned long long* tick_dev, unsigned long long by, hipStream_t stream)
{
    launch_begin();
    hipLaunchKernelGGL(tick_advance_kernel, dim3(1), dim3(1), 0, stream, tick_dev, by);
    return hipGetLastError();
}

hipError_t launch_reset(gymrs_env_kind kind, const ResetArgs& a, hipStream_t stream)
{
    launch_begin();
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
    launch_begin();
    if (n == 0) return hipSuccess;
    const uint64_t groups = (n + (gid0 & 3u) + 3) / 4; // aligned groups of four global ids that touch the shard
    const uint32_t grid = (uint32_t)((groups + kBlock - 1) / kBlock);
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

// ---------------------------------------------------------------------------------------------
// Reset-log fold on demand (off the hot path; the per-step kernel folds a FULL ring itself, see step_block).  The
// per-step kernel of a constant-reward env with statistics and no time limit records a step's re-armed lanes as
// done-masks: row (tick % kResetLogRows) of the ring, word (wave * VEC + k), bit = work-item of the wave, i.e.
// lane = wave * 64 * VEC + bit * VEC + k.  One work-item folds one word column: it walks the pending rows from the
// newest to the oldest, gives every lane whose bit it sees for the FIRST time its episode start (the tick after that
// row's step: episodes tile a lane's time axis, so only the most recent re-arm matters), counts all bits (= finished
// episodes) and zeroes what it read.  Runs before anything that reads ep_start or the counters (statistics, snapshot,
// clone, the rollout kernel) and when the launch shape changes.
constexpr int kFoldThreads = 64;

__global__ __launch_bounds__(kFoldThreads) void fold_reset_log_kernel(unsigned long long* __restrict__ log, uint32_t row_words,
                                                                      uint64_t first_tick, uint32_t pending, int vec,
                                                                      uint32_t* __restrict__ ep_start, uint64_t n,
                                                                      unsigned long long* __restrict__ block_stats)
{
    const uint32_t word = blockIdx.x * kFoldThreads + threadIdx.x;
    if (word >= row_words) return;
    const uint64_t lane0 = (uint64_t)(word / (uint32_t)vec) * (64u * (uint32_t)vec) + (word % (uint32_t)vec); // lane of bit 0
    if (lane0 >= n) return; // a column beyond the batch: the step kernel never writes there
    unsigned long long m[kResetLogRows];
#pragma unroll
    for (uint32_t i = 0; i < kResetLogRows; ++i) { // i steps back from the newest pending one; all loads in flight together
        const uint64_t t = first_tick + pending - 1 - i;
        m[i] = i < pending ? log[(size_t)((uint32_t)t & (kResetLogRows - 1u)) * row_words + word] : 0ull;
    }
    unsigned long long seen = 0, count = 0;
#pragma unroll
    for (uint32_t i = 0; i < kResetLogRows; ++i) {
        if (m[i] == 0) continue;
        const uint64_t t = first_tick + pending - 1 - i;
        log[(size_t)((uint32_t)t & (kResetLogRows - 1u)) * row_words + word] = 0; // the ring is all zero between folds
        count += (unsigned long long)__popcll(m[i]);
        unsigned long long fresh = m[i] & ~seen;
        seen |= m[i];
        while (fresh) {
            const uint32_t bit = (uint32_t)__ffsll((long long)fresh) - 1u;
            fresh &= fresh - 1;
            ep_start[lane0 + (uint64_t)bit * (uint32_t)vec] = (uint32_t)(t + 1); // the next episode started at the next tick
        }
    }
    if (count) atomicAdd(block_stats + (size_t)(word / 4u) * 2, count); // any slot will do: the read-out sums them all
}

hipError_t launch_fold_reset_log(unsigned long long* log, uint32_t row_words, uint64_t first_tick, uint32_t pending, int vec,
                                 uint32_t* ep_start, uint64_t n, unsigned long long* block_stats, hipStream_t stream)
{
    launch_begin();
    if (pending == 0) return hipSuccess;
    if (pending >= kResetLogRows) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fold_reset_log_kernel, dim3((row_words + kFoldThreads - 1) / kFoldThreads), dim3(kFoldThreads), 0, stream, log, row_words,
                       first_tick, pending, vec, ep_start, n, block_stats);
    return hipGetLastError();
}

// the per-step kernel tables live in one translation unit per env type
hipError_t launch_step_cartpole(int vec, uint32_t flags, const StepArgs& a, const void* consts, hipStream_t stream);
hipError_t launch_step_mountain_car(int vec, uint32_t flags, const StepArgs& a, const void* consts, hipStream_t stream);
hipError_t launch_step_pendulum(int vec, uint32_t flags, const StepArgs& a, const void* consts, hipStream_t stream);

hipError_t launch_step(gymrs_env_kind kind, int vec, uint32_t flags, const StepArgs& a, const void* consts,
                       hipStream_t stream)
{
    launch_begin();
    if (a.n == 0) return hipSuccess;
    switch (kind) {
    case GYMRS_CARTPOLE: return launch_step_cartpole(vec, flags, a, consts, stream);
    case GYMRS_MOUNTAIN_CAR: return launch_step_mountain_car(vec, flags, a, consts, stream);
    case GYMRS_PENDULUM: return launch_step_pendulum(vec, flags, a, consts, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace gymrs
