// gymrs_rollout.hip -- the fused multi-step (random-policy rollout) kernel, gfx950.
#include "gymrs_tile.h"

namespace gymrs {

// ---------------------------------------------------------------------------------------------
// Fused multi-step kernel: the caller loop of examples/cartpole.rs:15-30 -- draw a random
// action, step, reset on done, accumulate the return -- for every lane, n_steps iterations in ONE launch.
// State stays in registers between steps: HBM is touched once per launch, so this path is VALU-bound
// (physics + one Philox block per lane per 4 steps for the actions + the compacted reset pass), not
// HBM-bound.  Bit-identical to n_steps calls of gymrs_fill_actions + gymrs_step: it calls the same
// advance_tile, step after step, and leaves the arrays as the last of those steps would.
// REC (gymrs_rollout_record): additionally every step's observation, action, reward and flags are written to
// trajectory buffers -- what a random-policy data collection loop keeps.  Then HBM sees 22 B per CartPole
// lane-step (no state re-read, no launch per step) instead of 38 B + a launch + an action-generation kernel.
template <class Env, int VEC, uint32_t FLAGS, bool FULL, bool REC>
__device__ __forceinline__ void rollout_block(StepArgs a, const RolloutArgs& r, const typename Env::Consts& c,
                                              ResetLds<Env, VEC, kBlock>& lds)
{
    constexpr int kVec = VEC;
    using R = TileRegs<Env, VEC, FLAGS>;
    using Action = typename Env::Action;
    const uint64_t base = (uint64_t)blockIdx.x * (kBlock * kVec) + (uint64_t)threadIdx.x * kVec;
    R d;
    load_tile<Env, VEC, FLAGS, FULL, true>(a, base, d);
    unsigned long long resets = 0;
    double ret = 0.0, open = 0.0;
    const size_t wave_slot = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    unsigned long long* bs = a.block_stats + wave_slot * 2;
    if (R::STATS) {
        resets = bs[0];
        if (!Env::kConstReward) {
            ret = reinterpret_cast<const double*>(bs)[1];
            open = a.wave_open[wave_slot];
        }
    }
    StepOut<VEC> out;
    constexpr bool kDiscrete = sizeof(Action) == 1;
    constexpr int kBlocks = kVec / 4;     // a work-item's lanes are kVec consecutive global ids: kVec/4 aligned groups
    u32x4 blk[kBlocks];                    // of four lanes that share one action block (gymrs_philox.h) ...
    const uint64_t gid = a.gid0 + base;
    const bool aligned = (a.gid0 & 3u) == 0; // ... when the shard starts on a multiple of 4 (wave-uniform)
    const uint64_t tick0 = a.tick;
    uint64_t ustart = r.uniform_start;
    for (uint32_t k = 0; k < r.n_steps; ++k) {
        const uint64_t t = r.action_t0 + k;
        const uint64_t slot = kDiscrete ? (t >> 1) : t; // Discrete: a block serves two steps (16-bit halves)
        if (aligned) {
            if (!kDiscrete || k == 0 || (t & 1u) == 0) {
#pragma unroll
                for (int b = 0; b < kBlocks; ++b) blk[b] = action_block(r.action_seed, gid + 4 * b, slot);
            }
#pragma unroll
            for (int i = 0; i < kVec; ++i) {
                const uint32_t w = blk[i / 4].v[i % 4];
                if constexpr (kDiscrete)
                    d.act.v[i] = discrete_from_word(w, t, r.n_actions);
                else
                    d.act.v[i] = uniform_between(w, -r.max_torque, r.max_torque);
            }
        } else { // shard offset not a multiple of 4: every lane evaluates its own block (same values, slower)
#pragma unroll
            for (int i = 0; i < kVec; ++i) {
                const uint32_t w = action_word(r.action_seed, gid + i, slot);
                if constexpr (kDiscrete)
                    d.act.v[i] = discrete_from_word(w, t, r.n_actions);
                else
                    d.act.v[i] = uniform_between(w, -r.max_torque, r.max_torque);
            }
        }
        a.tick = tick0 + k;
        if (Env::kNeverTerminates && R::TLIM) { // the uniform episode clock of gymrs_engine.hip step_args()
            a.truncate_all = (a.tick + 1 - ustart >= c.max_steps) ? 1u : 0u;
            if (a.truncate_all && R::AUTO) ustart = a.tick + 1;
        }
        advance_tile<Env, VEC, FLAGS, FULL, true, kBlock>(a, c, base, d, lds, resets, ret, open, out, blockIdx.x);
        if constexpr (REC) {
            const uint64_t row = (uint64_t)k * r.rec_stride;
            constexpr int kObs = Env::kHasObsExtra ? 3 : Env::kState;
            float* obs = r.rec_obs + row * kObs;
            if constexpr (Env::kHasObsExtra) { // Pendulum: (cos, sin, theta_dot) as store_tile writes them
                Vec<float, kVec> oc, os;
#pragma unroll
                for (int i = 0; i < kVec; ++i) sincosf_(d.st[0].v[i], &os.v[i], &oc.v[i]);
                store_vec<float, kVec, true>(obs, base, a.n, FULL, oc);
                store_vec<float, kVec, true>(obs + r.rec_stride, base, a.n, FULL, os);
                store_vec<float, kVec, true>(obs + 2 * r.rec_stride, base, a.n, FULL, d.st[1]);
            } else {
#pragma unroll
                for (int j = 0; j < Env::kState; ++j) store_vec<float, kVec, true>(obs + j * r.rec_stride, base, a.n, FULL, d.st[j]);
            }
            store_vec<Action, kVec, true>(static_cast<Action*>(r.rec_action) + row, base, a.n, FULL, d.act);
            store_vec<float, kVec, true>(r.rec_reward + row, base, a.n, FULL, out.reward);
            store_vec<uint8_t, kVec, true>(r.rec_done + row, base, a.n, FULL, out.done);
            if (R::TLIM && r.rec_trunc) store_vec<uint8_t, kVec, true>(r.rec_trunc + row, base, a.n, FULL, out.trunc);
        }
    }
    store_tile<Env, VEC, FLAGS, FULL, true>(a, base, d, out);
    if (R::STATS && (threadIdx.x & 63u) == 0) {
        bs[0] = resets;
        if (!Env::kConstReward) {
            reinterpret_cast<double*>(bs)[1] = ret;
            a.wave_open[wave_slot] = open;
        }
    }
}

template <class Env, int VEC, uint32_t FLAGS, bool REC>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(16 / VEC, 16 / VEC))) void rollout_kernel(
    const StepArgs a, const RolloutArgs r, const typename Env::Consts c)
{
    constexpr int LPB = kBlock * VEC;
    __shared__ ResetLds<Env, VEC, kBlock> lds;
    if ((uint64_t)blockIdx.x * LPB + (uint64_t)((threadIdx.x >> 6) + 1) * (64 * VEC) <= a.n) // wave-uniform, see step_kernel
        rollout_block<Env, VEC, FLAGS, true, REC>(a, r, c, lds);
    else
        rollout_block<Env, VEC, FLAGS, false, REC>(a, r, c, lds);
}

template <class Env, int VEC, uint32_t FLAGS>
static hipError_t rollout_one(const StepArgs& a, const RolloutArgs& r, const void* consts, hipStream_t stream)
{
    launch_begin();
    if constexpr (VEC == 4) { // the recording variant exists at 4 lanes per work-item only
        if (r.rec_obs) {
            hipLaunchKernelGGL((rollout_kernel<Env, VEC, FLAGS, true>), dim3(step_grid(a.n, VEC)), dim3(kBlock), 0, stream, a, r,
                               *static_cast<const typename Env::Consts*>(consts));
            return hipGetLastError();
        }
    }
    if (r.rec_obs) return hipErrorInvalidValue;
    hipLaunchKernelGGL((rollout_kernel<Env, VEC, FLAGS, false>), dim3(step_grid(a.n, VEC)), dim3(kBlock), 0, stream, a, r,
                       *static_cast<const typename Env::Consts*>(consts));
    return hipGetLastError();
}

template <class Env, int VEC>
static hipError_t rollout_flags(uint32_t flags, const StepArgs& a, const RolloutArgs& r, const void* consts, hipStream_t stream)
{
    constexpr uint32_t A = GYMRS_AUTO_RESET, S = GYMRS_TRACK_STATS, T = GYMRS_TIME_LIMIT;
    if (!(flags & A)) flags &= ~S;
    switch (flags & (A | S | T)) {
    case 0: return rollout_one<Env, VEC, 0>(a, r, consts, stream);
    case A: return rollout_one<Env, VEC, A>(a, r, consts, stream);
    case A | S: return rollout_one<Env, VEC, A | S>(a, r, consts, stream);
    case T: return rollout_one<Env, VEC, T>(a, r, consts, stream);
    case A | T: return rollout_one<Env, VEC, A | T>(a, r, consts, stream);
    case A | S | T: return rollout_one<Env, VEC, A | S | T>(a, r, consts, stream);
    default: return hipErrorInvalidValue;
    }
}

template <class Env>
static hipError_t rollout_vec(int vec, uint32_t flags, const StepArgs& a, const RolloutArgs& r, const void* consts, hipStream_t stream)
{
    switch (vec) {
    case 4: return rollout_flags<Env, 4>(flags, a, r, consts, stream);
    case 8: return rollout_flags<Env, 8>(flags, a, r, consts, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_rollout(gymrs_env_kind kind, int vec, uint32_t flags, const StepArgs& a, const RolloutArgs& r,
                          const void* consts, hipStream_t stream)
{
    if (a.n == 0 || r.n_steps == 0) return hipSuccess;
    switch (kind) {
    case GYMRS_CARTPOLE: return rollout_vec<CartPoleT>(vec, flags, a, r, consts, stream);
    case GYMRS_MOUNTAIN_CAR: return rollout_vec<MountainCarT>(vec, flags, a, r, consts, stream);
    case GYMRS_PENDULUM: return rollout_vec<PendulumT>(vec, flags, a, r, consts, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace gymrs
