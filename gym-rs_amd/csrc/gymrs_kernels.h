// gymrs_kernels.h — launch interface between the engine (gymrs_engine.hip) and the gfx950 kernels
// (gymrs_step_<env>.hip, gymrs_rollout.hip, gymrs_aux.hip).  Plain structs passed by value as kernel arguments (uniform -> SGPRs).
#pragma once
#include <hip/hip_runtime.h>

#include "gymrs_physics.h"

namespace gymrs {

#ifdef GYMRS_EXP_BLOCK // (developer builds, HIP launches only: the workgroup size of every kernel that uses kBlock; profiles/r04_wave_variants.log)
constexpr int kBlock = GYMRS_EXP_BLOCK;
#else
constexpr int kBlock = 256; // work-items per workgroup of the small kernels (reset, fill, statistics) and of the rollout kernel
#endif
constexpr uint32_t kFlagNonTemporal = 0x100u; // internal launch flag (not an engine flag): non-temporal loads/stores
// Hints per class of access, for the launches of a chain (gymrs_aql.h): only the stores nobody reads again (reward, done,
// truncated, Pendulum's cos / sin), and additionally the state LOADS (the state stores stay plain: the next launch reads them
// out of the L2).  Measured per size in profiles/r03_chain_hints.log; the engine picks (chain_hint_bits).
constexpr uint32_t kFlagNtOut = 0x200u;
constexpr uint32_t kFlagNtStateLoads = 0x400u;
constexpr uint32_t kFlagHintMask = kFlagNonTemporal | kFlagNtOut | kFlagNtStateLoads;
// Rows of the reset log.  Measured at 2^20 CartPole lanes (128 KiB per row, no folding): a ring of <= 8 rows stays where a
// write is cheap (6.10 us per launch; 8 steps x 40 MB is about what passes through the 256 MB Infinity Cache before a line
// is evicted), 16 rows 6.22, >= 32 rows 6.40 (touching the next row one step ahead with a scalar load made it worse:
// 7.1); the scattered ep_start stores it replaces: 6.65.  Must be a power of two; the folding launch is written for 8.
constexpr uint32_t kResetLogRows = 8;

// Everything one step() launch needs.  Device pointers are SoA arrays of n lanes.
struct StepArgs {
    float* s[4];       // state: CartPole x,x_dot,theta,theta_dot | MountainCar position,velocity | Pendulum theta,theta_dot
    float* obs_cos;    // Pendulum only
    float* obs_sin;    // Pendulum only
    const void* action; // u8 (CartPole, MountainCar) or f32 (Pendulum)
    float* reward;
    uint8_t* done;
    uint8_t* truncated; // written only with GYMRS_TIME_LIMIT
    uint8_t* beyond;    // CartPole without auto-reset: steps_beyond_terminated.is_some()
    uint32_t* ep_start; // tick at which the lane's current episode started (low 32 bits)
    uint32_t* wave_clean; // [n_waves] constant-reward envs under auto-reset that elide the reward store: != 0 = the wave's part of `reward` holds the constant
    double* wave_open;  // [n_waves] Pendulum with GYMRS_TRACK_STATS: per-wavefront sum of the rewards of the open episodes
    unsigned long long* block_stats; // [n_waves][2] per-wavefront slots: finished episodes, sum of returns (f64 bits; Pendulum only)
    // Reset log (constant-reward envs with GYMRS_TRACK_STATS and without GYMRS_TIME_LIMIT): instead of one scattered 4-byte
    // ep_start store per re-armed lane and a read-modify-write of the wavefront's episode counter on every step, a
    // wavefront that re-armed lanes stores its VEC done-masks (one bit per lane) into row (tick % kResetLogRows) of a small
    // ring, and every kResetLogRows-th launch (fold_step) each wavefront folds its own column of the ring into ep_start and
    // its episode counter.  See step_block (gymrs_step_impl.h) and launch_fold_reset_log.
    unsigned long long* reset_log; // [kResetLogRows][reset_log_row_words]; word (wave * VEC + k), bit = work-item of the wave
    uint32_t reset_log_row_words;
    uint32_t fold_step;            // this launch folds the ring (wave-uniform branch in step_block)
    uint32_t elide_reward;         // constant-reward envs whose kernel does not elide the reward store unconditionally (CartPole): != 0 = this engine elides it (step_block)
    uint32_t* err;      // [0] number of invalid actions seen, [1] lowest offending lane (0xffffffff = none)
    uint32_t* err_seen; // mapped host words: [0] a wave that saw an invalid action sets it to 1: gymrs_sync looks there first and
                        // fetches err[] only then (no device-to-host copy per synchronisation); [1] a wave of a CHAIN launch that found
                        // itself on another XCD than the dispatcher's self-check saw for its workgroup index stores that index + 1
    uint64_t n;         // lanes in this engine
    uint64_t n_fast;    // n, or 0 when the action buffer is not aligned for the vector load (step_kernel)
    uint64_t gid0;      // global id of lane 0
    uint64_t seed;
    uint64_t tick;      // engine tick of this launch; when tick_base != NULL it is an OFFSET added to *tick_base
    const unsigned long long* tick_base; // device-resident tick for captured HIP graphs (NULL in eager launches)
    SampleBox box;      // reset sampling box, prepared on the host (gymrs_philox.h)
    uint32_t truncate_all; // envs that never terminate (Pendulum): this step hits the time limit for every lane
    uint32_t skip_trunc_store; // same envs: the `truncated` array already holds this step's (uniform) value
    unsigned long long* trace; // developer instrumentation (GYMRS_TRACE_TIMES builds), else NULL
    // Launches of a chain (gymrs_aql.h) carry no release fence between them, which is only right while tile i is stepped on the SAME
    // XCD (= the same L2) in every launch of the chain.  Where a queue's workgroups start is NOT a constant of the queue (round 4: it
    // differs between queues and changes while a queue sits idle), so the premise is checked per chain: the FIRST step launch after
    // aql_begin (xcc_check == 2) has its workgroups 0 .. 7 store {chain number, XCC id} into xcc_table (entry k at word k * kXccTableStride:
    // one cache line each; a plain store, which stays in that XCD's L2), every later launch of the chain (xcc_check == 1) compares the
    // XCC it runs on (HW_REG_XCC_ID) and the chain number with entry (blockIdx.x & 7) and reports a mismatch through err_seen[1]: a
    // workgroup on the wrong XCD does not see the entry at all (it reads an older chain's).  Between two chains everything is written back and
    // re-acquired, so a deal that changed THERE is harmless.  HIP launches (a release fence each) pass xcc_check == 0.
    uint32_t* xcc_table; // [8 * kXccTableStride] device words
    uint32_t xcc_check;
    uint32_t xcc_seq;    // the chain's number (24 bits)
    uint32_t trace_wpb; // wavefronts per workgroup of this launch (the stamps' index; blockDim would be a hidden kernel argument, which
                        // the engine's own dispatcher does not supply)
};

// The kernel-argument segment of step_kernel as the host sees it (the engine's own AQL dispatcher, gymrs_aql.h, fills it by
// hand; HIP launches marshal the same thing themselves): the parameters in order, naturally aligned.
template <class Consts>
struct StepKernArgs {
    float* s0;
    float* s1;
    float* s2;
    float* s3;
    const void* action;
    uint64_t n_fast;
    StepArgs rest;
    Consts c;
};

// The fused multi-step kernel (gymrs_rollout): n_steps consecutive steps of every lane in ONE launch, with
// the random-policy actions of gymrs_fill_actions(action_seed, action_t0 + k) generated in registers.
struct RolloutArgs {
    uint64_t action_seed, action_t0;
    uint64_t uniform_start; // Pendulum time limit: the tick at which every lane's current episode started
    uint32_t n_steps;
    uint32_t n_actions;     // Discrete(n) envs
    float max_torque;       // Pendulum
    // gymrs_rollout_record: every step's outputs go to trajectory buffers, row k = step k (NULL = not recorded)
    float* rec_obs;         // [n_steps][obs_dim][rec_stride]
    void* rec_action;       // [n_steps][rec_stride] u8 (f32 for Pendulum)
    float* rec_reward;      // [n_steps][rec_stride]
    uint8_t* rec_done;      // [n_steps][rec_stride]
    uint8_t* rec_trunc;     // [n_steps][rec_stride], written with GYMRS_TIME_LIMIT only; may be NULL
    uint64_t rec_stride;    // lanes per row: >= n, a multiple of 16
};

struct ResetArgs {
    float* s[4];
    float* obs_cos;
    float* obs_sin;
    float* reward;
    uint8_t* done;
    uint8_t* truncated;
    uint8_t* beyond;
    uint32_t* ep_start;
    uint64_t n, gid0, seed, tick;
    SampleBox box;
    // gymrs_reset_pcg64 (gymrs_pcg64.h): lane i draws its state from Pcg64::seed_from_u64(pcg_seeds ? pcg_seeds[i] :
    // seed + gid0 + i) over [pcg_low, pcg_low + pcg_scale) instead of from the Philox reset stream
    int pcg64;
    const uint64_t* pcg_seeds;
    double pcg_low[4], pcg_scale[4];
};

// Work-items per workgroup of a per-step launch: CartPole uses 512 once the launch still puts two such workgroups on every
// CU (measured 2 % faster there; small batches want many small workgroups), everything else kBlock.  launch_one
// (gymrs_step_impl.h) and the engine's AQL dispatcher both go by this.
// ... and only while the launch is at most two generations of waves: from 2^22 lanes on a CU refills better in units of 4 waves (round 4,
// profiles/r04_wave_variants.log: 2^22 lanes 25.7 -> 24.8 us, 2^23 48.7 -> 47.4; 2^21 12.0 vs 12.2 the other way).
constexpr int kCartPoleThreads = 512;
#ifndef GYMRS_EXP_BIG_BELOW // (developer builds: where the 512-work-item window of CartPole ends; profiles/r05_cartpole_2p21.log)
#define GYMRS_EXP_BIG_BELOW 2048
#endif
constexpr uint64_t kBigGroupsFrom = 512, kBigGroupsBelow = GYMRS_EXP_BIG_BELOW;
inline bool step_uses_big_groups(uint64_t n, int threads, int vec)
{
    return n >= (uint64_t)threads * vec * kBigGroupsFrom && n < (uint64_t)threads * vec * kBigGroupsBelow;
}
inline int step_threads_of(gymrs_env_kind kind, uint64_t n, int vec)
{
    return (kind == GYMRS_CARTPOLE && step_uses_big_groups(n, kCartPoleThreads, vec)) ? kCartPoleThreads : kBlock;
}

constexpr uint32_t kXccTableStride = 32; // words between two entries of StepArgs::xcc_table: one 128-byte line per entry

// Tiles (of `threads` * `vec` lanes) a workgroup of the per-step kernel steps one after the other: 1; developer builds (GYMRS_EXP_TILES,
// tools/devbuild.py) may ask for more (profiles/r04_two_tiles_per_workgroup.log).
#ifdef GYMRS_EXP_TILES
constexpr int kStepTiles = GYMRS_EXP_TILES;
#else
constexpr int kStepTiles = 1;
#endif

// Number of workgroups of `threads` work-items for n lanes at `vec` lanes per work-item (4 or 8).
inline uint32_t step_grid(uint64_t n, int vec, int threads = kBlock)
{
    const uint64_t per_block = (uint64_t)threads * vec;
    return (uint32_t)((n + per_block - 1) / per_block);
}

hipError_t launch_step(gymrs_env_kind kind, int vec, uint32_t flags, const StepArgs& a, const void* consts,
                       hipStream_t stream);
hipError_t launch_rollout(gymrs_env_kind kind, int vec, uint32_t flags, const StepArgs& a, const RolloutArgs& r,
                          const void* consts, hipStream_t stream);
hipError_t launch_reset(gymrs_env_kind kind, const ResetArgs& a, hipStream_t stream);
// Fold the `pending` (< kResetLogRows) not yet folded rows of the reset log (the steps first_tick .. first_tick + pending - 1,
// lanes-per-work-item `vec`) into ep_start (start tick of every lane's open episode) and the per-wavefront episode
// counters, and zero them again: on demand, off the hot path (the step kernel folds a full ring itself).
hipError_t launch_fold_reset_log(unsigned long long* log, uint32_t row_words, uint64_t first_tick, uint32_t pending, int vec,
                                 uint32_t* ep_start, uint64_t n, unsigned long long* block_stats, hipStream_t stream);
// wave_open[0] += sum of the others, others = 0 (before a launch whose lanes-per-wave differs from the last one's)
hipError_t launch_fold_open(double* wave_open, uint32_t n_slots, hipStream_t stream);
hipError_t launch_tick_advance(unsigned long long* tick_dev, unsigned long long by, hipStream_t stream);
hipError_t launch_fill_actions(gymrs_env_kind kind, void* actions, uint64_t n, uint64_t gid0, uint64_t seed, uint64_t t,
                               float max_torque, hipStream_t stream);
// A kernel launch (hipLaunchKernelGGL) reports its failure only through the calling thread's last-error word -- and since ROCm 7 that word keeps the last
// ERROR any earlier runtime call of the thread returned until somebody reads it (a successful call no longer overwrites it): a probe of this library
// that was allowed to fail, or an error the APPLICATION has not looked at.  Round 6's GPU suite met exactly that (an "invalid device ordinal" of a refused
// sharder creation surfaced three test files later as a failed reset).  Every launch site therefore reads the word once BEFORE it launches: what it
// returns afterwards is its own launch's.
inline void launch_begin() { (void)hipGetLastError(); }
constexpr int kStatsPartials = 256; // workgroups of the statistics read-out (= work-items of its finalize step)
struct StatsArgs {
    const uint32_t* ep_start;
    uint64_t n;
    uint32_t epoch;    // value reset() wrote into ep_start
    const unsigned long long* block_stats;
    uint32_t n_blocks;
    unsigned long long* partials; // [kStatsPartials][3] scratch: one {L, E, R} triple per workgroup of the read-out
    unsigned long long* base; // [kStatsBaseWords] {L, E, R (f64 bits)} at the last stats_clear: the BASELINE a read-out subtracts (see launch_stats)
    // the reset log's rows that are not folded yet: the read-out looks THROUGH them (read-only) instead of folding them first
    const unsigned long long* log;
    uint32_t log_row_words, log_pending;
    uint64_t log_first_tick;
    int log_vec;
    int track;                // GYMRS_TRACK_STATS set
    int reward_sign;          // +1 CartPole, -1 MountainCar (return = +-length), 0 Pendulum (summed)
    double n_steps;
    double* out4;             // {sum_return, sum_length, n_episodes, n_steps}
    double* host_out4;        // the same four into device-visible host memory (or NULL): a read-out without the copy engine
};
// mode 0 read, 1 clear (base = the totals now), 2 after reset (base = 0).
// The read-out WRITES NOTHING a step kernel reads (round 6; VERDICT r5 "next" #1c): gymrs_stats_clear used to zero the per-wavefront counters with a
// memset on the stream and to fold the reset log with an atomic-add kernel -- stream-side writes into memory the next launches (of the stream or of a chain on
// the engine's own queue) read-modify-write, the one place where round 5's stale episode counter could come from.  Now the counters only ever grow, written by
// the one wavefront that owns each slot; a clear remembers the totals and a read subtracts them (u64: exact; Pendulum's f64 return sum: to an ulp of the
// total), and rows of the reset log that are still pending are counted where they lie.
constexpr int kStatsBaseWords = 4;
hipError_t launch_stats(const StatsArgs& a, int mode, hipStream_t stream);
// The age of the oldest open episode, max over lanes of (uint32_t)(tick_ref - ep_start[lane]), from which the host derives
// the tick before which no lane's episode started (GYMRS_TIME_LIMIT elision, gymrs_engine.hip).  partials = kStatsPartials
// device words of scratch; the result goes to host_out2 (device-visible host memory): [0] = the age, then [1] = seq.
hipError_t launch_max_age(const uint32_t* ep_start, uint64_t n, uint32_t tick_ref, uint32_t* partials, uint32_t* host_out2, uint32_t seq,
                          hipStream_t stream);
} // namespace gymrs
