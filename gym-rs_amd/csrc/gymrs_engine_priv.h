// gymrs_engine_priv.h -- what the host-side translation units of the C ABI share: the engine object, the error convention and the few
// helpers more than one of them needs.  gymrs_engine.hip: creation, reset, the stepping paths (HIP launches, graphs, chains), gymrs_sync;
// gymrs_engine_io.hip: views and copies of the arrays, clone / snapshot, statistics, RCCL, the pub physics fields, the serde JSON view;
// gymrs_probe.hip: the copy probe (measurement).  Not installed, not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <random>
#include <string>
#include <vector>

#include "gymrs_amd.h"
#include "gymrs_aql.h"
#include "gymrs_json.h"
#include "gymrs_kernels.h"

using namespace gymrs;

#define GYMRS_HOST_INTERNAL __attribute__((visibility("hidden")))

// Every entry point returns a status instead of panicking; the message is kept per thread for gymrs_last_error().
GYMRS_HOST_INTERNAL gymrs_status fail(gymrs_status st, const std::string& msg);

#define HIP_TRY(expr)                                                                                              \
    do {                                                                                                           \
        hipError_t err_ = (expr);                                                                                  \
        if (err_ != hipSuccess)                                                                                    \
            return fail(GYMRS_EHIP, std::string(#expr) + ": " + hipGetErrorString(err_));                          \
    } while (0)

// RCCL entry points, resolved at first use so that the stepping path has no hard RCCL dependency.
struct NcclId128 { // ncclUniqueId (rccl.h): 128 opaque bytes, passed BY VALUE to ncclCommInitRank
    char internal[128];
};
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, NcclId128, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GroupStart)() = nullptr; // several devices driven by ONE thread (gymrs_allreduce_stats_multi): RCCL's calls are collective per
    int (*GroupEnd)() = nullptr;   // communicator, so one thread may only issue them for all of its devices inside a group
};

struct gymrs_engine {
    gymrs_env_kind kind;
    uint64_t n = 0, gid0 = 0;
    int device = 0;
    uint32_t flags = 0;
    int state_dim = 0, obs_dim = 0;
    union {
        CartPoleConsts cp;
        MountainCarConsts mc;
        PendulumConsts pd;
    } consts;
    union { // the pub physics fields as the caller set them (f64 like the reference's O64); consts is derived from them
        gymrs_cartpole_params cp;
        gymrs_mountain_car_params mc;
        gymrs_pendulum_params pd;
    } params;
    float max_torque = 2.0f;
    float lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};         // current reset box
    float dflt_lo[4] = {0, 0, 0, 0}, dflt_hi[4] = {0, 0, 0, 0}; // default reset box
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // device buffers
    float* s[4] = {nullptr, nullptr, nullptr, nullptr};
    float* obs_cos = nullptr;
    float* obs_sin = nullptr;
    float* reward = nullptr;
    uint8_t* done = nullptr;
    uint8_t* truncated = nullptr;
    uint8_t* beyond = nullptr;
    uint32_t* ep_start = nullptr;
    uint32_t* wave_clean = nullptr; // per-wavefront: its part of `reward` holds the env's constant reward (see step_block)
    int clean_shape = 0;           // lanes per workgroup row the flags were written with (vec * threads); 0 = all clear
    bool elide_reward = false;     // CartPole with GYMRS_AUTO_RESET from kElideRewardFromBytes per step on: the constant reward store is elided like MountainCar's
    double* wave_open = nullptr; // per-wavefront sum of the rewards of the open episodes (Pendulum + TRACK_STATS)
    int open_vec = 0;            // lanes per work-item of the launch that last updated wave_open (0 = none yet)
    int trunc_held = -1;         // Pendulum: the uniform value the `truncated` array holds (-1 = unknown: write it)
    unsigned long long* block_stats = nullptr;
    uint32_t n_stat_blocks = 0;
    // reset log (see StepArgs::reset_log): ring of done-mask rows the per-step kernel writes instead of scattered ep_start
    // stores; `log_pending` consecutive steps starting at tick `log_first_tick` have rows that are not folded yet
    unsigned long long* reset_log = nullptr;
    uint32_t log_row_words = 0, log_pending = 0;
    uint64_t log_first_tick = 0;
    int log_vec = 4; // lanes per work-item of the launches that wrote the pending rows
    void* pool = nullptr; // one allocation holding every per-lane array (see engine_create)
    size_t pool_bytes = 0;
    // Engines of up to kHostPoolMaxLanes lanes (the single-env mirrors: one lane) keep that allocation in mapped host
    // memory: the kernels reach it over the bus, and the host reads a step's results (state, reward, flags) and writes
    // actions and states with plain loads and stores after a stream synchronisation instead of three to five copy
    // commands (single-env step through the Python mirror: 156 -> ~40 us).
    char* pool_host = nullptr;       // host address of the pool (NULL = the pool is device memory)
    char* staging_host = nullptr;    // host address of action_staging when that is mapped host memory too
    int vec = 4; // lanes per work-item
    int nt_mode = 0; // 0 = automatic, 1 = always non-temporal, 2 = never
    unsigned long long* trace = nullptr; // developer instrumentation buffer (GYMRS_TRACE_TIMES builds)
    uint32_t* err = nullptr;
    volatile uint32_t* err_seen = nullptr; // mapped host words (StepArgs::err_seen): [0] a kernel saw an invalid action, [1] a chain launch ran on another XCD than expected
    uint32_t* err_seen_dev = nullptr;
    double* stats_dev = nullptr;
    volatile double* stats_host = nullptr; // mapped host memory the read-out kernel writes the same four doubles into
    double* stats_host_dev = nullptr;      // the device's address of it
    unsigned long long* stats_acc = nullptr;  // [kStatsPartials][3] scratch of the statistics read-out
    unsigned long long* stats_base = nullptr; // [kStatsBaseWords] the statistics BASELINE {sum of start ticks, episodes, returns} of the last gymrs_stats_clear
    uint32_t epoch = 1;                       // ep_start value written by the last reset()
    void* action_staging = nullptr; // for gymrs_step_host
    uint64_t seed = 0, tick = 0;
    uint64_t uniform_start = 0; // Pendulum: tick at which every lane's current episode started
    uint32_t max_steps = 0;
    double n_steps_total = 0;
    // captured HIP graph of `graph_steps` consecutive steps (gymrs_step_many use_graph)
    hipGraphExec_t graph_exec = nullptr;
    const void* graph_actions = nullptr;
    uint64_t graph_stride = 0;
    uint32_t graph_nbuf = 0, graph_steps = 0, graph_flags = 0;
    int graph_vec = 0;
    uint64_t graph_seed = 0;
    unsigned long long* tick_dev = nullptr;
    void* comm = nullptr; // ncclComm_t
    int n_ranks = 1, comm_rank = 0;
    // GYMRS_TIME_LIMIT elision (CartPole with all three flags): a launch whose tick cannot take any lane to the limit runs
    // the kernel WITHOUT the limit -- the reset-logged headline kernel -- and the `truncated` array stays all zero.  start_bound = a tick no open episode started before (ep_start only grows, so a
    // stale bound stays valid); refreshed asynchronously from the age of the oldest open episode (max_age_kernel).
    bool limit_elidable = false;
    uint64_t start_bound = 0;
    bool trunc_zero = false;            // the `truncated` array is known to hold zeros only
    uint32_t* age_dev = nullptr;        // device scratch of a refresh: one maximum per workgroup
    volatile uint32_t* age_host = nullptr; // mapped host memory the device writes {age, sequence number} into
    uint32_t* age_host_dev = nullptr;   // the device's address of it
    uint32_t age_seq = 0;               // sequence number of the refresh in flight
    bool age_pending = false;
    bool age_gave_up = false;           // a bounded wait for the refresh in flight ran out: do not wait for THIS refresh again
    bool age_near_done = false;         // the one refresh of this approach to the limit has been issued
    bool last_elided = false;           // the previous per-step launch ran without the limit
    uint64_t age_ref_tick = 0;          // tick the ages of the refresh in flight are measured from
    uint64_t age_next_refresh = 0;      // no new refresh before this tick (doubling back-off while the limit stays reachable)
    uint32_t age_backoff = 8;
    // The engine's own AQL dispatcher for chains of per-step launches (gymrs_aql.h): set up by gymrs_engine_create (or, for an engine created
    // under GYMRS_AQL=0, by the first gymrs_step_many that can use it); aql_why says why not when it stays NULL.
    AqlChain* aql = nullptr;
    bool aql_tried = false;
    uint32_t dev_hooks = 0;  // gymrs_dev_set_hooks (test / developer hooks of the chain path; 0 in production)
    bool chain_open = false; // gymrs_step_many is inside aql_begin .. aql_end: see stream_op_barrier
    bool chain_first = false; // the next step launch is the first of the open chain: it records the XCD table (StepArgs::xcc_table)
    std::string aql_why, aql_handover;
    uint64_t aql_chains = 0, aql_launches = 0; // for the serde view's engine extras (tests, diagnostics)
    uint32_t last_flags = 0;                   // launch flags (engine flags | hint bits) of the most recent per-step launch ...
    int last_path = 0;                         // ... and how it was submitted: 0 none yet, 1 HIP launch, 2 chain (the extras' "last_launch")
    uint64_t limit_elided_launches = 0; // for the serde view's engine extras (tests, diagnostics)
    uint64_t age_refreshes = 0, age_waits = 0, age_wait_ns = 0;
};

// What the reference's TYPES rule out (everything else is a `pub` f64 field the reference accepts as it is, NaN included):
// KinematicsIntegrator is a two-variant enum (cartpole.rs:380-387).
GYMRS_HOST_INTERNAL inline gymrs_status check_params(gymrs_env_kind kind, const void* params, const char* who)
{
    if (kind == GYMRS_CARTPOLE && params) {
        const int k = static_cast<const gymrs_cartpole_params*>(params)->kinematics_integrator;
        if (k != 0 && k != 1) return fail(GYMRS_EINVAL, std::string(who) + ": kinematics_integrator must be 0 (Euler) or 1 (Other)");
    }
    return GYMRS_OK;
}

GYMRS_HOST_INTERNAL inline const void* consts_ptr(const gymrs_engine* e)
{
    switch (e->kind) {
    case GYMRS_CARTPOLE: return &e->consts.cp;
    case GYMRS_MOUNTAIN_CAR: return &e->consts.mc;
    default: return &e->consts.pd;
    }
}

// Host address of one of the pool's arrays (engines whose pool is mapped host memory).
template <class T>
inline T* host_of(const gymrs_engine* e, T* dev)
{
    return reinterpret_cast<T*>(e->pool_host + (reinterpret_cast<char*>(dev) - static_cast<char*>(e->pool)));
}

// defined in gymrs_engine.hip
GYMRS_HOST_INTERNAL StatsArgs stats_args(const gymrs_engine* e);
GYMRS_HOST_INTERNAL gymrs_status fold_reset_log(gymrs_engine* e);
GYMRS_HOST_INTERNAL gymrs_status stream_sync_checked(gymrs_engine* e); // hipStreamSynchronize + what a chain reported meanwhile
GYMRS_HOST_INTERNAL void limit_restart(gymrs_engine* e, uint64_t bound, bool trunc_zero);
GYMRS_HOST_INTERNAL std::string aql_kernel_name(const gymrs_engine* e, uint32_t flags, int threads);
// defined in gymrs_engine_io.hip (RCCL is resolved there, at first use)
GYMRS_HOST_INTERNAL void comm_destroy(gymrs_engine* e);
