/*
 * gymrs_amd.h — C ABI of the MI355X-native batched classic-control stepper.
 *
 * This is the drop-in boundary for the ONE hot path of MathisWellmann/gym-rs:
 * Env::step() (+ the reset() that re-arms a finished lane) of CartPole / MountainCar
 * (/ a spec-derived Pendulum).  The reference has no FFI; its boundary is the trait pair in
 * src/core.rs.  Each entry point below cites the reference interface it replaces, with
 * paths relative to /root/reference.  The binding a gym-rs maintainer would add (a Rust
 * `extern "C"` block + `impl Env`) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - Plain pointers and sizes only.  Device pointers are HIP device addresses.
 *   - Every function returns a gymrs_status instead of panicking (the reference asserts:
 *     cartpole.rs:402-406); gymrs_last_error() gives the message for the calling thread.
 *   - The engine owns its device buffers.  The caller owns every host buffer it passes.
 *   - gymrs_step* are asynchronous on the engine's HIP stream; gymrs_sync() waits.
 *   - One engine is driven by one host thread at a time (the reference's methods take
 *     `&mut self`, core.rs:42-50).  Engines on different GPUs may be driven concurrently.
 *   - N independent envs ("lanes") live as SoA f32 arrays in HBM; lane i of an engine has the
 *     global id global_env_offset + i, which (with the seed and the engine tick) fully
 *     determines its reset draws, so results do not depend on how lanes are sharded over GPUs.
 */
#ifndef GYMRS_AMD_H
#define GYMRS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 5): + gymrs_sharded_*, gymrs_allreduce_stats_multi; - gymrs_copy_probe (a measurement tool now: tools/copy_probe) */
#define GYMRS_ABI_VERSION 3

typedef struct gymrs_engine gymrs_engine; /* opaque; owns device buffers + stream */

typedef enum {
    GYMRS_OK = 0,
    GYMRS_EINVAL = 1,  /* bad argument */
    GYMRS_EHIP = 2,    /* HIP runtime error */
    GYMRS_ENCCL = 3,   /* RCCL error */
    GYMRS_ENOMEM = 4,  /* allocation failed */
    GYMRS_EACTION = 5, /* an action outside the action space was seen (reference: assert! panic,
                          cartpole.rs:402-406, mountain_car.rs:402-406).  The offending lanes' STATE is
                          left untouched (the reference panics before touching the env); their reward /
                          done / truncated entries of that step read 0.  The engine tick still advances
                          for every lane, so with GYMRS_TIME_LIMIT / GYMRS_TRACK_STATS the episode clock
                          and the statistics of a lane whose action was rejected count the rejected
                          step: treat statistics as undefined once GYMRS_EACTION has been reported
                          (the reference would have aborted the process). */
} gymrs_status;

/* classical_control/mod.rs:1-4 exports cartpole and mountain_car.  Pendulum is NOT in the
 * reference (spec-derived from Gym's Pendulum-v1). */
typedef enum { GYMRS_CARTPOLE = 0, GYMRS_MOUNTAIN_CAR = 1, GYMRS_PENDULUM = 2 } gymrs_env_kind;

/* Engine flags.  The reference has none of these behaviours: with flags = 0 a
 * lane behaves exactly like one reference env (no auto-reset, no truncation). */
enum {
    GYMRS_AUTO_RESET = 1u,  /* a lane whose step returned done (or truncated) is re-armed inside
                               the same kernel: the caller loop of examples/cartpole.rs:23-28 */
    GYMRS_TRACK_STATS = 2u, /* accumulate {sum_return, sum_length, n_episodes} of finished episodes
                               (needs GYMRS_AUTO_RESET) */
    GYMRS_TIME_LIMIT = 4u,  /* truncated = (steps in episode >= max_episode_steps); the cap the
                               reference leaves to its callers (examples/cartpole.rs:18) */
};

/* Physics constants: the `pub` fields of the reference env structs, in f64 like the reference
 * (O64).  They are converted to f32 once at engine creation.  Pass NULL for the defaults. */
typedef struct {
    double gravity;                 /* 9.8    cartpole.rs:94  */
    double masscart;                /* 1.0    cartpole.rs:95  */
    double masspole;                /* 0.1    cartpole.rs:96  */
    double length;                  /* 0.5    cartpole.rs:97  */
    double force_mag;               /* 10.0   cartpole.rs:98  */
    double tau;                     /* 0.02   cartpole.rs:99  */
    double theta_threshold_radians; /* 12*2*pi/360  cartpole.rs:102 */
    double x_threshold;             /* 2.4    cartpole.rs:103 */
    int32_t kinematics_integrator;  /* 0 Euler (cartpole.rs:100), 1 Other (cartpole.rs:436-441) */
    uint32_t max_episode_steps;     /* used with GYMRS_TIME_LIMIT; 0 -> 500 (doc cartpole.rs:50) */
} gymrs_cartpole_params;

typedef struct {
    double min_position;  /* -1.2   mountain_car.rs:344 */
    double max_position;  /* 0.6    mountain_car.rs:345 */
    double max_speed;     /* 0.07   mountain_car.rs:346 */
    double goal_position; /* 0.5    mountain_car.rs:347 */
    double goal_velocity; /* 0.0    mountain_car.rs:348 */
    double force;         /* 0.001  mountain_car.rs:350 */
    double gravity;       /* 0.0025 mountain_car.rs:351 */
    uint32_t max_episode_steps; /* 0 -> 200 (doc mountain_car.rs:45) */
    uint32_t _pad;
} gymrs_mountain_car_params;

typedef struct { /* spec-derived (Gym Pendulum-v1), not in the reference */
    double max_speed;  /* 8    */
    double max_torque; /* 2    */
    double dt;         /* 0.05 */
    double g;          /* 10   */
    double m;          /* 1    */
    double l;          /* 1    */
    uint32_t max_episode_steps; /* 0 -> 200 */
    uint32_t _pad;
} gymrs_pendulum_params;

/* Fill *params (a gymrs_*_params of the right kind) with the defaults of CartPoleEnv::new /
 * MountainCarEnv::new (cartpole.rs:91-144, mountain_car.rs:341-390). */
gymrs_status gymrs_default_params(gymrs_env_kind kind, void* params);

/* ---- spaces: EnvProperties::action_space / observation_space (core.rs:86-89) -------------- */
/* Discrete(n) for CartPole (2, cartpole.rs:114) and MountainCar (3, mountain_car.rs:362);
 * *n = 0 for Pendulum, whose action space is the box [-max_torque, max_torque]. */
gymrs_status gymrs_action_space(gymrs_env_kind kind, uint32_t* n, double* box_low, double* box_high);
/* BoxR{low, high}: cartpole.rs:105-115, mountain_car.rs:353-364.  low/high hold *dim doubles
 * (capacity >= 4). */
gymrs_status gymrs_observation_space(gymrs_env_kind kind, const void* params, double* low, double* high,
                                     int* dim);
/* Discrete::contains (spaces/discrete.rs:14-19): value < n. */
int gymrs_discrete_contains(uint64_t n, uint64_t value);

/* ---- lifetime: CartPoleEnv::new / MountainCarEnv::new -------------------------------------- */
/* Like ::new (cartpole.rs:92,120), creation seeds from OS entropy and samples an initial state,
 * so the engine is steppable before the first gymrs_reset(). */
gymrs_status gymrs_engine_create(gymrs_env_kind kind, uint64_t n_envs, uint64_t global_env_offset, int device,
                                 const void* params /* NULL = defaults */, uint32_t flags, gymrs_engine** out);
/* Env::close (core.rs:56) + drop. */
gymrs_status gymrs_engine_destroy(gymrs_engine* e);

/* Use an externally created hipStream_t (e.g. the host framework's) instead of the engine's own. */
gymrs_status gymrs_set_stream(gymrs_engine* e, void* hip_stream);
gymrs_status gymrs_get_stream(gymrs_engine* e, void** hip_stream);

/* ---- Env::reset(seed, return_info, options) (core.rs:45-50) -------------------------------- */
/* has_seed = 0 re-seeds from OS entropy (seeding.rs:22); the same seed always gives the same
 * states.  bounds_low_high = obs_dim lows then obs_dim highs replacing the default
 * sampling box (`options`, cartpole.rs:352-364, mountain_car.rs:175-190), or NULL.
 * *seed_used (may be NULL) receives the seed number, like rand_random's second return value
 * (seeding.rs:21-26). */
gymrs_status gymrs_reset(gymrs_engine* e, int has_seed, uint64_t seed, const float* bounds_low_high,
                         uint64_t* seed_used);
/* Optional: the same reset, but every lane draws its state from the REFERENCE's own generator chain instead of the
 * Philox reset stream: `Pcg64::seed_from_u64(s)` (seeding.rs:21-26) and `Uniform::new(low, high).sample(rng)` over
 * f64 (cartpole.rs:293-297,317-324,352-364; mountain_car.rs:145,162-167,175-190), so that lane i holds, rounded once
 * to f32, the state the reference's `reset(Some(s), _, options)` returns for
 *     s = seeds_dev[i]                       (n_envs seed numbers on the device), or, with seeds_dev = NULL,
 *     s = seed + global_env_offset + i       (wrapping; a one-lane engine at offset 0 is the reference env itself).
 * bounds_low_high = obs_dim lows then obs_dim highs in f64 (the reference's BoxR is f64), or NULL for the defaults.
 * CartPole and MountainCar only (the reference has no Pendulum): GYMRS_EINVAL otherwise.  Everything else -- cleared
 * flags, statistics, the seed echo, the Philox key of later GYMRS_AUTO_RESET re-arms -- is as for gymrs_reset.
 * The third-party algorithms (rand 0.8, rand_pcg 0.3, rand_core 0.6) are restated from their publications and pinned
 * by rand_pcg's own known answers (tests/golden/pcg64.json); SURVEY App. B.2.
 * Rounding: the reference's f64 draw lies in [low, high); this call stores its NEAREST f32, so that a caller comparing start
 * states with gym-rs sees every component within half an f32 ulp of the reference's.  The stored value therefore lies in the
 * CLOSED f32 box [fl32(low), fl32(high)]: a draw within half an ulp of `high` rounds onto fl32(high) (probability ~2^-25 per
 * draw), which may exceed `high` itself (fl32(0.05) > 0.05).  gymrs_reset (Philox) keeps its samples strictly below
 * fl32(high); later GYMRS_AUTO_RESET re-arms of an engine reset through this call use that half-open f32 box. */
gymrs_status gymrs_reset_pcg64(gymrs_engine* e, int has_seed, uint64_t seed, const uint64_t* seeds_dev,
                               const double* bounds_low_high, uint64_t* seed_used);

/* ---- Env::step(action) (core.rs:42) --------------------------------------------------------- */
/* actions_dev: n_envs actions on the device: uint8_t for CartPole {0,1} / MountainCar {0,1,2},
 * float for Pendulum.  Asynchronous.  Results land in the arrays below.
 * (One exception to "asynchronous": a CartPole engine with all three flags runs the launches that cannot take any lane to
 * the time limit without the limit check, and learns the age of the oldest open episode from the device a few launches
 * before that knowledge runs out.  A caller that has queued steps far ahead of the GPU may wait in this call until the
 * GPU has reached that refresh -- at most once per approach to the limit; the queue does not run dry.)
 * Alignment: a buffer aligned to lanes_per_thread * sizeof(action) bytes (4 or 8 B for u8 actions, 16 or 32 B
 * for f32; any hipMalloc / torch allocation is) is read with one vector load per work-item.  Any other address
 * is accepted too and read lane by lane (every wavefront then takes the guarded per-lane code: correct, slower).
 * The same holds for every ring slot actions_dev + k * stride_bytes of gymrs_step_many. */
gymrs_status gymrs_step(gymrs_engine* e, const void* actions_dev);
/* Same with a host action buffer (copied first); for the single-env compatibility layer. */
gymrs_status gymrs_step_host(gymrs_engine* e, const void* actions_host);
/* n_steps consecutive step() calls; step t reads its actions at
 * actions_dev + (t % n_buffers) * stride_bytes.  use_graph != 0 replays a captured HIP graph of >= 32 steps
 * (a whole number of passes over the action ring; the remainder is launched eagerly): worth it when the step
 * kernel is shorter than a host launch (~3 us, i.e. small batches); results are identical.  Not available
 * for Pendulum with GYMRS_TIME_LIMIT (GYMRS_EINVAL).
 * Submission (round 6): HIP launches on the engine's stream -- every step's arrays are released when its launch ends, exactly what a loop of
 * gymrs_step enqueues.  GYMRS_AQL=1 in the environment OPTS IN to chains on the engine's own HSA queue (release only at the end of the call,
 * 25 % less per launch; one process per GPU: INTEGRATION.md "gymrs_step_many"); results are the same bits either way. */
gymrs_status gymrs_step_many(gymrs_engine* e, const void* actions_dev, uint64_t stride_bytes,
                             uint32_t n_buffers, uint32_t n_steps, int use_graph);
/* Fused random-policy rollout: exactly the effect of
 *     for (k = 0; k < n_steps; ++k) { gymrs_fill_actions(e, buf, action_seed, action_t0 + k); gymrs_step(e, buf); }
 * in ONE kernel launch with the state held in registers (the caller loop of examples/cartpole.rs:15-30 --
 * `rng.gen_range`, `step`, `reset` on done, `episode reward +=` -- for every lane).  Afterwards the state, the
 * reward/done/truncated arrays (those of the LAST step), the statistics and the tick are bit-identical to that
 * loop; the intermediate observations are never materialised, which is why this is a separate entry point and
 * not what bench.py's headline measures.  VALU-bound instead of HBM-bound. */
gymrs_status gymrs_rollout(gymrs_engine* e, uint32_t n_steps, uint64_t action_seed, uint64_t action_t0);
/* gymrs_rollout that also keeps the trajectory: what a random-policy data-collection loop around Env::step stores
 * (examples/cartpole.rs:15-30 with the observation, action, reward and done of every step kept).  Row k of each
 * buffer is step k; a row holds lane_stride lanes (>= n_envs, a multiple of 16); all buffers are device memory,
 * 16-byte aligned:
 *   obs      [n_steps][obs_dim][lane_stride] f32   observation AFTER the step (a re-armed lane shows its fresh state,
 *                                                  exactly what gymrs_obs_ptrs would show after that step)
 *   actions  [n_steps][lane_stride] u8 (f32 for Pendulum)   the action taken
 *   reward   [n_steps][lane_stride] f32,  done [n_steps][lane_stride] u8
 *   truncated[n_steps][lane_stride] u8    written with GYMRS_TIME_LIMIT only; may be NULL
 * Everything else is as gymrs_rollout (engine arrays, statistics and tick end up identical). */
typedef struct {
    float* obs;
    void* actions;
    float* reward;
    uint8_t* done;
    uint8_t* truncated;
    uint64_t lane_stride;
} gymrs_trajectory;
gymrs_status gymrs_rollout_record(gymrs_engine* e, uint32_t n_steps, uint64_t action_seed, uint64_t action_t0,
                                  const gymrs_trajectory* out);
/* `Env: Clone + Serialize` (core.rs:25; the serde-visible fields of cartpole.rs:51-87 / mountain_car.rs:46-80).
 * gymrs_engine_clone: a second engine (own stream, same device) with a deep copy of everything a step can observe:
 * lane state, episode bookkeeping, statistics, physics constants, reset box and the RNG position (seed, tick).
 * Like the reference's Clone (which drops the GUI handle, screen.rs:66-77) it does not copy the HIP stream binding,
 * the RCCL communicator or a captured graph.
 * gymrs_snapshot_*: the same content as an opaque host blob of gymrs_snapshot_size() bytes.  Loading requires an
 * engine created with the same kind, n_envs and flags (GYMRS_EINVAL otherwise); physics constants, reset box,
 * global_env_offset, seed and tick come from the snapshot.  A restored or cloned engine continues bit-identically
 * to the original. */
gymrs_status gymrs_engine_clone(gymrs_engine* src, gymrs_engine** out);
gymrs_status gymrs_snapshot_size(gymrs_engine* e, uint64_t* bytes);
gymrs_status gymrs_snapshot_save(gymrs_engine* e, void* host_buf, uint64_t bytes);
gymrs_status gymrs_snapshot_load(gymrs_engine* e, const void* host_buf, uint64_t bytes);
/* Wait for the stream; returns GYMRS_EACTION if any step since the last sync saw an invalid action. */
gymrs_status gymrs_sync(gymrs_engine* e);

/* ---- ActionReward{observation, reward, done, truncated} (core.rs:94-106), batched ------------ */
/* Zero-copy SoA device views, valid until destroy.  CartPole obs = (x, x_dot, theta, theta_dot)
 * (cartpole.rs:336-349), MountainCar obs = (position, velocity) (mountain_car.rs:193-197); for
 * these the observation IS the state array.  Pendulum obs = (cos, sin, theta_dot).
 * The views are for READING: a step does not rewrite an output that already holds the right value (MountainCar's -1.0,
 * CartPole's 1.0 on engines of >= 2^22 lanes, Pendulum's done / uniform truncated flag), so whoever overwrites the reward or
 * flag arrays between two steps finds them repaired only by gymrs_reset or a snapshot load. */
gymrs_status gymrs_obs_ptrs(gymrs_engine* e, float** out_ptrs /* capacity 4 */, int* obs_dim);
gymrs_status gymrs_state_ptrs(gymrs_engine* e, float** out_ptrs /* capacity 4 */, int* state_dim);
gymrs_status gymrs_reward_ptr(gymrs_engine* e, float** out);
gymrs_status gymrs_done_ptr(gymrs_engine* e, uint8_t** out);
gymrs_status gymrs_truncated_ptr(gymrs_engine* e, uint8_t** out);

/* Host copies (synchronising).  SoA: dim arrays of `count` floats, back to back.
 * get/set_state is the engine's Clone/Serialize equivalent (core.rs:25). */
gymrs_status gymrs_get_obs(gymrs_engine* e, uint64_t first, uint64_t count, float* host_out);
gymrs_status gymrs_get_state(gymrs_engine* e, uint64_t first, uint64_t count, float* host_out);
gymrs_status gymrs_set_state(gymrs_engine* e, uint64_t first, uint64_t count, const float* host_in);
gymrs_status gymrs_get_step_result(gymrs_engine* e, uint64_t first, uint64_t count, float* reward,
                                   uint8_t* done, uint8_t* truncated /* each may be NULL */);

/* ---- episode statistics (the all-reduce payload) --------------------------------------------- */
/* out = {sum_return, sum_length, n_episodes, n_steps} for this engine's lanes since the last gymrs_stats_clear / gymrs_reset (synchronising).
 * The read-out writes nothing a step reads (round 6): gymrs_stats_clear remembers the totals as a baseline that later read-outs subtract (integers:
 * exact; Pendulum's sum_return: a difference of two f64 sums) -- asynchronous on the engine's stream, no counter is zeroed. */
gymrs_status gymrs_stats(gymrs_engine* e, double out[4]);
gymrs_status gymrs_stats_clear(gymrs_engine* e);
/* Reduce the per-workgroup partials into 4 doubles in device memory (async on the engine's stream)
 * and return that device address, e.g. to hand it to an RCCL all-reduce. */
gymrs_status gymrs_stats_device(gymrs_engine* e, double** dev_out4);

/* RCCL over xGMI: one process per GPU.  Rank 0 makes an id, the host framework distributes the
 * 128 bytes, every rank joins, then gymrs_allreduce_stats sums the 4 doubles over all ranks. */
gymrs_status gymrs_comm_unique_id(uint8_t id_out[128]);
gymrs_status gymrs_comm_init(gymrs_engine* e, int n_ranks, int rank, const uint8_t id[128]);
gymrs_status gymrs_allreduce_stats(gymrs_engine* e, double out[4]);

/* ---- one batch over several GPUs, in ONE process ------------------------------------------- */
/* The in-process form of gymrs_allreduce_stats (`gymrs_allreduce_stats(gymrs_engine** shards, int n, double out[4])`): one host
 * thread holds every shard of a batch.  Shards on n DISTINCT devices: one RCCL communicator over exactly these engines (made on the first
 * call, all ranks inside one ncclGroupStart/End -- a bare gymrs_comm_init per engine from one thread would wait for the other ranks for
 * ever), then n grouped all-reduces of 32 bytes over xGMI, each on its engine's stream.  Shards sharing a device (RCCL refuses two ranks
 * on one GPU) or n = 1: the same four doubles are summed on the host.  *used_rccl (may be NULL) says which: 1 RCCL, 0 the host-side sum, -1 the
 * host-side sum BECAUSE RCCL could not be loaded or its communicator could not be made (round 6: first contact with RCCL does not cost the caller
 * its result; gymrs_last_error() then holds RCCL's message although the status is GYMRS_OK).  Synchronising; the calling thread's current HIP device
 * is left on the last shard's. */
gymrs_status gymrs_allreduce_stats_multi(gymrs_engine** shards, int n, double out[4], int* used_rccl);

/* A batch of n_total lanes cut into n_shards contiguous blocks, one engine per block on devices[r] (NULL = devices 0 .. n_shards-1; a
 * device may appear more than once), lane i of the batch carrying the global id global_env_offset + i whatever the cut -- so every result
 * below is bit-identical to ONE engine of n_total lanes (one exception: Pendulum's sum_return is a float sum over wavefront slots, and k engines
 * add their slots in another order than one -- equal to ~1e-12 relative; every integer statistic, every state bit, every reward / flag is identical).
 * Blocks are whole 1024-lane tiles dealt evenly (the first `tiles % n_shards` blocks hold one more), the ragged tail in the last.  Each engine is driven by its own host thread, bound to its device, for its
 * whole life: the calls below hand one command to every thread and return when all have ENQUEUED (the step calls stay asynchronous on
 * each engine's stream).  One gymrs_sharded is driven by one caller thread at a time (`&mut self`, core.rs:42-50).
 * Creation seeds every block with ONE OS-entropy seed (cartpole.rs:92,120).  Error messages name the shard and its device. */
typedef struct gymrs_sharded gymrs_sharded;
gymrs_status gymrs_sharded_create(gymrs_env_kind kind, uint64_t n_total, uint64_t global_env_offset, int n_shards,
                                  const int* devices, const void* params, uint32_t flags, gymrs_sharded** out);
gymrs_status gymrs_sharded_destroy(gymrs_sharded* h);
gymrs_status gymrs_sharded_count(gymrs_sharded* h, int* n_shards);
/* Block r: its engine (for the zero-copy views gymrs_obs_ptrs / gymrs_reward_ptr / ... and the stream; do not step it directly while
 * the sharder is in use), its first lane within the batch, its lane count and its device.  Any out pointer may be NULL. */
gymrs_status gymrs_sharded_shard(gymrs_sharded* h, int shard, gymrs_engine** engine, uint64_t* first_lane,
                                 uint64_t* n_lanes, int* device);
/* Env::reset for the whole batch: gymrs_reset on every block with the same seed (has_seed = 0: one fresh OS seed). */
gymrs_status gymrs_sharded_reset(gymrs_sharded* h, int has_seed, uint64_t seed, const float* bounds_low_high,
                                 uint64_t* seed_used);
/* Env::step / n_steps of them: actions_dev[r] = block r's action buffer (ring) ON ITS DEVICE, laid out as for gymrs_step / gymrs_step_many. */
gymrs_status gymrs_sharded_step(gymrs_sharded* h, const void* const* actions_dev);
gymrs_status gymrs_sharded_step_many(gymrs_sharded* h, const void* const* actions_dev, uint64_t stride_bytes,
                                     uint32_t n_buffers, uint32_t n_steps, int use_graph);
gymrs_status gymrs_sharded_fill_actions(gymrs_sharded* h, void* const* actions_dev, uint64_t seed, uint64_t t);
/* gymrs_rollout / gymrs_set_params on every block (the rollout's action stream is keyed by global lane ids: same bits as one engine). */
gymrs_status gymrs_sharded_rollout(gymrs_sharded* h, uint32_t n_steps, uint64_t action_seed, uint64_t action_t0);
gymrs_status gymrs_sharded_set_params(gymrs_sharded* h, const void* params);
/* Waits for every block's stream; GYMRS_EACTION etc. as gymrs_sync (the first failing shard is reported). */
gymrs_status gymrs_sharded_sync(gymrs_sharded* h);
/* {sum_return, sum_length, n_episodes, n_steps} of the whole batch = gymrs_allreduce_stats_multi over the blocks. */
gymrs_status gymrs_sharded_stats(gymrs_sharded* h, double out[4]);
gymrs_status gymrs_sharded_stats_clear(gymrs_sharded* h);
/* "rccl" | "host" | "host (RCCL unavailable, ...: <RCCL's message>)" | "none": how the last gymrs_sharded_stats summed. */
const char* gymrs_sharded_reduce_path(gymrs_sharded* h);
/* Host copies over lanes [first, first + count) of the BATCH, laid out exactly as gymrs_get_state / gymrs_get_step_result lay out an
 * engine's lanes (SoA: state_dim arrays of `count` floats back to back).  Synchronising. */
gymrs_status gymrs_sharded_get_state(gymrs_sharded* h, uint64_t first, uint64_t count, float* host_out);
gymrs_status gymrs_sharded_get_step_result(gymrs_sharded* h, uint64_t first, uint64_t count, float* reward,
                                           uint8_t* done, uint8_t* truncated /* each may be NULL */);

/* ---- the `pub` physics fields after construction (cartpole.rs:53-82, mountain_car.rs:49-62) --- */
/* In the reference the constants are public struct fields: `env.gravity = ...` between two step() calls touches
 * nothing else -- state, steps_beyond_terminated, the episode clock and the PRNG carry on (cartpole.rs:455-464 reads
 * the fields afresh on every step).  gymrs_set_params is that assignment for every lane of the engine: only the
 * constants change (the next launch uses them; a captured HIP graph is dropped); no device array, no tick, no
 * statistics are touched.  params = the env kind's params struct (not NULL).  gymrs_get_params reads them back
 * in f64 exactly as they were set (the f32 conversion happens per launch-constant block, not in this copy). */
gymrs_status gymrs_set_params(gymrs_engine* e, const void* params);
gymrs_status gymrs_get_params(gymrs_engine* e, void* params_out);

/* ---- `#[derive(Serialize)]` view (core.rs:25; cartpole.rs:51-87, mountain_car.rs:46-80) -------- */
/* What serde_json::to_string(&env) prints for the reference env that lane `lane` stands for: the serde-visible
 * fields in declaration order with the reference's field names -- CartPole: action_space, observation_space
 * {low, high}, render_mode, state {x, x_dot, theta, theta_dot}, metadata {render_modes, render_fps, marker},
 * gravity, masscart, masspole, length, force_mag, tau, kinematics_integrator ("Euler" | "Other"),
 * theta_threshold_radians, x_threshold, steps_beyond_terminated (null | 0; the reference counts further, the engine
 * keeps is_some()); MountainCar: min_position .. gravity, render_mode, action_space, observation_space, state
 * {position, velocity}, metadata.  Non-finite floats print as null (serde_json).  `rand_random` is
 * #[serde(skip_serializing)] in the reference and absent here; the GUI-only `renderer` / `screen` members are
 * omitted (RenderMode::None, out of scope).  Engine-side additions sit under one extra key "gymrs": {kind, n_envs,
 * global_env_id, flags, seed, tick, max_episode_steps}.  Pendulum (not in the reference) prints its params, state
 * and the "gymrs" object.
 * Writes at most cap bytes incl. the terminating NUL; *needed (may be NULL) receives the size required.  Returns
 * GYMRS_EINVAL when cap is too small (nothing useful written).  Synchronising (reads the lane's state). */
gymrs_status gymrs_env_json(gymrs_engine* e, uint64_t lane, char* buf, uint64_t cap, uint64_t* needed);
/* The inverse for the physics fields: parse a JSON object as printed above (unknown keys ignored, missing keys keep
 * the value already in *params, which the caller initialises, e.g. with gymrs_default_params) into the kind's
 * params struct.  If `state` (may be NULL, capacity 4) is given and the object has a "state", its numbers are stored
 * in field order and *state_dim (may be NULL) is set (0 when absent). */
gymrs_status gymrs_params_from_json(gymrs_env_kind kind, const char* json, void* params, double* state, int* state_dim);

/* ---- utilities --------------------------------------------------------------------------------- */
/* Random-policy actions for lane block [0, n_envs) at time t, written to actions_dev: the
 * `rng.gen_range(0..=1)` of examples/cartpole.rs:19, generated on the device so no PCIe traffic
 * sits in a timed loop.  Philox stream 1, key = seed, counter = (global id, t). */
gymrs_status gymrs_fill_actions(gymrs_engine* e, void* actions_dev, uint64_t seed, uint64_t t);
/* Engine tick (number of reset()/step() calls since the last seeded reset) and current seed. */
gymrs_status gymrs_get_tick(gymrs_engine* e, uint64_t* tick, uint64_t* seed);
/* Kernel tuning knobs for benchmarks; results never depend on them.  lanes_per_thread: 4 (default) or 8
 * lanes per work-item (16 was measured 4x slower everywhere and was removed in ABI 2).  memory_hint: 0 = automatic (every access non-temporal while one step's traffic is
 * <= 48 MiB or >= 1 GiB; in between only the stores nobody reads again -- reward, flags, Pendulum's cos / sin -- so that the Infinity
 * Cache keeps the state for the next step), 1 = every access non-temporal, 2 = none, 3 = only those stores. */
gymrs_status gymrs_set_tuning(gymrs_engine* e, int lanes_per_thread, int memory_hint);

const char* gymrs_last_error(void);
int gymrs_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GYMRS_AMD_H */
