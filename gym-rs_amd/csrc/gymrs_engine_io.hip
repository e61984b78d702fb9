// gymrs_engine_io.hip -- the parts of the C ABI (include/gymrs_amd.h) that look at an engine or copy it rather than step it: views and copies of the
// SoA arrays, clone / snapshot (Env: Clone + Serialize, core.rs:25), episode statistics and their RCCL all-reduce, the pub physics fields after
// construction, the #[derive(Serialize)] JSON view.  The engine object and the stepping paths: gymrs_engine_priv.h, gymrs_engine.hip.
#include "gymrs_engine_priv.h"

static RcclApi g_rccl;

void comm_destroy(gymrs_engine* e)
{
    if (e->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(e->comm);
    e->comm = nullptr;
}

extern "C" {

// ---------------------------------------------------------------------------------------------
gymrs_status gymrs_obs_ptrs(gymrs_engine* e, float** out_ptrs, int* obs_dim)
{
    if (!e || !out_ptrs || !obs_dim) return fail(GYMRS_EINVAL, "gymrs_obs_ptrs: NULL argument");
    if (e->kind == GYMRS_PENDULUM) {
        out_ptrs[0] = e->obs_cos;
        out_ptrs[1] = e->obs_sin;
        out_ptrs[2] = e->s[1]; // theta_dot: the observation column IS the state column
    } else {
        for (int j = 0; j < e->obs_dim; ++j) out_ptrs[j] = e->s[j];
    }
    *obs_dim = e->obs_dim;
    return GYMRS_OK;
}

gymrs_status gymrs_state_ptrs(gymrs_engine* e, float** out_ptrs, int* state_dim)
{
    if (!e || !out_ptrs || !state_dim) return fail(GYMRS_EINVAL, "gymrs_state_ptrs: NULL argument");
    for (int j = 0; j < e->state_dim; ++j) out_ptrs[j] = e->s[j];
    *state_dim = e->state_dim;
    return GYMRS_OK;
}

gymrs_status gymrs_reward_ptr(gymrs_engine* e, float** out)
{
    if (!e || !out) return fail(GYMRS_EINVAL, "gymrs_reward_ptr: NULL argument");
    *out = e->reward;
    return GYMRS_OK;
}

gymrs_status gymrs_done_ptr(gymrs_engine* e, uint8_t** out)
{
    if (!e || !out) return fail(GYMRS_EINVAL, "gymrs_done_ptr: NULL argument");
    *out = e->done;
    return GYMRS_OK;
}

gymrs_status gymrs_truncated_ptr(gymrs_engine* e, uint8_t** out)
{
    if (!e || !out) return fail(GYMRS_EINVAL, "gymrs_truncated_ptr: NULL argument");
    *out = e->truncated;
    return GYMRS_OK;
}

static gymrs_status range_check(const gymrs_engine* e, uint64_t first, uint64_t count, const char* who)
{
    if (first > e->n || count > e->n - first) return fail(GYMRS_EINVAL, std::string(who) + ": lane range out of bounds");
    return GYMRS_OK;
}

gymrs_status gymrs_get_obs(gymrs_engine* e, uint64_t first, uint64_t count, float* host_out)
{
    if (!e || !host_out) return fail(GYMRS_EINVAL, "gymrs_get_obs: NULL argument");
    if (gymrs_status st = range_check(e, first, count, "gymrs_get_obs")) return st;
    HIP_TRY(hipSetDevice(e->device));
    float* ptrs[4];
    int dim = 0;
    gymrs_obs_ptrs(e, ptrs, &dim);
    if (e->pool_host) { // small engine: the arrays ARE host memory (no copy-engine command on the single-env mirror path)
        if (gymrs_status st_ = stream_sync_checked(e)) return st_;
        for (int j = 0; j < dim; ++j) std::memcpy(host_out + (size_t)j * count, host_of(e, ptrs[j]) + first, count * sizeof(float));
        return GYMRS_OK;
    }
    for (int j = 0; j < dim; ++j)
        HIP_TRY(hipMemcpyAsync(host_out + (size_t)j * count, ptrs[j] + first, count * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (gymrs_status st_ = stream_sync_checked(e)) return st_;
    return GYMRS_OK;
}

gymrs_status gymrs_get_state(gymrs_engine* e, uint64_t first, uint64_t count, float* host_out)
{
    if (!e || !host_out) return fail(GYMRS_EINVAL, "gymrs_get_state: NULL argument");
    if (gymrs_status st = range_check(e, first, count, "gymrs_get_state")) return st;
    HIP_TRY(hipSetDevice(e->device));
    if (e->pool_host) {
        if (gymrs_status st_ = stream_sync_checked(e)) return st_;
        for (int j = 0; j < e->state_dim; ++j) std::memcpy(host_out + (size_t)j * count, host_of(e, e->s[j]) + first, count * sizeof(float));
        return GYMRS_OK;
    }
    for (int j = 0; j < e->state_dim; ++j)
        HIP_TRY(hipMemcpyAsync(host_out + (size_t)j * count, e->s[j] + first, count * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (gymrs_status st_ = stream_sync_checked(e)) return st_;
    return GYMRS_OK;
}

gymrs_status gymrs_set_state(gymrs_engine* e, uint64_t first, uint64_t count, const float* host_in)
{
    if (!e || !host_in) return fail(GYMRS_EINVAL, "gymrs_set_state: NULL argument");
    if (gymrs_status st = range_check(e, first, count, "gymrs_set_state")) return st;
    HIP_TRY(hipSetDevice(e->device));
    if (e->pool_host) { // (like the assignment below, this touches nothing but the state)
        HIP_TRY(hipStreamSynchronize(e->stream));
        for (int j = 0; j < e->state_dim; ++j) std::memcpy(host_of(e, e->s[j]) + first, host_in + (size_t)j * count, count * sizeof(float));
        if (e->kind == GYMRS_PENDULUM)
            for (uint64_t i = 0; i < count; ++i) sincosf_(host_in[i], host_of(e, e->obs_sin) + first + i, host_of(e, e->obs_cos) + first + i);
        return GYMRS_OK;
    }
    for (int j = 0; j < e->state_dim; ++j)
        HIP_TRY(hipMemcpyAsync(e->s[j] + first, host_in + (size_t)j * count, count * sizeof(float), hipMemcpyHostToDevice, e->stream));
    // like assigning the pub `state` field in the reference, this touches nothing else: the episode
    // (steps_beyond_terminated, elapsed steps) carries on; call gymrs_reset for a fresh one.
    if (e->kind == GYMRS_PENDULUM) {
        // keep the observation columns consistent with the new state
        std::vector<float> c(count), s(count);
        for (uint64_t i = 0; i < count; ++i) sincosf_(host_in[i], &s[i], &c[i]);
        HIP_TRY(hipMemcpyAsync(e->obs_cos + first, c.data(), count * sizeof(float), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(e->obs_sin + first, s.data(), count * sizeof(float), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    return GYMRS_OK;
}

gymrs_status gymrs_get_step_result(gymrs_engine* e, uint64_t first, uint64_t count, float* reward, uint8_t* done,
                                   uint8_t* truncated)
{
    if (!e) return fail(GYMRS_EINVAL, "gymrs_get_step_result: engine is NULL");
    if (gymrs_status st = range_check(e, first, count, "gymrs_get_step_result")) return st;
    HIP_TRY(hipSetDevice(e->device));
    if (e->pool_host) {
        if (gymrs_status st_ = stream_sync_checked(e)) return st_;
        if (reward) std::memcpy(reward, host_of(e, e->reward) + first, count * sizeof(float));
        if (done) std::memcpy(done, host_of(e, e->done) + first, count);
        if (truncated) std::memcpy(truncated, host_of(e, e->truncated) + first, count);
        return GYMRS_OK;
    }
    if (reward) HIP_TRY(hipMemcpyAsync(reward, e->reward + first, count * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (done) HIP_TRY(hipMemcpyAsync(done, e->done + first, count, hipMemcpyDeviceToHost, e->stream));
    if (truncated) HIP_TRY(hipMemcpyAsync(truncated, e->truncated + first, count, hipMemcpyDeviceToHost, e->stream));
    if (gymrs_status st_ = stream_sync_checked(e)) return st_;
    return GYMRS_OK;
}

// ---------------------------------------------------------------------------------------------
// `Env: Clone + Serialize` (core.rs:25).  Clone = a second engine with a deep copy of everything a step can
// observe (lane arrays, episode bookkeeping, statistics, RNG position = seed + tick).  The snapshot is the
// same content as a host blob; a restored engine continues bit-identically to the uninterrupted run.
extern "C++" {
namespace {
struct SnapshotHeader {
    char magic[8]; // "GYMRSNAP"
    uint32_t version, kind;
    uint64_t n, gid0;
    uint32_t flags, state_dim, epoch, n_stat_blocks, max_steps, open_vec;
    uint64_t seed, tick, uniform_start;
    double n_steps_total;
    float lo[4], hi[4], max_torque;
    uint32_t consts_bytes;
    unsigned char consts[96];
    unsigned char params[96]; // the f64 pub fields (gymrs_get_params / the serde view)
};
static_assert(sizeof(CartPoleConsts) <= 96 && sizeof(MountainCarConsts) <= 96 && sizeof(PendulumConsts) <= 96, "consts blob too small");
static_assert(sizeof(gymrs_cartpole_params) <= 96 && sizeof(gymrs_mountain_car_params) <= 96 && sizeof(gymrs_pendulum_params) <= 96, "params blob too small");
constexpr uint32_t kSnapshotVersion = 4; // 4: the statistics baseline is {L, E, R} (round 6), not L alone

struct Segment {
    void* dev;
    size_t bytes;
};
// the device arrays of a snapshot, in blob order
std::vector<Segment> snapshot_segments(const gymrs_engine* e)
{
    std::vector<Segment> v;
    const size_t n = (size_t)e->n;
    for (int j = 0; j < e->state_dim; ++j) v.push_back({e->s[j], n * 4});
    if (e->obs_cos) v.push_back({e->obs_cos, n * 4});
    if (e->obs_sin) v.push_back({e->obs_sin, n * 4});
    v.push_back({e->reward, n * 4});
    v.push_back({e->done, n});
    v.push_back({e->truncated, n});
    v.push_back({e->beyond, n});
    v.push_back({e->ep_start, n * 4});
    v.push_back({e->block_stats, (size_t)e->n_stat_blocks * 2 * sizeof(unsigned long long)});
    v.push_back({e->wave_open, (size_t)e->n_stat_blocks * sizeof(double)});
    v.push_back({e->stats_base, (size_t)kStatsBaseWords * sizeof(unsigned long long)});
    v.push_back({e->err, 2 * sizeof(uint32_t)});
    return v;
}
size_t consts_size(gymrs_env_kind k)
{
    return k == GYMRS_CARTPOLE ? sizeof(CartPoleConsts) : (k == GYMRS_MOUNTAIN_CAR ? sizeof(MountainCarConsts) : sizeof(PendulumConsts));
}
void drop_graph(gymrs_engine* e)
{
    if (e->graph_exec) {
        (void)hipGraphExecDestroy(e->graph_exec);
        e->graph_exec = nullptr;
    }
}
// the host-side scalars a step depends on
void copy_scalars(gymrs_engine* dst, const gymrs_engine* src)
{
    dst->consts = src->consts;
    dst->params = src->params;
    dst->max_torque = src->max_torque;
    dst->max_steps = src->max_steps;
    std::memcpy(dst->lo, src->lo, sizeof(dst->lo));
    std::memcpy(dst->hi, src->hi, sizeof(dst->hi));
    std::memcpy(dst->dflt_lo, src->dflt_lo, sizeof(dst->dflt_lo));
    std::memcpy(dst->dflt_hi, src->dflt_hi, sizeof(dst->dflt_hi));
    dst->epoch = src->epoch;
    dst->seed = src->seed;
    dst->tick = src->tick;
    dst->uniform_start = src->uniform_start;
    dst->n_steps_total = src->n_steps_total;
    dst->vec = src->vec;
    dst->nt_mode = src->nt_mode;
    dst->open_vec = src->open_vec;
    dst->trunc_held = src->trunc_held;
    // the clone's arrays are copies of the source's: what is known about them carries over (its own refresh starts afresh)
    limit_restart(dst, src->start_bound, src->trunc_zero);
    *dst->err_seen = 1u; // the error words are copied with the arrays: have the next gymrs_sync look at them
}
} // namespace
} // extern "C++"

gymrs_status gymrs_engine_clone(gymrs_engine* src, gymrs_engine** out)
{
    if (!src || !out) return fail(GYMRS_EINVAL, "gymrs_engine_clone: NULL argument");
    *out = nullptr;
    gymrs_engine* dst = nullptr;
    if (gymrs_status st = gymrs_engine_create(src->kind, src->n, src->gid0, src->device, nullptr, src->flags, &dst)) return st;
    copy_scalars(dst, src);
    if (gymrs_status st = fold_reset_log(src)) { // ep_start and the episode counters are copied below
        gymrs_engine_destroy(dst);
        return st;
    }
    if (gymrs_status st = stream_sync_checked(src)) { // everything queued on the source has happened -- and no chain of it tripped a check
        gymrs_engine_destroy(dst);
        return st;
    }
    hipError_t err = hipSuccess;
    const std::vector<Segment> from = snapshot_segments(src), to = snapshot_segments(dst);
    for (size_t i = 0; i < from.size() && err == hipSuccess; ++i)
        err = hipMemcpyAsync(to[i].dev, from[i].dev, from[i].bytes, hipMemcpyDefault, dst->stream); // (a small engine's pool is mapped host memory)
    if (err == hipSuccess) err = hipStreamSynchronize(dst->stream);
    if (err != hipSuccess) {
        gymrs_engine_destroy(dst);
        return fail(GYMRS_EHIP, std::string("gymrs_engine_clone: ") + hipGetErrorString(err));
    }
    *out = dst;
    return GYMRS_OK;
}

gymrs_status gymrs_snapshot_size(gymrs_engine* e, uint64_t* bytes)
{
    if (!e || !bytes) return fail(GYMRS_EINVAL, "gymrs_snapshot_size: NULL argument");
    size_t total = sizeof(SnapshotHeader);
    for (const Segment& sg : snapshot_segments(e)) total += sg.bytes;
    *bytes = total;
    return GYMRS_OK;
}

gymrs_status gymrs_snapshot_save(gymrs_engine* e, void* host_buf, uint64_t bytes)
{
    if (!e || !host_buf) return fail(GYMRS_EINVAL, "gymrs_snapshot_save: NULL argument");
    uint64_t need = 0;
    (void)gymrs_snapshot_size(e, &need);
    if (bytes < need) return fail(GYMRS_EINVAL, "gymrs_snapshot_save: buffer smaller than gymrs_snapshot_size");
    HIP_TRY(hipSetDevice(e->device));
    if (gymrs_status st = fold_reset_log(e)) return st; // the blob holds ep_start and the counters, never the log
    SnapshotHeader h;
    std::memset(&h, 0, sizeof(h));
    std::memcpy(h.magic, "GYMRSNAP", 8);
    h.version = kSnapshotVersion;
    h.kind = (uint32_t)e->kind;
    h.n = e->n;
    h.gid0 = e->gid0;
    h.flags = e->flags;
    h.state_dim = (uint32_t)e->state_dim;
    h.epoch = e->epoch;
    h.n_stat_blocks = e->n_stat_blocks;
    h.max_steps = e->max_steps;
    h.open_vec = (uint32_t)e->open_vec; // lanes per work-item the open-episode sums were last updated with
    h.seed = e->seed;
    h.tick = e->tick;
    h.uniform_start = e->uniform_start;
    h.n_steps_total = e->n_steps_total;
    std::memcpy(h.lo, e->lo, sizeof(h.lo));
    std::memcpy(h.hi, e->hi, sizeof(h.hi));
    h.max_torque = e->max_torque;
    h.consts_bytes = (uint32_t)consts_size(e->kind);
    std::memcpy(h.consts, consts_ptr(e), h.consts_bytes);
    std::memcpy(h.params, &e->params, sizeof(e->params));
    char* p = static_cast<char*>(host_buf);
    std::memcpy(p, &h, sizeof(h));
    p += sizeof(h);
    for (const Segment& sg : snapshot_segments(e)) {
        HIP_TRY(hipMemcpyAsync(p, sg.dev, sg.bytes, hipMemcpyDefault, e->stream));
        p += sg.bytes;
    }
    if (gymrs_status st_ = stream_sync_checked(e)) return st_;
    return GYMRS_OK;
}

gymrs_status gymrs_snapshot_load(gymrs_engine* e, const void* host_buf, uint64_t bytes)
{
    if (!e || !host_buf) return fail(GYMRS_EINVAL, "gymrs_snapshot_load: NULL argument");
    if (bytes < sizeof(SnapshotHeader)) return fail(GYMRS_EINVAL, "gymrs_snapshot_load: truncated snapshot");
    SnapshotHeader h;
    std::memcpy(&h, host_buf, sizeof(h));
    if (std::memcmp(h.magic, "GYMRSNAP", 8) != 0 || h.version != kSnapshotVersion)
        return fail(GYMRS_EINVAL, "gymrs_snapshot_load: not a gymrs snapshot of this version");
    if (h.kind != (uint32_t)e->kind || h.n != e->n || h.flags != e->flags || h.state_dim != (uint32_t)e->state_dim ||
        h.n_stat_blocks != e->n_stat_blocks || h.consts_bytes != consts_size(e->kind))
        return fail(GYMRS_EINVAL, "gymrs_snapshot_load: snapshot was taken from an engine of another kind / size / flags");
    uint64_t need = 0;
    (void)gymrs_snapshot_size(e, &need);
    if (bytes < need) return fail(GYMRS_EINVAL, "gymrs_snapshot_load: truncated snapshot");
    HIP_TRY(hipSetDevice(e->device));
    drop_graph(e); // seed and reset box are baked into a captured graph
    if (gymrs_status st = fold_reset_log(e)) return st; // empties the ring; what it folded into is overwritten below
    const char* p = static_cast<const char*>(host_buf) + sizeof(h);
    for (const Segment& sg : snapshot_segments(e)) {
        HIP_TRY(hipMemcpyAsync(sg.dev, p, sg.bytes, hipMemcpyDefault, e->stream));
        p += sg.bytes;
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->gid0 = h.gid0;
    e->epoch = h.epoch;
    e->max_steps = h.max_steps;
    e->seed = h.seed;
    e->tick = h.tick;
    e->uniform_start = h.uniform_start;
    e->n_steps_total = h.n_steps_total;
    std::memcpy(e->lo, h.lo, sizeof(e->lo));
    std::memcpy(e->hi, h.hi, sizeof(e->hi));
    e->max_torque = h.max_torque;
    e->open_vec = (int)h.open_vec;
    e->trunc_held = -1; // whatever the arrays held before the load: rewrite the flags on the next step
    limit_restart(e, 0, false); // nothing is known about the loaded episode clocks until a refresh has looked
    *e->err_seen = 1u;          // the error words came with the blob: have the next gymrs_sync look at them
    HIP_TRY(hipMemsetAsync(e->wave_clean, 0, (size_t)e->n_stat_blocks * sizeof(uint32_t), e->stream)); // and the rewards
    e->clean_shape = 0;
    std::memcpy(&e->consts, h.consts, h.consts_bytes);
    std::memcpy(&e->params, h.params, sizeof(e->params));
    return GYMRS_OK;
}

// ---------------------------------------------------------------------------------------------
gymrs_status gymrs_stats_device(gymrs_engine* e, double** dev_out4)
{
    if (!e || !dev_out4) return fail(GYMRS_EINVAL, "gymrs_stats_device: NULL argument");
    HIP_TRY(hipSetDevice(e->device));
    // (no fold: the read-out looks through the reset log's pending rows and writes nothing a step kernel reads -- launch_stats, gymrs_kernels.h)
    HIP_TRY(launch_stats(stats_args(e), 0, e->stream));
    *dev_out4 = e->stats_dev;
    return GYMRS_OK;
}

gymrs_status gymrs_stats(gymrs_engine* e, double out[4])
{
    if (!e || !out) return fail(GYMRS_EINVAL, "gymrs_stats: NULL argument");
    double* dev = nullptr;
    if (gymrs_status st = gymrs_stats_device(e, &dev)) return st;
    if (gymrs_status st_ = stream_sync_checked(e)) return st_; // the read-out kernel has written the four doubles into mapped host memory
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int j = 0; j < 4; ++j) out[j] = e->stats_host[j];
    return GYMRS_OK;
}

gymrs_status gymrs_stats_clear(gymrs_engine* e)
{
    if (!e) return fail(GYMRS_EINVAL, "gymrs_stats_clear: engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    // Baseline, not zeroing (round 6): one read-only pass remembers {sum of start ticks, finished episodes, sum of returns} as they are NOW -- pending rows
    // of the reset log included -- and every later read-out subtracts them.  Nothing the step kernels read is written from the stream.
    HIP_TRY(launch_stats(stats_args(e), 1, e->stream));
    e->n_steps_total = 0;
    return GYMRS_OK;
}

// ---- RCCL (one process per GPU; ncclAllReduce of 4 doubles over xGMI) --------------------------
static gymrs_status rccl_load()
{
    if (g_rccl.lib) return GYMRS_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* nm : names) {
        lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return fail(GYMRS_ENCCL, std::string("cannot load librccl: ") + dlerror());
    RcclApi api;
    api.lib = lib;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(lib, "ncclAllReduce"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(lib, "ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy)
        return fail(GYMRS_ENCCL, "librccl lacks an expected nccl* symbol");
    g_rccl = api;
    return GYMRS_OK;
}

static gymrs_status nccl_fail(const char* what, int rc)
{
    return fail(GYMRS_ENCCL, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"));
}

gymrs_status gymrs_comm_unique_id(uint8_t id_out[128])
{
    if (!id_out) return fail(GYMRS_EINVAL, "gymrs_comm_unique_id: NULL argument");
    if (gymrs_status st = rccl_load()) return st;
    if (int rc = g_rccl.GetUniqueId(id_out)) return nccl_fail("ncclGetUniqueId", rc);
    return GYMRS_OK;
}

gymrs_status gymrs_comm_init(gymrs_engine* e, int n_ranks, int rank, const uint8_t id[128])
{
    if (!e || !id) return fail(GYMRS_EINVAL, "gymrs_comm_init: NULL argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(GYMRS_EINVAL, "gymrs_comm_init: bad rank / n_ranks");
    if (gymrs_status st = rccl_load()) return st;
    HIP_TRY(hipSetDevice(e->device));
    NcclId128 uid;
    std::memcpy(uid.internal, id, 128);
    comm_destroy(e); // (a communicator of an earlier call, e.g. the in-process sharder's)
    if (int rc = g_rccl.CommInitRank(&e->comm, n_ranks, uid, rank)) return nccl_fail("ncclCommInitRank", rc);
    e->n_ranks = n_ranks;
    e->comm_rank = rank;
    return GYMRS_OK;
}

gymrs_status gymrs_allreduce_stats(gymrs_engine* e, double out[4])
{
    if (!e || !out) return fail(GYMRS_EINVAL, "gymrs_allreduce_stats: NULL argument");
    if (!e->comm) return fail(GYMRS_ENCCL, "gymrs_allreduce_stats: call gymrs_comm_init first");
    double* dev = nullptr;
    if (gymrs_status st = gymrs_stats_device(e, &dev)) return st;
    // ncclFloat64 = 8, ncclSum = 0 (rccl.h); 32 bytes per rank: latency-bound, rides the engine stream
    if (int rc = g_rccl.AllReduce(dev, dev, 4, 8, 0, e->comm, e->stream)) return nccl_fail("ncclAllReduce", rc);
    HIP_TRY(hipMemcpyAsync(out, dev, 4 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    return stream_sync_checked(e); // (like every synchronising read-out: a tripped chain fails THIS call)
}

// The in-process form (`gymrs_allreduce_stats(gymrs_engine** shards, int n, double out[4])`): ONE host thread holds every shard of a
// batch, one engine per GPU.  On distinct devices the sum is an RCCL all-reduce over xGMI: the communicators are made once, all ranks in one
// ncclGroupStart / ncclGroupEnd (what ncclCommInitAll does; a bare ncclCommInitRank per engine from one thread would wait for the other ranks
// for ever), and every later call issues the n all-reduces of 32 bytes inside one group, each on its engine's stream.  Where two shards share
// a device RCCL refuses the communicator ("duplicate GPU"), and nothing has to cross a link: the same four doubles are summed on the host.
// Either way every engine's statistics read-out runs on its own stream first and `out` holds the batch's totals when the call returns.
// stream_sync_checked with the shard named in the message: a read-out over several engines must say WHICH one a tripped chain belongs to
static gymrs_status sync_shard_checked(gymrs_engine* e, int index)
{
    const gymrs_status st = stream_sync_checked(e);
    if (st == GYMRS_OK) return st;
    return fail(st, "shard " + std::to_string(index) + " (device " + std::to_string(e->device) + "): " + gymrs_last_error());
}

// every engine's read-out on its own stream, the four doubles summed on the host
static gymrs_status host_sum_stats(gymrs_engine** shards, int n, double out[4])
{
    double total[4] = {0, 0, 0, 0};
    std::vector<double*> dev((size_t)n);
    for (int r = 0; r < n; ++r) // every read-out kernel is enqueued before the first wait
        if (gymrs_status st = gymrs_stats_device(shards[r], &dev[(size_t)r])) return st;
    for (int r = 0; r < n; ++r) {
        HIP_TRY(hipSetDevice(shards[r]->device));
        if (gymrs_status st = sync_shard_checked(shards[r], r)) return st;
        for (int j = 0; j < 4; ++j) total[j] += shards[r]->stats_host[j];
    }
    for (int j = 0; j < 4; ++j) out[j] = total[j];
    return GYMRS_OK;
}

gymrs_status gymrs_allreduce_stats_multi(gymrs_engine** shards, int n, double out[4], int* used_rccl)
{
    if (!shards || n < 1 || !out) return fail(GYMRS_EINVAL, "gymrs_allreduce_stats_multi: NULL argument or n < 1");
    if (used_rccl) *used_rccl = 0;
    bool distinct = true;
    for (int r = 0; r < n; ++r) {
        if (!shards[r]) return fail(GYMRS_EINVAL, "gymrs_allreduce_stats_multi: a shard is NULL");
        for (int q = 0; q < r; ++q) {
            if (shards[q] == shards[r]) return fail(GYMRS_EINVAL, "gymrs_allreduce_stats_multi: the same engine appears twice");
            distinct = distinct && shards[q]->device != shards[r]->device;
        }
    }
    // (test hook gymrs_dev_set_hooks bit 4 on shard 0: take the RCCL branch even for ONE shard -- a one-rank communicator made inside a group and a grouped
    // all-reduce -- so that the grouped code path runs on a one-GPU test box at all: tests/test_gpu_sharded_native.py)
    const bool force_rccl = n == 1 && (shards[0]->dev_hooks & 16u) != 0;
    if ((n == 1 && !force_rccl) || !distinct) return host_sum_stats(shards, n, out); // host-side sum: same interface, no link to cross
    // First contact with RCCL must not cost the caller its result (VERDICT r5 "next" #4: a first run on 8 devices should be boring): when the library
    // cannot be loaded or the grouped communicator cannot be made -- BEFORE any collective has been enqueued anywhere -- the same four doubles are summed on
    // the host, *used_rccl = -1 says so and gymrs_last_error() keeps RCCL's own message.  A failure of an all-reduce that was already enqueued is an error.
    auto fall_back = [&](gymrs_status why) -> gymrs_status {
        const std::string msg = gymrs_last_error();
        (void)why;
        if (gymrs_status st = host_sum_stats(shards, n, out)) return st;
        if (used_rccl) *used_rccl = -1;
        (void)fail(GYMRS_OK, "RCCL unavailable, statistics summed on the host: " + msg);
        return GYMRS_OK;
    };
    if ((shards[0]->dev_hooks & 32u) != 0) return fall_back(fail(GYMRS_ENCCL, "test hook: RCCL treated as unavailable")); // (tests: the fall-back itself)
    if (gymrs_status st = rccl_load()) return fall_back(st);
    if (!g_rccl.GroupStart || !g_rccl.GroupEnd) return fall_back(fail(GYMRS_ENCCL, "librccl lacks ncclGroupStart / ncclGroupEnd"));
    bool have = true;
    for (int r = 0; r < n; ++r) have = have && shards[r]->comm && shards[r]->n_ranks == n && shards[r]->comm_rank == r;
    if (!have) { // first call for this set of shards (or another set before): one communicator over exactly these engines, rank = index
        for (int r = 0; r < n; ++r) comm_destroy(shards[r]);
        NcclId128 uid;
        if (int rc = g_rccl.GetUniqueId(&uid)) return fall_back(nccl_fail("ncclGetUniqueId", rc));
        if (int rc = g_rccl.GroupStart()) return fall_back(nccl_fail("ncclGroupStart", rc));
        int bad = 0;
        hipError_t herr = hipSuccess; // (no early return between ncclGroupStart and ncclGroupEnd: an open group would swallow every later call of this thread)
        for (int r = 0; r < n && !bad && herr == hipSuccess; ++r) {
            herr = hipSetDevice(shards[r]->device);
            if (herr == hipSuccess) bad = g_rccl.CommInitRank(&shards[r]->comm, n, uid, r);
        }
        const int end = g_rccl.GroupEnd();
        if (bad || end || herr != hipSuccess) {
            for (int r = 0; r < n; ++r) shards[r]->comm = nullptr; // (a communicator of a failed group is not one to destroy)
            if (herr != hipSuccess) return fail(GYMRS_EHIP, std::string("gymrs_allreduce_stats_multi: hipSetDevice: ") + hipGetErrorString(herr));
            return fall_back(nccl_fail("ncclCommInitRank (grouped, one rank per device)", bad ? bad : end));
        }
        for (int r = 0; r < n; ++r) {
            shards[r]->n_ranks = n;
            shards[r]->comm_rank = r;
        }
    }
    std::vector<double*> dev((size_t)n);
    for (int r = 0; r < n; ++r)
        if (gymrs_status st = gymrs_stats_device(shards[r], &dev[(size_t)r])) return st;
    if (int rc = g_rccl.GroupStart()) return nccl_fail("ncclGroupStart", rc);
    int bad = 0;
    hipError_t herr = hipSuccess;
    for (int r = 0; r < n && !bad && herr == hipSuccess; ++r) {
        herr = hipSetDevice(shards[r]->device);
        if (herr == hipSuccess)
            bad = g_rccl.AllReduce(dev[(size_t)r], dev[(size_t)r], 4, 8 /* ncclFloat64 */, 0 /* ncclSum */, shards[r]->comm, shards[r]->stream);
    }
    const int end = g_rccl.GroupEnd();
    if (herr != hipSuccess) return fail(GYMRS_EHIP, std::string("gymrs_allreduce_stats_multi: hipSetDevice: ") + hipGetErrorString(herr));
    if (bad || end) return nccl_fail("ncclAllReduce (grouped)", bad ? bad : end);
    HIP_TRY(hipSetDevice(shards[0]->device));
    HIP_TRY(hipMemcpyAsync(out, dev[0], 4 * sizeof(double), hipMemcpyDeviceToHost, shards[0]->stream));
    gymrs_status first_bad = GYMRS_OK;
    std::string first_msg;
    for (int r = 0; r < n; ++r) { // every rank's copy of the sum is complete when the call returns (every stream is waited for, the FIRST failure is reported)
        HIP_TRY(hipSetDevice(shards[r]->device));
        if (gymrs_status st = sync_shard_checked(shards[r], r)) {
            if (first_bad == GYMRS_OK) {
                first_bad = st;
                first_msg = gymrs_last_error();
            }
        }
    }
    if (first_bad != GYMRS_OK) return fail(first_bad, first_msg);
    if (used_rccl) *used_rccl = 1;
    return GYMRS_OK;
}

// ---------------------------------------------------------------------------------------------
gymrs_status gymrs_fill_actions(gymrs_engine* e, void* actions_dev, uint64_t seed, uint64_t t)
{
    if (!e || !actions_dev) return fail(GYMRS_EINVAL, "gymrs_fill_actions: NULL argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(launch_fill_actions(e->kind, actions_dev, e->n, e->gid0, seed, t, e->max_torque, e->stream));
    return GYMRS_OK;
}

// ---- the pub physics fields after construction -------------------------------------------------------------------
gymrs_status gymrs_set_params(gymrs_engine* e, const void* params)
{
    if (!e || !params) return fail(GYMRS_EINVAL, "gymrs_set_params: NULL argument");
    if (gymrs_status st = check_params(e->kind, params, "gymrs_set_params")) return st;
    switch (e->kind) {
    case GYMRS_CARTPOLE:
        e->params.cp = *static_cast<const gymrs_cartpole_params*>(params);
        e->consts.cp = make_consts(e->params.cp);
        break;
    case GYMRS_MOUNTAIN_CAR:
        e->params.mc = *static_cast<const gymrs_mountain_car_params*>(params);
        e->consts.mc = make_consts(e->params.mc);
        break;
    case GYMRS_PENDULUM:
        e->params.pd = *static_cast<const gymrs_pendulum_params*>(params);
        e->consts.pd = make_consts(e->params.pd);
        e->max_steps = e->consts.pd.max_steps;
        e->max_torque = (float)e->params.pd.max_torque;
        break;
    }
    // Only the launch constants changed: state, steps_beyond_terminated, episode clocks, statistics, seed and tick
    // carry on, exactly like assigning a pub field of the reference struct between two step() calls.
    drop_graph(e); // the constants are baked into a captured graph
    return GYMRS_OK;
}

gymrs_status gymrs_get_params(gymrs_engine* e, void* params_out)
{
    if (!e || !params_out) return fail(GYMRS_EINVAL, "gymrs_get_params: NULL argument");
    switch (e->kind) {
    case GYMRS_CARTPOLE: *static_cast<gymrs_cartpole_params*>(params_out) = e->params.cp; break;
    case GYMRS_MOUNTAIN_CAR: *static_cast<gymrs_mountain_car_params*>(params_out) = e->params.mc; break;
    case GYMRS_PENDULUM: *static_cast<gymrs_pendulum_params*>(params_out) = e->params.pd; break;
    }
    return GYMRS_OK;
}

// ---- #[derive(Serialize)] view -------------------------------------------------------------------------------------
extern "C++" {
namespace {
json::Object engine_extras(const gymrs_engine* e, uint64_t lane, uint32_t max_episode_steps)
{
    json::Object g;
    g.str("kind", e->kind == GYMRS_CARTPOLE ? "CartPole" : (e->kind == GYMRS_MOUNTAIN_CAR ? "MountainCar" : "Pendulum"));
    g.uint("n_envs", e->n).uint("global_env_id", e->gid0 + lane).uint("flags", e->flags).uint("seed", e->seed).uint("tick", e->tick);
    g.uint("max_episode_steps", max_episode_steps);
    // chains of per-step launches that went through the engine's own AQL dispatcher (gymrs_aql.h), and why not if none can
    g.uint("aql_chains", e->aql_chains).uint("aql_launches", e->aql_launches);
    if (e->aql) g.str("aql_handover", e->aql_handover.c_str());
    if (e->last_path != 0) { // the most recent per-step launch as a kernel trace names it
        const int threads = step_threads_of(e->kind, e->n, e->vec);
        const uint32_t h = e->last_flags & kFlagHintMask;
        const char* hint = (h & kFlagNonTemporal) ? "nt" : (h == (kFlagNtOut | kFlagNtStateLoads) ? "so" : (h == kFlagNtOut ? "o" : "pl"));
        char buf[160];
        if (e->last_path == 2)
            std::snprintf(buf, sizeof(buf), "chain: %s", aql_kernel_name(e, e->last_flags, threads).c_str());
        else
            std::snprintf(buf, sizeof(buf), "HIP launch: gymrs::step_kernel<%s, %d, flags %u | hint %s, %d work-items>",
                          e->kind == GYMRS_CARTPOLE ? "CartPoleT" : (e->kind == GYMRS_MOUNTAIN_CAR ? "MountainCarT" : "PendulumT"), e->vec,
                          e->last_flags & 7u, hint, threads);
        g.str("last_launch", buf);
    }
    g.str("aql", e->aql ? "on" : (e->aql_tried ? e->aql_why.c_str() : "not tried"));
    // 1: the per-step kernel does not rewrite a wave's part of `reward` while it holds the env's constant reward (MountainCar always, CartPole from 128 MiB per step on)
    g.uint("reward_store_elided", ((e->flags & GYMRS_AUTO_RESET) && (e->kind == GYMRS_MOUNTAIN_CAR || e->elide_reward)) ? 1 : 0);
    if (e->limit_elidable) { // diagnostics of the time-limit elision: launches that ran without the limit, bound refreshes
        g.uint("time_limit_elided_launches", e->limit_elided_launches).uint("time_limit_refreshes", e->age_refreshes);
        g.uint("time_limit_waits", e->age_waits).uint("time_limit_wait_us", e->age_wait_ns / 1000);
    }
    return g;
}
json::Object metadata_json(std::initializer_list<const char*> modes, unsigned fps)
{
    std::string arr = "[";
    for (const char* m : modes) arr += (arr.size() > 1 ? "," : "") + json::quoted(m);
    arr += "]";
    json::Object m;
    m.raw("render_modes", arr).uint("render_fps", fps).null("marker"); // PhantomData serialises as a unit
    return m;
}
} // namespace
} // extern "C++"

gymrs_status gymrs_env_json(gymrs_engine* e, uint64_t lane, char* buf, uint64_t cap, uint64_t* needed)
{
    if (!e || (!buf && cap != 0)) return fail(GYMRS_EINVAL, "gymrs_env_json: NULL argument");
    if (lane >= e->n) return fail(GYMRS_EINVAL, "gymrs_env_json: lane out of range");
    HIP_TRY(hipSetDevice(e->device));
    float st[4] = {0, 0, 0, 0};
    uint8_t beyond = 0;
    if (e->pool_host) { // small engine: plain loads from the mapped pool
        if (gymrs_status st_ = stream_sync_checked(e)) return st_;
        for (int j = 0; j < e->state_dim; ++j) st[j] = host_of(e, e->s[j])[lane];
        if (e->kind == GYMRS_CARTPOLE) beyond = host_of(e, e->beyond)[lane];
    } else {
        for (int j = 0; j < e->state_dim; ++j) HIP_TRY(hipMemcpyAsync(&st[j], e->s[j] + lane, sizeof(float), hipMemcpyDeviceToHost, e->stream));
        if (e->kind == GYMRS_CARTPOLE) HIP_TRY(hipMemcpyAsync(&beyond, e->beyond + lane, 1, hipMemcpyDeviceToHost, e->stream));
        if (gymrs_status st_ = stream_sync_checked(e)) return st_;
    }
    double low[4], high[4];
    int dim = 0;
    json::Object o;
    switch (e->kind) {
    case GYMRS_CARTPOLE: { // field order of cartpole.rs:52-87
        const gymrs_cartpole_params& p = e->params.cp;
        gymrs_observation_space(e->kind, &p, low, high, &dim);
        static const char* names[4] = {"x", "x_dot", "theta", "theta_dot"};
        json::Object lo, hi, state;
        for (int j = 0; j < 4; ++j) {
            lo.num(names[j], low[j]);
            hi.num(names[j], high[j]);
            state.num(names[j], (double)st[j]);
        }
        json::Object space;
        space.obj("low", lo).obj("high", hi);
        o.uint("action_space", 2).obj("observation_space", space).str("render_mode", "None").obj("state", state);
        o.obj("metadata", metadata_json({"Human", "RgbArray"}, 50)); // cartpole.rs:265-270
        o.num("gravity", p.gravity).num("masscart", p.masscart).num("masspole", p.masspole).num("length", p.length);
        o.num("force_mag", p.force_mag).num("tau", p.tau).str("kinematics_integrator", p.kinematics_integrator == 0 ? "Euler" : "Other");
        o.num("theta_threshold_radians", p.theta_threshold_radians).num("x_threshold", p.x_threshold);
        // Option<usize>: the engine keeps is_some() (the count only feeds a warning in the reference, cartpole.rs:459-463)
        if (beyond && !(e->flags & GYMRS_AUTO_RESET))
            o.uint("steps_beyond_terminated", 0);
        else
            o.null("steps_beyond_terminated");
        o.obj("gymrs", engine_extras(e, lane, e->consts.cp.max_steps));
        break;
    }
    case GYMRS_MOUNTAIN_CAR: { // field order of mountain_car.rs:48-80
        const gymrs_mountain_car_params& p = e->params.mc;
        gymrs_observation_space(e->kind, &p, low, high, &dim);
        o.num("min_position", p.min_position).num("max_position", p.max_position).num("max_speed", p.max_speed);
        o.num("goal_position", p.goal_position).num("goal_velocity", p.goal_velocity).num("force", p.force).num("gravity", p.gravity);
        json::Object lo, hi, state, space;
        lo.num("position", low[0]).num("velocity", low[1]);
        hi.num("position", high[0]).num("velocity", high[1]);
        state.num("position", (double)st[0]).num("velocity", (double)st[1]);
        space.obj("low", lo).obj("high", hi);
        o.str("render_mode", "None").uint("action_space", 3).obj("observation_space", space).obj("state", state);
        o.obj("metadata", metadata_json({"Human", "RgbArray", "SingleRgbArray", "None"}, 30)); // mountain_car.rs:108-118
        o.obj("gymrs", engine_extras(e, lane, e->consts.mc.max_steps));
        break;
    }
    case GYMRS_PENDULUM: { // not in the reference: params, state, engine extras
        const gymrs_pendulum_params& p = e->params.pd;
        o.num("max_speed", p.max_speed).num("max_torque", p.max_torque).num("dt", p.dt).num("g", p.g).num("m", p.m).num("l", p.l);
        json::Object state;
        state.num("theta", (double)st[0]).num("theta_dot", (double)st[1]);
        o.str("render_mode", "None").obj("state", state);
        o.obj("gymrs", engine_extras(e, lane, e->consts.pd.max_steps));
        break;
    }
    }
    const std::string text = o.text();
    if (needed) *needed = text.size() + 1;
    if (cap < text.size() + 1) return fail(GYMRS_EINVAL, "gymrs_env_json: buffer too small (see *needed)");
    std::memcpy(buf, text.c_str(), text.size() + 1);
    return GYMRS_OK;
}

gymrs_status gymrs_params_from_json(gymrs_env_kind kind, const char* text, void* params, double* state, int* state_dim)
{
    if (!text || !params) return fail(GYMRS_EINVAL, "gymrs_params_from_json: NULL argument");
    if (kind != GYMRS_CARTPOLE && kind != GYMRS_MOUNTAIN_CAR && kind != GYMRS_PENDULUM)
        return fail(GYMRS_EINVAL, "gymrs_params_from_json: unknown env kind");
    json::Value root;
    json::Parser parser(text);
    if (!parser.parse(root) || root.kind != json::Value::ObjectK)
        return fail(GYMRS_EINVAL, std::string("gymrs_params_from_json: not a JSON object (stopped at offset ") +
                                      std::to_string(parser.where() - text) + ")");
    bool bad = false;
    auto number = [&](const char* key, double& dst) {
        const json::Value* v = root.get(key);
        if (!v) return;
        if (v->kind == json::Value::Number)
            dst = v->number;
        else if (v->kind == json::Value::Null)
            dst = std::numeric_limits<double>::quiet_NaN(); // serde_json prints non-finite floats as null
        else
            bad = true;
    };
    uint32_t* max_steps = nullptr;
    switch (kind) {
    case GYMRS_CARTPOLE: {
        auto* p = static_cast<gymrs_cartpole_params*>(params);
        number("gravity", p->gravity);
        number("masscart", p->masscart);
        number("masspole", p->masspole);
        number("length", p->length);
        number("force_mag", p->force_mag);
        number("tau", p->tau);
        number("theta_threshold_radians", p->theta_threshold_radians);
        number("x_threshold", p->x_threshold);
        if (const json::Value* v = root.get("kinematics_integrator")) {
            if (v->kind == json::Value::String && v->string == "Euler")
                p->kinematics_integrator = 0;
            else if (v->kind == json::Value::String && v->string == "Other")
                p->kinematics_integrator = 1;
            else
                bad = true;
        }
        max_steps = &p->max_episode_steps;
        break;
    }
    case GYMRS_MOUNTAIN_CAR: {
        auto* p = static_cast<gymrs_mountain_car_params*>(params);
        number("min_position", p->min_position);
        number("max_position", p->max_position);
        number("max_speed", p->max_speed);
        number("goal_position", p->goal_position);
        number("goal_velocity", p->goal_velocity);
        number("force", p->force);
        number("gravity", p->gravity);
        max_steps = &p->max_episode_steps;
        break;
    }
    case GYMRS_PENDULUM: {
        auto* p = static_cast<gymrs_pendulum_params*>(params);
        number("max_speed", p->max_speed);
        number("max_torque", p->max_torque);
        number("dt", p->dt);
        number("g", p->g);
        number("m", p->m);
        number("l", p->l);
        max_steps = &p->max_episode_steps;
        break;
    }
    }
    if (const json::Value* g = root.get("gymrs")) {
        const json::Value* v = g->kind == json::Value::ObjectK ? g->get("max_episode_steps") : nullptr;
        if (v && v->kind == json::Value::Number && v->number >= 0 && v->number <= 4294967295.0) *max_steps = (uint32_t)v->number;
    }
    int dim = 0;
    if (const json::Value* st = root.get("state")) {
        if (st->kind != json::Value::ObjectK) bad = true;
        if (!bad && state)
            for (const auto& f : st->fields) {
                if (dim >= 4 || (f.second.kind != json::Value::Number && f.second.kind != json::Value::Null)) {
                    bad = true;
                    break;
                }
                state[dim++] = f.second.kind == json::Value::Number ? f.second.number : std::numeric_limits<double>::quiet_NaN();
            }
    }
    if (state_dim) *state_dim = dim;
    if (bad) return fail(GYMRS_EINVAL, "gymrs_params_from_json: a known field has the wrong JSON type");
    return GYMRS_OK;
}

// Test / developer hooks (NOT in the header; tests and A/B tools bind it by name): bit 0 the next chains find a poisoned XCD table (stands in for a
// workgroup deal that changed under the engine), bit 1 no XCD check (what it costs), bit 2 CartPole chains with 256 work-items per workgroup, bit 3 HIP
// launches of gymrs_step_many start the CHAIN'S binary (the embedded code object) through hipModuleLaunchKernel, bit 4 gymrs_allreduce_stats_multi takes
// its grouped RCCL branch even for one shard, bit 5 that branch treats RCCL as unavailable (the host-sum fall-back of a failed first contact).  Until round 4 these were environment variables read on every gymrs_step_many.
gymrs_status gymrs_dev_set_hooks(gymrs_engine* e, uint32_t bits)
{
    if (!e) return fail(GYMRS_EINVAL, "gymrs_dev_set_hooks: engine is NULL");
    e->dev_hooks = bits;
    return GYMRS_OK;
}

// Developer hook (NOT in the header; tools/coldstart_amp.py binds it by name): the RAW bookkeeping arrays as device memory holds them, without the
// fold a statistics call or a snapshot would run first -- the post-mortem of a wrong episode count wants to see WHICH per-wavefront slots, which lanes'
// start ticks and which rows of the reset log differ from a clean engine's.  what: 0 the per-wavefront slots {episodes, return bits} (16 B each), 1 ep_start
// (4 B per lane), 2 the reset log (8 rows of log_row_words u64), 3 {n_stat_blocks, log_row_words, log_pending, log_first_tick lo / hi, tick lo / hi, epoch} as
// u32.  Waits for the engine's stream (plainly), then copies at most `bytes`; *copied = the array's size.
gymrs_status gymrs_dev_peek(gymrs_engine* e, int what, void* host_out, uint64_t bytes, uint64_t* copied)
{
    if (!e || !host_out) return fail(GYMRS_EINVAL, "gymrs_dev_peek: NULL argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const void* src = nullptr;
    uint64_t size = 0;
    uint32_t info[8] = {e->n_stat_blocks, e->log_row_words, e->log_pending, (uint32_t)e->log_first_tick, (uint32_t)(e->log_first_tick >> 32),
                        (uint32_t)e->tick, (uint32_t)(e->tick >> 32), e->epoch};
    switch (what) {
    case 0: src = e->block_stats; size = (uint64_t)e->n_stat_blocks * 16; break;
    case 1: src = e->ep_start; size = e->n * 4; break;
    case 2: src = e->reset_log; size = e->reset_log ? (uint64_t)kResetLogRows * e->log_row_words * 8 : 0; break;
    case 3:
        if (copied) *copied = sizeof(info);
        std::memcpy(host_out, info, bytes < sizeof(info) ? bytes : sizeof(info));
        return GYMRS_OK;
    default: return fail(GYMRS_EINVAL, "gymrs_dev_peek: unknown array");
    }
    if (copied) *copied = size;
    const uint64_t take = bytes < size ? bytes : size;
    if (take && src) {
        if (e->pool_host && what == 1)
            std::memcpy(host_out, host_of(e, e->ep_start), take);
        else
            HIP_TRY(hipMemcpy(host_out, src, take, hipMemcpyDeviceToHost));
    }
    return GYMRS_OK;
}

#ifdef GYMRS_TRACE_TIMES
// developer hook, not part of the ABI header: device buffer of 8 u64 stamps per wavefront
gymrs_status gymrs_dev_set_trace(gymrs_engine* e, unsigned long long* buf)
{
    e->trace = buf;
    return GYMRS_OK;
}
#endif

gymrs_status gymrs_get_tick(gymrs_engine* e, uint64_t* tick, uint64_t* seed)
{
    if (!e) return fail(GYMRS_EINVAL, "gymrs_get_tick: engine is NULL");
    if (tick) *tick = e->tick;
    if (seed) *seed = e->seed;
    return GYMRS_OK;
}

} // extern "C"
