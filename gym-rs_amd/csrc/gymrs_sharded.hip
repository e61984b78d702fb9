// gymrs_sharded.hip -- the in-process multi-GPU sharder of the C ABI (include/gymrs_amd.h, "one batch over several GPUs").
//
// lanes are fully independent (no cross-lane term anywhere in cartpole.rs:398-483 / mountain_car.rs:398-435), so a batch of
// n_total lanes is cut into contiguous blocks, one engine per block, lane i of block r carrying the global id offset + first[r] + i into the
// Philox counters -- results are identical for any number of blocks.  SURVEY 7.1 step 8: "one host thread + stream per device".  That is what
// this file is: a gymrs_sharded owns k engines and k worker threads; worker r is the ONE thread that ever drives engine r (the reference's
// `&mut self`, core.rs:42-50), bound to that engine's device for its whole life, so k launches are enqueued concurrently and no thread pays
// hipSetDevice or another device's launch cost.  The only exchange of the path, the four statistics doubles, goes through
// gymrs_allreduce_stats_multi (RCCL over xGMI on distinct devices, a host-side sum where shards share a device).
//
// Hand-over of a command: the caller writes the worker's mailbox and bumps its sequence number; a worker that has been idle for less than
// ~100 us is still spinning on that number and starts within a fraction of a microsecond (a per-step loop over a 6 us kernel cannot afford a
// futex wake-up per launch), one that has been idle longer sleeps on a condition variable.  Every call of this file returns when all
// workers have finished ENQUEUEING (gymrs_step* stay asynchronous on each engine's stream); gymrs_sharded_sync waits for the devices.
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "gymrs_engine_priv.h"

extern "C" gymrs_status gymrs_allreduce_stats_multi(gymrs_engine** shards, int n, double out[4], int* used_rccl);

namespace {

using Job = std::function<gymrs_status(gymrs_engine*&)>;

struct Worker {
    int index = 0, device = 0;
    uint64_t first = 0, count = 0; // this block's lanes: [first, first + count) of the batch
    gymrs_engine* eng = nullptr;
    std::thread thread;
    // mailbox
    Job job;
    std::atomic<uint64_t> posted{0}, finished{0};
    std::atomic<bool> sleeping{false}, quit{false};
    std::mutex mu;
    std::condition_variable cv;
    gymrs_status status = GYMRS_OK;
    std::string error;

    void run()
    {
        (void)hipSetDevice(device); // the worker's device for its whole life
        uint64_t seen = 0;
        for (;;) {
            // a command that follows the last one closely is picked up spinning; otherwise sleep
            bool got = false;
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t spin = 0;; ++spin) {
                if (posted.load(std::memory_order_acquire) != seen || quit.load(std::memory_order_acquire)) {
                    got = true;
                    break;
                }
                __builtin_ia32_pause();
                if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(100)) break;
            }
            if (!got) {
                std::unique_lock<std::mutex> lock(mu);
                sleeping.store(true, std::memory_order_seq_cst);
                cv.wait(lock, [&] { return posted.load(std::memory_order_seq_cst) != seen || quit.load(std::memory_order_seq_cst); });
                sleeping.store(false, std::memory_order_seq_cst);
            }
            if (posted.load(std::memory_order_acquire) == seen) return; // quit with nothing posted
            seen = posted.load(std::memory_order_acquire);
            status = job(eng);
            if (status != GYMRS_OK) error = gymrs_last_error(); // (per thread: carried to the caller's thread by wait())
            finished.store(seen, std::memory_order_release);
        }
    }

    void post(Job j)
    {
        job = std::move(j);
        posted.fetch_add(1, std::memory_order_seq_cst);
        if (sleeping.load(std::memory_order_seq_cst)) {
            std::lock_guard<std::mutex> lock(mu);
            cv.notify_one();
        }
    }

    gymrs_status wait()
    {
        const uint64_t want = posted.load(std::memory_order_relaxed);
        for (uint32_t spin = 0; finished.load(std::memory_order_acquire) != want; ++spin) {
            if (spin < 20000u)
                __builtin_ia32_pause();
            else
                std::this_thread::yield(); // a long call (creation, reset with its copies, a synchronise)
        }
        return status;
    }

    void stop()
    {
        quit.store(true, std::memory_order_seq_cst);
        {
            std::lock_guard<std::mutex> lock(mu);
            cv.notify_one();
        }
        if (thread.joinable()) thread.join();
    }
};

} // namespace

struct gymrs_sharded {
    gymrs_env_kind kind;
    uint64_t n_total = 0, gid0 = 0;
    uint32_t flags = 0;
    std::vector<Worker*> w;
    std::vector<gymrs_engine*> engines;
    int last_used_rccl = -1; // -1: no statistics call yet
    std::string reduce_path;

    // every worker runs its job concurrently; the first failure (lowest shard) is reported on the CALLER's thread
    gymrs_status all(const std::function<Job(int)>& make)
    {
        for (size_t r = 0; r < w.size(); ++r) w[r]->post(make((int)r));
        gymrs_status st = GYMRS_OK;
        std::string msg;
        for (size_t r = 0; r < w.size(); ++r) {
            const gymrs_status s = w[r]->wait();
            if (s != GYMRS_OK && st == GYMRS_OK) {
                st = s;
                msg = "shard " + std::to_string(r) + " (device " + std::to_string(w[r]->device) + "): " + w[r]->error;
            }
        }
        return st == GYMRS_OK ? GYMRS_OK : fail(st, msg);
    }
};

extern "C" {

gymrs_status gymrs_sharded_destroy(gymrs_sharded* h)
{
    if (!h) return GYMRS_OK;
    for (Worker* wk : h->w) { // each engine is destroyed by the thread that drove it, then the thread ends
        wk->post([](gymrs_engine*& e) {
            gymrs_status st = gymrs_engine_destroy(e);
            e = nullptr;
            return st;
        });
        (void)wk->wait();
        wk->stop();
        delete wk;
    }
    delete h;
    return GYMRS_OK;
}

gymrs_status gymrs_sharded_create(gymrs_env_kind kind, uint64_t n_total, uint64_t global_env_offset, int n_shards, const int* devices,
                                  const void* params, uint32_t flags, gymrs_sharded** out)
{
    if (!out) return fail(GYMRS_EINVAL, "gymrs_sharded_create: out is NULL");
    *out = nullptr;
    if (n_shards < 1 || n_shards > 64) return fail(GYMRS_EINVAL, "gymrs_sharded_create: n_shards must be in 1..64");
    if (n_total < (uint64_t)n_shards) return fail(GYMRS_EINVAL, "gymrs_sharded_create: fewer lanes than shards");
    gymrs_sharded* h = new (std::nothrow) gymrs_sharded();
    if (!h) return fail(GYMRS_ENOMEM, "gymrs_sharded_create: host allocation failed");
    h->kind = kind;
    h->n_total = n_total;
    h->gid0 = global_env_offset;
    h->flags = flags;
    // Contiguous blocks; every block but the last a whole number of 1024-lane tiles (n_total / n_shards rounded DOWN to tiles), the last takes
    // the rest: every block's arrays start where a tile of the unsharded batch starts, and the ragged tail stays in the last block.  (Nothing
    // depends on it -- lanes are independent.)
    const uint64_t tile = 1024;
    uint64_t per = n_total / (uint64_t)n_shards;
    if (per >= tile) per = per / tile * tile;
    uint64_t first = 0;
    for (int r = 0; r < n_shards; ++r) {
        const uint64_t count = r == n_shards - 1 ? n_total - first : per;
        Worker* wk = new Worker();
        wk->index = r;
        wk->device = devices ? devices[r] : r;
        wk->first = first;
        wk->count = count;
        first += count;
        h->w.push_back(wk);
        wk->thread = std::thread([wk] { wk->run(); });
    }
    // Engines are created ONE AT A TIME, each by its own worker: creation sets up an HSA queue, runs the dispatcher's self-check and times a
    // hand-over, all of which want the device (and, for shards sharing one, the per-device registry) to themselves.
    for (Worker* wk : h->w) {
        const uint64_t off = global_env_offset + wk->first, cnt = wk->count;
        const int dev = wk->device;
        wk->post([=](gymrs_engine*& e) { return gymrs_engine_create(kind, cnt, off, dev, params, flags, &e); });
        if (gymrs_status st = wk->wait()) {
            const std::string msg = "gymrs_sharded_create: shard " + std::to_string(wk->index) + " (device " + std::to_string(dev) + "): " + wk->error;
            gymrs_sharded_destroy(h);
            return fail(st, msg);
        }
    }
    for (Worker* wk : h->w) h->engines.push_back(wk->eng);
    // Like ::new (cartpole.rs:92,120) creation seeds from OS entropy -- ONE seed for the whole batch, or the blocks would not be one batch
    std::random_device rd;
    const uint64_t seed = ((uint64_t)rd() << 32) ^ (uint64_t)rd();
    if (gymrs_status st = h->all([=](int) { return [=](gymrs_engine*& e) { return gymrs_reset(e, 1, seed, nullptr, nullptr); }; })) {
        const std::string msg = gymrs_last_error();
        gymrs_sharded_destroy(h);
        return fail(st, msg);
    }
    *out = h;
    return GYMRS_OK;
}

gymrs_status gymrs_sharded_count(gymrs_sharded* h, int* n_shards)
{
    if (!h || !n_shards) return fail(GYMRS_EINVAL, "gymrs_sharded_count: NULL argument");
    *n_shards = (int)h->w.size();
    return GYMRS_OK;
}

gymrs_status gymrs_sharded_shard(gymrs_sharded* h, int shard, gymrs_engine** engine, uint64_t* first_lane, uint64_t* n_lanes, int* device)
{
    if (!h) return fail(GYMRS_EINVAL, "gymrs_sharded_shard: handle is NULL");
    if (shard < 0 || shard >= (int)h->w.size()) return fail(GYMRS_EINVAL, "gymrs_sharded_shard: shard index out of range");
    const Worker* wk = h->w[(size_t)shard];
    if (engine) *engine = wk->eng;
    if (first_lane) *first_lane = wk->first;
    if (n_lanes) *n_lanes = wk->count;
    if (device) *device = wk->device;
    return GYMRS_OK;
}

gymrs_status gymrs_sharded_reset(gymrs_sharded* h, int has_seed, uint64_t seed, const float* bounds_low_high, uint64_t* seed_used)
{
    if (!h) return fail(GYMRS_EINVAL, "gymrs_sharded_reset: handle is NULL");
    if (!has_seed) { // seeding.rs:22: a fresh seed from the OS -- drawn once, used by every block
        std::random_device rd;
        seed = ((uint64_t)rd() << 32) ^ (uint64_t)rd();
    }
    if (seed_used) *seed_used = seed;
    return h->all([=](int) { return [=](gymrs_engine*& e) { return gymrs_reset(e, 1, seed, bounds_low_high, nullptr); }; });
}

gymrs_status gymrs_sharded_step(gymrs_sharded* h, const void* const* actions_dev)
{
    if (!h || !actions_dev) return fail(GYMRS_EINVAL, "gymrs_sharded_step: NULL argument");
    return h->all([=](int r) {
        const void* a = actions_dev[r];
        return [=](gymrs_engine*& e) { return gymrs_step(e, a); };
    });
}

gymrs_status gymrs_sharded_step_many(gymrs_sharded* h, const void* const* actions_dev, uint64_t stride_bytes, uint32_t n_buffers, uint32_t n_steps,
                                     int use_graph)
{
    if (!h || !actions_dev) return fail(GYMRS_EINVAL, "gymrs_sharded_step_many: NULL argument");
    return h->all([=](int r) {
        const void* a = actions_dev[r];
        return [=](gymrs_engine*& e) { return gymrs_step_many(e, a, stride_bytes, n_buffers, n_steps, use_graph); };
    });
}

gymrs_status gymrs_sharded_fill_actions(gymrs_sharded* h, void* const* actions_dev, uint64_t seed, uint64_t t)
{
    if (!h || !actions_dev) return fail(GYMRS_EINVAL, "gymrs_sharded_fill_actions: NULL argument");
    return h->all([=](int r) {
        void* a = actions_dev[r];
        return [=](gymrs_engine*& e) { return gymrs_fill_actions(e, a, seed, t); };
    });
}

// The fused random-policy rollout (gymrs_rollout) on every block: the action stream is keyed by GLOBAL lane ids, so the batch does what one engine would.
gymrs_status gymrs_sharded_rollout(gymrs_sharded* h, uint32_t n_steps, uint64_t action_seed, uint64_t action_t0)
{
    if (!h) return fail(GYMRS_EINVAL, "gymrs_sharded_rollout: handle is NULL");
    return h->all([=](int) { return [=](gymrs_engine*& e) { return gymrs_rollout(e, n_steps, action_seed, action_t0); }; });
}

// `env.gravity = ...` for every lane of the batch (gymrs_set_params on every block).
gymrs_status gymrs_sharded_set_params(gymrs_sharded* h, const void* params)
{
    if (!h || !params) return fail(GYMRS_EINVAL, "gymrs_sharded_set_params: NULL argument");
    return h->all([=](int) { return [=](gymrs_engine*& e) { return gymrs_set_params(e, params); }; });
}

gymrs_status gymrs_sharded_sync(gymrs_sharded* h)
{
    if (!h) return fail(GYMRS_EINVAL, "gymrs_sharded_sync: handle is NULL");
    return h->all([](int) { return [](gymrs_engine*& e) { return gymrs_sync(e); }; });
}

gymrs_status gymrs_sharded_stats_clear(gymrs_sharded* h)
{
    if (!h) return fail(GYMRS_EINVAL, "gymrs_sharded_stats_clear: handle is NULL");
    return h->all([](int) { return [](gymrs_engine*& e) { return gymrs_stats_clear(e); }; });
}

gymrs_status gymrs_sharded_stats(gymrs_sharded* h, double out[4])
{
    if (!h || !out) return fail(GYMRS_EINVAL, "gymrs_sharded_stats: NULL argument");
    // The workers are idle (every call of this file returns after its jobs): the calling thread may drive all engines for this one call --
    // RCCL's grouped form wants ONE thread to issue every rank's all-reduce.
    int used = 0;
    if (gymrs_status st = gymrs_allreduce_stats_multi(h->engines.data(), (int)h->engines.size(), out, &used)) return st;
    h->last_used_rccl = used;
    return GYMRS_OK;
}

gymrs_status gymrs_sharded_get_state(gymrs_sharded* h, uint64_t first, uint64_t count, float* host_out)
{
    if (!h || !host_out) return fail(GYMRS_EINVAL, "gymrs_sharded_get_state: NULL argument");
    if (first > h->n_total || count > h->n_total - first) return fail(GYMRS_EINVAL, "gymrs_sharded_get_state: lane range out of bounds");
    // SoA over the REQUESTED lanes: dim arrays of `count` floats back to back, like gymrs_get_state
    const int dim = h->kind == GYMRS_CARTPOLE ? 4 : 2;
    return h->all([=](int r) {
        const Worker* wk = h->w[(size_t)r];
        const uint64_t lo = first > wk->first ? first : wk->first;
        const uint64_t hi = (first + count) < (wk->first + wk->count) ? (first + count) : (wk->first + wk->count);
        return [=](gymrs_engine*& e) -> gymrs_status {
            if (lo >= hi) return GYMRS_OK;
            std::vector<float> tmp((size_t)(hi - lo) * (size_t)dim);
            if (gymrs_status st = gymrs_get_state(e, lo - wk->first, hi - lo, tmp.data())) return st;
            for (int j = 0; j < dim; ++j)
                std::memcpy(host_out + (size_t)j * count + (lo - first), tmp.data() + (size_t)j * (hi - lo), (size_t)(hi - lo) * sizeof(float));
            return GYMRS_OK;
        };
    });
}

gymrs_status gymrs_sharded_get_step_result(gymrs_sharded* h, uint64_t first, uint64_t count, float* reward, uint8_t* done, uint8_t* truncated)
{
    if (!h) return fail(GYMRS_EINVAL, "gymrs_sharded_get_step_result: handle is NULL");
    if (first > h->n_total || count > h->n_total - first) return fail(GYMRS_EINVAL, "gymrs_sharded_get_step_result: lane range out of bounds");
    return h->all([=](int r) {
        const Worker* wk = h->w[(size_t)r];
        const uint64_t lo = first > wk->first ? first : wk->first;
        const uint64_t hi = (first + count) < (wk->first + wk->count) ? (first + count) : (wk->first + wk->count);
        return [=](gymrs_engine*& e) -> gymrs_status {
            if (lo >= hi) return GYMRS_OK;
            const uint64_t o = lo - first;
            return gymrs_get_step_result(e, lo - wk->first, hi - lo, reward ? reward + o : nullptr, done ? done + o : nullptr,
                                         truncated ? truncated + o : nullptr);
        };
    });
}

// How the last gymrs_sharded_stats summed: "rccl" (grouped all-reduce over the shards' communicators), "host" (shards share a device, or one
// shard), or "none" before the first call.
const char* gymrs_sharded_reduce_path(gymrs_sharded* h)
{
    if (!h || h->last_used_rccl < 0) return "none";
    return h->last_used_rccl ? "rccl" : "host";
}

} // extern "C"
