// gymrs_aql.hip -- the engine's own AQL dispatcher (host code only; see gymrs_aql.h for the why and the protocol).
// HSA next to HIP: the HIP runtime sits on the same ROCr instance, hsa_init() only takes a reference; device memory that
// hipMalloc returned is ordinary agent memory to a kernel dispatched through an HSA queue of the same agent.
#include "gymrs_aql.h"
#include "gymrs_kernels.h"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#if !defined(__HIP_DEVICE_COMPILE__)
#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#define GYMRS_STORE_FENCE() _mm_sfence()
#else
#define GYMRS_STORE_FENCE() __atomic_thread_fence(__ATOMIC_SEQ_CST) // (a full barrier also drains the write-combining buffers)
#endif
// The stand-alone code object of gymrs_step_aql.hip, built by build.py right before this file (into the object directory, which is on
// the assembler's include path) and embedded here.
__asm__(".section .rodata\n"
        ".balign 4096\n"
        ".global gymrs_aql_blob_begin\n"
        "gymrs_aql_blob_begin:\n"
        ".incbin \"gymrs_aql_kernels.hsaco\"\n"
        ".global gymrs_aql_blob_end\n"
        "gymrs_aql_blob_end:\n"
        ".previous\n");
extern "C" const char gymrs_aql_blob_begin[];
extern "C" const char gymrs_aql_blob_end[];
#endif

namespace gymrs {
namespace {

// The two HIP-launched ends of the hand-over (on the engine's stream).  Kernels, not stream memory operations: a
// hipStreamWaitValue32 leaves the stream's hardware queue POLLING for as long as the chain runs, and a polling queue next to
// the chain's queue cost every launch of the chain 0.25 us (profiles/r03_aql_engine_ab.log); one sleeping wavefront costs nothing.
__global__ __launch_bounds__(64) void aql_hip_set_flag(uint32_t* flag, uint32_t seq)
{
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Waits until the chain's last packet has stored `seq` (wrap-around safe).  Bounded by the 100 MHz real-time counter, not by a
// poll count: kStreamWaitSeconds, far beyond any chain (aql_begin cuts chains off long before); a wait that runs out reports it.
constexpr unsigned long long kStreamWaitTicks = 120ull * 100000000ull;
__global__ __launch_bounds__(64) void aql_hip_wait_flag(const uint32_t* flag, uint32_t seq, uint32_t* err)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const uint32_t v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(v - seq) >= 0) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > kStreamWaitTicks) {
            __hip_atomic_store(err, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(64);
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

constexpr uint32_t kQueuePackets = 4096;
constexpr uint32_t kKernargSlots = 8 * kQueuePackets; // a slot is rewritten only after kQueuePackets later packets were CONSUMED
constexpr uint32_t kFlushEvery = 64;
constexpr size_t kHostErrWords = 16; // the mapped host block a chain reports through (64 bytes; [0] is the word in use)
constexpr uint32_t kCalibItems = 1u << 18; // 16-byte items of aql_calibrate's probe chains (4 MB)

const char* hsa_err(hsa_status_t s)
{
    const char* m = nullptr;
    hsa_status_string(s, &m);
    return m ? m : "HSA error";
}

#define HSA_OK(expr, what)                                                     \
    do {                                                                       \
        const hsa_status_t st_ = (expr);                                       \
        if (st_ != HSA_STATUS_SUCCESS) {                                       \
            *why = std::string(what) + ": " + hsa_err(st_);                    \
            return false;                                                      \
        }                                                                      \
    } while (0)
#define HIP_OK(expr, what)                                                     \
    do {                                                                       \
        const hipError_t e_ = (expr);                                          \
        if (e_ != hipSuccess) {                                                \
            *why = std::string(what) + ": " + hipGetErrorString(e_);          \
            return false;                                                      \
        }                                                                      \
    } while (0)

struct DeviceCtx {
    bool tried = false, ok = false;
    std::string why;
    hsa_agent_t gpu{}, cpu{};
    hsa_amd_memory_pool_t gpu_pool{};
    hsa_executable_t exe{};
    std::map<std::string, AqlKernel> kernels;
    int concurrent_handover = -1; // -1 not tried yet, 1 works, 0 kernels of two queues do not run side by side here (see aql_create)
    // bumped whenever a chain object of this device is CREATED: which hand-over suits an engine depends on where its queues sit
    // among the process's queues, so an engine looks again (aql_calibrate, ~3 ms on its own stream) when a queue has appeared since
    // it last looked.  A queue that goes away moves nobody (destroying an engine used to re-calibrate every other one).
    std::atomic<uint32_t> epoch{0};
    float* calib_buf = nullptr; // scratch of aql_calibrate's probe chains (allocated once, at the self-check: no hipMalloc later)
    std::atomic<int> live{0}; // chain objects (= HSA queues) alive on this device, in use or parked
    // Chain objects whose engine is gone, parked for the next engine (guarded by g_mu): an HSA queue is created once and reused, not
    // destroyed and re-created with every engine -- engines come and go by the hundred in a test suite or a hyper-parameter sweep,
    // hardware queues should not (round 4: the dispatcher is set up at engine creation, which made every engine a queue).
    std::vector<AqlChain*> parked;
};

// Every chain object is a hardware queue of its own next to the HIP runtime's; a process that creates engines by the dozen must
// not oversubscribe the device's queue slots with them.  Engines beyond this many keep to HIP launches (they say why).
constexpr int kMaxChainsPerDevice = 8;

constexpr int kMaxDevices = 64;
DeviceCtx g_ctx[kMaxDevices];
std::mutex g_mu;

struct FindAgents {
    uint32_t want_bdf = 0, want_domain = 0;
    hsa_agent_t gpu{}, cpu{};
};

hsa_status_t on_agent(hsa_agent_t a, void* data)
{
    auto* f = static_cast<FindAgents*>(data);
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_CPU && f->cpu.handle == 0) f->cpu = a;
    if (t == HSA_DEVICE_TYPE_GPU) {
        uint32_t bdf = 0, domain = 0;
        hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
        hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
        if (bdf == f->want_bdf && domain == f->want_domain) f->gpu = a;
    }
    return HSA_STATUS_SUCCESS;
}

hsa_status_t on_gpu_pool(hsa_amd_memory_pool_t p, void* data)
{
    auto* out = static_cast<hsa_amd_memory_pool_t*>(data);
    hsa_amd_segment_t seg;
    uint32_t flags = 0;
    bool alloc = false;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && out->handle == 0) *out = p;
    return HSA_STATUS_SUCCESS;
}

uint16_t packet_header(int acquire, int release)
{
    return (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                      (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
}

void queue_error(hsa_status_t status, hsa_queue_t*, void* data)
{
    if (data) static_cast<std::atomic<int>*>(data)->store((int)status ? (int)status : -1);
    std::fprintf(stderr, "gymrs: AQL queue error: %s\n", hsa_err(status));
}

} // namespace

class AqlChain {
public:
    DeviceCtx* ctx = nullptr;
    int device = 0;
    hsa_queue_t* q = nullptr;
    char* kernarg = nullptr;     // device memory, written by the CPU through the PCIe BAR
    uint32_t* in_flag = nullptr; // device word a one-wave kernel on the HIP stream writes, the chain's first packet waits for
    uint32_t* out_flag = nullptr; // device word the chain's last packet writes, a one-wave kernel on the HIP stream waits for
    uint32_t* host_err = nullptr; // mapped host word: a wait that gave up
    uint32_t* host_err_dev = nullptr;
    uint32_t seq = 0;
    bool in_chain = false;
    // Synchronous hand-over: the host waits for the stream before a chain and for the chain after it.  Used where the two
    // one-wave kernels of the asynchronous hand-over cannot be in flight together -- under rocprofv3 --pmc, whose counter
    // collection serialises kernels across queues (the asynchronous form would wait out its bound there).
    bool sync_mode = false;
    // Decided per stream by aql_calibrate(): a kernel in flight on the stream's hardware queue next to the chain's queue -- the
    // asynchronous hand-over needs one -- is free on some placements of the two queues and costs the chain a factor of 3-6 on
    // others (the second and third engine of a process: profiles/r03_two_engines.log); a hipStreamWaitValue32 instead of the
    // kernel is no better (worse).  Where it costs, the chains of this engine use the synchronous hand-over.
    hipStream_t calibrated_for = nullptr;
    bool calibrated = false;
    uint32_t calibrated_epoch = 0;
    bool forced_sync = false; // device-wide: the asynchronous hand-over does not work here at all (see aql_create)
    hsa_signal_t done{};
    // 8 entries (one cache line each) the first step launch of every chain records {chain number, XCC} of its workgroups 0 .. 7 in and the
    // later launches of that chain compare against (StepArgs::xcc_table)
    uint32_t* xcc_table = nullptr;
    unsigned long long wait_ticks = 10ull * 100000000ull; // bound of the chain's first packet (100 MHz ticks)
    std::atomic<int> queue_status{0};
    AqlKernel k_wait, k_set;
    // packets written but not yet published (headers still INVALID)
    struct Staged {
        hsa_kernel_dispatch_packet_t* p;
        uint32_t header_and_setup;
        uint64_t idx;
    };
    std::vector<Staged> staged;
    void* last_kernarg = nullptr;

    bool stage(const AqlKernel& k, uint32_t grid_workitems, uint32_t wg, const void* args, size_t bytes, int acquire, int release, hsa_signal_t completion,
               std::string* why)
    {
        if (queue_status.load() != 0) {
            *why = "the AQL queue reported an error earlier";
            return false;
        }
        if (bytes > kAqlKernargSlot) {
            *why = "kernel arguments larger than a ring slot";
            return false;
        }
        const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
        if (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {
            publish(); // whatever is staged has to become visible before the queue can drain
            const auto t0 = std::chrono::steady_clock::now();
            while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
                    *why = "the AQL queue did not drain within 20 s";
                    return false;
                }
            }
        }
        char* slot = kernarg + (size_t)(idx % kKernargSlots) * kAqlKernargSlot;
        std::memcpy(slot, args, bytes);
        last_kernarg = slot;
        auto* p = &static_cast<hsa_kernel_dispatch_packet_t*>(q->base_address)[idx & (q->size - 1)];
        const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        // the slot may still carry the header of the packet that used it a lap ago: no valid header over a half-written body
        __atomic_store_n(reinterpret_cast<uint32_t*>(p), (uint32_t)(HSA_PACKET_TYPE_INVALID << HSA_PACKET_HEADER_TYPE), __ATOMIC_RELAXED);
        p->workgroup_size_x = (uint16_t)wg;
        p->workgroup_size_y = p->workgroup_size_z = 1;
        p->reserved0 = 0;
        p->grid_size_x = grid_workitems;
        p->grid_size_y = p->grid_size_z = 1;
        p->private_segment_size = k.private_bytes;
        p->group_segment_size = k.group_bytes;
        p->kernel_object = k.object;
        p->kernarg_address = slot;
        p->reserved2 = 0;
        p->completion_signal = completion;
        staged.push_back({p, (uint32_t)packet_header(acquire, release) | ((uint32_t)setup << 16), idx});
        // (developer knob for A/B runs: packets per publish)
        static const size_t flush_every = [] { const char* v = std::getenv("GYMRS_AQL_FLUSH"); return v ? (size_t)std::strtoul(v, nullptr, 0) : (size_t)kFlushEvery; }();
        if (staged.size() >= flush_every) publish();
        return true;
    }

    // Make the staged packets visible to the packet processor: first the kernel arguments (write-combined stores through the
    // BAR: drain them and read one back, so that they HAVE landed in device memory), then the headers, then the doorbell.
    void publish()
    {
        if (staged.empty()) return;
#if !defined(__HIP_DEVICE_COMPILE__)
        GYMRS_STORE_FENCE();
#endif
        if (last_kernarg) (void)*static_cast<volatile uint32_t*>(last_kernarg);
        // One doorbell per packet (posted writes, cheap): rocprofiler's queue interception faults on a doorbell that publishes
        // several packets at once (rocprofv3 --kernel-trace segfaulted inside its doorbell handler; with one packet per ring
        // the same run is traced, profiles/r03_kernel_trace_*).
        for (const Staged& s : staged) {
            __atomic_store_n(reinterpret_cast<uint32_t*>(s.p), s.header_and_setup, __ATOMIC_RELEASE);
            hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)s.idx);
        }
        staged.clear();
    }
};

namespace {

// One kernel of the loaded code object by name -> c->kernels (the caller holds g_mu or is the only user of c).
bool lookup_kernel(DeviceCtx* c, const std::string& name, std::string* why)
{
    const char* nm = name.c_str();
    hsa_executable_symbol_t sym;
    HSA_OK(hsa_executable_get_symbol_by_name(c->exe, (name + ".kd").c_str(), &c->gpu, &sym), nm);
    AqlKernel k;
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object), nm);
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_bytes), nm);
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group_bytes), nm);
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.private_bytes), nm);
    c->kernels[name] = k;
    return true;
}

bool load_code(DeviceCtx* c, std::string* why)
{
#if !defined(__HIP_DEVICE_COMPILE__)
    const size_t bytes = (size_t)(gymrs_aql_blob_end - gymrs_aql_blob_begin);
    hsa_code_object_reader_t reader;
    HSA_OK(hsa_code_object_reader_create_from_memory(gymrs_aql_blob_begin, bytes, &reader), "code object reader");
    HSA_OK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &c->exe), "hsa_executable_create_alt");
    HSA_OK(hsa_executable_load_agent_code_object(c->exe, c->gpu, reader, nullptr, nullptr), "loading the AQL code object");
    HSA_OK(hsa_executable_freeze(c->exe, nullptr), "hsa_executable_freeze");
    std::vector<std::string> names = {"gymrs_aql_wait_flag", "gymrs_aql_set_flag", "gymrs_aql_selfcheck"};
#ifndef GYMRS_AQL_ENDS_ONLY // (a tool's build of the dispatcher against its own code object: its kernels are looked up when asked for, aql_kernel)
    for (const char* env_threads : {"cartpole_f%d_t512", "cartpole_f%d_t256", "mountain_car_f%d_t256", "pendulum_f%d_t256"}) // (gymrs_step_aql.hip)
        for (int flags : {0, 1, 3, 4, 5, 7})
            for (const char* hint : {"_nt", "_o", "_so", "_pl"}) {
                char stem[64];
                std::snprintf(stem, sizeof(stem), env_threads, flags);
                names.push_back(std::string("gymrs_aql_") + stem + hint);
            }
#endif
    for (const std::string& name : names)
        if (!lookup_kernel(c, name, why)) return false;
    return true;
#else
    (void)c;
    (void)why;
    return false;
#endif
}

// `status` = where the queue's error callback reports (it must outlive the queue AND any late callback of the runtime's event thread):
// the chain object's own word for the long-lived objects, a static word for the self-check's short-lived queue.
bool make_queue(DeviceCtx* c, AqlChain* ch, std::string* why, std::atomic<int>* status = nullptr)
{
    static const uint32_t queue_packets = [] { const char* v = std::getenv("GYMRS_AQL_QUEUE"); return v ? (uint32_t)std::strtoul(v, nullptr, 0) : kQueuePackets; }();
    HSA_OK(hsa_queue_create(c->gpu, queue_packets, HSA_QUEUE_TYPE_SINGLE, queue_error, status ? status : &ch->queue_status, UINT32_MAX, UINT32_MAX, &ch->q), "hsa_queue_create");
    void* ka = nullptr;
    HSA_OK(hsa_amd_memory_pool_allocate(c->gpu_pool, (size_t)kKernargSlots * kAqlKernargSlot, 0, &ka), "kernel-argument ring");
    ch->kernarg = static_cast<char*>(ka);
    // the CPU writes the ring through the PCIe BAR (what HIP does with its own kernel arguments on this part)
    HSA_OK(hsa_amd_agents_allow_access(1, &c->cpu, nullptr, ka), "CPU access to device memory (large BAR)");
    return true;
}

struct SelfCheckArgs { // gymrs_aql_selfcheck (gymrs_step_aql.hip)
    float* x;
    uint32_t n4, first;
    uint32_t *xcc, *moved;
};

// The assumption the fence-free chain rests on, checked ON THIS DEVICE before the path is used: launches chained by the barrier
// bit alone (agent acquire, NO release) see each other's stores -- with grids that are not multiples of the XCD count and with
// one-workgroup launches in between (neither may change which XCD a workgroup index lands on) -- judged by the outcome (no lost
// update) and by the hardware's own word: each workgroup index reads HW_REG_XCC_ID in every launch and must stay where it was.
bool self_check(DeviceCtx* c, int device, std::string* why)
{
    AqlChain ch;
    ch.ctx = c;
    ch.device = device;
    bool ok = false;
    float* x = nullptr;
    uint32_t *flag = nullptr, *xcc = nullptr; // xcc[0 .. kGroups): the XCC a workgroup index ran on first, [kGroups .. 2 kGroups): it moved
    hsa_signal_t done{};
    constexpr uint32_t kLaunches = 96, kGroups = 1022; // 1022 workgroups: not a multiple of 8, the last one partial
    do {
        static std::atomic<int> self_check_queue_status{0}; // (`ch` lives on this stack frame; the callback's word must not)
        if (!make_queue(c, &ch, why, &self_check_queue_status)) break;
        const uint32_t n4 = (kGroups - 1u) * 256u + 77u;
        if (hipMalloc(&x, (size_t)n4 * 16) != hipSuccess || hipMalloc(&flag, 64) != hipSuccess || hipMalloc(&xcc, 2 * kGroups * sizeof(uint32_t)) != hipSuccess) {
            *why = "self-check: hipMalloc failed";
            break;
        }
        if (hipMemset(x, 0, (size_t)n4 * 16) != hipSuccess || hipMemset(xcc, 0, 2 * kGroups * sizeof(uint32_t)) != hipSuccess ||
            hipStreamSynchronize(nullptr) != hipSuccess) {
            *why = "self-check: hipMemset failed";
            break;
        }
        if (hsa_signal_create(1, 0, nullptr, &done) != HSA_STATUS_SUCCESS) {
            *why = "self-check: hsa_signal_create failed";
            break;
        }
        SelfCheckArgs args{x, n4, 1u, xcc, xcc + kGroups};
        struct {
            uint32_t* flag;
            uint32_t seq;
        } fargs{flag, 0};
        const AqlKernel& k = c->kernels["gymrs_aql_selfcheck"];
        const AqlKernel& ks = c->kernels["gymrs_aql_set_flag"];
        bool staged_ok = true;
        for (uint32_t t = 0; t < kLaunches && staged_ok; ++t) {
            const bool last = t + 1 == kLaunches;
            args.first = t == 0 ? 1u : 0u;
            staged_ok = ch.stage(k, kGroups * 256u, 256, &args, sizeof(args), HSA_FENCE_SCOPE_AGENT, last ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_NONE,
                                 last ? done : hsa_signal_t{0}, why);
            if (staged_ok && !last && (t % 3) != 2) { // one-workgroup launches in between, as the ends of real chains are
                fargs.seq = t;
                staged_ok = ch.stage(ks, 64, 64, &fargs, sizeof(fargs), HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_NONE, hsa_signal_t{0}, why);
            }
        }
        if (!staged_ok) break;
        ch.publish();
        if (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 10ull * 1000 * 1000 * 1000, HSA_WAIT_STATE_BLOCKED) >= 1) {
            *why = "self-check: the chain did not finish within 10 s";
            break;
        }
        std::vector<float> host((size_t)n4 * 4);
        if (hipMemcpy(host.data(), x, host.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            *why = "self-check: hipMemcpy failed";
            break;
        }
        size_t wrong = 0;
        for (float v : host) wrong += v != (float)kLaunches;
        if (wrong) {
            char buf[160];
            std::snprintf(buf, sizeof(buf), "self-check: %zu of %zu words lost an update in a fence-free chain of %u launches", wrong, host.size(), kLaunches);
            *why = buf;
            break;
        }
        std::vector<uint32_t> where(2 * kGroups);
        if (hipMemcpy(where.data(), xcc, where.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) {
            *why = "self-check: hipMemcpy failed";
            break;
        }
        uint32_t moved = 0, unset = 0;
        for (uint32_t g = 0; g < kGroups; ++g) {
            moved += where[kGroups + g] != 0;
            unset += where[g] == 0;
        }
        if (std::getenv("GYMRS_AQL_VERBOSE")) { // (developer knob: where the workgroup indices ran)
            std::fprintf(stderr, "gymrs: AQL self-check, XCC of workgroup 0..23 in %u chained launches:", kLaunches);
            for (uint32_t g = 0; g < 24; ++g) std::fprintf(stderr, " %u", where[g] - 1u);
            uint32_t per_xcc[16] = {0};
            for (uint32_t g = 0; g < kGroups; ++g) per_xcc[(where[g] - 1u) & 15u] += 1;
            std::fprintf(stderr, "; workgroups per XCC:");
            for (uint32_t k = 0; k < 16; ++k)
                if (per_xcc[k]) std::fprintf(stderr, " %u:%u", k, per_xcc[k]);
            std::fprintf(stderr, "; moved %u\n", moved);
        }
        if (moved || unset) {
            char buf[160];
            std::snprintf(buf, sizeof(buf), "self-check: %u of %u workgroup indices changed their XCD between the launches of a chain (%u never reported one)", moved,
                          kGroups, unset);
            *why = buf;
            break;
        }
        // The production kernels check every launch of a chain against a table of 8 entries (one per residue class of the workgroup index):
        // the deal must repeat with period 8 (round-robin over the XCDs; a partition with fewer XCDs repeats with a divisor of 8: fine too).
        uint32_t aperiodic = 0;
        for (uint32_t g = 0; g < kGroups; ++g) aperiodic += where[g] != where[g & 7u];
        if (aperiodic) {
            char buf[160];
            std::snprintf(buf, sizeof(buf), "self-check: the workgroup -> XCD deal does not repeat with period 8 (%u of %u indices differ from index mod 8)",
                          aperiodic, kGroups);
            *why = buf;
            break;
        }
        // (the table itself is recorded by the first launch of every chain: StepArgs::xcc_table)
        ok = true;
    } while (false);
    if (done.handle) hsa_signal_destroy(done);
    if (ch.q) hsa_queue_destroy(ch.q);
    if (ch.kernarg) hsa_amd_memory_pool_free(ch.kernarg);
    if (x) (void)hipFree(x);
    if (flag) (void)hipFree(flag);
    if (xcc) (void)hipFree(xcc);
    return ok;
}

bool init_device(DeviceCtx* c, int device, std::string* why)
{
    char bus[64] = {0};
    HIP_OK(hipDeviceGetPCIBusId(bus, sizeof(bus), device), "hipDeviceGetPCIBusId");
    unsigned dom = 0, b = 0, d = 0, f = 0;
    if (std::sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) != 4) {
        *why = std::string("unparsable PCI bus id ") + bus;
        return false;
    }
    HSA_OK(hsa_init(), "hsa_init");
    FindAgents find;
    find.want_bdf = (b << 8) | (d << 3) | f;
    find.want_domain = dom;
    HSA_OK(hsa_iterate_agents(on_agent, &find), "hsa_iterate_agents");
    if (find.gpu.handle == 0 || find.cpu.handle == 0) {
        *why = std::string("no HSA agent for HIP device at ") + bus;
        return false;
    }
    c->gpu = find.gpu;
    c->cpu = find.cpu;
    HSA_OK(hsa_amd_agent_iterate_memory_pools(c->gpu, on_gpu_pool, &c->gpu_pool), "memory pools");
    if (c->gpu_pool.handle == 0) {
        *why = "no coarse-grained device memory pool";
        return false;
    }
    if (!load_code(c, why)) return false;
    if (!self_check(c, device, why)) return false;
    void* p = nullptr;
    HIP_OK(hipMalloc(&p, (size_t)kCalibItems * 16), "hipMalloc (calibration scratch)");
    c->calib_buf = static_cast<float*>(p);
    return true;
}

} // namespace

static void aql_discard(AqlChain* c);

AqlChain* aql_create(int hip_device, std::string* why)
{
    std::string local;
    if (!why) why = &local;
    if (hip_device < 0 || hip_device >= kMaxDevices) {
        *why = "device index out of range";
        return nullptr;
    }
    DeviceCtx* c = &g_ctx[hip_device];
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (!c->tried) {
            c->tried = true;
            c->ok = init_device(c, hip_device, &c->why);
        }
        if (!c->ok) {
            *why = c->why;
            return nullptr;
        }
    }
    {   // a parked chain object first: its queue, ring and flags are idle (its engine synchronised its stream before letting go of it)
        std::lock_guard<std::mutex> lock(g_mu);
        while (!c->parked.empty()) {
            AqlChain* ch = c->parked.back();
            c->parked.pop_back();
            if (ch->queue_status.load() != 0) { // (a queue that reported an error is not handed on)
                aql_discard(ch);
                continue;
            }
            ch->in_chain = false;
            ch->staged.clear();
            ch->calibrated = false; // the next engine's stream is another one
            ch->sync_mode = ch->forced_sync;
            if (ch->host_err) std::memset(ch->host_err, 0, kHostErrWords * sizeof(uint32_t)); // every word, not only [0]
            return ch;
        }
    }
    if (c->live.fetch_add(1, std::memory_order_relaxed) >= kMaxChainsPerDevice) {
        c->live.fetch_sub(1, std::memory_order_relaxed);
        *why = "this process already drives " + std::to_string(kMaxChainsPerDevice) + " engines of this device through chains";
        return nullptr;
    }
    AqlChain* ch = new AqlChain();
    ch->ctx = c;
    ch->device = hip_device;
    ch->k_wait = c->kernels["gymrs_aql_wait_flag"];
    ch->k_set = c->kernels["gymrs_aql_set_flag"];
    bool ok = make_queue(c, ch, why);
    if (ok) {
        void* p = nullptr;
        ok = hipMalloc(&p, 64) == hipSuccess && hipMemset(p, 0, 64) == hipSuccess;
        ch->in_flag = static_cast<uint32_t*>(p);
        if (!ok) *why = "hipMalloc (hand-over flag) failed";
    }
    if (ok) {
        void* p = nullptr;
        ok = hipMalloc(&p, 64) == hipSuccess && hipMemset(p, 0, 64) == hipSuccess;
        ch->out_flag = static_cast<uint32_t*>(p);
        if (!ok) *why = "hipMalloc (hand-over flag) failed";
    }
    if (ok) {
        void* p = nullptr;
        ok = hipHostMalloc(&p, kHostErrWords * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess;
        ch->host_err = static_cast<uint32_t*>(p);
        if (ok) {
            std::memset(ch->host_err, 0, kHostErrWords * sizeof(uint32_t));
            ok = hipHostGetDevicePointer(reinterpret_cast<void**>(&ch->host_err_dev), p, 0) == hipSuccess;
        }
        if (!ok) *why = "hipHostMalloc (error word) failed";
    }
    if (ok) ok = hipStreamSynchronize(nullptr) == hipSuccess; // (the two hipMemset calls above; nothing else of the process is waited for)
    if (ok) ok = hsa_signal_create(1, 0, nullptr, &ch->done) == HSA_STATUS_SUCCESS;
    if (ok) {
        void* p = nullptr;
        ok = hipMalloc(&p, 8 * kXccTableStride * 4) == hipSuccess && hipMemset(p, 0, 8 * kXccTableStride * 4) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess;
        ch->xcc_table = static_cast<uint32_t*>(p);
        if (!ok) *why = "hipMalloc (XCD table) failed";
    }
    if (ok && c->concurrent_handover == 0) ch->sync_mode = ch->forced_sync = true;
    if (ok && std::getenv("GYMRS_AQL_SYNC")) { // (developer knob: synchronous hand-over everywhere)
        ch->sync_mode = ch->forced_sync = true;
        std::lock_guard<std::mutex> lock(g_mu);
        if (c->concurrent_handover < 0) c->concurrent_handover = 0;
    }
    if (ok && c->concurrent_handover < 0) {
        // The asynchronous hand-over itself, once per device, on a stream of its own: begin -> end must let the stream through.
        // It needs a kernel of the stream and a kernel of the chain in flight TOGETHER; where they are not (rocprofv3 --pmc
        // serialises kernels across queues) the chain's first packet runs out its -- here: short -- bound, and the chains of
        // this device use the synchronous hand-over instead.
        hipStream_t probe = nullptr;
        ok = hipStreamCreateWithFlags(&probe, hipStreamNonBlocking) == hipSuccess;
        ch->wait_ticks = 100000000ull; // 1 s
        if (ok) ok = aql_begin(ch, probe, why) && aql_end(ch, probe, why);
        if (ok) {
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t q = hipErrorNotReady;
            while ((q = hipStreamQuery(probe)) == hipErrorNotReady && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(8)) {
            }
            const uint32_t gave_up = aql_take_error(ch);
            if (q != hipSuccess) {
                ok = false;
                *why = "hand-over self-check: the stream did not come through an empty chain within 8 s";
                std::lock_guard<std::mutex> lock(g_mu);
                c->ok = false; // (the stuck stream and this chain's queue are left alone; no further attempts on this device)
                c->why = *why;
            } else {
                std::lock_guard<std::mutex> lock(g_mu);
                c->concurrent_handover = gave_up ? 0 : 1;
                ch->sync_mode = ch->forced_sync = gave_up != 0;
            }
        }
        ch->wait_ticks = 10ull * 100000000ull;
        if (ok && probe) (void)hipStreamDestroy(probe);
    }
    if (!ok) {
        if (why->find("hand-over self-check") == std::string::npos) aql_destroy(ch);
        else c->live.fetch_sub(1, std::memory_order_relaxed);
        return nullptr;
    }
    c->epoch.fetch_add(1, std::memory_order_relaxed);
    return ch;
}

// really gives the queue back (a chain object whose queue reported an error, or whose set-up failed half-way)
static void aql_discard(AqlChain* c)
{
    if (!c) return;
    if (c->ctx) c->ctx->live.fetch_sub(1, std::memory_order_relaxed);
    if (c->q) hsa_queue_destroy(c->q);
    if (c->done.handle) hsa_signal_destroy(c->done);
    if (c->kernarg) hsa_amd_memory_pool_free(c->kernarg);
    if (c->xcc_table) (void)hipFree(c->xcc_table);
    if (c->in_flag) (void)hipFree(c->in_flag);
    if (c->out_flag) (void)hipFree(c->out_flag);
    if (c->host_err) (void)hipHostFree(c->host_err);
    // (an object that owned a queue is not freed: the queue's error callback holds a pointer into it, and this is a rare error path)
    if (!c->q) delete c;
}

// The engine is done with its chain object (its stream is idle: every chain ended with a wait on it).  A complete, healthy object is
// parked for the next engine of the device; anything else is discarded.
void aql_destroy(AqlChain* c, bool discard)
{
    if (!c) return;
    if (discard) {
        aql_discard(c);
        return;
    }
    const bool complete = c->ctx && c->q && c->kernarg && c->xcc_table && c->in_flag && c->out_flag && c->host_err && c->done.handle;
    if (!complete || c->in_chain || c->queue_status.load() != 0) {
        aql_discard(c);
        return;
    }
    std::lock_guard<std::mutex> lock(g_mu);
    c->ctx->parked.push_back(c);
}

const void* aql_code_blob(size_t* bytes)
{
#if !defined(__HIP_DEVICE_COMPILE__)
    *bytes = (size_t)(gymrs_aql_blob_end - gymrs_aql_blob_begin);
    return gymrs_aql_blob_begin;
#else
    *bytes = 0;
    return nullptr;
#endif
}

bool aql_kernel(AqlChain* c, const char* name, AqlKernel* out)
{
    std::lock_guard<std::mutex> lock(g_mu); // (engines of one device may ask from different threads: the in-process sharder's workers)
    auto it = c->ctx->kernels.find(name);
    if (it == c->ctx->kernels.end()) { // not one of the names checked when the code object was loaded: look it up now
        std::string why;
        if (!lookup_kernel(c->ctx, name, &why)) return false;
        it = c->ctx->kernels.find(name);
    }
    *out = it->second;
    return true;
}

bool aql_begin(AqlChain* c, hipStream_t stream, std::string* why)
{
    if (c->in_chain) {
        *why = "aql_begin inside a chain";
        return false;
    }
    c->seq += 1;
    if (c->sync_mode) {
        HIP_OK(hipStreamSynchronize(stream), "hipStreamSynchronize (synchronous hand-over)");
        c->in_chain = true;
        return true;
    }
    // behind everything enqueued on the engine's stream so far ...
    launch_begin();
    hipLaunchKernelGGL(aql_hip_set_flag, dim3(1), dim3(64), 0, stream, c->in_flag, c->seq);
    HIP_OK(hipGetLastError(), "hand-over into the chain");
    // ... and the chain's first packet waits for it
    struct {
        const uint32_t* flag;
        uint32_t seq;
        uint32_t pad;
        uint32_t* err;
        unsigned long long max_ticks;
    } args{c->in_flag, c->seq, 0, c->host_err_dev, c->wait_ticks};
    if (!c->stage(c->k_wait, 64, 64, &args, sizeof(args), HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_NONE, hsa_signal_t{0}, why)) return false;
    c->in_chain = true;
    return true;
}

bool aql_dispatch(AqlChain* c, const AqlKernel& k, uint32_t grid_workitems, uint32_t workgroup_size, const void* kernarg, size_t bytes, std::string* why,
                  bool release, bool system_acquire)
{
    // barrier bit: after the previous packet has completed.  Agent-scope ACQUIRE (L1 and scalar caches start clean: free,
    // measured), NO release: the lines this launch leaves dirty in an XCD's L2 are read by the next launch on that same XCD.
    // (GYMRS_AQL_FENCES="<acquire><release>", digits 0 none / 1 agent / 2 system: developer knob for A/B runs)
    static const int acq = [] { const char* v = std::getenv("GYMRS_AQL_FENCES"); return v && v[0] >= '0' && v[0] <= '2' ? v[0] - '0' : (int)HSA_FENCE_SCOPE_AGENT; }();
    static const int rel = [] { const char* v = std::getenv("GYMRS_AQL_FENCES"); return v && v[0] && v[1] >= '0' && v[1] <= '2' ? v[1] - '0' : (int)HSA_FENCE_SCOPE_NONE; }();
    // release = true: HIP's own header on this packet (agent-scope acquire AND release): what the launch wrote is written back when it ends -- the
    // per-step-visible shape submitted through this queue (gymrs_engine.hip, GYMRS_AQL=2)
    // (round 5 tried, on this path, what the HIP runtime's own queues and packets have -- a completion signal per packet, a multi-producer / high-priority /
    // profiled queue, 3-D dispatches: none moved a launch by more than 0.02 us; the experiment switch is gone again, profiles/r05_visible_through_queue.log)
    // system_acquire: the FIRST launch of a chain.  The chain's opening packet (the one-wave wait for the stream) carries a system-scope acquire too, but the packet
    // processor performs a packet's acquire when it STARTS the packet -- i.e. possibly before the stream work the wait orders the chain against (a fold of the reset
    // log, gymrs_stats_clear's memset, a copy) has run -- and the wait kernel's own fence only reaches the XCD it runs on.  An acquire has to FOLLOW the synchronisation:
    // so the first launch behind the wait invalidates every L2 once more.  (Round 5's soak: one run in ~200 of 8 processes sharing a GPU kept, in one process, the
    // episode counters an XCD's L2 still held from before gymrs_stats_clear -- profiles/r05_suite_soak.log part 3.)
    return c->stage(k, grid_workitems, workgroup_size, kernarg, bytes, system_acquire ? (int)HSA_FENCE_SCOPE_SYSTEM : acq, release ? (int)HSA_FENCE_SCOPE_AGENT : rel,
                    hsa_signal_t{0}, why);
}

bool aql_end(AqlChain* c, hipStream_t stream, std::string* why)
{
    if (!c->in_chain) {
        *why = "aql_end outside a chain";
        return false;
    }
    c->in_chain = false;
    // Two packets close the chain.  The first exists for its END-OF-KERNEL system-scope release: everything the chain left
    // dirty in the L2s is written back when it completes (its own store goes to a scratch word).  The second starts after
    // that -- barrier bit -- and only then tells the stream to go on: a flag stored by the packet that also carries the
    // release would become visible BEFORE the write-back it announces.
    struct {
        uint32_t* flag;
        uint32_t seq;
    } scratch{c->in_flag + 8, c->seq}, args{c->out_flag, c->seq};
    if (c->sync_mode) { // the host itself waits for the packet whose release writes the chain back
        hsa_signal_store_relaxed(c->done, 1);
        if (!c->stage(c->k_set, 64, 64, &scratch, sizeof(scratch), HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_SYSTEM, c->done, why)) return false;
        c->publish();
        if (hsa_signal_wait_scacquire(c->done, HSA_SIGNAL_CONDITION_LT, 1, 120ull * 1000 * 1000 * 1000, HSA_WAIT_STATE_BLOCKED) >= 1) {
            *why = "a chain did not finish within 120 s (synchronous hand-over)";
            return false;
        }
        return true;
    }
    if (!c->stage(c->k_set, 64, 64, &scratch, sizeof(scratch), HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_SYSTEM, hsa_signal_t{0}, why)) return false;
    if (!c->stage(c->k_set, 64, 64, &args, sizeof(args), HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_SYSTEM, hsa_signal_t{0}, why)) return false;
    c->publish();
    launch_begin();
    hipLaunchKernelGGL(aql_hip_wait_flag, dim3(1), dim3(64), 0, stream, c->out_flag, c->seq, c->host_err_dev);
    HIP_OK(hipGetLastError(), "hand-over back to the stream");
    return true;
}

bool aql_is_synchronous(const AqlChain* c) { return c && c->sync_mode; }

// Does the asynchronous hand-over suit THIS pair of queues (the stream's and the chain's)?  Three short chains of the counting
// kernel each way on the engine's own stream, host clock around call + synchronise; the synchronous hand-over is taken when a
// chain behind the sleeping kernel is clearly slower (> 1.25 x).  ~3 ms, once per engine and stream.
const char* aql_calibrate(AqlChain* c, hipStream_t stream, bool own_stream)
{
    if (c->forced_sync) return "synchronous (kernels of two queues do not run side by side here)";
    static thread_local char note[160];
    const uint32_t epoch = c->ctx->epoch.load(std::memory_order_relaxed);
    if (c->calibrated && c->calibrated_for == stream && c->calibrated_epoch == epoch) return nullptr; // (unchanged)
    c->calibrated = true;
    c->calibrated_for = stream;
    c->calibrated_epoch = epoch;
    c->sync_mode = false;
    if (const char* v = std::getenv("GYMRS_AQL_HANDOVER")) { // (developer knob: "kernel" / "sync")
        c->sync_mode = v[0] == 's';
        return c->sync_mode ? "synchronous (forced)" : "asynchronous (forced)";
    }
    // A caller-provided stream is never waited for here (it may be blocked on work its owner has not submitted yet, or be under a
    // capture): its chains use the asynchronous hand-over, whose only waits are the bounded ones on the device.
    if (!own_stream) return "asynchronous (caller-provided stream: not calibrated)";
    float* x = c->ctx->calib_buf;
    if (!x) return "asynchronous (not calibrated)";
    constexpr uint32_t kN4 = kCalibItems;
    const AqlKernel k = c->ctx->kernels["gymrs_aql_selfcheck"];
    SelfCheckArgs args{x, kN4, 0u, nullptr, nullptr};
    std::string why;
    auto time_chains = [&](bool sync) -> double {
        c->sync_mode = sync;
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            if (hipStreamSynchronize(stream) != hipSuccess) return 1e30;
            const auto t0 = std::chrono::steady_clock::now();
            if (!aql_begin(c, stream, &why)) return 1e30;
            bool ok = true;
            for (int t = 0; t < 128 && ok; ++t) ok = aql_dispatch(c, k, kN4, 256, &args, sizeof(args), &why);
            if (!aql_end(c, stream, &why) || !ok) return 1e30;
            if (hipStreamSynchronize(stream) != hipSuccess) return 1e30;
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            best = us < best ? us : best;
        }
        return best;
    };
    const double t_async = time_chains(false), t_sync = time_chains(true);
    c->sync_mode = t_async > 1.25 * t_sync;
    std::snprintf(note, sizeof(note), "%s (128-launch probe chains: %.0f us asynchronous, %.0f us synchronous)",
                  c->sync_mode ? "synchronous: a kernel on the stream's queue slows this chain's queue" : "asynchronous", t_async, t_sync);
    return note;
}

uint32_t* aql_xcc_table(const AqlChain* c) { return c->xcc_table; }
uint32_t aql_chain_number(const AqlChain* c) { return c->seq & 0xffffffu; }

uint32_t aql_take_error(AqlChain* c)
{
    const uint32_t v = c->host_err ? c->host_err[0] : 0;
    if (v) c->host_err[0] = 0;
    return v | (c->queue_status.load() != 0 ? 2u : 0u);
}

} // namespace gymrs
