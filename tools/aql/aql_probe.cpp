// tools/aql/aql_probe.cpp -- developer experiment (VERDICT r2 "next" #5): what does the inter-kernel boundary cost, and how much of it
// goes away when consecutive launches are NOT separated by the AQL barrier bit?  Talks to the GPU through HSA directly (own
// queue, hand-written AQL dispatch packets), because HIP's any-order flag is documented as unsupported on gfx9.
//
//   g++ -O2 -std=c++17 aql_probe.cpp -I/opt/rocm/include -L/opt/rocm/lib -lhsa-runtime64 -o aql_probe
//   ./aql_probe aql_kernels.hsaco [lanes=1048576] [launches=4000]
//
// Variants (each: `launches` back-to-back dispatches of one kernel, wall time from the first doorbell to the last completion):
//   empty/barrier, empty/free                    the fixed cost of a dependent launch vs an independent one
//   stepish/barrier                              the step kernel's memory shape, launches ordered by the barrier bit (what HIP does)
//   stepish/free (RACY, timing only)             the same with the barrier bit cleared: launches overlap, no ordering at all
//   stepish/free + version words (1 = agent-scope release, 2 = vmcnt(0) only)
//                                                ordered per wavefront by a version word instead of per launch by the barrier bit;
//                                                every lane counts its launches, so a violated order is SEEN (x0 != launches)
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <immintrin.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hsa_status_t s_ = (x);                                                    \
        if (s_ != HSA_STATUS_SUCCESS) {                                           \
            const char* m_ = nullptr;                                             \
            hsa_status_string(s_, &m_);                                           \
            std::fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, m_); \
            std::exit(1);                                                         \
        }                                                                         \
    } while (0)

struct Args {
    float* s[4];
    const uint8_t* a;
    float* r;
    uint8_t* d;
    uint32_t* ver;
    uint32_t* err;
    uint32_t t, alu, versioned, pad;
#ifdef FAT_ARGS
    uint32_t fat[66];
#endif
};

static hsa_agent_t g_gpu, g_cpu;
static hsa_amd_memory_pool_t g_gpu_pool, g_kernarg_pool, g_host_pool;

static hsa_status_t on_agent(hsa_agent_t a, void*)
{
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && g_gpu.handle == 0) g_gpu = a;
    if (t == HSA_DEVICE_TYPE_CPU && g_cpu.handle == 0) g_cpu = a;
    return HSA_STATUS_SUCCESS;
}

static hsa_status_t on_gpu_pool(hsa_amd_memory_pool_t p, void*)
{
    hsa_amd_segment_t seg;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    bool alloc = false;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && g_gpu_pool.handle == 0) g_gpu_pool = p;
    return HSA_STATUS_SUCCESS;
}

static hsa_status_t on_cpu_pool(hsa_amd_memory_pool_t p, void*)
{
    hsa_amd_segment_t seg;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && g_kernarg_pool.handle == 0) g_kernarg_pool = p;
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED) && g_host_pool.handle == 0) g_host_pool = p;
    return HSA_STATUS_SUCCESS;
}

static void* gpu_alloc(size_t bytes)
{
    void* p = nullptr;
    CHECK(hsa_amd_memory_pool_allocate(g_gpu_pool, bytes, 0, &p));
    return p;
}

static void* host_alloc(hsa_amd_memory_pool_t pool, size_t bytes)
{
    void* p = nullptr;
    CHECK(hsa_amd_memory_pool_allocate(pool, bytes, 0, &p));
    CHECK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, p));
    return p;
}

struct Kernel {
    uint64_t object = 0;
    uint32_t kernarg = 0, group = 0, priv = 0;
};

static Kernel find_kernel(hsa_executable_t exe, const char* name)
{
    hsa_executable_symbol_t sym;
    CHECK(hsa_executable_get_symbol_by_name(exe, name, &g_gpu, &sym));
    Kernel k;
    CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
    CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
    CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
    return k;
}

static uint16_t header(bool barrier, int acquire, int release)
{
    return (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                      (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
}

int main(int argc, char** argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s aql_kernels.hsaco [lanes] [launches]\n", argv[0]);
        return 2;
    }
    const size_t n = argc > 2 ? std::strtoull(argv[2], nullptr, 0) : (1u << 20);
    const uint32_t launches = argc > 3 ? (uint32_t)std::strtoul(argv[3], nullptr, 0) : 4000;
    CHECK(hsa_init());
    CHECK(hsa_iterate_agents(on_agent, nullptr));
    CHECK(hsa_amd_agent_iterate_memory_pools(g_gpu, on_gpu_pool, nullptr));
    CHECK(hsa_amd_agent_iterate_memory_pools(g_cpu, on_cpu_pool, nullptr));
    if (!g_gpu.handle || !g_gpu_pool.handle || !g_kernarg_pool.handle) {
        std::fprintf(stderr, "no GPU agent / pools\n");
        return 1;
    }
    char name[64] = {0};
    hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_NAME, name);

    // code object
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    hsa_code_object_reader_t reader;
    CHECK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &reader));
    hsa_executable_t exe;
    CHECK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    CHECK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    CHECK(hsa_executable_freeze(exe, nullptr));
    const Kernel k_empty = find_kernel(exe, "empty_kernel.kd"), k_step = find_kernel(exe, "stepish_kernel.kd");

    hsa_queue_t* q = nullptr;
    const bool multi = std::getenv("AQL_QUEUE_MULTI") != nullptr, zero_hints = std::getenv("AQL_ZERO_HINTS") != nullptr;
    CHECK(hsa_queue_create(g_gpu, 4096, multi ? HSA_QUEUE_TYPE_MULTI : HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, zero_hints ? 0 : UINT32_MAX,
                           zero_hints ? 0 : UINT32_MAX, &q));
    if (std::getenv("AQL_PROFILING")) CHECK(hsa_amd_profiling_set_profiler_enabled(q, 1));
    hsa_signal_t done;
    CHECK(hsa_signal_create(1, 0, nullptr, &done));

    // data
    const size_t waves = n / 4 / 64;
    Args base{};
    for (int j = 0; j < 4; ++j) base.s[j] = (float*)gpu_alloc(n * 4 + 4352 * (j + 1)) + 1088 * (j + 1) / 4 * 4;
    base.a = (uint8_t*)gpu_alloc(n + 64);
    base.r = (float*)gpu_alloc(n * 4 + 64);
    base.d = (uint8_t*)gpu_alloc(n + 64);
    base.ver = (uint32_t*)gpu_alloc(waves * 4 + 64);
    uint32_t* err_host = (uint32_t*)host_alloc(g_host_pool, 64);
    base.err = err_host;
    std::vector<float> zeros(n, 0.0f);
    std::vector<float> host(n);
    // Kernel arguments: in DEVICE memory, written by the CPU through the PCIe BAR (what HIP does by default on this part,
    // HIP_FORCE_DEV_KERNARG): with the ring in host memory every scalar cache fetches its copy of the arguments over PCIe and
    // a 2^20-lane launch takes 28 us instead of 5.5 (the first version of this probe, profiles/r03_aql_probe_host_kernarg.log).
    const bool host_kernarg = std::getenv("AQL_HOST_KERNARG") != nullptr;
    // AQL_PREFILLED_KERNARG: device memory WITHOUT a CPU mapping, all slots filled up front by hsa_memory_copy (the launch number
    // `t` then repeats every 4096 launches, so only the unversioned variants are meaningful)
    const bool prefilled = std::getenv("AQL_PREFILLED_KERNARG") != nullptr;
    Args* kernargs = nullptr;
    if (host_kernarg) {
        kernargs = (Args*)host_alloc(g_kernarg_pool, sizeof(Args) * 4096 + 4096);
    } else {
        kernargs = (Args*)gpu_alloc(sizeof(Args) * 4096 + 4096);
        if (!prefilled) CHECK(hsa_amd_agents_allow_access(1, &g_cpu, nullptr, kernargs));
    }
    std::vector<Args> host_args(4096);
    auto fill = [&](void* dst, const void* src, size_t bytes) { CHECK(hsa_memory_copy(dst, src, bytes)); };

    std::printf("# agent %s, %zu lanes (%zu waves), %u launches per variant; %zu B read + %zu B written per launch\n", name, n, waves, launches,
                n * 17, n * 21);
    std::printf("%-44s %10s %10s %10s %s\n", "variant", "us/launch", "min3", "GB/s(38B)", "check");

    // ROCclr-style bookkeeping: a completion signal on EVERY packet (a ring of signals with a huge initial value: never re-armed)
    std::vector<hsa_signal_t> ring_sig(256);
    for (auto& sg : ring_sig) CHECK(hsa_signal_create((hsa_signal_value_t)1 << 40, 0, nullptr, &sg));
    bool sig_every = false;
    struct Pend { hsa_kernel_dispatch_packet_t* p; uint32_t h; uint64_t idx; };
    std::vector<Pend> pending;
    const size_t batch = std::getenv("AQL_BATCH") ? std::strtoul(std::getenv("AQL_BATCH"), nullptr, 0) : 1;
    const uint64_t depth = std::getenv("AQL_DEPTH") ? std::strtoul(std::getenv("AQL_DEPTH"), nullptr, 0) : 0;
    auto run = [&](const char* label, const Kernel& k, bool barrier, uint32_t versioned, uint32_t alu, int acquire, int release) {
        double best = 1e30, sum = 0;
        std::string check = "-";
        for (int rep = 0; rep < 4; ++rep) { // rep 0 = warm-up
            for (int j = 0; j < 4; ++j) fill(base.s[j], zeros.data(), n * 4);
            fill(base.ver, zeros.data(), waves * 4);
            fill((void*)base.a, zeros.data(), n);
            err_host[0] = err_host[1] = 0;
            hsa_signal_store_relaxed(done, 1);
            if (prefilled) {
                for (uint32_t i = 0; i < 4096; ++i) {
                    host_args[i] = base;
                    host_args[i].t = i;
                    host_args[i].alu = alu;
                    host_args[i].versioned = versioned;
                }
                fill(kernargs, host_args.data(), sizeof(Args) * 4096);
            }
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t t = 0; t < launches; ++t) {
                const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
                while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {
                }
                Args* ka = &kernargs[idx & 4095];
                Args tmp = base;
                tmp.t = t;
                tmp.alu = alu;
                tmp.versioned = versioned;
                if (!prefilled) std::memcpy((void*)ka, &tmp, sizeof(tmp));
                if (!host_kernarg && !prefilled) { // write-combined BAR stores: drain them, then make sure they have landed before the doorbell rings
                    _mm_sfence();
                    (void)*(volatile uint32_t*)&ka->pad;
                }
                auto* p = &((hsa_kernel_dispatch_packet_t*)q->base_address)[idx & (q->size - 1)];
                p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
                p->workgroup_size_x = 512;
                p->workgroup_size_y = p->workgroup_size_z = 1;
                p->grid_size_x = (uint32_t)(n / 4);
                p->grid_size_y = p->grid_size_z = 1;
                p->private_segment_size = k.priv;
                p->group_segment_size = k.group;
                p->kernel_object = k.object;
                p->kernarg_address = ka;
                p->reserved2 = 0;
                const bool last = t + 1 == launches;
                p->completion_signal = last ? done : (sig_every ? ring_sig[idx & 255] : hsa_signal_t{0});
                // the LAST packet always carries the barrier bit and a system-scope release: its completion = everything is done
                const uint16_t h = last ? header(true, acquire, HSA_FENCE_SCOPE_SYSTEM) : header(barrier, acquire, release);
                if (batch <= 1) {
                    __atomic_store_n((uint32_t*)p, (uint32_t)h | ((uint32_t)p->setup << 16), __ATOMIC_RELEASE);
                    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
                } else { // the engine's dispatcher: headers and ONE doorbell per `batch` packets
                    pending.push_back({p, (uint32_t)h | ((uint32_t)p->setup << 16), idx});
                    if (pending.size() >= batch || last) {
                        for (auto& pd : pending) __atomic_store_n((uint32_t*)pd.p, pd.h, __ATOMIC_RELEASE);
                        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)pending.back().idx);
                        pending.clear();
                    }
                }
                while (depth && idx - hsa_queue_load_read_index_scacquire(q) >= depth) { // pace the host like a slow launcher
                }
            }
            while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {
            }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / launches;
            if (rep > 0) {
                best = us < best ? us : best;
                sum += us;
            }
            if (k.object == k_step.object && rep == 3) {
                fill(host.data(), base.s[0], n * 4);
                size_t wrong = 0;
                for (size_t i = 0; i < n; ++i) wrong += host[i] != (float)launches;
                char buf[160];
                std::snprintf(buf, sizeof(buf), "lanes with x0 != %u: %zu; spins that ran out: %u", launches, wrong, err_host[0]);
                check = buf;
            }
        }
        std::printf("%-44s %10.3f %10.3f %10.0f %s\n", label, sum / 3, best, (double)n * 38 / (sum / 3 * 1e-6) / 1e9, check.c_str());
        std::fflush(stdout);
    };

    const int NONE = HSA_FENCE_SCOPE_NONE, AGENT = HSA_FENCE_SCOPE_AGENT;
    run("empty, barrier bit, no fences", k_empty, true, 0, 0, NONE, NONE);
    run("empty, barrier bit, agent fences", k_empty, true, 0, 0, AGENT, AGENT);
    run("empty, free (no barrier bit)", k_empty, false, 0, 0, NONE, NONE);
    sig_every = true;
    run("empty, barrier, signal on every packet", k_empty, true, 0, 0, NONE, NONE);
    run("empty, barrier, agent fences, signal every", k_empty, true, 0, 0, AGENT, AGENT);
    run("stepish alu=100, barrier, signal every", k_step, true, 0, 100, NONE, NONE);
    run("stepish alu=100, barrier, agent f., signal every", k_step, true, 0, 100, AGENT, AGENT);
    sig_every = false;
    for (uint32_t alu : {0u, 100u}) {
        char l[96];
        std::snprintf(l, sizeof(l), "stepish alu=%u, barrier, no fences", alu);
        run(l, k_step, true, 0, alu, NONE, NONE);
        std::snprintf(l, sizeof(l), "stepish alu=%u, barrier, agent fences", alu);
        run(l, k_step, true, 0, alu, AGENT, AGENT);
        std::snprintf(l, sizeof(l), "stepish alu=%u, barrier, agent ACQUIRE only", alu);
        run(l, k_step, true, 0, alu, AGENT, NONE);
        std::snprintf(l, sizeof(l), "stepish alu=%u, barrier, agent RELEASE only", alu);
        run(l, k_step, true, 0, alu, NONE, AGENT);
        std::snprintf(l, sizeof(l), "stepish alu=%u, barrier, system fences", alu);
        run(l, k_step, true, 0, alu, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM);
        std::snprintf(l, sizeof(l), "stepish alu=%u, free, RACY (timing only)", alu);
        run(l, k_step, false, 0, alu, NONE, NONE);
        if (!std::getenv("AQL_VERSION_WORDS")) continue;
        std::snprintf(l, sizeof(l), "stepish alu=%u, free + version words (1)", alu);
        run(l, k_step, false, 1, alu, NONE, NONE);
        std::snprintf(l, sizeof(l), "stepish alu=%u, free + version words (2)", alu);
        run(l, k_step, false, 2, alu, NONE, NONE);
        std::snprintf(l, sizeof(l), "stepish alu=%u, barrier + version words (2)", alu);
        run(l, k_step, true, 2, alu, NONE, NONE);
    }
    hsa_queue_destroy(q);
    hsa_shut_down();
    return 0;
}
