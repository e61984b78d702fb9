// tools/aql/visible_probe.cpp -- developer experiment (VERDICT r3 "next" #2): a release-free VISIBLE step?
// Own HSA queue, hand-written AQL packets (as aql_probe.cpp).  One "pair" = a step launch (17 B read + 21 B written per lane) and a
// reader launch with a permuted mapping (another XCD) that checks every value of that step and writes the next step's actions,
// which the step checks.  Per variant: the cache policy of the step's stores and of the reader's action stores, and the fence
// scopes of both packets.  Prints us per pair, us per step without a reader, and the number of STALE values either side saw.
//
//   g++ -O2 -std=c++17 visible_probe.cpp -I/opt/rocm/include -L/opt/rocm/lib -lhsa-runtime64 -o visible_probe
//   ./visible_probe visible_kernels.hsaco [lanes=1048576] [pairs=3000]
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

struct VArgs {
    float* s[4];
    uint8_t* a;
    float* r;
    uint8_t* d;
    uint32_t* err;
    uint32_t t, alu, n4, check;
};

static hsa_agent_t g_gpu, g_cpu;
static hsa_amd_memory_pool_t g_gpu_pool;

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

static void* gpu_alloc(size_t bytes)
{
    void* p = nullptr;
    CHECK(hsa_amd_memory_pool_allocate(g_gpu_pool, bytes, 0, &p));
    return p;
}

struct Kernel {
    uint64_t object = 0;
    uint32_t kernarg = 0, group = 0, priv = 0;
};

static Kernel find_kernel(hsa_executable_t exe, const char* name)
{
    hsa_executable_symbol_t sym;
    CHECK(hsa_executable_get_symbol_by_name(exe, (std::string(name) + ".kd").c_str(), &g_gpu, &sym));
    Kernel k;
    CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
    CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
    CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
    return k;
}

static uint16_t header(int acquire, int release)
{
    return (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                      (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
}

int main(int argc, char** argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s visible_kernels.hsaco [lanes] [pairs]\n", argv[0]);
        return 2;
    }
    const size_t n = argc > 2 ? std::strtoull(argv[2], nullptr, 0) : (1u << 20);
    const uint32_t pairs = argc > 3 ? (uint32_t)std::strtoul(argv[3], nullptr, 0) : 3000;
    const uint32_t alu = std::getenv("VIS_ALU") ? (uint32_t)std::strtoul(std::getenv("VIS_ALU"), nullptr, 0) : 100u;
    CHECK(hsa_init());
    CHECK(hsa_iterate_agents(on_agent, nullptr));
    CHECK(hsa_amd_agent_iterate_memory_pools(g_gpu, on_gpu_pool, nullptr));
    char name[64] = {0};
    hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_NAME, name);

    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    hsa_code_object_reader_t reader;
    CHECK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &reader));
    hsa_executable_t exe;
    CHECK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    CHECK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    CHECK(hsa_executable_freeze(exe, nullptr));

    hsa_queue_t* q = nullptr;
    CHECK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    hsa_signal_t done;
    CHECK(hsa_signal_create(1, 0, nullptr, &done));

    VArgs base{};
    for (int j = 0; j < 4; ++j) base.s[j] = (float*)gpu_alloc(n * 4 + 4352 * (j + 1)) + 1088 * (j + 1) / 4 * 4;
    base.a = (uint8_t*)gpu_alloc(n + 64);
    base.r = (float*)gpu_alloc(n * 4 + 64);
    base.d = (uint8_t*)gpu_alloc(n + 64);
    base.err = (uint32_t*)gpu_alloc(64);
    base.n4 = (uint32_t)(n / 4);
    base.alu = alu;
    std::vector<float> zeros(n, 0.0f);
    VArgs* kernargs = (VArgs*)gpu_alloc(sizeof(VArgs) * 4096 + 4096);
    CHECK(hsa_amd_agents_allow_access(1, &g_cpu, nullptr, kernargs)); // written through the PCIe BAR, as the engine's dispatcher does

    std::printf("# agent %s, %zu lanes, %u launches (pairs) per variant, %u dependent FMAs per lane; step: %zu B read + %zu B written per launch\n", name, n,
                pairs, alu, n * 17, n * 21);
    std::printf("%-78s %9s %9s %12s %12s\n", "variant (fence scopes: 0 none, 1 agent)", "us/pair", "min3", "stale@reader", "stale@step");

    auto dispatch = [&](const Kernel& k, const VArgs& args, int acquire, int release, hsa_signal_t completion) {
        const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
        while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {
        }
        VArgs* ka = &kernargs[idx & 4095];
        std::memcpy((void*)ka, &args, sizeof(args));
        _mm_sfence();
        (void)*(volatile uint32_t*)&ka->check;
        auto* p = &((hsa_kernel_dispatch_packet_t*)q->base_address)[idx & (q->size - 1)];
        p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        p->workgroup_size_x = 512;
        p->workgroup_size_y = p->workgroup_size_z = 1;
        p->grid_size_x = (uint32_t)((n / 4 + 511) / 512 * 512);
        p->grid_size_y = p->grid_size_z = 1;
        p->private_segment_size = k.priv;
        p->group_segment_size = k.group;
        p->kernel_object = k.object;
        p->kernarg_address = ka;
        p->reserved2 = 0;
        p->completion_signal = completion;
        __atomic_store_n((uint32_t*)p, (uint32_t)header(acquire, release) | ((uint32_t)p->setup << 16), __ATOMIC_RELEASE);
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
    };

    // reader == nullptr: steps only
    auto run = [&](const char* label, const char* step_name, const char* reader_name, int s_acq, int s_rel, int r_acq, int r_rel) {
        const Kernel ks = find_kernel(exe, step_name);
        Kernel kr{};
        if (reader_name) kr = find_kernel(exe, reader_name);
        double best = 1e30, sum = 0;
        uint32_t err[2] = {0, 0};
        for (int rep = 0; rep < 4; ++rep) { // rep 0 = warm-up
            for (int j = 0; j < 4; ++j) CHECK(hsa_memory_copy(base.s[j], zeros.data(), n * 4));
            CHECK(hsa_memory_copy(base.a, zeros.data(), n));
            CHECK(hsa_memory_copy(base.r, zeros.data(), n * 4));
            CHECK(hsa_memory_copy(base.d, zeros.data(), n));
            CHECK(hsa_memory_copy(base.err, zeros.data(), 64));
            hsa_signal_store_relaxed(done, 1);
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t t = 0; t < pairs; ++t) {
                VArgs a = base;
                a.t = t;
                a.check = reader_name ? 1u : 0u;
                const bool last = t + 1 == pairs;
                if (reader_name) {
                    dispatch(ks, a, s_acq, s_rel, hsa_signal_t{0});
                    dispatch(kr, a, r_acq, last ? (int)HSA_FENCE_SCOPE_SYSTEM : r_rel, last ? done : hsa_signal_t{0});
                } else {
                    dispatch(ks, a, s_acq, last ? (int)HSA_FENCE_SCOPE_SYSTEM : s_rel, last ? done : hsa_signal_t{0});
                }
            }
            while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {
            }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / pairs;
            if (rep > 0) {
                best = us < best ? us : best;
                sum += us;
            }
            CHECK(hsa_memory_copy(err, base.err, sizeof(err)));
        }
        std::printf("%-78s %9.3f %9.3f %12u %12u\n", label, sum / 3, best, err[0], err[1]);
        std::fflush(stdout);
    };

    const int N = HSA_FENCE_SCOPE_NONE, A = HSA_FENCE_SCOPE_AGENT;
    std::printf("# --- steps only (no reader): what the store policy and the release cost a back-to-back step ---\n");
    for (const char* ld : {"", "_ldnt"}) {
        for (const char* pol : {"plain", "nt", "sc1", "sc0sc1"}) {
            char kn[64], l[128];
            std::snprintf(kn, sizeof(kn), "vstep_%s%s", pol, ld);
            std::snprintf(l, sizeof(l), "step %s%s stores, acquire only", pol, ld);
            run(l, kn, nullptr, A, N, 0, 0);
            std::snprintf(l, sizeof(l), "step %s%s stores, acquire + RELEASE (HIP's header)", pol, ld);
            run(l, kn, nullptr, A, A, 0, 0);
        }
    }
    run("step sc1nt stores, acquire only", "vstep_sc1nt", nullptr, A, N, 0, 0);
    run("step sc1nt stores, acquire + RELEASE", "vstep_sc1nt", nullptr, A, A, 0, 0);
    std::printf("# --- step + permuted reader (another XCD) per pair; stale counts must be 0 for a variant to be usable ---\n");
    run("A  step nt, acq+REL | reader plain, acq+REL  (= two HIP launches)", "vstep_nt", "vreader_plain", A, A, A, A);
    run("A' step plain, acq+REL | reader plain, acq+REL", "vstep_plain", "vreader_plain", A, A, A, A);
    run("B  step plain, acq only | reader plain, acq only  (expected: STALE)", "vstep_plain", "vreader_plain", A, N, A, N);
    run("B' step nt, acq only | reader plain, acq only  (expected: STALE)", "vstep_nt", "vreader_plain", A, N, A, N);
    run("C  step sc1, acq only | reader sc1, acq only", "vstep_sc1", "vreader_sc1", A, N, A, N);
    run("C' step sc1 (nt loads), acq only | reader sc1, acq only", "vstep_sc1_ldnt", "vreader_sc1", A, N, A, N);
    run("D  step sc0sc1, acq only | reader sc0sc1, acq only", "vstep_sc0sc1", "vreader_sc0sc1", A, N, A, N);
    run("E  step sc1nt, acq only | reader sc1, acq only", "vstep_sc1nt", "vreader_sc1", A, N, A, N);
    run("F  step sc1, acq+REL | reader sc1, acq+REL  (release with nothing dirty: scan cost)", "vstep_sc1", "vreader_sc1", A, A, A, A);
    run("G  step sc1, acq only | reader plain, acq+REL", "vstep_sc1", "vreader_plain", A, N, A, A);
    run("H  step nt, acq+REL | reader sc1, acq only", "vstep_nt", "vreader_sc1", A, A, A, N);
    run("I  reader only x2 per pair (plain, acq only): what two small launches cost", "vreader_plain", "vreader_plain", A, N, A, N);
    hsa_queue_destroy(q);
    hsa_shut_down();
    return 0;
}
