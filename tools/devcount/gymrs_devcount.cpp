// tools/devcount/gymrs_devcount.cpp -- developer tool (VERDICT r3 "next" #1c): DEVICE-WIDE counter sampling through
// rocprofiler-sdk's device counting service, so that the fabric traffic of a FREE-RUNNING chain can be measured: rocprofv3 --pmc
// counts per dispatch and serialises kernels across queues, which is exactly what a chain must not be subjected to.  Here the
// counters of the whole agent run while the application does what it does; the application brackets a window with two samples.
//
//   g++ -O2 -std=c++17 -fPIC -shared gymrs_devcount.cpp -I/opt/rocm/include -L/opt/rocm/lib -lrocprofiler-sdk -o libgymrs_devcount.so
//   ROCP_TOOL_LIBRARIES=$PWD/libgymrs_devcount.so python chain_traffic.py      (the script dlopens the same file for the calls below)
//
//   int gymrs_devcount_start(const char* counters_csv);   0 = ok; the counters start counting (agent-wide)
//   int gymrs_devcount_sample(double* out, int n);        one value per requested counter (summed over its instances)
//   int gymrs_devcount_stop(void);
//   const char* gymrs_devcount_error(void);
#include <rocprofiler-sdk/registration.h>
#include <rocprofiler-sdk/rocprofiler.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {

std::string g_err;
rocprofiler_context_id_t g_ctx{};
rocprofiler_buffer_id_t g_buf{};
rocprofiler_agent_id_t g_agent{};
bool g_have_agent = false, g_configured = false, g_started = false;
rocprofiler_counter_config_id_t g_profile{.handle = 0};
std::vector<std::string> g_names;
std::map<uint64_t, int> g_slot_of_counter; // counter id handle -> index into g_names
size_t g_records = 0;

#define RP(call)                                                                          \
    do {                                                                                  \
        const rocprofiler_status_t s_ = (call);                                           \
        if (s_ != ROCPROFILER_STATUS_SUCCESS) {                                           \
            g_err = std::string(#call) + ": " + rocprofiler_get_status_string(s_);        \
            return -1;                                                                    \
        }                                                                                 \
    } while (0)

int tool_init(rocprofiler_client_finalize_t, void*)
{
    // first GPU agent
    auto cb = [](rocprofiler_agent_version_t, const void** agents, size_t n, void*) {
        for (size_t i = 0; i < n; ++i) {
            const auto* a = static_cast<const rocprofiler_agent_v0_t*>(agents[i]);
            if (a->type == ROCPROFILER_AGENT_TYPE_GPU && !g_have_agent) {
                g_agent = a->id;
                g_have_agent = true;
            }
        }
        return ROCPROFILER_STATUS_SUCCESS;
    };
    RP(rocprofiler_query_available_agents(ROCPROFILER_AGENT_INFO_VERSION_0, cb, sizeof(rocprofiler_agent_t), nullptr));
    if (!g_have_agent) {
        g_err = "no GPU agent";
        return -1;
    }
    RP(rocprofiler_create_context(&g_ctx));
    RP(rocprofiler_create_buffer(g_ctx, 4096, 2048, ROCPROFILER_BUFFER_POLICY_LOSSLESS,
                                 [](rocprofiler_context_id_t, rocprofiler_buffer_id_t, rocprofiler_record_header_t**, size_t, void*, uint64_t) {}, nullptr,
                                 &g_buf));
    rocprofiler_callback_thread_t th{};
    RP(rocprofiler_create_callback_thread(&th));
    RP(rocprofiler_assign_callback_thread(g_buf, th));
    RP(rocprofiler_configure_device_counting_service(
        g_ctx, g_buf, g_agent,
        [](rocprofiler_context_id_t ctx, rocprofiler_agent_id_t, rocprofiler_device_counting_agent_cb_t set_config, void*) {
            if (g_profile.handle != 0) set_config(ctx, g_profile);
        },
        nullptr));
    g_configured = true;
    return 0;
}

void tool_fini(void*) {}

} // namespace

extern "C" {

const char* gymrs_devcount_error(void) { return g_err.c_str(); }

int gymrs_devcount_start(const char* csv)
{
    if (!g_configured) {
        if (g_err.empty()) g_err = "the tool was not registered (start the process with ROCP_TOOL_LIBRARIES=<this library>)";
        return -1;
    }
    if (g_started) {
        g_err = "already started";
        return -1;
    }
    g_names.clear();
    g_slot_of_counter.clear();
    std::stringstream ss(csv);
    for (std::string item; std::getline(ss, item, ',');)
        if (!item.empty()) g_names.push_back(item);
    // the agent's counters by name
    std::vector<rocprofiler_counter_id_t> all;
    RP(rocprofiler_iterate_agent_supported_counters(
        g_agent,
        [](rocprofiler_agent_id_t, rocprofiler_counter_id_t* c, size_t n, void* u) {
            auto* v = static_cast<std::vector<rocprofiler_counter_id_t>*>(u);
            for (size_t i = 0; i < n; ++i) v->push_back(c[i]);
            return ROCPROFILER_STATUS_SUCCESS;
        },
        &all));
    std::vector<rocprofiler_counter_id_t> want;
    g_records = 0;
    for (size_t k = 0; k < g_names.size(); ++k) {
        bool found = false;
        for (const auto& c : all) {
            rocprofiler_counter_info_v1_t info;
            if (rocprofiler_query_counter_info(c, ROCPROFILER_COUNTER_INFO_VERSION_1, &info) != ROCPROFILER_STATUS_SUCCESS) continue;
            if (g_names[k] == info.name) {
                want.push_back(c);
                g_slot_of_counter[c.handle] = (int)k;
                g_records += info.dimensions_instances_count;
                found = true;
                break;
            }
        }
        if (!found) {
            g_err = "counter not supported on this agent: " + g_names[k];
            return -1;
        }
    }
    RP(rocprofiler_create_counter_config(g_agent, want.data(), want.size(), &g_profile));
    RP(rocprofiler_start_context(g_ctx));
    g_started = true;
    return 0;
}

int gymrs_devcount_sample(double* out, int n)
{
    if (!g_started) {
        g_err = "not started";
        return -1;
    }
    std::vector<rocprofiler_counter_record_t> recs(g_records + 64);
    size_t count = recs.size();
    RP(rocprofiler_sample_device_counting_service(g_ctx, {}, ROCPROFILER_COUNTER_FLAG_NONE, recs.data(), &count));
    for (int k = 0; k < n; ++k) out[k] = 0.0;
    for (size_t i = 0; i < count; ++i) {
        rocprofiler_counter_id_t cid{.handle = 0};
        rocprofiler_query_record_counter_id(recs[i].id, &cid);
        auto it = g_slot_of_counter.find(cid.handle);
        if (it != g_slot_of_counter.end() && it->second < n) out[it->second] += recs[i].counter_value;
    }
    return (int)count;
}

int gymrs_devcount_stop(void)
{
    if (!g_started) return 0;
    g_started = false;
    RP(rocprofiler_stop_context(g_ctx));
    return 0;
}

rocprofiler_tool_configure_result_t* rocprofiler_configure(uint32_t, const char*, uint32_t, rocprofiler_client_id_t* id)
{
    id->name = "gymrs_devcount";
    static rocprofiler_tool_configure_result_t cfg{sizeof(rocprofiler_tool_configure_result_t), &tool_init, &tool_fini, nullptr};
    return &cfg;
}
}
