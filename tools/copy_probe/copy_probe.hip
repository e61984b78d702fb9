// copy_probe.hip -- the copy yardstick ("also measure an in-repo stream-copy kernel on the box"): gymrs_tool_copy_probe, the copy floor of a
// launch of a given footprint and the HBM figure, submitted through HIP or through a chain of the library's dispatcher like the steps it is compared with.
// A MEASUREMENT TOOL, not part of the C ABI: until round 4 this was gymrs_copy_probe in include/gymrs_amd.h (VERDICT r4 "next" #8).  Built by
// tools/copy_probe/build.py into tools/copy_probe/libgymrs_copy_probe.so from this file, the dispatcher's source (gym-rs_amd/csrc/gymrs_aql.hip, compiled
// here against THIS tool's code object: copy_probe_aql.hip) -- bench.py and tools/size_sweep.py load it with ctypes.
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "copy_probe_kernels.h"
#include "gymrs_aql.h"

using namespace gymrs;

static thread_local std::string g_probe_error;
static int probe_fail(int st, const std::string& msg)
{
    g_probe_error = msg;
    return st;
}
enum { PROBE_OK = 0, PROBE_EINVAL = 1, PROBE_EHIP = 2, PROBE_ENOMEM = 4 }; // (the values gymrs_status had for the same outcomes)
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t err_ = (expr);                                                                      \
        if (err_ != hipSuccess) return probe_fail(PROBE_EHIP, std::string(#expr) + ": " + hipGetErrorString(err_)); \
    } while (0)

// The plain copy a step launch of the same size is compared with (copy_probe_body,
// gymrs_tile.h).  Two shapes: at a step's footprint a work-item moves 4 items like the step kernel's tiles; from 1.5 GiB per launch on
// (the HBM figure: 1 GiB + 1 GiB) one item per work-item -- measured on MI355X (profiles/r04_hbm_probe.log): 6.61 TB/s against 6.24 for the 4-item
// shape and 5.9-6.1 for every persistent grid-stride form; the guide's own float4 copy reads 6.29.
template <bool NTL, bool NTS, int ITEMS>
__global__ __launch_bounds__(kProbeBlock) void copy_probe_kernel(const uint32_t* src, uint64_t n_read16, uint32_t* dst,
                                                            uint64_t n_write16)
{
    copy_probe_body<NTL, NTS, ITEMS>(src, n_read16, dst, n_write16);
}

// hint: 0 none, 1 loads and stores non-temporal, 2 stores only; items: 16-byte items per work-item (kCopyProbeItems, or 1)
static hipError_t launch_copy_probe(const void* src, uint64_t n_read16, void* dst, uint64_t n_write16, int hint, int items_per_thread, hipStream_t stream)
{
    const uint64_t items = n_read16 > n_write16 ? n_read16 : n_write16;
    if (items == 0) return hipSuccess;
    const bool one = items_per_thread == 1;
    const uint64_t per_block = (uint64_t)kProbeBlock * (one ? 1 : kCopyProbeItems);
    const uint64_t grid = (items + per_block - 1) / per_block;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    const uint32_t* s = static_cast<const uint32_t*>(src);
    uint32_t* d = static_cast<uint32_t*>(dst);
    const dim3 g((uint32_t)grid), b(kProbeBlock);
#define GYMRS_COPY_LAUNCH(NTL_, NTS_)                                                                                                  \
    do {                                                                                                                               \
        if (one) hipLaunchKernelGGL((copy_probe_kernel<NTL_, NTS_, 1>), g, b, 0, stream, s, n_read16, d, n_write16);                   \
        else hipLaunchKernelGGL((copy_probe_kernel<NTL_, NTS_, kCopyProbeItems>), g, b, 0, stream, s, n_read16, d, n_write16);         \
    } while (0)
    if (hint == 1) GYMRS_COPY_LAUNCH(true, true);
    else if (hint == 2) GYMRS_COPY_LAUNCH(false, true);
    else GYMRS_COPY_LAUNCH(false, false);
#undef GYMRS_COPY_LAUNCH
    return hipGetLastError();
}



// `bytes` (a multiple of 4) of hashed 32-bit words at p: 1 MiB generated on the host, doubled on the device
static hipError_t fill_hashed_words(void* p, uint64_t bytes)
{
    const uint64_t seed_bytes = bytes < (1ull << 20) ? bytes : (1ull << 20);
    std::vector<uint32_t> host((size_t)(seed_bytes / 4));
    for (size_t i = 0; i < host.size(); ++i) {
        uint32_t h = (uint32_t)i * 2654435761u + 0x9e3779b9u;
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        h *= 3266489917u;
        h ^= h >> 16;
        host[i] = h | 1u; // (never a zero word)
    }
    hipError_t err = hipMemcpy(p, host.data(), host.size() * 4, hipMemcpyHostToDevice);
    for (uint64_t filled = host.size() * 4; err == hipSuccess && filled < bytes;) {
        const uint64_t chunk = filled < bytes - filled ? filled : bytes - filled;
        err = hipMemcpy(static_cast<char*>(p) + filled, p, chunk, hipMemcpyDeviceToDevice);
        filled += chunk;
    }
    return err;
}

extern "C" {

const char* gymrs_tool_copy_probe_error(void) { return g_probe_error.c_str(); }

/* Times `launches` back-to-back launches of a plain dwordx4 copy kernel that reads read_bytes and writes write_bytes per launch (private buffers on `device`,
 * HIP events on a private stream, after as many untimed warm-up launches) and returns the mean microseconds per launch.  With read/write sizes of one step's
 * traffic (four 16-byte items per work-item like the step kernel's tiles, IN PLACE like a step: the bytes read are the first bytes written) this is the floor a
 * step launch of that size can reach on this box (it includes the fixed cost of a dependent launch); from 1.5 GiB per launch on (1 GiB + 1 GiB: two buffers, one
 * item per work-item, the shape that streams fastest) it is the HBM bandwidth a kernel can actually get.  mode = hint (0 none, 1 non-temporal loads and stores,
 * 4 non-temporal stores only) | 8: one item per work-item instead of four | 2: the launches go through a chain of the dispatcher (acquire-only packets, one
 * release at the end: what gymrs_step_many's chains must be compared with; a step's footprint only) | 16: the source holds zeros.  Without 16 it holds hashed
 * 32-bit words: on this part lines of zeros move 3-14 % faster than any other content (profiles/r04_copy_content.log), and a step's arrays are not zeros.
 * Returns 0, or 1 (bad argument) / 2 (HIP, or the dispatcher is not available) / 4 (out of memory) with gymrs_tool_copy_probe_error(). */
int gymrs_tool_copy_probe(int device, uint64_t read_bytes, uint64_t write_bytes, uint32_t launches, int mode,
                              double* us_per_launch)
{
    if (!us_per_launch || launches == 0) return probe_fail(PROBE_EINVAL, "gymrs_tool_copy_probe: NULL output or zero launches");
    if (mode < 0 || mode > 31 || (mode & 5) == 5)
        return probe_fail(PROBE_EINVAL, "gymrs_tool_copy_probe: mode = hint (0 none, 1 loads and stores non-temporal, 4 stores only) | 2 for launches through a chain | 8 for one item per work-item | 16 for a source of zeros");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return probe_fail(PROBE_EHIP, "gymrs_tool_copy_probe: no HIP device available; this library has no CPU fallback");
    if (device < 0 || device >= n_dev) return probe_fail(PROBE_EINVAL, "gymrs_tool_copy_probe: device index out of range");
    HIP_TRY(hipSetDevice(device));
    const int non_temporal = (mode & 1) ? 1 : ((mode & 4) ? 2 : 0); // launch_copy_probe's hint
    const bool chained = (mode & 2) != 0;
    const uint64_t n_read = read_bytes / 16, n_write = write_bytes / 16;
    const bool big = (n_read + n_write) * 16 >= kCopyProbeBigBytes;
    const int items_per_thread = (big || (mode & 8)) ? 1 : kCopyProbeItems;
    if (chained && big) return probe_fail(PROBE_EINVAL, "gymrs_tool_copy_probe: the chained form is for a step's footprint (< 1.5 GiB per launch)");
    void *src = nullptr, *dst = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    AqlChain* chain = nullptr;
    std::string why;
    // At a step's footprint the copy works IN PLACE like a step does: the bytes it reads are the first bytes it writes (a step updates its
    // state where it lies and adds its outputs) -- inside a chain that is what lets the lines stay in the L2s.  The HBM figure
    // (>= 1.5 GiB per launch) copies from one buffer into another.
    const uint64_t span = (n_read > n_write ? n_read : n_write) * 16 + 256;
    hipError_t err = hipMalloc(&src, big ? n_read * 16 + 256 : span);
    if (err == hipSuccess && big) err = hipMalloc(&dst, n_write * 16 + 256);
    if (err == hipSuccess && !big) dst = src;
    // WHAT is copied matters on this part: lines of zeros move faster than anything else (a step's footprint at 2^22 CartPole lanes: 21.3 us for zeros,
    // 24.1-24.4 for 0x01 bytes, 1.0f everywhere, state-like floats or hashed words alike; 1 GiB -> 1 GiB: 6.69 vs 6.52 TB/s; profiles/r04_copy_content.log).
    // A step's arrays are not zeros, so the floor copies hashed 32-bit words; mode | 16 = the zeros every figure before round 4's last evidence set copied.
    if (err == hipSuccess) err = (mode & 16) ? hipMemset(src, 0, big ? n_read * 16 + 256 : span) : fill_hashed_words(src, big ? n_read * 16 + 256 : span);
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipEventCreate(&ev0);
    if (err == hipSuccess) err = hipEventCreate(&ev1);
    bool chain_failed = false;
    AqlKernel k;
    if (err == hipSuccess && chained) {
        // the same copy as launches of a CHAIN on a dispatcher queue of its own (acquire only, the release at the end of the chain): what a
        // chain's step has to be compared with -- the HIP-launched copy carries a release fence per launch, a chain's step does not
        chain = aql_create(device, &why);
        if (!chain || !aql_kernel(chain, (std::string(non_temporal == 1 ? "gymrs_aql_copy_probe_nt" : (non_temporal == 2 ? "gymrs_aql_copy_probe_st" : "gymrs_aql_copy_probe_pl")) + (items_per_thread == 1 ? "1" : "")).c_str(), &k)) chain_failed = true;
        if (!chain_failed) (void)aql_calibrate(chain, stream, true);
    }
    auto run = [&](uint32_t count) -> hipError_t {
        if (!chained) {
            hipError_t e2 = hipSuccess;
            for (uint32_t i = 0; i < count && e2 == hipSuccess; ++i) e2 = launch_copy_probe(src, n_read, dst, n_write, non_temporal, items_per_thread, stream);
            return e2;
        }
        const uint64_t items = n_read > n_write ? n_read : n_write;
        const uint64_t per_block = (uint64_t)kProbeBlock * items_per_thread;
        const uint32_t grid = (uint32_t)((items + per_block - 1) / per_block);
        CopyProbeKernArgs ka{static_cast<const uint32_t*>(src), n_read, static_cast<uint32_t*>(dst), n_write};
        if (k.kernarg_bytes != sizeof(ka)) {
            why = "kernel-argument segment of the copy kernel differs from what the dispatcher fills";
            chain_failed = true;
            return hipSuccess;
        }
        if (!aql_begin(chain, stream, &why)) {
            chain_failed = true;
            return hipSuccess;
        }
        bool ok = true;
        for (uint32_t i = 0; i < count && ok; ++i) ok = aql_dispatch(chain, k, grid * (uint32_t)kProbeBlock, (uint32_t)kProbeBlock, &ka, sizeof(ka), &why);
        if (!aql_end(chain, stream, &why) || !ok) chain_failed = true;
        return hipSuccess;
    };
    // warm-up: as many launches again, so that clocks and caches are where a long run has them
    if (err == hipSuccess && !chain_failed) err = run(launches + 3);
    if (err == hipSuccess) err = hipStreamSynchronize(stream);
    if (err == hipSuccess) err = hipEventRecord(ev0, stream);
    if (err == hipSuccess && !chain_failed) err = run(launches);
    if (err == hipSuccess) err = hipEventRecord(ev1, stream);
    if (err == hipSuccess) err = hipStreamSynchronize(stream);
    float ms = 0.0f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, ev0, ev1);
    if (chain) aql_destroy(chain);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (stream) (void)hipStreamDestroy(stream);
    (void)hipFree(src);
    if (dst != src) (void)hipFree(dst);
    if (err != hipSuccess) return probe_fail(err == hipErrorOutOfMemory ? PROBE_ENOMEM : PROBE_EHIP, std::string("gymrs_tool_copy_probe: ") + hipGetErrorString(err));
    if (chain_failed) return probe_fail(PROBE_EHIP, "gymrs_tool_copy_probe: the engine's own dispatcher is not available here (" + why + ")");
    *us_per_launch = (double)ms * 1e3 / (double)launches;
    return PROBE_OK;
}

} // extern "C"
