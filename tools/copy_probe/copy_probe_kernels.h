// copy_probe_kernels.h -- the copy yardstick's kernel body and constants (tools/copy_probe: a MEASUREMENT tool, not part of the C ABI;
// moved out of the product's headers in round 5, VERDICT r4 "next" #8).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace gymrs {

constexpr int kProbeBlock = 256; // work-items per workgroup, like the step kernels' kBlock
// reads n_read16 and writes n_write16 16-byte items; kCopyProbeItems items per work-item or one (the caller's choice;
// always one from kCopyProbeBigBytes per launch on: copy_probe.hip says why)
constexpr int kCopyProbeItems = 4;
constexpr uint64_t kCopyProbeBigBytes = 1536ull << 20;
struct CopyProbeKernArgs { // the kernel-argument segment of the copy kernels as the dispatcher fills it
    const uint32_t* src;
    uint64_t n_read16;
    uint32_t* dst;
    uint64_t n_write16;
};

// The copy probe's kernel body (launched through HIP from copy_probe.hip and, as gymrs_aql_copy_probe_*, through the library's
// dispatcher source compiled against this tool's own code object): a work-item moves ITEMS 16-byte items, its loads all in flight before its first store; a workgroup a
// contiguous chunk of 256 * ITEMS items.
template <bool NTL, bool NTS, int ITEMS>
__device__ __forceinline__ void copy_probe_body(const uint32_t* src, uint64_t n_read16, uint32_t* dst, uint64_t n_write16) // (src and dst may alias: in place)
{
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const uint64_t first = (uint64_t)blockIdx.x * (kProbeBlock * ITEMS) + threadIdx.x;
    u4 v[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint64_t i = first + (uint64_t)j * kProbeBlock;
        v[j] = u4{(uint32_t)i, 1u, 2u, 3u};
        if (i < n_read16) v[j] = NTL ? __builtin_nontemporal_load(reinterpret_cast<const u4*>(src) + i) : reinterpret_cast<const u4*>(src)[i];
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint64_t i = first + (uint64_t)j * kProbeBlock;
        if (i < n_write16) {
            if (NTS)
                __builtin_nontemporal_store(v[j], reinterpret_cast<u4*>(dst) + i);
            else
                reinterpret_cast<u4*>(dst)[i] = v[j];
        } else if (v[j].x == 0xdeadbeefu && v[j].y == 0x12345678u) { // keeps the load alive when nothing is written for this item
            dst[0] = v[j].z;
        }
    }
}


} // namespace gymrs
