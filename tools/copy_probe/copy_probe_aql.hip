// copy_probe_aql.hip -- the copy tool's OWN gfx950 code object (hipcc --cuda-device-only): the copy kernels under plain C names plus the three kernels the
// dispatcher needs in whatever code object it loads (the chain's two ends and the self-check: gymrs_step_aql.hip under GYMRS_AQL_ENDS_ONLY).  tools/copy_probe/build.py
// compiles gym-rs_amd/csrc/gymrs_aql.hip against this object for the tool; the product library's code object holds the step kernels and no copy kernel.
#define GYMRS_AQL_ENDS_ONLY 1
#include "gymrs_step_aql.hip"

#include "copy_probe_kernels.h"

// the copy through a chain: the floor a chain's step is compared with
#define GYMRS_AQL_COPY(NAME_, NTL_, NTS_)                                                                                                       \
    extern "C" __global__ __launch_bounds__(kProbeBlock) void NAME_(const uint32_t* src, uint64_t n_read16, uint32_t* dst, uint64_t n_write16)       \
    {                                                                                                                                         \
        copy_probe_body<NTL_, NTS_, kCopyProbeItems>(src, n_read16, dst, n_write16);                                                          \
    }                                                                                                                                         \
    extern "C" __global__ __launch_bounds__(kProbeBlock) void NAME_##1(const uint32_t* src, uint64_t n_read16, uint32_t* dst, uint64_t n_write16)    \
    {                                                                                                                                         \
        copy_probe_body<NTL_, NTS_, 1>(src, n_read16, dst, n_write16); /* one item per work-item */                                           \
    }
GYMRS_AQL_COPY(gymrs_aql_copy_probe_pl, false, false) // hints: none
GYMRS_AQL_COPY(gymrs_aql_copy_probe_nt, true, true)   // loads and stores
GYMRS_AQL_COPY(gymrs_aql_copy_probe_st, false, true)  // stores only

