// tools/aql/visible_kernels.hip -- developer experiment (VERDICT r3 "next" #2): can a step be made VISIBLE to a reader on
// another XCD without the end-of-kernel release fence (the L2 write-back HIP puts on every launch)?
// A "step" kernel of the real step's memory shape (17 B read + 21 B written per lane) whose OUTPUT stores carry a cache policy
// (plain / nt / sc1 = written through at agent scope / sc0 sc1 = system scope / sc1 nt), and a "reader" (the policy between two
// steps) that runs with a PERMUTED mapping -- work-item i reads the tile of work-item n4 - 1 - i, which the step's workgroup
// G - 1 - b wrote: with G = 512 workgroups that is always another XCD -- checks every value the step of launch t must have
// produced, and writes the next step's actions, which the step in turn checks.  A stale read on either side is COUNTED.
// Built as a stand-alone code object:  hipcc --cuda-device-only --no-gpu-bundle-output --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdint.h>

struct VArgs {
    float* s[4];      // "state": 4 arrays of n floats, read and written by the step
    uint8_t* a;       // "actions": n bytes, read by the step, written by the reader
    float* r;         // "reward": n floats, written by the step
    uint8_t* d;       // "done": n bytes, written by the step
    uint32_t* err;    // [0] stale values seen by the reader, [1] stale actions seen by the step (device memory, atomics)
    uint32_t t;       // launch number (of the pair)
    uint32_t alu;     // dependent FMAs per lane between the loads and the stores
    uint32_t n4;      // work-items (= lanes / 4)
    uint32_t check;   // 0 = no reader in this run: the step does not check its actions
};

typedef float f4 __attribute__((ext_vector_type(4)));

// cache policy of a store: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt
template <int P>
__device__ __forceinline__ void st16(f4* p, f4 v)
{
    if constexpr (P == 0) *p = v;
    else if constexpr (P == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (P == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <int P>
__device__ __forceinline__ void st4(uint32_t* p, uint32_t v)
{
    if constexpr (P == 0) *p = v;
    else if constexpr (P == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (P == 2) asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ uint32_t action_pattern(uint32_t t) { return 0x01010101u * (t & 1u); }

// LD = state loads non-temporal (what the chain's kernels do at this size)
template <int P, bool LD>
__device__ __forceinline__ void step_body(const VArgs& a)
{
    const uint32_t i = blockIdx.x * 512u + threadIdx.x;
    if (i >= a.n4) return;
    f4 x0, x1, x2, x3;
    if (LD) {
        x0 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.s[0]) + i);
        x1 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.s[1]) + i);
        x2 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.s[2]) + i);
        x3 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.s[3]) + i);
    } else {
        x0 = reinterpret_cast<const f4*>(a.s[0])[i];
        x1 = reinterpret_cast<const f4*>(a.s[1])[i];
        x2 = reinterpret_cast<const f4*>(a.s[2])[i];
        x3 = reinterpret_cast<const f4*>(a.s[3])[i];
    }
    const uint32_t act = reinterpret_cast<const uint32_t*>(a.a)[i];
    if (a.check && act != action_pattern(a.t)) atomicAdd(&a.err[1], 1u); // the reader of launch t - 1 wrote it
    f4 y = x1;
    for (uint32_t k = 0; k < a.alu; ++k) y = y * 0.999f + x2 * 0.001f;
    x0 += 1.0f;
    x1 = y;
    st16<P>(reinterpret_cast<f4*>(a.s[0]) + i, x0);
    st16<P>(reinterpret_cast<f4*>(a.s[1]) + i, x1);
    st16<P>(reinterpret_cast<f4*>(a.s[2]) + i, x2);
    st16<P>(reinterpret_cast<f4*>(a.s[3]) + i, x3);
    st16<P>(reinterpret_cast<f4*>(a.r) + i, f4{(float)a.t, (float)a.t, (float)a.t, (float)a.t});
    st4<P>(reinterpret_cast<uint32_t*>(a.d) + i, a.t * 0x9e3779b9u + i);
    if constexpr (P >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (asm stores are not in the compiler's count; s_endpgm would do)
}

// the reader: permuted mapping, checks launch t's outputs, writes launch t + 1's actions
template <int P>
__device__ __forceinline__ void reader_body(const VArgs& a)
{
    const uint32_t i = blockIdx.x * 512u + threadIdx.x;
    if (i >= a.n4) return;
    const uint32_t j = a.n4 - 1u - i;
    const f4 x0 = reinterpret_cast<const f4*>(a.s[0])[j];
    const f4 x1 = reinterpret_cast<const f4*>(a.s[1])[j];
    const f4 x2 = reinterpret_cast<const f4*>(a.s[2])[j];
    const f4 x3 = reinterpret_cast<const f4*>(a.s[3])[j];
    const f4 r = reinterpret_cast<const f4*>(a.r)[j];
    const uint32_t d = reinterpret_cast<const uint32_t*>(a.d)[j];
    const float want = (float)(a.t + 1u), tf = (float)a.t;
    uint32_t bad = 0;
    bad += x0.x != want || x0.y != want || x0.z != want || x0.w != want;
    bad += r.x != tf || r.y != tf || r.z != tf || r.w != tf;
    bad += d != a.t * 0x9e3779b9u + j;
    bad += (x1.x + x2.x + x3.x) == 12345.678f; // (keeps the other state loads alive; the arrays hold zeros)
    if (bad) atomicAdd(&a.err[0], bad);
    st4<P>(reinterpret_cast<uint32_t*>(a.a) + j, action_pattern(a.t + 1u));
    if constexpr (P >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#define STEP_KERNEL(NAME_, P_, LD_) extern "C" __global__ __launch_bounds__(512) void NAME_(VArgs a) { step_body<P_, LD_>(a); }
#define READER_KERNEL(NAME_, P_) extern "C" __global__ __launch_bounds__(512) void NAME_(VArgs a) { reader_body<P_>(a); }
STEP_KERNEL(vstep_plain, 0, false)
STEP_KERNEL(vstep_nt, 1, false)
STEP_KERNEL(vstep_sc1, 2, false)
STEP_KERNEL(vstep_sc0sc1, 3, false)
STEP_KERNEL(vstep_sc1nt, 4, false)
STEP_KERNEL(vstep_plain_ldnt, 0, true)
STEP_KERNEL(vstep_nt_ldnt, 1, true)
STEP_KERNEL(vstep_sc1_ldnt, 2, true)
STEP_KERNEL(vstep_sc0sc1_ldnt, 3, true)
READER_KERNEL(vreader_plain, 0)
READER_KERNEL(vreader_sc1, 2)
READER_KERNEL(vreader_sc0sc1, 3)
