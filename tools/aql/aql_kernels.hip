// tools/aql/aql_kernels.hip -- developer experiment (VERDICT r2 "next" #5): kernels for the raw-AQL launch-boundary probe.
// Built as a stand-alone code object:  hipcc --genco --offload-arch=gfx950 -O3 aql_kernels.hip -o aql_kernels.hsaco
#include <hip/hip_runtime.h>
#include <stdint.h>

struct Args {
    float* s[4];      // "state": 4 arrays of n floats, read and written
    const uint8_t* a; // "actions": n bytes, read
    float* r;         // "reward": n floats, written
    uint8_t* d;       // "done": n bytes, written
    uint32_t* ver;    // one version word per wavefront
    uint32_t* err;    // [0] spins that ran out, [1] observed version mismatches
    uint32_t t;       // launch number: a wave may start once ver[wave] == t, and publishes t + 1
    uint32_t alu;     // dependent FMAs per lane between the loads and the stores
    uint32_t versioned;
    uint32_t pad;
#ifdef FAT_ARGS
    uint32_t fat[66]; // 264 more bytes of kernel arguments (the step kernel's StepArgs), all of them read by every wave
#endif
};

extern "C" __global__ void empty_kernel(Args) {}

typedef float f4 __attribute__((ext_vector_type(4)));

// The step kernel's memory shape: a work-item owns 4 consecutive lanes: 4 x dwordx4 + 1 dword in, 5 x dwordx4 + 1 dword out
// (17 B + 21 B per lane), non-temporal, `alu` dependent FMAs in between.  Every lane's state word 0 is incremented by exactly 1
// per launch: after K launches it must read K -- any launch that read its tile before the previous one had written it shows.
extern "C" __global__ __launch_bounds__(512) void stepish_kernel(Args a)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 512u + threadIdx.x) >> 6);
    if (a.versioned) {
        // acquire: wait until the previous launch's wave of the same tile has published (wave-uniform loop)
        uint32_t v = 0, spins = 0;
        for (;;) {
            v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&a.ver[wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (v == a.t || ++spins >= (1u << 14)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (v != a.t && (threadIdx.x & 63) == 0) atomicAdd(&a.err[0], 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // invalidates this CU's vector L1
    }
    const size_t i = (size_t)blockIdx.x * 512u + threadIdx.x;
    f4 x0 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.s[0]) + i);
    f4 x1 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.s[1]) + i);
    f4 x2 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.s[2]) + i);
    f4 x3 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.s[3]) + i);
    const uint32_t act = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(a.a) + i);
    f4 y = x1;
    for (uint32_t k = 0; k < a.alu; ++k) y = y * 0.999f + x2 * 0.001f;
#ifdef FAT_ARGS
    {
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < 66; ++k) sum += a.fat[k];
        y += (float)sum; // (the host passes zeros)
    }
#endif
#ifdef FAT_CODE
    // ~10 KB of straight-line code every wave runs through (the step kernel is about that long)
#pragma unroll
    for (int k = 0; k < FAT_CODE; ++k) y = y * (1.0f + 1e-7f * (float)(k & 15)) + x2 * 1e-9f;
#endif
    x0 += 1.0f;
    x1 = y;
    __builtin_nontemporal_store(x0, reinterpret_cast<f4*>(a.s[0]) + i);
    __builtin_nontemporal_store(x1, reinterpret_cast<f4*>(a.s[1]) + i);
    __builtin_nontemporal_store(x2, reinterpret_cast<f4*>(a.s[2]) + i);
    __builtin_nontemporal_store(x3, reinterpret_cast<f4*>(a.s[3]) + i);
    __builtin_nontemporal_store(x3 + (float)(act & 1u), reinterpret_cast<f4*>(a.r) + i);
    __builtin_nontemporal_store(act ^ 0x01010101u, reinterpret_cast<uint32_t*>(a.d) + i);
    if (a.versioned) {
        // release: this wave's stores are in L2 (vmcnt(0)), then the version word.  versioned == 1: the compiler's agent-scope
        // release (buffer_wbl2 sc1 + vmcnt(0)); 2: only vmcnt(0) -- enough when producer and consumer share an XCD (= an L2)
        if (a.versioned == 1)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((threadIdx.x & 63) == 0) __hip_atomic_store(&a.ver[wave], a.t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
