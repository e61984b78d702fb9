// gymrs_step_aql.hip -- the per-step kernels once more, as a STAND-ALONE gfx950 code object with plain C names
// (hipcc --cuda-device-only --no-gpu-bundle-output -> gymrs_aql_kernels.hsaco, embedded into the library by gymrs_aql.cpp).
// The engine's own AQL dispatcher (gymrs_aql.h) loads it through HSA and writes the dispatch packets itself: the HIP
// runtime puts an agent-scope RELEASE fence (an L2 write-back) on every launch, which costs a back-to-back chain of
// 2^20-lane steps 1.6-1.9 us per launch (profiles/r03_aql_probe.log); a chain of gymrs_step_many launches needs that fence
// only on its last one -- tile i is stepped by workgroup i, hence by the same XCD and the same L2, in every launch.
// Same templates, same flags, same code as the kernels HIP launches (gymrs_step_<env>.hip): same bits.
// Every flag set of the launch table at 4 lanes per work-item (8 lanes per work-item is a tuning knob and stays on HIP launches), named
//   gymrs_aql_<env>_f<AUTO_RESET | TRACK_STATS | TIME_LIMIT as a number>_t<work-items per workgroup>_<hint variant>
// (TRACK_STATS without AUTO_RESET does not exist: the launch table drops it, and so does the engine before it asks for a name).
// GYMRS_AQL_ENDS_ONLY: only the chain's two ends and the self-check -- what the dispatcher needs in WHATEVER code object it loads (tools/copy_probe
// compiles the dispatcher against a code object of its own: these three kernels plus its copy kernels).
#include "gymrs_step_impl.h"

using namespace gymrs;

#ifndef GYMRS_AQL_ENDS_ONLY

#define GYMRS_AQL_STEP(NAME_, ENV_, FLAGS_, THREADS_)                                                                                     \
    extern "C" GYMRS_STEP_KERNEL_ATTRS(THREADS_, 4) void NAME_(float* s0, float* s1, float* s2, float* s3, const void* action,           \
                                                              uint64_t n_fast, const StepArgs rest, const ENV_::Consts c)               \
    {                                                                                                                                     \
        step_kernel_body<ENV_, 4, FLAGS_, THREADS_>(s0, s1, s2, s3, action, n_fast, rest, c);                                             \
    }

// hint variants: _nt every access, _o only the stores nobody reads again, _so those plus the state loads, _pl none
#define GYMRS_AQL_STEP_HINTS(PREFIX_, ENV_, FLAGS_, THREADS_)                     \
    GYMRS_AQL_STEP(PREFIX_##_nt, ENV_, (FLAGS_) | kFlagNonTemporal, THREADS_)     \
    GYMRS_AQL_STEP(PREFIX_##_o, ENV_, (FLAGS_) | kFlagNtOut, THREADS_)            \
    GYMRS_AQL_STEP(PREFIX_##_so, ENV_, (FLAGS_) | kFlagNtOut | kFlagNtStateLoads, THREADS_) \
    GYMRS_AQL_STEP(PREFIX_##_pl, ENV_, (FLAGS_), THREADS_)
static_assert(GYMRS_AUTO_RESET == 1u && GYMRS_TRACK_STATS == 2u && GYMRS_TIME_LIMIT == 4u, "the f<N> of the kernel names is these three bits");
#define GYMRS_AQL_STEP_FLAGSETS(ENV_NAME_, ENV_, THREADS_)                               \
    GYMRS_AQL_STEP_HINTS(gymrs_aql_##ENV_NAME_##_f0_t##THREADS_, ENV_, 0u, THREADS_)     \
    GYMRS_AQL_STEP_HINTS(gymrs_aql_##ENV_NAME_##_f1_t##THREADS_, ENV_, 1u, THREADS_)     \
    GYMRS_AQL_STEP_HINTS(gymrs_aql_##ENV_NAME_##_f3_t##THREADS_, ENV_, 3u, THREADS_)     \
    GYMRS_AQL_STEP_HINTS(gymrs_aql_##ENV_NAME_##_f4_t##THREADS_, ENV_, 4u, THREADS_)     \
    GYMRS_AQL_STEP_HINTS(gymrs_aql_##ENV_NAME_##_f5_t##THREADS_, ENV_, 5u, THREADS_)     \
    GYMRS_AQL_STEP_HINTS(gymrs_aql_##ENV_NAME_##_f7_t##THREADS_, ENV_, 7u, THREADS_)
GYMRS_AQL_STEP_FLAGSETS(cartpole, CartPoleT, 512)
GYMRS_AQL_STEP_FLAGSETS(cartpole, CartPoleT, 256)
GYMRS_AQL_STEP_FLAGSETS(mountain_car, MountainCarT, 256)
GYMRS_AQL_STEP_FLAGSETS(pendulum, PendulumT, 256)
#endif // GYMRS_AQL_ENDS_ONLY

// ---- the two ends of a chain: ordering against the engine's HIP stream -------------------------------------------------
// First packet of a chain: one wavefront waits until the HIP stream has reached the hipStreamWriteValue32 the engine put
// behind everything that was enqueued there before (flag >= seq, wrap-around safe).  Bounded: a stream that never gets
// there (blocked on work nobody submits) must not hang the queue -- the wait gives up after max_ticks (10 s), sets err[0] and
// lets the chain run (gymrs_sync reports it).
extern "C" __global__ __launch_bounds__(64) void gymrs_aql_wait_flag(const uint32_t* flag, uint32_t seq, uint32_t* err, unsigned long long max_ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
    for (;;) {
        const uint32_t v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(v - seq) >= 0) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            return;
        }
        if (__builtin_amdgcn_s_memrealtime() - t0 > max_ticks) break;
        __builtin_amdgcn_s_sleep(32);
    }
    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Last packet of a chain (dispatched with a system-scope release: everything the chain wrote is written back first): tells
// the HIP stream, which waits there with hipStreamWaitValue32, that the chain is done.
extern "C" __global__ __launch_bounds__(64) void gymrs_aql_set_flag(uint32_t* flag, uint32_t seq)
{
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- self-check of the assumption the fence-free chain rests on ---------------------------------------------------------
// Every work-item adds 1 to its own 16 bytes; launched as a chain WITHOUT release fences, with one-workgroup launches in
// between (they must not shift which XCD gets which workgroup) -- after K launches every word must read K.
// And the assumption itself, read from the hardware: every workgroup notes the XCC it runs on (HW_REG_XCC_ID) in the first
// launch and compares in every later one; a workgroup index that moved to another XCD marks its slot of `moved`.
extern "C" __global__ __launch_bounds__(256) void gymrs_aql_selfcheck(float* x, uint32_t n4, uint32_t first, uint32_t* xcc, uint32_t* moved)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (xcc && threadIdx.x == 0) {
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        id = (id & 0xfu) + 1u; // (0 = never written)
        if (first)
            xcc[blockIdx.x] = id;
        else if (xcc[blockIdx.x] != id)
            moved[blockIdx.x] = 1u;
    }
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n4) return;
    // plain accesses, like the state arrays of the real chains: the lines stay dirty in this XCD's L2 until the next launch reads
    // them there (streamed lines would leave the L2 early -- an easier case than the one the chains rest on)
    f4 v = reinterpret_cast<const f4*>(x)[i];
    v += 1.0f;
    reinterpret_cast<f4*>(x)[i] = v;
}
