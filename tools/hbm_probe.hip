// tools/hbm_probe.hip -- developer experiment (VERDICT r3 "next" #4): what does a pure memory kernel get out of HBM on this box,
// (a) as a plain 1 GiB -> 1 GiB copy (the guide measured 6.29 TB/s for a float4 copy; the in-library probe reads 5.97), and
// (b) in the STEP's access pattern at 2^24 CartPole lanes: 4 state arrays read and written (dwordx4 per work-item), actions read
// and done written as packed dwords, reward written -- 17 B read + 21 B written per lane, nine streams.
// Knobs: grid shape (one tile per workgroup vs persistent grid-stride), tiles in flight per work-item, workgroup size,
// non-temporal hints on loads / stores separately, the skew between the array bases.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/hbm_probe.hip -o tools/hbm_probe && tools/hbm_probe [lanes_log2=24]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#define HIP_OK(x)                                                                          \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ u4 ld16(const u4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void st16(u4* p, u4 v)
{
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <bool NT>
__device__ __forceinline__ uint32_t ld4(const uint32_t* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void st4(uint32_t* p, uint32_t v)
{
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// ---- (a) plain copy: ITEMS 16-byte items per work-item per iteration, all loads before the first store; persistent when the grid is
// smaller than the work (grid-stride over chunks of THREADS * ITEMS items)
template <int THREADS, int ITEMS, bool NTL, bool NTS>
__global__ __launch_bounds__(THREADS) void copy_kernel(const u4* __restrict__ src, u4* __restrict__ dst, uint64_t n16)
{
    const uint64_t chunk = (uint64_t)THREADS * ITEMS;
    for (uint64_t c = blockIdx.x; c * chunk < n16; c += gridDim.x) {
        const uint64_t first = c * chunk + threadIdx.x;
        u4 v[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint64_t i = first + (uint64_t)j * THREADS;
            if (i < n16) v[j] = ld16<NTL>(src + i);
        }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint64_t i = first + (uint64_t)j * THREADS;
            if (i < n16) st16<NTS>(dst + i, v[j]);
        }
    }
}

// (d) what the bytes ARE: fill a buffer with zeros / one byte value / hashed 32-bit words / small floats (a CartPole state: |x| < 0.05..2.4, mixed signs)
__global__ void fill_kernel(uint32_t* p, uint64_t n32, int kind)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        uint32_t v = 0;
        if (kind == 1) v = 0x01010101u;
        else if (kind == 2) v = 0xffffffffu;
        else if (kind == 3) v = h;
        else if (kind == 4) v = __builtin_bit_cast(uint32_t, ((float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.1f);   // uniform in [-0.05, 0.05)
        else if (kind == 5) v = __builtin_bit_cast(uint32_t, ((float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 4.8f);   // uniform in [-2.4, 2.4)
        else if (kind == 6) v = 0x3f800000u;                                                                             // 1.0f everywhere
        else if (kind == 7) v = (h & 0x01010101u);                                                                       // random 0/1 bytes (actions, flags)
        p[i] = v;
    }
}
static const char* fill_name(int kind)
{
    static const char* n[8] = {"zeros", "0x01 bytes", "0xff bytes", "hashed 32-bit words", "floats in [-0.05, 0.05)", "floats in [-2.4, 2.4)", "1.0f everywhere", "random 0/1 bytes"};
    return n[kind];
}

// ---- (b) the step's nine streams.  One "tile" of a work-item = 4 consecutive lanes: 4 x dwordx4 + 1 dword in, 5 x dwordx4 + 1 dword out.
struct Streams {
    const u4* s_in[4];
    u4* s_out[4]; // = s_in (in place, as the step does) or separate buffers
    const uint32_t* act;
    u4* reward;
    uint32_t* done;
    uint64_t n4; // work-item tiles
    uint32_t stag_units, stag_wgs; // (c): workgroup b < stag_wgs starts (b >> 8) * stag_units * 0.21 us late (the k-th workgroup a CU receives: k units)
    uint32_t delay_units;          // (c): after a tile's loads have landed the wave sleeps delay_units * 0.21 us (no VALU work) before it stores
    uint32_t mode;                 // (c): 0 loads and stores, 1 loads only (the stores sit behind a test that never holds), 2 stores only
};

// TILES tiles per work-item in flight (all their loads issued before the first store); persistent grid-stride over groups of tiles.
// ALU (round 4, the mid-size question): ALU rounds of 16 independent FMAs per work-item between a tile's loads and its stores -- the real step
// issues ~380 VALU instructions per wave there (ALU = 24) and holds its bytes for ~3 us; LDS_KB pads the workgroup's LDS like ResetLds does.
// AOS: the four state columns as ONE array of a float4 per LANE -- a work-item's 4 lanes are 64 contiguous bytes of s_in[0] (s_in[1..3] unused)
template <int THREADS, int TILES, bool NTL, bool NTS, int ALU = 0, int LDS_KB = 0, bool AOS = false>
__global__ __launch_bounds__(THREADS) void stream_kernel(Streams a)
{
    __shared__ uint32_t pad_lds[LDS_KB > 0 ? LDS_KB * 256 : 1];
    if (LDS_KB > 0 && a.n4 == 0xffffffffffull) pad_lds[threadIdx.x] = 1; // (never true: keeps the allocation)
    if (a.stag_units && blockIdx.x < a.stag_wgs)
        for (uint32_t i = 0; i < (blockIdx.x >> 8) * a.stag_units; ++i) __builtin_amdgcn_s_sleep(8);
    const uint64_t chunk = (uint64_t)THREADS * TILES;
    for (uint64_t c = blockIdx.x; c * chunk < a.n4; c += gridDim.x) {
        const uint64_t first = c * chunk + threadIdx.x;
        u4 v[TILES][4];
        uint32_t act[TILES];
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const uint64_t i = first + (uint64_t)t * THREADS;
            if (i < a.n4) {
                if (a.mode == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[t][j] = u4{(uint32_t)i, 1u, 2u, 3u};
                    act[t] = (uint32_t)i;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[t][j] = AOS ? ld16<NTL>(a.s_in[0] + 4 * i + j) : ld16<NTL>(a.s_in[j] + i);
                    act[t] = ld4<NTL>(a.act + i);
                }
            }
        }
        if (a.delay_units) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (uint32_t i = 0; i < a.delay_units; ++i) __builtin_amdgcn_s_sleep(8);
        }
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const uint64_t i = first + (uint64_t)t * THREADS;
            if (i < a.n4) {
                v[t][0].x += 1u;
                if constexpr (ALU > 0) {
                    float f[16];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f[4 * j + 0] = __builtin_bit_cast(float, v[t][j].x);
                        f[4 * j + 1] = __builtin_bit_cast(float, v[t][j].y);
                        f[4 * j + 2] = __builtin_bit_cast(float, v[t][j].z);
                        f[4 * j + 3] = __builtin_bit_cast(float, v[t][j].w);
                    }
                    for (int r = 0; r < ALU; ++r) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) f[q] = __builtin_fmaf(f[q], 1.0000001f, 1e-30f);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[t][j].y = __builtin_bit_cast(uint32_t, f[4 * j + 1]);
                        v[t][j].z = __builtin_bit_cast(uint32_t, f[4 * j + 2]);
                        v[t][j].w = __builtin_bit_cast(uint32_t, f[4 * j + 3]);
                        if (j) v[t][j].x = __builtin_bit_cast(uint32_t, f[4 * j]);
                    }
                }
                if (a.mode == 1 && (v[t][0].x ^ v[t][1].y ^ v[t][2].z ^ v[t][3].w ^ act[t]) != 0x9e3779b9u) continue; // loads only (the pool is zero-filled)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (AOS) st16<NTS>(a.s_out[0] + 4 * i + j, v[t][j]);
                    else st16<NTS>(a.s_out[j] + i, v[t][j]);
                }
                st16<NTS>(a.reward + i, v[t][3]);
                st4<NTS>(a.done + i, act[t] ^ 0x01010101u);
            }
        }
    }
}

static double time_launches(hipStream_t st, int launches, const auto& launch)
{
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    HIP_OK(hipStreamSynchronize(st));
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        HIP_OK(hipEventRecord(e0, st));
        for (int i = 0; i < launches; ++i) launch();
        HIP_OK(hipEventRecord(e1, st));
        HIP_OK(hipStreamSynchronize(st));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / launches;
        best = us < best ? us : best;
    }
    HIP_OK(hipEventDestroy(e0));
    HIP_OK(hipEventDestroy(e1));
    return best;
}

int main(int argc, char** argv)
{
    const int lanes_log2 = argc > 1 ? std::atoi(argv[1]) : 24;
    const bool phase_only = argc > 2 && !std::strcmp(argv[2], "phase"); // only part (c)
    const bool data_only = argc > 2 && !std::strcmp(argv[2], "data");   // only part (d)
    hipStream_t st;
    HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (data_only) {
        // ---------------- (d) does the CONTENT of the bytes matter? ----------------
        std::printf("# (d) the same copies over different CONTENT (every fill is followed by the same launches; the order A B A ... excludes time under load)\n");
        std::printf("%-96s %10s %10s\n", "variant", "us", "GB/s");
        {
            const uint64_t bytes = 1ull << 30, n16 = bytes / 16;
            u4 *src, *dst;
            HIP_OK(hipMalloc(&src, bytes + 4096));
            HIP_OK(hipMalloc(&dst, bytes + 4096));
            for (int kind : {0, 3, 0, 4, 1, 5, 6, 7, 2, 3, 0}) {
                hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, (uint32_t*)src, bytes / 4, kind);
                hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, (uint32_t*)dst, bytes / 4, 0);
                HIP_OK(hipStreamSynchronize(st));
                const uint32_t grid = (uint32_t)((n16 + 255) / 256);
                const double us = time_launches(st, 10, [&] { hipLaunchKernelGGL((copy_kernel<256, 1, true, true>), dim3(grid), dim3(256), 0, st, src, dst, n16); });
                char l[128];
                std::snprintf(l, sizeof(l), "1 GiB -> 1 GiB, one item per work-item, hinted | %s", fill_name(kind));
                std::printf("%-96s %10.2f %10.0f\n", l, us, 2.0 * bytes / (us * 1e-6) / 1e9);
                std::fflush(stdout);
            }
            HIP_OK(hipFree(src));
            HIP_OK(hipFree(dst));
        }
        {
            const uint64_t n = 1ull << lanes_log2, n4 = n / 4;
            const double alg = (double)n * 38.0;
            const size_t arr = n * 4, skew = 4352;
            char* pool;
            const size_t pool_bytes = arr * 9 + n * 2 + skew * 16 + (1 << 20);
            HIP_OK(hipMalloc(&pool, pool_bytes));
            Streams a{};
            size_t off = 0;
            int k = 0;
            auto carve = [&](size_t bytes) {
                char* p = pool + off + (size_t)(++k) * skew;
                off += (bytes + 4095) / 4096 * 4096;
                return p;
            };
            for (int j = 0; j < 4; ++j) { a.s_in[j] = (const u4*)carve(arr); a.s_out[j] = (u4*)a.s_in[j]; }
            a.reward = (u4*)carve(arr);
            a.act = (const uint32_t*)carve(n);
            a.done = (uint32_t*)carve(n);
            a.n4 = n4;
            const uint64_t full = (n4 + 511) / 512;
            for (int kind : {0, 3, 0, 4, 5, 1, 6, 7, 2, 3, 0}) {
                hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, (uint32_t*)pool, pool_bytes / 4, kind);
                HIP_OK(hipStreamSynchronize(st));
                for (int alu : {0, 24}) {
                    const double us = alu == 0
                        ? time_launches(st, 20, [&] { hipLaunchKernelGGL((stream_kernel<512, 1, false, true, 0, 0>), dim3((uint32_t)full), dim3(512), 0, st, a); })
                        : time_launches(st, 20, [&] { hipLaunchKernelGGL((stream_kernel<512, 1, false, true, 24, 0>), dim3((uint32_t)full), dim3(512), 0, st, a); });
                    char l[128];
                    std::snprintf(l, sizeof(l), "the step's nine streams at 2^%d lanes, in place, %2d FMA rounds | %s%s", lanes_log2, alu, fill_name(kind),
                                  alu ? " (then what the FMAs leave)" : "");
                    std::printf("%-96s %10.2f %10.0f\n", l, us, alg / (us * 1e-6) / 1e9);
                    std::fflush(stdout);
                }
            }
            // (e) the four state columns as one float4-per-lane array (s_in[0] spans 4 arrays' worth: the pool is contiguous), hashed words, 0 / 24 FMA rounds
            hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, (uint32_t*)pool, pool_bytes / 4, 3);
            HIP_OK(hipStreamSynchronize(st));
            for (int rep = 0; rep < 2; ++rep)
                for (int aos = 0; aos < 2; ++aos)
                    for (int alu : {0, 24}) {
                        double us;
                        if (aos)
                            us = alu ? time_launches(st, 20, [&] { hipLaunchKernelGGL((stream_kernel<512, 1, false, true, 24, 0, true>), dim3((uint32_t)full), dim3(512), 0, st, a); })
                                     : time_launches(st, 20, [&] { hipLaunchKernelGGL((stream_kernel<512, 1, false, true, 0, 0, true>), dim3((uint32_t)full), dim3(512), 0, st, a); });
                        else
                            us = alu ? time_launches(st, 20, [&] { hipLaunchKernelGGL((stream_kernel<512, 1, false, true, 24, 0>), dim3((uint32_t)full), dim3(512), 0, st, a); })
                                     : time_launches(st, 20, [&] { hipLaunchKernelGGL((stream_kernel<512, 1, false, true, 0, 0>), dim3((uint32_t)full), dim3(512), 0, st, a); });
                        char l[128];
                        std::snprintf(l, sizeof(l), "2^%d lanes, in place, %2d FMA rounds, hashed words | state as %s", lanes_log2, alu,
                                      aos ? "ONE array of a float4 per lane (6 streams)" : "four columns (9 streams)");
                        std::printf("%-96s %10.2f %10.0f\n", l, us, alg / (us * 1e-6) / 1e9);
                        std::fflush(stdout);
                    }
            HIP_OK(hipFree(pool));
        }
        return 0;
    }
    // ---------------- (a) 1 GiB -> 1 GiB ----------------
    if (!phase_only) {
        const uint64_t bytes = 1ull << 30, n16 = bytes / 16;
        u4 *src, *dst;
        HIP_OK(hipMalloc(&src, bytes + 4096));
        HIP_OK(hipMalloc(&dst, bytes + 4096));
        HIP_OK(hipMemset(src, 1, bytes));
        HIP_OK(hipMemset(dst, 0, bytes));
        std::printf("# (a) copy 1 GiB read + 1 GiB written per launch; GB/s = 2 GiB / time\n");
        std::printf("%-64s %10s %10s\n", "variant", "us", "GB/s");
        auto report = [&](const char* label, double us) {
            std::printf("%-64s %10.2f %10.0f\n", label, us, 2.0 * bytes / (us * 1e-6) / 1e9);
            std::fflush(stdout);
        };
#define COPY_VARIANT(THREADS_, ITEMS_, NTL_, NTS_, GRID_, LABEL_)                                                                       \
    {                                                                                                                                   \
        const uint64_t full = (n16 + (uint64_t)THREADS_ * ITEMS_ - 1) / ((uint64_t)THREADS_ * ITEMS_);                                 \
        const uint32_t grid = (uint32_t)((GRID_) == 0 ? full : (GRID_));                                                               \
        char l[128];                                                                                                                    \
        std::snprintf(l, sizeof(l), "%s: %d thr x %d items, nt loads %d stores %d, grid %u", LABEL_, THREADS_, ITEMS_, NTL_, NTS_, grid); \
        report(l, time_launches(st, 10, [&] {                                                                                          \
                   hipLaunchKernelGGL((copy_kernel<THREADS_, ITEMS_, NTL_, NTS_>), dim3(grid), dim3(THREADS_), 0, st, src, dst, n16);  \
               }));                                                                                                                     \
    }
        COPY_VARIANT(256, 4, true, true, 0, "library probe shape (nt)")
        COPY_VARIANT(256, 4, false, false, 0, "library probe shape (plain)")
        COPY_VARIANT(256, 4, true, false, 0, "one chunk per WG")
        COPY_VARIANT(256, 4, false, true, 0, "one chunk per WG")
        COPY_VARIANT(256, 1, true, true, 0, "one chunk per WG")
        COPY_VARIANT(256, 2, true, true, 0, "one chunk per WG")
        COPY_VARIANT(256, 8, true, true, 0, "one chunk per WG")
        COPY_VARIANT(512, 4, true, true, 0, "one chunk per WG")
        COPY_VARIANT(1024, 4, true, true, 0, "one chunk per WG")
        COPY_VARIANT(256, 4, true, true, 256 * 4, "persistent")
        COPY_VARIANT(256, 4, true, true, 256 * 8, "persistent")
        COPY_VARIANT(256, 4, true, true, 256 * 16, "persistent")
        COPY_VARIANT(256, 8, true, true, 256 * 4, "persistent")
        COPY_VARIANT(256, 8, true, true, 256 * 8, "persistent")
        COPY_VARIANT(512, 4, true, true, 256 * 4, "persistent")
        COPY_VARIANT(512, 8, true, true, 256 * 2, "persistent")
        COPY_VARIANT(512, 8, true, true, 256 * 4, "persistent")
        COPY_VARIANT(1024, 4, true, true, 256 * 2, "persistent")
        COPY_VARIANT(256, 4, false, false, 256 * 8, "persistent (plain)")
        COPY_VARIANT(256, 8, false, false, 256 * 8, "persistent (plain)")
        COPY_VARIANT(256, 4, true, false, 256 * 8, "persistent")
        COPY_VARIANT(256, 4, false, true, 256 * 8, "persistent")
        // hipMemcpyAsync device-to-device for reference (the runtime's own blit kernel)
        report("hipMemcpyAsync D2D", time_launches(st, 10, [&] { HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st)); }));
        HIP_OK(hipFree(src));
        HIP_OK(hipFree(dst));
    }
    // ---------------- (b) the step's streams ----------------
    {
        const uint64_t n = 1ull << lanes_log2, n4 = n / 4;
        const double alg = (double)n * 38.0;
        std::printf("# (b) the step's nine streams at 2^%d lanes: %.1f MB per launch (17 B read + 21 B written per lane)\n", lanes_log2, alg / 1e6);
        std::printf("%-84s %10s %10s\n", "variant", "us", "GB/s");
        for (uint64_t skew : {0ull, 4352ull, 256ull * 37, 256ull * 1021, 4096ull * 3 + 256}) {
            for (int inplace = 1; inplace >= 0; --inplace) {
                if (!inplace && skew != 4352) continue;
                if (phase_only && !(inplace && skew == 4352)) continue;
                // one pool, arrays carved with base k * skew apart from their power-of-two spacing (the engine: multiples of 4352 B)
                const size_t arr = n * 4;
                char* pool;
                const size_t pool_bytes = arr * 9 + n * 2 + skew * 16 + (1 << 20);
                HIP_OK(hipMalloc(&pool, pool_bytes));
                HIP_OK(hipMemset(pool, 0, pool_bytes));
                Streams a{};
                size_t off = 0;
                int k = 0;
                auto carve = [&](size_t bytes) {
                    char* p = pool + off + (size_t)(++k) * skew;
                    off += (bytes + 4095) / 4096 * 4096;
                    return p;
                };
                for (int j = 0; j < 4; ++j) a.s_in[j] = (const u4*)carve(arr);
                for (int j = 0; j < 4; ++j) a.s_out[j] = inplace ? (u4*)a.s_in[j] : (u4*)carve(arr);
                a.reward = (u4*)carve(arr);
                a.act = (const uint32_t*)carve(n);
                a.done = (uint32_t*)carve(n);
                a.n4 = n4;
                auto report = [&](const char* label, double us) {
                    char l[160];
                    std::snprintf(l, sizeof(l), "skew %6llu B %s | %s", (unsigned long long)skew, inplace ? "in place" : "out of place", label);
                    std::printf("%-84s %10.2f %10.0f\n", l, us, alg / (us * 1e-6) / 1e9);
                    std::fflush(stdout);
                };
#define STREAM_VARIANT(THREADS_, TILES_, NTL_, NTS_, GRID_)                                                                               \
    {                                                                                                                                     \
        const uint64_t full = (n4 + (uint64_t)THREADS_ * TILES_ - 1) / ((uint64_t)THREADS_ * TILES_);                                    \
        const uint32_t grid = (uint32_t)((GRID_) == 0 ? full : (GRID_));                                                                 \
        char l[96];                                                                                                                       \
        std::snprintf(l, sizeof(l), "%d thr x %d tiles, nt loads %d stores %d, grid %u%s", THREADS_, TILES_, NTL_, NTS_, grid,           \
                      (GRID_) == 0 ? "" : " (persistent)");                                                                               \
        report(l, time_launches(st, 20, [&] {                                                                                            \
                   hipLaunchKernelGGL((stream_kernel<THREADS_, TILES_, NTL_, NTS_>), dim3(grid), dim3(THREADS_), 0, st, a);              \
               }));                                                                                                                       \
    }
                if (phase_only) {
                    // ---------------- (c) is it the arithmetic, or WHEN the waves do what? ----------------
                    // A launch of several generations of waves starts its first generation all at once: every resident wave loads, then computes, then
                    // stores at the same time, and the next generation inherits the alignment (a slot frees when its workgroup ends).  (i) a pure delay
                    // between loads and stores (no VALU work at all) against the same time spent in FMAs; (ii) the first generation staggered by slot.
#define PHASE_VARIANT(THREADS_, ALU_, STAG_, DELAY_)                                                                                      \
    {                                                                                                                                     \
        const uint64_t full = (n4 + (uint64_t)THREADS_ - 1) / (uint64_t)THREADS_;                                                        \
        Streams b = a;                                                                                                                    \
        b.stag_units = (STAG_);                                                                                                           \
        b.stag_wgs = 256u * (THREADS_ == 512 ? 4 : 8);                                                                                    \
        b.delay_units = (DELAY_);                                                                                                         \
        char l[96];                                                                                                                       \
        std::snprintf(l, sizeof(l), "%d thr, %d FMA rounds, sleep %.2f us, stagger %.2f us per slot", THREADS_, ALU_, (DELAY_) * 0.213, (STAG_) * 0.213); \
        report(l, time_launches(st, 20, [&] {                                                                                            \
                   hipLaunchKernelGGL((stream_kernel<THREADS_, 1, false, true, ALU_, 0>), dim3((uint32_t)full), dim3(THREADS_), 0, st, b); \
               }));                                                                                                                       \
    }
                    PHASE_VARIANT(512, 0, 0, 0)
                    PHASE_VARIANT(512, 24, 0, 0)
                    for (int d : {1, 2, 4, 8, 16}) PHASE_VARIANT(512, 0, 0, d)
                    for (int g : {1, 2, 4, 8, 16}) PHASE_VARIANT(512, 24, g, 0)
                    for (int g : {2, 8}) PHASE_VARIANT(512, 0, g, 0)
                    for (int g : {2, 8}) PHASE_VARIANT(512, 0, g, 4)
                    PHASE_VARIANT(256, 24, 0, 0)
                    for (int g : {1, 2, 4, 8}) PHASE_VARIANT(256, 24, g, 0)
                    // (iii) which half of the traffic minds: the loads alone and the stores alone, in lockstep, staggered, delayed
                    for (uint32_t m : {1u, 2u}) {
                        a.mode = m;
                        std::printf("# mode %u: %s only (GB/s still counts 38 B per lane)\n", m, m == 1 ? "loads (17 B per lane)" : "stores (21 B per lane)");
                        PHASE_VARIANT(512, 0, 0, 0)
                        PHASE_VARIANT(512, 0, 2, 0)
                        PHASE_VARIANT(512, 0, 8, 0)
                        PHASE_VARIANT(512, 0, 0, 4)
                        PHASE_VARIANT(512, 24, 0, 0)
                    }
                    a.mode = 0;
                    // (iv) one generation per launch: the same streams as k launches of n / k lanes each (k HIP launches back to back: each pays its own
                    // release + acquire, which a chain's launches would not)
#define SPLIT_VARIANT(THREADS_, ALU_, K_)                                                                                                 \
    {                                                                                                                                     \
        const uint64_t part = n4 / (K_);                                                                                                  \
        const uint64_t full = (part + (uint64_t)THREADS_ - 1) / (uint64_t)THREADS_;                                                      \
        char l[96];                                                                                                                       \
        std::snprintf(l, sizeof(l), "%d thr, %d FMA rounds, as %d launches of %llu lanes", THREADS_, ALU_, K_, (unsigned long long)(part * 4)); \
        report(l, time_launches(st, 20, [&] {                                                                                            \
                   for (int q = 0; q < (K_); ++q) {                                                                                       \
                       Streams b = a;                                                                                                     \
                       for (int j = 0; j < 4; ++j) { b.s_in[j] = a.s_in[j] + q * part; b.s_out[j] = a.s_out[j] + q * part; }              \
                       b.reward = a.reward + q * part; b.act = a.act + q * part; b.done = a.done + q * part;                              \
                       b.n4 = part;                                                                                                       \
                       hipLaunchKernelGGL((stream_kernel<THREADS_, 1, false, true, ALU_, 0>), dim3((uint32_t)full), dim3(THREADS_), 0, st, b); \
                   }                                                                                                                      \
               }));                                                                                                                       \
    }
                    for (int k : {1, 2, 4, 8, 16}) SPLIT_VARIANT(512, 24, k)
                    for (int k : {1, 2, 4, 8}) SPLIT_VARIANT(512, 0, k)
                    for (int k : {2, 4, 8}) SPLIT_VARIANT(256, 24, k)
                    HIP_OK(hipFree(pool));
                    continue;
                }
                STREAM_VARIANT(512, 1, true, true, 0) // the step's launch shape today (CartPole: 512 work-items, every access hinted)
                if (skew == 4352) {
                    STREAM_VARIANT(256, 1, true, true, 0)
                    STREAM_VARIANT(512, 1, false, false, 0)
                    STREAM_VARIANT(512, 1, true, false, 0)
                    STREAM_VARIANT(512, 1, false, true, 0)
                    STREAM_VARIANT(512, 2, true, true, 0)
                    STREAM_VARIANT(256, 2, true, true, 0)
                    STREAM_VARIANT(512, 1, true, true, 256 * 2)
                    STREAM_VARIANT(512, 1, true, true, 256 * 4)
                    STREAM_VARIANT(256, 1, true, true, 256 * 4)
                    STREAM_VARIANT(256, 1, true, true, 256 * 8)
                    STREAM_VARIANT(512, 2, true, true, 256 * 2)
                    STREAM_VARIANT(256, 2, true, true, 256 * 4)
                    STREAM_VARIANT(256, 2, true, true, 256 * 8)
                    STREAM_VARIANT(256, 4, true, true, 256 * 4)
                    // the mid-size question: the same streams with the step's arithmetic between loads and stores (plain loads, hinted stores)
#define ALU_VARIANT(THREADS_, TILES_, ALU_, LDS_)                                                                                          \
    {                                                                                                                                     \
        const uint64_t full = (n4 + (uint64_t)THREADS_ * TILES_ - 1) / ((uint64_t)THREADS_ * TILES_);                                    \
        char l[96];                                                                                                                       \
        std::snprintf(l, sizeof(l), "%d thr x %d tiles, plain loads + nt stores, %d FMA rounds, %d KB LDS", THREADS_, TILES_, ALU_, LDS_); \
        report(l, time_launches(st, 20, [&] {                                                                                            \
                   hipLaunchKernelGGL((stream_kernel<THREADS_, TILES_, false, true, ALU_, LDS_>), dim3((uint32_t)full), dim3(THREADS_), 0, st, a); \
               }));                                                                                                                       \
    }
                    if (inplace) {
                        ALU_VARIANT(512, 1, 0, 0)
                        ALU_VARIANT(512, 1, 12, 0)
                        ALU_VARIANT(512, 1, 24, 0)
                        ALU_VARIANT(512, 1, 48, 0)
                        ALU_VARIANT(512, 1, 96, 0)
                        ALU_VARIANT(512, 1, 24, 36)
                        ALU_VARIANT(256, 1, 24, 0)
                        ALU_VARIANT(256, 1, 24, 18)
                        ALU_VARIANT(512, 2, 24, 0)
                        ALU_VARIANT(256, 2, 48, 0)
                    }
                } else {
                    STREAM_VARIANT(256, 2, true, true, 256 * 4)
                }
                HIP_OK(hipFree(pool));
            }
        }
    }
    return 0;
}
