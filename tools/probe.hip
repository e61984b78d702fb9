// tools/probe.hip — developer micro-benchmarks (not part of the product or the tests).
// Measures, on the box it runs on, (a) the streaming floor for the CartPole step's access pattern
// (same arrays, same bytes, trivial arithmetic), one tile per workgroup and a grid-stride persistent
// form; (b) the engine's step kernel under different flags / lanes-per-thread through the C ABI.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>
#include "gymrs_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int V> struct alignas(4 * V) F { float v[V]; };
template <int V> struct alignas(V) B { unsigned char v[V]; };

// one tile per workgroup: read 4 f32 arrays + u8 action, write 4 f32 + f32 reward + u8 done
template <int V>
__global__ __launch_bounds__(256) void copy_tile(float* s0, float* s1, float* s2, float* s3, const unsigned char* act,
                                                 float* rew, unsigned char* done, size_t n)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (i + V > n) return;
    F<V> a = *(F<V>*)(s0 + i), b = *(F<V>*)(s1 + i), c = *(F<V>*)(s2 + i), d = *(F<V>*)(s3 + i);
    B<V> u = *(const B<V>*)(act + i);
    F<V> r; B<V> dn;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        float f = u.v[k] ? 1.0f : -1.0f;
        a.v[k] += 0.02f * b.v[k]; b.v[k] += 0.02f * f; c.v[k] += 0.02f * d.v[k]; d.v[k] -= 0.02f * f;
        r.v[k] = 1.0f; dn.v[k] = a.v[k] > 2.4f;
    }
    *(F<V>*)(s0 + i) = a; *(F<V>*)(s1 + i) = b; *(F<V>*)(s2 + i) = c; *(F<V>*)(s3 + i) = d;
    *(F<V>*)(rew + i) = r; *(B<V>*)(done + i) = dn;
}

// copy_tile with non-temporal stores / loads (does the end-of-kernel write-back get cheaper?)
template <int V, int MODE>
__global__ __launch_bounds__(256) void copy_tile_nt(float* s0, float* s1, float* s2, float* s3, const unsigned char* act,
                                                    float* rew, unsigned char* done, size_t n)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 > n) return;
    f4 a, b, c, d;
    if (MODE & 2) {
        a = __builtin_nontemporal_load((f4*)(s0 + i)); b = __builtin_nontemporal_load((f4*)(s1 + i));
        c = __builtin_nontemporal_load((f4*)(s2 + i)); d = __builtin_nontemporal_load((f4*)(s3 + i));
    } else {
        a = *(f4*)(s0 + i); b = *(f4*)(s1 + i); c = *(f4*)(s2 + i); d = *(f4*)(s3 + i);
    }
    unsigned u = *(const unsigned*)(act + i);
    f4 r = {1.f, 1.f, 1.f, 1.f};
    unsigned dn = 0;
    for (int k = 0; k < 4; ++k) {
        float f = ((u >> (8 * k)) & 0xff) ? 1.0f : -1.0f;
        a[k] += 0.02f * b[k]; b[k] += 0.02f * f; c[k] += 0.02f * d[k]; d[k] -= 0.02f * f;
        dn |= (a[k] > 2.4f ? 1u : 0u) << (8 * k);
    }
    if (MODE & 1) {
        __builtin_nontemporal_store(a, (f4*)(s0 + i)); __builtin_nontemporal_store(b, (f4*)(s1 + i));
        __builtin_nontemporal_store(c, (f4*)(s2 + i)); __builtin_nontemporal_store(d, (f4*)(s3 + i));
        __builtin_nontemporal_store(r, (f4*)(rew + i)); __builtin_nontemporal_store(dn, (unsigned*)(done + i));
    } else {
        *(f4*)(s0 + i) = a; *(f4*)(s1 + i) = b; *(f4*)(s2 + i) = c; *(f4*)(s3 + i) = d; *(f4*)(rew + i) = r; *(unsigned*)(done + i) = dn;
    }
}

// copy_tile + K rounds of arithmetic per work-item between the loads and the stores: calibrates what one
// VALU instruction per wave costs in this launch shape.  MODE 0: 4 independent scalar FMA chains
// (one per lane of the work-item), MODE 1: the same as 2 packed chains.
typedef float pf2 __attribute__((ext_vector_type(2)));
template <int K, int MODE>
__global__ __launch_bounds__(256) void copy_alu(float* s0, float* s1, float* s2, float* s3, const unsigned char* act,
                                                float* rew, unsigned char* done, size_t n, float ca, float cb)
{
    constexpr int V = 4;
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (i + V > n) return;
    F<V> a = *(F<V>*)(s0 + i), b = *(F<V>*)(s1 + i), c = *(F<V>*)(s2 + i), d = *(F<V>*)(s3 + i);
    B<V> u = *(const B<V>*)(act + i);
    if (MODE == 0) {
#pragma unroll
        for (int r = 0; r < K; ++r) {
#pragma unroll
            for (int k = 0; k < V; ++k) a.v[k] = __builtin_fmaf(a.v[k], ca, cb);
        }
    } else {
        pf2 x = {a.v[0], a.v[1]}, y = {a.v[2], a.v[3]};
        const pf2 pa = {ca, ca}, pb = {cb, cb};
#pragma unroll
        for (int r = 0; r < K; ++r) {
            x = __builtin_elementwise_fma(x, pa, pb);
            y = __builtin_elementwise_fma(y, pa, pb);
        }
        a.v[0] = x.x; a.v[1] = x.y; a.v[2] = y.x; a.v[3] = y.y;
    }
    F<V> r; B<V> dn;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        float f = u.v[k] ? 1.0f : -1.0f;
        b.v[k] += 0.02f * f; c.v[k] += 0.02f * d.v[k]; d.v[k] -= 0.02f * f;
        r.v[k] = 1.0f; dn.v[k] = a.v[k] > 2.4f;
    }
    *(F<V>*)(s0 + i) = a; *(F<V>*)(s1 + i) = b; *(F<V>*)(s2 + i) = c; *(F<V>*)(s3 + i) = d;
    *(F<V>*)(rew + i) = r; *(B<V>*)(done + i) = dn;
}

// persistent grid-stride form with a register prefetch of the next tile
template <int V>
__global__ __launch_bounds__(256) void copy_persist(float* s0, float* s1, float* s2, float* s3, const unsigned char* act,
                                                    float* rew, unsigned char* done, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256 * V;
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (i + V > n) return;
    F<V> a = *(F<V>*)(s0 + i), b = *(F<V>*)(s1 + i), c = *(F<V>*)(s2 + i), d = *(F<V>*)(s3 + i);
    B<V> u = *(const B<V>*)(act + i);
    while (true) {
        size_t j = i + stride;
        bool more = j + V <= n;
        F<V> na, nb, nc, nd; B<V> nu;
        if (more) { na = *(F<V>*)(s0 + j); nb = *(F<V>*)(s1 + j); nc = *(F<V>*)(s2 + j); nd = *(F<V>*)(s3 + j); nu = *(const B<V>*)(act + j); }
        F<V> r; B<V> dn;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float f = u.v[k] ? 1.0f : -1.0f;
            a.v[k] += 0.02f * b.v[k]; b.v[k] += 0.02f * f; c.v[k] += 0.02f * d.v[k]; d.v[k] -= 0.02f * f;
            r.v[k] = 1.0f; dn.v[k] = a.v[k] > 2.4f;
        }
        *(F<V>*)(s0 + i) = a; *(F<V>*)(s1 + i) = b; *(F<V>*)(s2 + i) = c; *(F<V>*)(s3 + i) = d;
        *(F<V>*)(rew + i) = r; *(B<V>*)(done + i) = dn;
        if (!more) break;
        a = na; b = nb; c = nc; d = nd; u = nu; i = j;
    }
}

static float time_launches(hipStream_t st, int iters, const std::function<void()>& fn)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 200; ++i) fn();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

#include <functional>

int main(int argc, char** argv)
{
    size_t n = (argc > 1) ? (size_t)atoll(argv[1]) : (1u << 20);
    int iters = (argc > 2) ? atoi(argv[2]) : 2000;
    CK(hipSetDevice(0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *s[4], *rew; unsigned char *act, *done;
    for (auto& p : s) { CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); }
    CK(hipMalloc(&rew, n * 4)); CK(hipMalloc(&act, n * 32)); CK(hipMalloc(&done, n));
    CK(hipMemset(act, 1, n * 32));
    const double bytes = 38.0 * n;
    printf("n = %zu lanes, %d launches per measurement, algorithmic bytes per launch = %.1f MB\n", n, iters, bytes / 1e6);
    auto report = [&](const char* name, float us) { printf("%-44s %8.2f us/launch  %7.1f GB/s  frac(8TB/s) %.3f\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0); fflush(stdout); };

#define TILE(V) report("copy_tile V=" #V, time_launches(st, iters, [&] { hipLaunchKernelGGL(copy_tile<V>, dim3((n + 256 * V - 1) / (256 * V)), dim3(256), 0, st, s[0], s[1], s[2], s[3], act, rew, done, n); }))
    TILE(1); TILE(2); TILE(4);
#define PERS(V, G) report("copy_persist V=" #V " grid=" #G, time_launches(st, iters, [&] { hipLaunchKernelGGL(copy_persist<V>, dim3(G), dim3(256), 0, st, s[0], s[1], s[2], s[3], act, rew, done, n); }))
    PERS(1, 512); PERS(1, 1024); PERS(1, 2048); PERS(2, 512); PERS(2, 1024); PERS(4, 256); PERS(4, 512);

#define NT(M) report("copy_tile_nt mode=" #M " (1=nt stores, 2=nt loads)", time_launches(st, iters, [&] { hipLaunchKernelGGL((copy_tile_nt<4, M>), dim3((n + 1023) / 1024), dim3(256), 0, st, s[0], s[1], s[2], s[3], act, rew, done, n); }))
    NT(0); NT(1); NT(2); NT(3);
#define ALU(K, M) report("copy_alu V=4 K=" #K " mode=" #M, time_launches(st, iters, [&] { hipLaunchKernelGGL((copy_alu<K, M>), dim3((n + 1023) / 1024), dim3(256), 0, st, s[0], s[1], s[2], s[3], act, rew, done, n, 0.999f, 1e-3f); }))
    ALU(0, 0); ALU(25, 0); ALU(50, 0); ALU(100, 0); ALU(200, 0); ALU(400, 0); ALU(25, 1); ALU(50, 1); ALU(100, 1); ALU(200, 1); ALU(400, 1);
    // the engine's kernels through the C ABI
    struct Cfg { int kind; uint32_t flags; int vec, tiles, prio; };
    std::vector<Cfg> cfgs;
    const uint32_t AS = GYMRS_AUTO_RESET | GYMRS_TRACK_STATS;
    for (int vec : {4, 8, 16}) for (int tiles : {1}) for (int prio : {0}) {
        cfgs.push_back({0, AS, vec, tiles, prio});
    }
    for (int vec : {4, 8, 16}) for (int tiles : {1}) { cfgs.push_back({0, 0u, vec, tiles, 0}); cfgs.push_back({0, GYMRS_AUTO_RESET, vec, tiles, 0}); cfgs.push_back({1, AS, vec, tiles, 0}); cfgs.push_back({2, AS | GYMRS_TIME_LIMIT, vec, tiles, 0}); }
    for (const Cfg& cf : cfgs) {
        const int kind = cf.kind;
        const size_t asz = kind == 2 ? 4 : 1;
        gymrs_engine* e = nullptr;
        if (gymrs_engine_create((gymrs_env_kind)kind, n, 0, 0, nullptr, cf.flags, &e) != GYMRS_OK) { printf("create failed: %s\n", gymrs_last_error()); return 1; }
        gymrs_set_stream(e, st);
        if (gymrs_set_tuning(e, cf.vec, 0) != GYMRS_OK) { printf("tuning failed: %s\n", gymrs_last_error()); return 1; }
        gymrs_reset(e, 1, 0, nullptr, nullptr);
        for (int b = 0; b < 8; ++b) gymrs_fill_actions(e, act + (size_t)b * n * asz, 1, b);
        gymrs_step_many(e, act, n * asz, 8, 300, 0);
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        gymrs_step_many(e, act, n * asz, 8, iters, 0);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        char name[96]; snprintf(name, sizeof name, "engine kind=%d flags=%u vec=%d tiles=%d prio=%d", kind, cf.flags, cf.vec, cf.tiles, cf.prio);
        const double b2 = (kind == 0 ? 38.0 : kind == 1 ? 22.0 : 37.0) * n;
        float us = ms * 1e3f / iters;
        printf("%-50s %8.2f us/launch  %7.1f GB/s  frac(8TB/s) %.3f\n", name, us, b2 / us / 1e3, b2 / us / 1e3 / 8000.0); fflush(stdout);
        gymrs_engine_destroy(e);
    }
    // empty-kernel and small-copy floors: the fixed cost of one dependent launch
    report("empty kernel (1 block)", time_launches(st, iters, [&] { hipLaunchKernelGGL(copy_tile<4>, dim3(1), dim3(256), 0, st, s[0], s[1], s[2], s[3], act, rew, done, (size_t)0); }));
    for (size_t m : {(size_t)1 << 14, (size_t)1 << 16, (size_t)1 << 18, (size_t)1 << 19, n}) {
        char name[64]; snprintf(name, sizeof name, "copy_tile V=4 on %zu lanes (%.1f MB)", m, 38.0 * m / 1e6);
        float us = time_launches(st, iters, [&] { hipLaunchKernelGGL(copy_tile<4>, dim3((m + 1023) / 1024), dim3(256), 0, st, s[0], s[1], s[2], s[3], act, rew, done, m); });
        printf("%-50s %8.2f us/launch\n", name, us);
    }
    // lane-partitioned chains on separate streams, eager and as a captured HIP graph
    for (int parts : {1}) for (int graph : {0}) for (int vec : {4}) {
        std::vector<gymrs_engine*> es(parts); std::vector<hipStream_t> ss(parts);
        size_t np = n / parts;
        for (int p = 0; p < parts; ++p) {
            CK(hipStreamCreateWithFlags(&ss[p], hipStreamNonBlocking));
            gymrs_engine_create(GYMRS_CARTPOLE, np, p * np, 0, nullptr, GYMRS_AUTO_RESET | GYMRS_TRACK_STATS, &es[p]);
            gymrs_set_stream(es[p], ss[p]); gymrs_set_tuning(es[p], vec, 0); gymrs_reset(es[p], 1, 0, nullptr, nullptr);
            for (int b = 0; b < 8; ++b) gymrs_fill_actions(es[p], act + (size_t)b * n + p * np, 1, b);
        }
        for (int p = 0; p < parts; ++p) CK(hipStreamSynchronize(ss[p]));
        auto run = [&](int k) { for (int t = 0; t < k; ++t) for (int p = 0; p < parts; ++p) gymrs_step(es[p], act + (size_t)(t % 8) * n + p * np); };
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms = 0;
        if (!graph) {
            run(300);
            for (int p = 0; p < parts; ++p) CK(hipStreamSynchronize(ss[p]));
            CK(hipEventRecord(e0, ss[0]));
            run(iters);
            for (int p = 1; p < parts; ++p) { hipEvent_t j; CK(hipEventCreateWithFlags(&j, hipEventDisableTiming)); CK(hipEventRecord(j, ss[p])); CK(hipStreamWaitEvent(ss[0], j, 0)); }
            CK(hipEventRecord(e1, ss[0]));
            for (int p = 0; p < parts; ++p) CK(hipStreamSynchronize(ss[p]));
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= iters;
        } else {
            const int G = 64; // steps per graph (tick is baked at capture: timing probe only)
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(ss[0], hipStreamCaptureModeGlobal));
            hipEvent_t fork; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
            CK(hipEventRecord(fork, ss[0]));
            for (int p = 1; p < parts; ++p) CK(hipStreamWaitEvent(ss[p], fork, 0));
            run(G);
            for (int p = 1; p < parts; ++p) { hipEvent_t j; CK(hipEventCreateWithFlags(&j, hipEventDisableTiming)); CK(hipEventRecord(j, ss[p])); CK(hipStreamWaitEvent(ss[0], j, 0)); }
            CK(hipStreamEndCapture(ss[0], &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, ss[0]));
            CK(hipStreamSynchronize(ss[0]));
            const int reps = iters / G;
            CK(hipEventRecord(e0, ss[0]));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, ss[0]));
            CK(hipEventRecord(e1, ss[0]));
            CK(hipStreamSynchronize(ss[0]));
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= (reps * G);
        }
        char name[96]; snprintf(name, sizeof name, "cartpole AUTO|STATS vec=%d, %d chain(s), %s", vec, parts, graph ? "graph" : "eager");
        report(name, ms * 1e3f);
        for (auto e : es) gymrs_engine_destroy(e);
    }
    return 0;
}
