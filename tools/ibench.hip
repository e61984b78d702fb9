// tools/ibench.hip — instruction-cost microbenchmarks (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int MODE, int K>
__global__ __launch_bounds__(256) void ib(unsigned* out, unsigned seed, float fs)
{
    unsigned a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 77u, d = b + 1234567u;
    float fa = a * 1e-9f, fb = b * 1e-9f, fc = c * 1e-9f, fd = d * 1e-9f;
#pragma unroll 8
    for (int i = 0; i < K; ++i) {
        if (MODE == 0) { // v_mad_u64_u32 (4 independent chains)
            unsigned long long p0 = (unsigned long long)a * 0xD2511F53u, p1 = (unsigned long long)b * 0xCD9E8D57u;
            unsigned long long p2 = (unsigned long long)c * 0xD2511F53u, p3 = (unsigned long long)d * 0xCD9E8D57u;
            a = (unsigned)(p0 >> 32) ^ (unsigned)p0; b = (unsigned)(p1 >> 32) ^ (unsigned)p1; c = (unsigned)(p2 >> 32) ^ (unsigned)p2; d = (unsigned)(p3 >> 32) ^ (unsigned)p3;
        } else if (MODE == 1) { // mul_hi + mul_lo separately
            unsigned h0 = __umulhi(a, 0xD2511F53u), l0 = a * 0xD2511F53u, h1 = __umulhi(b, 0xCD9E8D57u), l1 = b * 0xCD9E8D57u;
            unsigned h2 = __umulhi(c, 0xD2511F53u), l2 = c * 0xD2511F53u, h3 = __umulhi(d, 0xCD9E8D57u), l3 = d * 0xCD9E8D57u;
            asm volatile("" : "+v"(h0), "+v"(l0), "+v"(h1), "+v"(l1));
            asm volatile("" : "+v"(h2), "+v"(l2), "+v"(h3), "+v"(l3));
            a = h0 ^ l0; b = h1 ^ l1; c = h2 ^ l2; d = h3 ^ l3;
        } else if (MODE == 2) { // mul_lo only
            a = a * 0xD2511F53u + 1u; b = b * 0xCD9E8D57u + 1u; c = c * 0xD2511F53u + 3u; d = d * 0xCD9E8D57u + 5u;
        } else if (MODE == 3) { // 24-bit mul
            a = __umul24(a, 0x511F53u) + 1u; b = __umul24(b, 0x9E8D57u) + 1u; c = __umul24(c, 0x511F53u) + 3u; d = __umul24(d, 0x9E8D57u) + 5u;
        } else if (MODE == 4) { // fma f32
            fa = __builtin_fmaf(fa, fs, 1e-3f); fb = __builtin_fmaf(fb, fs, 1e-3f); fc = __builtin_fmaf(fc, fs, 1e-3f); fd = __builtin_fmaf(fd, fs, 1e-3f);
        } else if (MODE == 5) { // xor/add
            a = (a ^ b) + 1u; b = (b ^ c) + 3u; c = (c ^ d) + 5u; d = (d ^ a) + 7u;
        } else if (MODE == 6) { // mul_hi only
            a = __umulhi(a, 0xD2511F53u) + 1u; b = __umulhi(b, 0xCD9E8D57u) + 1u; c = __umulhi(c, 0xD2511F53u) + 3u; d = __umulhi(d, 0xCD9E8D57u) + 5u;
        }
    }
    unsigned r = a ^ b ^ c ^ d ^ __float_as_uint(fa + fb + fc + fd);
    if (r == 0x12345678u) out[threadIdx.x] = r;
}

template <int MODE>
static void run(const char* name, int ops_per_iter)
{
    constexpr int K = 4096;
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned* out; CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 1024; // 4 waves per SIMD
    hipLaunchKernelGGL((ib<MODE, K>), dim3(blocks), dim3(256), 0, st, out, 1u, 0.999f);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((ib<MODE, K>), dim3(blocks), dim3(256), 0, st, out, 1u + i, 0.999f);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / 5;
    double wave_instrs_per_simd = 4.0 * K * ops_per_iter; // 4 waves per SIMD
    printf("%-28s %8.1f us  -> %.2f ns per wave-instruction per SIMD (%.1f cycles @2.4GHz)\n", name, us, us * 1e3 / wave_instrs_per_simd,
           us * 1e3 / wave_instrs_per_simd * 2.4);
}

int main()
{
    run<0>("v_mad_u64_u32 (+xor)", 4);
    run<1>("mul_hi + mul_lo (+xor)", 8);
    run<2>("v_mul_lo_u32 (+add)", 4);
    run<6>("v_mul_hi_u32 (+add)", 4);
    run<3>("v_mul_u32_u24 (+add)", 4);
    run<4>("v_fma_f32", 4);
    run<5>("xor+add", 8);
    return 0;
}
