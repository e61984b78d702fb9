// tools/pkbench.hip -- developer tool: issue cost of packed f32 VALU instructions next to their scalar forms on gfx950, with the SIMD as full as the step
// kernel keeps it (4 waves per SIMD) and with one wave per SIMD.  Every variant is 8 INDEPENDENT dependency chains of one instruction, written in inline
// assembly so that the compiler can neither pack nor unpack them; the figure is shader-clock cycles (s_memtime) per instruction per wave, times the waves
// that share the SIMD = the SIMD's issue cost per instruction.
//     hipcc --offload-arch=gfx950 -O3 tools/pkbench.hip -o tools/pkbench && tools/pkbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void pk(unsigned long long* cyc, float* sink, float s, int iters)
{
    float a[8];
    float2v p[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 1e-3f + i;
        p[i] = float2v{a[i], a[i] + 0.5f};
    }
    const float2v ss = {s, s}, cc = {1e-3f, 2e-3f};
    unsigned long long mask = 0x0f0f0f0f12345678ull ^ (unsigned long long)iters, m[8] = {}, q[8];
    unsigned u = 0x80000000u, u2 = threadIdx.x * 2654435761u, sr[8] = {};
    for (int i = 0; i < 8; ++i) q[i] = threadIdx.x * 0x9E3779B97F4A7C15ull + i;
    asm volatile("s_mov_b64 vcc, %0" : : "s"(MODE == 23 ? ~0ull : mask) : "vcc");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(cc.x));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(ss), "v"(cc));
                if (MODE == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(ss));
                if (MODE == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (MODE == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(ss));
                if (MODE == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(s));
                if (MODE == 7) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(p[i]) : "v"(ss), "v"(cc));
                if (MODE == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(p[i]) : "s"(ss), "v"(cc)); // an SGPR pair as the broadcast operand, as the compiler writes it
                if (MODE == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s) : );
                if (MODE == 10) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "s"(mask));
                if (MODE == 11) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(s), "v"(cc.x)); // destination not a source
                if (MODE == 12) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(s) : "vcc");
                if (MODE == 13) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m[i]) : "v"(a[i]), "v"(s));
                if (MODE == 14) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(u), "v"(u2) : "vcc");
                if (MODE == 15) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(u), "v"(s));
                if (MODE == 16) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(u));
                if (MODE == 17) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 18) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(s) : "vcc");
                if (MODE == 19) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(cc.x));
                if (MODE == 20) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(cc.x));
                if (MODE == 21) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sr[i]) : "v"(a[i]));
                if (MODE == 22) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(u), "v"(u2));
                if (MODE == 23) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s) : ); // as 9, but vcc = all lanes (set before the loop)
                if (MODE == 24) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(cc.x)); // an SGPR operand
                if (MODE == 25) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
                if (MODE == 26) asm volatile("v_cmp_nle_f32_e64 %0, |%1|, %2" : "=s"(m[i]) : "v"(a[i]), "s"(s));
                if (MODE == 27) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(p[i]) : "v"(ss));
                if (MODE == 28) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y + (float)m[i] + (float)q[i] + (float)sr[i];
    if (r == 12345.678f) sink[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int waves_per_simd)
{
    const int iters = 512, per_iter = 32;
    const int blocks = 256 * waves_per_simd; // 256-work-item groups = 4 waves = one per SIMD of a CU
    unsigned long long* cyc;
    float* sink;
    CK(hipMalloc(&cyc, blocks * 4 * 8));
    CK(hipMalloc(&sink, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((pk<MODE>), dim3(blocks), dim3(256), 0, 0, cyc, sink, 0.999f, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((pk<MODE>), dim3(blocks), dim3(256), 0, 0, cyc, sink, 0.999f, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 4);
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2], n = (double)iters * per_iter;
    // s_memtime ticks at a constant 100 MHz on this part; the kernel time gives the second view
    printf("%-44s %d wave(s)/SIMD: wave median %8.0f ticks; kernel %8.2f us -> %.2f ns per instruction per wave, %.2f ns of SIMD issue per instruction (%.2f cycles @ 2.4 GHz)\n", name,
           waves_per_simd, med, ms * 1e3, ms * 1e6 / n, ms * 1e6 / n / waves_per_simd, ms * 1e6 / n / waves_per_simd * 2.4);
    CK(hipFree(cyc));
    CK(hipFree(sink));
}

int main()
{
    for (int w : {4, 1}) {
        run<0>("v_fma_f32", w);
        run<1>("v_pk_fma_f32", w);
        run<8>("v_pk_fma_f32 (SGPR pair operand, op_sel_hi)", w);
        run<2>("v_mul_f32", w);
        run<3>("v_pk_mul_f32", w);
        run<4>("v_add_f32", w);
        run<5>("v_pk_add_f32", w);
        run<6>("v_mov_b32", w);
        run<9>("v_cndmask_b32", w);
        run<7>("v_fma_f64", w);
        run<27>("v_mul_f64", w);
        run<24>("v_fma_f32 (one SGPR operand)", w);
        run<10>("v_cndmask_b32_e64 (SGPR-pair mask)", w);
        run<11>("v_cndmask_b32 vcc, destination not a source", w);
        run<23>("v_cndmask_b32 vcc = all lanes", w);
        run<12>("v_cmp_lt_f32 vcc", w);
        run<13>("v_cmp_lt_f32_e64 -> SGPR pair", w);
        run<26>("v_cmp_nle_f32_e64 |v|, s -> SGPR pair", w);
        run<14>("v_mad_u64_u32", w);
        run<28>("v_lshl_add_u64", w);
        run<15>("v_bfi_b32", w);
        run<16>("v_xor_b32", w);
        run<22>("v_and_or_b32", w);
        run<17>("v_rcp_f32", w);
        run<18>("v_div_scale_f32", w);
        run<19>("v_div_fmas_f32", w);
        run<20>("v_div_fixup_f32", w);
        run<25>("v_cvt_f32_u32", w);
        run<21>("v_readlane_b32", w);
    }
    return 0;
}
