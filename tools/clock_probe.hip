// tools/clock_probe.hip -- developer tooling (not part of the product): what the shader clock is RIGHT NOW, measured on the
// device itself.  One wavefront reads s_memtime (a counter of the shader core clock) and s_memrealtime (the constant
// 100 MHz reference clock) around a dependent FMA chain: sclk = d(memtime) / d(memrealtime) * 100 MHz.  A second kernel
// is a pure VALU load over the whole chip (time ~ 1 / sclk) for cross-checking.   hipcc --offload-arch=gfx950 -shared -fPIC
#include <hip/hip_runtime.h>

__global__ void clk_kernel(unsigned long long* out, int iters)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    float x = (float)threadIdx.x;
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 1.0000001f, 0.5f);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
        out[2] = (unsigned long long)x;
    }
}

__global__ void alu_kernel(float* sink, int iters)
{
    float x = (float)threadIdx.x, y = (float)blockIdx.x;
    for (int i = 0; i < iters; ++i) {
        x = __builtin_fmaf(x, 1.0000001f, 0.5f);
        y = __builtin_fmaf(y, 0.9999999f, 0.25f);
    }
    if (x + y == 12345.678f) sink[0] = x;
}

static unsigned long long* g_out = nullptr;
static unsigned long long* g_out_dev = nullptr;
static hipStream_t g_stream = nullptr;
static hipEvent_t g_e0 = nullptr, g_e1 = nullptr;
static float* g_sink = nullptr;

static int init()
{
    if (g_out) return 0;
    if (hipHostMalloc(&g_out, 64, hipHostMallocMapped) != hipSuccess) return 1;
    if (hipHostGetDevicePointer((void**)&g_out_dev, g_out, 0) != hipSuccess) return 2;
    if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 3;
    if (hipEventCreate(&g_e0) != hipSuccess || hipEventCreate(&g_e1) != hipSuccess) return 4;
    if (hipMalloc(&g_sink, 64) != hipSuccess) return 5;
    return 0;
}

// sclk in MHz over ~`iters` dependent FMAs of one wavefront (20000 -> ~35 us); also the cycles per FMA seen
extern "C" int clock_probe(int iters, double* sclk_mhz, double* cycles_per_fma)
{
    if (int rc = init()) return rc;
    hipLaunchKernelGGL(clk_kernel, dim3(1), dim3(64), 0, g_stream, g_out_dev, iters);
    if (hipStreamSynchronize(g_stream) != hipSuccess) return 10;
    const double dc = (double)g_out[0], dr = (double)g_out[1];
    *sclk_mhz = dr > 0 ? dc / dr * 100.0 : 0.0;
    *cycles_per_fma = dc / (double)iters;
    return 0;
}

// microseconds of a chip-wide VALU-only kernel (2048 workgroups x 256 work-items, 2 x iters FMAs each)
extern "C" int alu_probe(int iters, double* us)
{
    if (int rc = init()) return rc;
    hipLaunchKernelGGL(alu_kernel, dim3(2048), dim3(256), 0, g_stream, g_sink, iters);
    hipEventRecord(g_e0, g_stream);
    hipLaunchKernelGGL(alu_kernel, dim3(2048), dim3(256), 0, g_stream, g_sink, iters);
    hipEventRecord(g_e1, g_stream);
    if (hipStreamSynchronize(g_stream) != hipSuccess) return 10;
    float ms = 0;
    hipEventElapsedTime(&ms, g_e0, g_e1);
    *us = ms * 1e3;
    return 0;
}
