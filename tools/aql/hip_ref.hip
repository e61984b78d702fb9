// tools/aql/hip_ref.hip -- the probe's stepish kernel launched through the HIP runtime (reference for aql_probe's numbers)
#include "aql_kernels.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 0) : (1u << 20);
    const int launches = argc > 2 ? std::atoi(argv[2]) : 4000;
    Args a{};
    for (int j = 0; j < 4; ++j) { CK(hipMalloc(&a.s[j], n * 4 + 8192)); CK(hipMemset(a.s[j], 0, n * 4)); }
    CK(hipMalloc((void**)&a.a, n + 64)); CK(hipMemset((void*)a.a, 0, n));
    CK(hipMalloc(&a.r, n * 4 + 64)); CK(hipMalloc(&a.d, n + 64)); CK(hipMalloc(&a.ver, n / 256 * 4 + 64)); CK(hipMalloc(&a.err, 64));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (unsigned alu : {0u, 100u}) {
        a.alu = alu; a.versioned = 0;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int t = 0; t < launches; ++t) { a.t = t; hipLaunchKernelGGL(stepish_kernel, dim3(n / 4 / 512), dim3(512), 0, st, a); }
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) std::printf("HIP launches, stepish alu=%u: %.3f us/launch\n", alu, ms * 1e3 / launches);
        }
    }
    return 0;
}
