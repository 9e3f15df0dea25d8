// Microbenchmark: does the time of a VALU instruction depend on how many (and which) lanes of the wave are enabled?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long laneMask, float a, float b) {
    uint32_t lane = threadIdx.x & 63;
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    if ((laneMask >> lane) & 1ull) {
        for (int i = 0; i < iters; i++) {   // four independent chains: issue-bound, not latency-bound
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}

int main() {
    float* out;
    const int blocks = 256 * 8, iters = 20000;   // 8 waves per SIMD
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct { const char* name; unsigned long long mask; } cases[] = {
        {"64 lanes", ~0ull}, {"low 32", 0xffffffffull}, {"high 32", 0xffffffff00000000ull}, {"low 16", 0xffffull},
        {"alternate lanes", 0x5555555555555555ull}, {"lanes 16-31", 0xffff0000ull}, {"1 lane", 1ull}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, c.mask, 0.999f, 0.001f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, c.mask, 0.999f, 0.001f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // per SIMD: 8 waves x iters x 8 instructions
        printf("%-16s: %8.1f us, %.2f ns per wave-instruction per SIMD\n", c.name, ms * 1e3, ms * 1e6 / (8.0 * iters * 8));
    }
    return 0;
}
