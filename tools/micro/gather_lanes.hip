// Microbenchmark: how the cost of a per-lane 64-byte gather (4 x 16 B from one record) depends on WHICH lanes of the
// wave are active.  Each active lane walks a pseudo-random dependent chain over `n` 64-byte records (n a power of 2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k(const float4* rec, uint32_t n, int steps, uint32_t* out, unsigned long long laneMask) {
    uint32_t lane = threadIdx.x & 63;
    uint32_t idx = ((blockIdx.x * 256 + threadIdx.x) * 2654435761u) & (n - 1);
    float acc = 0;
    if ((laneMask >> lane) & 1ull) {
        for (int s = 0; s < steps; s++) {
            const float4* r = rec + (size_t)idx * 4;
            float4 a = r[0], b = r[1], c = r[2], d = r[3];
            acc += a.x + b.y + c.z + d.w;
            idx = (__float_as_uint(a.x) ^ __float_as_uint(d.w) ^ (idx * 747796405u + lane)) & (n - 1);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(acc) ^ idx;
}

int main(int argc, char** argv) {
    uint32_t n = argc > 1 ? atoi(argv[1]) : 65536;
    int steps = 400, blocks = 1024;
    std::vector<float> h((size_t)n * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 8);
    float4* rec; uint32_t* out;
    CHECK(hipMalloc(&rec, h.size() * 4));
    CHECK(hipMemcpy(rec, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, 4096 * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct { const char* name; unsigned long long mask; } cases[] = {
        {"64 lanes", ~0ull}, {"32 contiguous", 0xffffffffull}, {"32 alternate lanes", 0x5555555555555555ull},
        {"16 contiguous", 0xffffull}, {"16 = one lane per quad", 0x1111111111111111ull}, {"16 = 4 full quads spread", 0x000f000f000f000full},
        {"8 contiguous", 0xffull}, {"4 contiguous (one quad)", 0xfull}, {"1 lane", 1ull}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, rec, n, steps, out, c.mask);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, rec, n, steps, out, c.mask);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s: %7.1f us, %.1f ns per wave-step per CU (16 waves per CU)\n", c.name, ms * 1e3, ms * 1e6 / steps / 16);
    }
    return 0;
}
