// Microbenchmark: cost of per-lane 64-byte record gathers (the BVH node fetch pattern) on one GPU.
// Each lane walks a pseudo-random dependent chain over `n` 64-byte records; variants differ in how the record is read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* rec, uint32_t n, int steps, uint32_t* out, int coherent) {
    extern __shared__ float4 cache[];
    uint32_t lane = threadIdx.x & 63;
    uint32_t idx = coherent ? (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u % n : (blockIdx.x * 256 + threadIdx.x) * 2654435761u % n;
    float acc = 0;
    if (MODE == 3) {  // LDS-resident records (first 1024 records copied in)
        for (int i = threadIdx.x; i < 1024 * 4; i += 256) cache[i] = rec[i];
        __syncthreads();
    }
    for (int s = 0; s < steps; s++) {
        float4 a, b, c, d;
        if (MODE == 0) { const float4* r = rec + (size_t)idx * 4; a = r[0]; b = r[1]; c = r[2]; d = r[3]; }
        else if (MODE == 1) { const float4* r = rec + (size_t)idx * 4; a = r[0]; b = a; c = a; d = a; }           // one 16-byte load
        else if (MODE == 2) { const float4* r = rec + (size_t)idx * 4; a = r[0]; b = r[1]; c = a; d = b; }        // two loads
        else if (MODE == 4 || MODE == 5) {
            // cooperative: instruction k fetches the records of lanes 16k..16k+15, four adjacent lanes reading the four
            // quarters of one record (coalesced 64-byte runs), then the wave transposes through LDS
            const int STR = MODE == 5 ? 5 : 4;   // record stride in float4 (5: padded against bank conflicts)
            float4* my = cache + (threadIdx.x >> 6) * (64 * 5);
            #pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t j = __shfl(idx, q * 16 + (lane >> 2));
                float4 piece = rec[(size_t)j * 4 + (lane & 3)];
                my[(q * 16 + (lane >> 2)) * STR + (lane & 3)] = piece;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            a = my[lane * STR]; b = my[lane * STR + 1]; c = my[lane * STR + 2]; d = my[lane * STR + 3];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        else { const float4* r = cache + (size_t)(idx & 1023) * 4; a = r[0]; b = r[1]; c = r[2]; d = r[3]; }      // LDS
        acc += a.x + b.y + c.z + d.w;
        idx = (__float_as_uint(a.x) ^ __float_as_uint(d.w) ^ (idx * 747796405u + lane)) & (n - 1);   // dependent next index (n is a power of two)
    }
    out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(acc) ^ idx;
}

int main(int argc, char** argv) {
    uint32_t n = argc > 1 ? atoi(argv[1]) : 65536;   // records (64 B each), a power of two: 65536 = 4 MB
    int steps = 400;
    std::vector<float> h((size_t)n * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 8);
    float4* rec; uint32_t* out;
    CHECK(hipMalloc(&rec, h.size() * 4));
    CHECK(hipMemcpy(rec, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, 4096 * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[6] = {"4x16B global", "1x16B global", "2x16B global", "4x16B LDS", "coop quads", "coop padded"};
    for (int coherent = 0; coherent < 1; coherent++)
        for (int mode = 0; mode < 6; mode++)
            for (int blocks : {256, 512, 1024, 2048}) {
                size_t lds = mode == 3 ? 65536 : (mode >= 4 ? 4 * 64 * 5 * 16 : 0);
                auto launch = [&]() {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), lds, 0, rec, n, steps, out, coherent);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), lds, 0, rec, n, steps, out, coherent);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), lds, 0, rec, n, steps, out, coherent);
                    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), lds, 0, rec, n, steps, out, coherent);
                    if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), lds, 0, rec, n, steps, out, coherent);
                    if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), lds, 0, rec, n, steps, out, coherent);
                };
                launch();
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                launch();
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                double laneSteps = (double)blocks * 256 * steps;
                printf("%s %-13s blocks %4d (%.1f waves/SIMD): %7.1f us, %.2f ns per wave-step, %.1f G lane-steps/s\n", coherent ? "coherent " : "divergent", names[mode], blocks,
                       blocks * 4 / 1024.0, ms * 1e3, ms * 1e6 / steps, laneSteps / ms / 1e6);
            }
    return 0;
}
