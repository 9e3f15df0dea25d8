// Exhaustive check: candidate fast reciprocal vs the compiler's IEEE 1.0f / x over all 2^32 bit patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float rcp_fast(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return r;
}
__device__ __forceinline__ float rcp_fast2(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return r;
}
__global__ void check(unsigned long long* bad, unsigned long long* bad2, uint32_t* firstBad) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t v = i; v < (1ull << 32); v += (uint64_t)gridDim.x * blockDim.x) {
        float x = __uint_as_float((uint32_t)v);
        float a = 1.0f / x;
        float b = rcp_fast(x), c = rcp_fast2(x);
        bool nanA = a != a;
        const uint32_t ex = ((uint32_t)v >> 23) & 0xffu;
        const bool inRange = ex >= 1 && ex <= 252;  // normal, and the reciprocal is normal too
        if (inRange && !(nanA ? (b != b) : __float_as_uint(a) == __float_as_uint(b))) { unsigned long long k = atomicAdd(bad, 1ull); if (k < 16) firstBad[k] = (uint32_t)v; }
        if (!(nanA ? (c != c) : __float_as_uint(a) == __float_as_uint(c))) { unsigned long long k = atomicAdd(bad2, 1ull); if (k < 16) firstBad[16 + k] = (uint32_t)v; }
    }
}
int main() {
    unsigned long long *bad, *bad2; uint32_t* fb;
    hipMalloc(&bad, 8); hipMalloc(&bad2, 8); hipMalloc(&fb, 128);
    hipMemset(bad, 0, 8); hipMemset(bad2, 0, 8); hipMemset(fb, 0, 128);
    check<<<4096, 256>>>(bad, bad2, fb);
    unsigned long long h = 0, h2 = 0; uint32_t hf[32];
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&h2, bad2, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, fb, 128, hipMemcpyDeviceToHost);
    printf("one-step mismatches with exponent field in [1,252]: %llu  two-step mismatches (all inputs): %llu\n", h, h2);
    for (int i = 0; i < 16 && i < (int)h; i++) printf(" one-step bad x=%08x\n", hf[i]);
    for (int i = 0; i < 16 && i < (int)h2; i++) printf(" two-step bad x=%08x\n", hf[16 + i]);
    return 0;
}
