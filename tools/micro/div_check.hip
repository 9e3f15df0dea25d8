// Check of the three-instruction quotient the RIS project's box test uses (rt_kernels.hip: quotientInRange) against the
// compiler's IEEE division, on the ranges the kernel restricts it to: divisor exponent field 87..167 (2^-40 .. 2^40),
// numerator magnitude 2^-60 .. 2^59.  2^32 random pairs (exponents uniform over the ranges, random mantissas and signs),
// then every pairing of "hard" mantissas (all ones, all zeros, one bit either side, alternating patterns) at every
// exponent pair of the ranges' corners and middles.  Prints the number of differing results (expected: 0).
// Build as the library is built: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off div_check.hip -o div_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float rcpNormalRange(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ float quotientInRange(float a, float d, float r) {
    const float q0 = a * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q0, a), r, q0);
}
__device__ __forceinline__ uint32_t pcg(uint32_t& s) {
    s = s * 747796405u + 2891336453u;
    uint32_t r = ((s >> ((s >> 28) + 4)) ^ s) * 277803737u;
    return (r >> 22) ^ r;
}
__device__ void compare(float a, float d, unsigned long long* bad, uint32_t* first) {
    const float want = a / d, got = quotientInRange(a, d, rcpNormalRange(d));
    if (__float_as_uint(want) != __float_as_uint(got)) {
        unsigned long long k = atomicAdd(bad, 1ull);
        if (k < 8) { first[2 * k] = __float_as_uint(a); first[2 * k + 1] = __float_as_uint(d); }
    }
}
__global__ void randomPairs(unsigned long long* bad, uint32_t* first, uint32_t perThread) {
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    for (uint32_t i = 0; i < perThread; i++) {
        const uint32_t ra = pcg(s), rd = pcg(s), re = pcg(s);
        const uint32_t ea = 67u + re % 120u, ed = 87u + (re >> 16) % 81u;
        const float a = __uint_as_float((ra & 0x807fffffu) | (ea << 23)), d = __uint_as_float((rd & 0x807fffffu) | (ed << 23));
        compare(a, d, bad, first);
    }
}
__constant__ uint32_t HARD[16] = {0x000000u, 0x000001u, 0x000002u, 0x7fffffu, 0x7ffffeu, 0x400000u, 0x400001u, 0x3fffffu,
                                  0x555555u, 0x2aaaaau, 0x7ffff0u, 0x00000fu, 0x600000u, 0x1fffffu, 0x7f0000u, 0x00ffffu};
__global__ void hardPairs(unsigned long long* bad, uint32_t* first) {
    // one thread per (numerator exponent, divisor exponent); all 16 x 16 mantissa pairs, both signs of the numerator,
    // and for every hard divisor mantissa 4096 random numerator mantissas as well
    const uint32_t ea = 67u + blockIdx.x, ed = 87u + threadIdx.x;
    if (ea > 186u || ed > 167u) return;
    uint32_t s = ea * 977u + ed;
    for (int i = 0; i < 16; i++) {
        const float d = __uint_as_float(HARD[i] | (ed << 23));
        for (int j = 0; j < 16; j++) {
            compare(__uint_as_float(HARD[j] | (ea << 23)), d, bad, first);
            compare(__uint_as_float(HARD[j] | (ea << 23) | 0x80000000u), d, bad, first);
        }
        for (int j = 0; j < 4096; j++) compare(__uint_as_float((pcg(s) & 0x807fffffu) | (ea << 23)), d, bad, first);
    }
}
int main() {
    unsigned long long* bad; uint32_t* first;
    hipMalloc(&bad, 16); hipMalloc(&first, 64); hipMemset(bad, 0, 16); hipMemset(first, 0, 64);
    randomPairs<<<4096, 256>>>(bad, first, 4096);        // 2^32 pairs
    hardPairs<<<120, 81>>>(bad + 1, first);
    unsigned long long h[2]; uint32_t f[16];
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(f, first, 64, hipMemcpyDeviceToHost);
    printf("random pairs: %llu of 4294967296 differ; hard-mantissa pairs: %llu of %llu differ\n", h[0], h[1], 120ull * 81 * 16 * (32 + 4096));
    for (int i = 0; i < 8 && i < (int)(h[0] + h[1]); i++) printf(" differs: a=%08x d=%08x\n", f[2 * i], f[2 * i + 1]);
    return h[0] + h[1] ? 1 : 0;
}
