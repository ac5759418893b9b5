// Does  q' = fma(fma(-q, d, x), inv, q)  with  inv = RN(1/d), q = RN(x * inv)  equal RN(x / d) on gfx950?
// (Markstein's correction step.)  Brute force over structured and random operand sets in the normal range.
// build: hipcc -O2 -ffp-contract=off --offload-arch=gfx950 tools/div_identity_probe.hip -o /tmp/div_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint32_t pcg(uint32_t x) { x = x * 747796405u + 2891336453u; x = ((x >> ((x >> 28) + 4)) ^ x) * 277803737u; return (x >> 22) ^ x; }

__device__ __forceinline__ bool check(float x, float d, unsigned long long *bad, float *ex) {
    const float inv = 1.0f / d;
    const float q = x * inv;
    const float r = __builtin_fmaf(-q, d, x);
    const float q2 = __builtin_fmaf(r, inv, q);
    const float ref = x / d;
    if (__float_as_uint(q2) != __float_as_uint(ref)) {
        if (atomicAdd(bad, 1ull) < 8) { ex[0] = x; ex[1] = d; ex[2] = q2; ex[3] = ref; }
        return false;
    }
    return true;
}

// mode 0: every mantissa of d (exponent 0) x 4096 random x; mode 1: random x, d with exponents in [-60, 60];
// mode 2: every mantissa of x (exponent 0) against 4096 random d; mode 3: d = all-ones-ish mantissas x random x
__global__ void probe(int mode, unsigned long long n, unsigned long long *bad, float *ex, uint32_t seed) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t h = pcg((uint32_t)i ^ seed), h2 = pcg(h + (uint32_t)(i >> 32) * 0x9e3779b9u);
        float x, d;
        if (mode == 0) {
            d = __uint_as_float(0x3f800000u | (uint32_t)(i & 0x7fffffu));
            x = __uint_as_float((h & 0x807fffffu) | ((uint32_t)(100 + (h2 % 56)) << 23));
        } else if (mode == 1) {
            x = __uint_as_float((h & 0x807fffffu) | ((uint32_t)(67 + (h2 % 121)) << 23));
            d = __uint_as_float((h2 & 0x807fffffu) | ((uint32_t)(67 + (pcg(h2) % 121)) << 23));
        } else if (mode == 2) {
            x = __uint_as_float(0x3f800000u | (uint32_t)(i & 0x7fffffu));
            d = __uint_as_float((h & 0x807fffffu) | ((uint32_t)(100 + (h2 % 56)) << 23));
        } else {
            d = __uint_as_float(0x3f800000u | (0x7fffffu - (uint32_t)(i & 0xfffu)));
            x = __uint_as_float((h & 0x807fffffu) | ((uint32_t)(100 + (h2 % 56)) << 23));
        }
        check(x, d, bad, ex);
    }
}

int main() {
    unsigned long long *bad; float *ex;
    hipMalloc(&bad, 8); hipMalloc(&ex, 16);
    const unsigned long long sizes[4] = {(1ull << 23) * 4096ull, 1ull << 35, (1ull << 23) * 4096ull, 1ull << 32};
    for (int mode = 0; mode < 4; mode++) {
        hipMemset(bad, 0, 8);
        hipLaunchKernelGGL(probe, dim3(256 * 32), dim3(256), 0, 0, mode, sizes[mode], bad, ex, 168u + mode);
        unsigned long long hb = 0; float he[4] = {0, 0, 0, 0};
        hipDeviceSynchronize();
        hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 16, hipMemcpyDeviceToHost);
        printf("mode %d: %llu pairs, %llu mismatches", mode, sizes[mode], hb);
        if (hb) printf("  e.g. x=%a d=%a corrected=%a ieee=%a", he[0], he[1], he[2], he[3]);
        printf("\n");
    }
    return 0;
}
