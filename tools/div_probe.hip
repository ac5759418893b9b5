// div_probe.hip -- is the shared-reciprocal quotient of mr_traverse.h (tri_test) bit for bit the compiler's IEEE division?
//
// Triangle::intersect divides three numerators by one denominator (Triangle.cpp:150-156).  The compiler expands every `/` into
// v_div_scale x2, v_rcp, four fma, v_mul, v_div_fmas, v_div_fixup.  When neither operand needs scaling (|x| in [2^-60, 2^60]) the
// scale steps return their inputs, v_div_fmas is a plain fma and v_div_fixup returns its first operand: the quotient is
//     r1 = fma(fma(-b, rcp(b), 1), rcp(b), rcp(b));  q = a * r1;  q1 = fma(fma(-b, q, a), r1, q);  fma(fma(-b, q1, a), r1, q1)
// and r1 depends on b alone -- three quotients can share it.  This probe compares that sequence with a / b on random operands
// of that range (uniform bit patterns: every exponent equally likely, plus near-tie constructions a = q * b +- few ulp) and
// reports every mismatch.   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/div_probe.hip -o div_probe && ./div_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
// a float with a uniformly random sign, an exponent in [-60, 60] and a random mantissa
__device__ __forceinline__ float in_range(uint32_t h) {
    const uint32_t e = 127u - 60u + (pcg(h ^ 0x9e3779b9u) % 121u);
    return __uint_as_float((h & 0x80000000u) | (e << 23) | (pcg(h) & 0x007FFFFFu));
}
__device__ __forceinline__ float shared_quot(float a, float b, float r1) {
    const float q = a * r1;
    const float q1 = fmaf(fmaf(-b, q, a), r1, q);
    return fmaf(fmaf(-b, q1, a), r1, q1);
}
__global__ void probe(unsigned long long n, unsigned seed, unsigned long long *bad, float *first_bad) {
    unsigned long long mine = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t h = pcg((uint32_t)i ^ seed) ^ (uint32_t)(i >> 32);
        const float b = in_range(h);
        float a = in_range(pcg(h + 1u));
        if ((i & 3ull) == 3ull) {                      // a quarter of the cases: numerators next to a rounding tie of the quotient
            const float q = in_range(pcg(h + 2u));
            const float p = q * b;                     // a ~ q * b, then a few ulp around it
            const uint32_t bits = __float_as_uint(p) + (pcg(h + 3u) % 5u) - 2u;
            const uint32_t e = (bits >> 23) & 0xFFu;
            if (e >= 67u && e <= 187u) a = __uint_as_float(bits);
        }
        const float r0 = __builtin_amdgcn_rcpf(b);
        const float r1 = fmaf(fmaf(-b, r0, 1.0f), r0, r0);
        const float want = a / b, got = shared_quot(a, b, r1);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            if (mine == 0 && atomicAdd(bad, 0ull) == 0ull) { first_bad[0] = a; first_bad[1] = b; first_bad[2] = want; first_bad[3] = got; }
            mine++;
        }
    }
    if (mine) atomicAdd(bad, mine);
}

int main() {
    unsigned long long *d_bad; float *d_first;
    CHECK(hipMalloc(&d_bad, 8)); CHECK(hipMalloc(&d_first, 16));
    unsigned long long total = 0, bad_total = 0;
    for (unsigned seed = 1; seed <= 8; seed++) {
        CHECK(hipMemset(d_bad, 0, 8));
        const unsigned long long n = 1ull << 32;
        hipLaunchKernelGGL(probe, dim3(256 * 32), dim3(256), 0, 0, n, seed * 0x85ebca6bu, d_bad, d_first);
        CHECK(hipDeviceSynchronize());
        unsigned long long bad = 0; float f[4];
        CHECK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(f, d_first, 16, hipMemcpyDeviceToHost));
        total += n; bad_total += bad;
        printf("seed %u: %llu operand pairs, %llu mismatches", seed, n, bad);
        if (bad) printf("  (first: %a / %a = %a, shared sequence %a)", f[0], f[1], f[2], f[3]);
        printf("\n"); fflush(stdout);
    }
    printf("%s: %llu pairs, %llu mismatches\n", bad_total ? "MISMATCH" : "OK", total, bad_total);
    return bad_total ? 1 : 0;
}
