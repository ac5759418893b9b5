// ta_probe.hip -- what does a divergent 64-byte record fetch cost the vector memory pipe on gfx950?
// Every lane needs one 64-byte record out of a table (L2-resident, 1.5 MB like the atrium's node array) at a random
// index, `iters` times in a dependent chain (the next index comes from the record, as in a tree walk).
//   mode 0: the lane reads its own record with 4 x global_load_dwordx4
//   mode 1: quad-cooperative: load k brings the record of quad lane k, lane j reads its j-th 16 bytes (no exchange:
//           the pieces are only summed -- this measures the memory side alone)
//   mode 2: own record, 2 x dwordx4 (32 of the 64 bytes)     mode 3: own record, 1 x dwordx4
//   mode 4: own record through 16 x global_load_dword
// build: hipcc -O3 --offload-arch=gfx950 tools/ta_probe.hip -o cse168-raytracer_amd/build/ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ int quad_bcast(int v, int k) {
    switch (k) {
        case 0: return __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true);
        case 1: return __builtin_amdgcn_mov_dpp(v, 0x55, 0xf, 0xf, true);
        case 2: return __builtin_amdgcn_mov_dpp(v, 0xaa, 0xf, 0xf, true);
        default: return __builtin_amdgcn_mov_dpp(v, 0xff, 0xf, 0xf, true);
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float4 *table, unsigned n_rec, int iters, float *out) {
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_rec;
    const int j = threadIdx.x & 3;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        float4 a, b, c, d;
        if (MODE == 0) {
            const float4 *r = table + 4 * (size_t)idx;
            a = r[0]; b = r[1]; c = r[2]; d = r[3];
        } else if (MODE == 1) {
            a = table[4 * (size_t)quad_bcast((int)idx, 0) + j];
            b = table[4 * (size_t)quad_bcast((int)idx, 1) + j];
            c = table[4 * (size_t)quad_bcast((int)idx, 2) + j];
            d = table[4 * (size_t)quad_bcast((int)idx, 3) + j];
        } else if (MODE == 2) {
            const float4 *r = table + 4 * (size_t)idx;
            a = r[0]; b = r[1]; c = a; d = b;
        } else if (MODE == 3) {
            a = table[4 * (size_t)idx]; b = a; c = a; d = a;
        } else {
            const float *r = reinterpret_cast<const float *>(table + 4 * (size_t)idx);
            a = make_float4(r[0], r[1], r[2], r[3]); b = make_float4(r[4], r[5], r[6], r[7]);
            c = make_float4(r[8], r[9], r[10], r[11]); d = make_float4(r[12], r[13], r[14], r[15]);
        }
        acc += a.x + b.y + c.z + d.w;
        // next index: depends on the loaded data (w of the first piece holds a random index as float bits)
        idx = (__float_as_uint(a.w) + (unsigned)it * 7919u + idx * 31u) % n_rec;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const unsigned n_rec = argc > 1 ? atoi(argv[1]) : 23350;       // the atrium's inner nodes
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    const unsigned lanes = 256u * 7u * 256u * 4u;                  // 4 chip-fulls of 7 workgroups per CU
    std::vector<float4> h(4 * (size_t)n_rec);
    srand(7);
    for (auto &v : h) { unsigned r = (unsigned)rand(); v = make_float4(1.f, 2.f, 3.f, 0.f); __builtin_memcpy(&v.w, &r, 4); }
    float4 *d_t; float *d_o;
    CK(hipMalloc(&d_t, h.size() * sizeof(float4))); CK(hipMalloc(&d_o, lanes * sizeof(float)));
    CK(hipMemcpy(d_t, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 5; mode++) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            switch (mode) {
                case 0: hipLaunchKernelGGL(probe<0>, dim3(lanes / 256), dim3(256), 0, 0, d_t, n_rec, iters, d_o); break;
                case 1: hipLaunchKernelGGL(probe<1>, dim3(lanes / 256), dim3(256), 0, 0, d_t, n_rec, iters, d_o); break;
                case 2: hipLaunchKernelGGL(probe<2>, dim3(lanes / 256), dim3(256), 0, 0, d_t, n_rec, iters, d_o); break;
                case 3: hipLaunchKernelGGL(probe<3>, dim3(lanes / 256), dim3(256), 0, 0, d_t, n_rec, iters, d_o); break;
                default: hipLaunchKernelGGL(probe<4>, dim3(lanes / 256), dim3(256), 0, 0, d_t, n_rec, iters, d_o); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double fetches = (double)lanes * iters;
        printf("TA_PROBE mode %d: %8.3f ms  %7.2f G record-fetches/s  = %6.1f cycles per wave-fetch per CU at 2.4 GHz (%u records, %d dependent fetches per lane)\n",
               mode, best, fetches / best / 1e6, best * 1e-3 * 2.4e9 * 256.0 / (fetches / 64.0), n_rec, iters);
    }
    return 0;
}
