// mr_bounce.hip -- specular secondary rays ("next" row f4 of SURVEY.md section 8): the recursion of
// Scene::traceScene (Scene.cpp:270-346) unrolled into wavefront bounces.
//   light_scale_kernel      what Phong::shade does with the shadow hit (Phong.cpp:97-113): opaque occluder -> 0,
//                           refractive occluder -> dot(N, l) (0 if negative or < epsilon), no occluder -> 1
//   shade_accumulate_kernel Phong::shade per ray with per-triangle materials, times the ray's path weight, added to
//                           its pixel (float atomics: the order of additions is not reproducible for multi-bounce
//                           frames; the single-bounce path of mr_shade_direct stays deterministic)
//   children_kernel         Ray::reflect / getReflectionCoefficient / refract (Ray.h:143-243) for every hit on a
//                           reflective or refractive material: up to three children per ray, each with
//                           weight * (specular | transmission*Rs | transmission*(1-Rs)) (Scene.cpp:302-336), written
//                           compacted by wave64 ballots + prefix sums
#include <hip/hip_runtime.h>

#include "mr_internal.h"
#include "mr_recursion.h"

namespace mr {
namespace {

using namespace rec;

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void light_scale_kernel(MeshMat m, const mr_ray *shadow_rays, const mr_hit *shadow_hits,
                                                             const uint32_t *src, const unsigned long long *count,
                                                             unsigned long long max_n, float *light_scale) {
    unsigned long long n = *count;
    if (n > max_n) n = max_n;
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < n; k += stride) {
        const float4 h = reinterpret_cast<const float4 *>(shadow_hits)[k];
        float scale = 1.0f;
        if (__float_as_uint(h.y) != MR_MISS)
            scale = light_scale_of(m, reinterpret_cast<const float4 *>(shadow_rays)[2 * k],
                                   reinterpret_cast<const float4 *>(shadow_rays)[2 * k + 1], h);
        light_scale[src[k]] = scale;
    }
}

struct AccumArgs {
    MeshMat m;
    const mr_ray *rays;
    const mr_hit *hits;
    const float *weights;         // rgb per ray or NULL (= 1)
    const uint32_t *pixels;       // pixel per ray or NULL (= ray index / spp)
    const float *light_scale;     // per ray
    LightArgs lt;
    float inv_spp;
    uint32_t spp;
    unsigned long long n;
    float *rgb;
};

__global__ __launch_bounds__(kBlock) void shade_accumulate_kernel(AccumArgs a) {
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    const unsigned long long n_round = (a.n + 63ull) & ~63ull;                            // whole waves: accumulate_runs shuffles
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < n_round; k += stride) {
        float v[3] = {0.f, 0.f, 0.f};
        uint32_t pix = 0xFFFFFFFFu;
        if (k < a.n) {
            pix = a.pixels ? a.pixels[k] : (uint32_t)(k / a.spp);
            const float4 h = reinterpret_cast<const float4 *>(a.hits)[k];
            const uint32_t prim = __float_as_uint(h.y);
            const float scale = prim != MR_MISS ? a.light_scale[k] : 0.0f;                // a miss: m_bgColor = 0 contributes nothing
            if (scale != 0.0f) {
                float P[3], N[3], diffuse[3], highlight, out[3];
                surface_point(a.m, a.rays, k, h, P, N);
                const float4 rb = reinterpret_cast<const float4 *>(a.rays)[2 * k + 1];
                phong_terms(a.lt, material_of(a.m, prim), P, N, rb.x, rb.y, rb.z, diffuse, highlight);
                phong_combine(diffuse, highlight, scale, out);
                float w[3] = {1.f, 1.f, 1.f};
                if (a.weights) { w[0] = a.weights[3 * k]; w[1] = a.weights[3 * k + 1]; w[2] = a.weights[3 * k + 2]; }
                for (int c = 0; c < 3; c++) v[c] = out[c] * w[c] * a.inv_spp;
            }
        }
        accumulate_runs(a.rgb, pix, v[0], v[1], v[2]);
    }
}

struct BounceArgs {
    MeshMat m;
    const mr_ray *rays;
    const mr_hit *hits;
    const float *weights;
    const uint32_t *pixels;
    const uint32_t *ids;          // path tracing: stable id per ray (NULL: the ray's index); children get ids derived from it
    uint32_t spp, hbase, bounce, kinds;
    unsigned long long n;
    ChildQueue out;
};

// PATH = false: Ray::reflect / getReflectionCoefficient / refract (Ray.h:143-243) for every hit on a reflective or
// refractive material, up to three children per ray.  PATH = true: the PATH_TRACING build of those generators plus
// Ray::random's diffuse bounce (an EXTENSION: the reference defines Ray::random but traceScene at HEAD never calls it,
// SURVEY.md section 8d, config 3), up to four.
//
// A workgroup takes chunks of kGenIter * kBlock rays: it first lists the rays of the chunk that have children at all (a hit
// on a material with the wanted terms) in LDS, in ray order, counts their children, reserves the chunk's output slots with
// one atomic, then generates from the list with full waves.  Run straight
// over the rays the generators work at the hit rate of the queue -- the diffuse bounce rays of an open scene hit something
// 2 % of the time, nearly every wave still holds a hit, and the launch cost what the dense first level costs.
constexpr int kGenIter = 8;

#ifndef MIRO_CHILDREN_WAVES
#define MIRO_CHILDREN_WAVES 5      /* 90 registers, no spill; unconstrained the scheduler took 160 (3 waves) for the same speed */
#endif
template <bool PATH>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PATH ? MIRO_CHILDREN_WAVES : 1, 8))) void children_kernel(BounceArgs a) {
    __shared__ unsigned s_list[kGenIter * kBlock];            // offsets into the chunk
    __shared__ unsigned s_cnt[kGenIter][kBlock / 64];
    __shared__ unsigned s_slot[kGenIter * kBlock];            // per listed ray: (children of earlier lanes of its wave << 3) | its own
    __shared__ unsigned long long s_chunk_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long chunk = (unsigned long long)kBlock * kGenIter;
    const unsigned long long n_chunks = (a.n + chunk - 1) / chunk;
    for (unsigned long long c = blockIdx.x; c < n_chunks; c += gridDim.x) {      // uniform per workgroup
        const unsigned long long base = c * chunk;
        unsigned long long wants[kGenIter];
#pragma unroll
        for (int it = 0; it < kGenIter; it++) {
            const unsigned long long k = base + (unsigned long long)it * kBlock + threadIdx.x;
            bool want = false;
            if (k < a.n) {
                const uint32_t prim = __float_as_uint(reinterpret_cast<const float4 *>(a.hits)[k].y);
                if (prim != MR_MISS) {
                    const float *mt = material_of(a.m, prim);
                    want = (any_pos(mt + 3) && (!PATH || (a.kinds & 1u))) || (any_pos(mt + 6) && (!PATH || (a.kinds & 2u))) ||
                           (PATH && any_pos(mt) && (a.kinds & 4u));
                }
            }
            wants[it] = __ballot(want);
            if (lane == 0) s_cnt[it][wave] = (unsigned)__popcll(wants[it]);
        }
        __syncthreads();
        unsigned listed = 0;
#pragma unroll
        for (int it = 0; it < kGenIter; it++) {
            unsigned at = listed;
            for (int w = 0; w < kBlock / 64; w++) {
                const unsigned cw = s_cnt[it][w];
                if (w < wave) at += cw;
                listed += cw;
            }
            if ((wants[it] >> lane) & 1ull)
                s_list[at + (unsigned)__popcll(wants[it] & ((1ull << lane) - 1ull))] = (unsigned)(it * kBlock) + threadIdx.x;
        }
        __syncthreads();
        // Two passes over the list, so that the whole chunk takes ONE reservation: a single counter word drains ~88 atomics
        // per microsecond, and one atomic per 256 listed rays (65 k for the 16.8 M rays of a bunny frame) was what the launch
        // took -- 0.75 ms of atomics around 0.4 ms of work.
        // pass 1: how many children each listed ray has.  Without refraction that is the material alone; a refractive hit
        // needs its Fresnel term (the hit point is rebuilt again in pass 2 -- for those rays only).
        const int rounds = (int)((listed + kBlock - 1) / kBlock);
        for (int it = 0; it < kGenIter; it++) {              // (not unrolled: the generators would be inlined eight times)
            if (it >= rounds) { if (lane == 63) s_cnt[it][wave] = 0; continue; }
            const unsigned j = (unsigned)it * kBlock + threadIdx.x;
            unsigned cnt = 0;
            if (j < listed) {
                const unsigned long long k = base + s_list[j];
                const float4 h = reinterpret_cast<const float4 *>(a.hits)[k];
                const uint32_t prim = __float_as_uint(h.y);
                const float *mt = material_of(a.m, prim);
                const bool refl = any_pos(mt + 3) && (!PATH || (a.kinds & 1u)), refr = any_pos(mt + 6) && (!PATH || (a.kinds & 2u));
                const bool diff = PATH && any_pos(mt) && (a.kinds & 4u);
                cnt = (refl ? 1u : 0u) + (diff ? 1u : 0u);
                if (refr) {
                    ChildGen<PATH> g;
                    bool emit[4];
                    g.mt = mt;
                    surface_point(a.m, a.rays, k, h, g.P, g.N);
                    const float4 rb = reinterpret_cast<const float4 *>(a.rays)[2 * k + 1];
                    g.d[0] = rb.x; g.d[1] = rb.y; g.d[2] = rb.z;
                    g.plan(refl, refr, diff, emit);
                    cnt += 1u + (emit[1] ? 1u : 0u);
                }
            }
            unsigned incl = cnt;
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned up = __shfl_up(incl, off, 64);
                if (lane >= off) incl += up;
            }
            if (j < listed) s_slot[j] = ((incl - cnt) << 3) | cnt;      // children of earlier lanes of my wave | my own
            if (lane == 63) s_cnt[it][wave] = incl;          // s_cnt is free again: the list is complete
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned tot = 0;
            for (int it = 0; it < kGenIter; it++)
                for (int w = 0; w < kBlock / 64; w++) tot += s_cnt[it][w];
            s_chunk_base = tot ? atomicAdd(a.out.count, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        // pass 2: generate, every ray's children next to one another
        unsigned long long round_base = s_chunk_base;
        for (int it = 0; it < rounds; it++) {
            unsigned before = 0, round_total = 0;
            for (int w = 0; w < kBlock / 64; w++) {
                const unsigned cw = s_cnt[it][w];
                if (w < wave) before += cw;
                round_total += cw;
            }
            const unsigned j = (unsigned)it * kBlock + threadIdx.x;
            const unsigned mine = j < listed ? s_slot[j] : 0u;
            if (mine & 7u) {
                const unsigned long long k = base + s_list[j];
                ChildGen<PATH> g;
                bool emit[4] = {false, false, false, false};
                const float4 h = reinterpret_cast<const float4 *>(a.hits)[k];
                const uint32_t prim = __float_as_uint(h.y);
                g.mt = material_of(a.m, prim);
                const bool refl = any_pos(g.mt + 3) && (!PATH || (a.kinds & 1u)), refr = any_pos(g.mt + 6) && (!PATH || (a.kinds & 2u));
                const bool diff = PATH && any_pos(g.mt) && (a.kinds & 4u);
                surface_point(a.m, a.rays, k, h, g.P, g.N);
                const float4 rb = reinterpret_cast<const float4 *>(a.rays)[2 * k + 1];
                g.d[0] = rb.x; g.d[1] = rb.y; g.d[2] = rb.z;
                g.w0[0] = g.w0[1] = g.w0[2] = 1.f;
                if (a.weights) { g.w0[0] = a.weights[3 * k]; g.w0[1] = a.weights[3 * k + 1]; g.w0[2] = a.weights[3 * k + 2]; }
                const uint32_t pix = a.pixels ? a.pixels[k] : (uint32_t)(k / a.spp);
                uint32_t id = 0;
                if (PATH) {
                    id = a.ids ? a.ids[k] : (uint32_t)k;
                    g.hray = pcg32(a.hbase ^ id) + a.bounce * 4u;
                }
                g.plan(refl, refr, diff, emit);
                write_children_at<PATH>(a.out, g, emit, pix, id, round_base + before + (mine >> 3));
            }
            round_base += round_total;
        }
        __syncthreads();                                                             // s_list / s_cnt are reused by the next chunk
    }
}

inline unsigned grid_for(unsigned long long n) {
    unsigned long long blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 256ull * 32ull) blocks = 256ull * 32ull;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

mr_status launch_shade_accumulate(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, const float *d_weights,
                                  const uint32_t *d_pixels, unsigned long long n, const mr_ray *d_shadow_rays,
                                  const mr_hit *d_shadow_hits, const uint32_t *d_shadow_src,
                                  const unsigned long long *d_shadow_count, float *d_light_scale, const mr_light &light,
                                  uint32_t spp, float *d_rgb, hipStream_t stream) {
    if (n == 0) return MR_OK;
    const MeshMat m = mesh_of(ds);
    hipLaunchKernelGGL(light_scale_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, m, d_shadow_rays, d_shadow_hits,
                       d_shadow_src, d_shadow_count, n, d_light_scale);
    MR_HIP_CHECK(hipGetLastError());
    AccumArgs a;
    a.m = m; a.rays = d_rays; a.hits = d_hits; a.weights = d_weights; a.pixels = d_pixels; a.light_scale = d_light_scale;
    for (int c = 0; c < 3; c++) { a.lt.L[c] = light.position[c]; a.lt.color[c] = light.color[c]; }
    a.lt.wattage = light.wattage; a.spp = spp; a.inv_spp = 1.0f / (float)spp; a.n = n; a.rgb = d_rgb;
    hipLaunchKernelGGL(shade_accumulate_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, a);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

mr_status launch_secondary_rays(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, const float *d_weights,
                                const uint32_t *d_pixels, unsigned long long n, uint32_t spp, mr_ray *d_out_rays,
                                float *d_out_weights, uint32_t *d_out_pixels, unsigned long long *d_count,
                                unsigned long long out_capacity, uint8_t *d_out_octants, hipStream_t stream) {
    MR_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), stream));
    if (n == 0) return MR_OK;
    BounceArgs a;
    a.m = mesh_of(ds); a.rays = d_rays; a.hits = d_hits; a.weights = d_weights; a.pixels = d_pixels; a.ids = nullptr;
    a.spp = spp; a.hbase = 0; a.bounce = 0; a.kinds = 0; a.n = n;
    a.out.rays = d_out_rays; a.out.weights = d_out_weights; a.out.pixels = d_out_pixels; a.out.ids = nullptr; a.out.count = d_count; a.out.capacity = out_capacity; a.out.octants = d_out_octants;
    hipLaunchKernelGGL(children_kernel<false>, dim3(grid_for((n + kGenIter - 1) / kGenIter)), dim3(kBlock), 0, stream, a);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

mr_status launch_path_rays(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, const float *d_weights,
                           const uint32_t *d_pixels, const uint32_t *d_ids, unsigned long long n, uint32_t spp, uint32_t seed,
                           uint32_t bounce, uint32_t kinds, mr_ray *d_out_rays, float *d_out_weights, uint32_t *d_out_pixels,
                           uint32_t *d_out_ids, unsigned long long *d_count, unsigned long long out_capacity, uint8_t *d_out_octants, hipStream_t stream) {
    MR_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), stream));
    if (n == 0) return MR_OK;
    BounceArgs a;
    a.m = mesh_of(ds); a.rays = d_rays; a.hits = d_hits; a.weights = d_weights; a.pixels = d_pixels; a.ids = d_ids;
    a.spp = spp; a.bounce = bounce; a.kinds = kinds; a.n = n;
    a.hbase = pcg32(seed);
    a.out.rays = d_out_rays; a.out.weights = d_out_weights; a.out.pixels = d_out_pixels; a.out.ids = d_out_ids; a.out.count = d_count; a.out.capacity = out_capacity; a.out.octants = d_out_octants;
    hipLaunchKernelGGL(children_kernel<true>, dim3(grid_for((n + kGenIter - 1) / kGenIter)), dim3(kBlock), 0, stream, a);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

}  // namespace mr
