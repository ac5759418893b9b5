// mr_level.hip -- ONE level of Scene::traceScene's recursion (Scene.cpp:270-346) for a queue of rays in ONE launch:
//
//   Scene::trace (Scene.cpp:278)  ->  Phong::shade: shadow ray towards the point light, Scene::trace, the occluder's
//   light scale, diffuse term + highlight (Phong.cpp:80-156), times the ray's path weight, added to its pixel  ->
//   the children of the next level: Ray::reflect / getReflectionCoefficient / refract (Scene.cpp:302-336), or their
//   PATH_TRACING build plus Ray::random, compacted into the next queue
//
// One ray per lane.  The batched pipeline runs a level as seven launches (mr_trace -> mr_gen_shadow_rays ->
// mr_trace_indirect -> light scale -> mr_shade_accumulate -> mr_gen_secondary_rays | mr_gen_path_rays) that pass hit
// records, shadow rays, their hits, source indices and light scales through HBM (100 bytes per ray written and read
// again); here the lane keeps the hit in registers, builds the shadow ray from it, traces it with the same LDS stack,
// shades, and emits its children.  HBM traffic per ray: the ray record in (32 B + weight, pixel, id), the children out.
// No buffer of the level's size exists besides the two queues.
//
// What it buys (profiles/r02_level_probe.log): the first level of a bunny 1024x1024x16 path trace takes 1.50 ms against
// the 1.84 ms of the seven batched launches.  What it does NOT buy: a level whose rays rarely hit anything (the diffuse
// bounce rays of the open bunny scene hit 2 % of the time) runs its second traversal and its generators at that lane
// utilisation, where the batched calls compact the shadow rays between launches -- 1.9 ms against 1.4 ms.  Gathering the
// hits of a workgroup into full waves through LDS before the second phase was built and measured (1.8 ms): the barrier
// per round couples four waves whose slowest lanes differ widely, and costs what the gathering saves.
//
// Every piece is the shared definition the batched kernels use (mr_traverse.h, mr_surface.h, mr_phong.h,
// mr_recursion.h): the children are the same bits as the batched generators' (tests/test_level.py compares the two
// queues as sets), the pixel sums differ only by the order of the float atomics.
#include <hip/hip_runtime.h>

#include "mr_internal.h"
#include "mr_phong.h"
#include "mr_recursion.h"
#include "mr_traverse.h"

namespace mr {
namespace {

using namespace rec;

struct LevelArgs {
    TraceParams tp;              // scene arrays, root box; tp.rays = the queue, tp.n its length
    MeshMat m;
    LightArgs lt;
    const float *weights;        // rgb per ray or NULL (= 1)
    const uint32_t *pixels;      // pixel per ray or NULL (= ray index / spp)
    const uint32_t *ids;         // path tracing: stable ray ids or NULL (= ray index)
    uint32_t spp, hbase, bounce, kinds;
    float inv_spp;
    float *rgb;
    ChildQueue out;              // out.rays == NULL: the last level, no children
    unsigned long long *counts;  // optional: [0] += rays traced, [1] += shadow rays traced
};

// VAR: the traversal variant of trace_ray (mr_traverse.h) for both rays; CHILDREN: 0 none, 1 specular, 2 path tracing.
// 80 registers (6 waves per SIMD): the generators spill a few values to scratch rather than costing the two traversals
// their occupancy.  The path-tracing form gets 96 (5 waves): its double arithmetic also keeps ~280 constants' worth of
// SGPRs spilled into VGPR lanes, and with 77-94 spilled VGPRs on top of that under an 80-register cap a build of this
// kernel faulted on the GPU; at 96 it spills <= 18 (tests/test_build_budget.py keeps every kernel below 48).
#ifndef MIRO_LEVEL_PATH_WAVES
#define MIRO_LEVEL_PATH_WAVES 5
#endif
template <int VAR, int CHILDREN>
__global__ __launch_bounds__(kTraceBlock) __attribute__((amdgpu_waves_per_eu(CHILDREN == 2 ? MIRO_LEVEL_PATH_WAVES : 6, 8))) void level_kernel(LevelArgs a) {
    extern __shared__ int s_stack[];                  // [stack_depth][kTraceBlock]
    __shared__ unsigned s_shadow_rays[kTraceBlock / 64];
    const int tid = threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * kTraceBlock;
    const unsigned long long n = a.tp.n;
    const unsigned long long n_round = (n + (unsigned long long)kTraceBlock - 1ull) / kTraceBlock * kTraceBlock;   // whole workgroups
    constexpr bool kObj = (VAR & 32) != 0;
    constexpr bool kPath = CHILDREN == 2;
    Stats st = {0ull, 0ull};
    unsigned my_shadow_rays = 0;

    for (unsigned long long k = (unsigned long long)xcd_block_id() * kTraceBlock + tid; k < n_round; k += stride) {
        const bool live = k < n;
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = make_float4(1.f, 1.f, 1.f, -1.f);
        if (live) {                                      // level->d_order: lane k works on ray d_order[k] of the queue
            const unsigned long long s0 = a.tp.order ? (unsigned long long)a.tp.order[k] : k;
            ra = reinterpret_cast<const float4 *>(a.tp.rays)[2 * s0]; rb = reinterpret_cast<const float4 *>(a.tp.rays)[2 * s0 + 1];
        }
        // ---- the ray (Scene.cpp:278)
        mr_hit h;
        {
            RayRegs r;
            ray_setup(r, ra, rb);
            Lane L;
            int plane_hit;
            trace_ray<true, false, false, VAR>(a.tp, r, rb.w, live, L, plane_hit, s_stack, tid, st);
            h = make_hit<kObj>(a.tp, L, plane_hit, rb.w);
        }
        const bool hit = live && h.prim != MR_MISS;

        // ---- Phong::shade.  The unoccluded terms are computed BEFORE the shadow ray is traced: four values stay live
        // across the second traversal instead of the hit point, the normal and the material pointer.
        float diffuse[3] = {0.f, 0.f, 0.f}, highlight = 0.0f;
        float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = make_float4(1.f, 1.f, 1.f, -1.f);
        if (hit) {
            float P[3], N[3];
            surface_point_od(a.m, ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, h.t, h.prim, h.beta, h.gamma, P, N);
            shadow_ray_of(P, a.lt.L[0], a.lt.L[1], a.lt.L[2], sa, sb);
            phong_terms(a.lt, material_of(a.m, h.prim), P, N, rb.x, rb.y, rb.z, diffuse, highlight);
            my_shadow_rays++;
        }
        float4 sh;
        {
            RayRegs r;
            ray_setup(r, sa, sb);
            Lane L;
            int plane_hit;
            trace_ray<true, false, false, VAR>(a.tp, r, sb.w, hit, L, plane_hit, s_stack, tid, st);      // closest hit: the occluder matters
            const mr_hit hs = make_hit<kObj>(a.tp, L, plane_hit, sb.w);
            sh = *reinterpret_cast<const float4 *>(&hs);
        }
        asm volatile("" ::: "memory");                   // the index is read again rather than kept in registers across the traversals
        unsigned long long src = k;
        if (live && a.tp.order) src = a.tp.order[k];
        uint32_t pix = 0xFFFFFFFFu;
        float w0[3] = {1.f, 1.f, 1.f}, v[3] = {0.f, 0.f, 0.f};
        if (live) pix = a.pixels ? a.pixels[src] : (uint32_t)(src / a.spp);
        if (hit) {
            if (a.weights) { w0[0] = a.weights[3 * src]; w0[1] = a.weights[3 * src + 1]; w0[2] = a.weights[3 * src + 2]; }
            float out[3];
            phong_combine(diffuse, highlight, light_scale_of(a.m, sa, sb, sh), out);
            for (int c = 0; c < 3; c++) v[c] = out[c] * w0[c] * a.inv_spp;
        }
        accumulate_runs(a.rgb, pix, v[0], v[1], v[2]);

        // ---- the children (Scene.cpp:302-336)
        if (CHILDREN != 0) {
            ChildGen<kPath> g;
            bool emit[4] = {false, false, false, false};
            uint32_t id = 0;
            if (hit) {
                g.mt = material_of(a.m, h.prim);
                const bool refl = any_pos(g.mt + 3) && (!kPath || (a.kinds & 1u)), refr = any_pos(g.mt + 6) && (!kPath || (a.kinds & 2u));
                const bool diff = kPath && any_pos(g.mt) && (a.kinds & 4u);
                if (refl || refr || diff) {
                    surface_point_od(a.m, ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, h.t, h.prim, h.beta, h.gamma, g.P, g.N);
                    g.d[0] = rb.x; g.d[1] = rb.y; g.d[2] = rb.z;
                    g.w0[0] = w0[0]; g.w0[1] = w0[1]; g.w0[2] = w0[2];
                    if (kPath) {
                        id = a.ids ? a.ids[src] : (uint32_t)src;
                        g.hray = pcg32(a.hbase ^ id) + a.bounce * 4u;
                    }
                    g.plan(refl, refr, diff, emit);
                }
            }
            write_children<kTraceBlock, kPath>(a.out, g, emit, pix, id);
        }
    }

    if (a.counts) {
        unsigned w = my_shadow_rays;
        for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off, 64);
        if ((tid & 63) == 0) s_shadow_rays[tid >> 6] = w;
        __syncthreads();
        if (tid == 0) {
            unsigned long long tot = 0;
            for (int j = 0; j < kTraceBlock / 64; j++) tot += s_shadow_rays[j];
            if (tot) atomicAdd(&a.counts[1], tot);
            if (blockIdx.x == 0) atomicAdd(&a.counts[0], n);
        }
    }
}

template <int VAR, int CHILDREN>
mr_status launch_level_t(const LevelArgs &a, hipStream_t stream) {
    const size_t lds = (size_t)a.tp.stack_depth * kTraceBlock * sizeof(int);
    if (lds > 150 * 1024) return fail(MR_ERR_INVALID, "traversal stack of depth %d does not fit in LDS", a.tp.stack_depth);
    if (lds > 48 * 1024)
        MR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&level_kernel<VAR, CHILDREN>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned long long blocks = (a.tp.n + kTraceBlock - 1) / kTraceBlock;
    if (blocks > (unsigned long long)kTraceGridCap) blocks = kTraceGridCap;
    hipLaunchKernelGGL((level_kernel<VAR, CHILDREN>), dim3((unsigned)blocks), dim3(kTraceBlock), lds, stream, a);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

template <int VAR>
mr_status launch_level_c(const LevelArgs &a, uint32_t children, hipStream_t stream) {
    switch (children) {
        case MR_LEVEL_LAST: return launch_level_t<VAR, 0>(a, stream);
        case MR_LEVEL_SPECULAR: return launch_level_t<VAR, 1>(a, stream);
        default: return launch_level_t<VAR, 2>(a, stream);
    }
}

}  // namespace

mr_status launch_level(const DeviceScene &ds, const mr_level_desc &ld, const mr_ray *d_rays, const float *d_weights,
                       const uint32_t *d_pixels, const uint32_t *d_ids, unsigned long long n, float *d_rgb, mr_ray *d_out_rays,
                       float *d_out_weights, uint32_t *d_out_pixels, uint32_t *d_out_ids, unsigned long long *d_out_count,
                       unsigned long long *d_counts, hipStream_t stream) {
    if (ld.children != MR_LEVEL_LAST) MR_HIP_CHECK(hipMemsetAsync(d_out_count, 0, sizeof(unsigned long long), stream));
    if (n == 0) return MR_OK;
    LevelArgs a;
    TraceParams &p = a.tp;
    p.nodes = ds.nodes; p.tris = ds.tris; p.tri_prim = ds.tri_prim; p.leaf_cnt_ext = ds.leaf_cnt_ext;
    for (int c = 0; c < 3; c++) { p.root_lo[c] = ds.root_lo[c]; p.root_hi[c] = ds.root_hi[c]; }
    p.root_ref = ds.root_ref;
    p.stack_depth = (int32_t)ds.stack_depth;
    p.rays = d_rays; p.hits = nullptr; p.n = n; p.n_dev = nullptr; p.stats = nullptr;
    p.planes = ds.planes; p.n_planes = ds.n_planes; p.n_spheres = ds.n_spheres;
    p.work_counter = nullptr; p.order = ld.d_order;
    a.m = mesh_of(ds);
    for (int c = 0; c < 3; c++) { a.lt.L[c] = ld.light.position[c]; a.lt.color[c] = ld.light.color[c]; }
    a.lt.wattage = ld.light.wattage;
    a.weights = d_weights; a.pixels = d_pixels; a.ids = d_ids;
    a.spp = ld.spp; a.inv_spp = 1.0f / (float)ld.spp; a.hbase = pcg32(ld.seed); a.bounce = ld.bounce; a.kinds = ld.path_kinds;
    a.rgb = d_rgb;
    a.out.rays = d_out_rays; a.out.weights = d_out_weights; a.out.pixels = d_out_pixels; a.out.ids = d_out_ids; a.out.count = d_out_count;
    a.out.capacity = ((unsigned long long)ld.out_capacity_hi << 32) | ld.out_capacity_lo;
    a.out.octants = ld.d_out_octants;
    a.counts = d_counts;

    const bool product = ld.flags & MR_MATH_PRODUCT, vote = ld.flags & MR_TRACE_INCOHERENT;
    if (ds.n_planes || ds.n_spheres) return product ? launch_level_c<43>(a, ld.children, stream) : launch_level_c<826>(a, ld.children, stream);
    if (vote) return product ? launch_level_c<73>(a, ld.children, stream) : launch_level_c<88>(a, ld.children, stream);
    return product ? launch_level_c<267>(a, ld.children, stream) : launch_level_c<1818>(a, ld.children, stream);
}

}  // namespace mr
