// mr_frame.hip -- the reference's per-pixel loop (Scene::raytraceImage, Scene.cpp:112-141) as ONE launch:
//
//   Camera::eyeRay (Camera.cpp:104-161)  ->  Scene::trace (Scene.cpp:278)  ->  Phong::shade's shadow ray towards the
//   point light (Phong.cpp:80-97)  ->  Scene::trace (Phong.cpp:97)  ->  Phong::shade's direct term (Phong.cpp:116-156)
//   ->  mean over the pixel's samples (Scene.cpp:126-139)
//
// One sample per lane.  The primary ray is generated in registers from the camera frame, traced, the shadow ray is
// built from the hit still in registers and traced by the same lane with the same LDS stack, the sample is shaded and
// the pixel's samples meet in an xor-butterfly.  No ray buffer exists: the batched pipeline (mr_gen_eye_rays ->
// mr_trace -> mr_gen_shadow_rays -> mr_trace_indirect -> mr_shade_direct) keeps 100 bytes per sample resident and moves
// 148 bytes per sample through HBM; this kernel writes 12 bytes per PIXEL (plus, on request, the two 16-byte hit
// records per sample, which is what parity tests compare).
//
// Every piece is the shared definition the batched kernels use (mr_eye.h, mr_traverse.h, mr_surface.h, mr_phong.h), so
// rays, hit records and pixels are the same bits as the batched pipeline's -- tests/test_frame.py compares them.
#include <hip/hip_runtime.h>

#include "mr_eye.h"
#include "mr_internal.h"
#include "mr_phong.h"
#include "mr_recursion.h"
#include "mr_surface.h"
#include "mr_tile.h"
#include "mr_traverse.h"

namespace mr {
namespace {

struct FrameArgs {
    EyeFrame eye;
    TraceParams tp;              // scene arrays, root box; rays / hits / n unused
    SurfacePtrs m;
    DirectLight lt;
    const float *mats;           // MAT kernels: the scene's material table (11 floats each) and the material of every object
    const uint32_t *prim_mat;
    mr_hit *hits;                // optional: primary hit record of sample k
    mr_hit *shadow_hits;         // optional: record of sample k's shadow ray (prim = MR_MISS, t = 0 when it has none)
    float *rgb;                  // [rows * W][3], window rows in band order, image order inside a row
    unsigned long long *counts;  // optional: [0] += primary rays, [1] += shadow rays traced by this launch
    // which chunks (kTraceBlock consecutive samples) a workgroup renders: see frame_schedule()
    uint32_t body_wgs, body_iters, tail_base, tail_chunks;      // tail_base = body_wgs * body_iters: the first tail chunk
    unsigned *tail_counter;      // hand-out counter of the tail chunks: 0 before the launch, left at 0 by its last reader
};

// the longest run of chunks one reading of the tail counter hands out: a quarter of a body workgroup's share, 1 ... 8
__device__ __forceinline__ unsigned tail_first_run(unsigned body_iters) {
    const unsigned q = body_iters >> 2;
    return q >= 8u ? 8u : q >= 4u ? 4u : q >= 2u ? 2u : 1u;
}
// A kernel argument read where it is used, every time (volatile): the tail's parameters are needed a few dozen times per launch,
// on a path a tenth of the workgroups take -- loaded once up front they would hold scalar registers through both traversals of
// every workgroup (the kernel already keeps 50 of them spilled in VGPR lanes; a first version that kept them lost 0.9 %).
template <typename T>
__device__ __forceinline__ T kernarg_now(size_t offset) {
    typedef const volatile T __attribute__((address_space(4))) *ptr_t;
    const char __attribute__((address_space(4))) *ka = (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    return *(ptr_t)(ka + offset);
}

// The launch's shape.  Workgroups b < body_wgs render chunks b', b' + body_wgs, ... (body_iters of them, b' = the XCD-aware
// permutation of b).  Frames of 60 000 chunks or more (1080p from 8 samples per pixel; 46 080 chunks: -4 %) keep their last tail_chunks chunks -- 6 % --
// out of that: kTailWgs further workgroups, the last to be dealt out, one per resident slot, PULL them in short runs from a
// counter until none is left.  Why (tools/wg_timeline.py, profiles/r03_wg_timeline.log): workgroups are dealt out in index order,
// seven per CU, every eighth to the same XCD whatever that XCD still has to do.  With 65 536 equal workgroups (8 chunks, 0.42 ms
// each on the bench frame) the last ones started at 97 % of the launch and the CUs drained for a whole workgroup lifetime, while
// the XCDs -- each renders its own regions of the image -- finished between 98 and 100 %: 18 % of the slots empty from 95 to
// 98 % of the launch, 87 % after that, 4.3 % of the launch's workgroup-slots unused in all.  Shorter workgroups throughout cost
// more than they return (profiles/r03_grid_ab.log).  A tail that whoever is free pulls from fills the drain and evens out the
// XCDs (slots unused: 2.0 %), and with it LONGER body workgroups pay: 32 chunks each (15 228 workgroups instead of 65 536) is
// +0.6 % over 8.  Bench frame 16.59 -> 16.93 Grays/s, 1080p x 16 spp 14.78 -> 15.26, BASELINE config 3 (bunny 1024^2 x 16) 41.0 ->
// 48.4; the 20-bunny scene at 16 spp loses 2.7 % (its dear chunks sit at the bottom of the image, i.e. in the tail); 4 spp and
// below are better off without (profiles/r03_tail_ab.log).  Every chunk is rendered by exactly one workgroup either way: the
// picture does not change.
#ifndef MIRO_TAIL_PERMILLE
#define MIRO_TAIL_PERMILLE 60
#endif
#ifndef MIRO_BODY_ITERS
#define MIRO_BODY_ITERS 32
#endif
#ifndef MIRO_TAIL_MIN_CHUNKS
#define MIRO_TAIL_MIN_CHUNKS 60000ull     /* 1080p x 8 spp, and an eighth of the bench frame (one of 8 ranks), are 64 800 */
#endif
constexpr uint32_t kTailWgs = 1792 * (256 / kTraceBlock);   // one per resident slot: 7 workgroups of 4 waves x 256 CUs
#ifndef MIRO_BODY_MIN_WGS
#define MIRO_BODY_MIN_WGS 14000
#endif
#ifndef MIRO_TAIL_PERMILLE_MID
#define MIRO_TAIL_PERMILLE_MID 60
#define MIRO_BODY_MIN_WGS_MID 14000
#endif
constexpr uint32_t kBodyMinWgs = MIRO_BODY_MIN_WGS;      // ~8 body workgroups per resident slot, at least
struct FrameShape { uint32_t body_wgs, body_iters, tail_chunks, grid; };
inline FrameShape frame_schedule(unsigned long long chunks) {
    FrameShape s;
    if (chunks < MIRO_TAIL_MIN_CHUNKS || MIRO_TAIL_PERMILLE == 0 || kTraceBlock != 256) {
        const unsigned long long cap = chunks < kFrameLargeChunks ? (unsigned long long)kTraceGridCap : (unsigned long long)kFrameGridCapLarge;
        s.body_wgs = (uint32_t)(chunks < cap ? chunks : cap);
        s.body_iters = (uint32_t)((chunks + s.body_wgs - 1) / s.body_wgs);
        s.tail_chunks = 0;
        s.grid = s.body_wgs;
        return s;
    }
    const bool mid = chunks < kFrameLargeChunks;      // below 2^18 chunks a launch is a few milliseconds: shorter shares, more tail
    const unsigned long long tail_target = chunks * (mid ? MIRO_TAIL_PERMILLE_MID : MIRO_TAIL_PERMILLE) / 1000;
    uint32_t m = MIRO_BODY_ITERS;
    while (m > 1 && (chunks - tail_target) / m < (mid ? (uint32_t)MIRO_BODY_MIN_WGS_MID : kBodyMinWgs)) m >>= 1;
    while ((unsigned long long)m * (unsigned long long)kFrameGridCapLarge < chunks) m <<= 1;
    s.body_iters = m;
    s.body_wgs = (uint32_t)((chunks - tail_target) / m);
    s.tail_chunks = (uint32_t)(chunks - (unsigned long long)s.body_wgs * m);
    s.grid = s.body_wgs + kTailWgs;
    return s;
}

// VAR: the traversal variant of trace_ray (mr_traverse.h) for both rays.
// SHADOW: 0 = the shadow ray is a closest-hit query as Phong.cpp:97; 1 = it stops at its first accepted hit (opaque scenes:
// same occlusion flag, hence the same picture); 2 = no shadow ray at all -- the reference's -DDISABLE_SHADOWS build
// (Phong.cpp:91), BASELINE config 2's "primary rays only".
// MAT: Phong::shade with the scene's per-object materials (mr_scene_set_materials) instead of the frame's uniform one:
// diffuse term and highlight from the hit's material (Phong.cpp:116-156) and light through a refractive occluder scaled by
// dot(N, l) of the occluder (Phong.cpp:99-113) -- the pieces of mr_recursion.h that mr_trace_level shades with.
#if defined(MIRO_WG_TIMES) && MIRO_TRACE_BLOCK == 256
// measurement build only (make VARIANT=_wgt FRAME_DEFS=-DMIRO_WG_TIMES; tools/wg_timeline.py): every workgroup leaves its start and
// end time (100 MHz constant clock), the CU it ran on and its XCD -- where the launch's idle VALU cycles sit
__device__ unsigned long long g_wg_times[4 * 131072];
#endif
template <int VAR, int SHADOW, bool MAT>
#ifndef MIRO_FRAME_WAVES
#define MIRO_FRAME_WAVES 7      /* 72 registers; 6 (80 registers, nothing spilled to scratch) is 6.5 % slower on the bench frame: profiles/r03_tail_ab.log */
#endif
__global__ __launch_bounds__(kTraceBlock) __attribute__((amdgpu_waves_per_eu(MIRO_FRAME_WAVES, 8))) void frame_kernel(FrameArgs a) {
#if defined(MIRO_WG_TIMES) && MIRO_TRACE_BLOCK == 256
    const unsigned long long wg_t0 = wall_clock64();
#endif
    extern __shared__ int s_stack[];                  // [stack_depth][kTraceBlock]
    __shared__ unsigned s_shadow_rays[kTraceBlock / 64];
    const int tid = threadIdx.x;
    const unsigned long long n = a.eye.n, n_round = (n + 63ull) & ~63ull;
    constexpr bool kObj = (VAR & 32) != 0;
    const uint32_t spp = a.eye.spp;
    const float inv_spp = 1.0f / (float)spp;
    Stats st = {0ull, 0ull};
    unsigned my_shadow_rays = 0;

    // body workgroups stride over their chunks as every trace kernel does; a tail workgroup (stride 0) asks the counter for its
    // next chunk instead -- values at or beyond tail_chunks mean "none left" (each tail workgroup reads exactly one such value)
    __shared__ unsigned s_pull, s_run[2];
    if (tid == 0) { s_run[0] = 0; s_run[1] = 0; }     // (read back by thread 0 only)
    unsigned long long idx, stride, limit;
    if (blockIdx.x < a.body_wgs) {
        const unsigned long long body_end = (unsigned long long)a.body_wgs * a.body_iters * kTraceBlock;
        idx = (unsigned long long)xcd_block_id_of(blockIdx.x, a.body_wgs, a.body_wgs >= (unsigned)kFrameGridCapLarge || a.tail_chunks != 0, a.tail_chunks != 0 ? 4096u : kXcdMinGrid) * kTraceBlock + tid;
        stride = (unsigned long long)a.body_wgs * kTraceBlock;
        limit = body_end < n_round ? body_end : n_round;
    } else {
        idx = 0; stride = 0; limit = n_round;
    }
    for (;; idx += stride) {
        if (__builtin_expect(stride == 0, 0)) {
            // One reading of the counter hands out a RUN of consecutive chunks: a quarter of a body workgroup's share at a time (at
            // most eight) for the first part of the tail, then half that, ... and one at a time for the rest (a chunk of background
            // costs a few microseconds: a reading per chunk was 2 % of a frame that is mostly background; runs as long as a body
            // workgroup's share bring the drain back).  s_run = {next chunk of the run, chunks left in it}.
            if (tid == 0) {
                unsigned left = s_run[1], next = s_run[0] + 1u;
                if (left == 0) {
                    const unsigned T = kernarg_now<uint32_t>(offsetof(FrameArgs, tail_chunks));
                    unsigned k = atomicAdd(kernarg_now<unsigned *>(offsetof(FrameArgs, tail_counter)), 1u);
                    s_pull = k;
                    unsigned at = 0;
                    next = 0xFFFFFFFFu;                // none left
                    for (unsigned size = tail_first_run(kernarg_now<uint32_t>(offsetof(FrameArgs, body_iters))); size >= 1; size >>= 1) {
                        const unsigned runs = size > 1 ? T / (4u * size) : T - at;      // a quarter of the tail per size; the rest one by one
                        if (k < runs) { next = at + k * size; left = size; break; }
                        k -= runs;
                        at += runs * size;
                    }
                }
                s_run[0] = next;
                s_run[1] = next == 0xFFFFFFFFu ? 0u : left - 1u;
            }
            __syncthreads();
            const unsigned c = s_run[0];
            __syncthreads();
            if (c == 0xFFFFFFFFu) break;              // the whole workgroup leaves together: there are barriers on this path
            idx = ((unsigned long long)kernarg_now<uint32_t>(offsetof(FrameArgs, tail_base)) + c) * kTraceBlock + tid;
            if (idx >= limit) continue;               // waves beyond the end of a ragged last chunk
        } else if (idx >= limit) break;               // (whole waves: limit is a multiple of 64)
        const bool live = idx < n;
        uint32_t x = 0, row = 0, y = 0, sm = 0;
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = make_float4(1.f, 1.f, 1.f, -1.f);
        if (live) {
            eye_sample_of(a.eye, idx, x, row, y, sm);
            eye_ray_of(a.eye, x, y, sm, ra, rb);
        }
        // ---- primary ray
        mr_hit h;
        {
            RayRegs r;
            ray_setup(r, ra, rb);
            Lane L;
            int plane_hit;
            trace_ray<true, false, false, VAR>(a.tp, r, rb.w, live, L, plane_hit, s_stack, tid, st);
            h = make_hit<kObj>(a.tp, L, plane_hit, rb.w);
        }
        if (a.hits && live) reinterpret_cast<float4 *>(a.hits)[idx] = *reinterpret_cast<const float4 *>(&h);

        // ---- shadow ray from the hit in registers (Phong.cpp:80-97).  The sample's colour is computed BEFORE the
        // shadow ray is traced and zeroed afterwards if the light is occluded (Phong.cpp:97-100): only three values
        // stay live across the second traversal instead of the hit point, the normal and the eye direction, which
        // keeps the kernel at the register count of the plain trace kernel.  (MAT: four -- the highlight apart, because a
        // refractive occluder scales the diffuse term only, Phong.cpp:146.)
        const bool hit = live && h.prim != MR_MISS;
        float c[3] = {0.f, 0.f, 0.f}, highlight = 0.0f;
        if (live && !hit) { c[0] = a.lt.bg[0]; c[1] = a.lt.bg[1]; c[2] = a.lt.bg[2]; }         // Scene.cpp:340
        float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = make_float4(1.f, 1.f, 1.f, -1.f);
        if (hit) {
            float P[3], N[3];
            if (MAT) {
                rec::MeshMat mm;
                mm.s = a.m; mm.mats = a.mats; mm.prim_mat = a.prim_mat;
                rec::LightArgs la;
                for (int k = 0; k < 3; k++) { la.L[k] = a.lt.L[k]; la.color[k] = a.lt.color[k]; }
                la.wattage = a.lt.wattage;
                rec::surface_point_od(mm, ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, h.t, h.prim, h.beta, h.gamma, P, N);
                rec::phong_terms(la, rec::material_of(mm, h.prim), P, N, rb.x, rb.y, rb.z, c, highlight);
            } else {
                surface_od<true>(a.m, ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, h.t, h.prim, h.beta, h.gamma, P, N);
            }
            if (SHADOW != 2) shadow_ray_of(P, a.lt.L[0], a.lt.L[1], a.lt.L[2], sa, sb);
            if (!MAT) phong_direct(a.lt, P, N, rb.x, rb.y, rb.z, c);                            // Phong.cpp:116-156
            if (SHADOW != 2) my_shadow_rays++;
        }
        if (SHADOW != 2) {
            RayRegs r;
            ray_setup(r, sa, sb);
            Lane L;
            int plane_hit;
            trace_ray<true, SHADOW == 1, false, VAR>(a.tp, r, sb.w, hit, L, plane_hit, s_stack, tid, st);
            mr_hit hs = make_hit<kObj>(a.tp, L, plane_hit, sb.w);
            if (!hit) { hs.t = 0.0f; hs.prim = MR_MISS; hs.beta = 0.0f; hs.gamma = 0.0f; }
            if (a.shadow_hits && live) reinterpret_cast<float4 *>(a.shadow_hits)[idx] = *reinterpret_cast<const float4 *>(&hs);
            if (MAT) {
                if (hit) {
                    rec::MeshMat mm;
                    mm.s = a.m; mm.mats = a.mats; mm.prim_mat = a.prim_mat;
                    rec::phong_combine(c, highlight, rec::light_scale_of(mm, sa, sb, *reinterpret_cast<const float4 *>(&hs)), c);
                }
            } else if (hs.prim != MR_MISS) {
                c[0] = 0.f; c[1] = 0.f; c[2] = 0.f;
            }
        } else {
            if (a.shadow_hits && live) reinterpret_cast<float4 *>(a.shadow_hits)[idx] = make_float4(0.0f, __uint_as_float(MR_MISS), 0.0f, 0.0f);
            if (MAT && hit) rec::phong_combine(c, highlight, 1.0f, c);
        }
        // the pixel's mean (Scene.cpp:126-139)
        // spp is a power of two <= 64 (checked by the host): a pixel's samples are `spp` consecutive, aligned lanes; the
        // summation tree depends on the sample index only -- the same tree as shade_samples_kernel's
        for (uint32_t off = 1; off < spp; off <<= 1) {
            c[0] += __shfl_xor(c[0], (int)off, 64);
            c[1] += __shfl_xor(c[1], (int)off, 64);
            c[2] += __shfl_xor(c[2], (int)off, 64);
        }
        if (live) eye_sample_of(a.eye, idx, x, row, y, sm);      // recomputed: cheaper than four live registers
        if (live && sm == 0) {
            if (spp > 1) { c[0] *= inv_spp; c[1] *= inv_spp; c[2] *= inv_spp; }
            float *o = a.rgb + 3 * ((size_t)row * a.eye.W + x);
            o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
        }
    }

    if (stride == 0 && tid == 0) {                    // the tail's last reader re-arms the counter for the next launch
        // every tail workgroup reads exactly one value at or beyond the number of runs
        const unsigned readers = gridDim.x - kernarg_now<uint32_t>(offsetof(FrameArgs, body_wgs));
        const unsigned T = kernarg_now<uint32_t>(offsetof(FrameArgs, tail_chunks));
        unsigned runs = 0, at = 0;
        for (unsigned size = tail_first_run(kernarg_now<uint32_t>(offsetof(FrameArgs, body_iters))); size >= 1; size >>= 1) {
            const unsigned r = size > 1 ? T / (4u * size) : T - at;
            runs += r;
            at += r * size;
        }
        if (s_pull == runs + readers - 1u) atomicExch(kernarg_now<unsigned *>(offsetof(FrameArgs, tail_counter)), 0u);
    }
    if (a.counts) {
        // rays traced.  Shadow rays: one atomic per workgroup (a single counter word drains ~88 atomics per microsecond:
        // a 1-spp frame's 8 100 workgroups are already a measurable 4 % with two words each); primary rays: the launch's
        // sample count, added once.
        unsigned w = my_shadow_rays;
        for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off, 64);
        if ((tid & 63) == 0) s_shadow_rays[tid >> 6] = w;
        __syncthreads();
        if (tid == 0) {
            unsigned long long tot = 0;
            for (int k = 0; k < kTraceBlock / 64; k++) tot += s_shadow_rays[k];
            if (tot) atomicAdd(&a.counts[1], tot);
            if (blockIdx.x == 0) atomicAdd(&a.counts[0], n);
        }
    }
#if defined(MIRO_WG_TIMES) && MIRO_TRACE_BLOCK == 256
    if (tid == 0 && blockIdx.x < 131072u) {
        g_wg_times[4 * blockIdx.x] = wg_t0;
        g_wg_times[4 * blockIdx.x + 1] = wall_clock64();
        g_wg_times[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
        g_wg_times[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
    }
#endif
}
#if defined(MIRO_WG_TIMES) && MIRO_TRACE_BLOCK == 256
}  // namespace
}  // namespace mr
extern "C" int mr_debug_wg_times(unsigned long long *out, unsigned n_words) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mr::g_wg_times), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
namespace mr {
namespace {
#endif

template <int VAR, int SHADOW, bool MAT>
mr_status launch_frame_t(FrameArgs a, hipStream_t stream) {
    const size_t lds = (size_t)a.tp.stack_depth * kTraceBlock * sizeof(int);
    if (lds > 150 * 1024) return fail(MR_ERR_INVALID, "traversal stack of depth %d does not fit in LDS", a.tp.stack_depth);
    if (lds > 48 * 1024)
        MR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&frame_kernel<VAR, SHADOW, MAT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const FrameShape shape = frame_schedule((a.eye.n + kTraceBlock - 1) / kTraceBlock);
    a.body_wgs = shape.body_wgs; a.body_iters = shape.body_iters; a.tail_chunks = shape.tail_chunks;
    a.tail_base = shape.body_wgs * shape.body_iters;
    if (shape.tail_chunks && !a.tail_counter) return fail(MR_ERR_STATE, "mr_render_direct: no hand-out counter for the tail of a large frame");
    hipLaunchKernelGGL((frame_kernel<VAR, SHADOW, MAT>), dim3(shape.grid), dim3(kTraceBlock), lds, stream, a);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

// the uniform-material frame: closest-hit or any-hit shadow rays in every traversal mode
template <int VAR>
mr_status launch_frame_plain(const FrameArgs &a, bool any, hipStream_t stream) {
    return any ? launch_frame_t<VAR, 1, false>(a, stream) : launch_frame_t<VAR, 0, false>(a, stream);
}
// the default (exact) traversal also comes without shadow rays (-DDISABLE_SHADOWS) and with per-object materials
template <int VAR>
mr_status launch_frame_default(const FrameArgs &a, bool any, bool no_shadows, bool mat, hipStream_t stream) {
    if (mat) return no_shadows ? launch_frame_t<VAR, 2, true>(a, stream) : launch_frame_t<VAR, 0, true>(a, stream);
    if (no_shadows) return launch_frame_t<VAR, 2, false>(a, stream);
    return launch_frame_plain<VAR>(a, any, stream);
}

}  // namespace

// This file is compiled twice (Makefile): with 256-thread workgroups as launch_frame_b256 and with 128-thread ones
// (-DMIRO_TRACE_BLOCK=128) as launch_frame_b128 -- small frames fill the chip more evenly in the finer grain (1 to 4 spp at
// 1080p: +4-6 %), large ones are 2 % better off with the coarser one; launch_frame (mr_internal.h) picks by sample count.
#ifndef MR_FRAME_ENTRY
#define MR_FRAME_ENTRY launch_frame_b256
#endif
mr_status MR_FRAME_ENTRY(const DeviceScene &ds, const mr_frame_desc &fd, float *d_rgb, mr_hit *d_hits, mr_hit *d_shadow_hits,
                         unsigned long long *d_counts, unsigned long long *work_counter, hipStream_t stream) {
    FrameArgs a;
    a.tail_counter = reinterpret_cast<unsigned *>(work_counter);
    const uint32_t spp = fd.spp;
    if (spp == 0 || spp > 64 || (spp & (spp - 1)))
        return fail(MR_ERR_INVALID, "mr_render_direct: spp must be a power of two <= 64 (got %u); use the batched pipeline", spp);
    if (fd.band_world == 0 || fd.band_rank >= fd.band_world || fd.band_rows == 0)
        return fail(MR_ERR_INVALID, "mr_render_direct: bad band description (%u rows, rank %u of %u)", fd.band_rows, fd.band_rank, fd.band_world);
    // rows of this window
    uint32_t rows = 0;
    if (fd.band_world == 1) {
        if (fd.y1 > fd.H || fd.y0 > fd.y1) return fail(MR_ERR_INVALID, "mr_render_direct: rows [%u,%u) outside the image", fd.y0, fd.y1);
        rows = fd.y1 - fd.y0;
    } else {
        const uint32_t nb = (fd.H + fd.band_rows - 1) / fd.band_rows;
        for (uint32_t b = fd.band_rank; b < nb; b += fd.band_world) {
            const uint32_t y0 = b * fd.band_rows, y1 = y0 + fd.band_rows < fd.H ? y0 + fd.band_rows : fd.H;
            rows += y1 - y0;
        }
    }
    a.eye = make_eye_frame(fd.camera, fd.W, fd.H, fd.band_world == 1 ? fd.y0 : 0, fd.band_world == 1 ? fd.y1 : rows, spp, fd.jitter, fd.seed,
                           fd.tiled != 0);
    a.eye.rows = rows;
    a.eye.n = (unsigned long long)rows * fd.W * spp;
    if (fd.band_world > 1) {
        a.eye.y0 = 0; a.eye.band_rows = fd.band_rows; a.eye.band_rank = fd.band_rank; a.eye.band_world = fd.band_world;
        // a ragged last band (H not a multiple of band_rows) only ever is the LAST window row group of its owner, so the
        // row -> image row formula of eye_sample_of holds for it as well
    }
    if (a.eye.n == 0) return MR_OK;
    if (a.eye.n >= (1ull << 32) * spp) return fail(MR_ERR_INVALID, "mr_render_direct: window too large");
    TraceParams &p = a.tp;
    p.nodes = ds.nodes; p.tris = ds.tris; p.tri_prim = ds.tri_prim; p.leaf_cnt_ext = ds.leaf_cnt_ext;
    for (int c = 0; c < 3; c++) { p.root_lo[c] = ds.root_lo[c]; p.root_hi[c] = ds.root_hi[c]; }
    p.root_ref = ds.root_ref;
    p.stack_depth = (int32_t)ds.stack_depth;
    p.rays = nullptr; p.hits = nullptr; p.n = a.eye.n; p.n_dev = nullptr; p.stats = nullptr;
    p.planes = ds.planes; p.n_planes = ds.n_planes; p.n_spheres = ds.n_spheres;
    p.work_counter = nullptr; p.order = nullptr;
    a.m = surface_ptrs(ds);
    for (int c = 0; c < 3; c++) {
        a.lt.L[c] = fd.light.position[c]; a.lt.color[c] = fd.light.color[c]; a.lt.diffuse[c] = fd.diffuse[c]; a.lt.bg[c] = 0.0f;
    }
    a.lt.wattage = fd.light.wattage;
    a.hits = d_hits; a.shadow_hits = d_shadow_hits; a.rgb = d_rgb; a.counts = d_counts;

    a.mats = ds.materials; a.prim_mat = ds.prim_material;
    const bool any = fd.flags & MR_TRACE_ANY, product = fd.flags & MR_MATH_PRODUCT, vote = fd.flags & MR_TRACE_INCOHERENT;
    const bool no_shadows = fd.flags & MR_FRAME_NO_SHADOWS, mat = ds.user_materials != 0;
    if (fd.flags & ~(uint32_t)(MR_TRACE_ANY | MR_MATH_PRODUCT | MR_TRACE_INCOHERENT | MR_FRAME_NO_SHADOWS))
        return fail(MR_ERR_INVALID, "mr_render_direct: flags may hold MR_TRACE_ANY, MR_MATH_PRODUCT, MR_TRACE_INCOHERENT, MR_FRAME_NO_SHADOWS only");
    if (any && no_shadows) return fail(MR_ERR_INVALID, "mr_render_direct: MR_TRACE_ANY and MR_FRAME_NO_SHADOWS exclude one another");
    if (any && ds.refractive)
        return fail(MR_ERR_STATE, "mr_render_direct: MR_TRACE_ANY in a scene with a refractive material -- the light through such an "
                                  "occluder depends on WHICH occluder is nearest (Phong.cpp:99-113)");
    if ((mat || no_shadows) && (product || vote || (mat && any)))
        return fail(MR_ERR_INVALID, "mr_render_direct: per-object materials and MR_FRAME_NO_SHADOWS come with the default traversal only "
                                    "(no MR_MATH_PRODUCT / MR_TRACE_INCOHERENT%s)", mat ? " / MR_TRACE_ANY" : "");
    if (ds.n_planes || ds.n_spheres)
        return product ? launch_frame_plain<43>(a, any, stream) : launch_frame_default<826>(a, any, no_shadows, mat, stream);
    if (vote) return product ? launch_frame_plain<73>(a, any, stream) : launch_frame_plain<88>(a, any, stream);
    if (product) return launch_frame_plain<267>(a, any, stream);
    return launch_frame_default<794>(a, any, no_shadows, mat, stream);
}

}  // namespace mr
