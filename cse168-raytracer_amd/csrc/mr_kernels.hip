// mr_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the Miro intersection path.
//
//   trace_kernel        Scene::trace -> BVH::intersect -> BVH::intersectChildren -> Triangle::intersect
//                       (Scene.cpp:214-268, BVH.cpp:438-658 scalar branch, Triangle.cpp:136-169)
//   eye_rays_kernel     Camera::eyeRay (Camera.cpp:104-161)
//   shadow_rays_kernel  Phong::shade shadow ray (Phong.cpp:80-97) + wave64 ballot compaction
//   hit_attrs_kernel    HitInfo::P / ::N (Triangle.cpp:160,162)
// The traversal itself (slab tests, triangle / sphere tests, LDS stack, control-flow modes) is mr_traverse.h, shared with
// the fused frame kernel of mr_frame.hip; this file holds the batched kernels around it and their launch logic.
//
// Compiled with -ffp-contract=off: in the default ("exact") mode every fp32 operation below is one
// individually rounded IEEE op in the reference's order, so t / beta / gamma are bit-identical to the
// reference's scalar build.  MR_MATH_FAST opts into explicit fmaf + v_rcp_f32.
//
// Traversal is pointer chasing, not a contraction: no MFMA.  One ray per lane, the pending-node stack
// lives in LDS as stack[level][thread] (consecutive lanes -> consecutive banks, conflict-free for
// ds_read/write_b32), nodes are 64-byte records fetched as 4 x dwordx4, triangles 48-byte records
// fetched as 3 x dwordx4 (layout in mr_internal.h).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mr_internal.h"
#include "mr_surface.h"
#include "mr_eye.h"
#include "mr_phong.h"
#include "mr_tile.h"
#include "mr_traverse.h"

namespace mr {
namespace {

template <bool EXACT, bool ANY, bool STATS, int VAR>
__global__ __launch_bounds__(kTraceBlock) void trace_kernel(TraceParams p) {
    extern __shared__ int s_stack[];                  // [stack_depth][kTraceBlock]
    const int tid = threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * kTraceBlock;
    Stats st = {0ull, 0ull};
    constexpr bool kWW = (VAR & (2 | 64)) != 0;      // wave-cooperative control flow: whole waves enter the loop

    // indirect batches: the ray count lives on the device (e.g. written by the shadow-ray compaction)
    unsigned long long n_rays = p.n;
    if (p.n_dev) { const unsigned long long nd = *p.n_dev; if (nd < n_rays) n_rays = nd; }
    const unsigned long long n_round = kWW ? ((n_rays + 63ull) & ~63ull) : n_rays;   // whole waves for __any

    for (unsigned long long idx = (unsigned long long)xcd_block_id() * kTraceBlock + tid; idx < n_round; idx += stride) {
        const bool live = idx < n_rays;
        // two dwordx4 loads per lane, 32-byte stride: every byte of the fetched lines is used
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = make_float4(1.f, 1.f, 1.f, -1.f);
        if (live) {                                       // mr_trace_grouped: the lane's ray is order[idx], its hit goes to hits[order[idx]]
            const unsigned long long src = p.order ? (unsigned long long)p.order[idx] : idx;
            ra = reinterpret_cast<const float4 *>(p.rays)[2 * src];
            rb = reinterpret_cast<const float4 *>(p.rays)[2 * src + 1];
        }
        RayRegs r;
        ray_setup(r, ra, rb);
        Lane L;
        int plane_hit;
        trace_ray<EXACT, ANY, STATS, VAR>(p, r, rb.w, live, L, plane_hit, s_stack, tid, st);
        if (live) {
            const mr_hit h = make_hit<(VAR & 32) != 0>(p, L, plane_hit, rb.w);
            asm volatile("" ::: "memory");                // the index is read again rather than kept in registers across the traversal
            const unsigned long long dst = p.order ? (unsigned long long)p.order[idx] : idx;
            reinterpret_cast<float4 *>(p.hits)[dst] = *reinterpret_cast<const float4 *>(&h);
        }
    }

    if (STATS) {
        // wave64 reduction, one atomic pair per wave
        for (int off = 32; off > 0; off >>= 1) {
            st.box += __shfl_down(st.box, off, 64);
            st.tri += __shfl_down(st.tri, off, 64);
        }
        if ((tid & 63) == 0) {
            atomicAdd(&p.stats[0], st.box);
            atomicAdd(&p.stats[1], st.tri);
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Ray order for bounce queues (mr_trace_grouped).  A generator writes its children in the order of their parents: the
// origins of neighbouring rays are neighbouring surface points, their directions are drawn independently -- the 64 rays
// of a wave point into all eight octants, run the generic slab tests, and a wave lasts as long as its unluckiest lane.
// Grouping the rays of a chunk by octant keeps the origins coherent and makes whole waves share a direction sign: the
// octant-specialised loops apply, and rays that leave the scene at once (the bunny's upward bounces) stop holding waves.
// order[] is a permutation inside each chunk of 2^chunk_log2 rays: stable counting sort on the octant (sign bits of d);
// no ray is moved -- the trace kernel gathers through the index (two 16-byte loads per lane either way).
// One workgroup per chunk, a wave owns a contiguous quarter of it; octants are kept as 3-bit fields in registers.
// ---------------------------------------------------------------------------------------------------
constexpr int kOrderMaxChunkLog2 = 14;    // chunks of up to 16 384 rays (one octant byte per ray in LDS)
constexpr int kOrderWaves = 16;           // waves of a workgroup, at most (blockDim = 64 * min(16, chunk / 64))
// BYTES: the octants come from a generator's d_out_octants (one byte per ray: 1/32 of the traffic of reading the rays)
template <bool BYTES>
__global__ __launch_bounds__(64 * kOrderWaves) void octant_order_kernel(const mr_ray *rays, const uint8_t *octants, unsigned long long n, uint32_t chunk_log2, uint32_t *order) {
    __shared__ unsigned s_cnt[kOrderWaves][8];
    __shared__ unsigned char s_oct[1 << kOrderMaxChunkLog2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = (int)(blockDim.x >> 6);
    const unsigned long long chunk = 1ull << chunk_log2, base = (unsigned long long)blockIdx.x * chunk;
    const unsigned per_wave = (unsigned)(chunk / (unsigned)waves), iters = per_wave / 64u;     // chunk >= 64 * waves
    const unsigned wfirst = (unsigned)wave * per_wave;          // a wave owns a contiguous part: the sort is stable
    unsigned cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};                 // wave totals (uniform)
    for (unsigned it = 0; it < iters; it += 4u) {               // four loads in flight per lane
        float4 rb[4];
        unsigned ob[4];
#pragma unroll
        for (unsigned u = 0; u < 4u; u++) {
            const unsigned long long idx = base + wfirst + (it + u) * 64u + (unsigned)lane;
            rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            ob[u] = 0;
            if (it + u < iters && idx < n) {
                if (BYTES) ob[u] = octants[idx] & 7u;
                else rb[u] = reinterpret_cast<const float4 *>(rays)[2 * idx + 1];
            }
        }
#pragma unroll
        for (unsigned u = 0; u < 4u; u++) {
            if (it + u >= iters) break;
            const unsigned local = wfirst + (it + u) * 64u + (unsigned)lane;
            unsigned oct = BYTES ? ob[u] : (rb[u].x < 0.0f ? 1u : 0u) | (rb[u].y < 0.0f ? 2u : 0u) | (rb[u].z < 0.0f ? 4u : 0u);
            if (base + local >= n) oct = 8;                     // 8: no ray here
            s_oct[local] = (unsigned char)oct;                  // read back by the same lane below
#pragma unroll
            for (int o = 0; o < 8; o++) cnt[o] += (unsigned)__popcll(__ballot(oct == (unsigned)o));
        }
    }
    if (lane < 8) {
        unsigned c = 0;
#pragma unroll
        for (int o = 0; o < 8; o++) c = lane == o ? cnt[o] : c;
        s_cnt[wave][lane] = c;
    }
    __syncthreads();
    // where this wave's rays of octant o start: all rays of smaller octants, then the same octant's rays of earlier waves
    unsigned off[8];
    {
        unsigned run = 0;
#pragma unroll
        for (int o = 0; o < 8; o++) {
            unsigned before = 0, total = 0;
            for (int w = 0; w < waves; w++) { const unsigned c = s_cnt[w][o]; total += c; if (w < wave) before += c; }
            off[o] = run + before;
            run += total;
        }
    }
    for (unsigned it = 0; it < iters; it++) {
        const unsigned local = wfirst + it * 64u + (unsigned)lane;
        const unsigned oct = s_oct[local];
#pragma unroll
        for (int o = 0; o < 8; o++) {
            const unsigned long long m = __ballot(oct == (unsigned)o);
            if (oct == (unsigned)o) order[base + off[o] + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)(base + local);
            off[o] += (unsigned)__popcll(m);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Persistent form for incoherent batches (secondary / shadow / random rays): a resident grid of waves pulls
// rays from a global counter.  Whenever at least REFILL_MIN lanes of a wave have finished their ray, the wave
// ballots the idle lanes, retires their hits, and re-arms them with fresh rays by prefix-sum over the ballot
// (wave64 active-ray compaction: lanes never wait for the slowest ray of a fixed group of 64).  Rays are handed
// out from a wave-local pool of kPoolChunk consecutive indices so that the global counter sees one atomic per
// chunk.  Per-ray work and results are those of trace_kernel: QUOT selects the default trace's exact quotients
// (correction step for regular rays, the reference's divisions otherwise), !QUOT the MR_MATH_PRODUCT arithmetic.
// ---------------------------------------------------------------------------------------------------
constexpr unsigned long long kPoolChunk = 1024;

// VOTE: the voting control flow of traverse() MODE 2 (one node step or one triangle test per iteration, whichever more
// lanes need) instead of while-while.
template <bool EXACT, bool ANY, bool QUOT, int REFILL_MIN, bool VOTE = false>
__global__ __launch_bounds__(kTraceBlock) void trace_persistent_kernel(TraceParams p, unsigned long long *next_ray) {
    extern __shared__ int s_stack[];
    const int tid = threadIdx.x, lane = tid & 63;
    Stats st = {0ull, 0ull};
    unsigned long long n_rays = p.n;
    if (p.n_dev) { const unsigned long long nd = *p.n_dev; if (nd < n_rays) n_rays = nd; }

    unsigned long long pool_next = 0, pool_end = 0;      // wave-uniform
    bool exhausted = false;                              // wave-uniform
    const unsigned long long kNone = ~0ull;
    unsigned long long my_idx = kNone;
    float tmax0 = 0.0f;
    bool safe_lane = true;
    RayRegs r = {};
    Lane L;
    L.best_t = 0.f; L.best_b = 0.f; L.best_g = 0.f; L.best_pos = -1; L.sp = 0; L.cur = kDone;
    L.lpos = 0; L.lend = 0;
    bool wave_safe = true;                               // wave-uniform: every armed lane's ray is safe_lane

    while (true) {
        const unsigned long long idle = __ballot(!L.have());
        const int n_idle = __popcll(idle);
        if (n_idle >= REFILL_MIN || idle == ~0ull) {
            // retire the rays of the idle lanes
            if (!L.have() && my_idx != kNone) {
                mr_hit h;
                if (L.best_pos >= 0) { h.t = L.best_t; h.prim = p.tri_prim[L.best_pos]; h.beta = L.best_b; h.gamma = L.best_g; }
                else { h.t = tmax0; h.prim = MR_MISS; h.beta = 0.0f; h.gamma = 0.0f; }
                reinterpret_cast<float4 *>(p.hits)[my_idx] = *reinterpret_cast<const float4 *>(&h);
                my_idx = kNone;
            }
            if (pool_next == pool_end && !exhausted) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(next_ray, kPoolChunk);
                base = __shfl(base, 0, 64);
                if (base >= n_rays) { exhausted = true; pool_next = pool_end = 0; }
                else { pool_next = base; pool_end = base + kPoolChunk < n_rays ? base + kPoolChunk : n_rays; }
            }
            if (pool_next < pool_end) {
                const unsigned long long cand = pool_next + (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
                if (!L.have() && cand < pool_end) {
                    my_idx = cand;
                    const float4 ra = reinterpret_cast<const float4 *>(p.rays)[2 * cand];
                    const float4 rb = reinterpret_cast<const float4 *>(p.rays)[2 * cand + 1];
                    ray_setup(r, ra, rb);
                    tmax0 = rb.w;
                    L.best_t = tmax0; L.best_b = 0.0f; L.best_g = 0.0f; L.best_pos = -1;
                    L.lpos = 0; L.lend = 0;
                    stack_reset(L, s_stack, tid);
                    float mn = -kInf, mx = kInf;
                    slab_axis<QUOT>(p.root_lo[0], p.root_hi[0], r.ox, r.dx, r.ix, mn, mx);
                    slab_axis<QUOT>(p.root_lo[1], p.root_hi[1], r.oy, r.dy, r.iy, mn, mx);
                    slab_axis<QUOT>(p.root_lo[2], p.root_hi[2], r.oz, r.dz, r.iz, mn, mx);
                    L.cur = !((mn > mx) || (mn > tmax0) || (mx < r.tmin)) ? p.root_ref : kDone;
                    safe_lane = QUOT ? lane_is_regular(r) : lane_is_nan_free(r);
                }
                const unsigned long long adv = pool_next + (unsigned)n_idle;
                pool_next = adv < pool_end ? adv : pool_end;
            }
            wave_safe = __all(safe_lane || !L.have());
        }
        if (!__any(L.have())) {
            if (exhausted && pool_next == pool_end) {
                if (my_idx != kNone) {              // rays that missed the root box in the last hand-out
                    mr_hit h;
                    h.t = tmax0; h.prim = MR_MISS; h.beta = 0.0f; h.gamma = 0.0f;
                    reinterpret_cast<float4 *>(p.hits)[my_idx] = *reinterpret_cast<const float4 *>(&h);
                }
                break;
            }
            continue;
        }
        if (VOTE) {
            const bool want_node = L.cur >= 0, want_tri = L.cur < 0 && L.cur != kDone;
            if (__popcll(__ballot(want_node)) >= __popcll(__ballot(want_tri))) {
                if (want_node) {
                    if (wave_safe) node_step<EXACT, false, QUOT ? 4 : 1, true>(p, r, L, s_stack, tid, st);
                    else node_step<EXACT, false, QUOT ? 3 : 0, QUOT>(p, r, L, s_stack, tid, st);
                }
            } else if (want_tri) {
                tri_step<EXACT, ANY, false, true, false>(p, r, L, s_stack, st);
            }
            continue;
        }
        if (wave_safe) {
            while (L.cur >= 0) node_step<EXACT, false, QUOT ? 4 : 1, true>(p, r, L, s_stack, tid, st);
        } else {
            while (L.cur >= 0) node_step<EXACT, false, QUOT ? 3 : 0, QUOT>(p, r, L, s_stack, tid, st);
        }
        if (L.have()) leaf_step<EXACT, ANY, false, true>(p, r, L, s_stack, tid, st);
    }
}

// ---------------------------------------------------------------------------------------------------
// Camera::eyeRay (Camera.cpp:104-161).  The camera frame is computed on the host exactly as the
// reference does; the per-pixel arithmetic below keeps the reference's operation order.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void eye_rays_kernel(EyeFrame f, mr_ray *rays) {
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < f.n; k += stride) {
        uint32_t x, row, y, sm;
        eye_sample_of(f, k, x, row, y, sm);
        float4 a, b;
        eye_ray_of(f, x, y, sm, a, b);
        reinterpret_cast<float4 *>(rays)[2 * k] = a;
        reinterpret_cast<float4 *>(rays)[2 * k + 1] = b;
    }
}

// ---------------------------------------------------------------------------------------------------
// shadow rays (Phong.cpp:80-97): origin P + l*eps, direction l = normalise(L - P), tMax = |L - P|.
// Hits are compacted wave-by-wave: ballot of hitting lanes, one atomicAdd per wave for the base,
// mbcnt prefix for the lane's slot.
// ---------------------------------------------------------------------------------------------------
// Compaction is hierarchical so that the global counter sees one atomic per 2048 rays, not one per wave
// (a single counter word saturates at ~88 atomics/us, MI355X_MICROARCH.md "dequeue"): each workgroup takes
// chunks of kBlock*kShIter rays; every wave ballots its kShIter sub-rows (wave64 __ballot + popcount
// prefix), wave totals meet in LDS, lane 0 reserves the chunk's range with one atomicAdd, and each lane
// writes at  chunk base + waves before mine + sub-rows before this one + lanes before mine.
constexpr int kShIter = 8;

__global__ __launch_bounds__(kBlock) void shadow_rays_kernel(SurfacePtrs m, const mr_ray *rays, const mr_hit *hits, unsigned long long n,
                                                             float Lx, float Ly, float Lz, mr_ray *out,
                                                             uint32_t *src, unsigned long long *count) {
    __shared__ unsigned s_wave_total[kBlock / 64];
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long chunk = (unsigned long long)kBlock * kShIter;
    const unsigned long long n_chunks = (n + chunk - 1) / chunk;
    for (unsigned long long c = blockIdx.x; c < n_chunks; c += gridDim.x) {      // uniform per workgroup
        float4 h[kShIter];
        unsigned prefix[kShIter];     // hits of this wave in earlier sub-rows + earlier lanes of this one
        unsigned wave_hits = 0;
#pragma unroll
        for (int it = 0; it < kShIter; it++) {
            const unsigned long long k = c * chunk + (unsigned long long)it * kBlock + threadIdx.x;
            bool is_hit = false;
            h[it] = make_float4(0.f, __uint_as_float(MR_MISS), 0.f, 0.f);
            if (k < n) {
                h[it] = reinterpret_cast<const float4 *>(hits)[k];
                is_hit = __float_as_uint(h[it].y) != MR_MISS;
            }
            const unsigned long long mask = __ballot(is_hit);
            prefix[it] = wave_hits + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
            wave_hits += (unsigned)__popcll(mask);
        }
        if (lane == 0) s_wave_total[wave] = wave_hits;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned tot = 0;
            for (int w = 0; w < kBlock / 64; w++) tot += s_wave_total[w];
            s_base = tot ? atomicAdd(count, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        unsigned long long wave_base = s_base;
        for (int w = 0; w < wave; w++) wave_base += s_wave_total[w];
#pragma unroll
        for (int it = 0; it < kShIter; it++) {
            if (__float_as_uint(h[it].y) == MR_MISS) continue;
            const unsigned long long k = c * chunk + (unsigned long long)it * kBlock + threadIdx.x;
            const unsigned long long slot = wave_base + prefix[it];
            float P[3];
            surface<false>(m, rays, k, h[it].x, __float_as_uint(h[it].y), h[it].z, h[it].w, P, nullptr);
            float4 a, b;
            shadow_ray_of(P, Lx, Ly, Lz, a, b);
            reinterpret_cast<float4 *>(out)[2 * slot] = a;
            reinterpret_cast<float4 *>(out)[2 * slot + 1] = b;
            if (src) src[slot] = (uint32_t)k;
        }
        __syncthreads();              // s_wave_total / s_base are reused by the next chunk
    }
}

__global__ __launch_bounds__(kBlock) void hit_attrs_kernel(SurfacePtrs m, const mr_ray *rays, const mr_hit *hits,
                                                           unsigned long long n, float *P, float *N) {
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < n; k += stride) {
        const float4 h = reinterpret_cast<const float4 *>(hits)[k];
        const uint32_t prim = __float_as_uint(h.y);
        float Pv[3] = {0.f, 0.f, 0.f}, Nv[3] = {0.f, 1.f, 0.f};               // HitInfo defaults (Ray.h:31-34)
        if (prim != MR_MISS) surface<true>(m, rays, k, h.x, prim, h.z, h.w, Pv, Nv);
        if (P) { P[3 * k] = Pv[0]; P[3 * k + 1] = Pv[1]; P[3 * k + 2] = Pv[2]; }
        if (N) { N[3 * k] = Nv[0]; N[3 * k + 1] = Nv[1]; N[3 * k + 2] = Nv[2]; }
    }
}

inline unsigned grid_for(unsigned long long n) {
    // memory/latency-bound kernels: cap the grid and grid-stride the rest (256 CUs x 8 blocks)
    unsigned long long blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 256ull * 32ull) blocks = 256ull * 32ull;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

// workgroups per trace launch.  -DMIRO_DEV builds read MIRO_TRACE_GRID_CAP once (A/B tooling); the shipped library
// has no environment switches on the launch path.
inline int trace_grid_cap() {
#ifdef MIRO_DEV
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MIRO_TRACE_GRID_CAP");
        v = e ? atoi(e) : kTraceGridCap;
        if (v < 1) v = 1;
    }
    return v;
#else
    return kTraceGridCap;
#endif
}

template <bool EXACT, bool ANY, bool STATS, int VAR>
mr_status launch_trace_t(const TraceParams &p, hipStream_t stream) {
    size_t lds = (size_t)p.stack_depth * kTraceBlock * sizeof(int);
#ifdef MIRO_DEV
    // occupancy experiments: MIRO_LDS_PAD bytes of unused LDS per workgroup (fewer workgroups per CU)
    { static const size_t pad = getenv("MIRO_LDS_PAD") ? (size_t)atoi(getenv("MIRO_LDS_PAD")) : 0; lds += pad; }
#endif
    if (lds > 160 * 1024) return fail(MR_ERR_INVALID, "traversal stack of depth %d does not fit in LDS", p.stack_depth);
    if (lds > 64 * 1024)
        MR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&trace_kernel<EXACT, ANY, STATS, VAR>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // Many short-lived workgroups balance better than a resident grid that strides over its rays: with exactly one
    // chip-full of workgroups (1792) the 33 M-ray frame runs 13 % slower than with the capped grid below, because the
    // hardware dispatcher rebalances at workgroup granularity while a static stride cannot.
    unsigned long long blocks = (p.n + kTraceBlock - 1) / kTraceBlock;
    const unsigned long long cap = (unsigned long long)trace_grid_cap();
    if (blocks > cap) blocks = cap;
    const unsigned grid = blocks ? (unsigned)blocks : 1u;
    hipLaunchKernelGGL((trace_kernel<EXACT, ANY, STATS, VAR>), dim3(grid), dim3(kTraceBlock), lds, stream, p);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

}  // namespace

template <bool EXACT, bool ANY, bool QUOT, int REFILL_MIN, bool VOTE = false>
static mr_status launch_persistent(const TraceParams &p, hipStream_t stream) {
    const size_t lds = (size_t)p.stack_depth * kTraceBlock * sizeof(int);
    if (lds > 160 * 1024) return fail(MR_ERR_INVALID, "traversal stack of depth %d does not fit in LDS", p.stack_depth);
    auto kern = &trace_persistent_kernel<EXACT, ANY, QUOT, REFILL_MIN, VOTE>;
    if (lds > 64 * 1024)
        MR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int dev = 0, cus = 256, per_cu = 1;
    MR_HIP_CHECK(hipGetDevice(&dev));
    MR_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    MR_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), kTraceBlock, lds));
    if (per_cu < 1) per_cu = 1;
    unsigned long long want = (p.n + kTraceBlock - 1) / kTraceBlock;
    unsigned long long grid = (unsigned long long)cus * (unsigned)per_cu;
    if (want < grid) grid = want;
    MR_HIP_CHECK(hipMemsetAsync(p.work_counter, 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kTraceBlock), lds, stream, p, p.work_counter);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

#ifdef MIRO_DEV
// development builds only (make DEV=1): MIRO_TRACE_VARIANT selects a control-flow / slab-test variant of the
// MR_MATH_PRODUCT kernel for tools/ab_variants.py:
//   0..15  one launch-time ray per lane (bit 0: min/max slabs, bit 1: while-while, bit 2: lean fma slabs,
//          bit 3: wave-uniform nodes through the scalar cache)
//   16,17  persistent waves with ballot/prefix re-arming of idle lanes (refill threshold 1 / 16 lanes)
static int trace_variant() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MIRO_TRACE_VARIANT");
        v = e ? (atoi(e) & 31) : 11;
    }
    return v;
}
template <bool ANY>
static mr_status launch_product(const TraceParams &p, hipStream_t stream) {
    switch (trace_variant()) {
        case 0: return launch_trace_t<true, ANY, false, 0>(p, stream);
        case 1: return launch_trace_t<true, ANY, false, 1>(p, stream);
        case 2: return launch_trace_t<true, ANY, false, 2>(p, stream);
        case 3: return launch_trace_t<true, ANY, false, 3>(p, stream);
        case 7: return launch_trace_t<true, ANY, false, 7>(p, stream);
        case 9: return launch_trace_t<true, ANY, false, 9>(p, stream);
        case 16: return launch_persistent<true, ANY, false, 1>(p, stream);
        case 17: return launch_persistent<true, ANY, false, 16>(p, stream);
        default: return launch_trace_t<true, ANY, false, 11>(p, stream);
    }
}
#else
// MR_MATH_PRODUCT: min/max slabs on (corner - o) * (1/d), while-while, wave-uniform nodes through the scalar cache
template <bool ANY>
static mr_status launch_product(const TraceParams &p, hipStream_t stream) {
    return launch_trace_t<true, ANY, false, 267>(p, stream);
}
#endif

mr_status launch_octant_order(const mr_ray *d_rays, const uint8_t *d_octants, unsigned long long n, uint32_t chunk_log2, uint32_t *d_order, hipStream_t stream) {
    if (n == 0) return MR_OK;
    if (chunk_log2 < 8 || chunk_log2 > (uint32_t)kOrderMaxChunkLog2) return fail(MR_ERR_INVALID, "ray order: chunks of 2^8 ... 2^%d rays", kOrderMaxChunkLog2);
    if (n > 0xFFFFFFFFull) return fail(MR_ERR_INVALID, "ray order: at most 2^32 - 1 rays per batch (32-bit indices)");
    const unsigned long long chunks = (n + (1ull << chunk_log2) - 1) >> chunk_log2;
    const unsigned waves = (1u << chunk_log2) / 64u < (unsigned)kOrderWaves ? (1u << chunk_log2) / 64u : (unsigned)kOrderWaves;
    if (d_octants) hipLaunchKernelGGL(octant_order_kernel<true>, dim3((unsigned)chunks), dim3(64u * waves), 0, stream, d_rays, d_octants, n, chunk_log2, d_order);
    else hipLaunchKernelGGL(octant_order_kernel<false>, dim3((unsigned)chunks), dim3(64u * waves), 0, stream, d_rays, d_octants, n, chunk_log2, d_order);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

mr_status launch_trace(const TraceParams &p, uint32_t flags, hipStream_t stream) {
    if (p.n == 0) return MR_OK;
    // a ray order (mr_trace_grouped) is gathered by the one-shot kernels only
    if (p.order) flags &= ~(uint32_t)MR_TRACE_PERSISTENT;
    const bool fast = flags & MR_MATH_FAST, any = flags & MR_TRACE_ANY, stats = flags & MR_COUNT_STATS;
    const bool product = flags & MR_MATH_PRODUCT, vote = flags & MR_TRACE_INCOHERENT;
    if (p.n_planes || p.n_spheres) {
        // scenes with spheres / planes: the exact kernels with the object dispatch compiled in (VAR bit 5); the fast
        // and persistent forms cover triangle scenes only
        if (stats) return any ? launch_trace_t<true, true, true, 32>(p, stream) : launch_trace_t<true, false, true, 32>(p, stream);
        if (vote) {
            if (product) return any ? launch_trace_t<true, true, false, 105>(p, stream) : launch_trace_t<true, false, false, 105>(p, stream);
            return any ? launch_trace_t<true, true, false, 120>(p, stream) : launch_trace_t<true, false, false, 120>(p, stream);
        }
        if (product) return any ? launch_trace_t<true, true, false, 43>(p, stream) : launch_trace_t<true, false, false, 43>(p, stream);
        return any ? launch_trace_t<true, true, false, 826>(p, stream) : launch_trace_t<true, false, false, 826>(p, stream);
    }
    if (stats) {
        // counting mode is diagnostic: always the literal-division kernel in the reference's control flow
        return any ? launch_trace_t<true, true, true, 0>(p, stream) : launch_trace_t<true, false, true, 0>(p, stream);
    }
    // MR_MATH_FAST: lean fma slabs + fmaf/rcp triangle test, with the same scalar-cache path as the exact kernels (VAR 15)
    if (fast) return any ? launch_trace_t<false, true, false, 15>(p, stream) : launch_trace_t<false, false, false, 15>(p, stream);
    if ((flags & MR_TRACE_PERSISTENT) && vote) {
        if (product) return any ? launch_persistent<true, true, false, 16, true>(p, stream) : launch_persistent<true, false, false, 16, true>(p, stream);
        return any ? launch_persistent<true, true, true, 16, true>(p, stream) : launch_persistent<true, false, true, 16, true>(p, stream);
    }
#ifdef MIRO_DEV
    // development builds only: MIRO_EXACT_CORRECTION=1 runs the default trace on the correction steps alone (VAR bit 9 clear)
    static const bool corr_only = getenv("MIRO_EXACT_CORRECTION") && atoi(getenv("MIRO_EXACT_CORRECTION")) != 0;
    if (corr_only && !product && !vote && !(flags & MR_TRACE_PERSISTENT)) {
        return any ? launch_trace_t<true, true, false, 282>(p, stream) : launch_trace_t<true, false, false, 282>(p, stream);
    }
#endif
    if (vote) {
        // MR_TRACE_INCOHERENT: the voting control flow (VAR bit 6) on the same arithmetic
        if (product) return any ? launch_trace_t<true, true, false, 73>(p, stream) : launch_trace_t<true, false, false, 73>(p, stream);
        // (on the correction steps alone: incoherent batches are bound by their fetches, and the guard's wave-wide branch
        // costs them 5 %, profiles/r02_guarded_products_ab.log)
        return any ? launch_trace_t<true, true, false, 88>(p, stream) : launch_trace_t<true, false, false, 88>(p, stream);
    }
    if (flags & MR_TRACE_PERSISTENT) {
        if (product) return any ? launch_persistent<true, true, false, 16>(p, stream) : launch_persistent<true, false, false, 16>(p, stream);
        return any ? launch_persistent<true, true, true, 16>(p, stream) : launch_persistent<true, false, true, 16>(p, stream);
    }
    // MR_MATH_PRODUCT: slab distances as products with the rounded 1/d (and its development variants)
    if (product) return any ? launch_product<true>(p, stream) : launch_product<false>(p, stream);
    // default: the reference's quotients -- guarded products, correction steps where a decision is close -- while-while,
    // scalar path, octant-specialised; waves whose rays point into several octants vote (VAR 16 | 2 | 8 | 256 | 512 | 1024)
    return any ? launch_trace_t<true, true, false, 1818>(p, stream) : launch_trace_t<true, false, false, 1818>(p, stream);
}

mr_status launch_eye_rays(const mr_camera &cam, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1,
                          uint32_t spp, uint32_t jitter, uint32_t seed, bool tiled, mr_ray *d_rays, hipStream_t stream) {
    const EyeFrame f = make_eye_frame(cam, W, H, y0, y1, spp, jitter, seed, tiled);
    if (f.n == 0) return MR_OK;
    hipLaunchKernelGGL(eye_rays_kernel, dim3(grid_for(f.n)), dim3(kBlock), 0, stream, f, d_rays);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

mr_status launch_shadow_rays(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, unsigned long long n,
                             const float light[3], mr_ray *d_out, uint32_t *d_src, unsigned long long *d_count,
                             hipStream_t stream) {
    MR_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), stream));
    if (n == 0) return MR_OK;
    if ((ds.spheres || ds.planes) && !d_rays)
        return fail(MR_ERR_INVALID, "the scene holds spheres / planes: their hit point is o + t*d, d_rays is required");
    hipLaunchKernelGGL(shadow_rays_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, surface_ptrs(ds), d_rays, d_hits, n,
                       light[0], light[1], light[2], d_out, d_src, d_count);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

mr_status launch_hit_attrs(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, unsigned long long n,
                           float *d_P, float *d_N, hipStream_t stream) {
    if (n == 0) return MR_OK;
    if ((ds.spheres || ds.planes) && !d_rays)
        return fail(MR_ERR_INVALID, "the scene holds spheres / planes: their hit point is o + t*d, d_rays is required");
    hipLaunchKernelGGL(hit_attrs_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, surface_ptrs(ds), d_rays, d_hits, n, d_P, d_N);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

}  // namespace mr
