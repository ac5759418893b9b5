// mr_internal.h -- shared declarations of the miro_hip library (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <cstdint>
#include <string>
#include <vector>

#include "miro_hip.h"

namespace mr {

// ---------------------------------------------------------------------------------------
// error plumbing: integer status + thread-local message (no exceptions across the ABI)
// ---------------------------------------------------------------------------------------
mr_status fail(mr_status code, const char *fmt, ...);
#define MR_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t e__ = (expr);                                                        \
        if (e__ != hipSuccess)                                                          \
            return mr::fail(MR_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                 \
                            hipGetErrorString(e__), __FILE__, __LINE__);                \
    } while (0)

// ---------------------------------------------------------------------------------------
// host tree, DFS pre-order (what BVH::build produces, BVH.h:29-63)
// ---------------------------------------------------------------------------------------
struct HostNode {
    float lo[3], hi[3];   // padded corners (m_corners[0], m_corners[1])
    int32_t is_leaf;
    int32_t a, b;         // inner: child node indices; leaf: [first, count) into leaf_prims
    int32_t depth;
};

struct HostTree {
    std::vector<HostNode> nodes;
    std::vector<uint32_t> leaf_prims;   // object indices in leaf order
    uint32_t n_leaves = 0, max_depth = 0, leaf_size = 4;
};

// Bounded objects in Scene::addObject order (Scene.h:20-25).  Object i is a triangle with indices vi/ni[3i..3i+2]
// or, when vi[3i] == kSphereSlot, the sphere number vi[3i+1] (a Sphere takes the next object index like any
// triangle would).  Planes are unbounded (Plane.h:27): they never enter the BVH and are scanned after it
// (Scene.cpp:220-230); a hit on plane k reports prim = kPlaneBit | k.
constexpr uint32_t kSphereSlot = 0xFFFFFFFFu;
constexpr uint32_t kPlaneBit = 0x80000000u;
constexpr uint32_t kSphereTag = 0x7fc00168u;   // bits of q2.w of a sphere's 48-byte leaf record (a NaN no product yields)

struct HostMesh {
    std::vector<float> v, n;        // xyz triples
    std::vector<uint32_t> vi, ni;   // 3 per object
    std::vector<float> spheres;     // cx, cy, cz, radius
    std::vector<float> planes;      // normal xyz, origin xyz
    std::vector<uint32_t> plane_material;
    uint32_t n_vertices() const { return (uint32_t)(v.size() / 3); }
    uint32_t n_normals() const { return (uint32_t)(n.size() / 3); }
    uint32_t n_triangles() const { return (uint32_t)(vi.size() / 3); }      // objects: triangles + spheres
    uint32_t n_spheres() const { return (uint32_t)(spheres.size() / 4); }
    uint32_t n_planes() const { return (uint32_t)(planes.size() / 6); }
    bool is_sphere(uint32_t obj) const { return !spheres.empty() && vi[3 * (size_t)obj] == kSphereSlot; }
};

// TriangleMesh::load semantics (TriangleMeshLoad.cpp:63-311): appends to `mesh`
mr_status load_obj(const char *path, const float *ctm16, HostMesh &mesh, uint32_t *n_tris);
// BVH::build semantics (BVH.cpp:60-339): identical tree to the reference's
mr_status build_reference_tree(const HostMesh &mesh, uint32_t leaf_size, HostTree &tree);

// ---------------------------------------------------------------------------------------
// device layout (see DESIGN.md "Data layout in HBM")
// ---------------------------------------------------------------------------------------
// Inner node i = 4 consecutive float4 (one 64-byte record, one cache-line half):
//   q0 = (c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y)
//   q1 = (c1.lo.x, c1.hi.x, c1.lo.y, c1.hi.y)
//   q2 = (c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z)
//   q3 = (ref0, ref1, -, -) as int bits
// ref >= 0: inner node index.  ref < 0: leaf, ~ref = (first << 4) | min(count, 15);
// count == 15 means "read leaf_cnt_ext[first]" (only leaves cut off at depth 32 get there).
// Triangle k (leaf order) = 3 consecutive float4 (48 bytes):
//   (A.x, A.y, A.z, BmA.x) (BmA.y, BmA.z, CmA.x, CmA.y) (CmA.z, n.x, n.y, n.z),  n = BmA x CmA
// A sphere's record in the same array: (c.x, c.y, c.z, radius) (0, 0, 0, 0) (0, 0, 0, kSphereTag bits).
// Plane k = 2 float4: (normal.xyz, material id bits) (origin.xyz, 0).
constexpr int kLeafCountBits = 4;
constexpr int kLeafCountMask = 15;
constexpr uint32_t kWorkCounters = 64;   // launches in flight on different streams each get their own counter
constexpr uint64_t kStageChunk = 1ull << 20;   // rays per chunk of a pipelined host-pointer trace (32 MiB up, 16 MiB down)

struct DeviceScene {
    float4   *nodes = nullptr;         // 4 * n_inner
    float4   *tris = nullptr;          // 3 * n_triangles (leaf order)
    uint32_t *tri_prim = nullptr;      // leaf order -> prim id
    uint32_t *leaf_cnt_ext = nullptr;  // leaf order position -> count (for count >= 15)
    // original indexed mesh, for HitInfo reconstruction and shadow-ray origins
    float    *v = nullptr, *n = nullptr;
    uint32_t *vi = nullptr, *ni = nullptr;
    // materials (11 floats each: diffuse, specular, transmission, shininess, refraction index) and the material of
    // every triangle; prim_material == nullptr means material 0 everywhere
    float    *materials = nullptr;
    uint32_t *prim_material = nullptr;
    uint32_t user_materials = 0;       // mr_scene_set_materials was called (otherwise `materials` is the one white Lambert)
    uint32_t refractive = 0;           // some material has a positive transmission component (Phong::isRefractive)
    float4   *spheres = nullptr;       // (c.xyz, radius) per sphere; nullptr when the scene has none
    float4   *planes = nullptr;        // 2 per plane; nullptr when the scene has none
    uint32_t n_spheres = 0, n_planes = 0;
    float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
    int32_t root_ref = 0;
    uint32_t n_inner = 0, n_tris = 0, stack_depth = 1;
    uint64_t bytes = 0;
};

struct TraceParams {
    const float4   *nodes;
    const float4   *tris;
    const uint32_t *tri_prim;
    const uint32_t *leaf_cnt_ext;
    float root_lo[3], root_hi[3];
    int32_t root_ref;
    int32_t stack_depth;
    const mr_ray *rays;
    mr_hit *hits;
    unsigned long long n;                 // number of rays (upper bound when n_dev is set)
    const unsigned long long *n_dev;      // optional device-resident ray count (mr_trace_indirect)
    unsigned long long *stats;   // [0] box tests, [1] triangle tests (MR_COUNT_STATS)
    const float4 *planes;                 // unbounded objects, scanned after the BVH (Scene.cpp:220-230)
    uint32_t n_planes, n_spheres;
    unsigned long long *work_counter;     // zeroed per launch: ray hand-out counter of the persistent kernel
    const uint32_t *order;                // optional: lane k traces ray order[k] (and writes hits[order[k]]): mr_trace_grouped
};

mr_status launch_trace(const TraceParams &p, uint32_t flags, hipStream_t stream);
// order[] = the batch's ray indices, grouped by direction octant inside consecutive chunks of 2^chunk_log2 rays (mr_kernels.hip)
mr_status launch_octant_order(const mr_ray *d_rays, const uint8_t *d_octants, unsigned long long n, uint32_t chunk_log2, uint32_t *d_order, hipStream_t stream);
mr_status launch_eye_rays(const mr_camera &cam, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1,
                          uint32_t spp, uint32_t jitter, uint32_t seed, bool tiled, mr_ray *d_rays, hipStream_t stream);
mr_status launch_shadow_rays(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits,
                             unsigned long long n, const float light[3], mr_ray *d_out, uint32_t *d_src,
                             unsigned long long *d_count, hipStream_t stream);
mr_status launch_hit_attrs(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, unsigned long long n,
                           float *d_P, float *d_N, hipStream_t stream);

mr_status launch_shade(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, unsigned long long n,
                       const mr_hit *d_shadow_hits, const uint32_t *d_shadow_src, const unsigned long long *d_shadow_count,
                       uint8_t *d_occluded, const mr_light &light, const float diffuse[3], uint32_t spp, float *d_rgb,
                       hipStream_t stream);
// final gather (Scene.cpp:285-299): queries of the diffuse hits (NaN normal elsewhere), then irradiance -> pixels
mr_status launch_gather_queries(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, unsigned long long n,
                                float *d_pos, float *d_nrm, hipStream_t stream);
mr_status launch_gather_accumulate(const float *d_irr_a, const float *d_irr_b, unsigned long long n, uint32_t spp,
                                   float *d_rgb, hipStream_t stream);
mr_status launch_tonemap(const float *d_rgb, unsigned long long n_values, uint8_t *d_out, hipStream_t stream);
mr_status launch_deinterleave(const float *d_recv, float *d_full, uint32_t W, uint32_t H, uint32_t band_rows, uint32_t world,
                              uint32_t shard_rows, uint32_t fpp, hipStream_t stream);
mr_status launch_untile(const float *d_slots, float *d_image, uint32_t W, uint32_t rows, uint32_t spp, uint32_t channels,
                        hipStream_t stream);

// the fused direct-light frame (mr_frame.hip)
mr_status launch_frame_b256(const DeviceScene &ds, const mr_frame_desc &fd, float *d_rgb, mr_hit *d_hits, mr_hit *d_shadow_hits,
                            unsigned long long *d_counts, unsigned long long *work_counter, hipStream_t stream);
mr_status launch_frame_b128(const DeviceScene &ds, const mr_frame_desc &fd, float *d_rgb, mr_hit *d_hits, mr_hit *d_shadow_hits,
                            unsigned long long *d_counts, unsigned long long *work_counter, hipStream_t stream);
// mr_frame.hip in its two workgroup sizes: frames of up to ~10 M samples (1080p at 4 spp) in 128-thread workgroups
inline mr_status launch_frame(const DeviceScene &ds, const mr_frame_desc &fd, float *d_rgb, mr_hit *d_hits, mr_hit *d_shadow_hits,
                              unsigned long long *d_counts, unsigned long long *work_counter, hipStream_t stream) {
    const unsigned long long rows = fd.band_world > 1 ? (fd.H + fd.band_world - 1) / fd.band_world : (fd.y1 > fd.y0 ? fd.y1 - fd.y0 : 0);
    const unsigned long long samples = rows * fd.W * fd.spp;
    return samples <= 10000000ull ? launch_frame_b128(ds, fd, d_rgb, d_hits, d_shadow_hits, d_counts, work_counter, stream)
                                  : launch_frame_b256(ds, fd, d_rgb, d_hits, d_shadow_hits, d_counts, work_counter, stream);
}

mr_status launch_shade_accumulate(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, const float *d_weights,
                                  const uint32_t *d_pixels, unsigned long long n, const mr_ray *d_shadow_rays,
                                  const mr_hit *d_shadow_hits, const uint32_t *d_shadow_src,
                                  const unsigned long long *d_shadow_count, float *d_light_scale, const mr_light &light,
                                  uint32_t spp, float *d_rgb, hipStream_t stream);
mr_status launch_secondary_rays(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, const float *d_weights,
                                const uint32_t *d_pixels, unsigned long long n, uint32_t spp, mr_ray *d_out_rays,
                                float *d_out_weights, uint32_t *d_out_pixels, unsigned long long *d_count,
                                unsigned long long out_capacity, uint8_t *d_out_octants, hipStream_t stream);

// PATH_TRACING generators (mr_bounce.hip): kinds bit 0 mirror, 1 refraction pair, 2 diffuse bounce
mr_status launch_path_rays(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, const float *d_weights,
                           const uint32_t *d_pixels, const uint32_t *d_ids, unsigned long long n, uint32_t spp, uint32_t seed,
                           uint32_t bounce, uint32_t kinds, mr_ray *d_out_rays, float *d_out_weights, uint32_t *d_out_pixels,
                           uint32_t *d_out_ids, unsigned long long *d_count, unsigned long long out_capacity, uint8_t *d_out_octants, hipStream_t stream);

mr_status launch_level(const DeviceScene &ds, const mr_level_desc &ld, const mr_ray *d_rays, const float *d_weights,
                       const uint32_t *d_pixels, const uint32_t *d_ids, unsigned long long n, float *d_rgb, mr_ray *d_out_rays,
                       float *d_out_weights, uint32_t *d_out_pixels, uint32_t *d_out_ids, unsigned long long *d_out_count,
                       unsigned long long *d_counts, hipStream_t stream);

// photon map on the device: three float4 planes in kd-tree heap order, 1-based (children of i: 2i, 2i+1)
struct PhotonMapDev {
    float4 *rec = nullptr;        // two per photon, one 32-byte record: (x, y, z, split axis as int bits), then the incoming
                                  // direction de-quantised through the reference's tables -- both halves in one cache line for the
                                  // divergent fetches of the reference-order pre-pass
    float4 *power = nullptr;      // (r, g, b, -)
    int32_t n = 0, half = 0;      // stored photons; nodes with index < half descend (PhotonMap.cpp:160,357)
    // Bounding boxes of the tree's BLOCKS (the 63 nodes of six levels below a block root r = 64^L + i, L = 0 .. layers - 1):
    // four float4 per block id = layer_base[L] + i: (lo, hi) of the block's own 63 photons, (lo, hi) of every photon below
    // its root.  The cooperative search finds the blocks that touch the search sphere with these instead of walking planes.
    float4 *boxes = nullptr;
    int32_t layers = 0, layer_base[4] = {0, 0, 0, 0};
    // hand-out counters of the estimate launches (mr_photon.hip): zero between launches, one per launch in flight
    unsigned *work_counters = nullptr;
};
constexpr uint32_t kPhotonWorkCounters = 16;
constexpr int kKnnMaxK = 512;     // nphotons limit of the wave-cooperative k-NN (PHOTON_SAMPLES = 500)
mr_status launch_irradiance(const PhotonMapDev &pm, unsigned *work_counter, const float *d_pos, const float *d_normal, unsigned long long nq,
                            float max_dist, uint32_t k, float *d_irrad, int32_t *d_found, float *d_r2, unsigned long long *d_stats,
                            hipStream_t stream);
constexpr int kPhotonStats = 12;  // see mr_photon_map_get_stats (miro_hip.h)

}  // namespace mr

struct mr_scene {
    int32_t device = 0;
    mr::HostMesh mesh;
    mr::HostTree tree;
    bool built = false;       // host tree exists
    bool on_device = false;   // device records uploaded
    mr::DeviceScene dev;
    unsigned long long *d_stats = nullptr;
    unsigned long long *d_work_counters = nullptr;   // ring of kWorkCounters hand-out counters
    std::atomic<uint32_t> next_counter{0};
    // grow-only staging buffers for host-pointer traces; stage_mutex serialises the calls that use them, so that
    // mr_trace may be called concurrently from several host threads (Scene::trace is const and re-entrant,
    // Scene.cpp:112-115 calls it from every OpenMP worker)
    void *d_stage_rays = nullptr, *d_stage_hits = nullptr;
    uint64_t stage_cap = 0;
    std::mutex stage_mutex;
    // large host-pointer traces are cut into chunks: upload of chunk k+1, kernel of chunk k and download of chunk k-1
    // overlap on these copy streams (events order them against the caller's stream)
    void *copy_in = nullptr, *copy_out = nullptr;   // hipStream_t
    std::vector<void *> stage_events;               // hipEvent_t, grow-only
    // grow-only per-primary-ray occlusion flags for mr_shade_direct
    uint8_t *d_occluded = nullptr;
    uint64_t occluded_cap = 0;
    // grow-only per-ray light attenuation for mr_shade_accumulate
    float *d_light_scale = nullptr;
    uint64_t light_scale_cap = 0;
    // materials (host copy; uploaded by mr_scene_set_materials / mr_bvh_build)
    std::vector<float> materials;          // 11 per material, clamped as the Phong constructor does
    std::vector<uint32_t> prim_material;   // empty: material 0 everywhere
};
