/*
 * miro_hip.h -- C ABI of the MI355X-native intersection path for the Miro ray tracer
 * (hallgeirl/cse168-raytracer).  Plain C, plain pointers and sizes; no C++ or torch types.
 *
 * The reference has no FFI layer: its hot path is the C++ surface
 *     bool Scene::trace(HitInfo&, const Ray&, float tMin, float tMax) const   Scene.h:38-39, Scene.cpp:214-268
 *     void BVH::build(Objects*, int depth)                                    BVH.h:33,      BVH.cpp:60-339
 *     bool BVH::intersect(HitInfo&, const Ray&, float tMin, float tMax) const BVH.h:35-36,   BVH.cpp:438-469
 * called from Scene.cpp:72 (build), :217/:278/:539 and Phong.cpp:97 (trace).  The entry points
 * below are what a binding for that surface binds: scene assembly (Scene::addObject,
 * TriangleMesh::load), BVH::build, and a *batched* Scene::trace (a single-ray call is a
 * batch of one).  cse168-raytracer_amd/host/miro_shim.hpp re-creates the C++ signatures on top
 * of these functions; INTEGRATION.md shows the reference-side glue.
 *
 * All functions return MR_OK (0) or a negative mr_status; mr_last_error() gives a
 * thread-local message.  No exception crosses this boundary.  There is NO CPU fallback:
 * without a HIP device every device-touching call fails with MR_ERR_HIP.
 */
#ifndef MIRO_HIP_H
#define MIRO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t mr_status;
enum {
    MR_OK          =  0,
    MR_ERR_INVALID = -1,   /* bad argument */
    MR_ERR_IO      = -2,   /* file could not be opened (TriangleMesh::load returns false, TriangleMeshLoad.cpp:66-71) */
    MR_ERR_NOMEM   = -3,
    MR_ERR_HIP     = -4,   /* HIP runtime error or no device */
    MR_ERR_STATE   = -5    /* call out of order (trace before build, add after build) */
};

/* Ray (Ray.h:40-84): origin, direction and the [tMin,tMax] interval that Scene::trace
 * takes as separate arguments.  32 bytes, two float4 loads per lane on the device. */
typedef struct mr_ray { float ox, oy, oz, tmin, dx, dy, dz, tmax; } mr_ray;

/* HitInfo (Ray.h:21-38) in its minimal device form.  prim indexes triangles in
 * Scene::addObject order (= concatenated mesh order); beta/gamma are the locals of
 * Triangle.cpp:155-156.  On a miss prim == MR_MISS, t == tmax (BVH.cpp:444), beta=gamma=0.
 * P, N, material, object are rebuilt by the shim (Triangle.cpp:160-166). 16 bytes. */
typedef struct mr_hit { float t; uint32_t prim; float beta, gamma; } mr_hit;
#define MR_MISS 0xFFFFFFFFu
/* A hit on an unbounded object (a Plane, kept in Scene::m_unboundedObjects, Scene.h:22-23) reports
 * prim = MR_PLANE_BIT | its index in that list; spheres are bounded objects and use ordinary indices.
 * beta = gamma = 0 for both; P = o + t*d (Sphere.cpp:61, Plane.cpp:42). */
#define MR_PLANE_BIT 0x80000000u

/* Indexed triangle mesh as TriangleMesh stores it (TriangleMesh.h:27-67). */
typedef struct mr_mesh_desc {
    const float    *vertices;  uint32_t n_vertices;   /* xyz triples */
    const float    *normals;   uint32_t n_normals;    /* xyz triples */
    const uint32_t *vidx;                             /* 3 per triangle */
    const uint32_t *nidx;                             /* 3 per triangle */
    uint32_t        n_triangles;
} mr_mesh_desc;

typedef struct mr_build_opts {
    uint32_t leaf_size;     /* OBJECTS_PER_LEAF (BVH.h:59,61): 4 = scalar reference build (default when 0) */
    uint32_t builder;       /* MR_BUILD_REFERENCE (0): the reference's binary-search split, identical tree */
    uint32_t host_only;     /* 1: build the tree on the host and skip the device upload (tree inspection,
                               CPU-only tooling); mr_trace on such a scene fails with MR_ERR_STATE */
    uint32_t layout;        /* storage order of the device records (MR_LAYOUT_*): a permutation of where the same nodes and
                               triangles live in HBM -- references, visiting order and hit records are unaffected */
    uint32_t reserved[4];
} mr_build_opts;
enum { MR_BUILD_REFERENCE = 0 };
enum {
    MR_LAYOUT_DFS      = 0,   /* inner nodes in depth-first pre-order (a node's first inner child is its neighbour) */
    MR_LAYOUT_PAIRS    = 1,   /* pre-order, padded so that a node and its first inner child always share one 128-byte line */
    MR_LAYOUT_TREELETS = 2,   /* the top 12 levels breadth-first, below them treelets of three levels stored contiguously */
    MR_LAYOUT_ALIGN_LEAVES = 16 /* OR-ed in: dummy triangle records in front of a leaf whenever that lets its triangles touch
                                   fewer 128-byte lines */
};

typedef struct mr_scene_info {
    uint32_t n_vertices, n_normals, n_triangles;
    uint32_t n_nodes, n_leaves, max_depth;            /* Stats::BVH_Nodes / BVH_LeafNodes (BVH.cpp:64,88) */
    uint32_t leaf_size, built;
    uint64_t device_bytes;                            /* node + triangle + index arrays resident in HBM */
    int32_t  device;
    uint32_t reserved[3];
} mr_scene_info;

typedef struct mr_camera {                            /* Camera.h:53-60 */
    float eye[3], lookat[3], up[3], fov_deg;
} mr_camera;

/* mr_trace flags */
enum {
    MR_TRACE_CLOSEST  = 0u,        /* Scene::trace semantics (closest hit, strict-less replacement) */
    MR_TRACE_ANY      = 1u << 0,   /* stop at the first accepted hit; legal for shadow batches only when the
                                      scene has no refractive material (Phong.cpp:99-113) */
    MR_RAYS_ON_DEVICE = 1u << 1,   /* rays is a device pointer */
    MR_HITS_ON_DEVICE = 1u << 2,   /* hits is a device pointer */
    MR_MATH_FAST      = 1u << 3,   /* fused multiply-add + v_rcp_f32: same primitive, t within 1e-5 relative of the
                                      reference, beta/gamma only within the formula's own rounding sensitivity
                                      (~|o-A|/|edge| ulps).  The default is the bit-exact IEEE expression tree
                                      of Triangle.cpp:150-156 and is what parity and the bench are quoted on */
    MR_COUNT_STATS    = 1u << 4,   /* accumulate -DSTATS counters (BVH.cpp:461,496,632,643) */
    MR_TRACE_PERSISTENT = 1u << 5, /* incoherent batches: resident waves pull rays from a counter and re-arm idle
                                      lanes by wave64 ballot + prefix sum.  Same arithmetic and hit records as the
                                      one-shot kernel (exact quotients, or products with MR_MATH_PRODUCT); on par with
                                      it on random rays, slower on coherent camera rays -- off by default;
                                      triangle scenes only: ignored, i.e. the default kernel runs, when the scene holds
                                      spheres or planes, and under MR_COUNT_STATS / MR_MATH_FAST) */
    MR_TRACE_INCOHERENT = 1u << 7, /* batch hint: the 64 rays of a wave do not share their path through the tree (secondary,
                                      random or 1-sample-per-pixel batches).  Selects the voting control flow: every
                                      iteration a wave runs the step (node test / triangle test) most of its lanes need,
                                      instead of running node tests until its slowest lane has found a leaf.  Same per-ray
                                      steps in the same order: identical hit records.  Combines with MR_TRACE_PERSISTENT.
                                      Without the hint the default kernel decides per wave: a wave whose rays point into
                                      several octants votes, one whose rays share an octant does not */
    MR_FRAME_NO_SHADOWS = 1u << 8, /* mr_render_direct only: the reference's -DDISABLE_SHADOWS build (Phong.cpp:91) -- no shadow ray
                                      is built or traced, every light reaches every hit (BASELINE config 2, "primary rays
                                      only"); default traversal only */
    MR_MATH_PRODUCT   = 1u << 6    /* slab distances as products (corner - o) * RN(1/d) instead of the reference's
                                      quotients (corner - o) / d (BVH.cpp:601-602), whose decisions the default reproduces
                                      exactly (products where the visit's comparisons are more than 16 ulp from a tie --
                                      provably the quotients' outcome --, one fma correction step per product otherwise:
                                      exactly the correctly rounded quotient for operands in the normal range; literal
                                      divisions for the rest).  A product differs
                                      from the quotient by <= 3 ulp, which can flip a box comparison only on a
                                      near-tie (observed: 0 of 3e8 rays in normal use, 2 of 1.3e8 when tMax is set one ulp
                                      above a known hit); t / beta / gamma of a hit are the same bits either way.
                                      About 25 % faster */
};

typedef struct mr_scene mr_scene;

/* ---- scene assembly: Scene::addObject / TriangleMesh::load / createSingleTriangle -------------- */
/* A scene lives on one device; every call that touches it makes that device the calling thread's current HIP
 * device (hipSetDevice) and leaves it so -- the intended deployment is one process (or thread) per GPU. */
mr_status mr_scene_create(int32_t device, mr_scene **out);
mr_status mr_scene_destroy(mr_scene *scene);
/* copies the arrays; triangles are appended in order (assignment2.cpp:449-461) */
mr_status mr_scene_add_mesh(mr_scene *scene, const mr_mesh_desc *mesh);
/* TriangleMesh::load(file, ctm) (TriangleMeshLoad.cpp:63-311); ctm = 16 floats row-major or NULL */
mr_status mr_scene_add_obj(mr_scene *scene, const char *path, const float *ctm16, uint32_t *n_triangles_out);
/* TriangleMesh::createSingleTriangle + setV1..3/setN1..3 (TriangleMeshLoad.cpp:15-56) */
mr_status mr_scene_add_triangle(mr_scene *scene, const float v[9], const float n[9]);
/* Sphere + setCenter/setRadius + Scene::addObject (Sphere.h:13-21, Sphere.cpp:28-69): the next bounded object;
 * *prim_out (may be NULL) receives its object index, which mr_hit.prim and prim_material use. */
mr_status mr_scene_add_sphere(mr_scene *scene, const float center[3], float radius, uint32_t *prim_out);
/* Plane + setNormal/setOrigin + Scene::addObject (Plane.h:22-27, Plane.cpp:33-48): the next unbounded object,
 * scanned after the BVH by every trace (Scene.cpp:220-230).  `material` indexes mr_scene_set_materials' table;
 * *index_out (may be NULL) receives the plane's index. */
mr_status mr_scene_add_plane(mr_scene *scene, const float normal[3], const float origin[3], uint32_t material,
                             uint32_t *index_out);

/* ---- BVH::build (BVH.cpp:60-339) via Scene::preCalc (Scene.cpp:50-84); uploads the scene ------- */
mr_status mr_bvh_build(mr_scene *scene, const mr_build_opts *opts);

mr_status mr_scene_get_info(const mr_scene *scene, mr_scene_info *info);
/* host copies of the merged mesh (for the shim's P/N/material reconstruction) */
mr_status mr_scene_get_mesh(const mr_scene *scene, mr_mesh_desc *out);
/* tree in DFS pre-order: corners6[n_nodes*6], meta3[n_nodes*3] = (is_leaf, child0|first, child1|count),
 * leaf_prims[n_triangles].  Any pointer may be NULL. */
mr_status mr_scene_export_tree(const mr_scene *scene, float *corners6, int32_t *meta3, uint32_t *leaf_prims);

/* ---- Scene::trace, batched (Scene.cpp:214-268 -> BVH.cpp:438-658 -> Triangle.cpp:136-169) ------- */
/* stream: a hipStream_t (NULL = default stream).  Host buffers are staged and the call returns when the hits are
 * in `hits`; batches above 2^20 rays are cut into chunks whose upload, trace and download overlap (fully so when the
 * buffers are pinned, see mr_host_alloc; pageable memory is staged by the HIP runtime on the calling thread).
 * With both buffers on the device the call only enqueues work on `stream`.
 * Threads: a built scene is immutable and mr_trace / mr_trace_indirect may be called on it from several host
 * threads at once, like the reference's const Scene::trace from its OpenMP workers (Scene.cpp:112-115); calls
 * with host buffers take turns on the scene's staging buffers.  mr_shade_direct and mr_shade_accumulate keep
 * per-scene scratch (occlusion flags, light scale): one such call in flight per scene. */
mr_status mr_trace(mr_scene *scene, const mr_ray *rays, uint64_t n_rays, mr_hit *hits,
                   uint32_t flags, void *stream);
/* Page-locked host memory for ray / hit buffers handed to mr_trace (the reference's callers keep Ray and HitInfo in
 * ordinary `new`-ed memory, Scene.cpp:117-140; pinned buffers let the copies run at PCIe speed and overlap the trace).
 * Usable from any device; free with mr_host_free (NULL is a no-op). */
mr_status mr_host_alloc(void **ptr, uint64_t bytes);
mr_status mr_host_free(void *ptr);
/* Same, for a batch whose size was produced on the device (the compacted shadow batch of
 * mr_gen_shadow_rays): traces min(*d_count, max_rays) rays without a host round trip.  All pointers are
 * device pointers; MR_RAYS_ON_DEVICE / MR_HITS_ON_DEVICE are implied. */
mr_status mr_trace_indirect(mr_scene *scene, const mr_ray *d_rays, const uint64_t *d_count, uint64_t max_rays,
                            mr_hit *d_hits, uint32_t flags, void *stream);
/* The same trace for a BOUNCE QUEUE (device buffers): a generator writes children in the order of their parents, so the
 * rays of a wave start at neighbouring surface points but point into all eight octants.  This call first writes into
 * d_order (n uint32) the ray indices grouped by direction octant inside consecutive chunks of 2^chunk_log2 rays (8 ... 14;
 * 0 = 14; a stable counting sort -- no ray is moved), then traces ray d_order[k] in lane k and stores its hit at
 * d_hits[d_order[k]]: the hit buffer is byte for byte that of mr_trace on the same rays; whole waves share a direction
 * sign (the octant-specialised loops apply) and rays that leave the scene at once stop holding waves.  Measured on the
 * stand-in atrium's diffuse-bounce queue: 5.2 -> 6.0 Grays/s; on the bunny's: 22 -> 31 (profiles/r03_octant_order.log).
 * flags: MR_TRACE_ANY, MR_MATH_PRODUCT, MR_TRACE_INCOHERENT.
 * d_octants (may be NULL): one byte per ray holding the sign bits of its direction (x<0 | y<0 << 1 | z<0 << 2), as the
 * generators write them (d_out_octants of mr_gen_secondary_rays / mr_gen_path_rays / mr_level_desc); the order is then made
 * from 1 byte per ray instead of the 32-byte rays -- the order kernel is bandwidth-bound, this is what it costs.  The bytes are
 * trusted to be the rays' octants: other values change the grouping, never the hit buffer.
 * chunk_log2 | MR_ORDER_GIVEN: d_order is an INPUT -- a permutation of 0 ... n-1 the caller made (mr_order_by_octant, or the
 * one an earlier call on the same rays left there); it is not checked, an index >= n reads and writes out of bounds.
 * mr_order_by_octant is the first half alone (d_rays may be NULL when d_octants is given): for mr_level_desc.d_order. */
#define MR_ORDER_GIVEN 0x80000000u
mr_status mr_order_by_octant(mr_scene *scene, const mr_ray *d_rays, const uint8_t *d_octants, uint64_t n, uint32_t chunk_log2,
                             uint32_t *d_order, void *stream);
mr_status mr_trace_grouped(mr_scene *scene, const mr_ray *d_rays, const uint8_t *d_octants, uint64_t n, mr_hit *d_hits,
                           uint32_t *d_order, uint32_t chunk_log2, uint32_t flags, void *stream);
/* -DSTATS counters accumulated by MR_COUNT_STATS traces (synchronises the device) */
mr_status mr_trace_get_stats(mr_scene *scene, uint64_t *box_tests, uint64_t *tri_tests, int32_t reset);

/* ---- callers of the path, on the device ("next" rows: Camera::eyeRay, Phong shadow ray) ---------- */
/* Camera::eyeRay (Camera.cpp:104-161) for rows [y0,y1), spp samples per pixel, ray index
 * ((y-y0)*W+x)*spp+s.  jitter=0: pixel centres (randomize=false).  d_rays: device pointer. */
mr_status mr_gen_eye_rays(mr_scene *scene, const mr_camera *cam, uint32_t W, uint32_t H,
                          uint32_t y0, uint32_t y1, uint32_t spp, uint32_t jitter, uint32_t seed,
                          mr_ray *d_rays, void *stream);
/* The same rays in TILED order, for frames with fewer than 64 samples per pixel: the reference traces pixel by pixel
 * (Scene.cpp:112-140) and has no batch order to keep, and a wave whose 64 rays cover a square of pixels shares far more
 * of its BVH path than one on a 64 x 1 strip.  A pixel's spp rays stay consecutive (slot p = rays p*spp .. p*spp+spp-1);
 * the window's rows are taken in groups of th, each group in blocks of tw pixels, each block row by row, with
 * th x tw = 8x8, 8x4, 4x4, 4x2, 2x2, 2x1 for spp = 1, 2, 4, 8, 16, 32 (image order for any other spp).
 * mr_tile_pixel_map writes, for the same window, pixel_of_slot[p] = (y - y0) * W + x into a HOST array of W * rows
 * entries: everything downstream that works per ray (trace, shadow rays, shade) is order-agnostic, and a frame
 * buffer shaded in slot order is scattered to image order with this map. */
mr_status mr_gen_eye_rays_tiled(mr_scene *scene, const mr_camera *cam, uint32_t W, uint32_t H,
                                uint32_t y0, uint32_t y1, uint32_t spp, uint32_t jitter, uint32_t seed,
                                mr_ray *d_rays, void *stream);
mr_status mr_tile_pixel_map(uint32_t W, uint32_t rows, uint32_t spp, uint32_t *pixel_of_slot);
/* The scatter itself, on the device: d_image[((y - y0) * W + x) * channels + c] = d_slots[p * channels + c] for every
 * pixel slot p of the window (channels = 3 for the float framebuffer of mr_shade_direct).  Not in place. */
mr_status mr_untile_pixels(mr_scene *scene, const float *d_slots, float *d_image, uint32_t W, uint32_t rows,
                           uint32_t spp, uint32_t channels, void *stream);
/* Phong::shade shadow ray (Phong.cpp:80-97) for every hit, compacted with a wave64 ballot /
 * prefix sum.  d_out needs room for n rays; d_src[k] = index of the originating ray (may be NULL);
 * d_count: device uint64 receiving the number of shadow rays (zeroed by the call).
 * The compacted order is wave-granular, not ray order; use d_src to match. */
mr_status mr_gen_shadow_rays(mr_scene *scene, const mr_ray *d_rays, const mr_hit *d_hits, uint64_t n,
                             const float light[3], mr_ray *d_out, uint32_t *d_src, uint64_t *d_count,
                             void *stream);
/* HitInfo::P and ::N as the object's intersect() leaves them (Triangle.cpp:160,162; Sphere.cpp:61-63;
 * Plane.cpp:42-44), device buffers of 3 floats per ray (either may be NULL).  d_rays (the rays the hits belong
 * to) may be NULL for scenes of triangles only. */
mr_status mr_hit_attrs(mr_scene *scene, const mr_ray *d_rays, const mr_hit *d_hits, uint64_t n, float *d_P, float *d_N,
                       void *stream);

/* ---- Phong::shade for one point light over a traced frame ("next" row: the consumer of the shadow batch) --- */
typedef struct mr_light {                             /* PointLight.h:8-59 */
    float position[3], color[3], wattage;
} mr_light;
/* For n primary rays (spp consecutive rays per pixel) with their hits, and the traced shadow batch of
 * mr_gen_shadow_rays (hits + source indices + device count): direct lighting of a uniform material with
 * diffuse colour `diffuse` as Phong::shade computes it (Phong.cpp:44-160; opaque occluders only: MR_ERR_STATE when the
 * scene's material table holds a refractive material -- mr_shade_accumulate handles those), normals
 * normalised as Scene::trace does (Scene.cpp:262), misses = background 0 (Scene.cpp:340,685), averaged over
 * the spp samples of each pixel (Scene.cpp:126-139) into d_rgb[(n/spp)*3] -- the linear float framebuffer
 * (tempImage, Scene.cpp:106).  All pointers are device pointers. */
mr_status mr_shade_direct(mr_scene *scene, const mr_ray *d_rays, const mr_hit *d_hits, uint64_t n,
                          const mr_hit *d_shadow_hits, const uint32_t *d_shadow_src, const uint64_t *d_shadow_count,
                          const mr_light *light, const float diffuse[3], uint32_t spp, float *d_rgb, void *stream);
/* ---- the whole direct-light frame step in ONE launch ------------------------------------------------------------
 * Scene::raytraceImage's loop for a window of rows (Scene.cpp:112-141): for every sample Camera::eyeRay ->
 * Scene::trace -> the shadow ray of Phong::shade (Phong.cpp:80-97) -> Scene::trace -> Phong::shade -> the pixel's
 * mean.  The same rays, hit records and pixels, bit for bit, as
 *   mr_gen_eye_rays[_tiled] -> mr_trace -> mr_gen_shadow_rays -> mr_trace_indirect -> mr_shade_direct [-> mr_untile_pixels]
 * without any ray buffer: rays live in registers, the shadow ray is built from the hit the lane still holds.
 * Rows of the window: [y0,y1) when band_world == 1; otherwise the interleaved bands of one rank of a multi-GPU frame
 * (bands of band_rows rows dealt round-robin: rank r owns bands r, r + band_world, ...; y0/y1 ignored).
 * spp: a power of two <= 64 (other counts: use the batched calls).  tiled: sample order of mr_gen_eye_rays_tiled over
 * the window's rows (sample k of d_hits / d_shadow_hits follows that order); d_rgb is always in image order.
 * flags: MR_MATH_PRODUCT, MR_TRACE_INCOHERENT (both rays), MR_TRACE_ANY (shadow ray only), MR_FRAME_NO_SHADOWS.
 * Material: without mr_scene_set_materials the uniform Phong material of `diffuse` (what mr_shade_direct shades, and the
 * batched equivalence above holds bit for bit); once the scene has a material table, Phong::shade uses the material of the
 * object that was hit (Phong.cpp:116-156) and lets light through refractive occluders scaled by dot(N, l) of the occluder
 * (Phong.cpp:99-113) -- `diffuse` is ignored, the frame equals mr_trace_level(MR_LEVEL_LAST) over the eye rays, and
 * MR_TRACE_ANY is refused when a material is refractive (MR_ERR_STATE); default traversal only. */
typedef struct mr_frame_desc {
    mr_camera camera;
    uint32_t  W, H, y0, y1;
    uint32_t  band_rows, band_rank, band_world;   /* band_world <= 1: the contiguous window [y0,y1) */
    uint32_t  spp, jitter, seed, tiled, flags;
    mr_light  light;
    float     diffuse[3];
    uint32_t  reserved[4];
} mr_frame_desc;
/* d_rgb: rows*W*3 floats (window rows in band order).  Optional device outputs (NULL to skip): d_hits / d_shadow_hits,
 * rows*W*spp records each -- the shadow record of a sample whose primary ray missed is {t = 0, prim = MR_MISS};
 * d_counts[2]: += primary rays, += shadow rays traced (not zeroed by the call).
 * Streams: a frame of 60 000 chunks of 256 samples or more hands the last 6 % of its chunks out through one of 64 per-scene
 * device counters (taken round-robin per call, re-armed by the launch itself, so a captured HIP graph replays): calls on one
 * scene may overlap on different streams, up to 64 at a time; a captured graph keeps its counter -- do not replay it while
 * other frames of the same scene are in flight on other streams. */
mr_status mr_render_direct(mr_scene *scene, const mr_frame_desc *frame, float *d_rgb, mr_hit *d_hits, mr_hit *d_shadow_hits,
                           uint64_t *d_counts, void *stream);

/* ---- multi-GPU frames: image rows dealt to the devices in interleaved bands -------------------
 * Bands of band_rows rows are dealt round-robin: band b (rows [b*band_rows, ...)) belongs to rank b % world and is that
 * rank's band number b / world; a rank keeps its rows in band order.  mr_band_locate answers, for image row y, which
 * rank owns it and where it sits in that rank's shard; mr_band_rows_of counts a rank's rows (host arithmetic, no device).
 * The reference hands rows to OpenMP workers two at a time (`schedule(dynamic, 2)`, Scene.cpp:113). */
mr_status mr_band_locate(uint32_t H, uint32_t band_rows, uint32_t world, uint32_t y, uint32_t *rank, uint32_t *local_row);
mr_status mr_band_rows_of(uint32_t H, uint32_t band_rows, uint32_t rank, uint32_t world, uint32_t *rows);
/* The de-interleave after the frame's single gather, on the device: d_recv holds `world` shards of shard_rows rows each
 * (shard_rows >= the largest rank's row count; rows of W pixels, floats_per_pixel floats per pixel: 3 for the float
 * framebuffer, 4 * spp for mr_hit records); row y of d_full (H rows) is copied from its owner's shard. */
mr_status mr_deinterleave_bands(mr_scene *scene, const float *d_recv, float *d_full, uint32_t W, uint32_t H, uint32_t band_rows,
                                uint32_t world, uint32_t shard_rows, uint32_t floats_per_pixel, void *stream);

/* ---- specular materials and secondary rays ("next" row: Scene::traceScene's recursion, Scene.cpp:302-336) -------- */
typedef struct mr_material {                          /* Phong(kd, ks, kt, shininess, refractIndex), Phong.h:10-14 */
    float diffuse[3], specular[3], transmission[3], shininess, refract_index;
} mr_material;
/* Materials of the scene (the Phong constructor's energy clamps are applied, Phong.cpp:12-33) and the material id
 * of every triangle in addObject order (NULL: material 0 everywhere).  Without this call every triangle is the white
 * Lambert of the BASELINE scenes.  May be called before or after mr_bvh_build. */
mr_status mr_scene_set_materials(mr_scene *scene, const mr_material *materials, uint32_t n_materials,
                                 const uint32_t *prim_material);
/* Phong::shade with per-triangle materials for a batch of rays of any bounce: d_weights (rgb per ray, NULL = 1) is the
 * product of reflection / transmission factors along the path, d_pixels (NULL = ray index / spp) the pixel the ray
 * contributes to; light through refractive occluders is attenuated as Phong.cpp:99-113 does (needs the shadow rays
 * themselves besides their hits).  Adds weight * L / spp to d_rgb[pixel] with float atomics. */
mr_status mr_shade_accumulate(mr_scene *scene, const mr_ray *d_rays, const mr_hit *d_hits, const float *d_weights,
                              const uint32_t *d_pixels, uint64_t n, const mr_ray *d_shadow_rays, const mr_hit *d_shadow_hits,
                              const uint32_t *d_shadow_src, const uint64_t *d_shadow_count, const mr_light *light, uint32_t spp,
                              float *d_rgb, void *stream);
/* Ray::reflect / getReflectionCoefficient / refract (Ray.h:143-243) for every hit on a reflective or refractive
 * material: up to three children per ray (room for 3n), compacted by wave64 ballot + prefix sum, each with its path
 * weight and pixel.  d_count: device uint64 receiving the number of children (zeroed by the call).
 * out_capacity: rays the output arrays have room for.  Children beyond it are counted but not stored: *d_count >
 * out_capacity afterwards means the queue was too small (3n always suffices) -- nothing is written out of bounds.
 * d_out_octants (may be NULL): one byte per child, the sign bits of its direction -- mr_trace_grouped's d_octants. */
mr_status mr_gen_secondary_rays(mr_scene *scene, const mr_ray *d_rays, const mr_hit *d_hits, const float *d_weights,
                                const uint32_t *d_pixels, uint64_t n, uint32_t spp, mr_ray *d_out_rays, float *d_out_weights,
                                uint32_t *d_out_pixels, uint64_t *d_count, uint64_t out_capacity, uint8_t *d_out_octants,
                                void *stream);

/* The PATH_TRACING build of those generators (Ray.h:149-158,235-239) plus Ray::random (Ray.h:124-140): every child is
 * drawn from a lobe (alignHemisphereToVector, Utility.h:34-50) around the mirror / refracted direction with
 * phi = acos(pow(u1, 1/(1+shininess))), or -- the diffuse bounce -- around the normal with phi = asin(sqrt(u1));
 * theta = 2 pi u2.  u1, u2 replace the reference's rand() by the counter-based generator of the eye-ray jitter, keyed by
 * (seed, ray id, bounce, child kind).  kinds selects the children: MR_PATH_MIRROR | MR_PATH_REFRACT (Scene.cpp:302-336)
 * | MR_PATH_DIFFUSE (an extension: traceScene at HEAD never calls Ray::random); up to four children per ray (room for
 * 4n).  d_ids (NULL = ray index) are stable ray ids, d_out_ids (may be NULL) receives the children's.  out_capacity as
 * in mr_gen_secondary_rays (4n always suffices; n when kinds == MR_PATH_DIFFUSE). */
enum { MR_PATH_MIRROR = 1u, MR_PATH_REFRACT = 2u, MR_PATH_DIFFUSE = 4u };
mr_status mr_gen_path_rays(mr_scene *scene, const mr_ray *d_rays, const mr_hit *d_hits, const float *d_weights,
                           const uint32_t *d_pixels, const uint32_t *d_ids, uint64_t n, uint32_t spp, uint32_t seed,
                           uint32_t bounce, uint32_t kinds, mr_ray *d_out_rays, float *d_out_weights, uint32_t *d_out_pixels,
                           uint32_t *d_out_ids, uint64_t *d_count, uint64_t out_capacity, uint8_t *d_out_octants, void *stream);

/* ---- one level of Scene::traceScene's recursion (Scene.cpp:270-346) in ONE launch -------------------------------
 * For every ray of the queue: Scene::trace -> Phong::shade (shadow ray, Scene::trace, the occluder's light scale,
 * diffuse term + highlight, Phong.cpp:80-156) times the ray's weight, added to its pixel -> the children of the next
 * level.  The same hit records, shadow rays and children, bit for bit, as
 *   mr_trace -> mr_gen_shadow_rays -> mr_trace_indirect -> mr_shade_accumulate -> mr_gen_secondary_rays | mr_gen_path_rays
 * without the buffers between them: the hit stays in the lane's registers, the shadow ray is built from it and traced
 * by the same lane.  The children arrive in another ORDER than the batched generators' (compare the queues as sets,
 * by ray id under path tracing); the pixel sums differ by the order of the float atomics, as two runs of
 * mr_shade_accumulate do.
 * children: MR_LEVEL_LAST (none: the queue's rays are shaded only; the d_out_* may be NULL), MR_LEVEL_SPECULAR (the
 * generators of mr_gen_secondary_rays, room for 3n) or MR_LEVEL_PATH (those of mr_gen_path_rays with path_kinds, seed
 * and bounce as there, room for 4n).  flags: MR_MATH_PRODUCT, MR_TRACE_INCOHERENT.  d_weights / d_pixels / d_ids /
 * d_out_ids may be NULL as in mr_gen_path_rays.  d_out_count: zeroed by the call, receives the number of children.
 * out_capacity_lo / _hi (a 64-bit count in two words): rays the output queue has room for, required (> 0) for a level with
 * children; children beyond it are counted, not stored (*d_out_count > capacity: queue too small, nothing out of bounds).
 * d_counts (may be NULL): [0] += rays traced, [1] += shadow rays traced. */
enum { MR_LEVEL_LAST = 0u, MR_LEVEL_SPECULAR = 1u, MR_LEVEL_PATH = 2u };
typedef struct mr_level_desc {
    mr_light light;
    uint32_t spp, flags, children;
    uint32_t path_kinds, seed, bounce;
    uint32_t out_capacity_lo, out_capacity_hi;
    uint32_t reserved;
    uint8_t *d_out_octants;       /* may be NULL: per child, the sign bits of its direction (mr_order_by_octant's input) */
    const uint32_t *d_order;      /* may be NULL: lane k works on ray d_order[k] of the queue (a permutation of 0 ... n-1) */
} mr_level_desc;
mr_status mr_trace_level(mr_scene *scene, const mr_level_desc *level, const mr_ray *d_rays, const float *d_weights,
                         const uint32_t *d_pixels, const uint32_t *d_ids, uint64_t n, float *d_rgb, mr_ray *d_out_rays,
                         float *d_out_weights, uint32_t *d_out_pixels, uint32_t *d_out_ids, uint64_t *d_out_count,
                         uint64_t *d_counts, void *stream);

/* sigmoid(6v-3) tone map + 8-bit quantisation (Scene.cpp:87-91,177-202; Image.cpp:44-50) */
mr_status mr_tonemap(mr_scene *scene, const float *d_rgb, uint64_t n_values, uint8_t *d_out, void *stream);

/* ---- photon map (BASELINE config 5): Photon_map of PhotonMap.h:42-105 ---------------------------------------- */
typedef struct mr_photon_map mr_photon_map;
mr_status mr_photon_map_create(int32_t device, uint32_t max_photons, mr_photon_map **out);     /* Photon_map(max_phot) */
mr_status mr_photon_map_destroy(mr_photon_map *map);
/* Photon_map::store (PhotonMap.cpp:255-289) for n photons: power, position, incoming direction (xyz triples);
 * photons beyond max_photons are dropped silently, as in the reference */
mr_status mr_photon_map_store(mr_photon_map *map, uint32_t n, const float *power, const float *pos, const float *dir);
mr_status mr_photon_map_scale(mr_photon_map *map, float scale);        /* scale_photon_power (:298-306) */
/* Photon_map::balance (:314-359): left-balanced kd-tree in heap order; uploads unless host_only */
mr_status mr_photon_map_balance(mr_photon_map *map, uint32_t host_only);
mr_status mr_photon_map_count(const mr_photon_map *map, uint32_t *stored);
/* balanced tree in heap order (any pointer may be NULL): pos[3n], plane[n], theta_phi[2n] (quantised direction), power[3n] */
mr_status mr_photon_map_export(const mr_photon_map *map, float *pos, int32_t *plane, uint8_t *theta_phi, float *power);
/* Photon_map::irradiance_estimate (:81-145), batched: for each query (surface position + normal, xyz triples on the
 * device) the nphotons (<= 512) nearest photons within max_dist whose incoming direction faces the normal;
 * d_irrad[3q..] = sum of their powers * (1/pi)/r^2.  d_found / d_r2 (optional) receive np.found and np.dist2[0]. */
mr_status mr_irradiance_estimate(mr_photon_map *map, const float *d_pos, const float *d_normal, uint64_t n_queries,
                                 float max_dist, uint32_t nphotons, float *d_irrad, int32_t *d_found, float *d_r2,
                                 void *stream);
/* Work counters of the estimates on this map, like MR_COUNT_STATS for the traversal (the reference has no counterpart:
 * Stats.h counts nothing in PhotonMap.cpp).  While enabled, every mr_irradiance_estimate / mr_final_gather on the map runs
 * the counting build of the kernel and adds to: [0] queries answered, [1] blocks of 63 kd-tree nodes examined, [2] photon
 * records (position + direction, 32 bytes) examined by the search, [3] radius tightenings (k-th-nearest selections),
 * [4] photon records examined by the reference-order pre-pass that finds the first overflow's victim
 * (PhotonMap.cpp:195-240), [5] searches repeated because a guessed radius did not hold the k nearest, [6] child-block boxes
 * measured (64 bytes each), [7] candidates buffered, [8] blocks expanded (64 child boxes each), [10] queries searched from
 * the pre-pass's safe radius (no guess available); [9], [11] unused. */
mr_status mr_photon_map_count_stats(mr_photon_map *map, int32_t enable);
mr_status mr_photon_map_get_stats(mr_photon_map *map, uint64_t counters[12], int32_t reset);

/* The photon-map term of Scene::traceScene (Scene.cpp:285-299) for a traced batch: for every ray whose hit has a
 * diffuse material (Phong::isDiffuse), irradiance_estimate on the global and on the caustic map (either may be NULL)
 * at the hit point with the normalised normal, and (irradiance + caustic) averaged over the pixel's spp samples
 * added to d_rgb[ray / spp].  d_scratch: 12 * n floats on the device (query positions, normals, two results). */
mr_status mr_final_gather(mr_scene *scene, mr_photon_map *global_map, mr_photon_map *caustic_map, const mr_ray *d_rays,
                          const mr_hit *d_hits, uint64_t n, float max_dist, uint32_t nphotons, uint32_t spp,
                          float *d_scratch, float *d_rgb, void *stream);

const char *mr_last_error(void);
const char *mr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MIRO_HIP_H */
