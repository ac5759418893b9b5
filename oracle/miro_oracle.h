/*
 * miro_oracle.h -- CPU restatement of the Miro ray tracer's intersection hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / the timed CPU figure -- never as a fallback for the
 * HIP path (cse168-raytracer_amd/ does not link, import or call anything in here).
 *
 * What it restates (citations are file:line in the reference, hallgeirl/cse168-raytracer):
 *   OBJ loader            TriangleMeshLoad.cpp:63-311, Matrix4x4.h:284-366,581-587
 *   object bounds/centre  Triangle.cpp:41-48,97-118
 *   BVH build             BVH.cpp:14-339 (scalar: 4 per leaf; SSE: 8 per leaf)
 *   closest-hit traversal BVH.cpp:438-469 (root), :471-511 (leaf), :587-651 (scalar inner)
 *   ray/triangle test     Triangle.cpp:136-169
 *   SSE packet path       BVH.cpp:91-166,342-434,513-584; SSE.h:15-49; Ray.h:51-60
 *   primary rays          Camera.cpp:104-161 (+ Camera.h:92-110 look-at)
 *   shadow rays           Phong.cpp:80-97, PointLight.h:41-52
 *   hit point / normal    Triangle.cpp:160-162, Scene.cpp:238-263
 *   STATS counters        BVH.cpp:64,88,461,496,632,643
 *   spheres / planes      Sphere.cpp:28-69, Sphere.h:19-21, Plane.cpp:33-48, Scene.cpp:220-230, Scene.h:20-25
 *
 * Parity pin: the reference cannot be compiled in this image (every translation unit
 * reaches <GL/glut.h> through Miro.h:26 -> OpenGL.h:10, and GLUT is not installed), and it
 * ships no tests.  The restatement is therefore pinned against the known-answer
 * counters the reference itself publishes (writeup/A2/Readme.tex:91-107) and the
 * counters recorded from the genuine scalar/SSE builds in BASELINE.md section 2
 * (tests/golden/kat_counters.json).  See DESIGN.md "Oracle".
 */
#ifndef MIRO_ORACLE_H
#define MIRO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float ox, oy, oz, tmin, dx, dy, dz, tmax; } orc_ray;   /* 32 B */
typedef struct { float t; uint32_t prim; float beta, gamma; } orc_hit;  /* 16 B */
typedef struct { float eye[3], lookat[3], up[3], fov_deg; } orc_camera;
typedef struct { uint64_t box_tests, tri_tests; } orc_counters;

#define ORC_MISS 0xFFFFFFFFu

typedef struct orc_scene orc_scene;

orc_scene *orc_scene_new(void);
void       orc_scene_free(orc_scene *);

/* TriangleMesh::load + addMeshTrianglesToScene.  ctm: 16 floats row-major (m11..m44) or NULL
 * for identity.  Returns number of triangles added, <0 on error. */
int orc_scene_add_obj(orc_scene *, const char *path, const float *ctm);
/* TriangleMesh::createSingleTriangle + setV1..3/setN1..3 + addObject. */
int orc_scene_add_triangle(orc_scene *, const float v[9], const float n[9]);
/* Append an indexed mesh given as flat arrays (synthetic scenes). */
int orc_scene_add_arrays(orc_scene *, int nv, const float *v, int nn, const float *n,
                         int nt, const uint32_t *vi, const uint32_t *ni);

/* Sphere (Sphere.h:13-20) as the next bounded object; returns its object (prim) index. */
int orc_scene_add_sphere(orc_scene *, const float center[3], float radius);
/* Plane (Plane.h:22-23) as the next unbounded object; hits report prim = ORC_PLANE_BIT | index.  Returns index. */
int orc_scene_add_plane(orc_scene *, const float normal[3], const float origin[3], uint32_t material);
#define ORC_PLANE_BIT 0x80000000u

int orc_scene_counts(const orc_scene *, int *nv, int *nn, int *nt);
const float    *orc_scene_vertices(const orc_scene *);
const float    *orc_scene_normals(const orc_scene *);
const uint32_t *orc_scene_vidx(const orc_scene *);
const uint32_t *orc_scene_nidx(const orc_scene *);

/* BVH::build.  leaf_size 4 = scalar build, 8 = SSE build (BVH.h:59,61). */
int  orc_scene_build(orc_scene *, int leaf_size);
void orc_scene_tree_stats(const orc_scene *, int *nodes, int *leaves, int *max_depth);
/* Export the tree in DFS pre-order for structural comparison with the product builder:
 * per node 6 floats (padded corners), is_leaf, child0/child1 or first/count into leaf_prims. */
int  orc_scene_export_tree(const orc_scene *, float *corners6, int32_t *meta3, uint32_t *leaf_prims);

/* Scene::trace -> BVH::intersect, scalar path.  counters may be NULL. */
void orc_trace(const orc_scene *, const orc_ray *rays, uint64_t n, orc_hit *hits,
               orc_counters *counters);
/* Brute force over all triangles in prim order with the same predicate (sanity only). */
void orc_trace_brute(const orc_scene *, const orc_ray *rays, uint64_t n, orc_hit *hits);

/* SSE packet path restatement -- the timed CPU baseline, NOT a parity oracle.  Scene must
 * have been built with leaf_size 8.  threads<=0: all cores. Returns threads used. */
int  orc_trace_sse(const orc_scene *, const orc_ray *rays, uint64_t n, orc_hit *hits,
                   int threads, orc_counters *counters);

/* Camera::eyeRay for every pixel (row-major, y*W+x), spp samples per pixel.
 * spp==1 && !jitter: pixel centre (dx=dy=0.5) like eyeRay(...,false).
 * jitter: dx,dy from the counter RNG documented in DESIGN.md (reference uses rand()).
 * Rows [y0,y1) only; ray index = ((y-y0)*W + x)*spp + s. */
void orc_eye_rays(const orc_camera *, int W, int H, int y0, int y1, int spp, int jitter,
                  uint32_t seed, orc_ray *rays);

/* Phong::shade shadow ray for each hit (one point light).  Writes rays compacted in ray
 * order, src[i] = index of the primary ray.  Returns number of shadow rays. */
/* sse_order=1 evaluates the hit point as the SSE packet code does, A + (beta*BmA + gamma*CmA)
 * (BVH.cpp:402), instead of Triangle.cpp:160's (A + beta*BmA) + gamma*CmA. */
uint64_t orc_shadow_rays(const orc_scene *, const orc_ray *rays, const orc_hit *hits, uint64_t n,
                         const float light[3], orc_ray *out, uint64_t *src, int sse_order);

/* HitInfo reconstruction: P (Triangle.cpp:160), N un-normalised (Triangle.cpp:162). */
void orc_hit_attrs(const orc_scene *, const orc_hit *hits, uint64_t n, float *P, float *N);

/* same for scenes with spheres / planes, whose P = o + t*d needs the ray (Sphere.cpp:61-63, Plane.cpp:42-45) */
void orc_hit_attrs_rays(const orc_scene *, const orc_ray *rays, const orc_hit *hits, uint64_t n, float *P, float *N);

/* Phong::shade for one point light + per-pixel sample average; occluded[i] = shadow ray of primary ray i
 * hit something.  rgb: (n/spp)*3 linear floats.  (miro_oracle_shade.c) */
void orc_shade_direct(const orc_scene *, const orc_ray *rays, const orc_hit *hits, uint64_t n,
                      const unsigned char *occluded, const float light[3], const float color[3],
                      float wattage, const float diffuse[3], int spp, float *rgb);
void orc_tonemap(const float *rgb, uint64_t n_values, unsigned char *out);
/* Scene::traceScene (Scene.cpp:270-346) per ray for reflective / refractive Phong materials (11 floats each:
 * diffuse, specular, transmission, shininess, refraction index, already clamped as Phong.cpp:12-33); returns the
 * number of Scene::trace calls.  (miro_oracle_shade.c) */
/* PATH_TRACING ray generators (miro_oracle_path.c): children of every hit, in ray order; returns their number */
uint64_t orc_path_rays(const orc_scene *, const float *materials, const uint32_t *prim_mat, const orc_ray *rays,
                       const orc_hit *hits, const float *weights, const uint32_t *pixels, const uint32_t *ids, uint64_t n,
                       uint32_t spp, uint32_t seed, uint32_t bounce, uint32_t kinds, orc_ray *out, float *out_w,
                       uint32_t *out_pix, uint32_t *out_id, uint32_t *out_kind);
uint64_t orc_trace_scene_pt(const orc_scene *, const float *materials, const uint32_t *prim_mat, const orc_ray *rays,
                            uint64_t n, const float light[3], const float color[3], float wattage, int depth, uint32_t seed,
                            uint32_t kinds, float *rgb);
void orc_miro_math(const float *x, const float *y, uint64_t n, float *out);
uint64_t orc_trace_scene(const orc_scene *, const float *materials, const uint32_t *prim_mat, const orc_ray *rays,
                         uint64_t n, const float light[3], const float color[3], float wattage, int depth, float *rgb);

/* Photon map (miro_oracle_photon.c): Photon_map::store / scale_photon_power / balance / irradiance_estimate */
typedef struct orc_pmap orc_pmap;
orc_pmap *orc_pmap_new(int max_photons);
void orc_pmap_free(orc_pmap *);
int  orc_pmap_count(const orc_pmap *);
void orc_pmap_store(orc_pmap *, int n, const float *power, const float *pos, const float *dir);
void orc_pmap_scale(orc_pmap *, float scale);
void orc_pmap_balance(orc_pmap *);
void orc_pmap_irradiance(const orc_pmap *, uint64_t nq, const float *pos, const float *normal, float max_dist,
                         int nphotons, float *irrad, int *found, float *r2);
void orc_pmap_irradiance_brute(const orc_pmap *, uint64_t nq, const float *pos, const float *normal, float max_dist,
                               int nphotons, float *irrad, int *found, float *r2);
void orc_pmap_export(const orc_pmap *, float *pos, int *plane, unsigned char *theta_phi, float *power);

uint32_t orc_hash(uint32_t x);

#ifdef __cplusplus
}
#endif
#endif
