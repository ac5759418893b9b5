/* miro_oracle_internal.h -- shared between the scalar and SSE halves of the oracle.
 * TEST INFRASTRUCTURE ONLY (see miro_oracle.h). */
#ifndef MIRO_ORACLE_INTERNAL_H
#define MIRO_ORACLE_INTERNAL_H

#include <stddef.h>
#include "miro_oracle.h"

typedef struct { float x, y, z; } v3;

/* Vector3.h:83-109,238-255 -- one rounded fp32 op per arithmetic operator */
static inline v3 v3sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline v3 v3add(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline v3 v3scale(v3 a, float s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }
static inline float v3dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3cross(v3 a, v3 b)
{
    v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}

/* one BVH node (BVH.h:29-63): padded corners, leaf flag, children or [first,count) */
typedef struct {
    float c[2][3];
    int leaf;
    int a, b;
    int depth;
} orc_node;

struct orc_scene {
    int nv, nn, nt;
    float *v, *n;            /* vertices / normals, xyz triples */
    uint32_t *vi, *ni;       /* per-triangle vertex / normal indices */
    v3 *omin, *omax, *ocen;  /* per-object bounds and centre */
    orc_node *nodes;
    int n_nodes, cap_nodes, n_leaves, max_depth, leaf_size;
    uint32_t *leaf_prims;
    int n_leaf_prims;
    void *sse;               /* packet cache of the SSE path */
    /* non-triangle objects.  A sphere is a bounded object like any triangle (Scene.h:20-25): it takes the next
     * object index, and its slot in vi/ni holds {ORC_SPHERE_SLOT, sphere index, 0}.  Planes are unbounded: they
     * live in Scene::m_unboundedObjects and are scanned after the BVH (Scene.cpp:220-230). */
    int nspheres, nplanes;
    float *spheres;          /* cx, cy, cz, radius */
    float *planes;           /* normal xyz, origin xyz */
    uint32_t *plane_mat;     /* material id of each plane (for orc_trace_scene) */
};

#define ORC_SPHERE_SLOT 0xFFFFFFFFu
static inline int orc_is_sphere(const orc_scene *s, uint32_t prim) { return s->nspheres && s->vi[3 * (size_t)prim] == ORC_SPHERE_SLOT; }

/* the object test of a leaf (Object::intersect): Triangle::intersect or Sphere::intersect */
int  orc_obj_test(const orc_scene *s, uint32_t prim, const orc_ray *r, float tMin, float tMax, orc_hit *out);
/* HitInfo::P and HitInfo::N exactly as the object's intersect() leaves them (N of a triangle un-normalised,
 * N of a sphere normalised, N of a plane as set); ray may be NULL for triangle hits */
void orc_surface(const orc_scene *s, const orc_ray *ray, const orc_hit *h, v3 *P, v3 *N, int sse_order);

/* context of the restated Scene::traceScene recursions (miro_oracle_shade.c, miro_oracle_path.c) */
typedef struct {
    const orc_scene *s;
    const float *mats;
    const uint32_t *prim_mat;
    v3 L, color;
    float wattage;
    uint64_t rays_traced;
} ts_ctx;
int  orc_ts_trace(ts_ctx *c, v3 o, v3 d, orc_hit *h, v3 *P, v3 *N);       /* Scene::trace + normalised N */
void orc_ts_shade(ts_ctx *c, v3 d, uint32_t prim, v3 P, v3 N, float out[3]);   /* Phong::shade */

int  orc_tri_test(const orc_scene *s, uint32_t prim, const orc_ray *r, float tMin, float tMax, orc_hit *out);
void orc_sse_prepare(orc_scene *s);
void orc_sse_free(orc_scene *s);

#endif
