/*
 * miro_oracle_sse.c -- restatement of the reference's SSE4.1 packet path: 1 ray x 4 triangles,
 * 1 ray x 2 boxes, 8 triangles per leaf, _mm_rcp_ps reciprocals, OpenMP over ray chunks.
 * TEST INFRASTRUCTURE ONLY (see miro_oracle.h).
 *
 * This is the TIMED CPU BASELINE of bench.py ("cpu_baseline", kind "port"), not a parity
 * oracle: _mm_rcp_ps (Ray.h:58, BVH.cpp:366) moves t by ~2e-4 relative.
 *
 *   Ray::setupSSE                 Ray.h:51-60
 *   packet cache build            BVH.cpp:91-166
 *   SSEintersectTriangles         BVH.cpp:342-412, SSE.h:15-49
 *   intersectTriangleList         BVH.cpp:414-434
 *   root slab test (scalar)       BVH.cpp:447-466
 *   intersectChildren, SSE branch BVH.cpp:478-491, 513-584
 *   OpenMP schedule(dynamic,2) rows  Scene.cpp:112-115
 */
#include "miro_oracle_internal.h"

#include <math.h>
#include <smmintrin.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { __m128 v[3]; } tuple3;                       /* SSEVectorTuple3, SSE.h:9-12 */

typedef struct {
    tuple3 A, BmA, CmA, normal, nA, nB, nC;                   /* SSETriangleCache, BVH.h:16-22 */
    int ntri;
    uint32_t prim[4];
} packet;

typedef struct {
    packet *packets;
    int *first, *count;      /* per node: packet range (leaves only) */
    __m128 *corners;         /* 2 per node: (min.xyz,0), (max.xyz,0) */
} sse_cache;

typedef struct {
    __m128 d_neg, o, d_rcp;  /* d_SSE = -d, o_SSE, d_SSE_rcp (Ray.h:54-59) */
    const orc_ray *r;
} sse_ray;

static inline tuple3 multi_cross(const tuple3 *a, const tuple3 *b)   /* SSE.h:15-23 */
{
    tuple3 o;
    o.v[0] = _mm_sub_ps(_mm_mul_ps(a->v[1], b->v[2]), _mm_mul_ps(a->v[2], b->v[1]));
    o.v[1] = _mm_sub_ps(_mm_mul_ps(a->v[2], b->v[0]), _mm_mul_ps(a->v[0], b->v[2]));
    o.v[2] = _mm_sub_ps(_mm_mul_ps(a->v[0], b->v[1]), _mm_mul_ps(a->v[1], b->v[0]));
    return o;
}
static inline __m128 multi_dot13(__m128 a, const tuple3 *b)          /* SSE.h:41-44 */
{
    return _mm_add_ps(_mm_mul_ps(_mm_shuffle_ps(a, a, _MM_SHUFFLE(0,0,0,0)), b->v[0]),
           _mm_add_ps(_mm_mul_ps(_mm_shuffle_ps(a, a, _MM_SHUFFLE(1,1,1,1)), b->v[1]),
                      _mm_mul_ps(_mm_shuffle_ps(a, a, _MM_SHUFFLE(2,2,2,2)), b->v[2])));
}
static inline __m128 multi_dot33(const tuple3 *a, const tuple3 *b)   /* SSE.h:46-49 */
{
    return _mm_add_ps(_mm_mul_ps(a->v[0], b->v[0]),
           _mm_add_ps(_mm_mul_ps(a->v[1], b->v[1]), _mm_mul_ps(a->v[2], b->v[2])));
}

void orc_sse_free(orc_scene *s)
{
    sse_cache *c = (sse_cache *)s->sse;
    if (!c) return;
    free(c->packets); free(c->first); free(c->count); free(c->corners);
    free(c);
    s->sse = NULL;
}

/* leaf triangle regrouping, BVH.cpp:91-166 */
void orc_sse_prepare(orc_scene *s)
{
    sse_cache *c = (sse_cache *)calloc(1, sizeof(sse_cache));
    int npk = 0;
    for (int i = 0; i < s->n_nodes; i++)
        if (s->nodes[i].leaf) npk += (s->nodes[i].b + 3) / 4;
    c->packets = (packet *)aligned_alloc(16, sizeof(packet) * (size_t)(npk + 1));
    memset(c->packets, 0, sizeof(packet) * (size_t)(npk + 1));
    c->first = (int *)calloc((size_t)s->n_nodes + 1, sizeof(int));
    c->count = (int *)calloc((size_t)s->n_nodes + 1, sizeof(int));
    c->corners = (__m128 *)aligned_alloc(16, sizeof(__m128) * 2 * (size_t)(s->n_nodes + 1));
    const v3 *V = (const v3 *)s->v, *N = (const v3 *)s->n;
    int pk = 0;
    for (int i = 0; i < s->n_nodes; i++) {
        const orc_node *nd = &s->nodes[i];
        c->corners[2*i]   = _mm_set_ps(0.0f, nd->c[0][2], nd->c[0][1], nd->c[0][0]);
        c->corners[2*i+1] = _mm_set_ps(0.0f, nd->c[1][2], nd->c[1][1], nd->c[1][0]);
        if (!nd->leaf) continue;
        c->first[i] = pk;
        for (int base = 0; base < nd->b; base += 4) {
            packet *p = &c->packets[pk++];
            float verts[3][12], normals[3][12];
            memset(verts, 0, sizeof(verts)); memset(normals, 0, sizeof(normals));
            p->ntri = nd->b - base < 4 ? nd->b - base : 4;
            for (int t = 0; t < p->ntri; t++) {
                uint32_t prim = s->leaf_prims[nd->a + base + t];
                p->prim[t] = prim;
                for (int k = 0; k < 3; k++) {
                    const float *vv = (const float *)&V[s->vi[3*prim + k]];
                    const float *nn = (const float *)&N[s->ni[3*prim + k]];
                    for (int d = 0; d < 3; d++) { verts[d][k*4+t] = vv[d]; normals[d][k*4+t] = nn[d]; }
                }
            }
            for (int d = 0; d < 3; d++) {
                p->A.v[d]   = _mm_loadu_ps(&verts[d][0]);
                p->nA.v[d]  = _mm_loadu_ps(&normals[d][0]);
                p->nB.v[d]  = _mm_loadu_ps(&normals[d][4]);
                p->nC.v[d]  = _mm_loadu_ps(&normals[d][8]);
                p->BmA.v[d] = _mm_sub_ps(_mm_loadu_ps(&verts[d][4]), p->A.v[d]);
                p->CmA.v[d] = _mm_sub_ps(_mm_loadu_ps(&verts[d][8]), p->A.v[d]);
            }
            p->normal = multi_cross(&p->BmA, &p->CmA);
        }
        c->count[i] = pk - c->first[i];
    }
    s->sse = c;
}

/* SSEintersectTriangles, BVH.cpp:342-412.  Returns winning lane or -1. */
static inline int packet_test(const packet *c, const sse_ray *ray, float tMin, float tMax,
                              float *outT, float *outBeta, float *outGamma)
{
    const float eps = 1e-4f;
    const __m128 one = _mm_set1_ps(1.0f);
    tuple3 RomA;
    for (int i = 0; i < 3; i++) {
        __m128 oi = i == 0 ? _mm_shuffle_ps(ray->o, ray->o, _MM_SHUFFLE(0,0,0,0))
                  : i == 1 ? _mm_shuffle_ps(ray->o, ray->o, _MM_SHUFFLE(1,1,1,1))
                           : _mm_shuffle_ps(ray->o, ray->o, _MM_SHUFFLE(2,2,2,2));
        RomA.v[i] = _mm_sub_ps(oi, c->A.v[i]);
    }
    __m128 ddotn = _mm_rcp_ps(multi_dot13(ray->d_neg, &c->normal));                 /* :366 */
    __m128 t = _mm_mul_ps(multi_dot33(&RomA, &c->normal), ddotn);
    tuple3 x1 = multi_cross(&RomA, &c->CmA), x2 = multi_cross(&c->BmA, &RomA);
    __m128 beta = _mm_mul_ps(multi_dot13(ray->d_neg, &x1), ddotn);
    __m128 gamma = _mm_mul_ps(multi_dot13(ray->d_neg, &x2), ddotn);
    int mask = _mm_movemask_ps(_mm_and_ps(_mm_cmpgt_ps(beta, _mm_set1_ps(-eps)),
                   _mm_and_ps(_mm_cmpgt_ps(gamma, _mm_set1_ps(-eps)),
                   _mm_and_ps(_mm_cmplt_ps(_mm_add_ps(gamma, beta), _mm_set1_ps(1.0f + eps)),
                   _mm_and_ps(_mm_cmpgt_ps(t, _mm_set1_ps(tMin)), _mm_cmplt_ps(t, _mm_set1_ps(tMax)))))));
    if (mask == 0) return -1;
    float tt[4];
    _mm_storeu_ps(tt, t);
    int best = -1;
    for (int i = 0; i < c->ntri; i++) {
        if ((mask & (1 << i)) == 0) continue;
        if (best == -1 || tt[i] < tt[best]) best = i;
    }
    if (best == -1) return -1;
    /* the reference computes P and N for all four lanes here (:396-404); keep the cost */
    __m128 alpha = _mm_sub_ps(_mm_sub_ps(one, beta), gamma);
    float sink[3][4];
    for (int i = 0; i < 3; i++) {
        __m128 P = _mm_add_ps(c->A.v[i], _mm_add_ps(_mm_mul_ps(beta, c->BmA.v[i]), _mm_mul_ps(gamma, c->CmA.v[i])));
        __m128 Nn = _mm_add_ps(_mm_mul_ps(alpha, c->nA.v[i]), _mm_add_ps(_mm_mul_ps(beta, c->nB.v[i]), _mm_mul_ps(gamma, c->nC.v[i])));
        _mm_storeu_ps(sink[i], _mm_add_ps(P, Nn));
    }
    float bb[4], gg[4];
    _mm_storeu_ps(bb, beta); _mm_storeu_ps(gg, gamma);
    *outT = tt[best] + 0.0f * sink[0][best];
    *outBeta = bb[best]; *outGamma = gg[best];
    return best;
}

/* BVH::intersectChildren, __SSE4_1__ branch (BVH.cpp:471-491, 513-584) */
static int isect_children_sse(const orc_scene *s, const sse_cache *c, int node, const sse_ray *ray,
                              float tMin, float tMax, orc_hit *minHit, uint64_t *tri_tests)
{
    int hit = 0;
    orc_hit tmp;
    const orc_node *nd = &s->nodes[node];
    minHit->t = tMax;
    if (nd->leaf) {
        for (int i = 0; i < c->count[node]; i++) {
            const packet *p = &c->packets[c->first[node] + i];
            float t, b, g;
            int best = packet_test(p, ray, tMin, minHit->t, &t, &b, &g);     /* :418 */
            if (best != -1 && t < minHit->t) {                               /* :421 */
                minHit->t = t; minHit->beta = b; minHit->gamma = g; minHit->prim = p->prim[best];
            }
            if (best != -1) hit = 1;                                         /* :431,:485 */
            *tri_tests += (uint64_t)p->ntri;
        }
        return hit;
    }
    int ch[2] = {nd->a, nd->b};
    __m128 t0 = _mm_mul_ps(_mm_sub_ps(c->corners[2*ch[0]],   ray->o), ray->d_rcp);
    __m128 t1 = _mm_mul_ps(_mm_sub_ps(c->corners[2*ch[0]+1], ray->o), ray->d_rcp);
    __m128 t2 = _mm_mul_ps(_mm_sub_ps(c->corners[2*ch[1]],   ray->o), ray->d_rcp);
    __m128 t3 = _mm_mul_ps(_mm_sub_ps(c->corners[2*ch[1]+1], ray->o), ray->d_rcp);
    __m128 tmin[2] = {_mm_min_ps(t0, t1), _mm_min_ps(t2, t3)};
    __m128 tmax[2] = {_mm_max_ps(t0, t1), _mm_max_ps(t2, t3)};
    /* horizontal max-of-mins / min-of-maxes over x,y,z (inline asm :528-547) */
    float out[4];
    for (int k = 0; k < 2; k++) {
        __m128 mn = tmin[k], mx = tmax[k];
        __m128 a = _mm_max_ss(mn, _mm_shuffle_ps(mn, mn, _MM_SHUFFLE(1,1,1,1)));
        a = _mm_max_ss(a, _mm_shuffle_ps(mn, mn, _MM_SHUFFLE(2,2,2,2)));
        __m128 b = _mm_min_ss(mx, _mm_shuffle_ps(mx, mx, _MM_SHUFFLE(1,1,1,1)));
        b = _mm_min_ss(b, _mm_shuffle_ps(mx, mx, _MM_SHUFFLE(2,2,2,2)));
        out[3 - 2*k] = _mm_cvtss_f32(a);     /* 3: minOverlap[0], 1: minOverlap[1] */
        out[2 - 2*k] = _mm_cvtss_f32(b);     /* 2: maxOverlap[0], 0: maxOverlap[1] */
    }
    int ind = 0;
    if (out[3] < out[1]) ind = 2;                                             /* :561-563 */
    for (int pass = 0; pass < 2; pass++) {
        if (!(out[ind+1] > out[ind] || out[ind+1] > minHit->t || out[ind] < tMin)) {   /* :565,:577 */
            if (isect_children_sse(s, c, ch[1 - (ind >> 1)], ray, tMin, minHit->t, &tmp, tri_tests)) {
                /* only t/prim of an actual improvement are meaningful; the reference copies tmp */
                if (tmp.t < minHit->t) *minHit = tmp;
                hit = 1;
            }
        }
        ind = (ind + 2) & 3;
    }
    return hit;
}

int orc_trace_sse(const orc_scene *s, const orc_ray *rays, uint64_t n, orc_hit *hits,
                  int threads, orc_counters *counters)
{
    const sse_cache *c = (const sse_cache *)s->sse;
    if (!c) return -1;
    uint64_t box = 0, tri = 0;
    int used = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    used = threads;
#else
    (void)threads;
#endif
    const int64_t nchunks = (int64_t)((n + 1023) / 1024);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+:box, tri)
#endif
    for (int64_t ck = 0; ck < nchunks; ck++) {
        uint64_t lo = (uint64_t)ck * 1024, hi = lo + 1024 < n ? lo + 1024 : n;
        for (uint64_t i = lo; i < hi; i++) {
            const orc_ray *r = &rays[i];
            sse_ray ray;
            ray.r = r;
            __m128 d = _mm_set_ps(r->dx, r->dz, r->dy, r->dx);
            ray.d_neg = _mm_sub_ps(_mm_setzero_ps(), d);
            ray.d_rcp = _mm_rcp_ps(d);
            ray.o = _mm_set_ps(r->ox, r->oz, r->oy, r->ox);
            orc_hit h;
            h.t = r->tmax; h.prim = ORC_MISS; h.beta = 0; h.gamma = 0;
            /* root test stays scalar with true divides even in the SSE build (BVH.cpp:447-466) */
            float mn = -INFINITY, mx = INFINITY;
            const float *o = &r->ox, *dd = &r->dx;
            for (int k = 0; k < 3; k++) {
                float t[2];
                t[0] = (s->nodes[0].c[0][k] - o[k]) / dd[k];
                t[1] = (s->nodes[0].c[1][k] - o[k]) / dd[k];
                int m = t[0] > t[1];
                if (t[m] > mn) mn = t[m];
                if (t[m ^ 1] < mx) mx = t[m ^ 1];
            }
            box++;
            if (!(mn > mx || mn > r->tmax || mx < r->tmin))
                isect_children_sse(s, c, 0, &ray, r->tmin, r->tmax, &h, &tri);
            if (h.prim == ORC_MISS) { h.t = r->tmax; h.beta = 0; h.gamma = 0; }
            hits[i] = h;
        }
    }
    if (counters) { counters->box_tests += box; counters->tri_tests += tri; }
    return used;
}
