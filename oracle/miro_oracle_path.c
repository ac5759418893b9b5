/*
 * miro_oracle_path.c -- restatement of the PATH_TRACING build of the secondary-ray generators:
 *   Ray::random                       Ray.h:124-140   (phi = asin(sqrt(frand())), theta = 2 PI frand(), around the normal)
 *   Ray::reflect  #ifdef PATH_TRACING Ray.h:149-158   (phi = acos(pow(frand(), 1/(1+shininess))), around d - 2(N.d)N)
 *   Ray::refract  #ifdef PATH_TRACING Ray.h:235-239   (same lobe around the refracted direction)
 *   Ray::alignToVector                Ray.h:86-91;  alignHemisphereToVector  Utility.h:34-50
 *   which children a hit spawns and their weights: Scene::traceScene, Scene.cpp:302-336
 * The reference draws from rand(); here frand() is the counter-based generator of the eye-ray jitter (orc_hash),
 * keyed by (seed, ray id, bounce, child kind) -- integer arithmetic, identical on the device.  sin / cos / asin / acos /
 * pow are include/miro_math.h (the one piece of source shared with the product, see its header): libm's last bit differs
 * between glibc and the device library, and the ray sets are compared bit for bit.
 * The diffuse child (kind 3, Ray::random) is an EXTENSION: traceScene at HEAD never calls Ray::random.
 * TEST INFRASTRUCTURE ONLY (see miro_oracle.h).
 */
#include "miro_oracle_internal.h"

#include <math.h>

#include "miro_math.h"

static inline float frand_of(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }
static inline int positive(const float *v) { return v[0] > 0.f || v[1] > 0.f || v[2] > 0.f; }
static inline v3 over(v3 a, float s) { float inv = 1.0f / s; return v3scale(a, inv); }   /* Vector3::operator/= */

/* alignHemisphereToVector (Utility.h:34-50) followed by Ray(origin + epsilon*dir, dir) (Ray.h:86-91) */
static void align_to_vector(v3 v, v3 origin, float theta, float phi, v3 *o, v3 *d)
{
    const float PI = 3.1415926535897932384626433832795028841972f;
    (void)PI;
    float u1 = mm_sinf(phi) * mm_cosf(theta);
    float u2 = mm_sinf(phi) * mm_sinf(theta);
    float u3 = mm_cosf(phi);
    v3 ez = {0, 0, 1}, ey = {0, 1, 0};
    v3 t1 = v3cross(ez, v);
    if (v3dot(t1, t1) < 1e-6) t1 = v3cross(ey, v);
    v3 aligned = v3add(v3add(v3scale(t1, u1), v3scale(v3cross(t1, v), u2)), v3scale(v, u3));
    aligned = over(aligned, sqrtf(v3dot(aligned, aligned)));
    *d = aligned;
    *o = v3add(origin, v3scale(aligned, 1e-4f));
}

static void lobe_angles(uint32_t hk, float shininess, float *theta, float *phi)
{
    const float PI = 3.1415926535897932384626433832795028841972f;
    *phi = mm_acosf01(mm_powf01(frand_of(orc_hash(hk)), 1.0f / (1.0f + shininess)));
    *theta = 2.0f * PI * frand_of(orc_hash(hk ^ 0x68bc21ebu));
}

/* Ray::reflect, PATH_TRACING branch */
static void reflect_pt(v3 d, v3 P, v3 N, uint32_t hk, float shininess, v3 *o, v3 *dr)
{
    float theta, phi;
    lobe_angles(hk, shininess, &theta, &phi);
    v3 d_reflect = v3sub(d, v3scale(N, 2 * v3dot(N, d)));
    align_to_vector(d_reflect, P, theta, phi, o, dr);
}

/* For n rays with their hits: the children of Scene::traceScene's recursion under PATH_TRACING, in ray order, kinds in the
 * order 0 mirror, 1 Fresnel reflection, 2 refraction, 3 diffuse bounce.  out arrays need room for 4n entries; returns
 * the number of children.  ids == NULL: ray index. */
uint64_t orc_path_rays(const orc_scene *s, const float *materials, const uint32_t *prim_mat, const orc_ray *rays,
                       const orc_hit *hits, const float *weights, const uint32_t *pixels, const uint32_t *ids, uint64_t n,
                       uint32_t spp, uint32_t seed, uint32_t bounce, uint32_t kinds, orc_ray *out, float *out_w,
                       uint32_t *out_pix, uint32_t *out_id, uint32_t *out_kind)
{
    const float PI = 3.1415926535897932384626433832795028841972f;
    const uint32_t base = orc_hash(seed);
    uint64_t m = 0;
    for (uint64_t k = 0; k < n; k++) {
        const orc_hit *h = &hits[k];
        if (h->prim == ORC_MISS) continue;
        const float *mt = (h->prim & ORC_PLANE_BIT) ? materials + 11 * (size_t)s->plane_mat[h->prim & ~ORC_PLANE_BIT]
                                                     : materials + 11 * (size_t)(prim_mat ? prim_mat[h->prim] : 0);
        const int refl = positive(mt + 3) && (kinds & 1u), refr = positive(mt + 6) && (kinds & 2u), diff = positive(mt) && (kinds & 4u);
        if (!refl && !refr && !diff) continue;
        v3 P, N;
        orc_surface(s, &rays[k], h, &P, &N, 0);
        N = over(N, sqrtf(v3dot(N, N)));                                     /* Scene.cpp:262 */
        const v3 d = {rays[k].dx, rays[k].dy, rays[k].dz};
        const float w0[3] = {weights ? weights[3 * k] : 1.f, weights ? weights[3 * k + 1] : 1.f, weights ? weights[3 * k + 2] : 1.f};
        const uint32_t pix = pixels ? pixels[k] : (uint32_t)(k / spp);
        const uint32_t id = ids ? ids[k] : (uint32_t)k;
        const uint32_t hray = orc_hash(base ^ id) + bounce * 4u;
        v3 co[4], cd[4];
        float cw[4][3];
        int emit[4] = {0, 0, 0, 0};
        if (refl) {                                                           /* Scene.cpp:302-312 */
            emit[0] = 1;
            reflect_pt(d, P, N, orc_hash(hray + 0u), mt[9], &co[0], &cd[0]);
            for (int c = 0; c < 3; c++) cw[0][c] = w0[c] * mt[3 + c];
        }
        if (refr) {                                                           /* Scene.cpp:315-336 */
            float n1, n2; v3 nn;
            if (v3dot(d, N) < 0) { n1 = 1.0f; n2 = mt[10]; nn = N; }
            else { n1 = mt[10]; n2 = 1.0f; nn.x = -N.x; nn.y = -N.y; nn.z = -N.z; }
            /* Ray::getReflectionCoefficient (Ray.h:168-199) */
            v3 md = {-d.x, -d.y, -d.z};
            float cosTheta = v3dot(md, nn);
            float sinTheta = mm_sinf(mm_acosf(cosTheta));
            float q = (n1 / n2) * sinTheta, p = q * q;
            float Rs = 1.0f;
            if (!(p > 1.f)) {
                float sq = sqrtf(1.f - p), fr = (n1 * cosTheta - sq) / (n1 * cosTheta + sq);
                Rs = fr * fr;
            }
            if (Rs > 0.01) {
                emit[1] = 1;
                reflect_pt(d, P, N, orc_hash(hray + 1u), mt[9], &co[1], &cd[1]);
                for (int c = 0; c < 3; c++) cw[1][c] = w0[c] * mt[6 + c] * Rs;
            }
            /* Ray::refract (Ray.h:202-243) */
            float dn = v3dot(d, nn);
            float energy = (float)(1 - (pow(n1, 2) * (1 - pow(dn, 2)) / pow(n2, 2)));
            emit[2] = 1;
            if (energy < 0) {
                reflect_pt(d, P, N, orc_hash(hray + 2u), mt[9], &co[2], &cd[2]);
            } else {
                v3 t = over(v3scale(v3sub(d, v3scale(nn, dn)), n1), n2);
                v3 d_r = v3sub(t, v3scale(nn, sqrtf(energy)));
                float theta, phi;
                lobe_angles(orc_hash(hray + 2u), mt[9], &theta, &phi);
                align_to_vector(d_r, P, theta, phi, &co[2], &cd[2]);
            }
            for (int c = 0; c < 3; c++) cw[2][c] = w0[c] * mt[6 + c] * (1.f - Rs);
        }
        if (diff) {                                                           /* Ray::random, Ray.h:124-140 */
            emit[3] = 1;
            const uint32_t hk = orc_hash(hray + 3u);
            float phi = mm_asinf01(sqrtf(frand_of(orc_hash(hk))));
            float theta = 2.0f * PI * frand_of(orc_hash(hk ^ 0x68bc21ebu));
            align_to_vector(N, P, theta, phi, &co[3], &cd[3]);
            for (int c = 0; c < 3; c++) cw[3][c] = w0[c] * mt[c];
        }
        for (int j = 0; j < 4; j++) {
            if (!emit[j]) continue;
            orc_ray *r = &out[m];
            r->ox = co[j].x; r->oy = co[j].y; r->oz = co[j].z; r->tmin = 0.0f;
            r->dx = cd[j].x; r->dy = cd[j].y; r->dz = cd[j].z; r->tmax = 1e12f;
            out_w[3 * m] = cw[j][0]; out_w[3 * m + 1] = cw[j][1]; out_w[3 * m + 2] = cw[j][2];
            out_pix[m] = pix;
            if (out_id) out_id[m] = orc_hash(id ^ (0x9e3779b9u * (uint32_t)(j + 1)));
            if (out_kind) out_kind[m] = (uint32_t)j;
            m++;
        }
    }
    return m;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Scene::traceScene (Scene.cpp:270-346) as the PATH_TRACING build runs it: the same recursion as orc_trace_scene, with the
 * lobe-sampled generators above in place of the mirror / refraction directions.  `id` is the ray's stable id (a primary
 * ray's index), `bounce` its recursion level; a child's id is orc_hash(id ^ 0x9e3779b9 * (kind + 1)) -- the keys the device
 * hands down its wavefront levels.  kinds bit 2 adds the diffuse bounce (extension, weight kd).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct { ts_ctx t; uint32_t base, kinds; } pt_ctx;

static int trace_scene_pt(pt_ctx *c, v3 o, v3 d, int depth, uint32_t id, uint32_t bounce, float res[3])
{
    const float PI = 3.1415926535897932384626433832795028841972f;
    res[0] = res[1] = res[2] = 0.0f;
    if (depth < 0) return 0;
    orc_hit h; v3 P, N;
    if (!orc_ts_trace(&c->t, o, d, &h, &P, &N)) return 1;
    --depth;
    const float *m = (h.prim & ORC_PLANE_BIT) ? c->t.mats + 11 * (size_t)c->t.s->plane_mat[h.prim & ~ORC_PLANE_BIT]
                                               : c->t.mats + 11 * (size_t)c->t.prim_mat[h.prim];
    orc_ts_shade(&c->t, d, h.prim, P, N, res);
    const uint32_t hray = orc_hash(c->base ^ id) + bounce * 4u;
    v3 ro, rd; float sub[3];
    if (positive(m + 3) && (c->kinds & 1u)) {
        reflect_pt(d, P, N, orc_hash(hray + 0u), m[9], &ro, &rd);
        if (trace_scene_pt(c, ro, rd, depth, orc_hash(id ^ (0x9e3779b9u * 1u)), bounce + 1, sub))
            for (int k = 0; k < 3; k++) res[k] += m[3 + k] * sub[k];
    }
    if (positive(m + 6) && (c->kinds & 2u)) {
        float n1, n2; v3 nn;
        if (v3dot(d, N) < 0) { n1 = 1.0f; n2 = m[10]; nn = N; }
        else { n1 = m[10]; n2 = 1.0f; nn.x = -N.x; nn.y = -N.y; nn.z = -N.z; }
        v3 md = {-d.x, -d.y, -d.z};
        float cosTheta = v3dot(md, nn);
        float sinTheta = mm_sinf(mm_acosf(cosTheta));
        float q = (n1 / n2) * sinTheta, p = q * q;
        float Rs = 1.0f;
        if (!(p > 1.f)) {
            float sq = sqrtf(1.f - p), fr = (n1 * cosTheta - sq) / (n1 * cosTheta + sq);
            Rs = fr * fr;
        }
        if (Rs > 0.01) {
            reflect_pt(d, P, N, orc_hash(hray + 1u), m[9], &ro, &rd);
            if (trace_scene_pt(c, ro, rd, depth, orc_hash(id ^ (0x9e3779b9u * 2u)), bounce + 1, sub))
                for (int k = 0; k < 3; k++) res[k] += m[6 + k] * sub[k] * Rs;
        }
        float dn = v3dot(d, nn);
        float energy = (float)(1 - (pow(n1, 2) * (1 - pow(dn, 2)) / pow(n2, 2)));
        if (energy < 0) {
            reflect_pt(d, P, N, orc_hash(hray + 2u), m[9], &ro, &rd);
        } else {
            v3 t = over(v3scale(v3sub(d, v3scale(nn, dn)), n1), n2);
            v3 d_r = v3sub(t, v3scale(nn, sqrtf(energy)));
            float theta, phi;
            lobe_angles(orc_hash(hray + 2u), m[9], &theta, &phi);
            align_to_vector(d_r, P, theta, phi, &ro, &rd);
        }
        if (trace_scene_pt(c, ro, rd, depth, orc_hash(id ^ (0x9e3779b9u * 3u)), bounce + 1, sub))
            for (int k = 0; k < 3; k++) res[k] += m[6 + k] * sub[k] * (1.f - Rs);
    }
    if (positive(m) && (c->kinds & 4u)) {
        const uint32_t hk = orc_hash(hray + 3u);
        float phi = mm_asinf01(sqrtf(frand_of(orc_hash(hk))));
        float theta = 2.0f * PI * frand_of(orc_hash(hk ^ 0x68bc21ebu));
        align_to_vector(N, P, theta, phi, &ro, &rd);
        if (trace_scene_pt(c, ro, rd, depth, orc_hash(id ^ (0x9e3779b9u * 4u)), bounce + 1, sub))
            for (int k = 0; k < 3; k++) res[k] += m[k] * sub[k];
    }
    return 1;
}

/* per-ray colours; ray i has id i; returns the number of Scene::trace calls made */
uint64_t orc_trace_scene_pt(const orc_scene *s, const float *materials, const uint32_t *prim_mat, const orc_ray *rays,
                            uint64_t n, const float light[3], const float color[3], float wattage, int depth, uint32_t seed,
                            uint32_t kinds, float *rgb)
{
    pt_ctx c;
    c.t.s = s; c.t.mats = materials; c.t.prim_mat = prim_mat;
    c.t.L.x = light[0]; c.t.L.y = light[1]; c.t.L.z = light[2];
    c.t.color.x = color[0]; c.t.color.y = color[1]; c.t.color.z = color[2];
    c.t.wattage = wattage; c.t.rays_traced = 0;
    c.base = orc_hash(seed); c.kinds = kinds;
    for (uint64_t i = 0; i < n; i++) {
        v3 o = {rays[i].ox, rays[i].oy, rays[i].oz}, d = {rays[i].dx, rays[i].dy, rays[i].dz};
        trace_scene_pt(&c, o, d, depth, (uint32_t)i, 0u, rgb + 3 * i);
    }
    return c.t.rays_traced;
}

/* miro_math.h's float functions, for tests/test_path_rays.py: out[5*i..] = sin, cos, asin01, acos01, pow01(x, y) */
void orc_miro_math(const float *x, const float *y, uint64_t n, float *out)
{
    for (uint64_t i = 0; i < n; i++) {
        out[5 * i] = mm_sinf(x[i]); out[5 * i + 1] = mm_cosf(x[i]);
        out[5 * i + 2] = mm_asinf01(x[i]); out[5 * i + 3] = mm_acosf01(x[i]);
        out[5 * i + 4] = mm_powf01(x[i], y[i]);
    }
}
