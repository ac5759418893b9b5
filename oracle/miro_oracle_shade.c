/*
 * miro_oracle_shade.c -- restatement of the caller of the shadow batch: Phong::shade for one point
 * light (Phong.cpp:44-160), Scene::trace's normal normalisation (Scene.cpp:238-263, bump height 0),
 * the miss colour (Scene.cpp:340, :683-686 with m_environment == 0, m_bgColor = 0), the per-pixel sample
 * average (Scene.cpp:126-139) and the tone map (Scene.cpp:87-91,177-202; Image.cpp:44-50).
 * TEST INFRASTRUCTURE ONLY (see miro_oracle.h).
 */
#include "miro_oracle_internal.h"

#include <math.h>
#include <stdlib.h>

static inline v3 divs(v3 a, float s) { float inv = 1.0f / s; return v3scale(a, inv); }

/* occluded[i] != 0: the shadow ray of primary ray i hit an (opaque) occluder (Phong.cpp:97-100) */
void orc_shade_direct(const orc_scene *s, const orc_ray *rays, const orc_hit *hits, uint64_t n,
                      const unsigned char *occluded, const float light[3], const float color[3],
                      float wattage, const float diffuse[3], int spp, float *rgb)
{
    const float PI = 3.1415926535897932384626433832795028841972f;
    v3 L = {light[0], light[1], light[2]};
    uint64_t npix = n / (uint64_t)spp;
    for (uint64_t pix = 0; pix < npix; pix++) {
        float acc[3] = {0, 0, 0};
        for (int sm = 0; sm < spp; sm++) {
            uint64_t k = pix * (uint64_t)spp + (uint64_t)sm;
            float c[3] = {0, 0, 0};                                 /* miss: m_bgColor = 0 */
            if (hits[k].prim != ORC_MISS && !occluded[k]) {
                v3 P, N;
                orc_surface(s, &rays[k], &hits[k], &P, &N, 0);
                N = divs(N, sqrtf(v3dot(N, N)));                    /* Scene.cpp:262 */
                v3 l = v3sub(L, P);
                float falloff = v3dot(l, l);
                l = divs(l, sqrtf(falloff));
                float nDotL = v3dot(N, l);
                float f2 = 1.0f / (falloff * 4.0f * PI * PI);       /* Phong.cpp:140 */
                float diff = nDotL * f2 * wattage / 1.0f;
                if (!(diff > 0.0f)) diff = 0.0f;                    /* std::max(0.0f, x) */
                for (int ch = 0; ch < 3; ch++) c[ch] = color[ch] * (diff * diffuse[ch] * diffuse[ch]) * 1.0f;
                /* highlight, Phong.cpp:149-156 (pow(float,int) promotes to double in C++11) */
                v3 md = {-rays[k].dx, -rays[k].dy, -rays[k].dz};
                v3 ml = {-l.x, -l.y, -l.z};
                v3 r = v3add(ml, v3scale(N, 2 * v3dot(l, N)));
                float e = v3dot(md, r);
                if (e > 1.f) e = 1.f;
                if (!(e > 0.0f)) e = 0.0f;
                float eDotr = (float)pow((double)e, 500.0);
                float hl = eDotr * f2 * wattage / 1.0f;
                if (!(hl > 0.0f)) hl = 0.0f;
                c[0] += hl; c[1] += hl; c[2] += hl;
            }
            acc[0] += c[0]; acc[1] += c[1]; acc[2] += c[2];
        }
        if (spp > 1) { float inv = 1.0f / (float)spp; acc[0] *= inv; acc[1] *= inv; acc[2] *= inv; }
        rgb[3*pix] = acc[0]; rgb[3*pix+1] = acc[1]; rgb[3*pix+2] = acc[2];
    }
}

/* tonemapValue = sigmoid(6v-3) (Scene.cpp:87-91, Utility.h:19-22: 1/(1+exp(-x)) in double), Map (Image.cpp:44-50) */
void orc_tonemap(const float *rgb, uint64_t n_values, unsigned char *out)
{
    for (uint64_t i = 0; i < n_values; i++) {
        float x = 6 * rgb[i] - 3;
        float v = (float)(1 / (1 + exp((double)-x)));
        float m = 255 * v;
        out[i] = m > 255 ? 255 : (unsigned char)m;
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * Scene::traceScene (Scene.cpp:270-346) for specular materials: the recursion over reflect / Fresnel / refract
 * rays (Ray.h:143-243) with Phong::shade (Phong.cpp:44-160, including the attenuation of light by refractive
 * occluders :97-113) at every hit.  Photon gathers contribute 0 (empty maps); misses return m_bgColor = 0.
 * materials: 11 floats each = diffuse[3], specular[3], transmission[3], shininess, refract_index, already
 * clamped as the Phong constructor does (Phong.cpp:12-33).  prim_mat: material id per triangle.
 * ------------------------------------------------------------------------------------------------------------ */
/* ts_ctx: miro_oracle_internal.h */

static inline const float *mat_of(const ts_ctx *c, uint32_t prim)
{
    if (prim & ORC_PLANE_BIT) return c->mats + 11 * (size_t)c->s->plane_mat[prim & ~ORC_PLANE_BIT];
    return c->mats + 11 * (size_t)c->prim_mat[prim];
}
static inline int any_pos(const float *v) { return v[0] > 0.f || v[1] > 0.f || v[2] > 0.f; }

/* Scene::trace: closest hit + normalised N (Scene.cpp:262) */
static int scene_trace(ts_ctx *c, v3 o, v3 d, float tmin, float tmax, orc_hit *h, v3 *P, v3 *N)
{
    orc_ray r = {o.x, o.y, o.z, tmin, d.x, d.y, d.z, tmax};
    orc_trace(c->s, &r, 1, h, NULL);
    c->rays_traced++;
    if (h->prim == ORC_MISS) return 0;
    float Pf[3], Nf[3];
    orc_hit_attrs_rays(c->s, &r, h, 1, Pf, Nf);
    v3 n = {Nf[0], Nf[1], Nf[2]};
    *N = divs(n, sqrtf(v3dot(n, n)));
    P->x = Pf[0]; P->y = Pf[1]; P->z = Pf[2];
    return 1;
}

static void phong_shade(ts_ctx *c, v3 d, uint32_t prim, v3 P, v3 N, float out[3])
{
    const float PI = 3.1415926535897932384626433832795028841972f, eps = 1e-4f;
    const float *m = mat_of(c, prim);
    out[0] = out[1] = out[2] = 0.0f;
    v3 l = v3sub(c->L, P);
    float falloff = v3dot(l, l);
    float dist = sqrtf(falloff);
    l = divs(l, dist);
    float intensity = 1.f;
    {   /* shadow ray, Phong.cpp:92-114 */
        orc_hit sh; v3 sP, sN;
        v3 so = v3add(P, v3scale(l, eps));
        if (scene_trace(c, so, l, 0.f, dist, &sh, &sP, &sN)) {
            const float *om = mat_of(c, sh.prim);
            if (!any_pos(om + 6)) return;                         /* opaque occluder */
            if (v3dot(sN, l) < 0) return;
            intensity = v3dot(sN, l);
            if (intensity < eps) return;
        }
    }
    float nDotL = v3dot(N, l);
    float f2 = 1.0f / (falloff * 4.0f * PI * PI);
    float diff = nDotL * f2 * c->wattage / 1.0f;
    if (!(diff > 0.0f)) diff = 0.0f;
    const float col[3] = {c->color.x, c->color.y, c->color.z};
    for (int k = 0; k < 3; k++) out[k] = col[k] * (diff * m[k] * m[k]) * intensity;      /* :146 */
    if (m[9] < INFINITY) {                                        /* :149-156 */
        v3 ml = {-l.x, -l.y, -l.z}, md = {-d.x, -d.y, -d.z};
        v3 r = v3add(ml, v3scale(N, 2 * v3dot(l, N)));
        float e = v3dot(md, r);
        if (e > 1.f) e = 1.f;
        if (!(e > 0.0f)) e = 0.0f;
        float hl = (float)pow((double)e, 500.0) * f2 * c->wattage / 1.0f;
        if (!(hl > 0.0f)) hl = 0.0f;
        out[0] += hl; out[1] += hl; out[2] += hl;
    }
}

/* Ray::reflect (Ray.h:143-165, non-path-tracing branch) */
static void ray_reflect(v3 d, v3 P, v3 N, v3 *o, v3 *dr)
{
    v3 r = v3sub(d, v3scale(N, 2 * v3dot(N, d)));
    r = divs(r, sqrtf(v3dot(r, r)));
    *dr = r;
    *o = v3add(P, v3scale(r, 1e-4f));
}

static void enter_or_exit(v3 d, v3 N, float index, float *n1, float *n2, v3 *n)
{
    if (v3dot(d, N) < 0) { *n1 = 1.0f; *n2 = index; *n = N; }
    else { *n1 = index; *n2 = 1.0f; n->x = -N.x; n->y = -N.y; n->z = -N.z; }
}

/* Ray::getReflectionCoefficient (Ray.h:168-199) */
static float fresnel(v3 d, v3 N, float index)
{
    float n1, n2; v3 n;
    enter_or_exit(d, N, index, &n1, &n2, &n);
    v3 md = {-d.x, -d.y, -d.z};
    float cosTheta = v3dot(md, n);
    float sinTheta = sinf(acosf(cosTheta));
    float p = powf((n1 / n2) * sinTheta, 2.f);
    if (p > 1.f) return 1;
    float sq = sqrtf(1.f - p);
    return powf((n1 * cosTheta - sq) / (n1 * cosTheta + sq), 2.f);
}

/* Ray::refract (Ray.h:202-243); returns via reflect on total internal reflection */
static void ray_refract(v3 d, v3 P, v3 N, float index, v3 *o, v3 *dr)
{
    float n1, n2; v3 n;
    enter_or_exit(d, N, index, &n1, &n2, &n);
    float dn = v3dot(d, n);
    float energy = (float)(1 - (pow(n1, 2) * (1 - pow(dn, 2)) / pow(n2, 2)));
    if (energy < 0) { ray_reflect(d, P, N, o, dr); return; }
    /* d_r = n1 * (d - n * dot(d, n)) / n2 - n * sqrt(energy)  -- Vector3 / float multiplies by 1/n2 */
    v3 t = v3scale(v3sub(d, v3scale(n, dn)), n1);
    t = divs(t, n2);
    *dr = v3sub(t, v3scale(n, sqrtf(energy)));
    *o = v3add(P, v3scale(*dr, 1e-4f));
}

static int trace_scene(ts_ctx *c, v3 o, v3 d, int depth, float res[3])
{
    res[0] = res[1] = res[2] = 0.0f;
    if (depth < 0) return 0;
    orc_hit h; v3 P, N;
    if (!scene_trace(c, o, d, 0.0f, 1e12f, &h, &P, &N)) return 1;   /* environment: m_bgColor = 0 */
    --depth;
    const float *m = mat_of(c, h.prim);
    phong_shade(c, d, h.prim, P, N, res);
    if (any_pos(m + 3)) {                                          /* reflective, Scene.cpp:302-312 */
        v3 ro, rd; float sub[3];
        ray_reflect(d, P, N, &ro, &rd);
        if (trace_scene(c, ro, rd, depth, sub)) for (int k = 0; k < 3; k++) res[k] += m[3 + k] * sub[k];
    }
    if (any_pos(m + 6)) {                                          /* refractive, :315-336 */
        float Rs = fresnel(d, N, m[10]);
        v3 ro, rd; float sub[3];
        ray_reflect(d, P, N, &ro, &rd);
        if (Rs > 0.01) if (trace_scene(c, ro, rd, depth, sub)) for (int k = 0; k < 3; k++) res[k] += m[6 + k] * sub[k] * Rs;
        ray_refract(d, P, N, m[10], &ro, &rd);
        if (trace_scene(c, ro, rd, depth, sub)) for (int k = 0; k < 3; k++) res[k] += m[6 + k] * sub[k] * (1.f - Rs);
    }
    return 1;
}

/* per-ray colours of Scene::traceScene(ray, result, depth); returns the number of Scene::trace calls made */
uint64_t orc_trace_scene(const orc_scene *s, const float *materials, const uint32_t *prim_mat, const orc_ray *rays,
                         uint64_t n, const float light[3], const float color[3], float wattage, int depth, float *rgb)
{
    ts_ctx c;
    c.s = s; c.mats = materials; c.prim_mat = prim_mat;
    c.L.x = light[0]; c.L.y = light[1]; c.L.z = light[2];
    c.color.x = color[0]; c.color.y = color[1]; c.color.z = color[2];
    c.wattage = wattage; c.rays_traced = 0;
    for (uint64_t i = 0; i < n; i++) {
        v3 o = {rays[i].ox, rays[i].oy, rays[i].oz}, d = {rays[i].dx, rays[i].dy, rays[i].dz};
        trace_scene(&c, o, d, depth, rgb + 3 * i);
    }
    return c.rays_traced;
}

/* the two pieces of Scene::traceScene that the PATH_TRACING recursion of miro_oracle_path.c shares with the one above */
int orc_ts_trace(ts_ctx *c, v3 o, v3 d, orc_hit *h, v3 *P, v3 *N) { return scene_trace(c, o, d, 0.0f, 1e12f, h, P, N); }
void orc_ts_shade(ts_ctx *c, v3 d, uint32_t prim, v3 P, v3 N, float out[3]) { phong_shade(c, d, prim, P, N, out); }
