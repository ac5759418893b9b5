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
    const v3 *V = (const v3 *)s->v, *Nn = (const v3 *)s->n;
    v3 L = {light[0], light[1], light[2]};
    uint64_t npix = n / (uint64_t)spp;
    for (uint64_t pix = 0; pix < npix; pix++) {
        float acc[3] = {0, 0, 0};
        for (int sm = 0; sm < spp; sm++) {
            uint64_t k = pix * (uint64_t)spp + (uint64_t)sm;
            float c[3] = {0, 0, 0};                                 /* miss: m_bgColor = 0 */
            if (hits[k].prim != ORC_MISS && !occluded[k]) {
                uint32_t p = hits[k].prim;
                float beta = hits[k].beta, gamma = hits[k].gamma;
                v3 A = V[s->vi[3*p]], B = V[s->vi[3*p+1]], C = V[s->vi[3*p+2]];
                v3 BmA = v3sub(B, A), CmA = v3sub(C, A);
                v3 P = v3add(v3add(A, v3scale(BmA, beta)), v3scale(CmA, gamma));
                v3 nA = Nn[s->ni[3*p]], nB = Nn[s->ni[3*p+1]], nC = Nn[s->ni[3*p+2]];
                v3 N = v3add(v3add(v3scale(nA, 1 - beta - gamma), v3scale(nB, beta)), v3scale(nC, gamma));
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
