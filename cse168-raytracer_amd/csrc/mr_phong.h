// mr_phong.h -- the two pieces of Phong::shade (Phong.cpp:44-160) that follow a hit, shared by the batched kernels
// (shadow_rays_kernel, shade kernels) and the fused frame kernel so that both produce the same bits:
//   shadow_ray_of   the shadow ray towards a point light (Phong.cpp:80-97)
//   phong_direct    direct light of an unoccluded hit: diffuse term + highlight (Phong.cpp:116-156), with the normal
//                   normalised as Scene::trace leaves it (Scene.cpp:262)
// Device code only.
#pragma once

#include <hip/hip_runtime.h>

namespace mr {

struct DirectLight {
    float L[3], color[3], diffuse[3], bg[3];
    float wattage;
};

// origin P + l*eps, direction l = normalise(L - P), tMin = 0, tMax = |L - P|   (PointLight::getLightDirection)
__device__ __forceinline__ void shadow_ray_of(const float P[3], float Lx, float Ly, float Lz, float4 &a, float4 &b) {
    constexpr float eps = 1e-4f;                                   // Miro.h:9
    const float Px = P[0], Py = P[1], Pz = P[2];
    float lx = Lx - Px, ly = Ly - Py, lz = Lz - Pz;
    const float falloff = (lx * lx + ly * ly) + lz * lz;
    const float len = sqrtf(falloff);
    const float inv = 1.0f / len;                                  // l /= sqrt(falloff)
    lx *= inv; ly *= inv; lz *= inv;
    a = make_float4(Px + lx * eps, Py + ly * eps, Pz + lz * eps, 0.0f);
    b = make_float4(lx, ly, lz, len);
}

// N: the un-normalised HitInfo::N; (dx, dy, dz): direction of the ray that produced the hit
__device__ __forceinline__ void phong_direct(const DirectLight &a, const float P[3], float N[3], float dx, float dy, float dz,
                                             float out[3]) {
    constexpr float kPI = 3.1415926535897932384626433832795028841972f;   // Miro.h:10
    {   // Scene.cpp:262 -- N.normalize()
        const float inv = 1.0f / sqrtf((N[0] * N[0] + N[1] * N[1]) + N[2] * N[2]);
        N[0] *= inv; N[1] *= inv; N[2] *= inv;
    }
    float l[3] = {a.L[0] - P[0], a.L[1] - P[1], a.L[2] - P[2]};
    const float falloff = (l[0] * l[0] + l[1] * l[1]) + l[2] * l[2];
    {
        const float inv = 1.0f / sqrtf(falloff);
        l[0] *= inv; l[1] *= inv; l[2] *= inv;
    }
    const float nDotL = (N[0] * l[0] + N[1] * l[1]) + N[2] * l[2];
    const float f2 = 1.0f / (falloff * 4.0f * kPI * kPI);                                   // Phong.cpp:140
    const float diff = fmaxf(0.0f, nDotL * f2 * a.wattage);
    for (int c = 0; c < 3; c++) out[c] = a.color[c] * (diff * a.diffuse[c] * a.diffuse[c]);  // :146
    // specular highlight (:149-156); Phong's default shininess 1 < infinity
    const float lDotN = (l[0] * N[0] + l[1] * N[1]) + l[2] * N[2];
    float eDotr = 0.0f;
    {
        const float two = 2 * lDotN;
        const float rx = -l[0] + two * N[0], ry = -l[1] + two * N[1], rz = -l[2] + two * N[2];
        eDotr = (-dx * rx + -dy * ry) + -dz * rz;
    }
    eDotr = powf(fmaxf(0.0f, fminf(1.0f, eDotr)), 500.0f);
    const float highlights = fmaxf(0.0f, eDotr * f2 * a.wattage);
    out[0] += highlights; out[1] += highlights; out[2] += highlights;
}

}  // namespace mr
