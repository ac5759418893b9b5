// mr_surface.h -- HitInfo::P / HitInfo::N of a hit record, shared by every kernel that continues from a hit
// (shadow-ray generation, HitInfo reconstruction, Phong shading, secondary rays).  Device code only.
//
//   triangle  P = (A + beta*(B-A)) + gamma*(C-A), N = ((1-beta-gamma)*nA + beta*nB) + gamma*nC   Triangle.cpp:160,162
//   sphere    P = o + t*d, N = (P - centre) / |P - centre|                                        Sphere.cpp:61-63
//   plane     P = o + t*d, N = the plane's normal as set                                          Plane.cpp:42-44
//
// N is what the object's intersect() leaves in HitInfo; Scene::trace normalises it afterwards (Scene.cpp:262).
#pragma once

#include <hip/hip_runtime.h>

#include "mr_internal.h"

namespace mr {

struct SurfacePtrs {
    const float *v, *n;
    const uint32_t *vi, *ni;
    const float4 *spheres;   // nullptr: triangles only
    const float4 *planes;    // nullptr: no unbounded objects
};

inline SurfacePtrs surface_ptrs(const DeviceScene &ds) {
    SurfacePtrs m;
    m.v = ds.v; m.n = ds.n; m.vi = ds.vi; m.ni = ds.ni; m.spheres = ds.spheres; m.planes = ds.planes;
    return m;
}

// The hit's ray given in registers (o, d): used by the fused frame kernel, and by surface() below once it has
// fetched the ray.  o / d are read only for spheres and planes.
template <bool WANT_N>
__device__ __forceinline__ void surface_od(const SurfacePtrs &m, float ox, float oy, float oz, float dx, float dy, float dz,
                                           float t, uint32_t prim, float beta, float gamma, float P[3], float N[3]) {
    if ((m.planes && (prim & kPlaneBit)) || (m.spheres && m.vi[3 * (size_t)prim] == kSphereSlot)) {
        P[0] = ox + t * dx; P[1] = oy + t * dy; P[2] = oz + t * dz;
        if (!WANT_N) return;
        if (prim & kPlaneBit) {
            const float4 pn = m.planes[2 * (size_t)(prim & ~kPlaneBit)];
            N[0] = pn.x; N[1] = pn.y; N[2] = pn.z;
        } else {
            const float4 sp = m.spheres[m.vi[3 * (size_t)prim + 1]];
            N[0] = P[0] - sp.x; N[1] = P[1] - sp.y; N[2] = P[2] - sp.z;
            const float inv = 1.0f / sqrtf((N[0] * N[0] + N[1] * N[1]) + N[2] * N[2]);   // N.normalize()
            N[0] *= inv; N[1] *= inv; N[2] *= inv;
        }
        return;
    }
    const size_t t3 = 3 * (size_t)prim;
    const uint32_t ia = m.vi[t3], ib = m.vi[t3 + 1], ic = m.vi[t3 + 2];
    for (int c = 0; c < 3; c++) {
        const float A = m.v[3 * (size_t)ia + c];
        const float BmA = m.v[3 * (size_t)ib + c] - A, CmA = m.v[3 * (size_t)ic + c] - A;
        P[c] = (A + beta * BmA) + gamma * CmA;
    }
    if (!WANT_N) return;
    const uint32_t ja = m.ni[t3], jb = m.ni[t3 + 1], jc = m.ni[t3 + 2];
    const float alpha = 1 - beta - gamma;
    for (int c = 0; c < 3; c++)
        N[c] = (alpha * m.n[3 * (size_t)ja + c] + beta * m.n[3 * (size_t)jb + c]) + gamma * m.n[3 * (size_t)jc + c];
}

// rays may be nullptr for triangle-only scenes (the host checks); k = index of the hit's ray
template <bool WANT_N>
__device__ __forceinline__ void surface(const SurfacePtrs &m, const mr_ray *rays, unsigned long long k, float t,
                                        uint32_t prim, float beta, float gamma, float P[3], float N[3]) {
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((m.planes && (prim & kPlaneBit)) || (m.spheres && m.vi[3 * (size_t)prim] == kSphereSlot)) {
        ra = reinterpret_cast<const float4 *>(rays)[2 * k];
        rb = reinterpret_cast<const float4 *>(rays)[2 * k + 1];
    }
    surface_od<WANT_N>(m, ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, t, prim, beta, gamma, P, N);
}

// material id of a hit: planes carry theirs, bounded objects look it up (nullptr table: material 0)
__device__ __forceinline__ uint32_t material_id(const SurfacePtrs &m, const uint32_t *prim_mat, uint32_t prim) {
    if (m.planes && (prim & kPlaneBit)) return __float_as_uint(m.planes[2 * (size_t)(prim & ~kPlaneBit)].w);
    return prim_mat ? prim_mat[prim] : 0u;
}

}  // namespace mr
