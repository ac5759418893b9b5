// mr_recursion.h -- the per-ray pieces of Scene::traceScene's recursion (Scene.cpp:270-346) and of Phong::shade with
// per-triangle materials (Phong.cpp:44-160), shared by the batched kernels of mr_bounce.hip and the fused level kernel
// of mr_level.hip so that both produce the same bits:
//   light_scale_of       what Phong::shade does with the shadow hit (Phong.cpp:97-113): opaque occluder -> 0,
//                        refractive occluder -> dot(N, l) (0 if negative or < epsilon), no occluder -> 1
//   phong_terms          diffuse term and highlight of a hit (Phong.cpp:116-156) for a material record
//   ChildGen<PATH>       Ray::reflect / getReflectionCoefficient / refract (Ray.h:143-243, Scene.cpp:302-336), or their
//                        PATH_TRACING build plus Ray::random (Ray.h:124-158,235-239)
//   write_children       wave64 ballot compaction of the children, one atomic per workgroup
// Device code only.
#pragma once

#include <hip/hip_runtime.h>

#include "miro_math.h"
#include "mr_internal.h"
#include "mr_surface.h"

namespace mr {
namespace rec {

constexpr float kEps = 1e-4f;                                                    // Miro.h:9
constexpr float kPI = 3.1415926535897932384626433832795028841972f;               // Miro.h:10
constexpr float kInf = __builtin_huge_valf();

struct MeshMat {
    SurfacePtrs s;
    const float *mats;            // 11 floats per material: diffuse, specular, transmission, shininess, index
    const uint32_t *prim_mat;     // NULL: material 0 everywhere
};

inline MeshMat mesh_of(const DeviceScene &ds) {
    MeshMat m;
    m.s = surface_ptrs(ds); m.mats = ds.materials; m.prim_mat = ds.prim_material;
    return m;
}

__device__ __forceinline__ const float *material_of(const MeshMat &m, uint32_t prim) {
    return m.mats + 11 * (size_t)material_id(m.s, m.prim_mat, prim);
}
__device__ __forceinline__ bool any_pos(const float *c) { return c[0] > 0.f || c[1] > 0.f || c[2] > 0.f; }

// HitInfo::P and the normalised N that Scene::trace hands to its callers (mr_surface.h, Scene.cpp:262), from the ray
// (origin o, direction d) and its hit record in registers
__device__ __forceinline__ void surface_point_od(const MeshMat &m, float ox, float oy, float oz, float dx, float dy, float dz,
                                                 float t, uint32_t prim, float beta, float gamma, float P[3], float N[3]) {
    surface_od<true>(m.s, ox, oy, oz, dx, dy, dz, t, prim, beta, gamma, P, N);
    const float inv = 1.0f / sqrtf((N[0] * N[0] + N[1] * N[1]) + N[2] * N[2]);
    N[0] *= inv; N[1] *= inv; N[2] *= inv;
}
__device__ __forceinline__ void surface_point(const MeshMat &m, const mr_ray *rays, unsigned long long k, const float4 h,
                                              float P[3], float N[3]) {
    const float4 ra = reinterpret_cast<const float4 *>(rays)[2 * k], rb = reinterpret_cast<const float4 *>(rays)[2 * k + 1];
    surface_point_od(m, ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, h.x, __float_as_uint(h.y), h.z, h.w, P, N);
}

// The factor Phong::shade puts on the light behind a shadow hit (Phong.cpp:97-113).  sa / sb: the shadow ray; sh: its
// hit record (prim = MR_MISS: no occluder).
__device__ __forceinline__ float light_scale_of(const MeshMat &m, const float4 sa, const float4 sb, const float4 sh) {
    const uint32_t prim = __float_as_uint(sh.y);
    float scale = 1.0f;
    if (prim != MR_MISS) {
        scale = 0.0f;
        const float *om = material_of(m, prim);
        if (any_pos(om + 6)) {                                            // refractive occluder (Phong.cpp:99-113)
            float P[3], N[3];
            surface_point_od(m, sa.x, sa.y, sa.z, sb.x, sb.y, sb.z, sh.x, prim, sh.z, sh.w, P, N);
            const float d = (N[0] * sb.x + N[1] * sb.y) + N[2] * sb.z;
            if (!(d < 0) && !(d < kEps)) scale = d;
        }
    }
    return scale;
}

struct LightArgs {
    float L[3], color[3], wattage;
};

// Phong::shade's direct light at a hit (Phong.cpp:116-156) in two parts: diffuse[c] (to be multiplied by the light scale,
// :146) and the highlight (:149-156, added unscaled; 0 for a material of infinite shininess).  The shaded value of a hit
// with light scale s != 0 is diffuse[c] * s + highlight, and 0 for s == 0 (Phong.cpp:100-103 skips the light).
// N normalised; (dx, dy, dz) the direction of the ray that produced the hit.
__device__ __forceinline__ void phong_terms(const LightArgs &a, const float *mt, const float P[3], const float N[3], float dx,
                                            float dy, float dz, float diffuse[3], float &highlight) {
    float l[3] = {a.L[0] - P[0], a.L[1] - P[1], a.L[2] - P[2]};
    const float falloff = (l[0] * l[0] + l[1] * l[1]) + l[2] * l[2];
    const float inv = 1.0f / sqrtf(falloff);
    l[0] *= inv; l[1] *= inv; l[2] *= inv;
    const float nDotL = (N[0] * l[0] + N[1] * l[1]) + N[2] * l[2];
    const float f2 = 1.0f / (falloff * 4.0f * kPI * kPI);
    const float diff = fmaxf(0.0f, nDotL * f2 * a.wattage);
    for (int c = 0; c < 3; c++) diffuse[c] = a.color[c] * (diff * mt[c] * mt[c]);            // Phong.cpp:146
    highlight = 0.0f;
    if (mt[9] < kInf) {                                                                       // :149-156
        const float two = 2 * ((l[0] * N[0] + l[1] * N[1]) + l[2] * N[2]);
        const float rx = -l[0] + two * N[0], ry = -l[1] + two * N[1], rz = -l[2] + two * N[2];
        float e = (-dx * rx + -dy * ry) + -dz * rz;
        e = powf(fmaxf(0.0f, fminf(1.0f, e)), 500.0f);
        highlight = fmaxf(0.0f, e * f2 * a.wattage);
    }
}
__device__ __forceinline__ void phong_combine(const float diffuse[3], float highlight, float scale, float out[3]) {
    if (scale == 0.0f) { out[0] = 0.f; out[1] = 0.f; out[2] = 0.f; return; }
    for (int c = 0; c < 3; c++) out[c] = diffuse[c] * scale + highlight;
}

// weight * L / spp added to the ray's pixel.  Called by ALL lanes of the wave (a lane without a contribution brings
// v = 0 and, if it has no ray at all, pix = ~0): rays of one pixel sit in adjacent lanes -- the spp samples of a pixel at
// the first level, the children of neighbouring parents later -- and 16 lanes adding to one address serialise in the L2
// (the atomics were 2 of the 3 ms of a bunny 1024x1024x16 level).  So every run of equal pixels is summed inside the wave
// first (a segmented suffix sum, six shuffle steps) and its first lane issues the run's three atomics.  Float atomics:
// the order of additions, here and in memory, is not reproducible.
__device__ __forceinline__ void accumulate_runs(float *rgb, uint32_t pix, float v0, float v1, float v2) {
    const int lane = threadIdx.x & 63;
    const uint32_t prev = __shfl_up(pix, 1, 64);
    const unsigned long long heads = __ballot(lane == 0 || prev != pix);          // bit 0 is always set
    // last lane of my run: one before the next head above me
    const unsigned long long above = lane == 63 ? 0ull : heads >> (lane + 1);
    const int last = above ? lane + (int)__builtin_ctzll(above) : 63;
    for (int d = 1; d < 64; d <<= 1) {
        const float a0 = __shfl_down(v0, d, 64), a1 = __shfl_down(v1, d, 64), a2 = __shfl_down(v2, d, 64);
        if (lane + d <= last) { v0 += a0; v1 += a1; v2 += a2; }
    }
    if (((heads >> lane) & 1ull) && pix != 0xFFFFFFFFu) {
        if (v0 != 0.0f) atomicAdd(&rgb[3 * (size_t)pix], v0);
        if (v1 != 0.0f) atomicAdd(&rgb[3 * (size_t)pix + 1], v1);
        if (v2 != 0.0f) atomicAdd(&rgb[3 * (size_t)pix + 2], v2);
    }
}

// ---- children -------------------------------------------------------------------------------------------------------
// up to four per ray: 0 = mirror reflection (weight x ks), 1 = Fresnel reflection (x kt Rs, if Rs > 0.01), 2 = refraction
// or its total internal reflection (x kt (1-Rs)) as Scene.cpp:302-336; 3 = the diffuse bounce of Ray::random (x kd), path
// tracing only.
//
// Generation is split in two so that a lane never holds more than one child: plan() decides which children exist (it
// needs the Fresnel coefficient, nothing else), the workgroup reserves their slots, make(kind, ...) then builds one child
// at a time and the caller stores it.  (All four children at once cost 36 registers on top of the generators' double
// arithmetic: the fused level kernel fell to 4 waves per SIMD.)
__device__ __forceinline__ void reflect_dir(const float d[3], const float N[3], float r[3]) {       // Ray.h:160-162
    const float two = 2 * ((N[0] * d[0] + N[1] * d[1]) + N[2] * d[2]);
    r[0] = d[0] - two * N[0]; r[1] = d[1] - two * N[1]; r[2] = d[2] - two * N[2];
    const float inv = 1.0f / sqrtf((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
    r[0] *= inv; r[1] *= inv; r[2] *= inv;
}

__host__ __device__ __forceinline__ uint32_t pcg32(uint32_t x) {
    const uint32_t state = x * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
__device__ __forceinline__ float unit01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// Ray::alignToVector (Ray.h:86-91): direction = alignHemisphereToVector(v, theta, phi) (Utility.h:34-50), origin = P +
// epsilon * direction
__device__ __forceinline__ void align_to_vector(const float v[3], const float P[3], float theta, float phi, float org[3], float dir[3]) {
    const float sp = mm_sinf(phi), cp = mm_cosf(phi), st = mm_sinf(theta), ct = mm_cosf(theta);
    const float u1 = sp * ct, u2 = sp * st, u3 = cp;
    const float ez[3] = {0.f, 0.f, 1.f}, ey[3] = {0.f, 1.f, 0.f};
    float t1[3], t2[3];
    cross3(ez, v, t1);
    if ((double)((t1[0] * t1[0] + t1[1] * t1[1]) + t1[2] * t1[2]) < 1e-6) cross3(ey, v, t1);     // float < double literal
    cross3(t1, v, t2);
    for (int c = 0; c < 3; c++) dir[c] = (t1[c] * u1 + t2[c] * u2) + v[c] * u3;
    const float inv = 1.0f / sqrtf((dir[0] * dir[0] + dir[1] * dir[1]) + dir[2] * dir[2]);
    for (int c = 0; c < 3; c++) { dir[c] *= inv; org[c] = P[c] + dir[c] * kEps; }
}

// PATH = false: Ray::reflect / getReflectionCoefficient / refract as compiled at HEAD (Ray.h:143-243).
// PATH = true: the PATH_TRACING build of those generators (Ray.h:149-158, 235-239) and Ray::random (Ray.h:124-140): every
// child direction is drawn from a lobe -- alignHemisphereToVector around the mirror direction / the refracted direction
// with phi = acos(pow(u1, 1/(1+shininess))), or around the normal with phi = asin(sqrt(u1)) (cosine-weighted) for the
// diffuse bounce -- theta = 2 pi u2.  The reference draws u1, u2 from rand(); here they come from the counter-based
// generator that jitters the eye rays, keyed by (seed, ray id, bounce, child kind), integer-exact on the device and in the
// oracle; the transcendentals are miro_math.h on both sides, so the ray sets are the same bits.
template <bool PATH>
struct ChildGen {
    const float *mt;             // the hit's material record
    float P[3], N[3], d[3], w0[3];
    float Rs;                    // Fresnel coefficient (set by plan() for a refractive material)
    uint32_t hray;               // PATH: pcg32(hbase ^ ray id) + bounce * 4, the ray's key at this bounce

    // entering / leaving: n1, n2 and the normal on the ray's side (Scene.cpp:317-327)
    __device__ __forceinline__ void interface(float &n1, float &n2, float nn[3]) const {
        const float index = mt[10];
        const float dN = (d[0] * N[0] + d[1] * N[1]) + d[2] * N[2];
        const bool enter = dN < 0;
        n1 = enter ? 1.0f : index; n2 = enter ? index : 1.0f;
        nn[0] = enter ? N[0] : -N[0]; nn[1] = enter ? N[1] : -N[1]; nn[2] = enter ? N[2] : -N[2];
    }

    // refl / refr / diff: the material reflects / refracts / scatters and the caller wants that child
    __device__ __forceinline__ void plan(bool refl, bool refr, bool diff, bool emit[4]) {
        emit[0] = refl; emit[1] = false; emit[2] = refr; emit[3] = PATH && diff;
        Rs = 1.0f;
        if (refr) {
            float n1, n2, nn[3];
            interface(n1, n2, nn);
            // Ray::getReflectionCoefficient (Ray.h:168-199); under PATH on the shared transcendentals
            const float cosT = (-d[0] * nn[0] + -d[1] * nn[1]) + -d[2] * nn[2];
            if (PATH) {
                const float sinT = mm_sinf(mm_acosf(cosT));
                const float q = (n1 / n2) * sinT, p = q * q;                  // powf(x, 2.f)
                if (!(p > 1.f)) {
                    const float sq = sqrtf(1.f - p), fr = (n1 * cosT - sq) / (n1 * cosT + sq);
                    Rs = fr * fr;
                }
            } else {
                const float sinT = sinf(acosf(cosT));
                const float p = powf((n1 / n2) * sinT, 2.f);
                if (!(p > 1.f)) {
                    const float sq = sqrtf(1.f - p);
                    Rs = powf((n1 * cosT - sq) / (n1 * cosT + sq), 2.f);
                }
            }
            emit[1] = Rs > 0.01f;
        }
    }

    // a lobe sample around v (PATH): Ray::reflect / refract's draw (Ray.h:149-158, 235-239)
    __device__ __forceinline__ void lobe(uint32_t kind, const float v[3], float org[3], float dir[3]) const {
        const float lobe_exp = 1.0f / (1.0f + mt[9]);
        const uint32_t hk = pcg32(hray + kind);
        const float phi = mm_acosf01(mm_powf01(unit01(pcg32(hk)), lobe_exp));
        const float theta = (2.0f * kPI) * unit01(pcg32(hk ^ 0x68bc21ebu));
        align_to_vector(v, P, theta, phi, org, dir);
    }
    // Ray::reflect (Ray.h:143-165): the mirror direction, or (PATH) a fresh draw around it per call
    __device__ __forceinline__ void reflect(uint32_t kind, float org[3], float dir[3]) const {
        if (PATH) {
            const float two = 2 * ((N[0] * d[0] + N[1] * d[1]) + N[2] * d[2]);
            const float dr[3] = {d[0] - two * N[0], d[1] - two * N[1], d[2] - two * N[2]};
            lobe(kind, dr, org, dir);
        } else {
            reflect_dir(d, N, dir);
            for (int c = 0; c < 3; c++) org[c] = P[c] + dir[c] * kEps;
        }
    }

    // child `kind` (one that plan() announced)
    __device__ __forceinline__ void make(int kind, float org[3], float dir[3], float wgt[3]) const {
        if (kind == 0) {                                                  // Scene.cpp:302-312
            reflect(0u, org, dir);
            for (int c = 0; c < 3; c++) wgt[c] = w0[c] * mt[3 + c];
        } else if (kind == 1) {                                           // Scene.cpp:315-330
            reflect(1u, org, dir);
            for (int c = 0; c < 3; c++) wgt[c] = w0[c] * mt[6 + c] * Rs;
        } else if (kind == 2) {                                           // Ray::refract (Ray.h:202-243)
            float n1, n2, nn[3];
            interface(n1, n2, nn);
            const float dn = (d[0] * nn[0] + d[1] * nn[1]) + d[2] * nn[2];
            const float energy = (float)(1 - (((double)n1 * (double)n1) * (1 - (double)dn * (double)dn) / ((double)n2 * (double)n2)));
            if (energy < 0) {
                reflect(2u, org, dir);
            } else {
                const float inv_n2 = 1.0f / n2, se = sqrtf(energy);
                if (PATH) {
                    float dr[3];
                    for (int c = 0; c < 3; c++) dr[c] = ((d[c] - nn[c] * dn) * n1) * inv_n2 - nn[c] * se;
                    lobe(2u, dr, org, dir);
                } else {
                    for (int c = 0; c < 3; c++) {
                        const float t = ((d[c] - nn[c] * dn) * n1) * inv_n2;
                        dir[c] = t - nn[c] * se;
                        org[c] = P[c] + dir[c] * kEps;
                    }
                }
            }
            for (int c = 0; c < 3; c++) wgt[c] = w0[c] * mt[6 + c] * (1.f - Rs);
        } else {                                                          // Ray::random (Ray.h:124-140)
            const uint32_t hk = pcg32(hray + 3u);
            const float phi = mm_asinf01(sqrtf(unit01(pcg32(hk))));
            const float theta = (2.0f * kPI) * unit01(pcg32(hk ^ 0x68bc21ebu));
            align_to_vector(N, P, theta, phi, org, dir);
            for (int c = 0; c < 3; c++) wgt[c] = w0[c] * mt[c];
        }
    }
};
__device__ __forceinline__ uint32_t child_id(uint32_t id, int kind) { return pcg32(id ^ (0x9e3779b9u * (uint32_t)(kind + 1))); }

// Output slots for a workgroup's children: every wave brings its count, ONE atomicAdd per workgroup reserves the range (a
// single counter word drains ~88 atomics per microsecond -- one atomic per wave made the generators atomic-bound: 2.7 ms for
// the 16.8 M rays of a bunny frame against 0.7 ms for tracing their children).  Called by all threads of the workgroup.
template <int BLOCK>
__device__ __forceinline__ unsigned long long workgroup_reserve(unsigned wave_total, unsigned long long *count) {
    __shared__ unsigned s_tot[BLOCK / 64];
    __shared__ unsigned long long s_base;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_tot[wave] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int w = 0; w < BLOCK / 64; w++) tot += s_tot[w];
        s_base = tot ? atomicAdd(count, (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
    unsigned long long base = s_base;
    for (int w = 0; w < wave; w++) base += s_tot[w];
    __syncthreads();                                      // s_tot / s_base are reused by the next round
    return base;
}

struct ChildQueue {
    mr_ray *rays;
    float *weights;
    uint32_t *pixels, *ids;       // ids may be NULL
    unsigned long long *count;
    unsigned long long capacity;  // rays the arrays have room for: a child whose slot lies beyond is counted, not stored
    uint8_t *octants;             // may be NULL: per child, the sign bits of its direction (x | y << 1 | z << 2) for mr_order_by_octant
};
__device__ __forceinline__ uint8_t octant_of(const float dir[3]) {
    return (uint8_t)((dir[0] < 0.0f ? 1u : 0u) | (dir[1] < 0.0f ? 2u : 0u) | (dir[2] < 0.0f ? 4u : 0u));
}

// wave64 compaction: one ballot per child kind; the workgroup's waves share one atomic.  Called by all threads of the
// workgroup (lanes without a ray bring emit[] = false); a lane's children are built one at a time, straight into their
// slots.
template <int BLOCK, bool PATH>
__device__ __forceinline__ void write_children(const ChildQueue &q, const ChildGen<PATH> &g, const bool emit[4], uint32_t pix, uint32_t id) {
    constexpr int NK = PATH ? 4 : 3;
    const int lane = threadIdx.x & 63;
    unsigned long long mk[NK];
    int cn[NK], tot = 0;
    for (int j = 0; j < NK; j++) { mk[j] = __ballot(emit[j]); cn[j] = __popcll(mk[j]); tot += cn[j]; }
    const unsigned long long base = workgroup_reserve<BLOCK>((unsigned)tot, q.count);
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned long long before = base;
#pragma unroll
    for (int j = 0; j < NK; j++) {
        const unsigned long long s = before + __popcll(mk[j] & lt);
        if (emit[j] && s < q.capacity) {
            float org[3], dir[3], wgt[3];
            g.make(j, org, dir, wgt);
            reinterpret_cast<float4 *>(q.rays)[2 * s] = make_float4(org[0], org[1], org[2], 0.0f);
            reinterpret_cast<float4 *>(q.rays)[2 * s + 1] = make_float4(dir[0], dir[1], dir[2], 1e12f);
            q.weights[3 * s] = wgt[0]; q.weights[3 * s + 1] = wgt[1]; q.weights[3 * s + 2] = wgt[2];
            q.pixels[s] = pix;
            if (q.ids) q.ids[s] = child_id(id, j);
            if (q.octants) q.octants[s] = octant_of(dir);
        }
        before += cn[j];
    }
}

// a lane's children, one after the other from slot `s` on (the caller has reserved the slots)
template <bool PATH>
__device__ __forceinline__ void write_children_at(const ChildQueue &q, const ChildGen<PATH> &g, const bool emit[4], uint32_t pix,
                                                  uint32_t id, unsigned long long s) {
    constexpr int NK = PATH ? 4 : 3;
#pragma unroll
    for (int j = 0; j < NK; j++) {
        if (emit[j]) {
            if (s < q.capacity) {
                float org[3], dir[3], wgt[3];
                g.make(j, org, dir, wgt);
                reinterpret_cast<float4 *>(q.rays)[2 * s] = make_float4(org[0], org[1], org[2], 0.0f);
                reinterpret_cast<float4 *>(q.rays)[2 * s + 1] = make_float4(dir[0], dir[1], dir[2], 1e12f);
                q.weights[3 * s] = wgt[0]; q.weights[3 * s + 1] = wgt[1]; q.weights[3 * s + 2] = wgt[2];
                q.pixels[s] = pix;
                if (q.ids) q.ids[s] = child_id(id, j);
                if (q.octants) q.octants[s] = octant_of(dir);
            }
            s++;
        }
    }
}

}  // namespace rec
}  // namespace mr
