// mr_bounce.hip -- specular secondary rays ("next" row f4 of SURVEY.md section 8): the recursion of
// Scene::traceScene (Scene.cpp:270-346) unrolled into wavefront bounces.
//   light_scale_kernel      what Phong::shade does with the shadow hit (Phong.cpp:97-113): opaque occluder -> 0,
//                           refractive occluder -> dot(N, l) (0 if negative or < epsilon), no occluder -> 1
//   shade_accumulate_kernel Phong::shade per ray with per-triangle materials, times the ray's path weight, added to
//                           its pixel (float atomics: the order of additions is not reproducible for multi-bounce
//                           frames; the single-bounce path of mr_shade_direct stays deterministic)
//   secondary_rays_kernel   Ray::reflect / getReflectionCoefficient / refract (Ray.h:143-243) for every hit on a
//                           reflective or refractive material: up to three children per ray, each with
//                           weight * (specular | transmission*Rs | transmission*(1-Rs)) (Scene.cpp:302-336), written
//                           compacted by wave64 ballots + prefix sums
#include <hip/hip_runtime.h>

#include "miro_math.h"
#include "mr_internal.h"
#include "mr_surface.h"

namespace mr {
namespace {

constexpr int kBlock = 256;
constexpr float kEps = 1e-4f;
constexpr float kPI = 3.1415926535897932384626433832795028841972f;
constexpr float kInf = __builtin_huge_valf();

struct MeshMat {
    SurfacePtrs s;
    const float *mats;            // 11 floats per material: diffuse, specular, transmission, shininess, index
    const uint32_t *prim_mat;     // NULL: material 0 everywhere
};

__device__ __forceinline__ const float *material_of(const MeshMat &m, uint32_t prim) {
    return m.mats + 11 * (size_t)material_id(m.s, m.prim_mat, prim);
}
__device__ __forceinline__ bool any_pos(const float *c) { return c[0] > 0.f || c[1] > 0.f || c[2] > 0.f; }

// HitInfo::P and the normalised N that Scene::trace hands to its callers (mr_surface.h, Scene.cpp:262)
__device__ __forceinline__ void surface_point(const MeshMat &m, const mr_ray *rays, unsigned long long k, const float4 h,
                                              float P[3], float N[3]) {
    surface<true>(m.s, rays, k, h.x, __float_as_uint(h.y), h.z, h.w, P, N);
    const float inv = 1.0f / sqrtf((N[0] * N[0] + N[1] * N[1]) + N[2] * N[2]);
    N[0] *= inv; N[1] *= inv; N[2] *= inv;
}

__global__ __launch_bounds__(kBlock) void light_scale_kernel(MeshMat m, const mr_ray *shadow_rays, const mr_hit *shadow_hits,
                                                             const uint32_t *src, const unsigned long long *count,
                                                             unsigned long long max_n, float *light_scale) {
    unsigned long long n = *count;
    if (n > max_n) n = max_n;
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < n; k += stride) {
        const float4 h = reinterpret_cast<const float4 *>(shadow_hits)[k];
        const uint32_t prim = __float_as_uint(h.y);
        float scale = 1.0f;
        if (prim != MR_MISS) {
            scale = 0.0f;
            const float *om = material_of(m, prim);
            if (any_pos(om + 6)) {                                    // refractive occluder (Phong.cpp:99-113)
                float P[3], N[3];
                surface_point(m, shadow_rays, k, h, P, N);
                const float4 rb = reinterpret_cast<const float4 *>(shadow_rays)[2 * k + 1];
                const float d = (N[0] * rb.x + N[1] * rb.y) + N[2] * rb.z;
                if (!(d < 0) && !(d < kEps)) scale = d;
            }
        }
        light_scale[src[k]] = scale;
    }
}

struct AccumArgs {
    MeshMat m;
    const mr_ray *rays;
    const mr_hit *hits;
    const float *weights;         // rgb per ray or NULL (= 1)
    const uint32_t *pixels;       // pixel per ray or NULL (= ray index / spp)
    const float *light_scale;     // per ray
    float L[3], color[3], wattage, inv_spp;
    uint32_t spp;
    unsigned long long n;
    float *rgb;
};

__global__ __launch_bounds__(kBlock) void shade_accumulate_kernel(AccumArgs a) {
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < a.n; k += stride) {
        const float4 h = reinterpret_cast<const float4 *>(a.hits)[k];
        const uint32_t prim = __float_as_uint(h.y);
        if (prim == MR_MISS) continue;                                // m_bgColor = 0 contributes nothing
        const float scale = a.light_scale[k];
        float out[3] = {0.f, 0.f, 0.f};
        if (scale != 0.0f) {
            const float *mt = material_of(a.m, prim);
            float P[3], N[3];
            surface_point(a.m, a.rays, k, h, P, N);
            float l[3] = {a.L[0] - P[0], a.L[1] - P[1], a.L[2] - P[2]};
            const float falloff = (l[0] * l[0] + l[1] * l[1]) + l[2] * l[2];
            const float inv = 1.0f / sqrtf(falloff);
            l[0] *= inv; l[1] *= inv; l[2] *= inv;
            const float nDotL = (N[0] * l[0] + N[1] * l[1]) + N[2] * l[2];
            const float f2 = 1.0f / (falloff * 4.0f * kPI * kPI);
            const float diff = fmaxf(0.0f, nDotL * f2 * a.wattage);
            for (int c = 0; c < 3; c++) out[c] = a.color[c] * (diff * mt[c] * mt[c]) * scale;     // Phong.cpp:146
            if (mt[9] < kInf) {                                                                       // :149-156
                const float4 rb = reinterpret_cast<const float4 *>(a.rays)[2 * k + 1];
                const float two = 2 * ((l[0] * N[0] + l[1] * N[1]) + l[2] * N[2]);
                const float rx = -l[0] + two * N[0], ry = -l[1] + two * N[1], rz = -l[2] + two * N[2];
                float e = (-rb.x * rx + -rb.y * ry) + -rb.z * rz;
                e = powf(fmaxf(0.0f, fminf(1.0f, e)), 500.0f);
                const float hl = fmaxf(0.0f, e * f2 * a.wattage);
                out[0] += hl; out[1] += hl; out[2] += hl;
            }
        }
        const uint32_t pix = a.pixels ? a.pixels[k] : (uint32_t)(k / a.spp);
        for (int c = 0; c < 3; c++) {
            const float w = a.weights ? a.weights[3 * k + c] : 1.0f;
            const float v = out[c] * w * a.inv_spp;
            if (v != 0.0f) atomicAdd(&a.rgb[3 * (size_t)pix + c], v);
        }
    }
}

struct BounceArgs {
    MeshMat m;
    const mr_ray *rays;
    const mr_hit *hits;
    const float *weights;
    const uint32_t *pixels;
    uint32_t spp;
    unsigned long long n;
    mr_ray *out_rays;
    float *out_weights;
    uint32_t *out_pixels;
    unsigned long long *count;
};

// Output slots for a workgroup's children: every wave brings its count, ONE atomicAdd per workgroup reserves the range (a
// single counter word drains ~88 atomics per microsecond -- one atomic per wave made the generators atomic-bound: 2.7 ms for
// the 16.8 M rays of a bunny frame against 0.7 ms for tracing their children).  Called by all threads of the workgroup.
__device__ __forceinline__ unsigned long long workgroup_reserve(unsigned wave_total, unsigned long long *count) {
    __shared__ unsigned s_tot[kBlock / 64];
    __shared__ unsigned long long s_base;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_tot[wave] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int w = 0; w < kBlock / 64; w++) tot += s_tot[w];
        s_base = tot ? atomicAdd(count, (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
    unsigned long long base = s_base;
    for (int w = 0; w < wave; w++) base += s_tot[w];
    __syncthreads();                                      // s_tot / s_base are reused by the next round
    return base;
}

__device__ __forceinline__ void reflect_dir(const float d[3], const float N[3], float r[3]) {       // Ray.h:160-162
    const float two = 2 * ((N[0] * d[0] + N[1] * d[1]) + N[2] * d[2]);
    r[0] = d[0] - two * N[0]; r[1] = d[1] - two * N[1]; r[2] = d[2] - two * N[2];
    const float inv = 1.0f / sqrtf((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
    r[0] *= inv; r[1] *= inv; r[2] *= inv;
}

__global__ __launch_bounds__(kBlock) void secondary_rays_kernel(BounceArgs a) {
    const int lane = threadIdx.x & 63;
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    const unsigned long long n_round = (a.n + (unsigned long long)kBlock - 1ull) / kBlock * kBlock;     // whole workgroups
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < n_round; k += stride) {
        // up to three children: 0 = mirror reflection, 1 = Fresnel reflection, 2 = refraction (or its total internal reflection)
        bool emit[3] = {false, false, false};
        float dir[3][3], org[3][3], wgt[3][3];
        uint32_t pix = 0;
        if (k < a.n) {
            const float4 h = reinterpret_cast<const float4 *>(a.hits)[k];
            const uint32_t prim = __float_as_uint(h.y);
            if (prim != MR_MISS) {
                const float *mt = material_of(a.m, prim);
                const bool refl = any_pos(mt + 3), refr = any_pos(mt + 6);
                if (refl || refr) {
                    float P[3], N[3];
                    surface_point(a.m, a.rays, k, h, P, N);
                    const float4 rb = reinterpret_cast<const float4 *>(a.rays)[2 * k + 1];
                    const float d[3] = {rb.x, rb.y, rb.z};
                    float w0[3] = {1.f, 1.f, 1.f};
                    if (a.weights) { w0[0] = a.weights[3 * k]; w0[1] = a.weights[3 * k + 1]; w0[2] = a.weights[3 * k + 2]; }
                    pix = a.pixels ? a.pixels[k] : (uint32_t)(k / a.spp);
                    float r[3];
                    reflect_dir(d, N, r);
                    if (refl) {                                                       // Scene.cpp:302-312
                        emit[0] = true;
                        for (int c = 0; c < 3; c++) { dir[0][c] = r[c]; org[0][c] = P[c] + r[c] * kEps; wgt[0][c] = w0[c] * mt[3 + c]; }
                    }
                    if (refr) {                                                       // Scene.cpp:315-336
                        const float index = mt[10];
                        const float dN = (d[0] * N[0] + d[1] * N[1]) + d[2] * N[2];
                        const bool enter = dN < 0;
                        const float n1 = enter ? 1.0f : index, n2 = enter ? index : 1.0f;
                        const float nn[3] = {enter ? N[0] : -N[0], enter ? N[1] : -N[1], enter ? N[2] : -N[2]};
                        // Ray::getReflectionCoefficient (Ray.h:168-199)
                        const float cosT = (-d[0] * nn[0] + -d[1] * nn[1]) + -d[2] * nn[2];
                        const float sinT = sinf(acosf(cosT));
                        const float p = powf((n1 / n2) * sinT, 2.f);
                        float Rs = 1.0f;
                        if (!(p > 1.f)) {
                            const float sq = sqrtf(1.f - p);
                            Rs = powf((n1 * cosT - sq) / (n1 * cosT + sq), 2.f);
                        }
                        if (Rs > 0.01f) {
                            emit[1] = true;
                            for (int c = 0; c < 3; c++) { dir[1][c] = r[c]; org[1][c] = P[c] + r[c] * kEps; wgt[1][c] = w0[c] * mt[6 + c] * Rs; }
                        }
                        // Ray::refract (Ray.h:202-243)
                        const float dn = (d[0] * nn[0] + d[1] * nn[1]) + d[2] * nn[2];
                        const float energy = (float)(1 - (((double)n1 * (double)n1) * (1 - (double)dn * (double)dn) / ((double)n2 * (double)n2)));
                        emit[2] = true;
                        if (energy < 0) {
                            for (int c = 0; c < 3; c++) { dir[2][c] = r[c]; org[2][c] = P[c] + r[c] * kEps; }
                        } else {
                            const float inv_n2 = 1.0f / n2, se = sqrtf(energy);
                            for (int c = 0; c < 3; c++) {
                                const float t = ((d[c] - nn[c] * dn) * n1) * inv_n2;
                                dir[2][c] = t - nn[c] * se;
                                org[2][c] = P[c] + dir[2][c] * kEps;
                            }
                        }
                        for (int c = 0; c < 3; c++) wgt[2][c] = w0[c] * mt[6 + c] * (1.f - Rs);
                    }
                }
            }
        }
        // wave64 compaction: one ballot per child kind; the workgroup's waves share one atomic
        const unsigned long long m0 = __ballot(emit[0]), m1 = __ballot(emit[1]), m2 = __ballot(emit[2]);
        const int c0 = __popcll(m0), c1 = __popcll(m1), c2 = __popcll(m2);
        const unsigned long long base = workgroup_reserve((unsigned)(c0 + c1 + c2), a.count);
        const unsigned long long lt = (1ull << lane) - 1ull;
        const unsigned long long slot[3] = {base + __popcll(m0 & lt), base + c0 + __popcll(m1 & lt), base + c0 + c1 + __popcll(m2 & lt)};
        for (int j = 0; j < 3; j++) {
            if (!emit[j]) continue;
            const unsigned long long s = slot[j];
            reinterpret_cast<float4 *>(a.out_rays)[2 * s] = make_float4(org[j][0], org[j][1], org[j][2], 0.0f);
            reinterpret_cast<float4 *>(a.out_rays)[2 * s + 1] = make_float4(dir[j][0], dir[j][1], dir[j][2], 1e12f);
            a.out_weights[3 * s] = wgt[j][0]; a.out_weights[3 * s + 1] = wgt[j][1]; a.out_weights[3 * s + 2] = wgt[j][2];
            a.out_pixels[s] = pix;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The PATH_TRACING build of the secondary-ray generators (Ray.h:149-158, 235-239) and Ray::random (Ray.h:124-140):
// every child direction is drawn from a lobe -- alignHemisphereToVector (Utility.h:34-50) around the mirror direction
// / the refracted direction with phi = acos(pow(u1, 1/(1+shininess))), or around the normal with phi = asin(sqrt(u1))
// (cosine-weighted) for the diffuse bounce -- theta = 2 pi u2.  The reference draws u1, u2 from rand(); here they come
// from the counter-based generator that jitters the eye rays, keyed by (seed, ray id, bounce, child kind), integer-exact
// on the device and in the oracle; the transcendentals are miro_math.h on both sides, so the ray sets are the same bits.
// Children: 0 mirror reflection (weight x ks), 1 Fresnel reflection (x kt Rs, if Rs > 0.01), 2 refraction or its total
// internal reflection (x kt (1-Rs)) as Scene.cpp:302-336; 3 the diffuse bounce of Ray::random (x kd) -- an EXTENSION:
// the reference defines Ray::random but traceScene at HEAD never calls it.
// ---------------------------------------------------------------------------------------------------
struct PathArgs {
    MeshMat m;
    const mr_ray *rays;
    const mr_hit *hits;
    const float *weights;
    const uint32_t *pixels;
    const uint32_t *ids;          // stable id per ray (NULL: the ray's index); children get ids derived from it
    uint32_t spp, hbase, bounce, kinds;
    unsigned long long n;
    mr_ray *out_rays;
    float *out_weights;
    uint32_t *out_pixels, *out_ids;
    unsigned long long *count;
};

__device__ __forceinline__ uint32_t pcg32(uint32_t x) {
    const uint32_t state = x * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
__device__ __forceinline__ float unit01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// Ray::alignToVector (Ray.h:86-91): direction = alignHemisphereToVector(v, theta, phi), origin = P + epsilon * direction
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

__global__ __launch_bounds__(kBlock) void path_rays_kernel(PathArgs a) {
    const int lane = threadIdx.x & 63;
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    const unsigned long long n_round = (a.n + (unsigned long long)kBlock - 1ull) / kBlock * kBlock;     // whole workgroups
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < n_round; k += stride) {
        bool emit[4] = {false, false, false, false};
        float dir[4][3], org[4][3], wgt[4][3];
        uint32_t pix = 0, id = 0;
        if (k < a.n) {
            const float4 h = reinterpret_cast<const float4 *>(a.hits)[k];
            const uint32_t prim = __float_as_uint(h.y);
            if (prim != MR_MISS) {
                const float *mt = material_of(a.m, prim);
                const bool refl = any_pos(mt + 3) && (a.kinds & 1u), refr = any_pos(mt + 6) && (a.kinds & 2u);
                const bool diff = any_pos(mt) && (a.kinds & 4u);
                if (refl || refr || diff) {
                    float P[3], N[3];
                    surface_point(a.m, a.rays, k, h, P, N);
                    const float4 rb = reinterpret_cast<const float4 *>(a.rays)[2 * k + 1];
                    const float d[3] = {rb.x, rb.y, rb.z};
                    float w0[3] = {1.f, 1.f, 1.f};
                    if (a.weights) { w0[0] = a.weights[3 * k]; w0[1] = a.weights[3 * k + 1]; w0[2] = a.weights[3 * k + 2]; }
                    pix = a.pixels ? a.pixels[k] : (uint32_t)(k / a.spp);
                    id = a.ids ? a.ids[k] : (uint32_t)k;
                    const uint32_t hray = pcg32(a.hbase ^ id) + a.bounce * 4u;
                    const float lobe_exp = 1.0f / (1.0f + mt[9]);
                    // Ray::reflect under PATH_TRACING (Ray.h:149-158): a fresh draw per call
                    auto reflect_pt = [&](uint32_t kind, float o_out[3], float d_out[3]) {
                        const uint32_t hk = pcg32(hray + kind);
                        const float phi = mm_acosf01(mm_powf01(unit01(pcg32(hk)), lobe_exp));
                        const float theta = (2.0f * kPI) * unit01(pcg32(hk ^ 0x68bc21ebu));
                        const float two = 2 * ((N[0] * d[0] + N[1] * d[1]) + N[2] * d[2]);
                        const float dr[3] = {d[0] - two * N[0], d[1] - two * N[1], d[2] - two * N[2]};
                        align_to_vector(dr, P, theta, phi, o_out, d_out);
                    };
                    if (refl) {                                                       // Scene.cpp:302-312
                        emit[0] = true;
                        reflect_pt(0u, org[0], dir[0]);
                        for (int c = 0; c < 3; c++) wgt[0][c] = w0[c] * mt[3 + c];
                    }
                    if (refr) {                                                       // Scene.cpp:315-336
                        const float index = mt[10];
                        const float dN = (d[0] * N[0] + d[1] * N[1]) + d[2] * N[2];
                        const bool enter = dN < 0;
                        const float n1 = enter ? 1.0f : index, n2 = enter ? index : 1.0f;
                        const float nn[3] = {enter ? N[0] : -N[0], enter ? N[1] : -N[1], enter ? N[2] : -N[2]};
                        // Ray::getReflectionCoefficient (Ray.h:168-199) on the shared transcendentals
                        const float cosT = (-d[0] * nn[0] + -d[1] * nn[1]) + -d[2] * nn[2];
                        const float sinT = mm_sinf(mm_acosf(cosT));
                        const float q = (n1 / n2) * sinT, p = q * q;                  // powf(x, 2.f)
                        float Rs = 1.0f;
                        if (!(p > 1.f)) {
                            const float sq = sqrtf(1.f - p), fr = (n1 * cosT - sq) / (n1 * cosT + sq);
                            Rs = fr * fr;
                        }
                        if (Rs > 0.01f) {
                            emit[1] = true;
                            reflect_pt(1u, org[1], dir[1]);
                            for (int c = 0; c < 3; c++) wgt[1][c] = w0[c] * mt[6 + c] * Rs;
                        }
                        // Ray::refract (Ray.h:202-243)
                        const float dn = (d[0] * nn[0] + d[1] * nn[1]) + d[2] * nn[2];
                        const float energy = (float)(1 - (((double)n1 * (double)n1) * (1 - (double)dn * (double)dn) / ((double)n2 * (double)n2)));
                        emit[2] = true;
                        if (energy < 0) {
                            reflect_pt(2u, org[2], dir[2]);
                        } else {
                            const float inv_n2 = 1.0f / n2, se = sqrtf(energy);
                            float dr[3];
                            for (int c = 0; c < 3; c++) dr[c] = ((d[c] - nn[c] * dn) * n1) * inv_n2 - nn[c] * se;
                            const uint32_t hk = pcg32(hray + 2u);
                            const float phi = mm_acosf01(mm_powf01(unit01(pcg32(hk)), lobe_exp));
                            const float theta = (2.0f * kPI) * unit01(pcg32(hk ^ 0x68bc21ebu));
                            align_to_vector(dr, P, theta, phi, org[2], dir[2]);
                        }
                        for (int c = 0; c < 3; c++) wgt[2][c] = w0[c] * mt[6 + c] * (1.f - Rs);
                    }
                    if (diff) {                                                       // Ray::random (Ray.h:124-140)
                        emit[3] = true;
                        const uint32_t hk = pcg32(hray + 3u);
                        const float phi = mm_asinf01(sqrtf(unit01(pcg32(hk))));
                        const float theta = (2.0f * kPI) * unit01(pcg32(hk ^ 0x68bc21ebu));
                        align_to_vector(N, P, theta, phi, org[3], dir[3]);
                        for (int c = 0; c < 3; c++) wgt[3][c] = w0[c] * mt[c];
                    }
                }
            }
        }
        // wave64 compaction: one ballot per child kind; the workgroup's waves share one atomic
        unsigned long long mk[4];
        int cn[4], tot = 0;
        for (int j = 0; j < 4; j++) { mk[j] = __ballot(emit[j]); cn[j] = __popcll(mk[j]); tot += cn[j]; }
        const unsigned long long base = workgroup_reserve((unsigned)tot, a.count);
        const unsigned long long lt = (1ull << lane) - 1ull;
        unsigned long long before = base;
        for (int j = 0; j < 4; j++) {
            if (emit[j]) {
                const unsigned long long s = before + __popcll(mk[j] & lt);
                reinterpret_cast<float4 *>(a.out_rays)[2 * s] = make_float4(org[j][0], org[j][1], org[j][2], 0.0f);
                reinterpret_cast<float4 *>(a.out_rays)[2 * s + 1] = make_float4(dir[j][0], dir[j][1], dir[j][2], 1e12f);
                a.out_weights[3 * s] = wgt[j][0]; a.out_weights[3 * s + 1] = wgt[j][1]; a.out_weights[3 * s + 2] = wgt[j][2];
                a.out_pixels[s] = pix;
                if (a.out_ids) a.out_ids[s] = pcg32(id ^ (0x9e3779b9u * (uint32_t)(j + 1)));
            }
            before += cn[j];
        }
    }
}

inline unsigned grid_for(unsigned long long n) {
    unsigned long long blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 256ull * 32ull) blocks = 256ull * 32ull;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

MeshMat mesh_of(const DeviceScene &ds) {
    MeshMat m;
    m.s = surface_ptrs(ds); m.mats = ds.materials; m.prim_mat = ds.prim_material;
    return m;
}

}  // namespace

mr_status launch_shade_accumulate(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, const float *d_weights,
                                  const uint32_t *d_pixels, unsigned long long n, const mr_ray *d_shadow_rays,
                                  const mr_hit *d_shadow_hits, const uint32_t *d_shadow_src,
                                  const unsigned long long *d_shadow_count, float *d_light_scale, const mr_light &light,
                                  uint32_t spp, float *d_rgb, hipStream_t stream) {
    if (n == 0) return MR_OK;
    const MeshMat m = mesh_of(ds);
    hipLaunchKernelGGL(light_scale_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, m, d_shadow_rays, d_shadow_hits,
                       d_shadow_src, d_shadow_count, n, d_light_scale);
    MR_HIP_CHECK(hipGetLastError());
    AccumArgs a;
    a.m = m; a.rays = d_rays; a.hits = d_hits; a.weights = d_weights; a.pixels = d_pixels; a.light_scale = d_light_scale;
    for (int c = 0; c < 3; c++) { a.L[c] = light.position[c]; a.color[c] = light.color[c]; }
    a.wattage = light.wattage; a.spp = spp; a.inv_spp = 1.0f / (float)spp; a.n = n; a.rgb = d_rgb;
    hipLaunchKernelGGL(shade_accumulate_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, a);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

mr_status launch_secondary_rays(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, const float *d_weights,
                                const uint32_t *d_pixels, unsigned long long n, uint32_t spp, mr_ray *d_out_rays,
                                float *d_out_weights, uint32_t *d_out_pixels, unsigned long long *d_count, hipStream_t stream) {
    MR_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), stream));
    if (n == 0) return MR_OK;
    BounceArgs a;
    a.m = mesh_of(ds); a.rays = d_rays; a.hits = d_hits; a.weights = d_weights; a.pixels = d_pixels; a.spp = spp; a.n = n;
    a.out_rays = d_out_rays; a.out_weights = d_out_weights; a.out_pixels = d_out_pixels; a.count = d_count;
    hipLaunchKernelGGL(secondary_rays_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, a);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

mr_status launch_path_rays(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, const float *d_weights,
                           const uint32_t *d_pixels, const uint32_t *d_ids, unsigned long long n, uint32_t spp, uint32_t seed,
                           uint32_t bounce, uint32_t kinds, mr_ray *d_out_rays, float *d_out_weights, uint32_t *d_out_pixels,
                           uint32_t *d_out_ids, unsigned long long *d_count, hipStream_t stream) {
    MR_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), stream));
    if (n == 0) return MR_OK;
    PathArgs a;
    a.m = mesh_of(ds); a.rays = d_rays; a.hits = d_hits; a.weights = d_weights; a.pixels = d_pixels; a.ids = d_ids;
    a.spp = spp; a.bounce = bounce; a.kinds = kinds; a.n = n;
    {   // host copy of pcg32
        const uint32_t state = seed * 747796405u + 2891336453u;
        const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
        a.hbase = (word >> 22u) ^ word;
    }
    a.out_rays = d_out_rays; a.out_weights = d_out_weights; a.out_pixels = d_out_pixels; a.out_ids = d_out_ids; a.count = d_count;
    hipLaunchKernelGGL(path_rays_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, a);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

}  // namespace mr
