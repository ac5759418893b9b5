// mr_shade.hip -- the caller of the shadow batch, on the device ("next" row f2 of SURVEY.md section 8):
//   occlusion scatter   shadow hits -> one flag per primary ray (Phong.cpp:97-100, opaque occluders)
//   shade_pixels        Phong::shade for one point light and a uniform Lambert/Phong material
//                       (Phong.cpp:44-160), Scene::trace's normal normalisation (Scene.cpp:262),
//                       the miss colour (Scene.cpp:340,685) and the per-pixel sample average
//                       (Scene.cpp:126-139) into a linear float RGB buffer (tempImage, Scene.cpp:106)
//   tonemap             sigmoid(6v-3) and the 8-bit mapping (Scene.cpp:87-91,177-202; Image.cpp:44-50)
// Floating point here is tolerance-parity (powf, expf differ from libm in the last ulps); the hot
// path's bit-exactness is unaffected.
#include <hip/hip_runtime.h>

#include "mr_internal.h"
#include "mr_phong.h"
#include "mr_surface.h"
#include "mr_tile.h"

namespace mr {
namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void occlusion_scatter_kernel(const mr_hit *shadow_hits, const uint32_t *src,
                                                                   const unsigned long long *count,
                                                                   unsigned long long max_n, uint8_t *occluded) {
    unsigned long long n = *count;
    if (n > max_n) n = max_n;
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < n; k += stride)
        occluded[src[k]] = shadow_hits[k].prim != MR_MISS ? 1 : 0;
}

struct ShadeArgs {
    SurfacePtrs m;
    const mr_ray *rays;
    const mr_hit *hits;
    const uint8_t *occluded;
    DirectLight lt;
    uint32_t spp;
    unsigned long long n_pixels;
    float *rgb;
};

__device__ __forceinline__ void shade_sample(const ShadeArgs &a, unsigned long long k, float out[3]) {
    const float4 h = reinterpret_cast<const float4 *>(a.hits)[k];
    const uint32_t prim = __float_as_uint(h.y);
    if (prim == MR_MISS) { out[0] = a.lt.bg[0]; out[1] = a.lt.bg[1]; out[2] = a.lt.bg[2]; return; }   // Scene.cpp:340
    out[0] = out[1] = out[2] = 0.0f;
    if (a.occluded[k]) return;                                                              // Phong.cpp:97-100
    float P[3], N[3];
    surface<true>(a.m, a.rays, k, h.x, prim, h.z, h.w, P, N);                               // HitInfo::P, ::N
    const float4 rb = reinterpret_cast<const float4 *>(a.rays)[2 * k + 1];
    phong_direct(a.lt, P, N, rb.x, rb.y, rb.z, out);
}

// one thread per pixel, samples summed in order (bitwise reproducible, independent of the grid)
__global__ __launch_bounds__(kBlock) void shade_pixels_kernel(ShadeArgs a) {
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    const float inv_spp = 1.0f / (float)a.spp;
    for (unsigned long long pix = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; pix < a.n_pixels; pix += stride) {
        float acc[3] = {0.f, 0.f, 0.f};
        for (uint32_t s = 0; s < a.spp; s++) {
            float c[3];
            shade_sample(a, pix * a.spp + s, c);
            acc[0] += c[0]; acc[1] += c[1]; acc[2] += c[2];
        }
        if (a.spp > 1) { acc[0] *= inv_spp; acc[1] *= inv_spp; acc[2] *= inv_spp; }          // Scene.cpp:139
        a.rgb[3 * pix] = acc[0]; a.rgb[3 * pix + 1] = acc[1]; a.rgb[3 * pix + 2] = acc[2];
    }
}

// one lane per sample (coalesced 16-byte hit / ray loads), then an xor-butterfly over the spp lanes of a pixel:
// spp must be a power of two <= 64, so that a pixel's samples never straddle a wave.  The summation tree depends on
// the sample index only, so the result is independent of how the frame is sharded.
__global__ __launch_bounds__(kBlock) void shade_samples_kernel(ShadeArgs a, unsigned long long n_samples) {
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    const float inv_spp = 1.0f / (float)a.spp;
    const unsigned long long n_round = (n_samples + 63ull) & ~63ull;
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < n_round; k += stride) {
        float c[3] = {0.f, 0.f, 0.f};
        if (k < n_samples) shade_sample(a, k, c);
        for (uint32_t off = 1; off < a.spp; off <<= 1) {
            c[0] += __shfl_xor(c[0], (int)off, 64);
            c[1] += __shfl_xor(c[1], (int)off, 64);
            c[2] += __shfl_xor(c[2], (int)off, 64);
        }
        if (k < n_samples && (k & (a.spp - 1)) == 0) {
            const unsigned long long pix = k / a.spp;
            if (a.spp > 1) { c[0] *= inv_spp; c[1] *= inv_spp; c[2] *= inv_spp; }
            a.rgb[3 * pix] = c[0]; a.rgb[3 * pix + 1] = c[1]; a.rgb[3 * pix + 2] = c[2];
        }
    }
}

// Queries of Scene::traceScene's photon-map term (Scene.cpp:285-292): hit point and the normalised normal of every
// hit whose material is diffuse (Phong::isDiffuse, Phong.cpp:39-42); a NaN normal marks "no query".
__global__ __launch_bounds__(kBlock) void gather_queries_kernel(SurfacePtrs m, const float *mats, const uint32_t *prim_mat,
                                                                const mr_ray *rays, const mr_hit *hits, unsigned long long n,
                                                                float *pos, float *nrm) {
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < n; k += stride) {
        const float4 h = reinterpret_cast<const float4 *>(hits)[k];
        const uint32_t prim = __float_as_uint(h.y);
        float P[3] = {0.f, 0.f, 0.f}, N[3];
        N[0] = N[1] = N[2] = __uint_as_float(0x7fc00000u);
        if (prim != MR_MISS) {
            const float *mt = mats + 11 * (size_t)material_id(m, prim_mat, prim);
            if (mt[0] > 0.f || mt[1] > 0.f || mt[2] > 0.f) {
                surface<true>(m, rays, k, h.x, prim, h.z, h.w, P, N);
                const float inv = 1.0f / sqrtf((N[0] * N[0] + N[1] * N[1]) + N[2] * N[2]);   // Scene.cpp:262
                N[0] *= inv; N[1] *= inv; N[2] *= inv;
            }
        }
        for (int c = 0; c < 3; c++) { pos[3 * k + c] = P[c]; nrm[3 * k + c] = N[c]; }
    }
}

// shadeResult += irradiance + caustic (Scene.cpp:298), averaged over the pixel's samples like the direct term
__global__ __launch_bounds__(kBlock) void gather_accumulate_kernel(const float *irr_a, const float *irr_b, unsigned long long n_pixels,
                                                                   uint32_t spp, float *rgb) {
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    const float inv = 1.0f / (float)spp;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < 3 * n_pixels; i += stride) {
        const unsigned long long pix = i / 3, c = i % 3;
        float acc = 0.0f;
        for (uint32_t s = 0; s < spp; s++) {
            const unsigned long long k = 3 * (pix * spp + s) + c;
            float v = 0.0f;
            if (irr_a) v = irr_a[k];
            if (irr_b) v = irr_a ? v + irr_b[k] : irr_b[k];
            acc += v;
        }
        rgb[i] += spp > 1 ? acc * inv : acc;
    }
}

__global__ __launch_bounds__(kBlock) void tonemap_kernel(const float *rgb, unsigned long long n_values, uint8_t *out) {
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < n_values; i += stride) {
        float v = rgb[i];
        v = 1.0f / (1.0f + expf(-(6.0f * v - 3.0f)));               // sigmoid(6v-3), Scene.cpp:89, Utility.h:19-22
        const float m = 255.0f * v;                                 // Image.cpp:44-50
        out[i] = m > 255.0f ? 255 : (uint8_t)m;
    }
}

// pixel slots of a tiled window (mr_gen_eye_rays_tiled) -> image order: `channels` floats per pixel
__global__ __launch_bounds__(kBlock) void untile_kernel(const float *slots, float *image, uint32_t W, uint32_t rows, TileShape t,
                                                        uint32_t channels) {
    const unsigned long long n = (unsigned long long)W * rows * channels;
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t slot = (uint32_t)(i / channels), c = (uint32_t)(i - (unsigned long long)slot * channels);
        uint32_t x, y;
        tile_decode(slot, W, rows, t, x, y);
        image[((unsigned long long)y * W + x) * channels + c] = slots[i];
    }
}

// image row y <- row `local` of rank `rank`'s shard (frame.band_rows rule): one thread per float
__global__ __launch_bounds__(kBlock) void deinterleave_kernel(const float *recv, float *full, uint32_t W, uint32_t H, uint32_t band_rows,
                                                              uint32_t world, uint32_t shard_rows, uint32_t fpp) {
    const unsigned long long row_floats = (unsigned long long)W * fpp, n = row_floats * H;
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t y = (uint32_t)(i / row_floats);
        const unsigned long long x = i - (unsigned long long)y * row_floats;
        const uint32_t band = y / band_rows, rank = band % world, local = (band / world) * band_rows + y % band_rows;
        full[i] = recv[((unsigned long long)rank * shard_rows + local) * row_floats + x];
    }
}

inline unsigned grid_for(unsigned long long n) {
    unsigned long long blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 256ull * 32ull) blocks = 256ull * 32ull;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

mr_status launch_shade(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, unsigned long long n,
                       const mr_hit *d_shadow_hits, const uint32_t *d_shadow_src, const unsigned long long *d_shadow_count,
                       uint8_t *d_occluded, const mr_light &light, const float diffuse[3], uint32_t spp, float *d_rgb,
                       hipStream_t stream) {
    if (n == 0) return MR_OK;
    hipLaunchKernelGGL(occlusion_scatter_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, d_shadow_hits, d_shadow_src,
                       d_shadow_count, n, d_occluded);
    MR_HIP_CHECK(hipGetLastError());
    ShadeArgs a;
    a.m = surface_ptrs(ds);
    a.rays = d_rays; a.hits = d_hits; a.occluded = d_occluded;
    for (int c = 0; c < 3; c++) { a.lt.L[c] = light.position[c]; a.lt.color[c] = light.color[c]; a.lt.diffuse[c] = diffuse[c]; a.lt.bg[c] = 0.0f; }
    a.lt.wattage = light.wattage;
    a.spp = spp;
    a.n_pixels = n / spp;
    a.rgb = d_rgb;
    if (spp <= 64 && (spp & (spp - 1)) == 0)
        hipLaunchKernelGGL(shade_samples_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, a, n);
    else
        hipLaunchKernelGGL(shade_pixels_kernel, dim3(grid_for(a.n_pixels)), dim3(kBlock), 0, stream, a);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

mr_status launch_gather_queries(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, unsigned long long n,
                                float *d_pos, float *d_nrm, hipStream_t stream) {
    if (n == 0) return MR_OK;
    hipLaunchKernelGGL(gather_queries_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, surface_ptrs(ds), ds.materials,
                       ds.prim_material, d_rays, d_hits, n, d_pos, d_nrm);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

mr_status launch_gather_accumulate(const float *d_irr_a, const float *d_irr_b, unsigned long long n, uint32_t spp,
                                   float *d_rgb, hipStream_t stream) {
    if (n == 0 || (!d_irr_a && !d_irr_b)) return MR_OK;
    hipLaunchKernelGGL(gather_accumulate_kernel, dim3(grid_for(3 * (n / spp))), dim3(kBlock), 0, stream, d_irr_a, d_irr_b,
                       n / spp, spp, d_rgb);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

mr_status launch_tonemap(const float *d_rgb, unsigned long long n_values, uint8_t *d_out, hipStream_t stream) {
    if (n_values == 0) return MR_OK;
    hipLaunchKernelGGL(tonemap_kernel, dim3(grid_for(n_values)), dim3(kBlock), 0, stream, d_rgb, n_values, d_out);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

mr_status launch_untile(const float *d_slots, float *d_image, uint32_t W, uint32_t rows, uint32_t spp, uint32_t channels,
                        hipStream_t stream) {
    const unsigned long long n = (unsigned long long)W * rows * channels;
    if (n == 0) return MR_OK;
    hipLaunchKernelGGL(untile_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, d_slots, d_image, W, rows, tile_shape(spp), channels);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

mr_status launch_deinterleave(const float *d_recv, float *d_full, uint32_t W, uint32_t H, uint32_t band_rows, uint32_t world,
                              uint32_t shard_rows, uint32_t fpp, hipStream_t stream) {
    const unsigned long long n = (unsigned long long)W * H * fpp;
    if (n == 0) return MR_OK;
    hipLaunchKernelGGL(deinterleave_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, d_recv, d_full, W, H, band_rows, world,
                       shard_rows, fpp);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

}  // namespace mr
