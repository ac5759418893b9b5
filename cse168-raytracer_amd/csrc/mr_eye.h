// mr_eye.h -- Camera::eyeRay (Camera.cpp:104-161) for the kernels that generate primary rays: the batched generator
// (eye_rays_kernel, mr_kernels.hip) and the fused frame kernel (mr_frame.hip) share one definition, so their rays
// are the same bits.  The camera frame is computed on the host exactly as the reference does (Camera.h:79-110,
// Camera.cpp:113-124); the per-sample arithmetic keeps the reference's operation order.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

#include "mr_internal.h"
#include "mr_tile.h"

namespace mr {

struct EyeFrame {
    float eye[3], u[3], v[3], w[3];
    float left, right, bottom, top;
    uint32_t W, H, y0, spp, jitter, hbase;
    uint32_t tiled, rows;        // tiled: ray order of mr_tile.h inside the window of `rows` rows
    // rows of the window in the image: contiguous from y0 (band_world == 1), or the interleaved bands of one rank of a
    // multi-GPU frame -- window row j is image row ((j / band_rows) * band_world + band_rank) * band_rows + j % band_rows
    uint32_t band_rows, band_rank, band_world;
    TileShape tile;
    unsigned long long n;        // samples in the window = rows * W * spp
};

// y0/y1: contiguous window.  For banded windows call set_bands() afterwards.
inline EyeFrame make_eye_frame(const mr_camera &cam, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1, uint32_t spp,
                               uint32_t jitter, uint32_t seed, bool tiled) {
    auto unit3 = [](float *a) {
        const float len = sqrtf((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
        const float inv = 1.0f / len;
        a[0] *= inv; a[1] *= inv; a[2] *= inv;
    };
    auto cross3 = [](const float *a, const float *b, float *o) {
        o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
    };
    EyeFrame f;
    float up[3] = {cam.up[0], cam.up[1], cam.up[2]};
    unit3(up);
    float view[3] = {cam.lookat[0] - cam.eye[0], cam.lookat[1] - cam.eye[1], cam.lookat[2] - cam.eye[2]};
    unit3(view);
    f.w[0] = -view[0]; f.w[1] = -view[1]; f.w[2] = -view[2];
    unit3(f.w);
    cross3(up, f.w, f.u);
    unit3(f.u);
    cross3(f.w, f.u, f.v);
    const float PI = 3.1415926535897932384626433832795028841972f;
    const float DegToRad = PI / 180.0f, HalfDegToRad = DegToRad / 2.0f;
    const float aspect = (float)W / (float)H;
    f.top = tanf(cam.fov_deg * HalfDegToRad);
    f.right = aspect * f.top; f.bottom = -f.top; f.left = -f.right;
    f.eye[0] = cam.eye[0]; f.eye[1] = cam.eye[1]; f.eye[2] = cam.eye[2];
    f.W = W; f.H = H; f.y0 = y0; f.spp = spp; f.jitter = jitter;
    f.tile = tile_shape(spp);
    f.rows = y1 - y0;
    f.band_rows = f.rows ? f.rows : 1; f.band_rank = 0; f.band_world = 1;
    f.tiled = tiled && (f.tile.th > 1 || f.tile.tw > 1) ? 1u : 0u;
    {   // host copy of pcg_hash
        uint32_t state = seed * 747796405u + 2891336453u;
        uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
        f.hbase = (word >> 22u) ^ word;
    }
    f.n = (unsigned long long)(y1 - y0) * W * spp;
    return f;
}

#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t eye_pcg(uint32_t x) {
    const uint32_t state = x * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

// sample k of the window -> pixel (x, window row, image row y) and sample number
__device__ __forceinline__ void eye_sample_of(const EyeFrame &f, unsigned long long k, uint32_t &x, uint32_t &row, uint32_t &y,
                                              uint32_t &sm) {
    const unsigned long long pix_local = k / f.spp;
    sm = (uint32_t)(k - pix_local * f.spp);
    row = (uint32_t)(pix_local / f.W);
    x = (uint32_t)(pix_local % f.W);
    if (f.tiled) tile_decode((uint32_t)pix_local, f.W, f.rows, f.tile, x, row);
    y = f.band_world == 1 ? f.y0 + row
                          : ((row / f.band_rows) * f.band_world + f.band_rank) * f.band_rows + row % f.band_rows;
}

// Camera::eyeRay for sample `sm` of pixel (x, y): a = (eye, tMin = 0), b = (direction, tMax = MIRO_TMAX)
__device__ __forceinline__ void eye_ray_of(const EyeFrame &f, uint32_t x, uint32_t y, uint32_t sm, float4 &a, float4 &b) {
    float dx = 0.5f, dy = 0.5f;
    if (f.jitter) {
        const uint32_t pix = y * f.W + x;
        const uint32_t h = eye_pcg(eye_pcg(f.hbase ^ pix) + sm);
        dx = (float)(eye_pcg(h) >> 8) * (1.0f / 16777216.0f);
        dy = (float)(eye_pcg(h ^ 0x68bc21ebu) >> 8) * (1.0f / 16777216.0f);
    }
    const float up = f.left + (f.right - f.left) * (((float)x + dx) / (float)f.W);
    const float vp = f.bottom + (f.top - f.bottom) * (((float)y + dy) / (float)f.H);
    float ddx = (up * f.u[0] + vp * f.v[0]) - f.w[0];
    float ddy = (up * f.u[1] + vp * f.v[1]) - f.w[1];
    float ddz = (up * f.u[2] + vp * f.v[2]) - f.w[2];
    const float len = sqrtf((ddx * ddx + ddy * ddy) + ddz * ddz);
    const float inv = 1.0f / len;
    a = make_float4(f.eye[0], f.eye[1], f.eye[2], 0.0f);
    b = make_float4(ddx * inv, ddy * inv, ddz * inv, 1e12f);     // MIRO_TMAX
}
#endif

}  // namespace mr
