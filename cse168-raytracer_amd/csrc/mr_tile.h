// mr_tile.h -- the tiled order of a frame's rays (mr_gen_eye_rays_tiled).
//
// Camera rays in image order put the 64 rays of a wave on a 64 x 1 pixel strip at 1 sample per pixel; the rays of a
// square tile share far more of their BVH path.  The tiled order keeps a pixel's samples consecutive (slot p holds rays
// p*spp .. p*spp+spp-1) and permutes the pixels: rows in groups of `th`, each group cut into blocks `tw` pixels wide,
// a block stored row by row, th * tw * spp = 64 whenever the window allows -- one block is one wave.  Ragged windows
// simply get shorter last groups / narrower last blocks.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define MR_HD __host__ __device__ __forceinline__
#else
#define MR_HD inline
#endif

namespace mr {

struct TileShape { uint32_t th, tw; };

// 64 / spp pixels per wave as the squarest th x tw; spp that is not a power of two (or > 64): image order
MR_HD TileShape tile_shape(uint32_t spp) {
    TileShape t = {1u, 1u};
    if (spp == 0 || spp > 64 || (spp & (spp - 1))) return t;
    uint32_t pixels = 64u / spp;            // 64, 32, 16, 8, 4, 2, 1
    while (pixels > 1) {                    // hand the factors of two to th and tw in turn, th first
        t.th *= 2; pixels /= 2;
        if (pixels > 1) { t.tw *= 2; pixels /= 2; }
    }
    return t;
}

// pixel slot -> (x, row inside the window) for a window of W x rows pixels
MR_HD void tile_decode(uint32_t slot, uint32_t W, uint32_t rows, TileShape t, uint32_t &x, uint32_t &y_local) {
    const uint32_t per_group = t.th * W;
    const uint32_t g = slot / per_group, pp = slot - g * per_group;
    const uint32_t left = rows - g * t.th, h = left < t.th ? left : t.th;        // rows in this group
    const uint32_t per_block = h * t.tw;
    const uint32_t xb = pp / per_block, pb = pp - xb * per_block;
    const uint32_t wl = W - xb * t.tw, w = wl < t.tw ? wl : t.tw;                // pixels across in this block
    const uint32_t r = pb / w, xi = pb - r * w;
    x = xb * t.tw + xi;
    y_local = g * t.th + r;
}

}  // namespace mr
