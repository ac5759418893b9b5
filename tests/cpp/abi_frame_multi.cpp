// One frame on N GPUs through the C ABI alone: ONE process, one mr_scene replica per device,
// the image rows dealt to the devices in interleaved bands, every device renders its bands with ONE launch
// (mr_render_direct: eye rays -> trace -> shadow rays -> trace -> Phong shade, no ray buffers), and the frame ends with
// the SINGLE gather of the shards to device 0 over RCCL (ncclGather inside one group, communicators from
// ncclCommInitAll) and the de-interleave on that device (mr_deinterleave_bands).  No Python, no torch: g++,
// include/miro_hip.h, libmiro_hip.so, the HIP runtime and librccl.
//
// mode "rgb":  gathers the float framebuffer (12 B per pixel), tone-maps it and writes a binary PPM -- with one device the
//              file equals abi_frame's byte for byte (tests/test_shim.py).
// mode "hits": gathers the primary mr_hit records instead (16 B per sample, samples in image order) and writes them raw:
//              the hit-buffer parity mode of SURVEY.md section 8e.
//
// A fourth word "virtual" rehearses the N-rank flow on ONE device: every rank's replica, launch and shard live on device 0 and
// the gather is N device-to-device copies instead of the RCCL call -- everything but the collective itself (band arithmetic,
// per-rank launches, shard layout, de-interleave) runs exactly as with N devices, which is how the 1-GPU test boxes check it.
//
// usage: abi_frame_multi <model.obj> <floor 9 floats | -> <W> <H> <spp> <eye xyz> <lookat xyz> <fov> <light xyz> <wattage>
//                        <out file> <devices> [band rows = 0: auto] [rgb | hits] [virtual]
// Prints "rays <primary> <shadow>" (totals over the devices) and "band <rows> devices <n>".
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "miro_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 10; } } while (0)
#define MR_OK_(x) do { if ((x) != MR_OK) { fprintf(stderr, "%s: %s\n", #x, mr_last_error()); return 11; } } while (0)
#define NCCL_OK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 12; } } while (0)

static bool parse3(const char *s, float v[3]) { return sscanf(s, "%f,%f,%f", v, v + 1, v + 2) == 3; }

// the largest band height <= 8 that gives every device the same number of bands (bench.py's rule), else 8
static uint32_t auto_band(uint32_t H, uint32_t world) {
    for (uint32_t b = 8; b >= 1; b--)
        if (H % b == 0 && (H / b) % world == 0) return b;
    return 8;
}

int main(int argc, char **argv) {
    if (argc < 13) { fprintf(stderr, "usage: see the head of tests/cpp/abi_frame_multi.cpp\n"); return 2; }
    const char *model = argv[1], *floor_s = argv[2];
    const uint32_t W = (uint32_t)atoi(argv[3]), H = (uint32_t)atoi(argv[4]), spp = (uint32_t)atoi(argv[5]);
    mr_frame_desc fd;
    memset(&fd, 0, sizeof(fd));
    if (!parse3(argv[6], fd.camera.eye) || !parse3(argv[7], fd.camera.lookat) || !parse3(argv[9], fd.light.position)) return 3;
    fd.camera.up[0] = 0; fd.camera.up[1] = 1; fd.camera.up[2] = 0;
    fd.camera.fov_deg = (float)atof(argv[8]);
    fd.light.color[0] = fd.light.color[1] = fd.light.color[2] = 1.0f;
    fd.light.wattage = (float)atof(argv[10]);
    fd.diffuse[0] = fd.diffuse[1] = fd.diffuse[2] = 1.0f;
    const char *out_path = argv[11];
    const int world = atoi(argv[12]);
    uint32_t band = argc > 13 ? (uint32_t)atoi(argv[13]) : 0;
    const bool hits_mode = argc > 14 && strcmp(argv[14], "hits") == 0;
    const bool virtual_ranks = argc > 15 && strcmp(argv[15], "virtual") == 0;
    if (world < 1 || W == 0 || H == 0 || spp == 0) return 3;
    if (band == 0) band = auto_band(H, (uint32_t)world);
    int have = 0;
    HIP_OK(hipGetDeviceCount(&have));
    if (have < (virtual_ranks ? 1 : world)) { fprintf(stderr, "%d devices asked for, %d present\n", world, have); return 6; }

    fd.W = W; fd.H = H; fd.spp = spp; fd.jitter = spp > 1; fd.seed = 168;
    fd.band_rows = band; fd.band_world = (uint32_t)world;
    fd.tiled = hits_mode ? 0 : 1;                       // hit records are gathered in image order; the picture is order-free

    // ---- one scene replica, one stream, one shard per device
    const uint32_t fpp = hits_mode ? 4 * spp : 3;       // floats per pixel of what is gathered
    uint32_t shard_rows = 0;
    std::vector<uint32_t> rows((size_t)world);
    for (int r = 0; r < world; r++) {
        MR_OK_(mr_band_rows_of(H, band, (uint32_t)r, (uint32_t)world, &rows[(size_t)r]));
        if (rows[(size_t)r] > shard_rows) shard_rows = rows[(size_t)r];
    }
    const size_t shard_floats = (size_t)shard_rows * W * fpp;
    std::vector<mr_scene *> scene((size_t)world, nullptr);
    std::vector<hipStream_t> stream((size_t)world);
    std::vector<float *> d_shard((size_t)world, nullptr), d_rgb((size_t)world, nullptr);
    std::vector<uint64_t *> d_counts((size_t)world, nullptr);
    std::vector<int> devs((size_t)world);
    for (int r = 0; r < world; r++) {
        devs[(size_t)r] = virtual_ranks ? 0 : r;
        HIP_OK(hipSetDevice(devs[(size_t)r]));
        MR_OK_(mr_scene_create(devs[(size_t)r], &scene[(size_t)r]));
        uint32_t ntri = 0;
        MR_OK_(mr_scene_add_obj(scene[(size_t)r], model, 0, &ntri));
        if (strcmp(floor_s, "-") != 0) {
            float f[9];
            if (sscanf(floor_s, "%f,%f,%f,%f,%f,%f,%f,%f,%f", f, f + 1, f + 2, f + 3, f + 4, f + 5, f + 6, f + 7, f + 8) != 9) return 4;
            const float up[9] = {0, 1, 0, 0, 1, 0, 0, 1, 0};
            MR_OK_(mr_scene_add_triangle(scene[(size_t)r], f, up));
        }
        MR_OK_(mr_bvh_build(scene[(size_t)r], 0));
        HIP_OK(hipStreamCreate(&stream[(size_t)r]));
        HIP_OK(hipMalloc((void **)&d_shard[(size_t)r], (shard_floats ? shard_floats : 1) * sizeof(float)));
        HIP_OK(hipMemsetAsync(d_shard[(size_t)r], 0, (shard_floats ? shard_floats : 1) * sizeof(float), stream[(size_t)r]));
        HIP_OK(hipMalloc((void **)&d_counts[(size_t)r], 2 * sizeof(uint64_t)));
        HIP_OK(hipMemsetAsync(d_counts[(size_t)r], 0, 2 * sizeof(uint64_t), stream[(size_t)r]));
        if (hits_mode) HIP_OK(hipMalloc((void **)&d_rgb[(size_t)r], ((size_t)rows[(size_t)r] * W * 3 + 1) * sizeof(float)));
    }
    std::vector<ncclComm_t> comm((size_t)world);
    if (!virtual_ranks) NCCL_OK(ncclCommInitAll(comm.data(), world, devs.data()));

    // ---- render: one launch per device, straight into its shard (rgb mode) or with the hit records as the shard
    for (int r = 0; r < world; r++) {
        if (rows[(size_t)r] == 0) continue;
        HIP_OK(hipSetDevice(devs[(size_t)r]));
        mr_frame_desc mine = fd;
        mine.band_rank = (uint32_t)r;
        if (world == 1) { mine.band_world = 1; mine.y0 = 0; mine.y1 = H; }
        if (hits_mode)
            MR_OK_(mr_render_direct(scene[(size_t)r], &mine, d_rgb[(size_t)r], (mr_hit *)d_shard[(size_t)r], 0, d_counts[(size_t)r], stream[(size_t)r]));
        else
            MR_OK_(mr_render_direct(scene[(size_t)r], &mine, d_shard[(size_t)r], 0, 0, d_counts[(size_t)r], stream[(size_t)r]));
    }

    // ---- the frame's single collective: every shard to device 0
    float *d_recv = nullptr, *d_full = nullptr;
    HIP_OK(hipSetDevice(0));
    HIP_OK(hipMalloc((void **)&d_recv, (shard_floats ? shard_floats : 1) * (size_t)world * sizeof(float)));
    HIP_OK(hipMalloc((void **)&d_full, (size_t)W * H * fpp * sizeof(float)));
    if (virtual_ranks) {
        for (int r = 0; r < world; r++) {
            HIP_OK(hipStreamSynchronize(stream[(size_t)r]));
            HIP_OK(hipMemcpyAsync(d_recv + (size_t)r * shard_floats, d_shard[(size_t)r], shard_floats * sizeof(float), hipMemcpyDeviceToDevice, stream[0]));
        }
    } else {
        NCCL_OK(ncclGroupStart());
        for (int r = 0; r < world; r++)
            NCCL_OK(ncclGather(d_shard[(size_t)r], r == 0 ? d_recv : nullptr, shard_floats, ncclFloat, 0, comm[(size_t)r], stream[(size_t)r]));
        NCCL_OK(ncclGroupEnd());
    }
    MR_OK_(mr_deinterleave_bands(scene[0], d_recv, d_full, W, H, band, (uint32_t)world, shard_rows, fpp, stream[0]));

    // ---- output
    unsigned long long n_primary = 0, n_shadow = 0;
    for (int r = 0; r < world; r++) {
        uint64_t c[2] = {0, 0};
        HIP_OK(hipSetDevice(devs[(size_t)r]));
        HIP_OK(hipMemcpyAsync(c, d_counts[(size_t)r], sizeof(c), hipMemcpyDeviceToHost, stream[(size_t)r]));
        HIP_OK(hipStreamSynchronize(stream[(size_t)r]));
        n_primary += c[0]; n_shadow += c[1];
    }
    HIP_OK(hipSetDevice(0));
    FILE *fp = fopen(out_path, "wb");
    if (!fp) return 5;
    if (hits_mode) {
        std::vector<float> h((size_t)W * H * fpp);
        HIP_OK(hipMemcpyAsync(h.data(), d_full, h.size() * sizeof(float), hipMemcpyDeviceToHost, stream[0]));
        HIP_OK(hipStreamSynchronize(stream[0]));
        fwrite(h.data(), sizeof(float), h.size(), fp);
    } else {
        uint8_t *d_rgb8 = nullptr;
        const size_t npix = (size_t)W * H;
        HIP_OK(hipMalloc((void **)&d_rgb8, npix * 3));
        MR_OK_(mr_tonemap(scene[0], d_full, npix * 3, d_rgb8, stream[0]));                                   // Scene.cpp:177-202
        std::vector<uint8_t> img(npix * 3);
        HIP_OK(hipMemcpyAsync(img.data(), d_rgb8, npix * 3, hipMemcpyDeviceToHost, stream[0]));
        HIP_OK(hipStreamSynchronize(stream[0]));
        fprintf(fp, "P6\n%u %u\n255\n", W, H);
        fwrite(img.data(), 1, img.size(), fp);
        hipFree(d_rgb8);
    }
    fclose(fp);
    printf("rays %llu %llu\n", n_primary, n_shadow);
    printf("band %u devices %d\n", band, world);

    for (int r = 0; r < world; r++) {
        HIP_OK(hipSetDevice(devs[(size_t)r]));
        if (!virtual_ranks) ncclCommDestroy(comm[(size_t)r]);
        hipFree(d_shard[(size_t)r]); hipFree(d_counts[(size_t)r]); hipFree(d_rgb[(size_t)r]);
        hipStreamDestroy(stream[(size_t)r]);
        mr_scene_destroy(scene[(size_t)r]);
    }
    hipFree(d_recv); hipFree(d_full);
    return 0;
}
