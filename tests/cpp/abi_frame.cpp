// One frame through the C ABI alone, the way a C++ host (the reference's Scene::raytraceImage, Scene.cpp:93-212, turned
// into batches) would drive it: no Python, no torch -- g++, include/miro_hip.h, libmiro_hip.so and the HIP runtime for the
// device buffers.
//
//   scene:  TriangleMesh::load + floor triangle + Scene::preCalc          mr_scene_add_obj / _add_triangle / mr_bvh_build
//   frame:  Camera::eyeRay -> Scene::trace -> Phong shadow rays -> trace -> Phong::shade -> tone map
//
// usage: abi_frame <model.obj> <floor 9 floats | -> <W> <H> <spp> <eye xyz> <lookat xyz> <fov> <light xyz> <wattage> <out.ppm>
// Writes a binary PPM (P6) and prints "rays <primary> <shadow>".
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "miro_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 10; } } while (0)
#define MR_OK_(x) do { if ((x) != MR_OK) { fprintf(stderr, "%s: %s\n", #x, mr_last_error()); return 11; } } while (0)

static bool parse3(const char *s, float v[3]) { return sscanf(s, "%f,%f,%f", v, v + 1, v + 2) == 3; }

int main(int argc, char **argv) {
    if (argc < 12) { fprintf(stderr, "usage\n"); return 2; }
    const char *model = argv[1], *floor_s = argv[2];
    const uint32_t W = (uint32_t)atoi(argv[3]), H = (uint32_t)atoi(argv[4]), spp = (uint32_t)atoi(argv[5]);
    mr_camera cam;
    mr_light light;
    if (!parse3(argv[6], cam.eye) || !parse3(argv[7], cam.lookat) || !parse3(argv[9], light.position)) return 3;
    cam.up[0] = 0; cam.up[1] = 1; cam.up[2] = 0;
    cam.fov_deg = (float)atof(argv[8]);
    light.color[0] = light.color[1] = light.color[2] = 1.0f;
    light.wattage = (float)atof(argv[10]);
    const char *out_path = argv[11];

    mr_scene *scene = 0;
    MR_OK_(mr_scene_create(0, &scene));
    uint32_t ntri = 0;
    MR_OK_(mr_scene_add_obj(scene, model, 0, &ntri));
    if (strcmp(floor_s, "-") != 0) {
        float f[9];
        if (sscanf(floor_s, "%f,%f,%f,%f,%f,%f,%f,%f,%f", f, f + 1, f + 2, f + 3, f + 4, f + 5, f + 6, f + 7, f + 8) != 9) return 4;
        const float up[9] = {0, 1, 0, 0, 1, 0, 0, 1, 0};
        MR_OK_(mr_scene_add_triangle(scene, f, up));
    }
    MR_OK_(mr_bvh_build(scene, 0));

    const uint64_t n = (uint64_t)W * H * spp, npix = (uint64_t)W * H;
    mr_ray *d_rays = 0, *d_srays = 0;
    mr_hit *d_hits = 0, *d_shits = 0;
    uint32_t *d_src = 0;
    uint64_t *d_count = 0;
    float *d_rgb = 0;
    uint8_t *d_rgb8 = 0;
    HIP_OK(hipMalloc((void **)&d_rays, n * sizeof(mr_ray)));
    HIP_OK(hipMalloc((void **)&d_srays, n * sizeof(mr_ray)));
    HIP_OK(hipMalloc((void **)&d_hits, n * sizeof(mr_hit)));
    HIP_OK(hipMalloc((void **)&d_shits, n * sizeof(mr_hit)));
    HIP_OK(hipMalloc((void **)&d_src, n * sizeof(uint32_t)));
    HIP_OK(hipMalloc((void **)&d_count, sizeof(uint64_t)));
    HIP_OK(hipMalloc((void **)&d_rgb, npix * 3 * sizeof(float)));
    HIP_OK(hipMalloc((void **)&d_rgb8, npix * 3));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    const float white[3] = {1.0f, 1.0f, 1.0f};
    MR_OK_(mr_gen_eye_rays(scene, &cam, W, H, 0, H, spp, spp > 1, 168, d_rays, stream));                    // Camera::eyeRay
    MR_OK_(mr_trace(scene, d_rays, n, d_hits, MR_RAYS_ON_DEVICE | MR_HITS_ON_DEVICE, stream));               // Scene.cpp:278
    MR_OK_(mr_gen_shadow_rays(scene, d_rays, d_hits, n, light.position, d_srays, d_src, d_count, stream));   // Phong.cpp:80-97
    MR_OK_(mr_trace_indirect(scene, d_srays, d_count, n, d_shits, 0, stream));                               // Phong.cpp:97
    MR_OK_(mr_shade_direct(scene, d_rays, d_hits, n, d_shits, d_src, d_count, &light, white, spp, d_rgb, stream));
    MR_OK_(mr_tonemap(scene, d_rgb, npix * 3, d_rgb8, stream));                                              // Scene.cpp:177-202
    std::vector<uint8_t> img(npix * 3);
    uint64_t n_shadow = 0;
    HIP_OK(hipMemcpyAsync(img.data(), d_rgb8, npix * 3, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(&n_shadow, d_count, sizeof(n_shadow), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));

    FILE *fp = fopen(out_path, "wb");
    if (!fp) return 5;
    fprintf(fp, "P6\n%u %u\n255\n", W, H);
    fwrite(img.data(), 1, img.size(), fp);
    fclose(fp);
    printf("rays %llu %llu\n", (unsigned long long)n, (unsigned long long)n_shadow);

    hipFree(d_rays); hipFree(d_srays); hipFree(d_hits); hipFree(d_shits); hipFree(d_src); hipFree(d_count); hipFree(d_rgb); hipFree(d_rgb8);
    hipStreamDestroy(stream);
    mr_scene_destroy(scene);
    return 0;
}
