// Drives the C++ shim the way the reference's scene code drives Scene/BVH (assignment2.cpp:24-70,449-461):
// load a mesh, add one Triangle object per face plus the floor triangle, preCalc(), then trace rays read from
// a file -- the first `n_single` through Scene::trace one call at a time, all of them through traceBatch --
// and dump t, P, N, object index for the Python test to compare with the oracle.
//
// usage: shim_render <model.obj> <floor 9 floats | -> <rays.bin> <out.bin> <n_single>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "miro_shim.hpp"

using namespace miro;

namespace miro { class Material { public: int id; }; }

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage\n"); return 2; }
    Scene scene;
    Material white; white.id = 7;
    TriangleMesh mesh;
    if (!mesh.load(argv[1])) { fprintf(stderr, "cannot load %s\n", argv[1]); return 3; }
    std::vector<Triangle *> keep;
    for (int i = 0; i < mesh.numTris(); ++i) {
        Triangle *t = new Triangle;
        t->setIndex(i);
        t->setMesh(&mesh);
        t->setMaterial(&white);
        scene.addObject(t);
        keep.push_back(t);
    }
    TriangleMesh floor;
    if (strcmp(argv[2], "-") != 0) {
        float f[9];
        if (sscanf(argv[2], "%f,%f,%f,%f,%f,%f,%f,%f,%f", f, f + 1, f + 2, f + 3, f + 4, f + 5, f + 6, f + 7, f + 8) != 9) return 4;
        floor.createSingleTriangle();
        floor.setV1(Vector3(f[0], f[1], f[2])); floor.setV2(Vector3(f[3], f[4], f[5])); floor.setV3(Vector3(f[6], f[7], f[8]));
        floor.setN1(Vector3(0, 1, 0)); floor.setN2(Vector3(0, 1, 0)); floor.setN3(Vector3(0, 1, 0));
        Triangle *t = new Triangle;
        t->setIndex(0);
        t->setMesh(&floor);
        t->setMaterial(&white);
        scene.addObject(t);
        keep.push_back(t);
    }
    scene.preCalc();

    FILE *fp = fopen(argv[3], "rb");
    if (!fp) return 5;
    fseek(fp, 0, SEEK_END);
    size_t n = (size_t)ftell(fp) / 32;
    fseek(fp, 0, SEEK_SET);
    std::vector<float> raw(n * 8);
    if (fread(raw.data(), 32, n, fp) != n) return 6;
    fclose(fp);
    std::vector<Ray> rays(n);
    for (size_t i = 0; i < n; i++) rays[i] = Ray(Vector3(raw[8 * i], raw[8 * i + 1], raw[8 * i + 2]), Vector3(raw[8 * i + 4], raw[8 * i + 5], raw[8 * i + 6]));

    size_t n_single = (size_t)atol(argv[5]);
    if (n_single > n) n_single = n;
    std::vector<HitInfo> single(n_single), batch(n);
    std::vector<char> single_hit(n_single), batch_hit(n);
    for (size_t i = 0; i < n_single; i++) single_hit[i] = scene.trace(single[i], rays[i]) ? 1 : 0;   // default tMin/tMax
    scene.traceBatch(rays.data(), n, batch.data(), reinterpret_cast<bool *>(batch_hit.data()));

    // single-ray calls and the batch must agree exactly
    for (size_t i = 0; i < n_single; i++) {
        if (single_hit[i] != batch_hit[i] || (single_hit[i] && (memcmp(&single[i].t, &batch[i].t, 4) || single[i].object != batch[i].object ||
                                                               memcmp(&single[i].P, &batch[i].P, 12) || memcmp(&single[i].N, &batch[i].N, 12)))) {
            fprintf(stderr, "single/batch mismatch at ray %zu\n", i);
            return 7;
        }
    }
    // out: per ray 8 floats: hit, t, P.xyz, N.xyz  + 1 int32 object index + material id
    const Objects &objs = *scene.objects();
    FILE *out = fopen(argv[4], "wb");
    if (!out) return 8;
    for (size_t i = 0; i < n; i++) {
        float rec[8] = {batch_hit[i] ? 1.0f : 0.0f, batch[i].t, batch[i].P.x, batch[i].P.y, batch[i].P.z, batch[i].N.x, batch[i].N.y, batch[i].N.z};
        int32_t meta[2] = {-1, -1};
        if (batch_hit[i]) {
            for (size_t k = 0; k < objs.size(); k++) if (objs[k] == batch[i].object) { meta[0] = (int32_t)k; break; }
            meta[1] = batch[i].material ? batch[i].material->id : -2;
        }
        fwrite(rec, 4, 8, out);
        fwrite(meta, 4, 2, out);
    }
    fclose(out);
    printf("shim_render: %zu rays, %zu single calls ok\n", n, n_single);
    for (size_t i = 0; i < keep.size(); i++) delete keep[i];
    return 0;
}
