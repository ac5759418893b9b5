// Drives the C++ shim the way the reference's scene code drives Scene/BVH (assignment2.cpp:24-70,449-461):
// load a mesh, add one Triangle object per face plus the floor triangle, preCalc(), then trace rays read from
// a file -- the first `n_single` through Scene::trace one call at a time, all of them through traceBatch --
// and dump t, P, N, object index for the Python test to compare with the oracle.
//
// With "@<objects.txt>" in place of the model, the scene is a list of spheres / planes / triangles in addObject order
// (assignment1.cpp:31-72 style), one per line: "s cx cy cz r" | "p nx ny nz ox oy oz" | "t 9 vertex + 9 normal
// floats" (hex floats, so that both sides hold the same fp32 values).
//
// usage: shim_render <model.obj | @objects.txt> <floor 9 floats | -> <rays.bin> <out.bin> <n_single>
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
    Material shiny; shiny.id = 8;
    Material ground; ground.id = 9;
    TriangleMesh mesh;
    std::vector<Object *> keep;
    std::vector<TriangleMesh *> singles;
    if (argv[1][0] == '@') {
        FILE *fo = fopen(argv[1] + 1, "r");
        if (!fo) return 3;
        char kind;
        while (fscanf(fo, " %c", &kind) == 1) {
            float f[18];
            const int want = kind == 's' ? 4 : (kind == 'p' ? 6 : 18);
            for (int k = 0; k < want; k++) if (fscanf(fo, "%f", &f[k]) != 1) return 4;
            if (kind == 's') {
                Sphere *sp = new Sphere;
                sp->setCenter(Vector3(f[0], f[1], f[2]));
                sp->setRadius(f[3]);
                sp->setMaterial(&shiny);
                scene.addObject(sp);
                keep.push_back(sp);
            } else if (kind == 'p') {
                Plane *pl = new Plane();
                pl->setNormal(Vector3(f[0], f[1], f[2]));
                pl->setOrigin(Vector3(f[3], f[4], f[5]));
                pl->setMaterial(&ground);
                scene.addObject(pl);
                keep.push_back(pl);
            } else {
                TriangleMesh *m = new TriangleMesh;
                m->createSingleTriangle();
                m->setV1(Vector3(f[0], f[1], f[2])); m->setV2(Vector3(f[3], f[4], f[5])); m->setV3(Vector3(f[6], f[7], f[8]));
                m->setN1(Vector3(f[9], f[10], f[11])); m->setN2(Vector3(f[12], f[13], f[14])); m->setN3(Vector3(f[15], f[16], f[17]));
                singles.push_back(m);
                Triangle *t = new Triangle;
                t->setIndex(0);
                t->setMesh(m);
                t->setMaterial(&white);
                scene.addObject(t);
                keep.push_back(t);
            }
        }
        fclose(fo);
    } else if (!mesh.load(argv[1])) { fprintf(stderr, "cannot load %s\n", argv[1]); return 3; }
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
    const Objects &unb = *scene.unboundedObjects();
    FILE *out = fopen(argv[4], "wb");
    if (!out) return 8;
    for (size_t i = 0; i < n; i++) {
        float rec[8] = {batch_hit[i] ? 1.0f : 0.0f, batch[i].t, batch[i].P.x, batch[i].P.y, batch[i].P.z, batch[i].N.x, batch[i].N.y, batch[i].N.z};
        int32_t meta[2] = {-1, -1};
        if (batch_hit[i]) {
            for (size_t k = 0; k < objs.size(); k++) if (objs[k] == batch[i].object) { meta[0] = (int32_t)k; break; }
            for (size_t k = 0; k < unb.size(); k++) if (unb[k] == batch[i].object) { meta[0] = -2 - (int32_t)k; break; }
            meta[1] = batch[i].material ? batch[i].material->id : -2;
        }
        fwrite(rec, 4, 8, out);
        fwrite(meta, 4, 2, out);
    }
    fclose(out);
    printf("shim_render: %zu rays, %zu single calls ok\n", n, n_single);
    for (size_t i = 0; i < keep.size(); i++) delete keep[i];
    for (size_t i = 0; i < singles.size(); i++) delete singles[i];
    return 0;
}
