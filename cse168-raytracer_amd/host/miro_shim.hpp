// miro_shim.hpp -- the reference's C++ surface for the intersection path, re-created on top of the
// C ABI of include/miro_hip.h so that the reference's render loop can keep calling
//
//     bool Scene::trace(HitInfo&, const Ray&, float tMin, float tMax) const        (Scene.h:38-39)
//     void BVH::build(Objects*, int depth)                                         (BVH.h:33)
//     bool BVH::intersect(HitInfo&, const Ray&, float tMin, float tMax) const      (BVH.h:35-36)
//
// with the work done by the HIP library.  Same names, argument meaning and defaults as the reference;
// the additions are the *Batch entry points (a single-ray call is a batch of one and pays a kernel
// launch: the render loop should hand over whole ray batches).
//
// Header-only, C++11, links against libmiro_hip.so only.  What is restated here from the reference:
// the value types (Vector3 subset, Ray, HitInfo, Ray.h:21-84), TriangleMesh/Triangle as data carriers
// (TriangleMesh.h:9-69, Triangle.h:12-45), HitInfo reconstruction P, N (Triangle.cpp:160-166) and
// Scene::trace's normal normalisation (Scene.cpp:262), Sphere / Plane as data carriers with their P, N
// (Sphere.cpp:61-63, Plane.cpp:42-44).  Nothing is intersected on the CPU.
#pragma once

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "miro_hip.h"

namespace miro {

const float MIRO_TMAX = 1e12f;   // Miro.h:8
const float epsilon = 1e-4f;     // Miro.h:9

struct Vector3 {
    float x, y, z;
    Vector3() : x(0), y(1), z(2) {}                    // sic, Vector3.h:26-27
    Vector3(float s) : x(s), y(s), z(s) {}
    Vector3(float a, float b, float c) : x(a), y(b), z(c) {}
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
    Vector3 operator+(const Vector3 &v) const { return Vector3(x + v.x, y + v.y, z + v.z); }
    Vector3 operator-(const Vector3 &v) const { return Vector3(x - v.x, y - v.y, z - v.z); }
    Vector3 operator-() const { return Vector3(-x, -y, -z); }
    Vector3 operator*(float a) const { return Vector3(x * a, y * a, z * a); }
    const Vector3 &operator/=(float a) { float inv = float(1) / a; x *= inv; y *= inv; z *= inv; return *this; }
    float length2() const { return x * x + y * y + z * z; }
    float length() const { return sqrtf(length2()); }
    const Vector3 &normalize() { return (*this /= length()); }
};
inline Vector3 operator*(float s, const Vector3 &v) { return Vector3(v.x * s, v.y * s, v.z * s); }

class Material;   // the caller's own type; the shim only carries the pointer (Object.h:17-18)

class Object {
public:
    Object() : m_material(0) {}
    virtual ~Object() {}
    void setMaterial(const Material *m) { m_material = m; }
    const Material *getMaterial() const { return m_material; }
    virtual bool isBounded() const { return true; }
protected:
    const Material *m_material;
};
typedef std::vector<Object *> Objects;

class TriangleMesh {
public:
    struct TupleI3 { unsigned int v[3]; };
    TriangleMesh() {}
    // TriangleMesh::load (TriangleMeshLoad.cpp:63-79): false when the file cannot be opened
    bool load(const char *file, const float *ctm16 = 0) {
        mr_scene *tmp = 0;
        if (mr_scene_create(0, &tmp) != MR_OK) return false;
        uint32_t nt = 0;
        bool ok = mr_scene_add_obj(tmp, file, ctm16, &nt) == MR_OK;
        if (ok) {
            mr_mesh_desc d;
            mr_scene_get_mesh(tmp, &d);
            m_vertices.assign(d.vertices, d.vertices + 3 * (size_t)d.n_vertices);
            m_normals.assign(d.normals, d.normals + 3 * (size_t)d.n_normals);
            m_vi.assign(d.vidx, d.vidx + 3 * (size_t)d.n_triangles);
            m_ni.assign(d.nidx, d.nidx + 3 * (size_t)d.n_triangles);
        }
        mr_scene_destroy(tmp);
        return ok;
    }
    // TriangleMesh::createSingleTriangle + setV1..3 / setN1..3 (TriangleMeshLoad.cpp:15-56)
    void createSingleTriangle() {
        m_vertices.assign(9, 0.0f); m_normals.assign(9, 0.0f);
        m_vi.assign({0u, 1u, 2u}); m_ni.assign({0u, 1u, 2u});
    }
    void setV1(const Vector3 &v) { setv(m_vertices, 0, v); }
    void setV2(const Vector3 &v) { setv(m_vertices, 1, v); }
    void setV3(const Vector3 &v) { setv(m_vertices, 2, v); }
    void setN1(const Vector3 &v) { setv(m_normals, 0, v); }
    void setN2(const Vector3 &v) { setv(m_normals, 1, v); }
    void setN3(const Vector3 &v) { setv(m_normals, 2, v); }
    int numTris() const { return (int)(m_vi.size() / 3); }
    Vector3 vertex(unsigned i) const { return Vector3(m_vertices[3 * i], m_vertices[3 * i + 1], m_vertices[3 * i + 2]); }
    Vector3 normal(unsigned i) const { return Vector3(m_normals[3 * i], m_normals[3 * i + 1], m_normals[3 * i + 2]); }
    const unsigned *vIndices(unsigned tri) const { return &m_vi[3 * tri]; }
    const unsigned *nIndices(unsigned tri) const { return &m_ni[3 * tri]; }
private:
    static void setv(std::vector<float> &a, int i, const Vector3 &v) { a[3 * i] = v.x; a[3 * i + 1] = v.y; a[3 * i + 2] = v.z; }
    std::vector<float> m_vertices, m_normals;
    std::vector<unsigned> m_vi, m_ni;
};

// Triangle (Triangle.h:12-45): a (mesh, index) pair
class Triangle : public Object {
public:
    Triangle(TriangleMesh *m = 0, unsigned int i = 0) : m_mesh(m), m_index(i) {}
    void setIndex(unsigned int i) { m_index = i; }
    unsigned int getIndex() const { return m_index; }
    void setMesh(TriangleMesh *m) { m_mesh = m; }
    TriangleMesh *getMesh() const { return m_mesh; }
private:
    TriangleMesh *m_mesh;
    unsigned int m_index;
};

// Sphere (Sphere.h:8-36) and Plane (Plane.h:12-34) as data carriers: intersected on the device
class Sphere : public Object {
public:
    Sphere() : m_center(), m_radius(0.0f) {}              // m_center = Vector3() = (0,1,2), sic
    void setCenter(const Vector3 &v) { m_center = v; }
    void setRadius(const float f) { m_radius = f; }
    const Vector3 &center() const { return m_center; }
    float radius() const { return m_radius; }
protected:
    Vector3 m_center;
    float m_radius;
};

class Plane : public Object {
public:
    Plane() : m_normal(0, 1, 0), m_origin(0, 0, 0) {}     // Plane.cpp:7-11
    void setNormal(Vector3 normal) { m_normal = normal; }
    void setOrigin(Vector3 origin) { m_origin = origin; }
    const Vector3 &normal() const { return m_normal; }
    const Vector3 &origin() const { return m_origin; }
    virtual bool isBounded() const { return false; }      // Plane.h:27
protected:
    Vector3 m_normal, m_origin;
};

class Ray {                                               // Ray.h:40-84
public:
    bool isDiffuse;
    Vector3 o, d;
    Ray() : isDiffuse(false), o(), d(Vector3(0.0f, 0.0f, 1.0f)) {}
    Ray(const Vector3 &o_, const Vector3 &d_) : isDiffuse(false), o(o_), d(d_) {}
};

class HitInfo {                                           // Ray.h:21-38
public:
    float t;
    Vector3 P, N;
    const Material *material;
    const Object *object;
    explicit HitInfo(float t_ = 0.0f, const Vector3 &P_ = Vector3(), const Vector3 &N_ = Vector3(0.0f, 1.0f, 0.0f))
        : t(t_), P(P_), N(N_), material(0), object(0) {}
};

class MiroHipError : public std::runtime_error {
public:
    MiroHipError(mr_status s, const std::string &what) : std::runtime_error(what), status(s) {}
    mr_status status;
};
inline void check(mr_status s) { if (s != MR_OK) throw MiroHipError(s, mr_last_error()); }

// BVH (BVH.h:29-63).  build() keeps the caller's Objects* like the reference does (BVH.cpp:84) and
// uploads their triangles; intersect() is a closest-hit query on the device.
class BVH {
public:
    BVH() : m_scene(0), m_objs(0), m_unbounded(0), m_device(0) {}
    ~BVH() { if (m_scene) mr_scene_destroy(m_scene); }
    void setDevice(int device) { m_device = device; }
    // Scene::m_unboundedObjects (Scene.h:22-23): scanned by the device after the BVH, as Scene::trace does
    // (Scene.cpp:220-230).  Call before build(); BVH::intersect on its own knows no unbounded objects.
    void setUnbounded(Objects *objs) { m_unbounded = objs; }

    void build(Objects *objs, int /*depth*/ = 0) {
        if (m_scene) { mr_scene_destroy(m_scene); m_scene = 0; }
        m_objs = objs;
        m_flat.clear();
        check(mr_scene_create(m_device, &m_scene));
        // objects go over in addObject order, so that prim index == object index: runs of triangles as meshes of
        // 3 vertices + 3 normals per triangle, spheres one by one
        const size_t n = objs->size();
        std::vector<float> v, nn;
        std::vector<uint32_t> idx;
        m_flat.resize(n);
        struct Flush {
            static void run(mr_scene *sc, std::vector<float> &v, std::vector<float> &nn, std::vector<uint32_t> &idx) {
                if (idx.empty()) return;
                mr_mesh_desc d;
                d.vertices = v.data(); d.n_vertices = (uint32_t)(v.size() / 3);
                d.normals = nn.data(); d.n_normals = (uint32_t)(nn.size() / 3);
                d.vidx = idx.data(); d.nidx = idx.data(); d.n_triangles = (uint32_t)(idx.size() / 3);
                check(mr_scene_add_mesh(sc, &d));
                v.clear(); nn.clear(); idx.clear();
            }
        };
        for (size_t i = 0; i < n; i++) {
            Flat &f = m_flat[i];
            if (Sphere *sp = dynamic_cast<Sphere *>((*objs)[i])) {
                Flush::run(m_scene, v, nn, idx);
                const float c[3] = {sp->center().x, sp->center().y, sp->center().z};
                uint32_t prim = 0;
                check(mr_scene_add_sphere(m_scene, c, sp->radius(), &prim));
                if (prim != i) throw MiroHipError(MR_ERR_STATE, "object order lost");
                f.sphere = true; f.A = sp->center();
                continue;
            }
            Triangle *t = dynamic_cast<Triangle *>((*objs)[i]);
            if (!t) throw MiroHipError(MR_ERR_INVALID, "bounded objects must be Triangle or Sphere");
            const TriangleMesh *m = t->getMesh();
            const unsigned *vi = m->vIndices(t->getIndex()), *ni = m->nIndices(t->getIndex());
            for (int k = 0; k < 3; k++) {
                Vector3 p = m->vertex(vi[k]), q = m->normal(ni[k]);
                idx.push_back((uint32_t)(v.size() / 3));
                v.push_back(p.x); v.push_back(p.y); v.push_back(p.z);
                nn.push_back(q.x); nn.push_back(q.y); nn.push_back(q.z);
            }
            f.sphere = false;
            f.A = m->vertex(vi[0]); f.BmA = m->vertex(vi[1]) - f.A; f.CmA = m->vertex(vi[2]) - f.A;
            f.nA = m->normal(ni[0]); f.nB = m->normal(ni[1]); f.nC = m->normal(ni[2]);
        }
        Flush::run(m_scene, v, nn, idx);
        for (size_t i = 0; m_unbounded && i < m_unbounded->size(); i++) {
            Plane *pl = dynamic_cast<Plane *>((*m_unbounded)[i]);
            if (!pl) throw MiroHipError(MR_ERR_INVALID, "unbounded objects must be Plane");
            const float nrm[3] = {pl->normal().x, pl->normal().y, pl->normal().z};
            const float org[3] = {pl->origin().x, pl->origin().y, pl->origin().z};
            check(mr_scene_add_plane(m_scene, nrm, org, 0, 0));
        }
        check(mr_bvh_build(m_scene, 0));
    }

    bool intersect(HitInfo &result, const Ray &ray, float tMin = 0.0f, float tMax = MIRO_TMAX) const {
        bool hit = false;
        intersectBatch(&ray, 1, &result, &hit, tMin, tMax);
        return hit;
    }

    // n rays sharing [tMin, tMax]; hit_flags may be 0.  Returns the number of hits.
    size_t intersectBatch(const Ray *rays, size_t n, HitInfo *results, bool *hit_flags, float tMin = 0.0f,
                          float tMax = MIRO_TMAX, uint32_t flags = MR_TRACE_CLOSEST) const {
        if (!m_scene) throw MiroHipError(MR_ERR_STATE, "BVH::build has not been called");
        std::vector<mr_ray> r(n);
        std::vector<mr_hit> h(n);
        for (size_t i = 0; i < n; i++) {
            r[i].ox = rays[i].o.x; r[i].oy = rays[i].o.y; r[i].oz = rays[i].o.z; r[i].tmin = tMin;
            r[i].dx = rays[i].d.x; r[i].dy = rays[i].d.y; r[i].dz = rays[i].d.z; r[i].tmax = tMax;
        }
        check(mr_trace(m_scene, r.data(), n, h.data(), flags & ~(uint32_t)(MR_RAYS_ON_DEVICE | MR_HITS_ON_DEVICE), 0));
        size_t hits = 0;
        for (size_t i = 0; i < n; i++) {
            const bool hit = h[i].prim != MR_MISS;
            if (hit_flags) hit_flags[i] = hit;
            results[i].t = h[i].t;                        // minHit.t = tMax on a miss (BVH.cpp:444)
            if (!hit) continue;
            hits++;
            if (h[i].prim & MR_PLANE_BIT) {                                         // Scene.cpp:223-229
                const Plane *pl = static_cast<const Plane *>((*m_unbounded)[h[i].prim & ~MR_PLANE_BIT]);
                results[i].P = rays[i].o + h[i].t * rays[i].d;                      // Plane.cpp:42
                results[i].N = pl->normal();                                        // Plane.cpp:44
                results[i].object = pl;
                results[i].material = pl->getMaterial();
                continue;
            }
            const Flat &f = m_flat[h[i].prim];
            if (f.sphere) {
                results[i].P = rays[i].o + h[i].t * rays[i].d;                      // Sphere.cpp:61
                results[i].N = (results[i].P - f.A);                                // Sphere.cpp:62-63
                results[i].N.normalize();
            } else {
                const float beta = h[i].beta, gamma = h[i].gamma;
                results[i].P = f.A + beta * f.BmA + gamma * f.CmA;                      // Triangle.cpp:160
                results[i].N = (1 - beta - gamma) * f.nA + beta * f.nB + gamma * f.nC;  // Triangle.cpp:162
            }
            results[i].object = (*m_objs)[h[i].prim];                               // BVH.cpp:506
            results[i].material = results[i].object->getMaterial();                 // Triangle.cpp:166
        }
        return hits;
    }

    mr_scene *handle() const { return m_scene; }

private:
    BVH(const BVH &);
    BVH &operator=(const BVH &);
    struct Flat { bool sphere; Vector3 A, BmA, CmA, nA, nB, nC; };       // sphere: A = centre
    mr_scene *m_scene;
    Objects *m_objs, *m_unbounded;
    std::vector<Flat> m_flat;
    int m_device;
};

// Scene (Scene.h:14-73), intersection part: addObject, preCalc -> BVH::build, trace.
class Scene {
public:
    void addObject(Object *pObj) {                        // Scene.h:20-25
        if (pObj->isBounded()) m_objects.push_back(pObj);
        else m_unboundedObjects.push_back(pObj);
    }
    const Objects *objects() const { return &m_objects; }
    const Objects *unboundedObjects() const { return &m_unboundedObjects; }
    void setDevice(int device) { m_bvh.setDevice(device); }
    void preCalc() { m_bvh.setUnbounded(&m_unboundedObjects); m_bvh.build(&m_objects); }   // Scene.cpp:50-84, the BVH part (:72)

    bool trace(HitInfo &minHit, const Ray &ray, float tMin = 0.0f, float tMax = MIRO_TMAX) const {   // Scene.cpp:214
        bool hit = false;
        traceBatch(&ray, 1, &minHit, &hit, tMin, tMax);
        return hit;
    }
    size_t traceBatch(const Ray *rays, size_t n, HitInfo *results, bool *hit_flags, float tMin = 0.0f,
                      float tMax = MIRO_TMAX, uint32_t flags = MR_TRACE_CLOSEST) const {
        std::vector<bool> dummy;
        std::vector<char> flagbuf(hit_flags ? 0 : n);
        bool *hf = hit_flags ? hit_flags : reinterpret_cast<bool *>(flagbuf.data());
        size_t hits = m_bvh.intersectBatch(rays, n, results, hf, tMin, tMax, flags);
        for (size_t i = 0; i < n; i++)
            if (hf[i]) results[i].N.normalize();          // Scene.cpp:262 (UV-lookup materials, bump height 0)
        return hits;
    }
    const BVH &bvh() const { return m_bvh; }

private:
    Objects m_objects, m_unboundedObjects;
    BVH m_bvh;
};

}  // namespace miro
