// mr_api.cpp -- the extern "C" boundary declared in include/miro_hip.h.
// Scene assembly and BVH::build run on the host; mr_bvh_build flattens the tree into the device
// layout of mr_internal.h and uploads it once; mr_trace only moves rays/hits and launches.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "mr_internal.h"
#include "mr_tile.h"

namespace mr {

static thread_local char g_err[512] = "";

mr_status fail(mr_status code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {

template <typename T>
mr_status upload(T *&dst, const T *src, size_t count, uint64_t &bytes) {
    const size_t sz = (count ? count : 1) * sizeof(T);
    MR_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&dst), sz));
    if (count) MR_HIP_CHECK(hipMemcpy(dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    bytes += sz;
    return MR_OK;
}

void release_device(mr_scene *s) {
    DeviceScene &d = s->dev;
    (void)hipFree(d.nodes); (void)hipFree(d.tris); (void)hipFree(d.tri_prim); (void)hipFree(d.leaf_cnt_ext);
    (void)hipFree(d.v); (void)hipFree(d.n); (void)hipFree(d.vi); (void)hipFree(d.ni);
    (void)hipFree(d.materials); (void)hipFree(d.prim_material);
    (void)hipFree(d.spheres); (void)hipFree(d.planes);
    d = DeviceScene();
    (void)hipFree(s->d_stats); s->d_stats = nullptr;
    (void)hipFree(s->d_work_counters); s->d_work_counters = nullptr;
    (void)hipFree(s->d_stage_rays); (void)hipFree(s->d_stage_hits);
    s->d_stage_rays = s->d_stage_hits = nullptr;
    s->stage_cap = 0;
    for (void *e : s->stage_events) (void)hipEventDestroy(static_cast<hipEvent_t>(e));
    s->stage_events.clear();
    if (s->copy_in) (void)hipStreamDestroy(static_cast<hipStream_t>(s->copy_in));
    if (s->copy_out) (void)hipStreamDestroy(static_cast<hipStream_t>(s->copy_out));
    s->copy_in = s->copy_out = nullptr;
    (void)hipFree(s->d_occluded);
    s->d_occluded = nullptr;
    s->occluded_cap = 0;
    (void)hipFree(s->d_light_scale);
    s->d_light_scale = nullptr;
    s->light_scale_cap = 0;
}

inline int32_t leaf_ref(uint32_t first, uint32_t count) {
    const uint32_t c = count < (uint32_t)kLeafCountMask ? count : (uint32_t)kLeafCountMask;
    return (int32_t)~((first << kLeafCountBits) | c);
}

mr_status upload_materials(mr_scene *s);

// Storage order of the inner nodes (MR_LAYOUT_*): the list of host node indices in the order their records are stored;
// -1 = an unused padding record.  Node identity, references and visiting order do not depend on it.
std::vector<int32_t> node_storage_order(const HostTree &t, uint32_t mode) {
    std::vector<int32_t> order;
    const auto inner = [&](int32_t n) { return n >= 0 && !t.nodes[(size_t)n].is_leaf; };
    if (t.nodes.empty() || t.nodes[0].is_leaf) return order;
    if (mode == MR_LAYOUT_PAIRS) {
        // pre-order; a node that has an inner child and would start in the second half of a line although it is not the
        // partner of the record before it moves to the next line: (node, first inner child) always share 128 bytes
        struct Item { int32_t node; bool partner; };
        std::vector<Item> stack{{0, false}};
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            const HostNode &nd = t.nodes[(size_t)it.node];
            const int32_t first = inner(nd.a) ? nd.a : (inner(nd.b) ? nd.b : -1);
            const int32_t other = (first == nd.a && inner(nd.b)) ? nd.b : -1;
            if (!it.partner && (order.size() & 1) && first >= 0) order.push_back(-1);
            const bool even = (order.size() & 1) == 0;
            order.push_back(it.node);
            if (other >= 0) stack.push_back({other, false});
            if (first >= 0) stack.push_back({first, even});
        }
        return order;
    }
    if (mode == MR_LAYOUT_TREELETS) {
        constexpr int kTop = 12, kTreelet = 3;
        std::vector<int32_t> frontier{0}, roots;
        for (int level = 0; level < kTop && !frontier.empty(); level++) {        // breadth-first top
            std::vector<int32_t> next;
            for (int32_t n : frontier) {
                order.push_back(n);
                for (int32_t c : {t.nodes[(size_t)n].a, t.nodes[(size_t)n].b})
                    if (inner(c)) next.push_back(c);
            }
            frontier.swap(next);
        }
        roots = frontier;
        // below: a treelet = the inner nodes of up to kTreelet levels under its root, breadth-first and contiguous; the
        // treelets under it follow depth-first
        std::vector<int32_t> stack(roots.rbegin(), roots.rend());
        while (!stack.empty()) {
            const int32_t root = stack.back();
            stack.pop_back();
            std::vector<int32_t> level{root}, below;
            for (int l = 0; l < kTreelet && !level.empty(); l++) {
                std::vector<int32_t> next;
                for (int32_t n : level) {
                    order.push_back(n);
                    for (int32_t c : {t.nodes[(size_t)n].a, t.nodes[(size_t)n].b})
                        if (inner(c)) next.push_back(c);
                }
                level.swap(next);
            }
            for (auto it = level.rbegin(); it != level.rend(); ++it) stack.push_back(*it);
        }
        return order;
    }
    for (size_t i = 0; i < t.nodes.size(); i++)
        if (!t.nodes[i].is_leaf) order.push_back((int32_t)i);
    return order;
}

// host tree -> device records
mr_status flatten_and_upload(mr_scene *s, uint32_t layout) {
    const HostTree &t = s->tree;
    const HostMesh &m = s->mesh;
    DeviceScene &d = s->dev;
    const uint32_t nt = m.n_triangles();
    if (nt >= (1u << (31 - kLeafCountBits)))
        return fail(MR_ERR_INVALID, "scene has %u triangles; the leaf reference encoding holds < %u", nt,
                    1u << (31 - kLeafCountBits));

    // inner nodes get their record index from the storage order (depth-first pre-order by default)
    const std::vector<int32_t> order = node_storage_order(t, layout & 15u);
    std::vector<int32_t> inner_id(t.nodes.size(), -1);
    const uint32_t n_inner = (uint32_t)order.size();
    for (uint32_t k = 0; k < n_inner; k++)
        if (order[k] >= 0) inner_id[(size_t)order[k]] = (int32_t)k;
    for (size_t i = 0; i < t.nodes.size(); i++)
        if (!t.nodes[i].is_leaf && inner_id[i] < 0) return fail(MR_ERR_STATE, "node %zu missing from the storage order", i);

    // leaves: where each one's triangles start in the record array.  MR_LAYOUT_ALIGN_LEAVES puts up to seven unused
    // 48-byte records in front of a leaf when that lowers the number of 128-byte lines its triangles touch.
    std::vector<uint32_t> leaf_first(t.nodes.size(), 0);
    uint32_t n_rec = 0;
    {
        std::vector<uint32_t> leaves;
        for (size_t i = 0; i < t.nodes.size(); i++)
            if (t.nodes[i].is_leaf) leaves.push_back((uint32_t)i);
        std::sort(leaves.begin(), leaves.end(), [&](uint32_t x, uint32_t y) { return t.nodes[x].a < t.nodes[y].a; });
        for (uint32_t li : leaves) {
            const uint32_t cnt = (uint32_t)t.nodes[li].b;
            if ((layout & MR_LAYOUT_ALIGN_LEAVES) && cnt > 0 && cnt <= 8) {
                const auto lines = [&](uint32_t first) { return (first * 48u + cnt * 48u - 1u) / 128u - (first * 48u) / 128u + 1u; };
                uint32_t best = 0;
                for (uint32_t pad = 1; pad < 8; pad++)
                    if (lines(n_rec + pad) < lines(n_rec + best)) best = pad;
                n_rec += best;
            }
            leaf_first[li] = n_rec;
            n_rec += cnt;
        }
    }
    if (n_rec >= (1u << (31 - kLeafCountBits))) return fail(MR_ERR_INVALID, "padded triangle array exceeds the leaf reference encoding");
    auto ref_of = [&](int32_t node) -> int32_t {
        const HostNode &nd = t.nodes[(size_t)node];
        return nd.is_leaf ? leaf_ref(leaf_first[(size_t)node], (uint32_t)nd.b) : inner_id[(size_t)node];
    };

    std::vector<float4> nodes((size_t)n_inner * 4, make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t i = 0; i < t.nodes.size(); i++) {
        const HostNode &nd = t.nodes[i];
        if (nd.is_leaf) continue;
        const HostNode &c0 = t.nodes[(size_t)nd.a], &c1 = t.nodes[(size_t)nd.b];
        float4 *q = &nodes[(size_t)inner_id[i] * 4];
        q[0] = make_float4(c0.lo[0], c0.hi[0], c0.lo[1], c0.hi[1]);
        q[1] = make_float4(c1.lo[0], c1.hi[0], c1.lo[1], c1.hi[1]);
        q[2] = make_float4(c0.lo[2], c0.hi[2], c1.lo[2], c1.hi[2]);
        // refs[2] = 1 marks an "irregular" node: some child corner is not finite (empty leaves keep [inf,-inf]) or lies
        // outside 0 / [2^-36, 2^60] in magnitude; the default trace then divides like the reference instead of using the
        // correction step (mr_kernels.hip: exact_quot)
        int32_t irregular = 0;
        for (const HostNode *c : {&c0, &c1})
            for (int k = 0; k < 3; k++)
                for (float v : {c->lo[k], c->hi[k]}) {
                    const float m = fabsf(v);
                    if (!(m == 0.0f || (m >= 0x1p-36f && m <= 0x1p60f))) irregular = 1;
                }
        int32_t refs[4] = {ref_of(nd.a), ref_of(nd.b), irregular, 0};
        memcpy(&q[3], refs, sizeof(refs));
    }

    // triangles pre-gathered in leaf order: A, B-A, C-A, (B-A)x(C-A)  (Triangle.cpp:143-151)
    std::vector<float4> tris((size_t)n_rec * 3, make_float4(0.f, 0.f, 0.f, 0.f));
    std::vector<uint32_t> cnt_ext(n_rec, 0), rec_prim(n_rec, MR_MISS);
    for (size_t li = 0; li < t.nodes.size(); li++) {
        const HostNode &leaf = t.nodes[li];
        if (!leaf.is_leaf) continue;
        for (int32_t j = 0; j < leaf.b; j++) {
            const uint32_t k = leaf_first[li] + (uint32_t)j;
            const uint32_t prim = t.leaf_prims[(size_t)leaf.a + (size_t)j];
            rec_prim[k] = prim;
            if (m.is_sphere(prim)) {
                const float *sp = &m.spheres[4 * (size_t)m.vi[3 * (size_t)prim + 1]];
                uint32_t tag = kSphereTag;
                float tagf;
                memcpy(&tagf, &tag, sizeof(tagf));
                tris[3 * (size_t)k + 0] = make_float4(sp[0], sp[1], sp[2], sp[3]);
                tris[3 * (size_t)k + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
                tris[3 * (size_t)k + 2] = make_float4(0.f, 0.f, 0.f, tagf);
                continue;
            }
            const float *A = &m.v[3 * (size_t)m.vi[3 * prim]];
            const float *B = &m.v[3 * (size_t)m.vi[3 * prim + 1]];
            const float *C = &m.v[3 * (size_t)m.vi[3 * prim + 2]];
            const float bx = B[0] - A[0], by = B[1] - A[1], bz = B[2] - A[2];
            const float cx = C[0] - A[0], cy = C[1] - A[1], cz = C[2] - A[2];
            const float nx = by * cz - bz * cy, ny = bz * cx - bx * cz, nz = bx * cy - by * cx;
            tris[3 * (size_t)k + 0] = make_float4(A[0], A[1], A[2], bx);
            tris[3 * (size_t)k + 1] = make_float4(by, bz, cx, cy);
            tris[3 * (size_t)k + 2] = make_float4(cz, nx, ny, nz);
        }
        if (leaf.b >= kLeafCountMask && leaf.b > 0) cnt_ext[(size_t)leaf_first[li]] = (uint32_t)leaf.b;
    }

    d.bytes = 0;
    mr_status st;
    if ((st = upload(d.nodes, nodes.data(), nodes.size(), d.bytes)) != MR_OK) return st;
    if ((st = upload(d.tris, tris.data(), tris.size(), d.bytes)) != MR_OK) return st;
    if ((st = upload(d.tri_prim, rec_prim.data(), rec_prim.size(), d.bytes)) != MR_OK) return st;
    if ((st = upload(d.leaf_cnt_ext, cnt_ext.data(), cnt_ext.size(), d.bytes)) != MR_OK) return st;
    if ((st = upload(d.v, m.v.data(), m.v.size(), d.bytes)) != MR_OK) return st;
    if ((st = upload(d.n, m.n.data(), m.n.size(), d.bytes)) != MR_OK) return st;
    if ((st = upload(d.vi, m.vi.data(), m.vi.size(), d.bytes)) != MR_OK) return st;
    if ((st = upload(d.ni, m.ni.data(), m.ni.size(), d.bytes)) != MR_OK) return st;
    d.n_spheres = m.n_spheres();
    d.n_planes = m.n_planes();
    if (d.n_spheres) {
        std::vector<float4> sp(d.n_spheres);
        for (uint32_t i = 0; i < d.n_spheres; i++)
            sp[i] = make_float4(m.spheres[4 * i], m.spheres[4 * i + 1], m.spheres[4 * i + 2], m.spheres[4 * i + 3]);
        if ((st = upload(d.spheres, sp.data(), sp.size(), d.bytes)) != MR_OK) return st;
    }
    if (d.n_planes) {
        std::vector<float4> pl(2 * (size_t)d.n_planes);
        for (uint32_t i = 0; i < d.n_planes; i++) {
            float matf;
            memcpy(&matf, &m.plane_material[i], sizeof(matf));
            pl[2 * i] = make_float4(m.planes[6 * i], m.planes[6 * i + 1], m.planes[6 * i + 2], matf);
            pl[2 * i + 1] = make_float4(m.planes[6 * i + 3], m.planes[6 * i + 4], m.planes[6 * i + 5], 0.f);
        }
        if ((st = upload(d.planes, pl.data(), pl.size(), d.bytes)) != MR_OK) return st;
    }
    const HostNode &root = t.nodes[0];
    memcpy(d.root_lo, root.lo, sizeof(d.root_lo));
    memcpy(d.root_hi, root.hi, sizeof(d.root_hi));
    d.root_ref = ref_of(0);
    d.n_inner = n_inner;
    d.n_tris = n_rec;
    d.stack_depth = t.max_depth + 1;   // pending far children (one per inner node on a root-to-leaf path, <= max_depth) + the kDone sentinel
    MR_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&s->d_stats), 2 * sizeof(unsigned long long)));
    MR_HIP_CHECK(hipMemset(s->d_stats, 0, 2 * sizeof(unsigned long long)));
    // two rings: [0, kWorkCounters) for the persistent trace kernels (zeroed before each launch), [kWorkCounters, 2 kWorkCounters) for
    // the tails of large frames (zero here, re-armed by the launch's last reader: mr_frame.hip)
    MR_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&s->d_work_counters), 2 * kWorkCounters * sizeof(unsigned long long)));
    MR_HIP_CHECK(hipMemset(s->d_work_counters, 0, 2 * kWorkCounters * sizeof(unsigned long long)));
    return upload_materials(s);
}

// default: one white Lambert = Phong(Vector3(1)) (Lambert.h:9, Phong.h:10-14: shininess 1, index 1)
mr_status upload_materials(mr_scene *s) {
    DeviceScene &d = s->dev;
    // the kernels index prim_material[] with every object of the scene: a table set before later geometry was added
    // would be read out of bounds (mr_surface.h material_id)
    if (!s->prim_material.empty() && s->prim_material.size() != s->mesh.n_triangles())
        return fail(MR_ERR_STATE, "mr_scene_set_materials covered %zu objects, the scene now holds %u: set the materials "
                                  "again after the last object was added", s->prim_material.size(), s->mesh.n_triangles());
    (void)hipFree(d.materials); (void)hipFree(d.prim_material);
    d.materials = nullptr; d.prim_material = nullptr;
    d.user_materials = s->materials.empty() ? 0u : 1u;
    d.refractive = 0;
    for (size_t i = 0; i + 10 < s->materials.size(); i += 11)
        if (s->materials[i + 6] > 0.f || s->materials[i + 7] > 0.f || s->materials[i + 8] > 0.f) d.refractive = 1;
    static const float white[11] = {1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1};
    const float *src = s->materials.empty() ? white : s->materials.data();
    const size_t nfl = s->materials.empty() ? 11 : s->materials.size();
    MR_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d.materials), nfl * sizeof(float)));
    MR_HIP_CHECK(hipMemcpy(d.materials, src, nfl * sizeof(float), hipMemcpyHostToDevice));
    if (!s->prim_material.empty()) {
        MR_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d.prim_material), s->prim_material.size() * sizeof(uint32_t)));
        MR_HIP_CHECK(hipMemcpy(d.prim_material, s->prim_material.data(), s->prim_material.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    return MR_OK;
}

mr_status ensure_stage(mr_scene *s, uint64_t n) {
    if (n <= s->stage_cap) return MR_OK;
    (void)hipFree(s->d_stage_rays); (void)hipFree(s->d_stage_hits);
    s->d_stage_rays = s->d_stage_hits = nullptr;
    s->stage_cap = 0;
    MR_HIP_CHECK(hipMalloc(&s->d_stage_rays, n * sizeof(mr_ray)));
    MR_HIP_CHECK(hipMalloc(&s->d_stage_hits, n * sizeof(mr_hit)));
    s->stage_cap = n;
    return MR_OK;
}

// copy streams and 2 events per chunk for the pipelined host-pointer trace
mr_status ensure_copy_pipeline(mr_scene *s, uint64_t n_chunks) {
    if (!s->copy_in) {
        hipStream_t a = nullptr;
        MR_HIP_CHECK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        s->copy_in = a;
    }
    if (!s->copy_out) {
        hipStream_t b = nullptr;
        MR_HIP_CHECK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
        s->copy_out = b;
    }
    while (s->stage_events.size() < 2 * n_chunks) {
        hipEvent_t e = nullptr;
        MR_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s->stage_events.push_back(e);
    }
    return MR_OK;
}

mr_status require_built(const mr_scene *s) {
    if (!s) return fail(MR_ERR_INVALID, "scene is NULL");
    if (!s->built) return fail(MR_ERR_STATE, "mr_bvh_build has not been called on this scene");
    return MR_OK;
}

mr_status require_device(const mr_scene *s) {
    mr_status st = require_built(s);
    if (st != MR_OK) return st;
    if (!s->on_device)
        return fail(MR_ERR_STATE, "scene was built host_only: nothing is resident on a device and there is no CPU fallback");
    return MR_OK;
}

}  // namespace
}  // namespace mr

using namespace mr;

extern "C" {

const char *mr_last_error(void) { return g_err; }
const char *mr_version(void) { return "miro_hip 0.2 (gfx950)"; }

mr_status mr_scene_create(int32_t device, mr_scene **out) {
    if (!out) return fail(MR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (device < 0) return fail(MR_ERR_INVALID, "device %d is negative", device);
    mr_scene *s = new (std::nothrow) mr_scene();
    if (!s) return fail(MR_ERR_NOMEM, "out of host memory");
    s->device = device;
    *out = s;
    return MR_OK;
}

mr_status mr_scene_destroy(mr_scene *s) {
    if (!s) return MR_OK;
    if (s->on_device) {
        (void)hipSetDevice(s->device);
        release_device(s);
    }
    delete s;
    return MR_OK;
}

mr_status mr_scene_add_mesh(mr_scene *s, const mr_mesh_desc *mesh) {
    if (!s || !mesh) return fail(MR_ERR_INVALID, "NULL argument");
    if (s->built) return fail(MR_ERR_STATE, "scene is immutable after mr_bvh_build");
    if ((mesh->n_vertices && !mesh->vertices) || (mesh->n_normals && !mesh->normals) ||
        (mesh->n_triangles && (!mesh->vidx || !mesh->nidx)))
        return fail(MR_ERR_INVALID, "mesh descriptor has NULL arrays");
    for (uint32_t i = 0; i < 3 * mesh->n_triangles; i++) {
        if (mesh->vidx[i] >= mesh->n_vertices) return fail(MR_ERR_INVALID, "vertex index %u out of range", mesh->vidx[i]);
        if (mesh->nidx[i] >= mesh->n_normals) return fail(MR_ERR_INVALID, "normal index %u out of range", mesh->nidx[i]);
    }
    HostMesh &m = s->mesh;
    const uint32_t vb = m.n_vertices(), nb = m.n_normals();
    m.v.insert(m.v.end(), mesh->vertices, mesh->vertices + 3 * (size_t)mesh->n_vertices);
    m.n.insert(m.n.end(), mesh->normals, mesh->normals + 3 * (size_t)mesh->n_normals);
    for (uint32_t i = 0; i < 3 * mesh->n_triangles; i++) {
        m.vi.push_back(mesh->vidx[i] + vb);
        m.ni.push_back(mesh->nidx[i] + nb);
    }
    return MR_OK;
}

mr_status mr_scene_add_obj(mr_scene *s, const char *path, const float *ctm16, uint32_t *n_triangles_out) {
    if (!s || !path) return fail(MR_ERR_INVALID, "NULL argument");
    if (s->built) return fail(MR_ERR_STATE, "scene is immutable after mr_bvh_build");
    return load_obj(path, ctm16, s->mesh, n_triangles_out);
}

mr_status mr_scene_add_triangle(mr_scene *s, const float v[9], const float n[9]) {
    static const uint32_t idx[3] = {0, 1, 2};
    mr_mesh_desc d;
    d.vertices = v; d.n_vertices = 3;
    d.normals = n;  d.n_normals = 3;
    d.vidx = idx; d.nidx = idx; d.n_triangles = 1;
    return mr_scene_add_mesh(s, &d);
}

mr_status mr_scene_add_sphere(mr_scene *s, const float center[3], float radius, uint32_t *prim_out) {
    if (!s || !center) return fail(MR_ERR_INVALID, "NULL argument");
    if (s->built) return fail(MR_ERR_STATE, "scene is immutable after mr_bvh_build");
    HostMesh &m = s->mesh;
    const uint32_t slot[3] = {kSphereSlot, m.n_spheres(), 0u};
    if (prim_out) *prim_out = m.n_triangles();
    m.vi.insert(m.vi.end(), slot, slot + 3);
    m.ni.insert(m.ni.end(), slot, slot + 3);
    m.spheres.insert(m.spheres.end(), center, center + 3);
    m.spheres.push_back(radius);
    return MR_OK;
}

mr_status mr_scene_add_plane(mr_scene *s, const float normal[3], const float origin[3], uint32_t material,
                             uint32_t *index_out) {
    if (!s || !normal || !origin) return fail(MR_ERR_INVALID, "NULL argument");
    if (s->built) return fail(MR_ERR_STATE, "scene is immutable after mr_bvh_build");
    HostMesh &m = s->mesh;
    if (index_out) *index_out = m.n_planes();
    m.planes.insert(m.planes.end(), normal, normal + 3);
    m.planes.insert(m.planes.end(), origin, origin + 3);
    m.plane_material.push_back(material);
    return MR_OK;
}

mr_status mr_bvh_build(mr_scene *s, const mr_build_opts *opts) {
    if (!s) return fail(MR_ERR_INVALID, "scene is NULL");
    uint32_t leaf = 4;
    if (opts) {
        if (opts->builder != MR_BUILD_REFERENCE) return fail(MR_ERR_INVALID, "unknown builder %u", opts->builder);
        if (opts->leaf_size) leaf = opts->leaf_size;
    }
    const bool host_only = opts && opts->host_only;
    if (!host_only) {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= 0)
            return fail(MR_ERR_HIP, "no HIP device available (%s); this library has no CPU fallback",
                        e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        if (s->device >= count) return fail(MR_ERR_INVALID, "device %d out of range [0,%d)", s->device, count);
        MR_HIP_CHECK(hipSetDevice(s->device));
    }
    if (s->on_device) release_device(s);
    s->built = false;
    s->on_device = false;
    mr_status st = build_reference_tree(s->mesh, leaf, s->tree);
    if (st != MR_OK) return st;
    s->built = true;
    if (host_only) return MR_OK;
    const uint32_t layout = opts ? opts->layout : 0u;
    if ((layout & 15u) > MR_LAYOUT_TREELETS || (layout & ~(uint32_t)(15u | MR_LAYOUT_ALIGN_LEAVES)))
        return fail(MR_ERR_INVALID, "unknown layout %u", layout);
    st = flatten_and_upload(s, layout);
    if (st != MR_OK) { release_device(s); return st; }
    s->on_device = true;
    return MR_OK;
}

mr_status mr_scene_get_info(const mr_scene *s, mr_scene_info *info) {
    if (!s || !info) return fail(MR_ERR_INVALID, "NULL argument");
    memset(info, 0, sizeof(*info));
    info->n_vertices = s->mesh.n_vertices();
    info->n_normals = s->mesh.n_normals();
    info->n_triangles = s->mesh.n_triangles();
    info->built = s->built ? 1u : 0u;
    info->device = s->device;
    if (s->built) {
        info->n_nodes = (uint32_t)s->tree.nodes.size();
        info->n_leaves = s->tree.n_leaves;
        info->max_depth = s->tree.max_depth;
        info->leaf_size = s->tree.leaf_size;
        info->device_bytes = s->dev.bytes;
    }
    return MR_OK;
}

mr_status mr_scene_get_mesh(const mr_scene *s, mr_mesh_desc *out) {
    if (!s || !out) return fail(MR_ERR_INVALID, "NULL argument");
    out->vertices = s->mesh.v.data(); out->n_vertices = s->mesh.n_vertices();
    out->normals = s->mesh.n.data();  out->n_normals = s->mesh.n_normals();
    out->vidx = s->mesh.vi.data();    out->nidx = s->mesh.ni.data();
    out->n_triangles = s->mesh.n_triangles();
    return MR_OK;
}

mr_status mr_scene_export_tree(const mr_scene *s, float *corners6, int32_t *meta3, uint32_t *leaf_prims) {
    mr_status st = require_built(s);
    if (st != MR_OK) return st;
    const HostTree &t = s->tree;
    for (size_t i = 0; i < t.nodes.size(); i++) {
        if (corners6) {
            memcpy(corners6 + 6 * i, t.nodes[i].lo, 3 * sizeof(float));
            memcpy(corners6 + 6 * i + 3, t.nodes[i].hi, 3 * sizeof(float));
        }
        if (meta3) {
            meta3[3 * i] = t.nodes[i].is_leaf;
            meta3[3 * i + 1] = t.nodes[i].a;
            meta3[3 * i + 2] = t.nodes[i].b;
        }
    }
    if (leaf_prims) memcpy(leaf_prims, t.leaf_prims.data(), t.leaf_prims.size() * sizeof(uint32_t));
    return MR_OK;
}

mr_status mr_trace(mr_scene *s, const mr_ray *rays, uint64_t n, mr_hit *hits, uint32_t flags, void *stream_v) {
    mr_status st = require_device(s);
    if (st != MR_OK) return st;
    if (n == 0) return MR_OK;
    if (!rays || !hits) return fail(MR_ERR_INVALID, "rays/hits is NULL");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    MR_HIP_CHECK(hipSetDevice(s->device));
    const bool rays_dev = flags & MR_RAYS_ON_DEVICE, hits_dev = flags & MR_HITS_ON_DEVICE;
    if ((rays_dev && (reinterpret_cast<uintptr_t>(rays) & 15)) || (hits_dev && (reinterpret_cast<uintptr_t>(hits) & 15)))
        return fail(MR_ERR_INVALID, "device ray/hit buffers must be 16-byte aligned");
    const mr_ray *d_rays = rays;
    mr_hit *d_hits = hits;
    // host buffers go through the scene's staging buffers: one such call at a time (the call is synchronous anyway)
    std::unique_lock<std::mutex> staged(s->stage_mutex, std::defer_lock);
    if (!rays_dev || !hits_dev) {
        staged.lock();
        if ((st = ensure_stage(s, n)) != MR_OK) return st;
        // batches above kStageChunk are uploaded chunk by chunk on the copy stream below; smaller ones in one piece here
        if (!rays_dev) {
            if (n <= kStageChunk)
                MR_HIP_CHECK(hipMemcpyAsync(s->d_stage_rays, rays, n * sizeof(mr_ray), hipMemcpyHostToDevice, stream));
            d_rays = static_cast<const mr_ray *>(s->d_stage_rays);
        }
        if (!hits_dev) d_hits = static_cast<mr_hit *>(s->d_stage_hits);
    }
    TraceParams p;
    p.nodes = s->dev.nodes; p.tris = s->dev.tris; p.tri_prim = s->dev.tri_prim; p.leaf_cnt_ext = s->dev.leaf_cnt_ext;
    memcpy(p.root_lo, s->dev.root_lo, sizeof(p.root_lo));
    memcpy(p.root_hi, s->dev.root_hi, sizeof(p.root_hi));
    p.root_ref = s->dev.root_ref;
    p.stack_depth = (int32_t)s->dev.stack_depth;
    p.rays = d_rays; p.hits = d_hits; p.n = n; p.n_dev = nullptr; p.stats = s->d_stats;
    p.planes = s->dev.planes; p.n_planes = s->dev.n_planes; p.n_spheres = s->dev.n_spheres;
    p.work_counter = s->d_work_counters + (s->next_counter.fetch_add(1) % kWorkCounters);
    p.order = nullptr;
    if (staged.owns_lock() && n > kStageChunk) {
        // host buffers, large batch: chunk k+1 uploads and chunk k-1 downloads while chunk k is traced (full overlap when
        // the caller's buffers are pinned -- mr_host_alloc --, since only then are the copies asynchronous to this thread)
        const uint64_t n_chunks = (n + kStageChunk - 1) / kStageChunk;
        if ((st = ensure_copy_pipeline(s, n_chunks)) != MR_OK) return st;
        // (alternating the uploads between two copy streams changes nothing: one queue already fills the link, 27 GB/s up)
        hipStream_t in = static_cast<hipStream_t>(s->copy_in), out = static_cast<hipStream_t>(s->copy_out);
        // one chunk; on any failure the three streams are drained before stage_mutex is released, so that the next
        // caller never shares the staging buffers with copies still in flight
        auto chunk = [&](uint64_t k) -> mr_status {
            const uint64_t off = k * kStageChunk, m = n - off < kStageChunk ? n - off : kStageChunk;
            hipEvent_t up = static_cast<hipEvent_t>(s->stage_events[2 * k]), done = static_cast<hipEvent_t>(s->stage_events[2 * k + 1]);
            if (!rays_dev) {
                MR_HIP_CHECK(hipMemcpyAsync(static_cast<mr_ray *>(s->d_stage_rays) + off, rays + off, m * sizeof(mr_ray), hipMemcpyHostToDevice, in));
                MR_HIP_CHECK(hipEventRecord(up, in));
                MR_HIP_CHECK(hipStreamWaitEvent(stream, up, 0));
            }
            p.rays = d_rays + off; p.hits = d_hits + off; p.n = m;
            const mr_status lst = launch_trace(p, flags, stream);
            if (lst != MR_OK) return lst;
            if (!hits_dev) {
                MR_HIP_CHECK(hipEventRecord(done, stream));
                MR_HIP_CHECK(hipStreamWaitEvent(out, done, 0));
                MR_HIP_CHECK(hipMemcpyAsync(hits + off, d_hits + off, m * sizeof(mr_hit), hipMemcpyDeviceToHost, out));
            }
            return MR_OK;
        };
        for (uint64_t k = 0; k < n_chunks && st == MR_OK; k++) st = chunk(k);
        const hipError_t e_in = hipStreamSynchronize(in), e_run = hipStreamSynchronize(stream), e_out = hipStreamSynchronize(out);
        if (st != MR_OK) return st;           // the failing call's message is the one mr_last_error() keeps
        MR_HIP_CHECK(e_in); MR_HIP_CHECK(e_run); MR_HIP_CHECK(e_out);
        return MR_OK;
    }
    if ((st = launch_trace(p, flags, stream)) != MR_OK) return st;
    if (!hits_dev) {
        MR_HIP_CHECK(hipMemcpyAsync(hits, d_hits, n * sizeof(mr_hit), hipMemcpyDeviceToHost, stream));
        MR_HIP_CHECK(hipStreamSynchronize(stream));
    } else if (!rays_dev) {
        MR_HIP_CHECK(hipStreamSynchronize(stream));   // the staged host rays may be reused by the caller
    }
    return MR_OK;
}

mr_status mr_host_alloc(void **ptr, uint64_t bytes) {
    if (!ptr) return fail(MR_ERR_INVALID, "ptr is NULL");
    *ptr = nullptr;
    if (bytes == 0) return MR_OK;
    MR_HIP_CHECK(hipHostMalloc(ptr, bytes, hipHostMallocPortable));
    return MR_OK;
}

mr_status mr_host_free(void *ptr) {
    if (ptr) MR_HIP_CHECK(hipHostFree(ptr));
    return MR_OK;
}

mr_status mr_trace_indirect(mr_scene *s, const mr_ray *d_rays, const uint64_t *d_count, uint64_t max_rays,
                            mr_hit *d_hits, uint32_t flags, void *stream_v) {
    mr_status st = require_device(s);
    if (st != MR_OK) return st;
    if (max_rays == 0) return MR_OK;
    if (!d_rays || !d_hits || !d_count) return fail(MR_ERR_INVALID, "NULL argument");
    if ((reinterpret_cast<uintptr_t>(d_rays) & 15) || (reinterpret_cast<uintptr_t>(d_hits) & 15) ||
        (reinterpret_cast<uintptr_t>(d_count) & 7))
        return fail(MR_ERR_INVALID, "device ray/hit buffers must be 16-byte aligned, the count 8-byte aligned");
    MR_HIP_CHECK(hipSetDevice(s->device));
    TraceParams p;
    p.nodes = s->dev.nodes; p.tris = s->dev.tris; p.tri_prim = s->dev.tri_prim; p.leaf_cnt_ext = s->dev.leaf_cnt_ext;
    memcpy(p.root_lo, s->dev.root_lo, sizeof(p.root_lo));
    memcpy(p.root_hi, s->dev.root_hi, sizeof(p.root_hi));
    p.root_ref = s->dev.root_ref;
    p.stack_depth = (int32_t)s->dev.stack_depth;
    p.rays = d_rays; p.hits = d_hits; p.n = max_rays;
    p.n_dev = reinterpret_cast<const unsigned long long *>(d_count);
    p.stats = s->d_stats;
    p.planes = s->dev.planes; p.n_planes = s->dev.n_planes; p.n_spheres = s->dev.n_spheres;
    p.work_counter = s->d_work_counters + (s->next_counter.fetch_add(1) % kWorkCounters);
    p.order = nullptr;
    return launch_trace(p, flags, static_cast<hipStream_t>(stream_v));
}

mr_status mr_order_by_octant(mr_scene *s, const mr_ray *d_rays, const uint8_t *d_octants, uint64_t n, uint32_t chunk_log2,
                             uint32_t *d_order, void *stream) {
    mr_status st = require_device(s);
    if (st != MR_OK) return st;
    if (n == 0) return MR_OK;
    if ((!d_rays && !d_octants) || !d_order) return fail(MR_ERR_INVALID, "NULL argument");
    if ((reinterpret_cast<uintptr_t>(d_rays) & 15) || (reinterpret_cast<uintptr_t>(d_order) & 3))
        return fail(MR_ERR_INVALID, "the ray buffer must be 16-byte aligned, the order buffer 4-byte aligned");
    MR_HIP_CHECK(hipSetDevice(s->device));
    return launch_octant_order(d_rays, d_octants, n, chunk_log2 ? chunk_log2 : 14u, d_order, static_cast<hipStream_t>(stream));
}

mr_status mr_trace_grouped(mr_scene *s, const mr_ray *d_rays, const uint8_t *d_octants, uint64_t n, mr_hit *d_hits, uint32_t *d_order,
                           uint32_t chunk_log2, uint32_t flags, void *stream_v) {
    mr_status st = require_device(s);
    if (st != MR_OK) return st;
    if (n == 0) return MR_OK;
    if (!d_rays || !d_hits || !d_order) return fail(MR_ERR_INVALID, "NULL argument");
    if ((reinterpret_cast<uintptr_t>(d_rays) & 15) || (reinterpret_cast<uintptr_t>(d_hits) & 15) || (reinterpret_cast<uintptr_t>(d_order) & 3))
        return fail(MR_ERR_INVALID, "device ray/hit buffers must be 16-byte aligned, the order buffer 4-byte aligned");
    if (flags & (MR_MATH_FAST | MR_COUNT_STATS))
        return fail(MR_ERR_INVALID, "mr_trace_grouped: flags may hold MR_TRACE_ANY, MR_MATH_PRODUCT, MR_TRACE_INCOHERENT only");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    MR_HIP_CHECK(hipSetDevice(s->device));
    if (!(chunk_log2 & MR_ORDER_GIVEN)) {
        if (chunk_log2 == 0) chunk_log2 = 14;
        st = launch_octant_order(d_rays, d_octants, n, chunk_log2, d_order, stream);
        if (st != MR_OK) return st;
    }
    TraceParams p;
    p.nodes = s->dev.nodes; p.tris = s->dev.tris; p.tri_prim = s->dev.tri_prim; p.leaf_cnt_ext = s->dev.leaf_cnt_ext;
    memcpy(p.root_lo, s->dev.root_lo, sizeof(p.root_lo));
    memcpy(p.root_hi, s->dev.root_hi, sizeof(p.root_hi));
    p.root_ref = s->dev.root_ref;
    p.stack_depth = (int32_t)s->dev.stack_depth;
    p.rays = d_rays; p.hits = d_hits; p.n = n; p.n_dev = nullptr;
    p.stats = s->d_stats;
    p.planes = s->dev.planes; p.n_planes = s->dev.n_planes; p.n_spheres = s->dev.n_spheres;
    p.work_counter = nullptr;
    p.order = d_order;
    return launch_trace(p, flags & ~(uint32_t)(MR_RAYS_ON_DEVICE | MR_HITS_ON_DEVICE | MR_TRACE_PERSISTENT), stream);
}

mr_status mr_trace_get_stats(mr_scene *s, uint64_t *box_tests, uint64_t *tri_tests, int32_t reset) {
    mr_status st = require_device(s);
    if (st != MR_OK) return st;
    MR_HIP_CHECK(hipSetDevice(s->device));
    unsigned long long h[2] = {0, 0};
    MR_HIP_CHECK(hipDeviceSynchronize());
    MR_HIP_CHECK(hipMemcpy(h, s->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    if (box_tests) *box_tests = h[0];
    if (tri_tests) *tri_tests = h[1];
    if (reset) MR_HIP_CHECK(hipMemset(s->d_stats, 0, sizeof(h)));
    return MR_OK;
}

mr_status mr_gen_eye_rays(mr_scene *s, const mr_camera *cam, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1,
                          uint32_t spp, uint32_t jitter, uint32_t seed, mr_ray *d_rays, void *stream) {
    if (!s || !cam || !d_rays) return fail(MR_ERR_INVALID, "NULL argument");
    if (W == 0 || H == 0 || spp == 0 || y1 < y0 || y1 > H) return fail(MR_ERR_INVALID, "bad image window");
    if (reinterpret_cast<uintptr_t>(d_rays) & 15) return fail(MR_ERR_INVALID, "d_rays must be 16-byte aligned");
    MR_HIP_CHECK(hipSetDevice(s->device));
    return launch_eye_rays(*cam, W, H, y0, y1, spp, jitter, seed, false, d_rays, static_cast<hipStream_t>(stream));
}

mr_status mr_gen_eye_rays_tiled(mr_scene *s, const mr_camera *cam, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1,
                                uint32_t spp, uint32_t jitter, uint32_t seed, mr_ray *d_rays, void *stream) {
    if (!s || !cam || !d_rays) return fail(MR_ERR_INVALID, "NULL argument");
    if (W == 0 || H == 0 || spp == 0 || y1 < y0 || y1 > H) return fail(MR_ERR_INVALID, "bad image window");
    if ((uint64_t)W * (y1 - y0) > 0xFFFFFFFFull) return fail(MR_ERR_INVALID, "window of more than 2^32-1 pixels");
    if (reinterpret_cast<uintptr_t>(d_rays) & 15) return fail(MR_ERR_INVALID, "d_rays must be 16-byte aligned");
    MR_HIP_CHECK(hipSetDevice(s->device));
    return launch_eye_rays(*cam, W, H, y0, y1, spp, jitter, seed, true, d_rays, static_cast<hipStream_t>(stream));
}

mr_status mr_tile_pixel_map(uint32_t W, uint32_t rows, uint32_t spp, uint32_t *pixel_of_slot) {
    if (!pixel_of_slot && (uint64_t)W * rows) return fail(MR_ERR_INVALID, "pixel_of_slot is NULL");
    if ((uint64_t)W * rows > 0xFFFFFFFFull) return fail(MR_ERR_INVALID, "window of more than 2^32-1 pixels");
    const mr::TileShape t = mr::tile_shape(spp);
    const uint32_t n = W * rows;
    for (uint32_t p = 0; p < n; p++) {
        uint32_t x, yl;
        mr::tile_decode(p, W, rows, t, x, yl);
        pixel_of_slot[p] = yl * W + x;
    }
    return MR_OK;
}

mr_status mr_gen_shadow_rays(mr_scene *s, const mr_ray *d_rays, const mr_hit *d_hits, uint64_t n, const float light[3],
                             mr_ray *d_out, uint32_t *d_src, uint64_t *d_count, void *stream) {
    mr_status st = require_device(s);
    if (st != MR_OK) return st;
    if (!d_hits || !d_out || !d_count || !light) return fail(MR_ERR_INVALID, "NULL argument");
    if (n > 0xFFFFFFFFull) return fail(MR_ERR_INVALID, "at most 2^32-1 rays per shadow batch");
    MR_HIP_CHECK(hipSetDevice(s->device));
    return launch_shadow_rays(s->dev, d_rays, d_hits, n, light, d_out, d_src,
                              reinterpret_cast<unsigned long long *>(d_count), static_cast<hipStream_t>(stream));
}

mr_status mr_shade_direct(mr_scene *s, const mr_ray *d_rays, const mr_hit *d_hits, uint64_t n, const mr_hit *d_shadow_hits,
                          const uint32_t *d_shadow_src, const uint64_t *d_shadow_count, const mr_light *light,
                          const float diffuse[3], uint32_t spp, float *d_rgb, void *stream) {
    mr_status st = require_device(s);
    if (st != MR_OK) return st;
    if (!d_rays || !d_hits || !d_shadow_hits || !d_shadow_src || !d_shadow_count || !light || !diffuse || !d_rgb)
        return fail(MR_ERR_INVALID, "NULL argument");
    if (spp == 0 || n % spp != 0) return fail(MR_ERR_INVALID, "n (%llu) must be a multiple of spp (%u)", (unsigned long long)n, spp);
    // this call shades a uniform material and knows occluders as a flag per ray (opaque occluders, see the header): in a
    // scene with a refractive material the light behind such an occluder is attenuated, not removed (Phong.cpp:99-113) --
    // mr_shade_accumulate (batched) and mr_render_direct / mr_trace_level (one launch) implement that
    if (s->dev.refractive)
        return fail(MR_ERR_STATE, "mr_shade_direct: the scene has a refractive material; use mr_shade_accumulate, mr_render_direct "
                                  "or mr_trace_level, which let light through refractive occluders as Phong.cpp:99-113 does");
    MR_HIP_CHECK(hipSetDevice(s->device));
    if (n > s->occluded_cap) {     // grow-only scratch; size it with a first call outside any graph capture
        (void)hipFree(s->d_occluded);
        s->d_occluded = nullptr;
        s->occluded_cap = 0;
        MR_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&s->d_occluded), n));
        s->occluded_cap = n;
    }
    return launch_shade(s->dev, d_rays, d_hits, n, d_shadow_hits, d_shadow_src,
                        reinterpret_cast<const unsigned long long *>(d_shadow_count), s->d_occluded, *light, diffuse, spp,
                        d_rgb, static_cast<hipStream_t>(stream));
}

mr_status mr_band_locate(uint32_t H, uint32_t band_rows, uint32_t world, uint32_t y, uint32_t *rank, uint32_t *local_row) {
    if (band_rows == 0 || world == 0 || y >= H) return fail(MR_ERR_INVALID, "mr_band_locate: row %u of %u, bands of %u over %u ranks", y, H, band_rows, world);
    const uint32_t band = y / band_rows;
    if (rank) *rank = band % world;
    if (local_row) *local_row = (band / world) * band_rows + y % band_rows;
    return MR_OK;
}

mr_status mr_band_rows_of(uint32_t H, uint32_t band_rows, uint32_t rank, uint32_t world, uint32_t *rows) {
    if (band_rows == 0 || world == 0 || rank >= world || !rows) return fail(MR_ERR_INVALID, "mr_band_rows_of: bad arguments");
    uint32_t n = 0;
    const uint32_t nb = (H + band_rows - 1) / band_rows;
    for (uint32_t b = rank; b < nb; b += world) n += (b + 1) * band_rows <= H ? band_rows : H - b * band_rows;
    *rows = n;
    return MR_OK;
}

mr_status mr_deinterleave_bands(mr_scene *s, const float *d_recv, float *d_full, uint32_t W, uint32_t H, uint32_t band_rows,
                                uint32_t world, uint32_t shard_rows, uint32_t floats_per_pixel, void *stream) {
    mr_status st = require_device(s);
    if (st != MR_OK) return st;
    if (!d_recv || !d_full) return fail(MR_ERR_INVALID, "NULL argument");
    if (band_rows == 0 || world == 0 || floats_per_pixel == 0) return fail(MR_ERR_INVALID, "mr_deinterleave_bands: bad band description");
    for (uint32_t r = 0; r < world; r++) {
        uint32_t rows = 0;
        mr_band_rows_of(H, band_rows, r, world, &rows);
        if (rows > shard_rows) return fail(MR_ERR_INVALID, "rank %u owns %u rows, the shards hold %u", r, rows, shard_rows);
    }
    MR_HIP_CHECK(hipSetDevice(s->device));
    return launch_deinterleave(d_recv, d_full, W, H, band_rows, world, shard_rows, floats_per_pixel, static_cast<hipStream_t>(stream));
}

mr_status mr_render_direct(mr_scene *s, const mr_frame_desc *frame, float *d_rgb, mr_hit *d_hits, mr_hit *d_shadow_hits,
                           uint64_t *d_counts, void *stream) {
    mr_status st = require_device(s);
    if (st != MR_OK) return st;
    if (!frame || !d_rgb) return fail(MR_ERR_INVALID, "NULL argument");
    if (frame->W == 0 || frame->H == 0) return fail(MR_ERR_INVALID, "empty image");
    if ((reinterpret_cast<uintptr_t>(d_hits) & 15) || (reinterpret_cast<uintptr_t>(d_shadow_hits) & 15) ||
        (reinterpret_cast<uintptr_t>(d_counts) & 7) || (reinterpret_cast<uintptr_t>(d_rgb) & 3))
        return fail(MR_ERR_INVALID, "hit buffers must be 16-byte aligned, counters 8-byte aligned");
    mr_frame_desc fd = *frame;
    if (fd.band_world <= 1) { fd.band_world = 1; fd.band_rank = 0; if (fd.band_rows == 0) fd.band_rows = 1; }
    MR_HIP_CHECK(hipSetDevice(s->device));
    return launch_frame(s->dev, fd, d_rgb, d_hits, d_shadow_hits, reinterpret_cast<unsigned long long *>(d_counts),
                        s->d_work_counters + kWorkCounters + (s->next_counter.fetch_add(1) % kWorkCounters), static_cast<hipStream_t>(stream));
}

mr_status mr_scene_set_materials(mr_scene *s, const mr_material *mats, uint32_t n_mats, const uint32_t *prim_material) {
    if (!s || !mats || n_mats == 0) return fail(MR_ERR_INVALID, "NULL argument or no materials");
    const uint32_t nt = s->mesh.n_triangles();
    if (prim_material)
        for (uint32_t i = 0; i < nt; i++)
            if (prim_material[i] >= n_mats) return fail(MR_ERR_INVALID, "triangle %u has material %u of %u", i, prim_material[i], n_mats);
    for (uint32_t i = 0; i < s->mesh.n_planes(); i++)
        if (s->mesh.plane_material[i] >= n_mats)
            return fail(MR_ERR_INVALID, "plane %u has material %u of %u", i, s->mesh.plane_material[i], n_mats);
    s->materials.resize(11 * (size_t)n_mats);
    for (uint32_t i = 0; i < n_mats; i++) {
        float *o = &s->materials[11 * (size_t)i];
        // energy balance of the Phong constructor (Phong.cpp:12-33)
        for (int c = 0; c < 3; c++) {
            const float ks = mats[i].specular[c];
            const float kt = fmaxf(fminf(mats[i].transmission[c], 1.0f - ks), 0.f);
            const float kd = fmaxf(fminf(mats[i].diffuse[c], 1.0f - ks - kt), 0.f);
            o[c] = kd; o[3 + c] = ks; o[6 + c] = kt;
        }
        o[9] = mats[i].shininess;
        o[10] = mats[i].refract_index;
    }
    if (prim_material) s->prim_material.assign(prim_material, prim_material + nt);
    else s->prim_material.clear();
    if (s->on_device) {
        MR_HIP_CHECK(hipSetDevice(s->device));
        MR_HIP_CHECK(hipDeviceSynchronize());
        return upload_materials(s);
    }
    return MR_OK;
}

mr_status mr_shade_accumulate(mr_scene *s, const mr_ray *d_rays, const mr_hit *d_hits, const float *d_weights,
                              const uint32_t *d_pixels, uint64_t n, const mr_ray *d_shadow_rays, const mr_hit *d_shadow_hits,
                              const uint32_t *d_shadow_src, const uint64_t *d_shadow_count, const mr_light *light, uint32_t spp,
                              float *d_rgb, void *stream) {
    mr_status st = require_device(s);
    if (st != MR_OK) return st;
    if (!d_rays || !d_hits || !d_shadow_rays || !d_shadow_hits || !d_shadow_src || !d_shadow_count || !light || !d_rgb)
        return fail(MR_ERR_INVALID, "NULL argument");
    if (spp == 0) return fail(MR_ERR_INVALID, "spp is 0");
    MR_HIP_CHECK(hipSetDevice(s->device));
    if (n > s->light_scale_cap) {
        (void)hipFree(s->d_light_scale);
        s->d_light_scale = nullptr;
        s->light_scale_cap = 0;
        MR_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&s->d_light_scale), n * sizeof(float)));
        s->light_scale_cap = n;
    }
    return launch_shade_accumulate(s->dev, d_rays, d_hits, d_weights, d_pixels, n, d_shadow_rays, d_shadow_hits, d_shadow_src,
                                   reinterpret_cast<const unsigned long long *>(d_shadow_count), s->d_light_scale, *light, spp,
                                   d_rgb, static_cast<hipStream_t>(stream));
}

mr_status mr_gen_secondary_rays(mr_scene *s, const mr_ray *d_rays, const mr_hit *d_hits, const float *d_weights,
                                const uint32_t *d_pixels, uint64_t n, uint32_t spp, mr_ray *d_out_rays, float *d_out_weights,
                                uint32_t *d_out_pixels, uint64_t *d_count, uint64_t out_capacity, uint8_t *d_out_octants, void *stream) {
    mr_status st = require_device(s);
    if (st != MR_OK) return st;
    if (!d_rays || !d_hits || !d_out_rays || !d_out_weights || !d_out_pixels || !d_count) return fail(MR_ERR_INVALID, "NULL argument");
    if (spp == 0) return fail(MR_ERR_INVALID, "spp is 0");
    if (n / spp > 0xFFFFFFFFull) return fail(MR_ERR_INVALID, "too many pixels");
    if (out_capacity == 0 && n > 0) return fail(MR_ERR_INVALID, "mr_gen_secondary_rays: out_capacity is 0 (room for 3n rays always suffices)");
    MR_HIP_CHECK(hipSetDevice(s->device));
    return launch_secondary_rays(s->dev, d_rays, d_hits, d_weights, d_pixels, n, spp, d_out_rays, d_out_weights, d_out_pixels,
                                 reinterpret_cast<unsigned long long *>(d_count), out_capacity, d_out_octants, static_cast<hipStream_t>(stream));
}

mr_status mr_gen_path_rays(mr_scene *s, const mr_ray *d_rays, const mr_hit *d_hits, const float *d_weights,
                           const uint32_t *d_pixels, const uint32_t *d_ids, uint64_t n, uint32_t spp, uint32_t seed,
                           uint32_t bounce, uint32_t kinds, mr_ray *d_out_rays, float *d_out_weights, uint32_t *d_out_pixels,
                           uint32_t *d_out_ids, uint64_t *d_count, uint64_t out_capacity, uint8_t *d_out_octants, void *stream) {
    mr_status st = require_device(s);
    if (st != MR_OK) return st;
    if (!d_rays || !d_hits || !d_out_rays || !d_out_weights || !d_out_pixels || !d_count) return fail(MR_ERR_INVALID, "NULL argument");
    if (spp == 0) return fail(MR_ERR_INVALID, "spp is 0");
    if (n / spp > 0xFFFFFFFFull || n > 0xFFFFFFFFull) return fail(MR_ERR_INVALID, "too many rays for 32-bit ray ids");
    if (kinds == 0 || (kinds & ~7u)) return fail(MR_ERR_INVALID, "kinds must be a combination of MR_PATH_MIRROR | MR_PATH_REFRACT | MR_PATH_DIFFUSE");
    if (out_capacity == 0 && n > 0) return fail(MR_ERR_INVALID, "mr_gen_path_rays: out_capacity is 0 (room for 4n rays always suffices)");
    MR_HIP_CHECK(hipSetDevice(s->device));
    return launch_path_rays(s->dev, d_rays, d_hits, d_weights, d_pixels, d_ids, n, spp, seed, bounce, kinds, d_out_rays,
                            d_out_weights, d_out_pixels, d_out_ids, reinterpret_cast<unsigned long long *>(d_count), out_capacity,
                            d_out_octants, static_cast<hipStream_t>(stream));
}

mr_status mr_trace_level(mr_scene *s, const mr_level_desc *level, const mr_ray *d_rays, const float *d_weights,
                         const uint32_t *d_pixels, const uint32_t *d_ids, uint64_t n, float *d_rgb, mr_ray *d_out_rays,
                         float *d_out_weights, uint32_t *d_out_pixels, uint32_t *d_out_ids, uint64_t *d_out_count,
                         uint64_t *d_counts, void *stream) {
    mr_status st = require_device(s);
    if (st != MR_OK) return st;
    if (!level || !d_rays || !d_rgb) return fail(MR_ERR_INVALID, "NULL argument");
    if (level->children > MR_LEVEL_PATH) return fail(MR_ERR_INVALID, "mr_trace_level: children must be MR_LEVEL_LAST, _SPECULAR or _PATH");
    if (level->children != MR_LEVEL_LAST && (!d_out_rays || !d_out_weights || !d_out_pixels || !d_out_count))
        return fail(MR_ERR_INVALID, "mr_trace_level: a level with children needs the output queue");
    if (level->children != MR_LEVEL_LAST && n > 0 && level->out_capacity_lo == 0 && level->out_capacity_hi == 0)
        return fail(MR_ERR_INVALID, "mr_trace_level: out_capacity is 0 (room for %un rays always suffices)", level->children == MR_LEVEL_PATH ? 4u : 3u);
    if (level->spp == 0) return fail(MR_ERR_INVALID, "spp is 0");
    if (n > 0xFFFFFFFFull) return fail(MR_ERR_INVALID, "mr_trace_level: at most 2^32 - 1 rays per call");
    if (level->children == MR_LEVEL_PATH && (level->path_kinds == 0 || (level->path_kinds & ~7u)))
        return fail(MR_ERR_INVALID, "path_kinds must be a combination of MR_PATH_MIRROR | MR_PATH_REFRACT | MR_PATH_DIFFUSE");
    if (level->flags & ~(uint32_t)(MR_MATH_PRODUCT | MR_TRACE_INCOHERENT))
        return fail(MR_ERR_INVALID, "mr_trace_level: flags may hold MR_MATH_PRODUCT, MR_TRACE_INCOHERENT only");
    if ((reinterpret_cast<uintptr_t>(d_rays) & 15) || (reinterpret_cast<uintptr_t>(d_out_rays) & 15) ||
        (reinterpret_cast<uintptr_t>(d_out_count) & 7) || (reinterpret_cast<uintptr_t>(d_counts) & 7))
        return fail(MR_ERR_INVALID, "ray queues must be 16-byte aligned, counters 8-byte aligned");
    if (level->reserved != 0) return fail(MR_ERR_INVALID, "mr_level_desc.reserved must be 0");
    if (reinterpret_cast<uintptr_t>(level->d_order) & 3) return fail(MR_ERR_INVALID, "mr_level_desc.d_order must be 4-byte aligned");
    MR_HIP_CHECK(hipSetDevice(s->device));
    return launch_level(s->dev, *level, d_rays, d_weights, d_pixels, d_ids, n, d_rgb, d_out_rays, d_out_weights, d_out_pixels,
                        d_out_ids, reinterpret_cast<unsigned long long *>(d_out_count),
                        reinterpret_cast<unsigned long long *>(d_counts), static_cast<hipStream_t>(stream));
}

mr_status mr_tonemap(mr_scene *s, const float *d_rgb, uint64_t n_values, uint8_t *d_out, void *stream) {
    if (!s || !d_rgb || !d_out) return fail(MR_ERR_INVALID, "NULL argument");
    MR_HIP_CHECK(hipSetDevice(s->device));
    return launch_tonemap(d_rgb, n_values, d_out, static_cast<hipStream_t>(stream));
}

mr_status mr_untile_pixels(mr_scene *s, const float *d_slots, float *d_image, uint32_t W, uint32_t rows, uint32_t spp,
                           uint32_t channels, void *stream) {
    if (!s || !d_slots || !d_image) return fail(MR_ERR_INVALID, "NULL argument");
    if (d_slots == d_image) return fail(MR_ERR_INVALID, "mr_untile_pixels does not work in place");
    if (channels == 0 || spp == 0) return fail(MR_ERR_INVALID, "channels and spp must be positive");
    if ((uint64_t)W * rows > 0xFFFFFFFFull) return fail(MR_ERR_INVALID, "window of more than 2^32-1 pixels");
    MR_HIP_CHECK(hipSetDevice(s->device));
    return launch_untile(d_slots, d_image, W, rows, spp, channels, static_cast<hipStream_t>(stream));
}

mr_status mr_hit_attrs(mr_scene *s, const mr_ray *d_rays, const mr_hit *d_hits, uint64_t n, float *d_P, float *d_N,
                       void *stream) {
    mr_status st = require_device(s);
    if (st != MR_OK) return st;
    if (!d_hits) return fail(MR_ERR_INVALID, "d_hits is NULL");
    MR_HIP_CHECK(hipSetDevice(s->device));
    return launch_hit_attrs(s->dev, d_rays, d_hits, n, d_P, d_N, static_cast<hipStream_t>(stream));
}

}  // extern "C"
