// mr_photon.cpp -- host side of the photon map (config 5 of BASELINE.json): the reference's Photon_map
// (PhotonMap.h:42-105; Jensen's kd-tree as vendored and modified by the reference) behind the C ABI:
//   mr_photon_map_store    Photon_map::store              PhotonMap.cpp:255-289  (direction quantised to 2 bytes)
//   mr_photon_map_scale    Photon_map::scale_photon_power PhotonMap.cpp:298-306
//   mr_photon_map_balance  Photon_map::balance            PhotonMap.cpp:314-359,409-476 (left-balanced kd-tree in
//                          heap order; split axis = longest extent of the running bounding box) + upload
//   mr_irradiance_estimate Photon_map::irradiance_estimate PhotonMap.cpp:81-145 -> csrc/mr_photon.hip
// Own structure: photons live in SoA vectors; the balance works on an index permutation with std::nth_element
// under a total order (coordinate, then storage index), which yields the unique left-balanced tree; the device
// gets three float4 planes in heap order (position+axis, de-quantised direction, power).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <numeric>

#include "mr_internal.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

using namespace mr;

namespace {

struct DirTables {
    float costheta[256], sintheta[256], cosphi[256], sinphi[256];
    DirTables() {
        for (int i = 0; i < 256; i++) {                            // PhotonMap.cpp:47-53
            const double angle = double(i) * (1.0 / 256.0) * M_PI;
            costheta[i] = (float)cos(angle);
            sintheta[i] = (float)sin(angle);
            cosphi[i] = (float)cos(2.0 * angle);
            sinphi[i] = (float)sin(2.0 * angle);
        }
    }
};
const DirTables &tables() { static DirTables t; return t; }

}  // namespace

struct mr_photon_map {
    int32_t device = 0;
    uint32_t max_photons = 0;
    // storage order (1-based in the reference; 0-based here)
    std::vector<float> pos, power;           // xyz triples
    std::vector<uint8_t> theta, phi;
    uint32_t prev_scale = 0;
    float bbox_min[3] = {1e8f, 1e8f, 1e8f}, bbox_max[3] = {-1e8f, -1e8f, -1e8f};
    // heap order after balance: heap[i] = storage index of the photon at kd-tree node i+1, with its split axis
    std::vector<uint32_t> heap;
    std::vector<int16_t> plane;
    bool balanced = false, on_device = false;
    PhotonMapDev dev;
    unsigned long long *d_stats = nullptr;   // work counters of the estimates (mr_photon_map_count_stats), else NULL
    std::atomic<uint32_t> next_counter{0};   // which of dev.work_counters the next estimate launch takes
    uint32_t count() const { return (uint32_t)theta.size(); }
};

namespace {

// the median rule of balance_segment (PhotonMap.cpp:421-430): left-balanced split of [start, end]
inline int left_balanced_median(int start, int end) {
    const int n = end - start + 1;
    int median = 1;
    while (4 * median <= n) median += median;
    if (3 * median <= n) { median += median; median += start - 1; }
    else median = end - median + 1;
    return median;
}

struct Balancer {
    mr_photon_map &m;
    std::vector<uint32_t> org;      // permutation being partitioned, 1-based segment positions
    float bmin[3], bmax[3];

    explicit Balancer(mr_photon_map &pm) : m(pm) {
        org.resize(m.count() + 1);
        std::iota(org.begin(), org.end(), 0u);      // org[p] = storage index p-1 for p >= 1 (slot 0 unused)
        for (uint32_t p = 1; p < org.size(); p++) org[p] = p - 1;
        memcpy(bmin, m.bbox_min, sizeof(bmin));
        memcpy(bmax, m.bbox_max, sizeof(bmax));
    }

    void segment(int index, int start, int end) {
        const int median = left_balanced_median(start, end);
        int axis = 2;                                               // PhotonMap.cpp:436-441
        if ((bmax[0] - bmin[0]) > (bmax[1] - bmin[1]) && (bmax[0] - bmin[0]) > (bmax[2] - bmin[2])) axis = 0;
        else if ((bmax[1] - bmin[1]) > (bmax[2] - bmin[2])) axis = 1;
        const float *pos = m.pos.data();
        std::nth_element(org.begin() + start, org.begin() + median, org.begin() + end + 1,
                         [pos, axis](uint32_t a, uint32_t b) {
                             const float pa = pos[3 * (size_t)a + axis], pb = pos[3 * (size_t)b + axis];
                             return pa != pb ? pa < pb : a < b;
                         });
        m.heap[index - 1] = org[median];
        m.plane[index - 1] = (int16_t)axis;
        const float split = pos[3 * (size_t)org[median] + axis];
        if (median > start) {
            if (start < median - 1) {
                const float keep = bmax[axis];
                bmax[axis] = split;
                segment(2 * index, start, median - 1);
                bmax[axis] = keep;
            } else {
                m.heap[2 * index - 1] = org[start];
            }
        }
        if (median < end) {
            if (median + 1 < end) {
                const float keep = bmin[axis];
                bmin[axis] = split;
                segment(2 * index + 1, median + 1, end);
                bmin[axis] = keep;
            } else {
                m.heap[2 * index] = org[end];
            }
        }
    }
};

void release(mr_photon_map *m) {
    (void)hipFree(m->dev.rec); (void)hipFree(m->dev.power); (void)hipFree(m->dev.boxes); (void)hipFree(m->dev.work_counters); (void)hipFree(m->d_stats);
    m->dev = PhotonMapDev();
    m->d_stats = nullptr;
    m->on_device = false;
}

}  // namespace

extern "C" {

mr_status mr_photon_map_create(int32_t device, uint32_t max_photons, mr_photon_map **out) {
    if (!out) return fail(MR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (device < 0) return fail(MR_ERR_INVALID, "device %d is negative", device);
    mr_photon_map *m = new (std::nothrow) mr_photon_map();
    if (!m) return fail(MR_ERR_NOMEM, "out of host memory");
    m->device = device;
    m->max_photons = max_photons;
    *out = m;
    return MR_OK;
}

mr_status mr_photon_map_destroy(mr_photon_map *m) {
    if (!m) return MR_OK;
    if (m->on_device) { (void)hipSetDevice(m->device); release(m); }
    delete m;
    return MR_OK;
}

mr_status mr_photon_map_store(mr_photon_map *m, uint32_t n, const float *power, const float *pos, const float *dir) {
    if (!m || (n && (!power || !pos || !dir))) return fail(MR_ERR_INVALID, "NULL argument");
    if (m->balanced) return fail(MR_ERR_STATE, "photon map is immutable after mr_photon_map_balance");
    for (uint32_t i = 0; i < n; i++) {
        if (m->count() >= m->max_photons) break;                    // PhotonMap.cpp:260-261: silently full
        for (int k = 0; k < 3; k++) {
            const float p = pos[3 * (size_t)i + k];
            m->pos.push_back(p);
            if (p < m->bbox_min[k]) m->bbox_min[k] = p;
            if (p > m->bbox_max[k]) m->bbox_max[k] = p;
            m->power.push_back(power[3 * (size_t)i + k]);
        }
        const int theta = int(acos(dir[3 * (size_t)i + 2]) * (256.0 / M_PI));
        m->theta.push_back(theta > 255 ? 255 : (uint8_t)theta);
        const int phi = int(atan2(dir[3 * (size_t)i + 1], dir[3 * (size_t)i]) * (256.0 / (2.0 * M_PI)));
        m->phi.push_back(phi > 255 ? 255 : (phi < 0 ? (uint8_t)(phi + 256) : (uint8_t)phi));
    }
    return MR_OK;
}

mr_status mr_photon_map_scale(mr_photon_map *m, float scale) {
    if (!m) return fail(MR_ERR_INVALID, "photon map is NULL");
    if (m->balanced) return fail(MR_ERR_STATE, "photon map is immutable after mr_photon_map_balance");
    // the reference's loop runs from prev_scale (1-based, initially 1) to stored: it re-scales the last photon of
    // the previous batch (PhotonMap.cpp:301-305); kept
    const uint32_t from = m->prev_scale == 0 ? 0 : m->prev_scale - 1;
    for (uint32_t i = from; i < m->count(); i++)
        for (int k = 0; k < 3; k++) m->power[3 * (size_t)i + k] *= scale;
    m->prev_scale = m->count();
    return MR_OK;
}

mr_status mr_photon_map_balance(mr_photon_map *m, uint32_t host_only) {
    if (!m) return fail(MR_ERR_INVALID, "photon map is NULL");
    const uint32_t n = m->count();
    m->heap.assign(n, 0);
    m->plane.assign(n, 0);
    if (n > 1) {
        Balancer b(*m);
        b.segment(1, 1, (int)n);
    } else if (n == 1) {
        m->heap[0] = 0;
    }
    m->balanced = true;
    if (host_only) return MR_OK;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(MR_ERR_HIP, "no HIP device available (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (m->device >= count) return fail(MR_ERR_INVALID, "device %d out of range [0,%d)", m->device, count);
    MR_HIP_CHECK(hipSetDevice(m->device));
    if (m->on_device) release(m);
    // heap-ordered planes, 1-based (slot 0 unused) so that children of i are 2i, 2i+1
    std::vector<float4> a(n + 1), d(n + 1), p(n + 1);
    const DirTables &t = tables();
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t s = m->heap[i];
        float w;
        const int32_t pl = m->plane[i];
        memcpy(&w, &pl, 4);
        a[i + 1] = make_float4(m->pos[3 * (size_t)s], m->pos[3 * (size_t)s + 1], m->pos[3 * (size_t)s + 2], w);
        const uint8_t th = m->theta[s], ph = m->phi[s];
        d[i + 1] = make_float4(t.sintheta[th] * t.cosphi[ph], t.sintheta[th] * t.sinphi[ph], t.costheta[th], 0.0f);   // :66-71
        p[i + 1] = make_float4(m->power[3 * (size_t)s], m->power[3 * (size_t)s + 1], m->power[3 * (size_t)s + 2], 0.0f);
    }
    a[0] = d[0] = p[0] = make_float4(0, 0, 0, 0);
    const size_t bytes = (size_t)(n + 1) * sizeof(float4);
    std::vector<float4> rec(2 * (size_t)(n + 1));
    for (uint32_t i = 0; i <= n; i++) { rec[2 * (size_t)i] = a[i]; rec[2 * (size_t)i + 1] = d[i]; }
    MR_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&m->dev.rec), 2 * bytes));
    MR_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&m->dev.power), bytes));
    MR_HIP_CHECK(hipMemcpy(m->dev.rec, rec.data(), 2 * bytes, hipMemcpyHostToDevice));
    MR_HIP_CHECK(hipMemcpy(m->dev.power, p.data(), bytes, hipMemcpyHostToDevice));
    m->dev.n = (int32_t)n;
    m->dev.half = (int32_t)n / 2 - 1;                               // half_stored_photons, PhotonMap.cpp:357
    // block boxes: bounds of every subtree (bottom-up over the heap), then per block root its own six levels and its subtree
    {
        const float inf = INFINITY;
        std::vector<float> slo(3 * (size_t)(n + 1), inf), shi(3 * (size_t)(n + 1), -inf);
        for (uint32_t j = n; j >= 1; j--) {
            const float pj[3] = {a[j].x, a[j].y, a[j].z};
            for (int c = 0; c < 3; c++) {
                float lo = pj[c], hi = pj[c];
                for (uint32_t ch = 2 * j; ch <= 2 * j + 1 && ch <= n; ch++) {
                    lo = std::min(lo, slo[3 * (size_t)ch + c]);
                    hi = std::max(hi, shi[3 * (size_t)ch + c]);
                }
                slo[3 * (size_t)j + c] = lo; shi[3 * (size_t)j + c] = hi;
            }
        }
        int layers = 0;
        size_t total = 0;
        for (uint64_t first = 1; first <= n && layers < 4; first <<= 6) { m->dev.layer_base[layers++] = (int32_t)total; total += (size_t)first; }
        m->dev.layers = layers;
        std::vector<float4> boxes(4 * std::max<size_t>(total, 1), make_float4(inf, inf, inf, 0.f));
        for (size_t i = 0; i < boxes.size(); i += 2) boxes[i + 1] = make_float4(-inf, -inf, -inf, 0.f);
        uint64_t first = 1;
        for (int L = 0; L < layers; L++, first <<= 6) {
            for (uint64_t r = first; r < 2 * first && r <= n; r++) {
                float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
                for (int l = 0; l < 6; l++)
                    for (uint64_t o = 0; o < (1ull << l); o++) {
                        const uint64_t j = (r << l) + o;
                        if (j > n) break;
                        const float pj[3] = {a[j].x, a[j].y, a[j].z};
                        for (int c = 0; c < 3; c++) { lo[c] = std::min(lo[c], pj[c]); hi[c] = std::max(hi[c], pj[c]); }
                    }
                float4 *b = &boxes[4 * ((size_t)m->dev.layer_base[L] + (size_t)(r - first))];
                b[0] = make_float4(lo[0], lo[1], lo[2], 0.f);
                b[1] = make_float4(hi[0], hi[1], hi[2], 0.f);
                b[2] = make_float4(slo[3 * r], slo[3 * r + 1], slo[3 * r + 2], 0.f);
                b[3] = make_float4(shi[3 * r], shi[3 * r + 1], shi[3 * r + 2], 0.f);
            }
        }
        MR_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&m->dev.boxes), boxes.size() * sizeof(float4)));
        MR_HIP_CHECK(hipMemcpy(m->dev.boxes, boxes.data(), boxes.size() * sizeof(float4), hipMemcpyHostToDevice));
    }
    if (!m->dev.work_counters) {
        MR_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&m->dev.work_counters), kPhotonWorkCounters * sizeof(unsigned)));
        MR_HIP_CHECK(hipMemset(m->dev.work_counters, 0, kPhotonWorkCounters * sizeof(unsigned)));
    }
    m->on_device = true;
    return MR_OK;
}

mr_status mr_photon_map_count(const mr_photon_map *m, uint32_t *stored) {
    if (!m || !stored) return fail(MR_ERR_INVALID, "NULL argument");
    *stored = m->count();
    return MR_OK;
}

mr_status mr_photon_map_export(const mr_photon_map *m, float *pos, int32_t *plane, uint8_t *theta_phi, float *power) {
    if (!m) return fail(MR_ERR_INVALID, "photon map is NULL");
    if (!m->balanced) return fail(MR_ERR_STATE, "mr_photon_map_balance has not been called");
    for (uint32_t i = 0; i < m->count(); i++) {
        const uint32_t s = m->heap[i];
        for (int k = 0; k < 3; k++) {
            if (pos) pos[3 * (size_t)i + k] = m->pos[3 * (size_t)s + k];
            if (power) power[3 * (size_t)i + k] = m->power[3 * (size_t)s + k];
        }
        if (plane) plane[i] = m->plane[i];
        if (theta_phi) { theta_phi[2 * (size_t)i] = m->theta[s]; theta_phi[2 * (size_t)i + 1] = m->phi[s]; }
    }
    return MR_OK;
}

mr_status mr_irradiance_estimate(mr_photon_map *m, const float *d_pos, const float *d_normal, uint64_t n_queries,
                                 float max_dist, uint32_t nphotons, float *d_irrad, int32_t *d_found, float *d_r2,
                                 void *stream) {
    if (!m) return fail(MR_ERR_INVALID, "photon map is NULL");
    if (!m->balanced) return fail(MR_ERR_STATE, "mr_photon_map_balance has not been called");
    if (!m->on_device) return fail(MR_ERR_STATE, "photon map was balanced host_only: nothing is resident on a device and there is no CPU fallback");
    if (n_queries == 0) return MR_OK;
    if (!d_pos || !d_normal || !d_irrad) return fail(MR_ERR_INVALID, "NULL argument");
    if (nphotons == 0 || nphotons > kKnnMaxK) return fail(MR_ERR_INVALID, "nphotons must be in [1, %d]", kKnnMaxK);
    MR_HIP_CHECK(hipSetDevice(m->device));
    return launch_irradiance(m->dev, m->dev.work_counters + (m->next_counter.fetch_add(1) % kPhotonWorkCounters), d_pos, d_normal, n_queries, max_dist, nphotons, d_irrad, d_found, d_r2, m->d_stats,
                             static_cast<hipStream_t>(stream));
}

mr_status mr_photon_map_count_stats(mr_photon_map *m, int32_t enable) {
    if (!m) return fail(MR_ERR_INVALID, "photon map is NULL");
    if (!m->on_device) return fail(MR_ERR_STATE, "photon map is not resident on a device");
    MR_HIP_CHECK(hipSetDevice(m->device));
    if (enable && !m->d_stats) {
        MR_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&m->d_stats), kPhotonStats * sizeof(unsigned long long)));
        MR_HIP_CHECK(hipMemset(m->d_stats, 0, kPhotonStats * sizeof(unsigned long long)));
    } else if (!enable && m->d_stats) {
        MR_HIP_CHECK(hipDeviceSynchronize());
        (void)hipFree(m->d_stats);
        m->d_stats = nullptr;
    }
    return MR_OK;
}

mr_status mr_photon_map_get_stats(mr_photon_map *m, uint64_t out[12], int32_t reset) {
    if (!m || !out) return fail(MR_ERR_INVALID, "NULL argument");
    if (!m->d_stats) return fail(MR_ERR_STATE, "mr_photon_map_count_stats has not been enabled on this map");
    MR_HIP_CHECK(hipSetDevice(m->device));
    MR_HIP_CHECK(hipDeviceSynchronize());
    unsigned long long h[kPhotonStats];
    MR_HIP_CHECK(hipMemcpy(h, m->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < kPhotonStats; i++) out[i] = h[i];
    if (reset) MR_HIP_CHECK(hipMemset(m->d_stats, 0, sizeof(h)));
    return MR_OK;
}

mr_status mr_final_gather(mr_scene *s, mr_photon_map *global_map, mr_photon_map *caustic_map, const mr_ray *d_rays,
                          const mr_hit *d_hits, uint64_t n, float max_dist, uint32_t nphotons, uint32_t spp,
                          float *d_scratch, float *d_rgb, void *stream_v) {
    if (!s || !s->built || !s->on_device) return fail(MR_ERR_STATE, "scene is not resident on a device");
    if (!d_rays || !d_hits || !d_scratch || !d_rgb) return fail(MR_ERR_INVALID, "NULL argument");
    if (spp == 0 || n % spp != 0) return fail(MR_ERR_INVALID, "n (%llu) must be a multiple of spp (%u)", (unsigned long long)n, spp);
    if (nphotons == 0 || nphotons > kKnnMaxK) return fail(MR_ERR_INVALID, "nphotons must be in [1, %d]", kKnnMaxK);
    mr_photon_map *maps[2] = {global_map, caustic_map};
    for (mr_photon_map *m : maps) {
        if (!m) continue;
        if (!m->balanced || !m->on_device) return fail(MR_ERR_STATE, "photon map is not balanced and resident on a device");
        if (m->device != s->device) return fail(MR_ERR_INVALID, "photon map lives on device %d, the scene on %d", m->device, s->device);
    }
    if (n == 0) return MR_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    MR_HIP_CHECK(hipSetDevice(s->device));
    float *d_pos = d_scratch, *d_nrm = d_scratch + 3 * n;
    float *d_irr[2] = {d_scratch + 6 * n, d_scratch + 9 * n};
    mr_status st = launch_gather_queries(s->dev, d_rays, d_hits, n, d_pos, d_nrm, stream);
    if (st != MR_OK) return st;
    for (int i = 0; i < 2; i++) {
        if (!maps[i]) { d_irr[i] = nullptr; continue; }
        st = launch_irradiance(maps[i]->dev, maps[i]->dev.work_counters + (maps[i]->next_counter.fetch_add(1) % kPhotonWorkCounters), d_pos, d_nrm, n, max_dist, nphotons, d_irr[i], nullptr, nullptr, maps[i]->d_stats, stream);
        if (st != MR_OK) return st;
    }
    return launch_gather_accumulate(d_irr[0], d_irr[1], n, spp, d_rgb, stream);
}

}  // extern "C"
