// mr_build.cpp -- host-side BVH construction that yields the SAME tree as the reference's
// BVH::build (BVH.cpp:60-339), so that traversal order -- and with it tie-breaking between
// equal-t hits and the -DSTATS counters -- is identical to the reference's.
//
// The algorithm (per node): pad the box by +-epsilon; leaf if <= leaf_size objects or depth 32;
// otherwise for each axis run a 32-step bisection of the split plane, always moving the plane into
// the costlier side (cost = N * surface area), tracking child bounds incrementally, and keep the
// cheapest (axis, plane, child boxes) seen; finally partition by centroid < plane.
//
// Own structure, not a translation: objects are rows of a flat PrimRef table (bounds + centroid
// pre-computed, Triangle.cpp:41-48,97-118), the two sides of a trial split are index ranges in two
// scratch arrays owned by the builder (no per-node vectors), and nodes go into one DFS-ordered array.
#include <cmath>
#include <cstring>
#include <limits>

#include "mr_internal.h"

namespace mr {
namespace {

constexpr float kEps = 1e-4f;   // Miro.h:9
constexpr int kMaxDepth = 32;   // BVH.h:57
constexpr int kBisectSteps = 32;
constexpr float kInf = std::numeric_limits<float>::infinity();

struct PrimRef { float lo[3], hi[3], ctr[3]; };

struct Box {
    float lo[3], hi[3];
    void clear() { for (int k = 0; k < 3; k++) { lo[k] = kInf; hi[k] = -kInf; } }
    void grow(const PrimRef &p) {
        for (int k = 0; k < 3; k++) {
            if (hi[k] < p.hi[k]) hi[k] = p.hi[k];
            if (lo[k] > p.lo[k]) lo[k] = p.lo[k];
        }
    }
    // 2 * sum of pairwise extents, accumulated in the order (y,z), (z,x), (x,y) (BVH.cpp:41-51)
    float area2() const {
        float a = 0;
        for (int d = 0; d < 3; d++) {
            int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
            a += (hi[d1] - lo[d1]) * (hi[d2] - lo[d2]);
        }
        return 2 * a;
    }
};

class RefBuilder {
public:
    RefBuilder(const HostMesh &mesh, uint32_t leaf_size, HostTree &out)
        : leaf_size_((int)leaf_size), tree_(out) {
        const uint32_t nt = mesh.n_triangles();
        prims_.resize(nt);
        for (uint32_t i = 0; i < nt; i++) {
            PrimRef &p = prims_[i];
            if (mesh.is_sphere(i)) {      // Sphere.h:19-21: m_center -/+ Vector3(m_radius), centre = m_center
                const float *sp = &mesh.spheres[4 * (size_t)mesh.vi[3 * (size_t)i + 1]];
                for (int k = 0; k < 3; k++) { p.lo[k] = sp[k] - sp[3]; p.hi[k] = sp[k] + sp[3]; p.ctr[k] = sp[k]; }
                continue;
            }
            const float *a = &mesh.v[3 * (size_t)mesh.vi[3 * i + 0]];
            const float *b = &mesh.v[3 * (size_t)mesh.vi[3 * i + 1]];
            const float *c = &mesh.v[3 * (size_t)mesh.vi[3 * i + 2]];
            const float third = 1.0f / 3.0f;      // Vector3::operator/(3): multiply by rounded 1/3
            for (int k = 0; k < 3; k++) {
                float mn = a[k], mx = a[k];
                if (b[k] < mn) mn = b[k];
                if (b[k] > mx) mx = b[k];
                if (c[k] < mn) mn = c[k];
                if (c[k] > mx) mx = c[k];
                p.lo[k] = mn; p.hi[k] = mx;
                float ba = b[k] - a[k], ca = c[k] - a[k];
                p.ctr[k] = (a[k] + ba * third) + ca * third;     // Triangle.cpp:45-47
            }
        }
        side_[0].resize(nt + 1);
        side_[1].resize(nt + 1);
        tree_.nodes.clear();
        tree_.leaf_prims.clear();
        tree_.leaf_prims.reserve(nt);
        tree_.n_leaves = 0;
        tree_.max_depth = 0;
        tree_.leaf_size = leaf_size;
    }

    void run() {
        std::vector<uint32_t> all(prims_.size());
        for (uint32_t i = 0; i < all.size(); i++) all[i] = i;
        Box none;
        none.clear();
        node(all, 0, none);
    }

private:
    Box bounds_of(const uint32_t *idx, int n) const {
        Box b;
        b.clear();
        for (int i = 0; i < n; i++) b.grow(prims_[idx[i]]);
        return b;
    }
    static float cost(const Box &b, int n) { return n == 0 ? 0.0f : (float)n * b.area2(); }

    struct Split { float cost = kInf, plane = 0.0f; int axis = 0; Box child[2]; };

    // one axis of the search (BVH.cpp:178-302); updates `best` in place
    void bisect_axis(const std::vector<uint32_t> &objs, int axis, const Box &padded, Split &best) {
        uint32_t *S[2] = {side_[0].data(), side_[1].data()};
        int cnt[2] = {0, 0}, frozen[2] = {0, 0};
        float plane = (padded.hi[axis] + padded.lo[axis]) / 2.0f, lo = padded.lo[axis], hi = padded.hi[axis];
        for (uint32_t o : objs) {
            int s = prims_[o].ctr[axis] < plane ? 0 : 1;
            S[s][cnt[s]++] = o;
        }
        Box box[2] = {bounds_of(S[0], cnt[0]), bounds_of(S[1], cnt[1])};

        auto consider = [&]() {
            float c0 = cost(box[0], cnt[0]), c1 = cost(box[1], cnt[1]);
            if (c0 + c1 < best.cost) {
                best.cost = c0 + c1; best.axis = axis; best.plane = plane;
                best.child[0] = box[0]; best.child[1] = box[1];
            }
            return c0 > c1 ? 0 : 1;       // the costlier side (ties -> right)
        };

        for (int step = 0; step < kBisectSteps; step++) {
            const int heavy = consider(), light = heavy ^ 1;
            if (heavy == 0) hi = plane; else lo = plane;
            plane = (lo + hi) / 2;
            frozen[light] = cnt[light];   // what is on the light side now can never move back
            bool boundary_left = false;
            // scan the movable tail of the heavy side from the back; swap-remove keeps the scan valid
            for (int i = cnt[heavy] - 1; i >= frozen[heavy]; i--) {
                const uint32_t o = S[heavy][i];
                const PrimRef &p = prims_[o];
                const bool moves = heavy == 0 ? (p.ctr[axis] > plane) : (p.ctr[axis] < plane);
                if (!moves) continue;
                for (int k = 0; k < 3; k++) {
                    if (p.hi[k] > box[light].hi[k]) box[light].hi[k] = p.hi[k];
                    if (p.lo[k] < box[light].lo[k]) box[light].lo[k] = p.lo[k];
                    if (!boundary_left &&
                        (p.hi[k] >= box[heavy].hi[k] - kEps || p.lo[k] <= box[heavy].lo[k] + kEps))
                        boundary_left = true;
                }
                S[light][cnt[light]++] = o;
                S[heavy][i] = S[heavy][cnt[heavy] - 1];
                S[heavy][cnt[heavy] - 1] = o;
                cnt[heavy]--;
            }
            if (boundary_left) box[heavy] = bounds_of(S[heavy], cnt[heavy]);
        }
        consider();
    }

    // returns node index; `given` = box handed down by the parent (lo[0]==inf: compute it)
    int node(const std::vector<uint32_t> &objs, int depth, Box given) {
        const int me = (int)tree_.nodes.size();
        tree_.nodes.emplace_back();
        if (given.lo[0] == kInf) given = bounds_of(objs.data(), (int)objs.size());
        for (int k = 0; k < 3; k++) { given.lo[k] -= kEps; given.hi[k] += kEps; }
        {
            HostNode &nd = tree_.nodes[me];
            memcpy(nd.lo, given.lo, sizeof(nd.lo));
            memcpy(nd.hi, given.hi, sizeof(nd.hi));
            nd.depth = depth;
        }
        if ((uint32_t)depth > tree_.max_depth) tree_.max_depth = (uint32_t)depth;

        if ((int)objs.size() <= leaf_size_ || depth >= kMaxDepth) {
            HostNode &nd = tree_.nodes[me];
            nd.is_leaf = 1;
            nd.a = (int32_t)tree_.leaf_prims.size();
            nd.b = (int32_t)objs.size();
            tree_.leaf_prims.insert(tree_.leaf_prims.end(), objs.begin(), objs.end());
            tree_.n_leaves++;
            return me;
        }

        Split best;
        for (int s = 0; s < 2; s++)   // Vector3 bestCorners[2][2] default to (0,1,2) (BVH.cpp:173)
            for (int k = 0; k < 3; k++) { best.child[s].lo[k] = (float)k; best.child[s].hi[k] = (float)k; }
        for (int axis = 0; axis < 3; axis++) bisect_axis(objs, axis, given, best);

        std::vector<uint32_t> part[2];
        for (uint32_t o : objs) part[prims_[o].ctr[best.axis] < best.plane ? 0 : 1].push_back(o);
        int kids[2];
        for (int s = 0; s < 2; s++) {
            kids[s] = node(part[s], depth + 1, best.child[s]);
            std::vector<uint32_t>().swap(part[s]);
        }
        HostNode &nd = tree_.nodes[me];
        nd.is_leaf = 0;
        nd.a = kids[0];
        nd.b = kids[1];
        return me;
    }

    int leaf_size_;
    HostTree &tree_;
    std::vector<PrimRef> prims_;
    std::vector<uint32_t> side_[2];
};

}  // namespace

mr_status build_reference_tree(const HostMesh &mesh, uint32_t leaf_size, HostTree &tree) {
    if (leaf_size == 0) leaf_size = 4;
    RefBuilder b(mesh, leaf_size, tree);
    b.run();
    return MR_OK;
}

}  // namespace mr
