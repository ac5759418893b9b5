// mr_obj.cpp -- OBJ ingestion with the semantics of the reference loader, so that vertex and
// normal values (and therefore hit t / P / N) agree with TriangleMesh::load:
//   * records are read in chunks of at most 79 characters (fgets(line, 80, fp), TriangleMeshLoad.cpp:119,180)
//   * only "v", "vn", "vt", "f" records; faces are triangles; indices are 1-based v, v/t, v/t/n, v//n (:81-111)
//   * v -> ctm * v with w = 1 (Matrix4x4.h:581-587); vn -> normalise((ctm^-1)^T * n) (:176-178,:184-197)
//   * a face whose LAST corner has no normal index gets three copies of its face normal (:252-281);
//     those synthesised normals are later replaced by the average of all normals incident on the
//     vertex, the accumulator starting at (0,1,2) as Vector3() does (:287-308, Vector3.h:27)
// Not a translation: the file is slurped once and scanned in place; per-vertex incidence lists are
// a CSR built from counted references rather than vector-of-vectors.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mr_internal.h"

namespace mr {
namespace {

struct F3 { float x, y, z; };

inline F3 sub(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 add(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline F3 mul(F3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline F3 cross(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// Vector3::operator/=(float) multiplies by a rounded reciprocal (Vector3.h:138-143)
inline F3 div_by(F3 a, float s) { float inv = 1.0f / s; return mul(a, inv); }
inline F3 unit(F3 a) { return div_by(a, sqrtf(dot(a, a))); }

struct Affine {          // rows of a 4x4, row-major
    float m[16];
    F3 apply(F3 u) const {
        return {m[0] * u.x + m[1] * u.y + m[2] * u.z + m[3],
                m[4] * u.x + m[5] * u.y + m[6] * u.z + m[7],
                m[8] * u.x + m[9] * u.y + m[10] * u.z + m[11]};
    }
};

// inverse-transpose by cofactors, rounding as Matrix4x4::invert does (fp32 minors, 1.0/det in double)
Affine inverse_transpose(const Affine &A) {
    const float *a = A.m;
    auto at = [&](int r, int c) { return a[r * 4 + c]; };
    // 2x2 minors of row pairs (2,3), (1,3), (1,2) -- zero-based rows
    float p[3][6];
    const int rowpair[3][2] = {{2, 3}, {1, 3}, {1, 2}};
    const int colpair[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
    for (int k = 0; k < 3; k++)
        for (int j = 0; j < 6; j++)
            p[k][j] = at(rowpair[k][0], colpair[j][0]) * at(rowpair[k][1], colpair[j][1]) -
                      at(rowpair[k][0], colpair[j][1]) * at(rowpair[k][1], colpair[j][0]);
    // 3x3 minors sd[i][j]: delete row i, column j.  Expansion along the remaining top-most row,
    // using the 2x2 minors of the two remaining lower rows (index k selects which row pair).
    float sd[4][4];
    for (int i = 0; i < 4; i++) {
        int top = (i == 0) ? 1 : 0;            // first remaining row
        int k = (i <= 1) ? 0 : (i == 2 ? 1 : 2);   // lower row pair: (2,3) / (1,3) / (1,2)
        for (int j = 0; j < 4; j++) {
            int c[3], n = 0;
            for (int q = 0; q < 4; q++) if (q != j) c[n++] = q;
            auto minor2 = [&](int ca, int cb) {
                for (int w = 0; w < 6; w++)
                    if (colpair[w][0] == ca && colpair[w][1] == cb) return p[k][w];
                return 0.0f;
            };
            sd[i][j] = at(top, c[0]) * minor2(c[1], c[2]) - at(top, c[1]) * minor2(c[0], c[2]) +
                       at(top, c[2]) * minor2(c[0], c[1]);
        }
    }
    float det = at(0, 0) * sd[0][0] - at(0, 1) * sd[0][1] + at(0, 2) * sd[0][2] - at(0, 3) * sd[0][3];
    float detInv = (float)(1.0 / (double)det);
    Affine R;
    // inverse(r,c) = (-1)^(r+c) sd[c][r] * detInv ; transposed -> out(r,c) = (-1)^(r+c) sd[r][c] * detInv
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            float s = sd[r][c] * detInv;
            R.m[r * 4 + c] = ((r + c) & 1) ? -s : s;
        }
    return R;
}

struct Corner { int v, t, n; };

// "12", "12/5", "12/5/7", "12//7" with atoi semantics (missing field -> 0)
Corner parse_corner(const char *w) {
    Corner c{0, 0, 0};
    c.v = atoi(w);
    const char *s1 = strchr(w, '/');
    if (s1) {
        c.t = atoi(s1 + 1);
        const char *s2 = strchr(s1 + 1, '/');
        if (s2) {
            // the reference keeps only the LAST slash-separated field as the normal
            const char *last = strrchr(w, '/');
            c.n = atoi(last + 1);
        }
    }
    return c;
}

}  // namespace

mr_status load_obj(const char *path, const float *ctm16, HostMesh &mesh, uint32_t *n_tris_out) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return fail(MR_ERR_IO, "cannot open \"%s\" for reading", path);
    std::vector<char> buf;
    {
        fseek(fp, 0, SEEK_END);
        long sz = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        buf.resize((size_t)(sz > 0 ? sz : 0) + 1);
        size_t got = sz > 0 ? fread(buf.data(), 1, (size_t)sz, fp) : 0;
        buf.resize(got + 1);
        buf[got] = '\0';
        fclose(fp);
    }
    Affine ctm;
    if (ctm16) memcpy(ctm.m, ctm16, sizeof(ctm.m));
    else { const float id[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1}; memcpy(ctm.m, id, sizeof(id)); }
    const Affine nctm = inverse_transpose(ctm);

    std::vector<F3> verts, normals;
    std::vector<char> synthesised;            // per normal slot: must be averaged
    std::vector<uint32_t> vidx, nidx;
    std::vector<std::pair<uint32_t, uint32_t>> incidence;   // (vertex, normal slot) in encounter order

    // walk the file in fgets(…, 80) records: up to 79 bytes, ending after '\n'
    const size_t total = buf.size() - 1;
    size_t pos = 0;
    char rec[80];
    while (pos < total) {
        size_t len = 0;
        while (pos < total && len < 79) {
            char ch = buf[pos++];
            rec[len++] = ch;
            if (ch == '\n') break;
        }
        rec[len] = '\0';
        if (rec[0] == 'v') {
            float x = 0, y = 0, z = 0;
            if (rec[1] == 'n') {
                sscanf(rec + 2, "%f %f %f", &x, &y, &z);
                normals.push_back(unit(nctm.apply({x, y, z})));
                synthesised.push_back(0);
            } else if (rec[1] == 't') {
                // texture coordinates do not take part in intersection
            } else {
                sscanf(rec + 1, "%f %f %f", &x, &y, &z);
                verts.push_back(ctm.apply({x, y, z}));
            }
        } else if (rec[0] == 'f') {
            char w[3][80] = {"", "", ""};
            sscanf(rec + 1, "%79s %79s %79s", w[0], w[1], w[2]);
            Corner c[3];
            uint32_t tri_v[3], tri_n[3] = {0, 0, 0};
            for (int k = 0; k < 3; k++) {
                c[k] = parse_corner(w[k]);
                tri_v[k] = (uint32_t)(c[k].v - 1);
                if (c[k].n) {
                    tri_n[k] = (uint32_t)(c[k].n - 1);
                    incidence.emplace_back(tri_v[k], tri_n[k]);
                }
            }
            for (int k = 0; k < 3; k++)
                if (tri_v[k] >= verts.size())
                    return fail(MR_ERR_IO, "\"%s\": face references vertex %u before it is defined", path, tri_v[k] + 1);
            if (!c[2].n) {
                F3 fn = unit(cross(sub(verts[tri_v[1]], verts[tri_v[0]]), sub(verts[tri_v[2]], verts[tri_v[0]])));
                for (int k = 0; k < 3; k++) {
                    tri_n[k] = (uint32_t)normals.size();
                    normals.push_back(fn);
                    synthesised.push_back(1);
                }
                for (int k = 0; k < 3; k++) incidence.emplace_back(tri_v[k], tri_n[k]);
            }
            for (int k = 0; k < 3; k++) { vidx.push_back(tri_v[k]); nidx.push_back(tri_n[k]); }
        }
    }
    for (uint32_t ni : nidx)
        if (ni >= normals.size()) return fail(MR_ERR_IO, "\"%s\": normal index %u out of range", path, ni + 1);
    // An explicit normal index on corner 0 or 1 of a face whose last corner has none never reaches nidx (the face's
    // slots are overwritten by the synthesised ones) but still takes part in the smoothing pass below, as in the
    // reference (TriangleMeshLoad.cpp:226-248 record it before :252 decides): it has to be in range as well.
    for (auto &e : incidence)
        if (e.second >= normals.size())
            return fail(MR_ERR_IO, "\"%s\": normal index %u out of range", path, e.second + 1);

    // smooth the synthesised normals: CSR of (vertex -> incident normal slots) in encounter order
    {
        std::vector<uint32_t> start(verts.size() + 1, 0);
        for (auto &e : incidence) start[e.first + 1]++;
        for (size_t i = 0; i < verts.size(); i++) start[i + 1] += start[i];
        std::vector<uint32_t> fill(start.begin(), start.end() - 1), slots(incidence.size());
        for (auto &e : incidence) slots[fill[e.first]++] = e.second;
        for (size_t vtx = 0; vtx < verts.size(); vtx++) {
            uint32_t b = start[vtx], e = start[vtx + 1];
            if (b == e) continue;
            F3 acc{0.0f, 1.0f, 2.0f};
            for (uint32_t k = b; k < e; k++) acc = add(acc, normals[slots[k]]);
            acc = unit(div_by(acc, (float)(e - b)));
            for (uint32_t k = b; k < e; k++)
                if (synthesised[slots[k]]) normals[slots[k]] = acc;
        }
    }

    const uint32_t vbase = mesh.n_vertices(), nbase = mesh.n_normals();
    for (auto &p : verts) { mesh.v.push_back(p.x); mesh.v.push_back(p.y); mesh.v.push_back(p.z); }
    for (auto &p : normals) { mesh.n.push_back(p.x); mesh.n.push_back(p.y); mesh.n.push_back(p.z); }
    for (uint32_t i : vidx) mesh.vi.push_back(i + vbase);
    for (uint32_t i : nidx) mesh.ni.push_back(i + nbase);
    if (n_tris_out) *n_tris_out = (uint32_t)(vidx.size() / 3);
    return MR_OK;
}

}  // namespace mr
