/*
 * miro_oracle.c -- scalar CPU restatement of the Miro intersection hot path.
 * TEST INFRASTRUCTURE ONLY (see miro_oracle.h).  Plain C99; must be compiled with
 * -ffp-contract=off and without -ffast-math / -march=native so that every fp32
 * operation is an individually rounded IEEE op, as in the reference's scalar build
 * (g++ -mno-sse4.1).
 *
 * Each function cites the reference file:line it follows.  Expression trees (operand
 * order, a/b computed as a*(1/b) where Vector3::operator/ does so) are kept identical
 * because the GPU "exact" kernels are compared bit-for-bit against this file.
 */
#include "miro_oracle_internal.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ Vector3.h */
/* Vector3::operator/(float): multiplies by a rounded reciprocal (Vector3.h:125-129) */
static inline v3 v3divs(v3 a, float s) { float inv = 1.0f / s; return v3scale(a, inv); }
/* Vector3::normalize (Vector3.h:205-208): *this /= length() */
static inline v3 v3normalize(v3 a) { return v3divs(a, sqrtf(v3dot(a, a))); }

/* ------------------------------------------------------------------ scene storage */
orc_scene *orc_scene_new(void)
{
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    return s;
}

void orc_scene_free(orc_scene *s)
{
    if (!s) return;
    free(s->v); free(s->n); free(s->vi); free(s->ni);
    free(s->omin); free(s->omax); free(s->ocen);
    free(s->nodes); free(s->leaf_prims);
    free(s->spheres); free(s->planes); free(s->plane_mat);
    orc_sse_free(s);
    free(s);
}

static void scene_grow(orc_scene *s, int add_v, int add_n, int add_t)
{
    s->v  = (float *)realloc(s->v,  sizeof(float) * 3 * (size_t)(s->nv + add_v + 1));
    s->n  = (float *)realloc(s->n,  sizeof(float) * 3 * (size_t)(s->nn + add_n + 1));
    s->vi = (uint32_t *)realloc(s->vi, sizeof(uint32_t) * 3 * (size_t)(s->nt + add_t + 1));
    s->ni = (uint32_t *)realloc(s->ni, sizeof(uint32_t) * 3 * (size_t)(s->nt + add_t + 1));
}

int orc_scene_counts(const orc_scene *s, int *nv, int *nn, int *nt)
{
    if (nv) *nv = s->nv;
    if (nn) *nn = s->nn;
    if (nt) *nt = s->nt;
    return s->nt;
}
const float    *orc_scene_vertices(const orc_scene *s) { return s->v; }
const float    *orc_scene_normals(const orc_scene *s)  { return s->n; }
const uint32_t *orc_scene_vidx(const orc_scene *s)     { return s->vi; }
const uint32_t *orc_scene_nidx(const orc_scene *s)     { return s->ni; }

int orc_scene_add_arrays(orc_scene *s, int nv, const float *v, int nn, const float *n,
                         int nt, const uint32_t *vi, const uint32_t *ni)
{
    scene_grow(s, nv, nn, nt);
    memcpy(s->v + 3 * (size_t)s->nv, v, sizeof(float) * 3 * (size_t)nv);
    memcpy(s->n + 3 * (size_t)s->nn, n, sizeof(float) * 3 * (size_t)nn);
    for (int i = 0; i < 3 * nt; i++) {
        s->vi[3 * (size_t)s->nt + i] = vi[i] + (uint32_t)s->nv;
        s->ni[3 * (size_t)s->nt + i] = ni[i] + (uint32_t)s->nn;
    }
    s->nv += nv; s->nn += nn; s->nt += nt;
    return nt;
}

/* TriangleMesh::createSingleTriangle + setV1..3 / setN1..3 (TriangleMeshLoad.cpp:15-56) */
int orc_scene_add_triangle(orc_scene *s, const float v[9], const float n[9])
{
    static const uint32_t idx[3] = {0, 1, 2};
    return orc_scene_add_arrays(s, 3, v, 3, n, 1, idx, idx);
}

/* Sphere as a bounded object (Scene::addObject, Scene.h:20-25) */
int orc_scene_add_sphere(orc_scene *s, const float center[3], float radius)
{
    scene_grow(s, 0, 0, 1);
    s->spheres = (float *)realloc(s->spheres, sizeof(float) * 4 * (size_t)(s->nspheres + 1));
    memcpy(s->spheres + 4 * (size_t)s->nspheres, center, sizeof(float) * 3);
    s->spheres[4 * (size_t)s->nspheres + 3] = radius;
    for (int k = 0; k < 3; k++) {
        s->vi[3 * (size_t)s->nt + k] = k == 0 ? ORC_SPHERE_SLOT : (k == 1 ? (uint32_t)s->nspheres : 0u);
        s->ni[3 * (size_t)s->nt + k] = s->vi[3 * (size_t)s->nt + k];
    }
    s->nspheres++;
    return s->nt++;
}

int orc_scene_add_plane(orc_scene *s, const float normal[3], const float origin[3], uint32_t material)
{
    s->planes = (float *)realloc(s->planes, sizeof(float) * 6 * (size_t)(s->nplanes + 1));
    s->plane_mat = (uint32_t *)realloc(s->plane_mat, sizeof(uint32_t) * (size_t)(s->nplanes + 1));
    memcpy(s->planes + 6 * (size_t)s->nplanes, normal, sizeof(float) * 3);
    memcpy(s->planes + 6 * (size_t)s->nplanes + 3, origin, sizeof(float) * 3);
    s->plane_mat[s->nplanes] = material;
    return s->nplanes++;
}

/* ------------------------------------------------------------------ Matrix4x4.h */
/* Matrix4x4::invert (Matrix4x4.h:308-366): cofactor expansion in fp32, detInv = 1.0/det
 * evaluated in double then rounded; followed by transpose (:284-306). m is row-major. */
static void mat_invert_transpose(const float a[16], float out[16])
{
#define M(r, c) a[(r - 1) * 4 + (c - 1)]
    float T34C12 = M(3,1)*M(4,2)-M(3,2)*M(4,1), T34C13 = M(3,1)*M(4,3)-M(3,3)*M(4,1);
    float T34C14 = M(3,1)*M(4,4)-M(3,4)*M(4,1), T34C23 = M(3,2)*M(4,3)-M(3,3)*M(4,2);
    float T34C24 = M(3,2)*M(4,4)-M(3,4)*M(4,2), T34C34 = M(3,3)*M(4,4)-M(3,4)*M(4,3);
    float T24C12 = M(2,1)*M(4,2)-M(2,2)*M(4,1), T24C13 = M(2,1)*M(4,3)-M(2,3)*M(4,1);
    float T24C14 = M(2,1)*M(4,4)-M(2,4)*M(4,1), T24C23 = M(2,2)*M(4,3)-M(2,3)*M(4,2);
    float T24C24 = M(2,2)*M(4,4)-M(2,4)*M(4,2), T24C34 = M(2,3)*M(4,4)-M(2,4)*M(4,3);
    float T23C12 = M(2,1)*M(3,2)-M(2,2)*M(3,1), T23C13 = M(2,1)*M(3,3)-M(2,3)*M(3,1);
    float T23C14 = M(2,1)*M(3,4)-M(2,4)*M(3,1), T23C23 = M(2,2)*M(3,3)-M(2,3)*M(3,2);
    float T23C24 = M(2,2)*M(3,4)-M(2,4)*M(3,2), T23C34 = M(2,3)*M(3,4)-M(2,4)*M(3,3);

    float sd11 = M(2,2)*T34C34 - M(2,3)*T34C24 + M(2,4)*T34C23;
    float sd12 = M(2,1)*T34C34 - M(2,3)*T34C14 + M(2,4)*T34C13;
    float sd13 = M(2,1)*T34C24 - M(2,2)*T34C14 + M(2,4)*T34C12;
    float sd14 = M(2,1)*T34C23 - M(2,2)*T34C13 + M(2,3)*T34C12;
    float sd21 = M(1,2)*T34C34 - M(1,3)*T34C24 + M(1,4)*T34C23;
    float sd22 = M(1,1)*T34C34 - M(1,3)*T34C14 + M(1,4)*T34C13;
    float sd23 = M(1,1)*T34C24 - M(1,2)*T34C14 + M(1,4)*T34C12;
    float sd24 = M(1,1)*T34C23 - M(1,2)*T34C13 + M(1,3)*T34C12;
    float sd31 = M(1,2)*T24C34 - M(1,3)*T24C24 + M(1,4)*T24C23;
    float sd32 = M(1,1)*T24C34 - M(1,3)*T24C14 + M(1,4)*T24C13;
    float sd33 = M(1,1)*T24C24 - M(1,2)*T24C14 + M(1,4)*T24C12;
    float sd34 = M(1,1)*T24C23 - M(1,2)*T24C13 + M(1,3)*T24C12;
    float sd41 = M(1,2)*T23C34 - M(1,3)*T23C24 + M(1,4)*T23C23;
    float sd42 = M(1,1)*T23C34 - M(1,3)*T23C14 + M(1,4)*T23C13;
    float sd43 = M(1,1)*T23C24 - M(1,2)*T23C14 + M(1,4)*T23C12;
    float sd44 = M(1,1)*T23C23 - M(1,2)*T23C13 + M(1,3)*T23C12;
    float det = M(1,1)*sd11 - M(1,2)*sd12 + M(1,3)*sd13 - M(1,4)*sd14;
    float detInv = (float)(1.0 / (double)det);
#undef M
    float inv[16] = {
         sd11*detInv, -sd21*detInv,  sd31*detInv, -sd41*detInv,
        -sd12*detInv,  sd22*detInv, -sd32*detInv,  sd42*detInv,
         sd13*detInv, -sd23*detInv,  sd33*detInv, -sd43*detInv,
        -sd14*detInv,  sd24*detInv, -sd34*detInv,  sd44*detInv };
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) out[r * 4 + c] = inv[c * 4 + r];
}

/* operator*(Matrix4x4, Vector3) (Matrix4x4.h:581-587): w == 1, fourth row ignored */
static inline v3 mat_xform(const float m[16], v3 u)
{
    v3 r;
    r.x = m[0]*u.x + m[1]*u.y + m[2]*u.z  + m[3];
    r.y = m[4]*u.x + m[5]*u.y + m[6]*u.z  + m[7];
    r.z = m[8]*u.x + m[9]*u.y + m[10]*u.z + m[11];
    return r;
}

/* ------------------------------------------------------------------ OBJ loader */
/* getIndices (TriangleMeshLoad.cpp:81-111): "v", "v/t", "v/t/n", "v//n"; atoi semantics */
static void get_indices(char *word, int *vindex, int *tindex, int *nindex)
{
    char *nullstr = (char *)" ";
    char *tp = nullstr, *np = nullstr;
    for (char *p = word; *p != '\0'; p++) {
        if (*p == '/') {
            if (tp == nullstr) tp = p + 1; else np = p + 1;
            *p = '\0';
        }
    }
    *vindex = atoi(word);
    *tindex = atoi(tp);
    *nindex = atoi(np);
}

typedef struct { int *d; int n, cap; } ivec;
static void ivec_push(ivec *v, int x)
{
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 4; v->d = (int *)realloc(v->d, sizeof(int) * (size_t)v->cap); }
    v->d[v->n++] = x;
}

/* TriangleMesh::load / loadObj (TriangleMeshLoad.cpp:63-311) */
int orc_scene_add_obj(orc_scene *s, const char *path, const float *ctm_in)
{
    static const float ident[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1};
    const float *ctm = ctm_in ? ctm_in : ident;
    FILE *fp = fopen(path, "rb");
    if (!fp) return -1;

    int nv = 0, nn = 0, nf = 0;
    char line[81];
    while (fgets(line, 80, fp) != 0) {                       /* :119 -- 79-char reads */
        if (line[0] == 'v') { if (line[1] == 'n') nn++; else if (line[1] == 't') { } else nv++; }
        else if (line[0] == 'f') nf++;
    }
    fseek(fp, 0, SEEK_SET);

    int ncap = nv > nf * 3 ? nv : nf * 3;                    /* :138 */
    /* the reference's array overflows when vn records and normal-less faces mix (nn + 3*faces can pass :138's
     * size); give the restatement room for the worst case so such files behave as if the array were large enough */
    if (ncap < nn + 3 * nf) ncap = nn + 3 * nf;
    v3 *normals = (v3 *)calloc((size_t)ncap + 1, sizeof(v3));
    v3 *verts = (v3 *)calloc((size_t)nv + 1, sizeof(v3));
    char *fix = (char *)calloc((size_t)ncap + 1, 1);
    ivec *nbr = (ivec *)calloc((size_t)nv + 1, sizeof(ivec));
    uint32_t *vidx = (uint32_t *)calloc((size_t)nf * 3 + 3, sizeof(uint32_t));
    uint32_t *nidx = (uint32_t *)calloc((size_t)nf * 3 + 3, sizeof(uint32_t));
    int ntris = 0, nvertices = 0, nnormals = 0;
    int bad = 0;   /* an index the reference would chase out of bounds (undefined behaviour there): refuse the file */

    float nctm[16];
    mat_invert_transpose(ctm, nctm);                         /* :176-178 */

    while (fgets(line, 80, fp) != 0) {
        if (line[0] == 'v') {
            if (line[1] == 'n') {                            /* :184-197 */
                float x = 0, y = 0, z = 0;
                sscanf(&line[2], "%f %f %f\n", &x, &y, &z);
                v3 nrm = {x, y, z};
                normals[nnormals] = v3normalize(mat_xform(nctm, nrm));
                fix[nnormals] = 0;
                nnormals++;
            } else if (line[1] == 't') {
                /* texture coordinates are not on the intersection path */
            } else {                                         /* :206-216 */
                float x = 0, y = 0, z = 0;
                sscanf(&line[1], "%f %f %f\n", &x, &y, &z);
                v3 p = {x, y, z};
                verts[nvertices++] = mat_xform(ctm, p);
            }
        } else if (line[0] == 'f') {                         /* :218-284 */
            char s1[80] = "", s2[80] = "", s3[80] = "";
            int v, t, n;
            sscanf(&line[1], "%s %s %s\n", s1, s2, s3);
            char *ss[3] = {s1, s2, s3};
            for (int k = 0; k < 3; k++) {
                get_indices(ss[k], &v, &t, &n);
                if (v < 1 || v > nvertices || n < 0 || n > ncap) { bad = 1; v = 1; n = 0; if (nvertices == 0) break; }
                vidx[3 * ntris + k] = (uint32_t)(v - 1);
                if (n) {
                    nidx[3 * ntris + k] = (uint32_t)(n - 1);
                    if (v >= 1 && v <= nv) ivec_push(&nbr[v - 1], n - 1);
                }
            }
            if (bad) break;
            if (!n) {                                        /* :252-281 (n of the LAST corner) */
                v3 e1 = v3sub(verts[vidx[3 * ntris + 1]], verts[vidx[3 * ntris + 0]]);
                v3 e2 = v3sub(verts[vidx[3 * ntris + 2]], verts[vidx[3 * ntris + 0]]);
                for (int i = 0; i < 3; i++) {
                    normals[nnormals] = v3normalize(v3cross(e1, e2));
                    fix[nnormals] = 1;
                    nnormals++;
                }
                nidx[3 * ntris + 0] = (uint32_t)(nnormals - 3);
                nidx[3 * ntris + 1] = (uint32_t)(nnormals - 2);
                nidx[3 * ntris + 2] = (uint32_t)(nnormals - 1);
                for (int k = 0; k < 3; k++)
                    ivec_push(&nbr[vidx[3 * ntris + k]], (int)nidx[3 * ntris + k]);
            }
            ntris++;
        }
    }
    fclose(fp);
    /* a normal slot nobody ever wrote (neither a vn record nor a synthesised face normal): refused, as the product does */
    for (int i = 0; i < 3 * ntris && !bad; i++)
        if (nidx[i] >= (uint32_t)nnormals) bad = 1;
    if (bad) {
        for (int i = 0; i < nv; i++) free(nbr[i].d);
        free(nbr); free(fix); free(normals); free(verts); free(vidx); free(nidx);
        return -2;
    }

    /* normal averaging (:287-308).  NB `Vector3 avg;` starts at (0,1,2) (Vector3.h:27). */
    for (int i = 0; i < nvertices; i++) {
        if (nbr[i].n == 0) continue;
        v3 avg = {0.0f, 1.0f, 2.0f};
        for (int j = 0; j < nbr[i].n; j++) avg = v3add(avg, normals[nbr[i].d[j]]);
        avg = v3divs(avg, (float)nbr[i].n);
        avg = v3normalize(avg);
        for (int j = 0; j < nbr[i].n; j++)
            if (fix[nbr[i].d[j]]) normals[nbr[i].d[j]] = avg;
    }

    /* addMeshTrianglesToScene (assignment2.cpp:449-461): triangles in file order */
    int nn_used = nnormals;
    orc_scene_add_arrays(s, nvertices, (const float *)verts, nn_used, (const float *)normals,
                         ntris, vidx, nidx);
    for (int i = 0; i < nv; i++) free(nbr[i].d);
    free(nbr); free(fix); free(normals); free(verts); free(vidx); free(nidx);
    return ntris;
}

/* ------------------------------------------------------------------ object pre-calc */
/* Triangle::updateMinMax (Triangle.cpp:97-118), Triangle::center (:41-48) */
static void precalc_objects(orc_scene *s)
{
    free(s->omin); free(s->omax); free(s->ocen);
    s->omin = (v3 *)malloc(sizeof(v3) * (size_t)(s->nt + 1));
    s->omax = (v3 *)malloc(sizeof(v3) * (size_t)(s->nt + 1));
    s->ocen = (v3 *)malloc(sizeof(v3) * (size_t)(s->nt + 1));
    const v3 *V = (const v3 *)s->v;
    for (int i = 0; i < s->nt; i++) {
        if (orc_is_sphere(s, (uint32_t)i)) {     /* Sphere.h:19-21: m_center -/+ Vector3(m_radius), m_center */
            const float *sp = s->spheres + 4 * (size_t)s->vi[3*i+1];
            v3 c = {sp[0], sp[1], sp[2]}, r = {sp[3], sp[3], sp[3]};
            s->omin[i] = v3sub(c, r); s->omax[i] = v3add(c, r); s->ocen[i] = c;
            continue;
        }
        v3 a = V[s->vi[3*i]], b = V[s->vi[3*i+1]], c = V[s->vi[3*i+2]];
        v3 mn = a, mx = a;
        v3 vs[2] = {b, c};
        for (int k = 0; k < 2; k++) {
            if (vs[k].x < mn.x) mn.x = vs[k].x;
            if (vs[k].y < mn.y) mn.y = vs[k].y;
            if (vs[k].z < mn.z) mn.z = vs[k].z;
            if (vs[k].x > mx.x) mx.x = vs[k].x;
            if (vs[k].y > mx.y) mx.y = vs[k].y;
            if (vs[k].z > mx.z) mx.z = vs[k].z;
        }
        s->omin[i] = mn; s->omax[i] = mx;
        v3 BmA = v3sub(b, a), CmA = v3sub(c, a);
        s->ocen[i] = v3add(v3add(a, v3divs(BmA, 3.0f)), v3divs(CmA, 3.0f));
    }
}

/* ------------------------------------------------------------------ BVH build */
static inline float comp(const v3 *p, int d) { return ((const float *)p)[d]; }

/* getCornerPoints (BVH.cpp:14-38) */
static void corner_points(const orc_scene *s, const int *objs, int n, float c[2][3])
{
    for (int i = 0; i < 3; i++) { c[0][i] = INFINITY; c[1][i] = -INFINITY; }
    for (int i = 0; i < n; i++) {
        const v3 *mx = &s->omax[objs[i]], *mn = &s->omin[objs[i]];
        for (int j = 0; j < 3; j++) {
            if (c[1][j] < comp(mx, j)) c[1][j] = comp(mx, j);
            if (c[0][j] > comp(mn, j)) c[0][j] = comp(mn, j);
        }
    }
}

/* getArea (BVH.cpp:41-51), getCost (:53-58) */
static float box_area(float c[2][3])
{
    float area = 0;
    for (int d = 0; d < 3; d++) {
        int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
        area += (c[1][d1] - c[0][d1]) * (c[1][d2] - c[0][d2]);
    }
    return 2 * area;
}
static float box_cost(float c[2][3], int n)
{
    if (n == 0) return 0;
    return ((float)n) * box_area(c);
}

static int new_node(orc_scene *s)
{
    if (s->n_nodes == s->cap_nodes) {
        s->cap_nodes = s->cap_nodes ? s->cap_nodes * 2 : 1024;
        s->nodes = (orc_node *)realloc(s->nodes, sizeof(orc_node) * (size_t)s->cap_nodes);
    }
    memset(&s->nodes[s->n_nodes], 0, sizeof(orc_node));
    return s->n_nodes++;
}

/* BVH::build (BVH.cpp:60-339).  c = corners handed down by the parent (bestCorners) or
 * c[0][0]==inf for the root.  Returns the node index; nodes are numbered in DFS pre-order. */
static int build_rec(orc_scene *s, const int *objs, int n, int depth, float c[2][3])
{
    const float eps = 1e-4f;                                         /* Miro.h:9 */
    int me = new_node(s);
    if (c[0][0] == INFINITY) corner_points(s, objs, n, c);            /* :69-72 */
    for (int i = 0; i < 3; i++) { c[0][i] -= eps; c[1][i] += eps; }   /* :75-79 */
    memcpy(s->nodes[me].c, c, sizeof(float) * 6);
    s->nodes[me].depth = depth;
    if (depth > s->max_depth) s->max_depth = depth;

    if (n <= s->leaf_size || depth >= 32) {                           /* :82 */
        s->nodes[me].leaf = 1;
        s->nodes[me].a = s->n_leaf_prims;
        s->nodes[me].b = n;
        for (int i = 0; i < n; i++) s->leaf_prims[s->n_leaf_prims++] = (uint32_t)objs[i];
        s->n_leaves++;
        return me;
    }

    float bestCost = INFINITY, bestPosition = 0.0f;
    int bestDim = 0;
    float bestCorners[2][2][3];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++)           /* Vector3() = (0,1,2) */
        { bestCorners[a][b][0] = 0; bestCorners[a][b][1] = 1; bestCorners[a][b][2] = 2; }

    int *ch[2];
    ch[0] = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    ch[1] = (int *)malloc(sizeof(int) * (size_t)(n + 1));

    for (int dim = 0; dim < 3; dim++) {                               /* :178 */
        float current = (c[1][dim] + c[0][dim]) / 2.0f, beg = c[0][dim], end = c[1][dim];
        int nocheck[2] = {0, 0};
        int cn[2] = {0, 0};
        float corners[2][2][3];
        for (int i = 0; i < n; i++) {                                 /* :187-193 */
            if (comp(&s->ocen[objs[i]], dim) < current) ch[0][cn[0]++] = objs[i];
            else ch[1][cn[1]++] = objs[i];
        }
        corner_points(s, ch[0], cn[0], corners[0]);
        corner_points(s, ch[1], cn[1], corners[1]);

        for (int sd = 0; sd < 32; sd++) {                             /* :198 */
            float costLeft = box_cost(corners[0], cn[0]), costRight = box_cost(corners[1], cn[1]);
            if (costLeft + costRight < bestCost) {                    /* :204-217 */
                bestCost = costLeft + costRight; bestDim = dim; bestPosition = current;
                memcpy(bestCorners, corners, sizeof(corners));
            }
            int canShrink = 0;
            int largest = (costLeft > costRight ? 0 : 1);             /* :223 */
            int smallest = (largest + 1) % 2;
            if (largest == 0) end = current; else beg = current;
            current = (beg + end) / 2;
            nocheck[smallest] = cn[smallest];                         /* :236 */

            for (int i = cn[largest] - 1; i >= nocheck[largest]; i--) {   /* :240 */
                int o = ch[largest][i];
                float ctr = comp(&s->ocen[o], dim);
                if ((largest == 0 && ctr > current) || (largest == 1 && ctr < current)) {
                    const v3 *cmax = &s->omax[o], *cmin = &s->omin[o];
                    for (int j = 0; j < 3; j++) {                     /* :254-273 */
                        if (comp(cmax, j) > corners[smallest][1][j]) corners[smallest][1][j] = comp(cmax, j);
                        if (comp(cmin, j) < corners[smallest][0][j]) corners[smallest][0][j] = comp(cmin, j);
                        if (!canShrink) {
                            if (comp(cmax, j) >= corners[largest][1][j] - eps ||
                                comp(cmin, j) <= corners[largest][0][j] + eps)
                                canShrink = 1;
                        }
                    }
                    ch[smallest][cn[smallest]++] = o;                 /* :275-277 */
                    ch[largest][i] = ch[largest][cn[largest] - 1];
                    ch[largest][cn[largest] - 1] = o;
                    cn[largest]--;
                }
            }
            if (canShrink) corner_points(s, ch[largest], cn[largest], corners[largest]);  /* :282-285 */
        }
        float costLeft = box_cost(corners[0], cn[0]), costRight = box_cost(corners[1], cn[1]);
        if (costLeft + costRight < bestCost) {                        /* :288-302 */
            bestCost = costLeft + costRight; bestDim = dim; bestPosition = current;
            memcpy(bestCorners, corners, sizeof(corners));
        }
    }

    int cn[2] = {0, 0};                                               /* :313-319 */
    for (int i = 0; i < n; i++) {
        if (comp(&s->ocen[objs[i]], bestDim) < bestPosition) ch[0][cn[0]++] = objs[i];
        else ch[1][cn[1]++] = objs[i];
    }
    for (int i = 0; i < 2; i++) {                                     /* :321-337 */
        float cc[2][3];
        memcpy(cc, bestCorners[i], sizeof(cc));
        int child = build_rec(s, ch[i], cn[i], depth + 1, cc);
        if (i == 0) s->nodes[me].a = child; else s->nodes[me].b = child;
    }
    free(ch[0]); free(ch[1]);
    return me;
}

int orc_scene_build(orc_scene *s, int leaf_size)
{
    precalc_objects(s);                                               /* Scene.cpp:56-60 */
    free(s->nodes); s->nodes = NULL; s->n_nodes = s->cap_nodes = 0;
    free(s->leaf_prims);
    s->leaf_prims = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(s->nt + 1));
    s->n_leaf_prims = 0; s->n_leaves = 0; s->max_depth = 0;
    s->leaf_size = leaf_size;
    int *objs = (int *)malloc(sizeof(int) * (size_t)(s->nt + 1));
    for (int i = 0; i < s->nt; i++) objs[i] = i;
    float c[2][3] = {{INFINITY, 0, 0}, {0, 0, 0}};
    build_rec(s, objs, s->nt, 0, c);                                  /* Scene.cpp:72 */
    free(objs);
    orc_sse_free(s);
    if (leaf_size == 8 && s->nspheres == 0) orc_sse_prepare(s);   /* the packet restatement covers triangle leaves only */
    return s->n_nodes;
}

void orc_scene_tree_stats(const orc_scene *s, int *nodes, int *leaves, int *max_depth)
{
    if (nodes) *nodes = s->n_nodes;
    if (leaves) *leaves = s->n_leaves;
    if (max_depth) *max_depth = s->max_depth;
}

int orc_scene_export_tree(const orc_scene *s, float *corners6, int32_t *meta3, uint32_t *leaf_prims)
{
    for (int i = 0; i < s->n_nodes; i++) {
        memcpy(corners6 + 6 * (size_t)i, s->nodes[i].c, sizeof(float) * 6);
        meta3[3 * (size_t)i + 0] = s->nodes[i].leaf;
        meta3[3 * (size_t)i + 1] = s->nodes[i].a;
        meta3[3 * (size_t)i + 2] = s->nodes[i].b;
    }
    memcpy(leaf_prims, s->leaf_prims, sizeof(uint32_t) * (size_t)s->n_leaf_prims);
    return s->n_nodes;
}

/* ------------------------------------------------------------------ intersection */
/* Triangle::intersect (Triangle.cpp:136-169), scalar branch */
int orc_tri_test(const orc_scene *s, uint32_t prim, const orc_ray *r, float tMin, float tMax, orc_hit *out)
{
    const float eps = 1e-4f;
    const v3 *V = (const v3 *)s->v;
    v3 A = V[s->vi[3*prim]], B = V[s->vi[3*prim+1]], C = V[s->vi[3*prim+2]];
    v3 o = {r->ox, r->oy, r->oz}, d = {r->dx, r->dy, r->dz};
    v3 md = {-d.x, -d.y, -d.z};
    v3 BmA = v3sub(B, A), CmA = v3sub(C, A);
    v3 normal = v3cross(BmA, CmA);
    float ddotn = v3dot(md, normal);
    v3 omA = v3sub(o, A);
    float t = v3dot(omA, normal) / ddotn;
    float beta = v3dot(md, v3cross(omA, CmA)) / ddotn;
    float gamma = v3dot(md, v3cross(BmA, omA)) / ddotn;
    if (beta < -eps || gamma < -eps || beta + gamma > 1 + eps || t < tMin || t > tMax) return 0;
    out->t = t; out->beta = beta; out->gamma = gamma; out->prim = prim;
    return 1;
}

/* Sphere::intersect (Sphere.cpp:28-69) */
static int sphere_test(const orc_scene *s, uint32_t prim, const orc_ray *r, float tMin, float tMax, orc_hit *out)
{
    const float *sp = s->spheres + 4 * (size_t)s->vi[3 * (size_t)prim + 1];
    v3 o = {r->ox, r->oy, r->oz}, d = {r->dx, r->dy, r->dz}, cen = {sp[0], sp[1], sp[2]};
    const float radius = sp[3];
    v3 toO = v3sub(o, cen);
    const float a = v3dot(d, d);                                      /* ray.d.length2() */
    const float b = v3dot(v3scale(d, 2), toO);                        /* dot(2*ray.d, toO) */
    const float c = v3dot(toO, toO) - radius * radius;
    const float discrim = b * b - 4.0f * a * c;
    if (discrim < 0) return 0;
    const float sqrt_discrim = sqrtf(discrim);
    const float t[2] = {(-b - sqrt_discrim) / (2.0f * a), (-b + sqrt_discrim) / (2.0f * a)};
    float tt;
    if ((t[0] > tMin) && (t[0] < tMax)) tt = t[0];
    else if ((t[1] > tMin) && (t[1] < tMax)) tt = t[1];
    else return 0;
    out->t = tt; out->beta = 0.0f; out->gamma = 0.0f; out->prim = prim;
    return 1;
}

int orc_obj_test(const orc_scene *s, uint32_t prim, const orc_ray *r, float tMin, float tMax, orc_hit *out)
{
    if (orc_is_sphere(s, prim)) return sphere_test(s, prim, r, tMin, tMax, out);
    return orc_tri_test(s, prim, r, tMin, tMax, out);
}

/* Plane::intersect (Plane.cpp:33-48); fabs() of a float compared with the double literal 1e-6 */
static int plane_test(const orc_scene *s, int i, const orc_ray *r, float tMin, float tMax, orc_hit *out)
{
    const float *pl = s->planes + 6 * (size_t)i;
    v3 n = {pl[0], pl[1], pl[2]}, org = {pl[3], pl[4], pl[5]};
    v3 o = {r->ox, r->oy, r->oz}, d = {r->dx, r->dy, r->dz};
    float ndotd = v3dot(n, d);
    if ((double)fabsf(ndotd) < 1e-6) return 0;
    float t = v3dot(n, v3sub(org, o)) / ndotd;
    if (t < tMin || t > tMax) return 0;
    out->t = t; out->beta = 0.0f; out->gamma = 0.0f; out->prim = ORC_PLANE_BIT | (uint32_t)i;
    return 1;
}

/* slab test shared by BVH::intersect (BVH.cpp:447-458) and the scalar child test (:597-608) */
static inline void slab(const float c[2][3], const orc_ray *r, float *minOverlap, float *maxOverlap)
{
    float mn = -INFINITY, mx = INFINITY;
    const float *o = &r->ox, *d = &r->dx;
    for (int i = 0; i < 3; i++) {
        float t[2];
        t[0] = (c[0][i] - o[i]) / d[i];
        t[1] = (c[1][i] - o[i]) / d[i];
        int m = t[0] > t[1];
        if (t[m] > mn) mn = t[m];
        if (t[m ^ 1] < mx) mx = t[m ^ 1];
    }
    *minOverlap = mn; *maxOverlap = mx;
}

/* BVH::intersectChildren, scalar (BVH.cpp:471-511, 587-657) */
static int isect_children(const orc_scene *s, int node, const orc_ray *r, float tMin, float tMax,
                          orc_hit *minHit, orc_counters *ctr)
{
    int hit = 0;
    orc_hit tmp;
    const orc_node *nd = &s->nodes[node];
    minHit->t = tMax;                                                 /* :477 */
    if (nd->leaf) {
        for (int i = 0; i < nd->b; i++) {                             /* :493-509 */
            uint32_t prim = s->leaf_prims[nd->a + i];
            if (!orc_is_sphere(s, prim)) ctr->tri_tests++;             /* :496 counts Triangle objects only */
            if (orc_obj_test(s, prim, r, tMin, minHit->t, &tmp)) {
                if (tmp.t < minHit->t) { hit = 1; *minHit = tmp; }
            }
        }
        return hit;
    }
    float minT = INFINITY, minTother = INFINITY;
    int minIndex = -1, otherIndex = -1;
    for (int i = 0; i < 2; i++) {                                     /* :593-624 */
        int child = i == 0 ? nd->a : nd->b;
        float mn, mx;
        slab(s->nodes[child].c, r, &mn, &mx);
        if (mn > mx || mn > tMax || mx < tMin) continue;
        if (minT > mn) { minTother = minT; otherIndex = minIndex; minT = mn; minIndex = i; }
        else if (minTother > mn) { otherIndex = i; minTother = mn; }
    }
    if (minIndex == -1) return 0;
    ctr->box_tests++;                                                 /* :632 */
    if (isect_children(s, minIndex == 0 ? nd->a : nd->b, r, tMin, minHit->t, &tmp, ctr)) {
        *minHit = tmp; hit = 1;
    }
    if (otherIndex != -1) {
        ctr->box_tests++;                                             /* :643 */
        if (isect_children(s, otherIndex == 0 ? nd->a : nd->b, r, tMin, minHit->t, &tmp, ctr)) {
            *minHit = tmp; hit = 1;
        }
    }
    return hit;
}

/* Scene::trace (Scene.cpp:214-231): BVH::intersect (BVH.cpp:438-469), then the unbounded-object scan (:220-230),
 * each plane tested against the caller's tMin / tMax and kept when there was no hit yet or it is strictly nearer. */
void orc_trace(const orc_scene *s, const orc_ray *rays, uint64_t n, orc_hit *hits, orc_counters *counters)
{
    orc_counters ctr = {0, 0};
    for (uint64_t i = 0; i < n; i++) {
        const orc_ray *r = &rays[i];
        orc_hit h;
        h.t = r->tmax; h.prim = ORC_MISS; h.beta = 0; h.gamma = 0;
        int hit = 0;
        if (s->n_nodes > 0) {
            float mn, mx;
            slab(s->nodes[0].c, r, &mn, &mx);
            ctr.box_tests++;                                          /* :461 */
            if (!(mn > mx || mn > r->tmax || mx < r->tmin))
                hit = isect_children(s, 0, r, r->tmin, r->tmax, &h, &ctr);
        }
        if (!hit) { h.t = r->tmax; h.prim = ORC_MISS; h.beta = 0; h.gamma = 0; }
        for (int k = 0; k < s->nplanes; k++) {
            orc_hit tmp;
            if (plane_test(s, k, r, r->tmin, r->tmax, &tmp) && (!hit || tmp.t < h.t)) { hit = 1; h = tmp; }
        }
        hits[i] = h;
    }
    if (counters) { counters->box_tests += ctr.box_tests; counters->tri_tests += ctr.tri_tests; }
}

void orc_trace_brute(const orc_scene *s, const orc_ray *rays, uint64_t n, orc_hit *hits)
{
    for (uint64_t i = 0; i < n; i++) {
        orc_hit best, tmp;
        best.t = rays[i].tmax; best.prim = ORC_MISS; best.beta = 0; best.gamma = 0;
        int hit = 0;
        for (int p = 0; p < s->nt; p++)
            if (orc_obj_test(s, (uint32_t)p, &rays[i], rays[i].tmin, best.t, &tmp))
                if (tmp.t < best.t) { best = tmp; hit = 1; }
        for (int k = 0; k < s->nplanes; k++)
            if (plane_test(s, k, &rays[i], rays[i].tmin, rays[i].tmax, &tmp) && (!hit || tmp.t < best.t)) { best = tmp; hit = 1; }
        hits[i] = best;
    }
}

/* ------------------------------------------------------------------ ray generators */
/* PCG-RXS-M-XS-32 output hash; shared verbatim (integer ops) with the device ray generator */
uint32_t orc_hash(uint32_t x)
{
    uint32_t state = x * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
static inline float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

/* Camera::setLookAt/setViewDir/setUp (Camera.h:79-110) + Camera::eyeRay (Camera.cpp:104-161) */
void orc_eye_rays(const orc_camera *cam, int W, int H, int y0, int y1, int spp, int jitter,
                  uint32_t seed, orc_ray *rays)
{
    const float PI = 3.1415926535897932384626433832795028841972f;     /* Miro.h:10 */
    const float DegToRad = PI / 180.0f;
    const float HalfDegToRad = DegToRad / 2.0f;                       /* Camera.cpp:15 */
    v3 eye = {cam->eye[0], cam->eye[1], cam->eye[2]};
    v3 look = {cam->lookat[0], cam->lookat[1], cam->lookat[2]};
    v3 up = {cam->up[0], cam->up[1], cam->up[2]};
    up = v3normalize(up);
    v3 viewDir = v3normalize(v3sub(look, eye));
    v3 mv = {-viewDir.x, -viewDir.y, -viewDir.z};
    v3 wDir = v3normalize(mv);                                        /* :115 */
    v3 uDir = v3normalize(v3cross(up, wDir));
    v3 vDir = v3cross(wDir, uDir);
    float aspect = (float)W / (float)H;
    float top = tanf(cam->fov_deg * HalfDegToRad);                    /* :121 */
    float right = aspect * top, bottom = -top, left = -right;
    uint32_t base = orc_hash(seed);
    uint64_t k = 0;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < W; x++)
            for (int sm = 0; sm < spp; sm++, k++) {
                float dx = 0.5f, dy = 0.5f;                           /* :127 */
                if (jitter) {
                    uint32_t pix = (uint32_t)(y * W + x);
                    uint32_t b = orc_hash(orc_hash(base ^ pix) + (uint32_t)sm);
                    dx = u01(orc_hash(b));
                    dy = u01(orc_hash(b ^ 0x68bc21ebu));
                }
                float u = left + (right - left) * (((float)x + dx) / (float)W);     /* :157 */
                float v = bottom + (top - bottom) * (((float)y + dy) / (float)H);   /* :158 */
                v3 dir = v3sub(v3add(v3scale(uDir, u), v3scale(vDir, v)), wDir);
                dir = v3normalize(dir);                               /* :160 */
                orc_ray *r = &rays[k];
                r->ox = eye.x; r->oy = eye.y; r->oz = eye.z; r->tmin = 0.0f;
                r->dx = dir.x; r->dy = dir.y; r->dz = dir.z; r->tmax = 1e12f;       /* MIRO_TMAX */
            }
}

static inline void hit_point(const orc_scene *s, const orc_hit *h, v3 *P, int sse_order)
{
    const v3 *V = (const v3 *)s->v;
    v3 A = V[s->vi[3*h->prim]], B = V[s->vi[3*h->prim+1]], C = V[s->vi[3*h->prim+2]];
    v3 BmA = v3sub(B, A), CmA = v3sub(C, A);
    if (sse_order)   /* A + (beta*BmA + gamma*CmA), BVH.cpp:402 */
        *P = v3add(A, v3add(v3scale(BmA, h->beta), v3scale(CmA, h->gamma)));
    else             /* (A + beta*BmA) + gamma*CmA, Triangle.cpp:160 */
        *P = v3add(v3add(A, v3scale(BmA, h->beta)), v3scale(CmA, h->gamma));
}

void orc_surface(const orc_scene *s, const orc_ray *ray, const orc_hit *h, v3 *P, v3 *N, int sse_order)
{
    const uint32_t p = h->prim;
    if ((p & ORC_PLANE_BIT) || orc_is_sphere(s, p)) {
        v3 o = {ray->ox, ray->oy, ray->oz}, d = {ray->dx, ray->dy, ray->dz};
        *P = v3add(o, v3scale(d, h->t));                              /* ray.o + t*ray.d (Sphere.cpp:61, Plane.cpp:42) */
        if (p & ORC_PLANE_BIT) {
            const float *pl = s->planes + 6 * (size_t)(p & ~ORC_PLANE_BIT);
            N->x = pl[0]; N->y = pl[1]; N->z = pl[2];                  /* Plane.cpp:44 */
        } else {
            const float *sp = s->spheres + 4 * (size_t)s->vi[3 * (size_t)p + 1];
            v3 cen = {sp[0], sp[1], sp[2]};
            *N = v3normalize(v3sub(*P, cen));                         /* Sphere.cpp:62-63 */
        }
        return;
    }
    hit_point(s, h, P, sse_order);
    const v3 *Nn = (const v3 *)s->n;
    v3 nA = Nn[s->ni[3*p]], nB = Nn[s->ni[3*p+1]], nC = Nn[s->ni[3*p+2]];
    *N = v3add(v3add(v3scale(nA, 1 - h->beta - h->gamma), v3scale(nB, h->beta)), v3scale(nC, h->gamma));  /* Triangle.cpp:162 */
}

/* Phong::shade shadow ray (Phong.cpp:80-97) for a PointLight (PointLight.h:41-52) */
uint64_t orc_shadow_rays(const orc_scene *s, const orc_ray *rays, const orc_hit *hits, uint64_t n,
                         const float light[3], orc_ray *out, uint64_t *src, int sse_order)
{
    const float eps = 1e-4f;
    uint64_t k = 0;
    v3 L = {light[0], light[1], light[2]};
    for (uint64_t i = 0; i < n; i++) {
        if (hits[i].prim == ORC_MISS) continue;
        v3 P, Nunused;
        orc_surface(s, rays ? &rays[i] : NULL, &hits[i], &P, &Nunused, sse_order);
        v3 l = v3sub(L, P);                                           /* getLightDirection */
        float falloff = v3dot(l, l);
        float len = sqrtf(falloff);
        l = v3divs(l, len);                                           /* l /= sqrt(falloff) */
        v3 org = v3add(P, v3scale(l, eps));                           /* hit.P+(l*epsilon) */
        orc_ray *r = &out[k];
        r->ox = org.x; r->oy = org.y; r->oz = org.z; r->tmin = 0.0f;
        r->dx = l.x; r->dy = l.y; r->dz = l.z; r->tmax = len;
        if (src) src[k] = i;
        k++;
    }
    return k;
}

void orc_hit_attrs_rays(const orc_scene *s, const orc_ray *rays, const orc_hit *hits, uint64_t n, float *Pout, float *Nout)
{
    for (uint64_t i = 0; i < n; i++) {
        v3 P = {0, 0, 0}, N = {0, 1, 0};                              /* HitInfo defaults Ray.h:31-34 */
        if (hits[i].prim != ORC_MISS) orc_surface(s, rays ? &rays[i] : NULL, &hits[i], &P, &N, 0);
        if (Pout) { Pout[3*i] = P.x; Pout[3*i+1] = P.y; Pout[3*i+2] = P.z; }
        if (Nout) { Nout[3*i] = N.x; Nout[3*i+1] = N.y; Nout[3*i+2] = N.z; }
    }
}

void orc_hit_attrs(const orc_scene *s, const orc_hit *hits, uint64_t n, float *Pout, float *Nout)
{
    orc_hit_attrs_rays(s, NULL, hits, n, Pout, Nout);
}
