// mr_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the Miro intersection path.
//
//   trace_kernel        Scene::trace -> BVH::intersect -> BVH::intersectChildren -> Triangle::intersect
//                       (Scene.cpp:214-268, BVH.cpp:438-658 scalar branch, Triangle.cpp:136-169)
//   eye_rays_kernel     Camera::eyeRay (Camera.cpp:104-161)
//   shadow_rays_kernel  Phong::shade shadow ray (Phong.cpp:80-97) + wave64 ballot compaction
//   hit_attrs_kernel    HitInfo::P / ::N (Triangle.cpp:160,162)
//
// Compiled with -ffp-contract=off: in the default ("exact") mode every fp32 operation below is one
// individually rounded IEEE op in the reference's order, so t / beta / gamma are bit-identical to the
// reference's scalar build.  MR_MATH_FAST opts into explicit fmaf + v_rcp_f32.
//
// Traversal is pointer chasing, not a contraction: no MFMA.  One ray per lane, the pending-node stack
// lives in LDS as stack[level][thread] (consecutive lanes -> consecutive banks, conflict-free for
// ds_read/write_b32), nodes are 64-byte records fetched as 4 x dwordx4, triangles 48-byte records
// fetched as 3 x dwordx4 (layout in mr_internal.h).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mr_internal.h"
#include "mr_surface.h"
#include "mr_tile.h"

namespace mr {
namespace {

constexpr int kBlock = 256;          // 4 waves per workgroup
#ifndef MIRO_TRACE_BLOCK
#define MIRO_TRACE_BLOCK 256
#endif
constexpr int kTraceBlock = MIRO_TRACE_BLOCK;   // threads per workgroup of the trace kernels (their LDS stack is [depth][kTraceBlock])
constexpr int kTraceGridCap = 32768; // workgroups per trace launch (see launch_trace_t)
// Default 11 = min/max slabs on (corner - o) * (1/d) + while-while + wave-uniform nodes/leaves through the scalar
// cache: bit-identical to variant 0 (the literal select form) on 3 scenes x (33 M primary + 33 M shadow + 16 M random rays), tools/ab_variants.py.  Variant 7 (lean fma
// slabs) is NOT: a handful of shadow rays per 33 M change (boxes ending within ~1e-6 of a ray origin on a surface),
// for a 2 % gain -- it is used by MR_MATH_FAST only.
constexpr int kDefaultVariant = 11;  // see trace_kernel's VAR and trace_variant()
constexpr float kEps = 1e-4f;        // Miro.h:9
constexpr float kInf = __builtin_huge_valf();

struct Stats { unsigned long long box, tri; };

// ---------------------------------------------------------------------------------------------------
// slab test of one box.  EXACT keeps the reference's predicate structure literally (BVH.cpp:599-608):
// NaNs (0 * inf when the origin sits on a slab plane of an axis the ray does not move along) fall
// through every comparison.  `inv` is 1/d, correctly rounded; STRICT divides instead (bit-equal to
// the reference's (corner - o) / d, used when the -DSTATS counters must match exactly).
// ---------------------------------------------------------------------------------------------------
template <bool STRICT>
__device__ __forceinline__ void slab_axis(float lo, float hi, float o, float d, float inv, float &mn, float &mx) {
    float t0, t1;
    if (STRICT) { t0 = (lo - o) / d; t1 = (hi - o) / d; }
    else        { t0 = (lo - o) * inv; t1 = (hi - o) * inv; }
    const bool m = t0 > t1;
    const float tnear = m ? t1 : t0, tfar = m ? t0 : t1;
    if (tnear > mn) mn = tnear;
    if (tfar < mx) mx = tfar;
}

struct RayRegs {
    float ox, oy, oz, dx, dy, dz;     // origin, direction
    float ix, iy, iz;                 // 1/d
    float mx_, my_, mz_;              // -d (Triangle.cpp:152 uses dot(-r.d, ...))
    float tmin;
    float nox, noy, noz;              // -(o * 1/d): slab distance = fma(corner, 1/d, nox)   (lean slab form)
};

__device__ __forceinline__ void ray_setup(RayRegs &r, const float4 ra, const float4 rb) {
    r.ox = ra.x; r.oy = ra.y; r.oz = ra.z; r.tmin = ra.w;
    r.dx = rb.x; r.dy = rb.y; r.dz = rb.z;
    r.ix = 1.0f / r.dx; r.iy = 1.0f / r.dy; r.iz = 1.0f / r.dz;
    r.mx_ = -r.dx; r.my_ = -r.dy; r.mz_ = -r.dz;
    r.nox = -(r.ox * r.ix); r.noy = -(r.oy * r.iy); r.noz = -(r.oz * r.iz);
}

// A slab distance (corner - o) * (1/d) -- or fma(corner, 1/d, -(o/d)) -- can only be NaN as 0*inf, inf*0 or
// inf-inf: with o, d, 1/d and o/d all finite (corners are finite or +-inf) none of these can occur, and the
// select form of the reference and the min/max forms take the same decisions.
__device__ __forceinline__ bool lane_is_nan_free(const RayRegs &r) {
    return (__builtin_fabsf(r.ox) < kInf) && (__builtin_fabsf(r.oy) < kInf) && (__builtin_fabsf(r.oz) < kInf) &&
           (__builtin_fabsf(r.dx) < kInf) && (__builtin_fabsf(r.dy) < kInf) && (__builtin_fabsf(r.dz) < kInf) &&
           (__builtin_fabsf(r.ix) < kInf) && (__builtin_fabsf(r.iy) < kInf) && (__builtin_fabsf(r.iz) < kInf) &&
           (__builtin_fabsf(r.nox) < kInf) && (__builtin_fabsf(r.noy) < kInf) && (__builtin_fabsf(r.noz) < kInf);
}

// three-input min/max in one VALU op; inline asm so that no canonicalising v_max x,x is inserted
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float o;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}
__device__ __forceinline__ float vmin3(float a, float b, float c) {
    float o;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}

// two-input min/max as single VALU ops (no canonicalising v_max x,x in front, no NaN quieting: callers guarantee
// NaN-free operands or want exactly the hardware's minNum/maxNum behaviour)
__device__ __forceinline__ float vmax2(float a, float b) {
    float o;
    asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
    return o;
}
__device__ __forceinline__ float vmin2(float a, float b) {
    float o;
    asm("v_min_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
    return o;
}

// min/max slab test of one child box for NaN-free rays on the reference's own products (corner - o) * (1/d):
// 6 sub + 6 mul + 3 min + 3 max + max3 + min3.  Same entry/exit values as the select chain of slab_axis (up to
// the sign of a zero, which no comparison sees).
__device__ __forceinline__ void slab_box_minmax(float lox, float hix, float loy, float hiy, float loz, float hiz,
                                                const RayRegs &r, float &mn, float &mx) {
    const float ax = (lox - r.ox) * r.ix, bx = (hix - r.ox) * r.ix;
    const float ay = (loy - r.oy) * r.iy, by = (hiy - r.oy) * r.iy;
    const float az = (loz - r.oz) * r.iz, bz = (hiz - r.oz) * r.iz;
    mn = vmax3(vmin2(ax, bx), vmin2(ay, by), vmin2(az, bz));
    mx = vmin3(vmax2(ax, bx), vmax2(ay, by), vmax2(az, bz));
}

// The reference's quotient (corner - o) / d without a division: with inv = RN(1/d) and q = RN(a * inv), one
// correction step q' = fma(fma(-q, d, a), inv, q) is the correctly rounded a / d (Markstein's theorem; checked on this
// GPU against v_div_* over 1.1e11 operand pairs incl. every mantissa of a and of d, tools/div_identity_probe.hip) as
// long as nothing under- or overflows on the way.  Callers guarantee that: "regular" rays (lane_is_regular) on
// "regular" nodes (flag in the node record) keep a, d, q well inside the normal range.
__device__ __forceinline__ float exact_quot(float a, float d, float inv) {
    const float q = a * inv;
    return __builtin_fmaf(__builtin_fmaf(-q, d, a), inv, q);
}

// min/max slab test on those exact quotients: the reference's entry/exit distances themselves (up to the sign of a
// zero), 12 VALU more per box than slab_box_minmax and none of its tie caveats.
__device__ __forceinline__ void slab_box_exactq(float lox, float hix, float loy, float hiy, float loz, float hiz,
                                                const RayRegs &r, float &mn, float &mx) {
    const float ax = exact_quot(lox - r.ox, r.dx, r.ix), bx = exact_quot(hix - r.ox, r.dx, r.ix);
    const float ay = exact_quot(loy - r.oy, r.dy, r.iy), by = exact_quot(hiy - r.oy, r.dy, r.iy);
    const float az = exact_quot(loz - r.oz, r.dz, r.iz), bz = exact_quot(hiz - r.oz, r.dz, r.iz);
    mn = vmax3(vmin2(ax, bx), vmin2(ay, by), vmin2(az, bz));
    mx = vmin3(vmax2(ax, bx), vmax2(ay, by), vmax2(az, bz));
}

// magnitudes for which exact_quot is safe: direction components in [2^-40, 2^40], origin components 0 or in
// [2^-36, 2^60] (node corners obey the same bound when the node record's flag is clear, mr_api.cpp), so that a non-zero
// corner - o is at least 2^-59 and every intermediate stays a normal number
__device__ __forceinline__ bool regular_dir(float d) { const float a = __builtin_fabsf(d); return a >= 0x1p-40f && a <= 0x1p40f; }
__device__ __forceinline__ bool regular_pos(float o) { const float a = __builtin_fabsf(o); return a == 0.0f || (a >= 0x1p-36f && a <= 0x1p60f); }
__device__ __forceinline__ bool lane_is_regular(const RayRegs &r) {
    return regular_dir(r.dx) && regular_dir(r.dy) && regular_dir(r.dz) && regular_pos(r.ox) && regular_pos(r.oy) && regular_pos(r.oz);
}

// Lean slab test of one child box for NaN-free rays: 6 fma + 3 min + 3 max + max3 + min3.  Entry/exit
// distances differ from (corner - o) * (1/d) by rounding only; the decisions taken from them (cull, order) are
// protected by the epsilon padding of every box (BVH.cpp:75-79) -- see DESIGN.md section 5.
__device__ __forceinline__ void slab_box_lean(float lox, float hix, float loy, float hiy, float loz, float hiz,
                                              const RayRegs &r, float &mn, float &mx) {
    const float ax = fmaf(lox, r.ix, r.nox), bx = fmaf(hix, r.ix, r.nox);
    const float ay = fmaf(loy, r.iy, r.noy), by = fmaf(hiy, r.iy, r.noy);
    const float az = fmaf(loz, r.iz, r.noz), bz = fmaf(hiz, r.iz, r.noz);
    mn = vmax3(fminf(ax, bx), fminf(ay, by), fminf(az, bz));
    mx = vmin3(fmaxf(ax, bx), fmaxf(ay, by), fmaxf(az, bz));
}

// ---------------------------------------------------------------------------------------------------
// Triangle::intersect (Triangle.cpp:150-158).  q0..q2 = the 48-byte record.  Returns true when the
// reference's reject test passes with tMax = best; outputs t, beta, gamma.
// ---------------------------------------------------------------------------------------------------
template <bool EXACT>
__device__ __forceinline__ bool tri_test(const float4 q0, const float4 q1, const float4 q2, const RayRegs &r,
                                         float tmax, float &t, float &beta, float &gamma) {
    const float Ax = q0.x, Ay = q0.y, Az = q0.z;
    const float Bx = q0.w, By = q1.x, Bz = q1.y;      // B - A
    const float Cx = q1.z, Cy = q1.w, Cz = q2.x;      // C - A
    const float nx = q2.y, ny = q2.z, nz = q2.w;      // (B-A) x (C-A)
    const float px = r.ox - Ax, py = r.oy - Ay, pz = r.oz - Az;   // o - A
    if (EXACT) {
        const float ddotn = (r.mx_ * nx + r.my_ * ny) + r.mz_ * nz;
        t = ((px * nx + py * ny) + pz * nz) / ddotn;
        // cross(o-A, C-A)
        const float ux = py * Cz - pz * Cy, uy = pz * Cx - px * Cz, uz = px * Cy - py * Cx;
        beta = ((r.mx_ * ux + r.my_ * uy) + r.mz_ * uz) / ddotn;
        // cross(B-A, o-A)
        const float wx = By * pz - Bz * py, wy = Bz * px - Bx * pz, wz = Bx * py - By * px;
        gamma = ((r.mx_ * wx + r.my_ * wy) + r.mz_ * wz) / ddotn;
    } else {
        const float ddotn = fmaf(r.mz_, nz, fmaf(r.my_, ny, r.mx_ * nx));
        const float rcp = __builtin_amdgcn_rcpf(ddotn);
        t = fmaf(pz, nz, fmaf(py, ny, px * nx)) * rcp;
        const float ux = fmaf(py, Cz, -(pz * Cy)), uy = fmaf(pz, Cx, -(px * Cz)), uz = fmaf(px, Cy, -(py * Cx));
        beta = fmaf(r.mz_, uz, fmaf(r.my_, uy, r.mx_ * ux)) * rcp;
        const float wx = fmaf(By, pz, -(Bz * py)), wy = fmaf(Bz, px, -(Bx * pz)), wz = fmaf(Bx, py, -(By * px));
        gamma = fmaf(r.mz_, wz, fmaf(r.my_, wy, r.mx_ * wx)) * rcp;
    }
    // reject iff beta < -eps || gamma < -eps || beta+gamma > 1+eps || t < tMin || t > tMax  (:158)
    const bool reject = (beta < -kEps) || (gamma < -kEps) || (beta + gamma > 1 + kEps) || (t < r.tmin) || (t > tmax);
    return !reject;
}

// Sphere::intersect (Sphere.cpp:28-69) on the record (c.xyz, radius): the quadratic in the reference's order of
// operations, true divisions, strict range test on both roots.
__device__ __forceinline__ bool sphere_test(const float4 q0, const RayRegs &r, float tmax, float &t) {
    const float tx = r.ox - q0.x, ty = r.oy - q0.y, tz = r.oz - q0.z;       // toO = ray.o - m_center
    const float a = (r.dx * r.dx + r.dy * r.dy) + r.dz * r.dz;               // ray.d.length2()
    const float b = ((r.dx * 2) * tx + (r.dy * 2) * ty) + (r.dz * 2) * tz;   // dot(2*ray.d, toO)
    const float c = ((tx * tx + ty * ty) + tz * tz) - q0.w * q0.w;
    const float discrim = b * b - 4.0f * a * c;
    if (discrim < 0) return false;
    const float sq = sqrtf(discrim);
    const float t0 = (-b - sq) / (2.0f * a), t1 = (-b + sq) / (2.0f * a);
    if ((t0 > r.tmin) && (t0 < tmax)) { t = t0; return true; }
    if ((t1 > r.tmin) && (t1 < tmax)) { t = t1; return true; }
    return false;
}

// the object test of a leaf slot: Triangle::intersect, or Sphere::intersect when OBJ and the record carries the tag
template <bool EXACT, bool OBJ>
__device__ __forceinline__ bool object_test(const float4 q0, const float4 q1, const float4 q2, const RayRegs &r,
                                            float tmax, float &t, float &beta, float &gamma) {
    if (OBJ && __float_as_uint(q2.w) == kSphereTag) {
        beta = 0.0f; gamma = 0.0f;
        return sphere_test(q0, r, tmax, t);
    }
    return tri_test<EXACT>(q0, q1, q2, r, tmax, t, beta, gamma);
}

// ---------------------------------------------------------------------------------------------------
// closest-hit / any-hit traversal, one ray per lane.
// VAR bit 0: when no lane of the wave can produce a NaN in a slab product (o, d, 1/d all finite -- wave-uniform
//            test via __all), the select chains of the slab test collapse to v_min/v_max, which give the same
//            decisions (they differ only in the sign of a zero);
// VAR bit 1: "while-while" control flow: lanes run inner nodes until each holds a leaf (or is done), then the
//            wave does the leaves together -- same per-lane visiting order, better SIMD utilisation in the
//            triangle loop.
// ---------------------------------------------------------------------------------------------------
// `cur` is the node the lane is at: >= 0 inner node, < 0 leaf reference, kDone = no more work.  `sp` is the BYTE
// offset in LDS of the lane's next free stack slot (slots of one lane are kTraceBlock * 4 bytes apart); the bottom slot
// of every lane holds kDone, so a pop needs no emptiness test: popping the sentinel ends the ray.
constexpr int kDone = (int)0x80000000;
constexpr int kStackStride = kTraceBlock * (int)sizeof(int);
struct Lane {
    float best_t, best_b, best_g;
    int best_pos;
    int sp, cur;
    __device__ __forceinline__ bool have() const { return cur != kDone; }
};

__device__ __forceinline__ void stack_push(Lane &L, int *s_stack, int v) {
    *reinterpret_cast<int *>(reinterpret_cast<char *>(s_stack) + L.sp) = v;
    L.sp += kStackStride;
}
__device__ __forceinline__ int stack_pop(Lane &L, int *s_stack) {
    L.sp -= kStackStride;
    return *reinterpret_cast<int *>(reinterpret_cast<char *>(s_stack) + L.sp);
}
__device__ __forceinline__ void stack_reset(Lane &L, int *s_stack, int tid) {
    L.sp = tid * (int)sizeof(int);
    stack_push(L, s_stack, kDone);
}

// One 64-byte node record through the scalar data cache: when every active lane of the wave sits at the same
// node (coherent camera / shadow rays near the top of the tree), one s_load_dwordx16 replaces 64 lanes x 4
// global_load_dwordx4 -- the vector L1 (64 B/clk/CU) is what bounds this kernel (profiles/r01_pmc_sq.txt).
typedef float v16f __attribute__((ext_vector_type(16)));
__device__ __forceinline__ v16f load_node_scalar(const float4 *nodes, int cur_uniform) {
    const float4 *ptr = nodes + 4 * (size_t)cur_uniform;
    v16f v;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ptr) : "memory");
    return v;
}

template <bool EXACT, bool STATS, int SLAB>
__device__ __forceinline__ void node_slabs(const float4 q0, const float4 q1, const float4 q2, const RayRegs &r,
                                           float &mn0, float &mx0, float &mn1, float &mx1);

// the post-test bookkeeping of BVH.cpp:609-651: near child first (ties -> child 0), far child pushed, else pop
template <bool STATS, bool SAFE>
__device__ __forceinline__ void node_decide(float mn0, float mx0, float mn1, float mx1, int ref0, int ref1,
                                            const RayRegs &r, Lane &L, int *s_stack, Stats &st) {
    // tMax of this call == best_t: nothing changed since the node was entered
    bool h0, h1;
    if (SAFE) {
        // mn, mx are not NaN here; (mn > mx || mn > best) == (mn > minNum(mx, best)) also when best is NaN
        h0 = !((mn0 > vmin2(mx0, L.best_t)) || (mx0 < r.tmin));
        h1 = !((mn1 > vmin2(mx1, L.best_t)) || (mx1 < r.tmin));
    } else {
        h0 = !((mn0 > mx0) || (mn0 > L.best_t) || (mx0 < r.tmin));
        h1 = !((mn1 > mx1) || (mn1 > L.best_t) || (mx1 < r.tmin));
    }
    const bool one_first = h1 && (!h0 || (mn0 > mn1));
    // (a select-only formulation with predicated push/pop was measured 5 % slower than this branch nest)
    if (h0 && h1) {
        stack_push(L, s_stack, one_first ? ref0 : ref1);
        L.cur = one_first ? ref1 : ref0;
        if (STATS) st.box++;
    } else if (h0 || h1) {
        L.cur = h0 ? ref0 : ref1;
        if (STATS) st.box++;
    } else {
        L.cur = stack_pop(L, s_stack);        // the far child is entered unconditionally (:640-650); kDone at the bottom
        if (STATS && L.cur != kDone) st.box++;
    }
}

// SLAB: 0 = select form (the reference's NaN semantics) on (corner - o) * (1/d), 1 = min/max on the same products,
//       2 = lean fma form, 3 = select form on the reference's true quotients (corner - o) / d
// SCALAR: try the wave-uniform scalar-load path first
template <bool EXACT, bool STATS, int SLAB, bool SCALAR = false>
__device__ __forceinline__ void node_step(const TraceParams &p, const RayRegs &r, Lane &L, int *s_stack, int tid, Stats &st) {
    float mn0, mx0, mn1, mx1;
    constexpr bool kSafe = SLAB == 1 || SLAB == 2 || SLAB == 4;
    if (SCALAR) {
        const int cur0 = __builtin_amdgcn_readfirstlane(L.cur);
        if (__all(L.cur == cur0)) {
            const v16f v = load_node_scalar(p.nodes, cur0);
            const float4 q0 = make_float4(v[0], v[1], v[2], v[3]), q1 = make_float4(v[4], v[5], v[6], v[7]);
            const float4 q2 = make_float4(v[8], v[9], v[10], v[11]);
            if (SLAB == 4 && __float_as_int(v[14]) != 0) {   // irregular node (wave-uniform): the reference's own divisions
                node_slabs<EXACT, STATS, 3>(q0, q1, q2, r, mn0, mx0, mn1, mx1);
                node_decide<STATS, false>(mn0, mx0, mn1, mx1, __float_as_int(v[12]), __float_as_int(v[13]), r, L, s_stack, st);
                return;
            }
            node_slabs<EXACT, STATS, SLAB>(q0, q1, q2, r, mn0, mx0, mn1, mx1);
            node_decide<STATS, kSafe>(mn0, mx0, mn1, mx1, __float_as_int(v[12]), __float_as_int(v[13]), r, L, s_stack, st);
            return;
        }
    }
    // ---- inner node: test both children (BVH.cpp:593-624)
    const float4 *nd = p.nodes + 4 * (size_t)L.cur;
    const float4 q0 = nd[0], q1 = nd[1], q2 = nd[2];
    const int4 q3 = *reinterpret_cast<const int4 *>(nd + 3);
    if (SLAB == 4 && q3.z != 0) {
        node_slabs<EXACT, STATS, 3>(q0, q1, q2, r, mn0, mx0, mn1, mx1);
        node_decide<STATS, false>(mn0, mx0, mn1, mx1, q3.x, q3.y, r, L, s_stack, st);
        return;
    }
    node_slabs<EXACT, STATS, SLAB>(q0, q1, q2, r, mn0, mx0, mn1, mx1);
    node_decide<STATS, kSafe>(mn0, mx0, mn1, mx1, q3.x, q3.y, r, L, s_stack, st);
}

template <bool EXACT, bool STATS, int SLAB>
__device__ __forceinline__ void node_slabs(const float4 q0, const float4 q1, const float4 q2, const RayRegs &r,
                                           float &mn0, float &mx0, float &mn1, float &mx1) {
    if (SLAB == 2) {
        slab_box_lean(q0.x, q0.y, q0.z, q0.w, q2.x, q2.y, r, mn0, mx0);
        slab_box_lean(q1.x, q1.y, q1.z, q1.w, q2.z, q2.w, r, mn1, mx1);
    } else if (SLAB == 4) {
        slab_box_exactq(q0.x, q0.y, q0.z, q0.w, q2.x, q2.y, r, mn0, mx0);
        slab_box_exactq(q1.x, q1.y, q1.z, q1.w, q2.z, q2.w, r, mn1, mx1);
    } else if (EXACT && (SLAB == 0 || SLAB == 3)) {
        mn0 = -kInf; mx0 = kInf; mn1 = -kInf; mx1 = kInf;
        slab_axis<SLAB == 3>(q0.x, q0.y, r.ox, r.dx, r.ix, mn0, mx0);
        slab_axis<SLAB == 3>(q0.z, q0.w, r.oy, r.dy, r.iy, mn0, mx0);
        slab_axis<SLAB == 3>(q2.x, q2.y, r.oz, r.dz, r.iz, mn0, mx0);
        slab_axis<SLAB == 3>(q1.x, q1.y, r.ox, r.dx, r.ix, mn1, mx1);
        slab_axis<SLAB == 3>(q1.z, q1.w, r.oy, r.dy, r.iy, mn1, mx1);
        slab_axis<SLAB == 3>(q2.z, q2.w, r.oz, r.dz, r.iz, mn1, mx1);
    } else {
        slab_box_minmax(q0.x, q0.y, q0.z, q0.w, q2.x, q2.y, r, mn0, mx0);
        slab_box_minmax(q1.x, q1.y, q1.z, q1.w, q2.z, q2.w, r, mn1, mx1);
    }
}

// one 48-byte triangle record through the scalar data cache (all active lanes at the same leaf)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_tri_scalar(const float4 *tris, unsigned pos_uniform, float4 &q0, float4 &q1, float4 &q2) {
    const float4 *ptr = tris + 3 * (size_t)pos_uniform;
    v4f a, b, c;
    asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_load_dwordx4 %2, %3, 0x20\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c) : "s"(ptr) : "memory");
    q0 = make_float4(a[0], a[1], a[2], a[3]);
    q1 = make_float4(b[0], b[1], b[2], b[3]);
    q2 = make_float4(c[0], c[1], c[2], c[3]);
}

template <bool EXACT, bool ANY, bool STATS, bool SCALAR = false, bool OBJ = false>
__device__ __forceinline__ void leaf_step(const TraceParams &p, const RayRegs &r, Lane &L, int *s_stack, int tid, Stats &st) {
    // ---- leaf (BVH.cpp:493-509)
    const unsigned bits = ~(unsigned)L.cur;
    const unsigned first = bits >> kLeafCountBits;
    unsigned cnt = bits & kLeafCountMask;
    if (cnt == kLeafCountMask) cnt = p.leaf_cnt_ext[first];
    bool done = false;
    bool uniform = false;
    if (SCALAR) {
        const int cur0 = __builtin_amdgcn_readfirstlane(L.cur);
        uniform = __all(L.cur == cur0);
        if (uniform) {
            const unsigned first0 = (unsigned)__builtin_amdgcn_readfirstlane((int)first);
            const unsigned cnt0 = (unsigned)__builtin_amdgcn_readfirstlane((int)cnt);
            for (unsigned k = 0; k < cnt0; k++) {
                float4 q0, q1, q2;
                load_tri_scalar(p.tris, first0 + k, q0, q1, q2);
                if (!(ANY && done)) {
                    float t, b, g;
                    const bool ok = object_test<EXACT, OBJ>(q0, q1, q2, r, L.best_t, t, b, g);
                    if (ok && t < L.best_t) {
                        L.best_t = t; L.best_b = b; L.best_g = g; L.best_pos = (int)(first0 + k);
                        if (ANY) done = true;
                    }
                }
            }
        }
    }
    if (!uniform) {
        for (unsigned k = 0; k < cnt; k++) {
            const float4 *tr = p.tris + 3 * (size_t)(first + k);
            float t, b, g;
            const bool ok = object_test<EXACT, OBJ>(tr[0], tr[1], tr[2], r, L.best_t, t, b, g);
            if (ok && t < L.best_t) {             // strict-less replacement (:500)
                L.best_t = t; L.best_b = b; L.best_g = g; L.best_pos = (int)(first + k);
                if (ANY) { done = true; break; }
            }
        }
    }
    if (STATS) {
        if (OBJ) {   // Stats::Ray_Tri_Intersect counts Triangle objects only (the dynamic_cast of BVH.cpp:496)
            for (unsigned k = 0; k < cnt; k++)
                if (__float_as_uint(p.tris[3 * (size_t)(first + k) + 2].w) != kSphereTag) st.tri++;
        } else {
            st.tri += cnt;
        }
    }
    if (ANY && done) {
        L.cur = kDone;
    } else {
        L.cur = stack_pop(L, s_stack);
        if (STATS && L.cur != kDone) st.box++;
    }
}

template <bool EXACT, bool ANY, bool STATS, int SLAB, bool WW, bool SCALAR, bool OBJ = false>
__device__ __forceinline__ void traverse(const TraceParams &p, const RayRegs &r, Lane &L, int *s_stack, int tid, Stats &st) {
    if (WW) {
        while (__any(L.have())) {
            while (L.cur >= 0) node_step<EXACT, STATS, SLAB, SCALAR>(p, r, L, s_stack, tid, st);
            if (L.have()) leaf_step<EXACT, ANY, STATS, SCALAR, OBJ>(p, r, L, s_stack, tid, st);
        }
    } else {
        while (L.have()) {
            if (L.cur >= 0) node_step<EXACT, STATS, SLAB, SCALAR>(p, r, L, s_stack, tid, st);
            else leaf_step<EXACT, ANY, STATS, false, OBJ>(p, r, L, s_stack, tid, st);
        }
    }
}

template <bool EXACT, bool ANY, bool STATS, int VAR>
__global__ __launch_bounds__(kTraceBlock) void trace_kernel(TraceParams p) {
    extern __shared__ int s_stack[];                  // [stack_depth][kTraceBlock]
    const int tid = threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * kTraceBlock;
    Stats st = {0ull, 0ull};
    // VAR bit 4: every slab distance is the reference's true quotient (the default trace; MR_COUNT_STATS implies it)
    constexpr bool kStrict = STATS || (VAR & 16);
    constexpr int kBaseSlab = kStrict ? 3 : 0;
    constexpr bool kMinMax = !kStrict && (VAR & 1);
    constexpr bool kWW = (VAR & 2) != 0;
    constexpr int kSafeSlab = (VAR & 4) ? 2 : 1;      // slab form for waves whose rays cannot produce a NaN
    constexpr bool kScalar = (VAR & 8) != 0;          // wave-uniform nodes through the scalar cache
    constexpr bool kObj = (VAR & 32) != 0;            // the scene holds spheres and / or planes

    // indirect batches: the ray count lives on the device (e.g. written by the shadow-ray compaction)
    unsigned long long n_rays = p.n;
    if (p.n_dev) { const unsigned long long nd = *p.n_dev; if (nd < n_rays) n_rays = nd; }
    const unsigned long long n_round = kWW ? ((n_rays + 63ull) & ~63ull) : n_rays;   // whole waves for __any

    for (unsigned long long idx = (unsigned long long)blockIdx.x * kTraceBlock + tid; idx < n_round; idx += stride) {
        const bool live = idx < n_rays;
        // two dwordx4 loads per lane, 32-byte stride: every byte of the fetched lines is used
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = make_float4(1.f, 1.f, 1.f, -1.f);
        if (live) {
            ra = reinterpret_cast<const float4 *>(p.rays)[2 * idx];
            rb = reinterpret_cast<const float4 *>(p.rays)[2 * idx + 1];
        }
        RayRegs r;
        ray_setup(r, ra, rb);
        const float tmax0 = rb.w;

        Lane L;
        L.best_t = tmax0;                             // minHit.t = tMax (BVH.cpp:444)
        L.best_b = 0.0f; L.best_g = 0.0f;
        L.best_pos = -1;                              // leaf-order position of the winning triangle
        stack_reset(L, s_stack, tid);                 // this lane's LDS stack: the kDone sentinel only
        {   // BVH::intersect root test (BVH.cpp:447-466)
            float mn = -kInf, mx = kInf;
            slab_axis<kStrict>(p.root_lo[0], p.root_hi[0], r.ox, r.dx, r.ix, mn, mx);
            slab_axis<kStrict>(p.root_lo[1], p.root_hi[1], r.oy, r.dy, r.iy, mn, mx);
            slab_axis<kStrict>(p.root_lo[2], p.root_hi[2], r.oz, r.dz, r.iz, mn, mx);
            if (STATS && live) st.box++;
            L.cur = (live && !((mn > mx) || (mn > tmax0) || (mx < r.tmin))) ? p.root_ref : kDone;
        }

        if (kMinMax) {
            // a slab product (corner - o) * (1/d) can only be NaN as 0*inf or inf*0 or from a non-finite origin:
            // with o, d and 1/d all finite in every lane the select form and the min/max form decide identically
            if (__all(lane_is_nan_free(r) || !live)) traverse<EXACT, ANY, STATS, kSafeSlab, kWW, kScalar, kObj>(p, r, L, s_stack, tid, st);
            else traverse<EXACT, ANY, STATS, 0, kWW, false, kObj>(p, r, L, s_stack, tid, st);
        } else if (kStrict && !STATS) {
            // the default trace: exact quotients by the correction step where every lane's ray is regular (then the lanes'
            // quotients are NaN-free too and the min/max form decides like the select form); the reference's own
            // divisions otherwise
            if (__all(lane_is_regular(r) || !live)) traverse<EXACT, ANY, STATS, 4, kWW, kScalar, kObj>(p, r, L, s_stack, tid, st);
            else traverse<EXACT, ANY, STATS, 3, kWW, kScalar, kObj>(p, r, L, s_stack, tid, st);
        } else {
            traverse<EXACT, ANY, STATS, kBaseSlab, kWW, kStrict && kScalar, kObj>(p, r, L, s_stack, tid, st);
        }

        // Scene::trace's scan of the unbounded objects (Scene.cpp:220-230): every plane is tested against the
        // caller's tMin / tMax (Plane.cpp:33-48) and kept when nothing was hit yet or it is strictly nearer
        int plane_hit = -1;
        if (kObj && live && !(ANY && L.best_pos >= 0)) {
            for (uint32_t k = 0; k < p.n_planes; k++) {
                const float4 pn = p.planes[2 * k], po = p.planes[2 * k + 1];
                const float ndotd = (pn.x * r.dx + pn.y * r.dy) + pn.z * r.dz;
                if ((double)__builtin_fabsf(ndotd) < 1e-6) continue;          // fabs(float) < double literal
                const float t = ((pn.x * (po.x - r.ox) + pn.y * (po.y - r.oy)) + pn.z * (po.z - r.oz)) / ndotd;
                if (t < r.tmin || t > tmax0) continue;
                if ((L.best_pos < 0 && plane_hit < 0) || t < L.best_t) { L.best_t = t; plane_hit = (int)k; }
            }
        }

        if (live) {
            mr_hit h;
            if (kObj && plane_hit >= 0) {
                h.t = L.best_t; h.prim = kPlaneBit | (uint32_t)plane_hit; h.beta = 0.0f; h.gamma = 0.0f;
            } else if (L.best_pos >= 0) {
                h.t = L.best_t; h.prim = p.tri_prim[L.best_pos]; h.beta = L.best_b; h.gamma = L.best_g;
            } else {
                h.t = tmax0; h.prim = MR_MISS; h.beta = 0.0f; h.gamma = 0.0f;
            }
            reinterpret_cast<float4 *>(p.hits)[idx] = *reinterpret_cast<const float4 *>(&h);
        }
    }

    if (STATS) {
        // wave64 reduction, one atomic pair per wave
        for (int off = 32; off > 0; off >>= 1) {
            st.box += __shfl_down(st.box, off, 64);
            st.tri += __shfl_down(st.tri, off, 64);
        }
        if ((tid & 63) == 0) {
            atomicAdd(&p.stats[0], st.box);
            atomicAdd(&p.stats[1], st.tri);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Persistent form for incoherent batches (secondary / shadow / random rays): a resident grid of waves pulls
// rays from a global counter.  Whenever at least REFILL_MIN lanes of a wave have finished their ray, the wave
// ballots the idle lanes, retires their hits, and re-arms them with fresh rays by prefix-sum over the ballot
// (wave64 active-ray compaction: lanes never wait for the slowest ray of a fixed group of 64).  Rays are handed
// out from a wave-local pool of kPoolChunk consecutive indices so that the global counter sees one atomic per
// chunk.  Per-ray work and results are those of trace_kernel: QUOT selects the default trace's exact quotients
// (correction step for regular rays, the reference's divisions otherwise), !QUOT the MR_MATH_PRODUCT arithmetic.
// ---------------------------------------------------------------------------------------------------
constexpr unsigned long long kPoolChunk = 1024;

template <bool EXACT, bool ANY, bool QUOT, int REFILL_MIN>
__global__ __launch_bounds__(kTraceBlock) void trace_persistent_kernel(TraceParams p, unsigned long long *next_ray) {
    extern __shared__ int s_stack[];
    const int tid = threadIdx.x, lane = tid & 63;
    Stats st = {0ull, 0ull};
    unsigned long long n_rays = p.n;
    if (p.n_dev) { const unsigned long long nd = *p.n_dev; if (nd < n_rays) n_rays = nd; }

    unsigned long long pool_next = 0, pool_end = 0;      // wave-uniform
    bool exhausted = false;                              // wave-uniform
    const unsigned long long kNone = ~0ull;
    unsigned long long my_idx = kNone;
    float tmax0 = 0.0f;
    bool safe_lane = true;
    RayRegs r = {};
    Lane L;
    L.best_t = 0.f; L.best_b = 0.f; L.best_g = 0.f; L.best_pos = -1; L.sp = 0; L.cur = kDone;

    while (true) {
        const unsigned long long idle = __ballot(!L.have());
        const int n_idle = __popcll(idle);
        if (n_idle >= REFILL_MIN || idle == ~0ull) {
            // retire the rays of the idle lanes
            if (!L.have() && my_idx != kNone) {
                mr_hit h;
                if (L.best_pos >= 0) { h.t = L.best_t; h.prim = p.tri_prim[L.best_pos]; h.beta = L.best_b; h.gamma = L.best_g; }
                else { h.t = tmax0; h.prim = MR_MISS; h.beta = 0.0f; h.gamma = 0.0f; }
                reinterpret_cast<float4 *>(p.hits)[my_idx] = *reinterpret_cast<const float4 *>(&h);
                my_idx = kNone;
            }
            if (pool_next == pool_end && !exhausted) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(next_ray, kPoolChunk);
                base = __shfl(base, 0, 64);
                if (base >= n_rays) { exhausted = true; pool_next = pool_end = 0; }
                else { pool_next = base; pool_end = base + kPoolChunk < n_rays ? base + kPoolChunk : n_rays; }
            }
            if (pool_next < pool_end) {
                const unsigned long long cand = pool_next + (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
                if (!L.have() && cand < pool_end) {
                    my_idx = cand;
                    const float4 ra = reinterpret_cast<const float4 *>(p.rays)[2 * cand];
                    const float4 rb = reinterpret_cast<const float4 *>(p.rays)[2 * cand + 1];
                    ray_setup(r, ra, rb);
                    tmax0 = rb.w;
                    L.best_t = tmax0; L.best_b = 0.0f; L.best_g = 0.0f; L.best_pos = -1;
                    stack_reset(L, s_stack, tid);
                    float mn = -kInf, mx = kInf;
                    slab_axis<QUOT>(p.root_lo[0], p.root_hi[0], r.ox, r.dx, r.ix, mn, mx);
                    slab_axis<QUOT>(p.root_lo[1], p.root_hi[1], r.oy, r.dy, r.iy, mn, mx);
                    slab_axis<QUOT>(p.root_lo[2], p.root_hi[2], r.oz, r.dz, r.iz, mn, mx);
                    L.cur = !((mn > mx) || (mn > tmax0) || (mx < r.tmin)) ? p.root_ref : kDone;
                    safe_lane = QUOT ? lane_is_regular(r) : lane_is_nan_free(r);
                }
                const unsigned long long adv = pool_next + (unsigned)n_idle;
                pool_next = adv < pool_end ? adv : pool_end;
            }
        }
        if (!__any(L.have())) {
            if (exhausted && pool_next == pool_end) {
                if (my_idx != kNone) {              // rays that missed the root box in the last hand-out
                    mr_hit h;
                    h.t = tmax0; h.prim = MR_MISS; h.beta = 0.0f; h.gamma = 0.0f;
                    reinterpret_cast<float4 *>(p.hits)[my_idx] = *reinterpret_cast<const float4 *>(&h);
                }
                break;
            }
            continue;
        }
        if (__all(safe_lane || !L.have())) {
            while (L.cur >= 0) node_step<EXACT, false, QUOT ? 4 : 1, true>(p, r, L, s_stack, tid, st);
        } else {
            while (L.cur >= 0) node_step<EXACT, false, QUOT ? 3 : 0, QUOT>(p, r, L, s_stack, tid, st);
        }
        if (L.have()) leaf_step<EXACT, ANY, false, true>(p, r, L, s_stack, tid, st);
    }
}

// ---------------------------------------------------------------------------------------------------
// Camera::eyeRay (Camera.cpp:104-161).  The camera frame is computed on the host exactly as the
// reference does; the per-pixel arithmetic below keeps the reference's operation order.
// ---------------------------------------------------------------------------------------------------
struct EyeFrame {
    float eye[3], u[3], v[3], w[3];
    float left, right, bottom, top;
    uint32_t W, H, y0, spp, jitter, hbase;
    uint32_t tiled, rows;        // tiled: ray order of mr_tile.h inside the window of `rows` rows
    TileShape tile;
    unsigned long long n;
};

__device__ __forceinline__ uint32_t pcg_hash(uint32_t x) {
    const uint32_t state = x * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
__device__ __forceinline__ float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

__global__ __launch_bounds__(kBlock) void eye_rays_kernel(EyeFrame f, mr_ray *rays) {
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < f.n; k += stride) {
        const unsigned long long pix_local = k / f.spp;
        const uint32_t sm = (uint32_t)(k - pix_local * f.spp);
        uint32_t y = f.y0 + (uint32_t)(pix_local / f.W), x = (uint32_t)(pix_local % f.W);
        if (f.tiled) {
            uint32_t yl;
            tile_decode((uint32_t)pix_local, f.W, f.rows, f.tile, x, yl);
            y = f.y0 + yl;
        }
        float dx = 0.5f, dy = 0.5f;
        if (f.jitter) {
            const uint32_t pix = y * f.W + x;
            const uint32_t b = pcg_hash(pcg_hash(f.hbase ^ pix) + sm);
            dx = u01(pcg_hash(b));
            dy = u01(pcg_hash(b ^ 0x68bc21ebu));
        }
        const float up = f.left + (f.right - f.left) * (((float)x + dx) / (float)f.W);
        const float vp = f.bottom + (f.top - f.bottom) * (((float)y + dy) / (float)f.H);
        float ddx = (up * f.u[0] + vp * f.v[0]) - f.w[0];
        float ddy = (up * f.u[1] + vp * f.v[1]) - f.w[1];
        float ddz = (up * f.u[2] + vp * f.v[2]) - f.w[2];
        const float len = sqrtf((ddx * ddx + ddy * ddy) + ddz * ddz);
        const float inv = 1.0f / len;
        float4 a = make_float4(f.eye[0], f.eye[1], f.eye[2], 0.0f);
        float4 b = make_float4(ddx * inv, ddy * inv, ddz * inv, 1e12f);     // MIRO_TMAX
        reinterpret_cast<float4 *>(rays)[2 * k] = a;
        reinterpret_cast<float4 *>(rays)[2 * k + 1] = b;
    }
}

// ---------------------------------------------------------------------------------------------------
// shadow rays (Phong.cpp:80-97): origin P + l*eps, direction l = normalise(L - P), tMax = |L - P|.
// Hits are compacted wave-by-wave: ballot of hitting lanes, one atomicAdd per wave for the base,
// mbcnt prefix for the lane's slot.
// ---------------------------------------------------------------------------------------------------
// Compaction is hierarchical so that the global counter sees one atomic per 2048 rays, not one per wave
// (a single counter word saturates at ~88 atomics/us, MI355X_MICROARCH.md "dequeue"): each workgroup takes
// chunks of kBlock*kShIter rays; every wave ballots its kShIter sub-rows (wave64 __ballot + popcount
// prefix), wave totals meet in LDS, lane 0 reserves the chunk's range with one atomicAdd, and each lane
// writes at  chunk base + waves before mine + sub-rows before this one + lanes before mine.
constexpr int kShIter = 8;

__global__ __launch_bounds__(kBlock) void shadow_rays_kernel(SurfacePtrs m, const mr_ray *rays, const mr_hit *hits, unsigned long long n,
                                                             float Lx, float Ly, float Lz, mr_ray *out,
                                                             uint32_t *src, unsigned long long *count) {
    __shared__ unsigned s_wave_total[kBlock / 64];
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long chunk = (unsigned long long)kBlock * kShIter;
    const unsigned long long n_chunks = (n + chunk - 1) / chunk;
    for (unsigned long long c = blockIdx.x; c < n_chunks; c += gridDim.x) {      // uniform per workgroup
        float4 h[kShIter];
        unsigned prefix[kShIter];     // hits of this wave in earlier sub-rows + earlier lanes of this one
        unsigned wave_hits = 0;
#pragma unroll
        for (int it = 0; it < kShIter; it++) {
            const unsigned long long k = c * chunk + (unsigned long long)it * kBlock + threadIdx.x;
            bool is_hit = false;
            h[it] = make_float4(0.f, __uint_as_float(MR_MISS), 0.f, 0.f);
            if (k < n) {
                h[it] = reinterpret_cast<const float4 *>(hits)[k];
                is_hit = __float_as_uint(h[it].y) != MR_MISS;
            }
            const unsigned long long mask = __ballot(is_hit);
            prefix[it] = wave_hits + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
            wave_hits += (unsigned)__popcll(mask);
        }
        if (lane == 0) s_wave_total[wave] = wave_hits;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned tot = 0;
            for (int w = 0; w < kBlock / 64; w++) tot += s_wave_total[w];
            s_base = tot ? atomicAdd(count, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        unsigned long long wave_base = s_base;
        for (int w = 0; w < wave; w++) wave_base += s_wave_total[w];
#pragma unroll
        for (int it = 0; it < kShIter; it++) {
            if (__float_as_uint(h[it].y) == MR_MISS) continue;
            const unsigned long long k = c * chunk + (unsigned long long)it * kBlock + threadIdx.x;
            const unsigned long long slot = wave_base + prefix[it];
            float P[3];
            surface<false>(m, rays, k, h[it].x, __float_as_uint(h[it].y), h[it].z, h[it].w, P, nullptr);
            const float Px = P[0], Py = P[1], Pz = P[2];
            float lx = Lx - Px, ly = Ly - Py, lz = Lz - Pz;           // PointLight::getLightDirection
            const float falloff = (lx * lx + ly * ly) + lz * lz;
            const float len = sqrtf(falloff);
            const float inv = 1.0f / len;                              // l /= sqrt(falloff)
            lx *= inv; ly *= inv; lz *= inv;
            float4 a = make_float4(Px + lx * kEps, Py + ly * kEps, Pz + lz * kEps, 0.0f);
            float4 b = make_float4(lx, ly, lz, len);
            reinterpret_cast<float4 *>(out)[2 * slot] = a;
            reinterpret_cast<float4 *>(out)[2 * slot + 1] = b;
            if (src) src[slot] = (uint32_t)k;
        }
        __syncthreads();              // s_wave_total / s_base are reused by the next chunk
    }
}

__global__ __launch_bounds__(kBlock) void hit_attrs_kernel(SurfacePtrs m, const mr_ray *rays, const mr_hit *hits,
                                                           unsigned long long n, float *P, float *N) {
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < n; k += stride) {
        const float4 h = reinterpret_cast<const float4 *>(hits)[k];
        const uint32_t prim = __float_as_uint(h.y);
        float Pv[3] = {0.f, 0.f, 0.f}, Nv[3] = {0.f, 1.f, 0.f};               // HitInfo defaults (Ray.h:31-34)
        if (prim != MR_MISS) surface<true>(m, rays, k, h.x, prim, h.z, h.w, Pv, Nv);
        if (P) { P[3 * k] = Pv[0]; P[3 * k + 1] = Pv[1]; P[3 * k + 2] = Pv[2]; }
        if (N) { N[3 * k] = Nv[0]; N[3 * k + 1] = Nv[1]; N[3 * k + 2] = Nv[2]; }
    }
}

inline unsigned grid_for(unsigned long long n) {
    // memory/latency-bound kernels: cap the grid and grid-stride the rest (256 CUs x 8 blocks)
    unsigned long long blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 256ull * 32ull) blocks = 256ull * 32ull;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

// development switch: MIRO_TRACE_GRID_CAP = most workgroups a trace launch uses (threads stride over the rest)
inline int trace_grid_cap() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MIRO_TRACE_GRID_CAP");
        v = e ? atoi(e) : kTraceGridCap;
        if (v < 1) v = 1;
    }
    return v;
}

template <bool EXACT, bool ANY, bool STATS, int VAR>
mr_status launch_trace_t(const TraceParams &p, hipStream_t stream) {
    const size_t lds = (size_t)p.stack_depth * kTraceBlock * sizeof(int);
    if (lds > 160 * 1024) return fail(MR_ERR_INVALID, "traversal stack of depth %d does not fit in LDS", p.stack_depth);
    if (lds > 64 * 1024)
        MR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&trace_kernel<EXACT, ANY, STATS, VAR>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // Many short-lived workgroups balance better than a resident grid that strides over its rays: with exactly one
    // chip-full of workgroups (1792) the 33 M-ray frame runs 13 % slower than with the capped grid below, because the
    // hardware dispatcher rebalances at workgroup granularity while a static stride cannot.
    unsigned long long blocks = (p.n + kTraceBlock - 1) / kTraceBlock;
    const unsigned long long cap = (unsigned long long)trace_grid_cap();
    if (blocks > cap) blocks = cap;
    const unsigned grid = blocks ? (unsigned)blocks : 1u;
    hipLaunchKernelGGL((trace_kernel<EXACT, ANY, STATS, VAR>), dim3(grid), dim3(kTraceBlock), lds, stream, p);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

}  // namespace

// development switch: MIRO_TRACE_VARIANT selects the control-flow / slab-test variant of the exact kernel:
//   0..15  one launch-time ray per lane (bit 0: min/max slabs, bit 1: while-while, bit 2: lean fma slabs,
//          bit 3: wave-uniform nodes through the scalar cache)
//   16,17  persistent waves with ballot/prefix re-arming of idle lanes (refill threshold 1 / 16 lanes)
static int trace_variant() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MIRO_TRACE_VARIANT");
        v = e ? (atoi(e) & 31) : kDefaultVariant;
    }
    return v;
}

template <bool EXACT, bool ANY, bool QUOT, int REFILL_MIN>
static mr_status launch_persistent(const TraceParams &p, hipStream_t stream) {
    const size_t lds = (size_t)p.stack_depth * kTraceBlock * sizeof(int);
    if (lds > 160 * 1024) return fail(MR_ERR_INVALID, "traversal stack of depth %d does not fit in LDS", p.stack_depth);
    auto kern = &trace_persistent_kernel<EXACT, ANY, QUOT, REFILL_MIN>;
    if (lds > 64 * 1024)
        MR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int dev = 0, cus = 256, per_cu = 1;
    MR_HIP_CHECK(hipGetDevice(&dev));
    MR_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    MR_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), kTraceBlock, lds));
    if (per_cu < 1) per_cu = 1;
    unsigned long long want = (p.n + kTraceBlock - 1) / kTraceBlock;
    unsigned long long grid = (unsigned long long)cus * (unsigned)per_cu;
    if (want < grid) grid = want;
    MR_HIP_CHECK(hipMemsetAsync(p.work_counter, 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kTraceBlock), lds, stream, p, p.work_counter);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

template <bool ANY>
static mr_status launch_exact(const TraceParams &p, hipStream_t stream) {
    switch (trace_variant()) {
        case 0: return launch_trace_t<true, ANY, false, 0>(p, stream);
        case 1: return launch_trace_t<true, ANY, false, 1>(p, stream);
        case 2: return launch_trace_t<true, ANY, false, 2>(p, stream);
        case 3: return launch_trace_t<true, ANY, false, 3>(p, stream);
        case 7: return launch_trace_t<true, ANY, false, 7>(p, stream);
        case 9: return launch_trace_t<true, ANY, false, 9>(p, stream);
        case 16: return launch_persistent<true, ANY, false, 1>(p, stream);
        case 17: return launch_persistent<true, ANY, false, 16>(p, stream);
        default: return launch_trace_t<true, ANY, false, 11>(p, stream);
    }
}

mr_status launch_trace(const TraceParams &p, uint32_t flags, hipStream_t stream) {
    if (p.n == 0) return MR_OK;
    const bool fast = flags & MR_MATH_FAST, any = flags & MR_TRACE_ANY, stats = flags & MR_COUNT_STATS;
    const bool product = flags & MR_MATH_PRODUCT;
    if (p.n_planes || p.n_spheres) {
        // scenes with spheres / planes: the exact kernels with the object dispatch compiled in (VAR bit 5); the fast
        // and persistent forms cover triangle scenes only
        if (stats) return any ? launch_trace_t<true, true, true, 32>(p, stream) : launch_trace_t<true, false, true, 32>(p, stream);
        if (product) return any ? launch_trace_t<true, true, false, 43>(p, stream) : launch_trace_t<true, false, false, 43>(p, stream);
        return any ? launch_trace_t<true, true, false, 58>(p, stream) : launch_trace_t<true, false, false, 58>(p, stream);
    }
    if (stats) {
        // counting mode is diagnostic: always the literal-division kernel in the reference's control flow
        return any ? launch_trace_t<true, true, true, 0>(p, stream) : launch_trace_t<true, false, true, 0>(p, stream);
    }
    // MR_MATH_FAST: lean fma slabs + fmaf/rcp triangle test, with the same scalar-cache path as the exact kernels (VAR 15)
    if (fast) return any ? launch_trace_t<false, true, false, 15>(p, stream) : launch_trace_t<false, false, false, 15>(p, stream);
    if (flags & MR_TRACE_PERSISTENT) {
        if (product) return any ? launch_persistent<true, true, false, 16>(p, stream) : launch_persistent<true, false, false, 16>(p, stream);
        return any ? launch_persistent<true, true, true, 16>(p, stream) : launch_persistent<true, false, true, 16>(p, stream);
    }
    // MR_MATH_PRODUCT: slab distances as products with the rounded 1/d (and its development variants)
    if (product) return any ? launch_exact<true>(p, stream) : launch_exact<false>(p, stream);
    // default: the reference's quotients by the correction step, while-while, scalar path (VAR 16 | 2 | 8)
    return any ? launch_trace_t<true, true, false, 26>(p, stream) : launch_trace_t<true, false, false, 26>(p, stream);
}

mr_status launch_eye_rays(const mr_camera &cam, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1,
                          uint32_t spp, uint32_t jitter, uint32_t seed, bool tiled, mr_ray *d_rays, hipStream_t stream) {
    // camera frame on the host, in the reference's order of operations (Camera.h:79-110, Camera.cpp:113-124)
    auto unit3 = [](float *a) {
        const float len = sqrtf((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
        const float inv = 1.0f / len;
        a[0] *= inv; a[1] *= inv; a[2] *= inv;
    };
    auto cross3 = [](const float *a, const float *b, float *o) {
        o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
    };
    EyeFrame f;
    float up[3] = {cam.up[0], cam.up[1], cam.up[2]};
    unit3(up);
    float view[3] = {cam.lookat[0] - cam.eye[0], cam.lookat[1] - cam.eye[1], cam.lookat[2] - cam.eye[2]};
    unit3(view);
    f.w[0] = -view[0]; f.w[1] = -view[1]; f.w[2] = -view[2];
    unit3(f.w);
    cross3(up, f.w, f.u);
    unit3(f.u);
    cross3(f.w, f.u, f.v);
    const float PI = 3.1415926535897932384626433832795028841972f;
    const float DegToRad = PI / 180.0f, HalfDegToRad = DegToRad / 2.0f;
    const float aspect = (float)W / (float)H;
    f.top = tanf(cam.fov_deg * HalfDegToRad);
    f.right = aspect * f.top; f.bottom = -f.top; f.left = -f.right;
    f.eye[0] = cam.eye[0]; f.eye[1] = cam.eye[1]; f.eye[2] = cam.eye[2];
    f.W = W; f.H = H; f.y0 = y0; f.spp = spp; f.jitter = jitter;
    f.tile = tile_shape(spp);
    f.rows = y1 - y0;
    f.tiled = tiled && (f.tile.th > 1 || f.tile.tw > 1) ? 1u : 0u;
    {   // host copy of pcg_hash
        uint32_t state = seed * 747796405u + 2891336453u;
        uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
        f.hbase = (word >> 22u) ^ word;
    }
    f.n = (unsigned long long)(y1 - y0) * W * spp;
    if (f.n == 0) return MR_OK;
    hipLaunchKernelGGL(eye_rays_kernel, dim3(grid_for(f.n)), dim3(kBlock), 0, stream, f, d_rays);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

mr_status launch_shadow_rays(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, unsigned long long n,
                             const float light[3], mr_ray *d_out, uint32_t *d_src, unsigned long long *d_count,
                             hipStream_t stream) {
    MR_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), stream));
    if (n == 0) return MR_OK;
    if ((ds.spheres || ds.planes) && !d_rays)
        return fail(MR_ERR_INVALID, "the scene holds spheres / planes: their hit point is o + t*d, d_rays is required");
    hipLaunchKernelGGL(shadow_rays_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, surface_ptrs(ds), d_rays, d_hits, n,
                       light[0], light[1], light[2], d_out, d_src, d_count);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

mr_status launch_hit_attrs(const DeviceScene &ds, const mr_ray *d_rays, const mr_hit *d_hits, unsigned long long n,
                           float *d_P, float *d_N, hipStream_t stream) {
    if (n == 0) return MR_OK;
    if ((ds.spheres || ds.planes) && !d_rays)
        return fail(MR_ERR_INVALID, "the scene holds spheres / planes: their hit point is o + t*d, d_rays is required");
    hipLaunchKernelGGL(hit_attrs_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, surface_ptrs(ds), d_rays, d_hits, n, d_P, d_N);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

}  // namespace mr
