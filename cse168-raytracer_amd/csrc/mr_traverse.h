// mr_traverse.h -- device-side machinery of the intersection path, shared by the batched trace kernels
// (mr_kernels.hip) and the fused frame kernel (mr_frame.hip): slab tests, Triangle::intersect, Sphere::intersect,
// the per-lane LDS stack and the "while-while" traversal loop.  Device code only; include from .hip files.
//
//   BVH::intersect / intersectChildren   BVH.cpp:438-658 (scalar branch)
//   Triangle::intersect                  Triangle.cpp:136-169
//   Sphere::intersect                    Sphere.cpp:28-69
//
// Compiled with -ffp-contract=off: in the default ("exact") mode every fp32 operation below is one individually
// rounded IEEE op in the reference's order, so t / beta / gamma are bit-identical to the reference's scalar build.
#pragma once

#include <hip/hip_runtime.h>

#include "mr_internal.h"

namespace mr {
namespace {

constexpr int kBlock = 256;          // 4 waves per workgroup
#ifndef MIRO_TRACE_BLOCK
#define MIRO_TRACE_BLOCK 256
#endif
constexpr int kTraceBlock = MIRO_TRACE_BLOCK;   // threads per workgroup of the trace kernels (their LDS stack is [depth][kTraceBlock])
#ifndef MIRO_GRID_CAP
#define MIRO_GRID_CAP 32768
#endif
constexpr int kTraceGridCap = MIRO_GRID_CAP; // workgroups per trace launch (see launch_trace_t)
// the fused frame kernel on frames of 2^18 chunks or more (1080p at 64 spp is 518 400): twice the workgroups, and XCD runs of 256
// chunks instead of 64 -- +1.1 % there, while frames of 4 to 16 samples per pixel lose 1.5-4 % to either (profiles/r03_grid_ab.log)
#ifndef MIRO_CAP_LARGE
#define MIRO_CAP_LARGE 65536
#define MIRO_RUN_LARGE 256
#endif
constexpr int kFrameGridCapLarge = MIRO_CAP_LARGE;
constexpr unsigned long long kFrameLargeChunks = 1ull << 18;

// Workgroup ids go round-robin to the 8 XCDs, each with its own L2 (not coherent with the others').  In the plain order
// every XCD traces every eighth 256-ray chunk of the whole image; here an XCD gets runs of kXcdRun consecutive chunks --
// one region of the image, whole cache lines of the output to itself: +4.9 % on the bench frame, +3 % at 16 and 4 spp
// (profiles/r02_xcd_runs.log; DESIGN.md section 4.14 for what the counters show).  Only for grids of kXcdMinGrid workgroups or more: a 1-spp frame is 8 100 workgroups,
// little more than four per resident slot, and there the uneven cost of the regions shows as idle XCDs (-5 %).  Round 3: launches that
// reach kFrameGridCapLarge workgroups (the fused frame kernel at 64 samples per pixel) take runs of kXcdRunLarge chunks.
#ifndef MIRO_XCD_MIN_GRID
#define MIRO_XCD_MIN_GRID 16384
#endif
constexpr unsigned kXcdRun = 64, kXcdRunLarge = MIRO_RUN_LARGE, kXcdMinGrid = MIRO_XCD_MIN_GRID;
__device__ __forceinline__ unsigned xcd_block_id_of(unsigned b, unsigned G, bool large, unsigned min_grid = kXcdMinGrid) {
    if (G < min_grid) return b;
    const unsigned run = large ? kXcdRunLarge : kXcdRun;
    const unsigned xcd = b & 7u, slot = b >> 3, grp = slot / run, k = slot - grp * run;
    return (grp + 1u) * (8u * run) <= G ? grp * (8u * run) + xcd * run + k : b;      // the ragged tail keeps the plain order
}
__device__ __forceinline__ unsigned xcd_block_id() {
    return xcd_block_id_of(blockIdx.x, gridDim.x, gridDim.x >= (unsigned)kFrameGridCapLarge);   // only the large-frame launch has that many workgroups
}
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

// |a - b| of two bit patterns as unsigned integers, and a three-way unsigned minimum: one VALU op each
__device__ __forceinline__ unsigned vsad(float a, float b) {
    unsigned o;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(o) : "v"(a), "v"(b));
    return o;
}
__device__ __forceinline__ unsigned vmin3u(unsigned a, unsigned b, unsigned c) {
    unsigned o;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
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

// Octant-specialised forms (OCT bit k set: the ray's direction component k is negative; the octant is uniform over the
// wave).  Rounding is monotone -- RN(a * inv), fma(c, inv, n) and the correctly rounded quotient all grow with the corner
// when 1/d > 0 and shrink when 1/d < 0 -- and a box has lo <= hi on every axis (boxes that do not are "irregular" and take
// another path), so the smaller of an axis' two slab distances is the one of the lo corner for d > 0 and of the hi corner
// for d < 0: the six per-box min/max of the generic forms become a compile-time choice of operand.  Same entry / exit
// values (up to the sign of a zero), 12 VALU fewer per two-child visit.
template <int OCT, int SLAB>
__device__ __forceinline__ void slab_box_oct(float lox, float hix, float loy, float hiy, float loz, float hiz,
                                             const RayRegs &r, float &mn, float &mx) {
    const float nx = (OCT & 1) ? hix : lox, fx = (OCT & 1) ? lox : hix;
    const float ny = (OCT & 2) ? hiy : loy, fy = (OCT & 2) ? loy : hiy;
    const float nz = (OCT & 4) ? hiz : loz, fz = (OCT & 4) ? loz : hiz;
    if (SLAB == 4) {
        mn = vmax3(exact_quot(nx - r.ox, r.dx, r.ix), exact_quot(ny - r.oy, r.dy, r.iy), exact_quot(nz - r.oz, r.dz, r.iz));
        mx = vmin3(exact_quot(fx - r.ox, r.dx, r.ix), exact_quot(fy - r.oy, r.dy, r.iy), exact_quot(fz - r.oz, r.dz, r.iz));
    } else if (SLAB == 2) {
        mn = vmax3(fmaf(nx, r.ix, r.nox), fmaf(ny, r.iy, r.noy), fmaf(nz, r.iz, r.noz));
        mx = vmin3(fmaf(fx, r.ix, r.nox), fmaf(fy, r.iy, r.noy), fmaf(fz, r.iz, r.noz));
    } else {
        mn = vmax3((nx - r.ox) * r.ix, (ny - r.oy) * r.iy, (nz - r.oz) * r.iz);
        mx = vmin3((fx - r.ox) * r.ix, (fy - r.oy) * r.iy, (fz - r.oz) * r.iz);
    }
}
__device__ __forceinline__ int octant_of(const RayRegs &r) {
    return (r.dx < 0.0f ? 1 : 0) | (r.dy < 0.0f ? 2 : 0) | (r.dz < 0.0f ? 4 : 0);
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
        // (Leaving with "rejected" as soon as t alone rejects the triangle in every active lane -- skipping the two other
        // divisions and both cross products -- was measured: 16.21 vs 16.32 Grays/s on the bench frame, 7.56 vs 7.65 at 1 spp:
        // the wave-wide test costs more than the rare whole-wave rejection saves, profiles/r03_t_first_ab.log.)
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
    int lpos, lend;      // voting traversal only: next / one-past-last triangle of the leaf in progress (lpos == lend: not begun)
    __device__ __forceinline__ bool have() const { return cur != kDone; }
};

// L.sp is the lane's stack pointer as an ABSOLUTE LDS address (the base of the dynamic LDS block is added once, in
// stack_reset): a push or pop is one ds instruction on that register -- through a generic `s_stack + offset` the compiler
// emitted a v_add with the (link-time) base in front of every one of them.
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ void stack_push(Lane &L, int *s_stack, int v) {
    *reinterpret_cast<lds_int *>((unsigned)L.sp) = v;
    L.sp += kStackStride;
}
__device__ __forceinline__ int stack_pop(Lane &L, int *s_stack) {
    L.sp -= kStackStride;
    return *reinterpret_cast<lds_int *>((unsigned)L.sp);
}
__device__ __forceinline__ void stack_reset(Lane &L, int *s_stack, int tid) {
    L.sp = (int)(unsigned)reinterpret_cast<__UINTPTR_TYPE__>((lds_int *)s_stack) + tid * (int)sizeof(int);
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

template <bool EXACT, bool STATS, int SLAB, int OCT = 8>
__device__ __forceinline__ void node_slabs(const float4 q0, const float4 q1, const float4 q2, const RayRegs &r,
                                           float &mn0, float &mx0, float &mn1, float &mx1);
template <bool EXACT, bool STATS, int SLAB, int OCT = 8>
__device__ __forceinline__ void node_slabs_guarded(const float4 q0, const float4 q1, const float4 q2, const RayRegs &r, float best_t,
                                                   float &mn0, float &mx0, float &mn1, float &mx1, float &k0, float &k1);

// the post-test bookkeeping of BVH.cpp:609-651: near child first (ties -> child 0), far child pushed, else pop
// HAVE_K: the caller already holds k0 = minNum(mx0, best_t), k1 = minNum(mx1, best_t) (node_slabs_guarded computes them for
// its tie test: recomputing them here cost two VALU instructions per visit of the hot loop)
template <bool STATS, bool SAFE, bool HAVE_K = false>
__device__ __forceinline__ void node_decide(float mn0, float mx0, float mn1, float mx1, int ref0, int ref1,
                                            const RayRegs &r, Lane &L, int *s_stack, Stats &st, float k0 = 0.0f, float k1 = 0.0f) {
    // tMax of this call == best_t: nothing changed since the node was entered
    bool h0, h1;
    if (SAFE) {
        // mn, mx are not NaN here; (mn > mx || mn > best) == (mn > minNum(mx, best)) also when best is NaN
        if (!HAVE_K) { k0 = vmin2(mx0, L.best_t); k1 = vmin2(mx1, L.best_t); }
        h0 = !((mn0 > k0) || (mx0 < r.tmin));
        h1 = !((mn1 > k1) || (mx1 < r.tmin));
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
template <bool EXACT, bool STATS, int SLAB, bool SCALAR = false, int OCT = 8>
__device__ __forceinline__ void node_step(const TraceParams &p, const RayRegs &r, Lane &L, int *s_stack, int tid, Stats &st) {
    float mn0, mx0, mn1, mx1;
    constexpr bool kSafe = SLAB == 1 || SLAB == 2 || SLAB == 4 || SLAB == 5 || SLAB == 6;
    constexpr bool kGuarded = SLAB == 5 || SLAB == 6;
    if (SCALAR) {
        const int cur0 = __builtin_amdgcn_readfirstlane(L.cur);
        if (__all(L.cur == cur0)) {
            const v16f v = load_node_scalar(p.nodes, cur0);
            const float4 q0 = make_float4(v[0], v[1], v[2], v[3]), q1 = make_float4(v[4], v[5], v[6], v[7]);
            const float4 q2 = make_float4(v[8], v[9], v[10], v[11]);
            if ((SLAB == 4 || kGuarded) && __float_as_int(v[14]) != 0) {   // irregular node (wave-uniform): the reference's own divisions
                node_slabs<EXACT, STATS, 3>(q0, q1, q2, r, mn0, mx0, mn1, mx1);
                node_decide<STATS, false>(mn0, mx0, mn1, mx1, __float_as_int(v[12]), __float_as_int(v[13]), r, L, s_stack, st);
                return;
            }
            float k0 = 0.0f, k1 = 0.0f;
            node_slabs_guarded<EXACT, STATS, SLAB, OCT>(q0, q1, q2, r, L.best_t, mn0, mx0, mn1, mx1, k0, k1);
            node_decide<STATS, kSafe, kGuarded>(mn0, mx0, mn1, mx1, __float_as_int(v[12]), __float_as_int(v[13]), r, L, s_stack, st, k0, k1);
            return;
        }
    }
    // ---- inner node: test both children (BVH.cpp:593-624)
    const float4 *nd = p.nodes + 4 * (size_t)L.cur;
    const float4 q0 = nd[0], q1 = nd[1], q2 = nd[2];
    const int4 q3 = *reinterpret_cast<const int4 *>(nd + 3);
    if ((SLAB == 4 || kGuarded) && q3.z != 0) {
        node_slabs<EXACT, STATS, 3>(q0, q1, q2, r, mn0, mx0, mn1, mx1);
        node_decide<STATS, false>(mn0, mx0, mn1, mx1, q3.x, q3.y, r, L, s_stack, st);
        return;
    }
    float k0 = 0.0f, k1 = 0.0f;
    node_slabs_guarded<EXACT, STATS, SLAB, OCT>(q0, q1, q2, r, L.best_t, mn0, mx0, mn1, mx1, k0, k1);
    node_decide<STATS, kSafe, kGuarded>(mn0, mx0, mn1, mx1, q3.x, q3.y, r, L, s_stack, st, k0, k1);
}

template <bool EXACT, bool STATS, int SLAB, int OCT>
__device__ __forceinline__ void node_slabs(const float4 q0, const float4 q1, const float4 q2, const RayRegs &r,
                                           float &mn0, float &mx0, float &mn1, float &mx1) {
    if (OCT < 8 && (SLAB == 1 || SLAB == 2 || SLAB == 4)) {
        slab_box_oct<OCT, SLAB>(q0.x, q0.y, q0.z, q0.w, q2.x, q2.y, r, mn0, mx0);
        slab_box_oct<OCT, SLAB>(q1.x, q1.y, q1.z, q1.w, q2.z, q2.w, r, mn1, mx1);
    } else if (SLAB == 2) {
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

// SLAB 5, "guarded products": the slab distances of a regular ray in a regular node as products (corner - o) * RN(1/d)
// -- 24 VALU fewer per two-child visit than the correction steps of SLAB 4 -- whenever the decisions taken from them are
// PROVABLY the ones the reference's quotients give, and the quotients themselves otherwise.
//   * q~ = RN(a * RN(1/d)) = (a/d)(1+e), |e| <= 2^-23 + 2^-48, against q = RN(a/d) = (a/d)(1+e'), |e'| <= 2^-24 (no
//     under- or overflow: lane_is_regular, regular nodes): q~ and q are less than 4 ulps apart, have the same sign, and
//     are zero together.  max3 / min3 / min with best_t are monotone, so each of mn0, mx0, mn1, mx1, min(mx, best_t)
//     computed from products is less than 4 ulps from the same expression on quotients.
//   * node_decide takes five comparisons from them: mn0 > min(mx0, best), mx0 < tmin, the same two for child 1, and
//     mn0 > mn1.  Two floats further than 8 ulps apart compare the same way after each moves by less than 4.  The ulp
//     distance of two floats of one sign is the difference of their bit patterns (v_sad_u32); patterns of opposite sign
//     are 2^31 apart, and there the comparison is decided by the signs, which are exact.
//   * so: if in every lane all five pairs are more than 16 patterns apart (margin of two), the product decisions stand;
//     if any lane has a closer pair the whole wave recomputes the node with exact quotients (SLAB 4) -- about one visit
//     in a few thousand.
// Same hits, same visiting order, same bits as SLAB 4 (tests: test_gpu_parity, the fuzz campaigns run both).
template <bool EXACT, bool STATS, int SLAB, int OCT>
__device__ __forceinline__ void node_slabs_guarded(const float4 q0, const float4 q1, const float4 q2, const RayRegs &r, float best_t,
                                                   float &mn0, float &mx0, float &mn1, float &mx1, float &k0, float &k1) {
    if (SLAB != 5 && SLAB != 6) {
        node_slabs<EXACT, STATS, SLAB, OCT>(q0, q1, q2, r, mn0, mx0, mn1, mx1);
        return;
    }
    node_slabs<EXACT, STATS, 1, OCT>(q0, q1, q2, r, mn0, mx0, mn1, mx1);
    // k = minNum(exit, best_t) of both children and the smallest of the pattern distances, in ONE asm block (separate
    // asm statements are fenced by hazard no-ops: three s_nop per visit when the two v_min stood alone)
    unsigned near, t0, t1;
    if (SLAB == 6) {
        // tMin == 0 in every lane of the wave (camera, shadow and bounce rays: all of them): "exit < tMin" is a SIGN test,
        // and a product and its quotient have the same sign and are zero together -- that comparison needs no guard.
        // Three pairs are left: entry against min(exit, best) for each child, entry against entry.
        asm("v_min_f32 %3, %7, %9\n\tv_min_f32 %4, %8, %9\n\t"
            "v_sad_u32 %0, %5, %3, 0\n\tv_sad_u32 %1, %6, %4, 0\n\tv_sad_u32 %2, %5, %6, 0\n\tv_min3_u32 %0, %0, %1, %2"
            : "=&v"(near), "=&v"(t0), "=&v"(t1), "=&v"(k0), "=&v"(k1)
            : "v"(mn0), "v"(mn1), "v"(mx0), "v"(mx1), "v"(best_t));
    } else {
        asm("v_min_f32 %3, %8, %10\n\tv_min_f32 %4, %9, %10\n\t"
            "v_sad_u32 %0, %5, %3, 0\n\tv_sad_u32 %1, %8, %6, 0\n\tv_sad_u32 %2, %7, %4, 0\n\tv_min3_u32 %0, %0, %1, %2\n\t"
            "v_sad_u32 %1, %9, %6, 0\n\tv_sad_u32 %2, %5, %7, 0\n\tv_min3_u32 %0, %0, %1, %2"
            : "=&v"(near), "=&v"(t0), "=&v"(t1), "=&v"(k0), "=&v"(k1)
            : "v"(mn0), "v"(r.tmin), "v"(mn1), "v"(mx0), "v"(mx1), "v"(best_t));
    }
    if (__any(near <= 16u)) {
        node_slabs<EXACT, STATS, 4, OCT>(q0, q1, q2, r, mn0, mx0, mn1, mx1);
        k0 = vmin2(mx0, best_t); k1 = vmin2(mx1, best_t);
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

// One triangle of the lane's current leaf (voting traversal): the leaf is decoded on its first step, popped after its
// last.  Same tests in the same order with the same running best_t as leaf_step.  When every participating lane is at
// the start of the same leaf, the whole leaf goes through the scalar cache in this one step (the coherent case).
template <bool EXACT, bool ANY, bool STATS, bool SCALAR, bool OBJ>
__device__ __forceinline__ void tri_step(const TraceParams &p, const RayRegs &r, Lane &L, int *s_stack, Stats &st) {
    bool done = false;
    if (SCALAR) {
        const int cur0 = __builtin_amdgcn_readfirstlane(L.cur);
        if (__all(L.cur == cur0 && L.lpos == L.lend)) {
            const unsigned bits0 = ~(unsigned)cur0;
            const unsigned first0 = bits0 >> kLeafCountBits;
            unsigned cnt0 = bits0 & kLeafCountMask;
            if (cnt0 == kLeafCountMask) cnt0 = p.leaf_cnt_ext[first0];
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
            if (STATS) {
                if (OBJ) { for (unsigned k = 0; k < cnt0; k++) if (__float_as_uint(p.tris[3 * (size_t)(first0 + k) + 2].w) != kSphereTag) st.tri++; }
                else st.tri += cnt0;
            }
            if (ANY && done) { L.cur = kDone; }
            else { L.cur = stack_pop(L, s_stack); if (STATS && L.cur != kDone) st.box++; }
            return;
        }
    }
    if (L.lpos == L.lend) {                          // first step in this leaf
        const unsigned bits = ~(unsigned)L.cur;
        const unsigned first = bits >> kLeafCountBits;
        unsigned cnt = bits & kLeafCountMask;
        if (cnt == kLeafCountMask) cnt = p.leaf_cnt_ext[first];
        L.lpos = (int)first; L.lend = (int)(first + cnt);
        if (STATS) {
            if (OBJ) { for (unsigned k = 0; k < cnt; k++) if (__float_as_uint(p.tris[3 * (size_t)(first + k) + 2].w) != kSphereTag) st.tri++; }
            else st.tri += cnt;
        }
    }
    if (L.lpos < L.lend) {
        const float4 *tr = p.tris + 3 * (size_t)L.lpos;
        float t, b, g;
        const bool ok = object_test<EXACT, OBJ>(tr[0], tr[1], tr[2], r, L.best_t, t, b, g);
        if (ok && t < L.best_t) {                    // strict-less replacement (BVH.cpp:500)
            L.best_t = t; L.best_b = b; L.best_g = g; L.best_pos = L.lpos;
            if (ANY) done = true;
        }
        L.lpos++;
    }
    if (ANY && done) {
        L.cur = kDone; L.lpos = L.lend;
    } else if (L.lpos == L.lend) {                   // leaf finished (or empty): on to the next pending node
        L.cur = stack_pop(L, s_stack);
        if (STATS && L.cur != kDone) st.box++;
    }
}

// MODE 0: one step of whatever the lane needs per iteration (the reference's control flow, lane by lane)
// MODE 1: "while-while": lanes run inner nodes until each holds a leaf (or is done), then the wave does the leaves
// MODE 2: voting: every iteration the wave counts the lanes that need a node step and those that need a triangle test
//         and runs the step the majority needs; the others wait one round.  In while-while a wave's node loop lasts as
//         long as its slowest lane's search for a leaf (incoherent batches: 14 of 64 lanes active per VALU
//         instruction, profiles/r02_before_random); with the vote at least half of the unfinished lanes are active in every step.  The order of
//         every lane's own steps -- and so its hit record -- is the same in all three modes.
template <bool EXACT, bool ANY, bool STATS, int SLAB, int MODE, bool SCALAR, bool OBJ = false, int OCT = 8>
__device__ __forceinline__ void traverse(const TraceParams &p, const RayRegs &r, Lane &L, int *s_stack, int tid, Stats &st) {
    if (MODE == 2) {
        L.lpos = 0; L.lend = 0;
        while (true) {
            const bool want_node = L.cur >= 0, want_tri = L.cur < 0 && L.cur != kDone;
            const unsigned long long m_node = __ballot(want_node), m_tri = __ballot(want_tri);
            if ((m_node | m_tri) == 0ull) break;
            if (__popcll(m_node) >= __popcll(m_tri)) {
                if (want_node) node_step<EXACT, STATS, SLAB, SCALAR, OCT>(p, r, L, s_stack, tid, st);
            } else {
                if (want_tri) tri_step<EXACT, ANY, STATS, SCALAR, OBJ>(p, r, L, s_stack, st);
            }
        }
    } else if (MODE == 1) {
        while (__any(L.have())) {
            while (L.cur >= 0) node_step<EXACT, STATS, SLAB, SCALAR, OCT>(p, r, L, s_stack, tid, st);
            if (L.have()) leaf_step<EXACT, ANY, STATS, SCALAR, OBJ>(p, r, L, s_stack, tid, st);
        }
    } else {
        while (L.have()) {
            if (L.cur >= 0) node_step<EXACT, STATS, SLAB, SCALAR>(p, r, L, s_stack, tid, st);
            else leaf_step<EXACT, ANY, STATS, false, OBJ>(p, r, L, s_stack, tid, st);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// One ray per lane from the root test to the unbounded-object scan: Scene::trace (Scene.cpp:214-230) ->
// BVH::intersect (BVH.cpp:438-469) -> intersectChildren.  `live` = the lane holds a ray; the call is made by whole
// waves (the while-while loop votes with __any).  Result in L (best_t / best_pos / beta / gamma) and plane_hit.
// VAR bit 0: min/max slabs on the products (corner - o) * (1/d) for waves that cannot produce a NaN;
//     bit 1: while-while control flow (bit 6: the voting control flow instead); bit 2: lean fma slabs (MR_MATH_FAST); bit 3: wave-uniform nodes and leaves
//     through the scalar cache; bit 4: every slab distance is the reference's true quotient (the default trace;
//     MR_COUNT_STATS implies it); bit 5: the scene holds spheres and / or planes; bit 8: octant-specialised slab tests
//     for waves whose rays share an octant.
// ---------------------------------------------------------------------------------------------------
template <bool EXACT, bool ANY, bool STATS, int VAR>
__device__ __forceinline__ void trace_ray(const TraceParams &p, const RayRegs &r, float tmax0, bool live, Lane &L,
                                          int &plane_hit, int *s_stack, int tid, Stats &st) {
    constexpr bool kStrict = STATS || (VAR & 16);
    constexpr int kBaseSlab = kStrict ? 3 : 0;
    constexpr bool kMinMax = !kStrict && (VAR & 1);
    constexpr int kWW = (VAR & 64) ? 2 : ((VAR & 2) ? 1 : 0);   // control flow of traverse()
    constexpr int kSafeSlab = (VAR & 4) ? 2 : 1;      // slab form for waves whose rays cannot produce a NaN
    constexpr bool kScalar = (VAR & 8) != 0;          // wave-uniform nodes through the scalar cache
    constexpr bool kObj = (VAR & 32) != 0;            // the scene holds spheres and / or planes

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

    // VAR bit 8: when the wave's live rays all point into one octant the slab tests take their near / far corners by
    // position (slab_box_oct): eight copies of the loop, chosen once per ray batch of the wave
    constexpr bool kOct = (VAR & 256) != 0;
    constexpr int kExactSlab = (VAR & 512) ? 5 : 4;   // VAR bit 9: guarded products (node_slabs_guarded) instead of the correction steps
    constexpr int kGoodSlab = kMinMax ? kSafeSlab : kExactSlab;
    constexpr int kOctSlab = kGoodSlab == 5 ? 6 : kGoodSlab;     // the octant loops' guard assumes tMin == 0 (node_slabs_guarded)
    const bool good_wave = kMinMax ? __all(lane_is_nan_free(r) || !live) : ((kStrict && !STATS) ? __all(lane_is_regular(r) || !live) : false);
    bool done_oct = false;
    if (kOct && (kMinMax || (kStrict && !STATS)) && good_wave) {
        const unsigned long long m_live = __ballot(live);
        if (m_live) {
            const int oct = octant_of(r);
            const int oct0 = __builtin_amdgcn_readlane(oct, __ffsll((long long)m_live) - 1);
            if (__all(!live || oct == oct0) && (kOctSlab != 6 || __all(!live || r.tmin == 0.0f))) {
                done_oct = true;
                switch (oct0) {
                    case 0: traverse<EXACT, ANY, STATS, kOctSlab, kWW, kScalar, kObj, 0>(p, r, L, s_stack, tid, st); break;
                    case 1: traverse<EXACT, ANY, STATS, kOctSlab, kWW, kScalar, kObj, 1>(p, r, L, s_stack, tid, st); break;
                    case 2: traverse<EXACT, ANY, STATS, kOctSlab, kWW, kScalar, kObj, 2>(p, r, L, s_stack, tid, st); break;
                    case 3: traverse<EXACT, ANY, STATS, kOctSlab, kWW, kScalar, kObj, 3>(p, r, L, s_stack, tid, st); break;
                    case 4: traverse<EXACT, ANY, STATS, kOctSlab, kWW, kScalar, kObj, 4>(p, r, L, s_stack, tid, st); break;
                    case 5: traverse<EXACT, ANY, STATS, kOctSlab, kWW, kScalar, kObj, 5>(p, r, L, s_stack, tid, st); break;
                    case 6: traverse<EXACT, ANY, STATS, kOctSlab, kWW, kScalar, kObj, 6>(p, r, L, s_stack, tid, st); break;
                    default: traverse<EXACT, ANY, STATS, kOctSlab, kWW, kScalar, kObj, 7>(p, r, L, s_stack, tid, st); break;
                }
            }
        }
    }
    if (done_oct) {
    } else if (kMinMax) {
        // a slab product (corner - o) * (1/d) can only be NaN as 0*inf or inf*0 or from a non-finite origin:
        // with o, d and 1/d all finite in every lane the select form and the min/max form decide identically
        if (good_wave) traverse<EXACT, ANY, STATS, kSafeSlab, kWW, kScalar, kObj>(p, r, L, s_stack, tid, st);
        else traverse<EXACT, ANY, STATS, 0, kWW, false, kObj>(p, r, L, s_stack, tid, st);
    } else if (kStrict && !STATS) {
        // the default trace: where every lane's ray is regular, the quotients' decisions from guarded products (VAR bit 9)
        // or from the correction steps (then the lanes' quotients are NaN-free too and the min/max form decides like the
        // select form); the reference's own divisions otherwise
        // (a wave whose rays point into several octants is an incoherent one: bound by its record fetches, it gains nothing
        // from the guarded products and would pay for their wave-wide branch -- the correction steps alone here)
        // VAR bit 10: ... and the voting control flow suits it better (random rays 3.86 -> 4.09 Grays/s, the atrium's bounce
        // rays 4.96 -> 5.31, 1-spp shadow rays in image order 3.73 -> 4.31 without the caller's MR_TRACE_INCOHERENT hint)
        constexpr int kMixedFlow = ((VAR & 1024) && kWW == 1) ? 2 : kWW;
        if (good_wave) traverse<EXACT, ANY, STATS, 4, kMixedFlow, kScalar, kObj>(p, r, L, s_stack, tid, st);
        else traverse<EXACT, ANY, STATS, 3, kWW, kScalar, kObj>(p, r, L, s_stack, tid, st);
    } else {
        traverse<EXACT, ANY, STATS, kBaseSlab, kWW, kStrict && kScalar, kObj>(p, r, L, s_stack, tid, st);
    }

    // Scene::trace's scan of the unbounded objects (Scene.cpp:220-230): every plane is tested against the
    // caller's tMin / tMax (Plane.cpp:33-48) and kept when nothing was hit yet or it is strictly nearer
    plane_hit = -1;
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
}

// the mr_hit record of a finished lane (HitInfo in its device form, miro_hip.h)
template <bool OBJ>
__device__ __forceinline__ mr_hit make_hit(const TraceParams &p, const Lane &L, int plane_hit, float tmax0) {
    mr_hit h;
    if (OBJ && plane_hit >= 0) {
        h.t = L.best_t; h.prim = kPlaneBit | (uint32_t)plane_hit; h.beta = 0.0f; h.gamma = 0.0f;
    } else if (L.best_pos >= 0) {
        h.t = L.best_t; h.prim = p.tri_prim[L.best_pos]; h.beta = L.best_b; h.gamma = L.best_g;
    } else {
        h.t = tmax0; h.prim = MR_MISS; h.beta = 0.0f; h.gamma = 0.0f;
    }
    return h;
}


}  // namespace
}  // namespace mr
