// mr_photon.hip -- Photon_map::irradiance_estimate / locate_photons (PhotonMap.cpp:81-243) on gfx950:
// one wave64 per query, the k-nearest candidate set in LDS.
//
// The reference walks the implicit kd-tree one node at a time with a 500-entry max-heap per query.  What it returns is a SET --
// the k nearest facing photons inside the radius (minus one, see below) -- and a set can be found in any order.  Here a wave
// works on BLOCKS of the heap-ordered tree: the 63 nodes of six consecutive levels below a block root b are b*2^l + o
// (contiguous per level, so the loads coalesce), one node per lane, and every block carries two bounding boxes made at
// balance time (of its own 63 photons, and of everything below its root).  A search is two interleaved phases: EXPAND hands
// each lane one of a pending block's 64 child blocks and measures the query's distance to the child's boxes -- children
// whose own box reaches into the radius are listed, children whose subtree box does are expanded in turn --, EXAMINE tests
// the 63 photons of a listed block (distance, facing) and appends accepted ones to the LDS candidate buffer by ballot +
// prefix sum, four listed blocks per step with all their loads in flight together.  When the buffer is nearly full the wave
// selects the k smallest distances (4-pass radix select on the float bits, LDS histogram) and the k-th becomes the new
// radius.  (Rounds 1-2 walked the kd-tree's planes block by block in the reference's order, with the lanes of a block
// pulling their ancestors' decisions through shuffles: a serial chain of ~2 300 cycles per block and 93 blocks per query
// where the boxes need 61 independent ones; 41.5 -> 24 ms per 2.07 M queries, results identical.)
//
// Kept from the reference: nodes with index >= stored/2-1 do not descend (the photons below them are never found); the normalising radius stays
// max_dist^2 unless more than k candidates were seen; strict comparisons; and the FIRST-OVERFLOW REPLACEMENT
// (PhotonMap.cpp:195-240): when candidate k+1 arrives the reference heapifies the k it holds and replaces the heap
// root -- the farthest of the first k, m* -- by the newcomer even when the newcomer is farther still, and only from
// then on does the radius follow the heap root.  m* therefore never takes part in the result: what the reference
// returns is the k nearest of (all candidates minus m*).  Which photon m* is depends on the reference's visiting
// order (near side first, a node after its subtrees), so a first pass walks exactly that order, one query per lane,
// until k+1 candidates have been seen (first_overflow below); the wave-cooperative search then skips m*.
// found and the radius are the reference's on every query.  Not kept: the order in which the powers are summed
// (float rounding, ~1e-6 relative), and which of two photons at exactly the same squared distance is dropped when they
// tie for m* or at the k-th place (the reference's heap layout decides; distances, found and radius are unaffected).
#include <hip/hip_runtime.h>

#include "mr_internal.h"

namespace mr {
namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kCap = 704;          // candidate slots per wave: kTighten + one block of 63 candidates + 1
constexpr int kTighten = 640;      // compress (tighten the radius) once this many candidates are buffered
#ifndef MIRO_PHOTON_GUESS_MARGIN
#define MIRO_PHOTON_GUESS_MARGIN 1.02f
#endif
constexpr float kGuessMargin = MIRO_PHOTON_GUESS_MARGIN;   // on the guessed squared radius (see the kernel)

// A wave's LDS (dynamic, sized at launch): the candidate buffer, the radix-select histogram, the stack of blocks whose
// children are still to be measured (only blocks with blocks below them wait there: 64 x (layers - 2) entries, at least 64)
// and the list of blocks to examine.  7 936 bytes per wave for a 200 000-photon map: five waves per SIMD (round 2's fixed
// 9 216-byte layout allowed four; the fifth is worth 21 %).
struct WaveLds {
    float *d2;               // [kCap]
    int *idx;                // [kCap]
    unsigned *hist;          // [256]
    int *stack;              // [stack_cap] pending block roots whose children are still to be tested
    int *list;               // [kList] blocks to examine
    float *lb;               // [kList] squared distance from the query to each listed block's own box
};
#ifndef MIRO_PHOTON_EXAMINE
#define MIRO_PHOTON_EXAMINE 4
#endif
#ifndef MIRO_PHOTON_SMALL_BATCH
#define MIRO_PHOTON_SMALL_BATCH 16
#endif
#ifndef MIRO_PHOTON_TAIL_QUERIES
#define MIRO_PHOTON_TAIL_QUERIES 102400
#endif
constexpr unsigned kSmallBatch = MIRO_PHOTON_SMALL_BATCH;     // queries per batch at the end of a launch (see the kernel)
constexpr unsigned long long kTailQueries = MIRO_PHOTON_TAIL_QUERIES;   // how many queries that end is: 1.25 small batches per resident wave
constexpr int kExamine = MIRO_PHOTON_EXAMINE;   // listed blocks examined per step (their loads are in flight together)
constexpr int kList = 128;         // an expansion adds up to 64 entries and runs only while at most kList - 64 are waiting
__host__ __device__ inline int wave_lds_words(int stack_cap) { return 2 * kCap + 256 + stack_cap + 2 * kList; }
__device__ __forceinline__ WaveLds wave_lds(int *base, int stack_cap) {
    WaveLds w;
    w.d2 = reinterpret_cast<float *>(base);
    w.idx = base + kCap;
    w.hist = reinterpret_cast<unsigned *>(base + 2 * kCap);
    w.stack = base + 2 * kCap + 256;
    w.list = base + 2 * kCap + 256 + stack_cap;
    w.lb = reinterpret_cast<float *>(base + 2 * kCap + 256 + stack_cap + kList);
    return w;
}

// a cross-lane hand-off through wave-private LDS: release + acquire at wavefront scope and a wave barrier, so that neither
// the compiler nor the memory model may move LDS accesses across it (no instruction is emitted beyond a waitcnt)
__device__ __forceinline__ void lds_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// k-th smallest (1-based rank `k`) of d2[0..count): returns its bit pattern and how many entries equal to it belong
// to the k smallest.  Non-negative floats order like their bit patterns.
__device__ __forceinline__ unsigned radix_select(const WaveLds &w, int count, int k, int lane, int &eq_keep) {
    unsigned prefix = 0;
    int remaining = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = lane; b < 256; b += 64) w.hist[b] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        for (int i = lane; i < count; i += 64) {
            const unsigned key = __float_as_uint(w.d2[i]);
            const bool in = shift == 24 || (key >> (shift + 8)) == (prefix >> (shift + 8));
            if (in) atomicAdd(&w.hist[(key >> shift) & 255u], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // lane holds bins 4*lane .. 4*lane+3
        const unsigned h0 = w.hist[4 * lane], h1 = w.hist[4 * lane + 1], h2 = w.hist[4 * lane + 2], h3 = w.hist[4 * lane + 3];
        const int local = (int)(h0 + h1 + h2 + h3);
        int incl = local;
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        const unsigned long long m = __ballot(incl >= remaining);
        const int tl = __ffsll((long long)m) - 1;            // first lane whose cumulative count reaches the rank
        int bin = 0, before = 0;
        if (lane == tl) {
            int c = incl - local;
            if (c + (int)h0 >= remaining) { bin = 4 * lane; before = c; }
            else if (c + (int)(h0 + h1) >= remaining) { bin = 4 * lane + 1; before = c + (int)h0; }
            else if (c + (int)(h0 + h1 + h2) >= remaining) { bin = 4 * lane + 2; before = c + (int)(h0 + h1); }
            else { bin = 4 * lane + 3; before = c + (int)(h0 + h1 + h2); }
        }
        bin = __shfl(bin, tl, 64);
        before = __shfl(before, tl, 64);
        prefix |= (unsigned)bin << shift;
        remaining -= before;
    }
    eq_keep = remaining;
    return prefix;
}

// keep the k smallest candidates at the front of the buffer; returns the k-th distance
__device__ __forceinline__ float compress(const WaveLds &w, int &count, int k, int lane) {
    int eq_keep;
    const unsigned kth = radix_select(w, count, k, lane, eq_keep);
    int out = 0, eq_seen = 0;
    for (int base = 0; base < count; base += 64) {
        const int i = base + lane;
        float d = 0.0f;
        int id = 0;
        bool lt = false, eq = false;
        if (i < count) {
            d = w.d2[i]; id = w.idx[i];
            const unsigned key = __float_as_uint(d);
            lt = key < kth; eq = key == kth;
        }
        const unsigned long long meq = __ballot(eq);
        const int eq_rank = eq_seen + __popcll(meq & ((1ull << lane) - 1ull));
        const bool keep = lt || (eq && eq_rank < eq_keep);
        const unsigned long long mk = __ballot(keep);
        if (keep) {
            const int pos = out + __popcll(mk & ((1ull << lane) - 1ull));
            w.d2[pos] = d; w.idx[pos] = id;                  // pos <= i: never overtakes an unread slot
        }
        out += __popcll(mk);
        eq_seen += __popcll(meq);
    }
    count = out;
    return __uint_as_float(kth);
}

// locate_photons (PhotonMap.cpp:152-243) in the reference's own order while np->dist2[0] is still max_dist^2, i.e. up
// to the first overflow: near child first, the far child if the splitting plane is inside max_dist, the node itself
// after both.  No stack: the path is implicit in the heap index, and whether a child was the near one is kept as one bit
// per level.  Returns the index of the farthest of the first k candidates (earliest among equals) once candidate k+1
// has been seen; 0 when the whole (reachable) tree holds at most k candidates.  radius_after = the reference's np->dist2[0]
// right after that replacement: the largest distance among the first k+1 candidates without m* -- every later
// replacement only shrinks it, so the cooperative search may start from it instead of from max_dist^2.
__device__ __forceinline__ int first_overflow(const PhotonMapDev &pm, float qx, float qy, float qz, float nx, float ny, float nz,
                                              float md2, int k, float &radius_after, unsigned &visits) {
    radius_after = md2;
    if (pm.n < 1) return 0;
    // max_dist is 1e10 in the reference (Miro.h:17): larger than any distance between the query and a photon.  When the whole
    // map's box lies inside max_dist of the query, so does every splitting plane, and "the far child if the plane is inside
    // max_dist" (:168-172) needs no look at the plane -- one record fetch less per far-side step of a walk that is bound by them.
    bool all_inside;
    {
        const float4 lo = pm.boxes[2], hi = pm.boxes[3];                 // box of everything below the root
        const float ex = fmaxf(fabsf(qx - lo.x), fabsf(qx - hi.x)), ey = fmaxf(fabsf(qy - lo.y), fabsf(qy - hi.y)),
                    ez = fmaxf(fabsf(qz - lo.z), fabsf(qz - hi.z));
        const float far2 = fmaxf(ex, fmaxf(ey, ez));
        all_inside = far2 * far2 < md2;                                  // side * side <= far2 * far2 for every plane (monotone)
    }
    int i = 1, cnt = 0, best_i = 0;
    float best_d2 = -1.0f, second_d2 = -1.0f;
    unsigned near_right = 0;              // bit d: at the ancestor of depth d the near child was the right one
    bool descend = true;
    while (true) {
        if (descend && i < pm.half) {                                       // :160-172, going down
            const float4 A = pm.rec[2 * i];
            const int plane = __float_as_int(A.w);
            const float side = (plane == 0 ? qx : (plane == 1 ? qy : qz)) - (plane == 0 ? A.x : (plane == 1 ? A.y : A.z));
            const int d = 31 - __clz(i);
            const unsigned right = side > 0.0f ? 1u : 0u;
            near_right = (near_right & ~(1u << d)) | (right << d);
            i = 2 * i + (int)right;
            continue;
        }
        // the photon at node i (:177-186)
        {
            visits++;
            const float4 A = pm.rec[2 * i], D = pm.rec[2 * i + 1];
            float dd = A.x - qx;
            float d2 = dd * dd;
            dd = A.y - qy; d2 += dd * dd;
            dd = A.z - qz; d2 += dd * dd;
            const float facing = (D.x * nx + D.y * ny) + D.z * nz;
            if (d2 < md2 && facing < 0.0f) {
                if (++cnt > k) {                                            // candidate k+1: the heap root goes (:222-238)
                    radius_after = fmaxf(second_d2, d2);
                    return best_i;
                }
                if (d2 > best_d2) { second_d2 = best_d2; best_d2 = d2; best_i = i; }
                else if (d2 > second_d2) second_d2 = d2;
            }
        }
        // back up: a near child hands over to its far sibling (if the plane is inside the radius), a far child to the parent
        if (i == 1) return 0;
        const int parent = i >> 1;
        const int d = 31 - __clz(parent);
        const bool was_near = (unsigned)(i & 1) == ((near_right >> d) & 1u);
        descend = false;
        if (was_near && all_inside) { i ^= 1; descend = true; continue; }      // every plane is inside max_dist: no need to look
        if (was_near) {
            const float4 A = pm.rec[2 * parent];
            const int plane = __float_as_int(A.w);
            const float side = (plane == 0 ? qx : (plane == 1 ? qy : qz)) - (plane == 0 ? A.x : (plane == 1 ? A.y : A.z));
            if (side * side < md2) { i ^= 1; descend = true; continue; }
        }
        i = parent;
    }
}

// STATS: work counters of the launch, added to stats[] (mr_photon_map_get_stats, miro_hip.h): queries answered, blocks examined,
// photon records examined by the search (32 bytes each: position + direction), radius tightenings, photon records examined by
// the reference-order pre-pass, searches repeated with the safe radius, child boxes measured, candidates, blocks expanded.
// Five waves per SIMD (96 registers) is what 7 680 bytes of LDS per wave allow; the fifth wave is worth 21 % (50.3 -> 41.5 ms),
// so the register allocation is held to it.
template <bool STATS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(STATS ? 4 : 5, 8))) void irradiance_kernel(PhotonMapDev pm, const float *qpos, const float *qnrm,
                                                            unsigned long long nq, float max_dist, int k, float *irrad,
                                                            int *found_out, float *r2_out, unsigned long long *stats, int stack_cap,
                                                            unsigned *work_counter) {
    extern __shared__ int s_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const WaveLds w = wave_lds(s_lds + wv * wave_lds_words(stack_cap), stack_cap);
    // node of this lane inside a block: level lv (0..5), offset within the level
    const int lv = 31 - __clz(lane + 1);
    const int off_in_level = lane + 1 - (1 << lv);
    const bool node_lane = lane < 63;

    const float md2 = max_dist * max_dist;
    unsigned long long st_queries = 0, st_blocks = 0, st_records = 0, st_tighten = 0, st_prepass = 0, st_retries = 0;
    unsigned long long st_reached = 0, st_cands = 0, st_top = 0, st_mid = 0, st_unguessed = 0;
    // A wave takes a BATCH of consecutive queries at a time: first every lane finds its own query's m* (first_overflow), then
    // the wave searches the batch's queries one after the other.  Batches are handed out by a counter (one atomic per batch, no
    // barrier: waves share nothing): 64 queries each, and kSmallBatch each for the last kTailQueries of the launch.  Dealt out
    // statically -- batch b to wave b mod (waves of the grid) -- a wave got one or two batches of 3.7 ms each in a 23.5 ms launch
    // (config 5: 32 400 batches over 20 480 waves, 5 120 resident) and the launch drained for the length of a batch.  Batches of 64
    // from the counter: 23.2 -> 22.8 ms; the last 5 % in batches of 16: 22.15 ms (smaller or more of them cost more than they
    // return: the pre-pass of a batch of 8 runs on 8 lanes; profiles/r03_photon_batches.log).  The counter is zero before the
    // launch; the wave that reads the last value re-arms it.
    const unsigned long long n_big = (nq > kTailQueries ? nq - kTailQueries : 0ull) / 64ull;   // batches of 64
    const unsigned long long n_small = (nq - 64ull * n_big + kSmallBatch - 1) / kSmallBatch;  // batches of kSmallBatch (the last one ragged)
    unsigned pulled = 0;
    for (;;) {
      if (lane == 0) pulled = atomicAdd(work_counter, 1u);
      pulled = (unsigned)__builtin_amdgcn_readfirstlane((int)pulled);
      if (pulled >= n_big + n_small) break;
      const unsigned long long base = pulled < n_big ? 64ull * pulled : 64ull * n_big + (unsigned long long)kSmallBatch * (pulled - n_big);
      const unsigned long long bend0 = base + (pulled < n_big ? 64ull : (unsigned long long)kSmallBatch);
      const unsigned long long bend = bend0 < nq ? bend0 : nq;                                 // end of this batch
      int my_mstar = 0;
      float my_radius = md2;
      {
          const unsigned long long qa = base + (unsigned)lane;
          unsigned visits = 0;
          if (qa < bend) {
              const float ax = qnrm[3 * qa];
              if (ax == ax)
                  my_mstar = first_overflow(pm, qpos[3 * qa], qpos[3 * qa + 1], qpos[3 * qa + 2], ax, qnrm[3 * qa + 1], qnrm[3 * qa + 2], md2, k,
                                            my_radius, visits);
          }
          if (STATS) {
              for (int off = 32; off > 0; off >>= 1) visits += __shfl_xor(visits, off, 64);
              st_prepass += visits;
          }
      }
#ifdef MIRO_PHOTON_PREPASS_ONLY      /* timing probe: the reference-order pre-pass alone (results are NOT the estimate) */
      if (base + (unsigned)lane < bend) { irrad[3 * (base + lane)] = my_radius; if (found_out) found_out[base + lane] = my_mstar; }
      continue;
#endif
      const int n_here = (int)(bend - base);
      // The previous query of this wave (the neighbouring pixel) and its final radius: a GUESS for this one's.  If both
      // queries saw the same candidates the k-th nearest of this one would lie within sqrt(prev) + |q - q_prev| (triangle
      // inequality); they need not (the facing test depends on the normal, m* differs), so the guess is verified: a search
      // inside the guessed radius that finds at least k candidates has found the k nearest (anything nearer than the k-th
      // is inside the radius too) and the result is the full search's; one that finds fewer is repeated with the safe
      // radius of the pre-pass.  A tighter start means fewer blocks that touch the sphere.
      bool prev_ok = false;
      float prev_r2 = 0.0f, pqx = 0.0f, pqy = 0.0f, pqz = 0.0f;
      for (int t = 0; t < n_here; t++) {
        const unsigned long long q = base + (unsigned)t;
        const int mstar = __shfl(my_mstar, t, 64);            // 0: this query never overflows
        const float radius0 = __shfl(my_radius, t, 64);
        const float qx = qpos[3 * q], qy = qpos[3 * q + 1], qz = qpos[3 * q + 2];
        const float nx = qnrm[3 * q], ny = qnrm[3 * q + 1], nz = qnrm[3 * q + 2];
        if (nx != nx) {                                       // NaN normal = "no query here" (mr_final_gather: miss / non-diffuse hit)
            if (lane == 0) {
                irrad[3 * q] = 0.0f; irrad[3 * q + 1] = 0.0f; irrad[3 * q + 2] = 0.0f;
                if (found_out) found_out[q] = 0;
                if (r2_out) r2_out[q] = 0.0f;
            }
            continue;
        }
        // np.dist2[0]: max_dist^2 (PhotonMap.cpp:99) until the first overflow; for a query that overflows, the reference's
        // radius right after the overflow bounds everything that can still enter the result -- one ulp is added because the
        // photon AT that distance is in the set while candidates must be strictly nearer than the radius
        const float safe_r2 = mstar != 0 ? __uint_as_float(__float_as_uint(radius0) + 1u) : md2;
        float r2 = safe_r2;
        bool guessed = false;
        if (prev_ok && mstar != 0) {
            float dx = qx - pqx, dy = qy - pqy, dz = qz - pqz;
            const float g = sqrtf(prev_r2) + sqrtf((dx * dx + dy * dy) + dz * dz);
            const float g2 = (g * g) * kGuessMargin;               // strictly beyond the would-be k-th photon; only a guess anyway
            if (g2 < safe_r2) { r2 = g2; guessed = true; }
        }
        int count;
        bool evicted;
        if (STATS) { st_queries++; if (!guessed) st_unguessed++; }
      search_again:
        count = 0;
        evicted = false;
        // Two phases share one loop.  EXPAND: a pending block root r (LDS stack, newest first) hands each lane one of its 64
        // child blocks: the lane reads that child's two boxes and computes the squared distance from the query to the box of
        // the child's own 63 photons and to the box of everything below it -- per axis e = lo - q, hi - q or 0, summed in the
        // order the photon test sums, so that (floating-point operations being monotone) no photon inside a box computes
        // nearer than its box.  Children whose own box reaches into the radius go on the LIST, children whose subtree box does
        // and which have blocks below them go on the stack.  EXAMINE: a listed block's 63 photons are tested, one per lane,
        // and accepted ones appended to the candidate buffer.  Which blocks are examined no longer depends on one another:
        // the result is the set of facing photons inside the radius whichever way they are found.
        int sp = 0, nlist = 0;
        if (pm.n >= 1) {
            if (lane == 0) { w.stack[0] = 1; w.list[0] = 1; w.lb[0] = 0.0f; }
            sp = pm.layers > 1 ? 1 : 0;
            nlist = 1;
        }
        lds_handoff();
        while (sp > 0 || nlist > 0) {
            if (sp > 0 && nlist <= kList - 64) {
                // ---- expand
                sp--;
                const int r = w.stack[sp];                                      // broadcast
                const int L = (31 - __clz(r)) / 6;                              // r's layer; its children are layer L + 1
                const int child = (r << 6) + lane;
                const bool valid = child <= pm.n;
                float d_own = 0.0f, d_sub = 0.0f;
                if (valid) {
                    const float4 *bx = pm.boxes + 4 * (size_t)(pm.layer_base[L + 1] + (child - (1 << (6 * (L + 1)))));
                    const float4 olo = bx[0], ohi = bx[1], slo = bx[2], shi = bx[3];
                    float e = qx < olo.x ? olo.x - qx : (qx > ohi.x ? ohi.x - qx : 0.0f);
                    d_own = e * e;
                    e = qy < olo.y ? olo.y - qy : (qy > ohi.y ? ohi.y - qy : 0.0f); d_own += e * e;
                    e = qz < olo.z ? olo.z - qz : (qz > ohi.z ? ohi.z - qz : 0.0f); d_own += e * e;
                    e = qx < slo.x ? slo.x - qx : (qx > shi.x ? shi.x - qx : 0.0f);
                    d_sub = e * e;
                    e = qy < slo.y ? slo.y - qy : (qy > shi.y ? shi.y - qy : 0.0f); d_sub += e * e;
                    e = qz < slo.z ? slo.z - qz : (qz > shi.z ? shi.z - qz : 0.0f); d_sub += e * e;
                }
                if (STATS) { st_top++; st_reached += (unsigned)__popcll(__ballot(valid)); }
                const bool look = valid && d_own < r2;
                const bool down = valid && d_sub < r2 && ((long long)child << 6) <= (long long)pm.n;
                const unsigned long long ml = __ballot(look), md = __ballot(down), lt = (1ull << lane) - 1ull;
                if (look) { const int pos = nlist + __popcll(ml & lt); w.list[pos] = child; w.lb[pos] = d_own; }
                if (down) w.stack[sp + __popcll(md & lt)] = child;
                nlist += __popcll(ml);
                sp += __popcll(md);
                lds_handoff();                                                  // list / stack entries are read by every lane
                continue;
            }
            // ---- examine: up to kExamine listed blocks per step -- all their records are requested before any is tested
            // (the steps no longer depend on one another, only on the candidate count and the radius)
            int jb[kExamine];
            bool go[kExamine], vb[kExamine];
            float4 Ab[kExamine], Db[kExamine];
#pragma unroll
            for (int u = 0; u < kExamine; u++) {
                go[u] = false; jb[u] = 0;
                if (nlist > 0) {                                                // wave-uniform
                    nlist--;
                    const int bu = w.list[nlist];                               // broadcast
                    go[u] = w.lb[nlist] < r2;                                   // else: the radius has shrunk below this block meanwhile
                    jb[u] = (bu << lv) + off_in_level;                          // < 2^24 (checked at launch)
                }
                vb[u] = go[u] && node_lane && jb[u] <= pm.n;
                Ab[u] = make_float4(0.f, 0.f, 0.f, 0.f); Db[u] = Ab[u];
                if (vb[u]) { Ab[u] = pm.rec[2 * jb[u]]; Db[u] = pm.rec[2 * jb[u] + 1]; }
            }
#pragma unroll
            for (int u = 0; u < kExamine; u++) {
                if (!go[u]) continue;                                           // wave-uniform
                const bool valid = vb[u];
                const int j = jb[u];
                const float4 A = Ab[u], D = Db[u];
                if (STATS) { st_blocks++; st_records += (unsigned)__popcll(__ballot(valid)); }
                // the photon itself (:177-186)
                float dd = A.x - qx;
                float d2 = dd * dd;
                dd = A.y - qy; d2 += dd * dd;
                dd = A.z - qz; d2 += dd * dd;
                const float facing = (D.x * nx + D.y * ny) + D.z * nz;
                // the reference never descends below a node of index >= half_stored_photons (:160): the two or three photons
                // whose parent is such a node are never found by it -- nor here; m* never takes part either (see the header)
                const bool cand = valid && d2 < r2 && facing < 0.0f && j != mstar && (j == 1 || (j >> 1) < pm.half);
                const unsigned long long mc = __ballot(cand);
                if (cand) {
                    const int pos = count + __popcll(mc & ((1ull << lane) - 1ull));
                    w.d2[pos] = d2; w.idx[pos] = j;
                }
                count += __popcll(mc);
                if (STATS) st_cands += (unsigned)__popcll(mc);
                if (count > kTighten && count > k) {          // keep the k nearest so far; the k-th is the new radius
                    lds_handoff();                            // the candidates appended above are read by other lanes
                    r2 = compress(w, count, k, lane);
                    evicted = true;
                    if (STATS) st_tighten++;
                }
            }
        }
        if (guessed && count < k) {                           // the guessed radius does not hold the k nearest: the safe one
            guessed = false;
            r2 = safe_r2;
            if (STATS) st_retries++;
            lds_handoff();
            goto search_again;
        }
        if (count > k) { lds_handoff(); r2 = compress(w, count, k, lane); evicted = true; if (STATS) st_tighten++; }
        else if (mstar != 0) {
            // exactly k candidates besides m*: the reference's heap holds them all and its root is the farthest
            lds_handoff();
            float mx = 0.0f;
            for (int i = lane; i < count; i += 64) mx = fmaxf(mx, w.d2[i]);
            for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            r2 = mx; evicted = true;
        }
        const float r2_final = evicted ? r2 : md2;            // dist2[0] only moves once the heap overflows (:240)
        prev_ok = evicted; prev_r2 = r2_final; pqx = qx; pqy = qy; pqz = qz;
        lds_handoff();
        float sr = 0.f, sg = 0.f, sb = 0.f;
        for (int i = lane; i < count; i += 64) {
            const float4 P = pm.power[w.idx[i]];
            sr += P.x; sg += P.y; sb += P.z;
        }
        for (int off = 32; off > 0; off >>= 1) {
            sr += __shfl_xor(sr, off, 64); sg += __shfl_xor(sg, off, 64); sb += __shfl_xor(sb, off, 64);
        }
        if (lane == 0) {
            const float tmp = (float)((1.0 / 3.14159265358979323846) / (double)r2_final);     // :136
            irrad[3 * q] = sr * tmp; irrad[3 * q + 1] = sg * tmp; irrad[3 * q + 2] = sb * tmp;
            if (found_out) found_out[q] = count;
            if (r2_out) r2_out[q] = r2_final;
        }
      }
    }
    // every wave of the grid reads exactly one value at or beyond the number of batches: the last of them re-arms the counter
    if (lane == 0 && pulled == (unsigned)(n_big + n_small) + gridDim.x * kWaves - 1u) atomicExch(work_counter, 0u);
    if (STATS && lane == 0 && stats) {
        atomicAdd(&stats[0], st_queries); atomicAdd(&stats[1], st_blocks); atomicAdd(&stats[2], st_records);
        atomicAdd(&stats[3], st_tighten); atomicAdd(&stats[4], st_prepass); atomicAdd(&stats[5], st_retries);
        atomicAdd(&stats[6], st_reached); atomicAdd(&stats[7], st_cands); atomicAdd(&stats[8], st_top); atomicAdd(&stats[9], st_mid);
        atomicAdd(&stats[10], st_unguessed);
    }
}

}  // namespace

mr_status launch_irradiance(const PhotonMapDev &pm, unsigned *work_counter, const float *d_pos, const float *d_normal, unsigned long long nq,
                            float max_dist, uint32_t k, float *d_irrad, int32_t *d_found, float *d_r2, unsigned long long *d_stats,
                            hipStream_t stream) {
    static_assert(kTighten + 64 <= kCap && kKnnMaxK <= kTighten, "candidate buffer must hold k plus one block");
    if (pm.n >= (1 << 24))
        return fail(MR_ERR_INVALID, "photon maps of 2^24 photons or more need a fifth layer of blocks");
    // layers of 6-level blocks that hold nodes; an expanded block leaves at most 64 roots of the next layer pending
    int layers = 1;
    while (layers < 4 && (1ll << (6 * layers)) <= (long long)pm.n) layers++;
    const int stack_cap = layers > 2 ? 64 * (layers - 2) : 64;      // only blocks that have blocks below them wait on the stack
    const size_t lds = (size_t)kWaves * wave_lds_words(stack_cap) * sizeof(int);
    if (!work_counter) return fail(MR_ERR_STATE, "photon map without hand-out counters (not uploaded?)");
    if (nq >= (1ull << 31)) return fail(MR_ERR_INVALID, "at most 2^31 - 1 queries per estimate launch");
    // every resident slot gets a workgroup (five per CU, a few more in case fewer registers are taken): they pull their batches
    unsigned long long blocks = (nq + kWaves - 1) / kWaves;
    if (blocks > 256ull * 6ull) blocks = 256ull * 6ull;
    if (d_stats)
        hipLaunchKernelGGL(irradiance_kernel<true>, dim3((unsigned)blocks), dim3(kBlock), lds, stream, pm, d_pos, d_normal, nq, max_dist,
                           (int)k, d_irrad, d_found, d_r2, d_stats, stack_cap, work_counter);
    else
        hipLaunchKernelGGL(irradiance_kernel<false>, dim3((unsigned)blocks), dim3(kBlock), lds, stream, pm, d_pos, d_normal, nq, max_dist,
                           (int)k, d_irrad, d_found, d_r2, d_stats, stack_cap, work_counter);
    MR_HIP_CHECK(hipGetLastError());
    return MR_OK;
}

}  // namespace mr
