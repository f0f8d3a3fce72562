// ptk_kernels_coopk.hpp -- k > 1 (k <= 64): the long searches of a batch finished by a whole wavefront each.
//
// What this is for (profiles/r05_notes.txt item 2, tools/wave_trace.py).  knn_reg_kernel runs every query to its end
// in its lane, and a lane takes ~4.4 us per leaf it visits once its wavefront's other lanes have finished (two or
// three dependent memory round trips per turn and nobody to share the issue slots with).  On BASELINE config 3
// (knn = 16, 7.2 M queries) the mean query visits 12 leaves, 57 queries visit more than 500 and one visits 1 049:
// the launch is over its 112 514 wavefronts after 3.7 ms and then waits another millisecond for five of them, one
// lane each, that started in the first microsecond.  No order of the batch shortens a chain of dependent visits;
// only more lanes per query do.
//
// So the general kernel is CAPPED (traverse<..., CAPPED>): a query that has entered more than `cap` far children
// stops, leaves its k-list in its output row and hands its stack over -- every pending far child that can still
// matter with the state it would be entered with (Task, ptk_kernels.hpp) -- and knn_coop_kernel gives each such query
// a wavefront: the 64 lanes share a pool of subtrees and a bound, every lane advances its own subtree by one node per
// step and keeps a k-list of its own that starts as a copy of the handed-over list.  At the end the lists are merged.
//
// Why the merged result is the reference's (kd_tree_search.hpp:52-105, search_visitor.hpp:83-123).  The capped
// traversal IS the reference's up to the hand-over, so the handed-over list L0 is the reference's list at that point
// and its pending subtrees are exactly what the reference has left to look at.  Let S be the points of those
// subtrees, and D the k-th smallest float distance over L0 and S.  Assume (a) the k + 1 smallest distances over
// L0 and S are pairwise different, and (b) every point p of S among the k nearest has `gmax(p) <= D`: no far child
// on the way from the hand-over to p's leaf has a float box distance above D (on scans the point that DEFINES a
// leaf's bounding plane often sits one rounding below the incrementally updated box distance of its own leaf, so the
// certificate must not ask for `gmax(p) <= d(p)`: a fifth of the long queries of config 3 would fail it; and since
// the reference's list cannot hold p before it has reached p, its bound on the way to p is in fact never below the
// RUNNER-UP distance D' -- the (k + 1)-th smallest -- so `gmax(p) <= D'` is enough, with D' bounded from below by
// what the lanes still hold, what they turned away or pushed out, and the box distances of what they pruned).  The
// reference's bound is never below D (its list holds k of these points), so by (b) it enters every far child on the way to such a p
// (`max() >= box distance`, :99), reaches p, and accepts it: by (a) fewer than k points are nearer than p, so its
// bound then is strictly above d(p).  An accepted p is only ever displaced by k nearer points, of which there are
// none.  Points outside the k nearest never end up in the reference's list either (the k nearest get in, see above,
// and push them out or keep them out).  Hence the reference's final list is the k nearest of L0 and S by distance --
// which is what the merge below produces, whatever the order the wavefront visited S in.
//
// Equal distances.  The long searches are the ones with a ring of points all about equally far away, and two float
// distances out of hundreds of nearly equal ones do coincide: 33 of the 31 073 queries of config 3 that visit more
// than 160 leaves have a tie among their 17 nearest.  The reference keeps equal distances in the order it visits them
// (strict comparisons, search_visitor.hpp:30,107), i.e. in depth-first order, near child first: dfs_before() of
// ptk_kernels.hpp decides that for two record positions without knowing anything else.  So a query whose merge meets
// equal distances (in one round, in consecutive rounds, or turned away at the edge of a lane's list) gets a SECOND
// sweep of its subtrees with the bound fixed at D: every point at a distance <= D is collected (there are k of them
// plus the ties), ranked by (distance, handed-over entries in their order, then depth-first order), and the first k
// are the row.  By the argument above -- read with "nearer, or as far and earlier" for "nearer", and with D' = the
// distance of the (k + 1)-th entry of that ranking (or D, the bound everything was collected under, when the ranking has
// only k entries) -- that is the reference's row: before the reference has reached a chosen point p its list holds k
// OTHER points or fewer, its bound is therefore at least D', so (b) `gmax(p) <= D'` makes it enter every far child on
// the way to p; it then accepts p (fewer than k points are ahead of p in the ranking, so its k-th entry then is
// strictly farther) and nothing displaces p afterwards.
//
// What (b) must NOT be compared with (the parity failure of profiles/r05_notes.txt item 24, closed in
// profiles/r06_notes.txt item 1): the value `dk` of the merge's k-th ROUND.  A round takes the smallest head of all
// lanes and pops every lane whose head equals it -- several different points of one distance count as one round -- so
// with ties dk lies ABOVE the true k-th distance (60 000 points on a line, knn = 33: 55 points within dk, 43 ulps above
// the 33rd distance T).  dk is a sound bound to collect the second sweep's entries with, and nothing else.  The box
// distances themselves never differ between the two searches: a node's box distance is a function of its root path
// alone (`nbd - old_offset + new_offset` with the offsets the path set, kd_tree_search.hpp:91-99), and Task carries
// {nbd, off[3]} down every path exactly as traverse<> does (tools/trace_box_distance.py replays the reference in numpy
// and prints both per far child of a point's root path: equal on every step; the reference turns away from the leaf
// of point 3505 because that leaf's box distance has drifted one ulp above T, which `gmax(p) <= D' = T` detects --
// the query is then searched by knn_redo_kernel).  The first sweep's certificate never had the flaw: without ties
// every round pops exactly one point, so its D and D' are the true k-th and (k + 1)-th distances.
//
// When (b) cannot be shown, or a bound is in the denormal or overflow range, or the pool and its HBM spill (or the
// 64 slots of the second sweep: points on a grid) overflowed, the query goes to knn_redo_kernel: the reference search
// from the root, one lane, as before.  A lane prunes a subtree only when its box distance exceeds bound * (1 + 2^-10), the margin
// knn1_coop_kernel derives: a pruned subtree then holds no point at a float distance <= the bound, in particular
// none that ties with the k-th.
#pragma once

#include "ptk_kernels.hpp"

namespace ptk {

// (the words of the counters block this search adds to -- kKnnWhyPool .. kKnnTieSweeps -- are named in ptk_kernels.hpp,
// next to the other counters: the unit that reads them back does not include this file)
constexpr uint32_t kKnnTieSlots = 64;        // points at a distance <= D the second sweep can rank
constexpr uint32_t kKnnPosFlag = 0x80000000u;  // second sweep: the entry is a record position (else: its rank in the handed-over list)

// KnnRegPolicy (ptk_kernels.hpp) with a third word per entry: the largest box distance of a far child on the way to
// the point (0 for the entries handed over: the reference has visited those).
template <int K>
struct KnnCertPolicy {
  float ld[K];
  int32_t li[K];
  float lg[K];
  __device__ __forceinline__ void init(uint32_t k) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      ld[j] = (uint32_t)j + k >= (uint32_t)K ? 3.402823466e+38f : __uint_as_float(0xFF800000u);  // unused slots: -inf
      li[j] = 0;
      lg[j] = 0.0f;
    }
  }
  __device__ __forceinline__ void visit(int32_t idx, float d, float g) {
    if (ld[K - 1] > d) {
      bool below[K];
#pragma unroll
      for (int j = 0; j < K; ++j) below[j] = d < ld[j];
#pragma unroll
      for (int j = K - 1; j >= 1; --j) {
        li[j] = below[j - 1] ? li[j - 1] : (below[j] ? idx : li[j]);
        lg[j] = below[j - 1] ? lg[j - 1] : (below[j] ? g : lg[j]);
        ld[j] = f_med3(ld[j - 1], ld[j], d);
      }
      li[0] = below[0] ? idx : li[0];
      lg[0] = below[0] ? g : lg[0];
      ld[0] = below[0] ? d : ld[0];
    }
  }
};

__device__ __forceinline__ float wave_min_f32(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const float o = __shfl_xor(v, d);
    v = o < v ? o : v;
  }
  return v;
}

// LDS of a wavefront of the cooperative search, in 32-bit words: the pool [6][POOL], the shared bound, the row the merge
// produced {index, distance}[max(K, 32)], the entries of the second sweep {distance, tag, box distance}[kKnnTieSlots].
constexpr uint32_t knn_coop_row(uint32_t K) { return K > 32u ? K : 32u; }  // entries of the merged row
constexpr uint32_t knn_coop_lds_words(uint32_t pool, uint32_t K = 32u) {
  return 6u * pool + 1u + 2u * knn_coop_row(K) + 3u * kKnnTieSlots;
}

// One sweep of a query's pending subtrees by the 64 lanes of the wavefront (see the head of this file).
//   COLLECT = false: every lane keeps a k-list (`pol`), the shared bound *gbest follows the smallest k-th distance
//                    any lane holds; tie_d = the last distance a lane met that was EQUAL to its k-th.
//   COLLECT = true:  the bound stays at `fixed` (= D); every point at a distance <= D becomes an entry
//                    {distance, position | kKnnPosFlag, box distance} behind the n_ent entries already there.
// Returns false if a subtree was lost (pool and spill full).
template <int K, int POOL, bool COLLECT, class M = MetricL2>
__device__ __forceinline__ bool knn_coop_sweep(
    const DevTree& t, float qx, float qy, float qz, const Task* __restrict__ src, uint32_t nt, PTK_LDS uint32_t* pool,
    PTK_LDS uint32_t* gbest, Task* __restrict__ spill_w, uint32_t spill_cap, KnnCertPolicy<K>& pol, float& tie_d,
    float& drop_min, float& prune_min, float fixed, PTK_LDS uint32_t* ent, uint32_t& n_ent) {
  const uint4* __restrict__ nodes = t.nodes;
  const float4* __restrict__ pts = t.pts;
  const uint32_t lane = threadIdx.x;
  const uint64_t below = (1ull << lane) - 1ull;
  bool ok = true;
  uint32_t count = nt;  // subtrees in the pool (uniform)
  for (uint32_t i = lane; i < nt; i += 64u) {  // the capped traversal's stack, next-to-visit on top
    const Task tk = src[i];
    const uint32_t sl = nt - 1u - i;
    pool[0 * POOL + sl] = tk.ref;
    pool[1 * POOL + sl] = __float_as_uint(tk.nbd);
    pool[2 * POOL + sl] = __float_as_uint(tk.off0);
    pool[3 * POOL + sl] = __float_as_uint(tk.off1);
    pool[4 * POOL + sl] = __float_as_uint(tk.off2);
    pool[5 * POOL + sl] = __float_as_uint(tk.gmax);
  }
  if (lane == 0) *gbest = __float_as_uint(COLLECT ? fixed : pol.ld[K - 1]);
  bool busy = false;
  uint32_t ref = 0;
  uint32_t spill_n = 0;  // tasks parked in HBM (uniform)
  float nbd = 0.0f, off0 = 0.0f, off1 = 0.0f, off2 = 0.0f, gmax = 0.0f;

  for (;;) {
    // A drained pool takes back what had to be parked in HBM (the newest first, up to half a pool).
    if (count == 0u && spill_n != 0u) {  // (uniform)
      const uint32_t m = spill_n < (uint32_t)(POOL / 2) ? spill_n : (uint32_t)(POOL / 2);
      for (uint32_t i = lane; i < m; i += 64u) {
        const Task tk = spill_w[spill_n - m + i];
        pool[0 * POOL + i] = tk.ref;
        pool[1 * POOL + i] = __float_as_uint(tk.nbd);
        pool[2 * POOL + i] = __float_as_uint(tk.off0);
        pool[3 * POOL + i] = __float_as_uint(tk.off1);
        pool[4 * POOL + i] = __float_as_uint(tk.off2);
        pool[5 * POOL + i] = __float_as_uint(tk.gmax);
      }
      count = m;
      spill_n -= m;
    }
    // (the ballot is also where the lanes meet after the pool and the bound were written)
    const bool want = !busy;
    const uint64_t wmask = __ballot(want);
    const float best = __uint_as_float(*gbest);
    const float bm = f_add(best, f_mul(best, 0.0009765625f));  // bound * (1 + 2^-10): see the head of this file
    bool fresh = false;  // taken in the handed-over form: the parent branch has to be read first
    if (want) {
      const uint32_t rank = (uint32_t)__popcll(wmask & below);
      if (rank < count) {
        const uint32_t sl = count - 1u - rank;
        ref = pool[0 * POOL + sl];
        nbd = __uint_as_float(pool[1 * POOL + sl]);
        off0 = __uint_as_float(pool[2 * POOL + sl]);
        off1 = __uint_as_float(pool[3 * POOL + sl]);
        off2 = __uint_as_float(pool[4 * POOL + sl]);
        const uint32_t gb = pool[5 * POOL + sl];
        gmax = __uint_as_float(gb & 0x7FFFFFFFu);
        fresh = (gb >> 31) != 0u;
        busy = bm >= nbd;  // the bound may have tightened since the subtree was kept
        if (!busy) prune_min = nbd < prune_min ? nbd : prune_min;
      }
    }
    {
      const uint32_t nw = (uint32_t)__popcll(wmask);
      count -= nw < count ? nw : count;
    }

    // One node per lane.
    bool push = false;
    uint32_t p_ref = 0;
    float p_nbd = 0.0f, p_off0 = 0.0f, p_off1 = 0.0f, p_off2 = 0.0f, p_gmax = 0.0f;
    bool hit[4] = {false, false, false, false};  // COLLECT: the points of this step at a distance <= D
    float hit_d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t hit_pos = 0;
    if (busy) {
      const bool is_leaf = !fresh && (ref & kLeafBit) != 0u;
      const uint32_t lv = ref & 0x7FFFFFFFu;
      const uint32_t begin = lv >> t.cbits;
      const uint32_t cnt = lv & t.cmask;
      const uint4* from = is_leaf ? reinterpret_cast<const uint4*>(pts + begin)
                                  : nodes + (fresh ? (ref & kRecIdxMask) : (ref & kBranchIdxMask));
      const uint4 w0 = *from;
      if (!is_leaf) {
        // A branch, or (fresh) the parent branch of a pending record whose far child is entered as traverse()
        // enters it: the same arithmetic with the side given instead of chosen, and nothing kept.
        const uint32_t axis = fresh ? (ref >> 28) & 3u : (ref >> 29) & 3u;
        const float left_max = __uint_as_float(w0.x);
        const float right_min = __uint_as_float(w0.y);
        const float v = sel3(axis, qx, qy, qz);
        const bool near_left = f_sub(f_sub(f_add(left_max, right_min), v), v) > 0.0f;
        const bool go_left = fresh ? (ref & kRecSide) != 0u : near_left;
        const float dv = f_sub(go_left ? right_min : left_max, v);
        const float new_off = M::one(dv);
        const uint32_t far_ref = go_left ? w0.w : w0.z;
        if (fresh) {
          off0 = axis == 0 ? new_off : off0;
          off1 = axis == 1 ? new_off : off1;
          off2 = axis == 2 ? new_off : off2;
          ref = far_ref;
        } else {
          const float far_nbd = f_add(f_sub(nbd, sel3(axis, off0, off1, off2)), new_off);
          if (bm >= far_nbd) {
            push = true;
            p_ref = far_ref;
            p_nbd = far_nbd;
            p_off0 = axis == 0 ? new_off : off0;
            p_off1 = axis == 1 ? new_off : off1;
            p_off2 = axis == 2 ? new_off : off2;
            p_gmax = gmax < far_nbd ? far_nbd : gmax;
          } else {
            prune_min = far_nbd < prune_min ? far_nbd : prune_min;
          }
          ref = go_left ? w0.z : w0.w;
        }
      } else {
        float4 p[4];
        p[0] = make_float4(__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w));
#pragma unroll
        for (int u = 1; u < 4; ++u) p[u] = pts[begin + u];
        hit_pos = begin;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if ((uint32_t)u < cnt) {
            PTK_KEEP4(p[u]);
            float dx = f_sub(qx, p[u].x);
            float dy = f_sub(qy, p[u].y);
            float dz = f_sub(qz, p[u].z);
            PTK_SCALAR(dx);
            PTK_SCALAR(dy);
            PTK_SCALAR(dz);
            const float d = point_distance3<M>(dx, dy, dz);
            if constexpr (COLLECT) {
              hit[u] = d <= fixed;
              hit_d[u] = d;
            } else {
              // Equal distances at the edge of this lane's list: a point turned away because it is exactly as far as
              // the k-th, or a k-th pushed out whose equal stays behind as the new k-th ([.., D, D] -> [.., c, .., D]).
              const float last = pol.ld[K - 1];
              if (d == last) tie_d = d;
              pol.visit(__float_as_int(p[u].w), d, gmax);
              if (d < last && pol.ld[K - 1] == last) tie_d = last;
              const float gone = d < last ? last : d;  // what left this lane's sight: the old k-th, or the point itself
              drop_min = gone < drop_min ? gone : drop_min;
            }
          }
        }
        if (cnt > 4u) {
          ref = kLeafBit | ((begin + 4u) << t.cbits) | (cnt - 4u);
        } else {
          busy = false;
        }
        if constexpr (!COLLECT) {
          if (pol.ld[K - 1] < best) lds_min_u32(gbest, __float_as_uint(pol.ld[K - 1]));
        }
      }
    }

    if constexpr (COLLECT) {  // the hits of this step go behind the entries (all lanes take part in the ballots)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint64_t hm = __ballot(hit[u]);
        if (hit[u]) {
          const uint32_t at = n_ent + (uint32_t)__popcll(hm & below);
          if (at < kKnnTieSlots) {
            ent[0 * kKnnTieSlots + at] = __float_as_uint(hit_d[u]);
            ent[1 * kKnnTieSlots + at] = kKnnPosFlag | (hit_pos + (uint32_t)u);
            ent[2 * kKnnTieSlots + at] = __float_as_uint(gmax);
          }
        }
        n_ent += (uint32_t)__popcll(hm);
      }
    }

    // Far children kept in this step go onto the pool.
    const uint64_t pmask = __ballot(push);
    if (push) {
      const uint32_t sl = count + (uint32_t)__popcll(pmask & below);
      if (sl < (uint32_t)POOL) {
        pool[0 * POOL + sl] = p_ref;
        pool[1 * POOL + sl] = __float_as_uint(p_nbd);
        pool[2 * POOL + sl] = __float_as_uint(p_off0);
        pool[3 * POOL + sl] = __float_as_uint(p_off1);
        pool[4 * POOL + sl] = __float_as_uint(p_off2);
        pool[5 * POOL + sl] = __float_as_uint(p_gmax);
      } else if (spill_n + (sl - (uint32_t)POOL) < spill_cap) {  // no room in LDS: parked in HBM
        Task tk;
        tk.ref = p_ref;
        tk.nbd = p_nbd;
        tk.off0 = p_off0;
        tk.off1 = p_off1;
        tk.off2 = p_off2;
        tk.gmax = p_gmax;
        spill_w[spill_n + (sl - (uint32_t)POOL)] = tk;
      }
    }
    count += (uint32_t)__popcll(pmask);
    if (count > (uint32_t)POOL) {
      spill_n += count - (uint32_t)POOL;
      count = (uint32_t)POOL;
      if (spill_n > spill_cap) {  // a subtree was lost: this query cannot be certified here
        ok = false;
        count = 0;
        spill_n = 0;
        busy = false;
      }
    }
    if (__ballot(busy) == 0ull && count == 0u && spill_n == 0u) break;
  }
  return ok;
}

// grid: any number of one-wavefront blocks; block b takes entries b, b + grid, ... of the hand-over list.  (Handing
// the entries out through a counter instead -- a free wavefront takes the next one -- was measured: every entry
// through the counter 0.1 ms slower (thousands of returning atomics on one word at the start), only the entries past
// the first grid through it no different; asking for 5 / 6 / 8 wavefronts per SIMD instead of the 4 its 118 VGPRs
// allow: no different / slower / spills.  profiles/r05_notes.txt item 11.)
// `redo_word`: the word of ho.meta that counts redo_list.  `ranges`: per branch, the record range of its subtree
// (dfs_before).  What the pool (LDS) has no room for is parked in the wavefront's run of `spill_cap` tasks of `spill`
// (HBM) and comes back when the pool has drained: 64 lanes keeping a far child each fill a pool of 128 in two steps
// while the bound is still wide, and a query that loses a subtree has to be searched again from the root by ONE lane
// -- the long searches this kernel exists for.
// M: metric_l2_squared or (r06) metric_l1 -- both box distances are sums of per-axis terms and lower bounds of the point
// distances below them, which is all the argument at the head of this file uses (see knn1_phase1u_kernel for why
// metric_lpinf / metric_lninf cannot come here).
template <int K, int POOL, class M = MetricL2>
__global__ __launch_bounds__(64) void knn_coop_kernel(
    DevTree t, const uint2* __restrict__ ranges, const float* __restrict__ queries, uint32_t dim, uint32_t k,
    Neighbor* __restrict__ out, Handover ho, uint32_t* __restrict__ redo_list, uint32_t redo_word,
    Task* __restrict__ spill, uint32_t spill_cap) {
  static_assert(POOL >= (int)kMaxTasks, "the pool must hold what a query starts with");
  PTK_TRACE_BEGIN_SEL(5);
  typedef PTK_LDS uint32_t LdsU32;
  Task* const spill_w = spill + (uint64_t)blockIdx.x * spill_cap;
  const uint32_t lane = threadIdx.x;
  LdsU32* pool = (LdsU32*)ptk_smem;  // [field][slot]
  LdsU32* gbest = pool + 6 * POOL;   // bits of the smallest k-th distance any lane holds
  constexpr uint32_t kRow = knn_coop_row((uint32_t)K);
  LdsU32* row = gbest + 1;           // the merged row: index [kRow], distance bits [kRow]
  LdsU32* ent = row + 2u * kRow;     // second sweep: distance bits, tag, box distance bits [kKnnTieSlots] each
  // (queries that found the list full were not listed: they finished in their lanes, Handover::full_keeps)
  const uint32_t n_heavy = ho.full_keeps != 0u && ho.meta[ho.counter] > ho.max_heavy ? ho.max_heavy : ho.meta[ho.counter];
  const float kInf = __uint_as_float(0x7F800000u);

  for (uint32_t entry = blockIdx.x; entry < n_heavy; entry += gridDim.x) {  // (uniform)
    const uint32_t qi = ho.heavy_list[entry];
    const uint32_t nt = ho.ntasks[entry];
    const Task* src = ho.tasks + (uint64_t)entry * kMaxTasks;
    float qx, qy, qz;
    load_query(queries, dim, qi, qx, qy, qz);
    // The list handed over: every lane starts with a copy (slot j of K holds entry j + k - K, as KnnRegPolicy keeps it).
    KnnCertPolicy<K> pol;
    pol.init(k);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if ((uint32_t)j + k >= (uint32_t)K) {
        const Neighbor nb = out[(uint64_t)qi * k + ((uint32_t)j + k - (uint32_t)K)];
        pol.ld[j] = nb.distance;
        pol.li[j] = nb.index;
      }
    }
    bool failed = nt == kTasksRedo || nt == kTasksFromRoot || nt > kMaxTasks;  // (uniform)
    float tie_d = -1.0f;  // the last distance this lane saw that was EQUAL to its k-th (a tie at the edge of its list)
    float drop_min = kInf, prune_min = kInf;  // the nearest point this lane let go of / box distance it pruned at
    uint32_t n_ent = 0;
    if (!failed)
      failed = !knn_coop_sweep<K, POOL, false, M>(t, qx, qy, qz, src, nt, pool, gbest, spill_w, spill_cap, pol, tie_d,
                                               drop_min, prune_min, 0.0f, ent, n_ent);

    // The k nearest of what the lanes hold (the handed-over entries are in every list that has not displaced them:
    // equal heads with one index are one point), k + 1 rounds: the last one looks at the runner-up.
    bool tie = false;
    float prev = -1.0f, g_all = 0.0f, runner_up = kInf;
    for (uint32_t r = 0; r <= k; ++r) {  // (uniform)
      float hd = kInf, hg = 0.0f;
      int32_t hi = 0;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if ((uint32_t)j + k == (uint32_t)K) {
          hd = pol.ld[j];
          hi = pol.li[j];
          hg = pol.lg[j];
        }
      }
      const float m = wave_min_f32(hd);
      if (m == prev) tie = true;  // two different points equally far (or nothing left: k-th == FLT_MAX twice)
      if (r == k) {
        runner_up = m;
        break;
      }
      const bool mine = hd == m;
      const uint64_t owners = __ballot(mine);
      const int first = (int)__builtin_ctzll(owners);
      const int32_t i0 = __shfl(hi, first);
      const float g0 = __shfl(hg, first);
      if (__ballot(mine && hi != i0) != 0ull) tie = true;
      g_all = g0 > g_all ? g0 : g_all;
      if (lane == 0) {
        row[r] = (uint32_t)i0;
        row[kRow + r] = __float_as_uint(m);
      }
      if (mine) {  // the head leaves this lane's list
#pragma unroll
        for (int j = 0; j < K - 1; ++j) {
          if ((uint32_t)j + k >= (uint32_t)K) {
            pol.ld[j] = pol.ld[j + 1];
            pol.li[j] = pol.li[j + 1];
            pol.lg[j] = pol.lg[j + 1];
          }
        }
        pol.ld[K - 1] = kInf;
        pol.li[K - 1] = 0;
      }
      prev = m;
    }
    // prev = D, the k-th distance.  A lane that turned a point away because it was exactly as far as its k-th then: if
    // that is the final k-th, the reference's visit order decides between them.
    if (__ballot(tie_d == prev) != 0ull) tie = true;
    const float dk = prev;
    const bool range = !(dk >= 1e-30f && dk <= 1e30f);  // (the error bounds assume no underflow or overflow; D = 0 is
                                                        // k points AT the query)
    // Certificate (b): no box distance on the way to one of the k nearest above the runner-up distance D', which is at
    // least: the nearest entry the lanes still hold, the nearest point a lane turned away or pushed out, the smallest
    // box distance a lane pruned at less its rounding (a float box distance exceeds the exact one by a relative 2^-12
    // at most and a float point distance is within five roundings of the exact one: knn1_coop_kernel) -- and D itself.
    float d_next = wave_min_f32(drop_min);
    {
      const float pr = wave_min_f32(prune_min);
      const float pr_low = f_sub(pr, f_mul(pr, 0.00048828125f));  // x (1 - 2^-11)
      d_next = pr_low < d_next ? pr_low : d_next;
      d_next = runner_up < d_next ? runner_up : d_next;
      d_next = d_next < dk ? dk : d_next;
    }
    bool box = !(g_all <= d_next);
    bool crowded = false;

    if (tie && !failed && !range) {
      // The second sweep: every point at a distance <= D, ranked.  First the handed-over entries (tag = their rank).
      const Neighbor seed = lane < k ? out[(uint64_t)qi * k + lane] : Neighbor{0, kInf};
      const bool in = lane < k && seed.distance <= dk;
      const uint64_t sm = __ballot(in);
      if (in) {
        const uint32_t at = (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
        ent[0 * kKnnTieSlots + at] = __float_as_uint(seed.distance);
        ent[1 * kKnnTieSlots + at] = lane;
        ent[2 * kKnnTieSlots + at] = 0u;
      }
      n_ent = (uint32_t)__popcll(sm);
      failed = !knn_coop_sweep<K, POOL, true, M>(t, qx, qy, qz, src, nt, pool, gbest, spill_w, spill_cap, pol, tie_d,
                                              drop_min, prune_min, dk, ent, n_ent);
      crowded = n_ent > kKnnTieSlots;
      if (lane == 0) atomicAdd(&ho.meta[kKnnTieSweeps], 1u);
      // (the ballots of the sweep's last step are behind every write of an entry)
      if (!failed && !crowded) {
        const bool have = lane < n_ent;
        const float d_i = have ? __uint_as_float(ent[0 * kKnnTieSlots + lane]) : kInf;
        const uint32_t tag_i = have ? ent[1 * kKnnTieSlots + lane] : 0u;
        const float g_i = have ? __uint_as_float(ent[2 * kKnnTieSlots + lane]) : 0.0f;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n_ent; ++j) {  // (uniform)
          const float d_j = __uint_as_float(ent[0 * kKnnTieSlots + j]);
          const uint32_t tag_j = ent[1 * kKnnTieSlots + j];
          bool first_j = d_j < d_i;
          if (have && j != lane && d_j == d_i) {  // as far: the handed-over entries in their order, then depth-first order
            const bool pos_i = (tag_i & kKnnPosFlag) != 0u, pos_j = (tag_j & kKnnPosFlag) != 0u;
            if (!pos_i || !pos_j) {
              first_j = !pos_j && (pos_i || tag_j < tag_i);
            } else {
              first_j = dfs_before(t, ranges, qx, qy, qz, tag_j & ~kKnnPosFlag, tag_i & ~kKnnPosFlag);
            }
          }
          rank += have && first_j ? 1u : 0u;
        }
        const bool chosen = have && rank < k;
        // Certificate of the second sweep -- (b) again, with the runner-up taken from the ranking: D2 = the distance
        // of the first entry NOT chosen (rank k), or dk when every entry is chosen (whatever was not collected lies
        // beyond dk).  Before the reference has reached a chosen point p its list holds k other points or fewer, so
        // its bound is at least the distance of the (k + 1)-th entry of the ranking: D2.  `dk` itself is NOT that
        // bound: it is the value of the merge's k-th ROUND, and a round that pops several different points of one
        // distance counts them once -- with ties dk lies above the true k-th distance (it is a sound bound to COLLECT
        // with, no more).  Comparing box distances with dk is the parity failure of profiles/r05_notes.txt item 24 (a
        // line of points: the reference turns away from a leaf whose box distance has drifted one ulp above its k-th
        // distance T; dk was 43 ulps above T); traced in profiles/r06_notes.txt item 1, tools/trace_box_distance.py.
        const float d2 = wave_min_f32(have && !chosen ? d_i : dk);
        box = __ballot(chosen && !(g_i <= d2)) != 0ull;
        // (every index is fetched before any row entry is written: the handed-over ones come from the row itself)
        int32_t idx_i = 0;
        if (chosen) {
          idx_i = (tag_i & kKnnPosFlag) != 0u ? __float_as_int(t.pts[tag_i & ~kKnnPosFlag].w)
                                              : out[(uint64_t)qi * k + tag_i].index;
        }
        if (__ballot(chosen) != 0ull && !box && chosen) {
          Neighbor nb;
          nb.index = idx_i;
          nb.distance = d_i;
          out[(uint64_t)qi * k + rank] = nb;
        }
      }
    } else if (!failed && !range && !box) {
      // (the ballot is behind lane 0's writes of the row)
      if (__ballot(lane < k) != 0ull && lane < k) {
        Neighbor nb;
        nb.index = (int32_t)row[lane];
        nb.distance = __uint_as_float(row[kRow + lane]);
        out[(uint64_t)qi * k + lane] = nb;
      }
    }
    if ((failed || crowded || box || range) && lane == 0) {
      redo_list[atomicAdd(&ho.meta[redo_word], 1u)] = qi;
      atomicAdd(&ho.meta[failed ? kKnnWhyPool : (crowded ? kKnnWhyTie : (box ? kKnnWhyBox : kKnnWhyRange))], 1u);
    }
  }
  PTK_TRACE_END_ANY();
}

// The reference search from the root for the queries the cooperative search could not certify (rows as knn_reg_kernel
// stores them).
template <int K, int S, int OVF, int LEAFB, class M = MetricL2>
__global__ __launch_bounds__(64) void knn_redo_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim, uint32_t k, float e_inv, Neighbor* __restrict__ out,
    const uint32_t* __restrict__ meta, uint32_t redo_word, const uint32_t* __restrict__ redo_list) {
  const uint32_t n = meta[redo_word];
  Record spill[OVF > 0 ? OVF : 1];
  for (uint32_t i = blockIdx.x * 64u + threadIdx.x; i < n; i += gridDim.x * 64u) {
    const uint32_t qi = redo_list[i];
    float qx, qy, qz;
    load_query(queries, dim, qi, qx, qy, qz);
    pad_query<M>(dim, qy, qz);
    Stack<S, OVF, 64> st;
    st.init((LdsWord*)ptk_smem, threadIdx.x, spill);
    KnnRegPolicy<K> pol;
    pol.init(k, e_inv);
    traverse<LEAFB, false, M>(t, qx, qy, qz, pol, st);
    pol.store(out + (uint64_t)qi * k);
  }
}

}  // namespace ptk
