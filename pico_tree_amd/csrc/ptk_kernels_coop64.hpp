// ptk_kernels_coop64.hpp -- double precision, dim <= 3: the long searches of a k-NN batch (k <= 32; the radius search: below) finished by a
// whole wavefront each.  ptk_kernels_coopk.hpp in double; the argument for why the merged row is the reference's is
// at the head of that file and is not repeated here -- it uses nothing of the scalar type but
//   * the box distance of a node is a function of its root path alone (`nbd - old_offset + new_offset`,
//     kd_tree_search.hpp:91-99): Task64 carries {nbd, off[3]} down every path exactly as traverse64_3 does;
//   * a subtree is pruned only when its box distance exceeds bound * (1 + 2^-10) -- the margin derived for FLOAT
//     roundings (a float box distance is within a relative 2^-12 of the exact one, a float point distance within five
//     roundings); a double rounds 2^29 times finer, so the same margin holds a fortiori;
//   * strict comparisons keep equal distances in visit order (search_visitor.hpp:30,107), which dfs_before64() decides
//     for two record positions.
//
// Why it exists (tools/time_f64_sizes.py, profiles/r06_notes.txt item 12): the double kernels ran every query to its
// end in its lane, and on BASELINE config 2's scan ANY batch then takes as long as the longest search of the cloud --
// 150 000 queries 2.9 ms at knn = 1 and 4.7 ms at knn = 16, where the float32 searches (capped since r02 / r05) take
// 0.2 and 0.45 ms.
//
// k = 1 goes the same way (a list of one): there is no two-phase search in double.
#pragma once

#include "ptk_kernels_f64.hpp"

namespace ptk {

// KnnCertPolicy (ptk_kernels_coopk.hpp) over doubles: the k entries live in the LAST k of the K slots (the rest hold
// -inf and never move); `lg` = the largest box distance of a far child on the way to the point (0 for the entries
// handed over: the reference has visited those).
template <int K>
struct KnnCert64Policy {
  double ld[K];
  int32_t li[K];
  double lg[K];
  __device__ __forceinline__ void init(uint32_t k) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      ld[j] = (uint32_t)j + k >= (uint32_t)K ? kDblMax : __longlong_as_double((long long)0xFFF0000000000000ull);
      li[j] = 0;
      lg[j] = 0.0;
    }
  }
  __device__ __forceinline__ void visit(int32_t idx, double d, double g) {
    if (ld[K - 1] > d) {
      bool below[K];
#pragma unroll
      for (int j = 0; j < K; ++j) below[j] = d < ld[j];
#pragma unroll
      for (int j = K - 1; j >= 1; --j) {
        li[j] = below[j - 1] ? li[j - 1] : (below[j] ? idx : li[j]);
        lg[j] = below[j - 1] ? lg[j - 1] : (below[j] ? g : lg[j]);
        ld[j] = below[j - 1] ? ld[j - 1] : (below[j] ? d : ld[j]);
      }
      li[0] = below[0] ? idx : li[0];
      lg[0] = below[0] ? g : lg[0];
      ld[0] = below[0] ? d : ld[0];
    }
  }
};

__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const double o = __shfl_xor(v, d);
    v = o < v ? o : v;
  }
  return v;
}

// min() on a 64-bit LDS word shared by the lanes of a wavefront (ds_min_u64): the bits of a non-negative double order
// as the double does.
#if defined(__HIPCC__)
__device__ __forceinline__ void lds_min_u64(PTK_LDS unsigned long long* p, unsigned long long v) {
  __hip_atomic_fetch_min((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#else
inline void lds_min_u64(unsigned long long* p, unsigned long long v) {
  if (v < *p) *p = v;
}
#endif

// dfs_before() of ptk_kernels.hpp on the double tree: true if, in the reference's depth-first order for query q, the
// point at record position `a` comes before the one at `b` (a != b).
__device__ inline bool dfs_before64(const DevTree64& t, double q0, double q1, double q2, uint32_t a, uint32_t b) {
  uint32_t ref = t.root_ref;
  for (;;) {
    if (ref & kLeafBit) return a < b;
    const Node64 nd = t.nodes[ref];
    uint32_t mid;  // first record of the right child
    if (nd.right_ref & kLeafBit) {
      mid = (nd.right_ref & 0x7FFFFFFFu) >> t.cbits;
    } else {
      mid = t.ranges[nd.right_ref].x;
    }
    const bool a_left = a < mid, b_left = b < mid;
    if (a_left == b_left) {
      ref = a_left ? nd.left_ref : nd.right_ref;
      continue;
    }
    const double v = sel3d(nd.axis, q0, q1, q2);
    const bool go_left = d_sub(d_sub(d_add(nd.left_max, nd.right_min), v), v) > 0.0;
    return go_left == a_left;
  }
}

constexpr uint32_t kKnn64TieSlots = 64;           // points at a distance <= D the second sweep can rank
constexpr uint32_t kKnn64PosFlag = 0x80000000u;   // second sweep: the entry is a record position (else: its rank in the handed-over list)
constexpr uint32_t kKnn64Row = 32;                // entries of the merged row (k <= 32)

// LDS of a wavefront, doubles first: pool {nbd, off0, off1, off2, gmax}[POOL], the shared bound, the merged row's
// distances [kKnn64Row], the second sweep's {distance, box distance}[kKnn64TieSlots]; then 32-bit words: pool refs
// [POOL], the merged row's indices [kKnn64Row], the second sweep's tags [kKnn64TieSlots].
constexpr uint32_t knn64_coop_lds_doubles(uint32_t pool) { return 5u * pool + 1u + kKnn64Row + 2u * kKnn64TieSlots; }
constexpr uint32_t knn64_coop_lds_bytes(uint32_t pool) {
  return knn64_coop_lds_doubles(pool) * 8u + (pool + kKnn64Row + kKnn64TieSlots) * 4u;
}

typedef PTK_LDS double LdsF64;
typedef PTK_LDS unsigned long long LdsU64;

template <int POOL>
struct Pool64 {
  LdsF64* f;    // [5][POOL]
  LdsU32* ref;  // [POOL]
  __device__ __forceinline__ void put(uint32_t sl, const Task64& tk) {
    ref[sl] = tk.ref;
    f[0 * POOL + sl] = tk.nbd;
    f[1 * POOL + sl] = tk.off0;
    f[2 * POOL + sl] = tk.off1;
    f[3 * POOL + sl] = tk.off2;
    f[4 * POOL + sl] = tk.gmax;
  }
};

// One sweep of a query's pending subtrees by the 64 lanes of the wavefront: knn_coop_sweep of ptk_kernels_coopk.hpp.
// U points of a leaf per step.
template <int K, int POOL, bool COLLECT, class M, int U>
__device__ __forceinline__ bool knn64_coop_sweep(
    const DevTree64& t, double q0, double q1, double q2, const Task64* __restrict__ src, uint32_t nt, Pool64<POOL> pool,
    LdsU64* gbest, Task64* __restrict__ spill_w, uint32_t spill_cap, KnnCert64Policy<K>& pol, double& tie_d,
    double& drop_min, double& prune_min, double fixed, LdsF64* ent_d, LdsU32* ent_tag, LdsF64* ent_g, uint32_t& n_ent) {
  const Node64* __restrict__ nodes = t.nodes;
  const double* __restrict__ pts = t.pts;
  const uint32_t last = t.n_points - 1;
  const uint32_t lane = threadIdx.x;
  const uint64_t below = (1ull << lane) - 1ull;
  bool ok = true;
  uint32_t count = nt;  // subtrees in the pool (uniform)
  for (uint32_t i = lane; i < nt; i += 64u) pool.put(nt - 1u - i, src[i]);  // the capped traversal's stack, next-to-visit on top
  if (lane == 0) *gbest = (unsigned long long)__double_as_longlong(COLLECT ? fixed : pol.ld[K - 1]);
  bool busy = false;
  uint32_t ref = 0;
  uint32_t spill_n = 0;  // tasks parked in HBM (uniform)
  double nbd = 0.0, off0 = 0.0, off1 = 0.0, off2 = 0.0, gmax = 0.0;

  for (;;) {
    // A drained pool takes back what had to be parked in HBM (the newest first, up to half a pool).
    if (count == 0u && spill_n != 0u) {  // (uniform)
      const uint32_t m = spill_n < (uint32_t)(POOL / 2) ? spill_n : (uint32_t)(POOL / 2);
      for (uint32_t i = lane; i < m; i += 64u) pool.put(i, spill_w[spill_n - m + i]);
      count = m;
      spill_n -= m;
    }
    // (the ballot is also where the lanes meet after the pool and the bound were written)
    const bool want = !busy;
    const uint64_t wmask = __ballot(want);
    const double best = __longlong_as_double((long long)*gbest);
    const double bm = d_add(best, d_mul(best, 0.0009765625));  // bound * (1 + 2^-10): see the head of this file
    bool fresh = false;  // taken in the handed-over form: the parent branch has to be read first
    if (want) {
      const uint32_t rank = (uint32_t)__popcll(wmask & below);
      if (rank < count) {
        const uint32_t sl = count - 1u - rank;
        ref = pool.ref[sl];
        nbd = pool.f[0 * POOL + sl];
        off0 = pool.f[1 * POOL + sl];
        off1 = pool.f[2 * POOL + sl];
        off2 = pool.f[3 * POOL + sl];
        const long long gb = __double_as_longlong(pool.f[4 * POOL + sl]);
        gmax = __longlong_as_double(gb & 0x7FFFFFFFFFFFFFFFll);
        fresh = gb < 0;
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
    Task64 pt{};
    bool hit[U];
    double hit_d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      hit[u] = false;
      hit_d[u] = 0.0;
    }
    uint32_t hit_pos = 0;
    if (busy) {
      const bool is_leaf = !fresh && (ref & kLeafBit) != 0u;
      if (!is_leaf) {
        // A branch, or (fresh) the parent branch of a pending record whose far child is entered as traverse64_3 enters
        // it: the same arithmetic with the side given instead of chosen, and nothing kept.
        const Node64 nd = nodes[fresh ? (ref & 0x3FFFFFFFu) : ref];
        const double v = sel3d(nd.axis, q0, q1, q2);
        const bool near_left = d_sub(d_sub(d_add(nd.left_max, nd.right_min), v), v) > 0.0;
        const bool go_left = fresh ? (ref & kRecSide) != 0u : near_left;  // (pending: kRecSide = the far child is the right one)
        const double dv = d_sub(go_left ? nd.right_min : nd.left_max, v);
        const double new_off = M::one(dv);
        const uint32_t far_ref = go_left ? nd.right_ref : nd.left_ref;
        if (fresh) {
          off0 = nd.axis == 0 ? new_off : off0;
          off1 = nd.axis == 1 ? new_off : off1;
          off2 = nd.axis == 2 ? new_off : off2;
          ref = far_ref;
        } else {
          const double far_nbd = d_add(d_sub(nbd, sel3d(nd.axis, off0, off1, off2)), new_off);
          if (bm >= far_nbd) {
            push = true;
            pt.ref = far_ref;
            pt.nbd = far_nbd;
            pt.off0 = nd.axis == 0 ? new_off : off0;
            pt.off1 = nd.axis == 1 ? new_off : off1;
            pt.off2 = nd.axis == 2 ? new_off : off2;
            pt.gmax = gmax < far_nbd ? far_nbd : gmax;
          } else {
            prune_min = far_nbd < prune_min ? far_nbd : prune_min;
          }
          ref = go_left ? nd.left_ref : nd.right_ref;
        }
      } else {
        const uint32_t lv = ref & 0x7FFFFFFFu;
        const uint32_t begin = lv >> t.cbits;
        const uint32_t cnt = lv & t.cmask;
        double px[U], py[U], pz[U];
        int32_t pi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {  // every load of the round before the first use
          const uint32_t pu = begin + u <= last ? begin + u : last;  // in range past the leaf's end too
          const double4 a = *reinterpret_cast<const double4*>(pts + (uint64_t)pu * kStride64D3);  // {x, y, z, index}
          px[u] = a.x;
          py[u] = a.y;
          pz[u] = a.z;
          pi[u] = (int32_t)__double_as_longlong(a.w);
        }
        hit_pos = begin;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if ((uint32_t)u < cnt) {
            const double d = M::acc(M::acc(M::one(d_sub(q0, px[u])), d_sub(q1, py[u])), d_sub(q2, pz[u]));
            if constexpr (COLLECT) {
              hit[u] = d <= fixed;
              hit_d[u] = d;
            } else {
              // Equal distances at the edge of this lane's list: a point turned away because it is exactly as far as
              // the k-th, or a k-th pushed out whose equal stays behind as the new k-th ([.., D, D] -> [.., c, .., D]).
              const double lastd = pol.ld[K - 1];
              if (d == lastd) tie_d = d;
              pol.visit(pi[u], d, gmax);
              if (d < lastd && pol.ld[K - 1] == lastd) tie_d = lastd;
              const double gone = d < lastd ? lastd : d;  // what left this lane's sight: the old k-th, or the point itself
              drop_min = gone < drop_min ? gone : drop_min;
            }
          }
        }
        if (cnt > (uint32_t)U) {
          ref = kLeafBit | ((begin + (uint32_t)U) << t.cbits) | (cnt - (uint32_t)U);
        } else {
          busy = false;
        }
        if constexpr (!COLLECT) {
          if (pol.ld[K - 1] < best) lds_min_u64(gbest, (unsigned long long)__double_as_longlong(pol.ld[K - 1]));
        }
      }
    }

    if constexpr (COLLECT) {  // the hits of this step go behind the entries (all lanes take part in the ballots)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t hm = __ballot(hit[u]);
        if (hit[u]) {
          const uint32_t at = n_ent + (uint32_t)__popcll(hm & below);
          if (at < kKnn64TieSlots) {
            ent_d[at] = hit_d[u];
            ent_tag[at] = kKnn64PosFlag | (hit_pos + (uint32_t)u);
            ent_g[at] = gmax;
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
        pool.put(sl, pt);
      } else if (spill_n + (sl - (uint32_t)POOL) < spill_cap) {  // no room in LDS: parked in HBM
        spill_w[spill_n + (sl - (uint32_t)POOL)] = pt;
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

// Before the capped launch: the counters zeroed, the description of the hand-over list where the kernels read it.
PTK_GLOBAL __launch_bounds__(64) void knn64_handover_init_kernel(Handover64 ho, Handover64* __restrict__ at) {
  if (threadIdx.x < kMetaWords) ho.meta[threadIdx.x] = 0u;
  if (threadIdx.x == 0) *at = ho;
}

// The capped launch: one query per lane, the reference traversal with the k-list in registers (Knn64RegPolicy<K>) --
// the row is stored either way; a query that stops leaves its list there for the cooperative search to start from.
template <class M, int K>
__global__ __launch_bounds__(64) void knn64_capped_kernel(
    DevTree64 t, const double* __restrict__ queries, const uint32_t* __restrict__ perm, uint64_t q0, uint64_t nq,
    uint32_t k, Neighbor64* __restrict__ out, Rec64* __restrict__ stack, uint32_t slots, uint32_t cap,
    const Handover64* __restrict__ ho) {
  const uint64_t i = (uint64_t)xcd_runs(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[q0 + i] : q0 + i;
  Knn64RegPolicy<K> pol;
  pol.init(k, 1.0);
  const double* row = queries + qi * t.dim;
  const double x0 = row[0];
  const double x1 = t.dim > 1 ? row[1] : metric64_pad<M>();
  const double x2 = t.dim > 2 ? row[2] : metric64_pad<M>();
  Stack64 st;
  st.init(0, 0, stack, slots);
  Trav64State ts{};
  bool resume = false;
  for (;;) {
    if (traverse64_3<M, Knn64RegPolicy<K>, true>(t, x0, x1, x2, pol, st, cap, &ts, resume)) break;
    // (the list's description is read here, where a query stops, and not held in scalar registers through the
    // traversal: as a kernel argument by value its ten words were -- knn64_capped_kernel<16> 103 scalar registers against
    // the uncapped kernel's 74, the k-list's compare masks competing for them, and 6-9 % of its time)
    const uint32_t h = atomicAdd(&ho->meta[ho->counter], 1u);
    if (h < ho->max_heavy) {
      hand_over64(ho, (uint32_t)qi, h, ts, st, pol.max());
      break;
    }
    cap = 0xFFFFFFFFu;  // no room in the list: this lane finishes its query itself
    resume = true;
  }
  pol.store(out + qi * k);
}

// grid: any number of one-wavefront blocks; block b takes entries b, b + grid, ... of the hand-over list
// (knn_coop_kernel of ptk_kernels_coopk.hpp; the comments on the merge and the certificates are there).
template <int K, int POOL, class M>
__global__ __launch_bounds__(64) void knn64_coop_kernel(
    DevTree64 t, const double* __restrict__ queries, uint32_t k, Neighbor64* __restrict__ out, Handover64 ho,
    uint32_t* __restrict__ redo_list, uint32_t redo_word, Task64* __restrict__ spill, uint32_t spill_cap) {
  static_assert(POOL >= (int)kMaxTasks, "the pool must hold what a query starts with");
  static_assert(K <= (int)kKnn64Row, "the merged row");
  constexpr int U = K >= 32 ? 2 : 4;
  Task64* const spill_w = spill + (uint64_t)blockIdx.x * spill_cap;
  const uint32_t lane = threadIdx.x;
  LdsF64* fd = (LdsF64*)ptk_smem;
  Pool64<POOL> pool;
  pool.f = fd;
  LdsU64* gbest = (LdsU64*)(fd + 5 * POOL);
  LdsF64* row_d = fd + 5 * POOL + 1;
  LdsF64* ent_d = row_d + kKnn64Row;
  LdsF64* ent_g = ent_d + kKnn64TieSlots;
  LdsU32* wd = (LdsU32*)(ent_g + kKnn64TieSlots);
  pool.ref = wd;
  LdsU32* row_i = wd + POOL;
  LdsU32* ent_tag = row_i + kKnn64Row;
  const uint32_t listed = ho.meta[ho.counter];
  const uint32_t n_heavy = listed > ho.max_heavy ? ho.max_heavy : listed;  // (queries that found the list full finished in their lanes)
  const double kInf = __longlong_as_double(0x7FF0000000000000ll);

  for (uint32_t entry = blockIdx.x; entry < n_heavy; entry += gridDim.x) {  // (uniform)
    const uint32_t qi = ho.heavy_list[entry];
    const uint32_t nt = ho.ntasks[entry];
    const Task64* src = ho.tasks + (uint64_t)entry * kMaxTasks;
    const double* qrow = queries + (uint64_t)qi * t.dim;
    const double q0 = qrow[0];
    const double q1 = t.dim > 1 ? qrow[1] : metric64_pad<M>();
    const double q2 = t.dim > 2 ? qrow[2] : metric64_pad<M>();
    // The list handed over: every lane starts with a copy (slot j of K holds entry j + k - K).
    KnnCert64Policy<K> pol;
    pol.init(k);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if ((uint32_t)j + k >= (uint32_t)K) {
        const Neighbor64 nb = out[(uint64_t)qi * k + ((uint32_t)j + k - (uint32_t)K)];
        pol.ld[j] = nb.distance;
        pol.li[j] = nb.index;
      }
    }
    bool failed = nt == kTasksRedo || nt == kTasksFromRoot || nt > kMaxTasks;  // (uniform)
    double tie_d = -1.0;  // the last distance this lane saw that was EQUAL to its k-th (a tie at the edge of its list)
    double drop_min = kInf, prune_min = kInf;  // the nearest point this lane let go of / box distance it pruned at
    uint32_t n_ent = 0;
    if (!failed)
      failed = !knn64_coop_sweep<K, POOL, false, M, U>(t, q0, q1, q2, src, nt, pool, gbest, spill_w, spill_cap, pol, tie_d,
                                                       drop_min, prune_min, 0.0, ent_d, ent_tag, ent_g, n_ent);

    // The k nearest of what the lanes hold (the handed-over entries are in every list that has not displaced them:
    // equal heads with one index are one point), k + 1 rounds: the last one looks at the runner-up.
    bool tie = false;
    double prev = -1.0, g_all = 0.0, runner_up = kInf;
    for (uint32_t r = 0; r <= k; ++r) {  // (uniform)
      double hd = kInf, hg = 0.0;
      int32_t hi = 0;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if ((uint32_t)j + k == (uint32_t)K) {
          hd = pol.ld[j];
          hi = pol.li[j];
          hg = pol.lg[j];
        }
      }
      const double m = wave_min_f64(hd);
      if (m == prev) tie = true;  // two different points equally far (or nothing left: k-th == DBL_MAX twice)
      if (r == k) {
        runner_up = m;
        break;
      }
      const bool mine = hd == m;
      const uint64_t owners = __ballot(mine);
      const int first = (int)__builtin_ctzll(owners);
      const int32_t i0 = __shfl(hi, first);
      const double g0 = __shfl(hg, first);
      if (__ballot(mine && hi != i0) != 0ull) tie = true;
      g_all = g0 > g_all ? g0 : g_all;
      if (lane == 0) {
        row_i[r] = (uint32_t)i0;
        row_d[r] = m;
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
    const double dk = prev;
    const bool range = !(dk >= 1e-280 && dk <= 1e280);  // (the error bounds assume no underflow or overflow; D = 0 is k
                                                        // points AT the query)
    // Certificate (b): no box distance on the way to one of the k nearest above the runner-up distance D'
    // (ptk_kernels_coopk.hpp).
    double d_next = wave_min_f64(drop_min);
    {
      const double pr = wave_min_f64(prune_min);
      const double pr_low = d_sub(pr, d_mul(pr, 0.00048828125));  // x (1 - 2^-11)
      d_next = pr_low < d_next ? pr_low : d_next;
      d_next = runner_up < d_next ? runner_up : d_next;
      d_next = d_next < dk ? dk : d_next;
    }
    bool box = !(g_all <= d_next);
    bool crowded = false;

    if (tie && !failed && !range) {
      // The second sweep: every point at a distance <= D, ranked.  First the handed-over entries (tag = their rank).
      Neighbor64 seed;
      seed.index = 0;
      seed.pad_ = 0;
      seed.distance = kInf;
      if (lane < k) seed = out[(uint64_t)qi * k + lane];
      const bool in = lane < k && seed.distance <= dk;
      const uint64_t sm = __ballot(in);
      if (in) {
        const uint32_t at = (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
        ent_d[at] = seed.distance;
        ent_tag[at] = lane;
        ent_g[at] = 0.0;
      }
      n_ent = (uint32_t)__popcll(sm);
      failed = !knn64_coop_sweep<K, POOL, true, M, U>(t, q0, q1, q2, src, nt, pool, gbest, spill_w, spill_cap, pol, tie_d,
                                                      drop_min, prune_min, dk, ent_d, ent_tag, ent_g, n_ent);
      crowded = n_ent > kKnn64TieSlots;
      if (lane == 0) atomicAdd(&ho.meta[kKnnTieSweeps], 1u);
      // (the ballots of the sweep's last step are behind every write of an entry)
      if (!failed && !crowded) {
        const bool have = lane < n_ent;
        const double d_i = have ? ent_d[lane] : kInf;
        const uint32_t tag_i = have ? ent_tag[lane] : 0u;
        const double g_i = have ? ent_g[lane] : 0.0;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n_ent; ++j) {  // (uniform)
          const double d_j = ent_d[j];
          const uint32_t tag_j = ent_tag[j];
          bool first_j = d_j < d_i;
          if (have && j != lane && d_j == d_i) {  // as far: the handed-over entries in their order, then depth-first order
            const bool pos_i = (tag_i & kKnn64PosFlag) != 0u, pos_j = (tag_j & kKnn64PosFlag) != 0u;
            if (!pos_i || !pos_j) {
              first_j = !pos_j && (pos_i || tag_j < tag_i);
            } else {
              first_j = dfs_before64(t, q0, q1, q2, tag_j & ~kKnn64PosFlag, tag_i & ~kKnn64PosFlag);
            }
          }
          rank += have && first_j ? 1u : 0u;
        }
        const bool chosen = have && rank < k;
        // (the certificate of the second sweep: D2 = the distance of the first entry NOT chosen, or dk when every entry
        // is chosen -- ptk_kernels_coopk.hpp)
        const double d2 = wave_min_f64(have && !chosen ? d_i : dk);
        box = __ballot(chosen && !(g_i <= d2)) != 0ull;
        // (every index is fetched before any row entry is written: the handed-over ones come from the row itself)
        int32_t idx_i = 0;
        if (chosen) {
          idx_i = (tag_i & kKnn64PosFlag) != 0u ? t.index[tag_i & ~kKnn64PosFlag] : out[(uint64_t)qi * k + tag_i].index;
        }
        if (__ballot(chosen) != 0ull && !box && chosen) {
          Neighbor64 nb;
          nb.index = idx_i;
          nb.pad_ = 0;
          nb.distance = d_i;
          out[(uint64_t)qi * k + rank] = nb;
        }
      }
    } else if (!failed && !range && !box) {
      // (the ballot is behind lane 0's writes of the row)
      if (__ballot(lane < k) != 0ull && lane < k) {
        Neighbor64 nb;
        nb.index = (int32_t)row_i[lane];
        nb.pad_ = 0;
        nb.distance = row_d[lane];
        out[(uint64_t)qi * k + lane] = nb;
      }
    }
    if ((failed || crowded || box || range) && lane == 0) {
      redo_list[atomicAdd(&ho.meta[redo_word], 1u)] = qi;
      atomicAdd(&ho.meta[failed ? kKnnWhyPool : (crowded ? kKnnWhyTie : (box ? kKnnWhyBox : kKnnWhyRange))], 1u);
    }
  }
}

// The reference search from the root for the queries the cooperative search could not certify.  Block b owns stack
// columns [b * slots * 64, ...) of the launch's block, as every double kernel does.
template <class M, int K>
__global__ __launch_bounds__(64) void knn64_redo_kernel(
    DevTree64 t, const double* __restrict__ queries, uint32_t k, Neighbor64* __restrict__ out,
    const uint32_t* __restrict__ meta, uint32_t redo_word, const uint32_t* __restrict__ redo_list, Rec64* __restrict__ stack,
    uint32_t slots) {
  const uint32_t n = meta[redo_word];
  for (uint32_t i = blockIdx.x * 64u + threadIdx.x; i < n; i += gridDim.x * 64u) {
    const uint32_t qi = redo_list[i];
    Knn64RegPolicy<K> pol;
    pol.init(k, 1.0);
    search64<M, true>(t, queries, qi, pol, stack, slots);
    pol.store(out + (uint64_t)qi * k);
  }
}


// ---- the radius search: the long queries of a double batch finished by a wavefront each ------------------------------
// ptk_kernels_coopr.hpp in double.  The radius visitor never changes its bound (search_visitor.hpp:127-156), so the SET
// of leaves the reference visits below a pending subtree is decided node by node whatever the order the nodes are looked
// at in (a node's box distance is a function of its root path alone), and only the ORDER of the leaves in the row is the
// traversal's: every subtree in the shared pool carries a KEY -- the handed-over task it belongs to, then one bit per
// branch below it, 0 for the child the reference enters first -- the leaves with hits are listed as {key, leaf entry} in
// whatever order the lanes meet them, and a bitonic sort of the wavefront's entries (at most 512, in LDS) restores the
// reference's visit order.  No certificate is needed, so every metric of the double kernels takes this path.
//
//   count pass   radius64_capped_kernel<M, false>: a query that has entered more than `cap` far children stops, its count
//                so far stays in counts[row], its stack goes to the hand-over list, flag[row] = 1;
//                radius64_coop_count_kernel: a wavefront per such query, the sorted entries to `entries`, the total to
//                counts[row]; what it cannot finish (more than 512 leaves with hits, pool and spill full, the entry block
//                exhausted) is counted again from the root by one lane (radius64_redo_kernel<M, false>).
//   fill pass    radius64_capped_kernel<M, true>: the same traversal stops at the same place (flag[row]) having written
//                the row's first hits; radius64_coop_replay_kernel writes the rest from the sorted entries -- same
//                arithmetic as the leaf scan (bit-identical distances); rows whose entries were lost are filled again
//                from the root by one lane (radius64_redo_kernel<M, true>).
constexpr uint32_t kRc64MaxEntries = 512;    // leaves with hits of one query the sort holds (LDS: 8 KB)
constexpr uint32_t kRc64Lost = 0xFFFFFFFFu;  // RadiusHeavy64::run_n: this query's entries are not here (searched again)
constexpr uint32_t kRc64KeyTop = 57;         // the first path bit of a key (bits 63:58 = the task)
constexpr uint32_t kRc64KeyLow = 6;          // bits 5:0 = the piece of a large leaf
constexpr uint32_t kRc64MaskBits = 32;       // points per entry (a larger leaf is listed in pieces)
constexpr uint32_t kRc64MaxDepth = kRc64KeyTop - kRc64KeyLow;  // branches below a task a key has room for
constexpr uint32_t kMetaRc64Entries = 28;    // word of the counters block: entries handed out of RadiusHeavy64::entries
constexpr uint32_t kMetaRc64Over = 29;       // ... rows the fill pass must fill again from the root

// What a call keeps of its handed-over queries from the count pass to the fill pass.
struct RadiusHeavy64 {
  uint32_t* meta;      // the counters block (kMetaHeavy, kMetaRedo, kMetaRc64Entries, kMetaRc64Over)
  uint32_t* rows;      // [max_heavy] row of hand-over h
  uint32_t* own;       // [max_heavy] hits its lane had found before it stopped
  uint32_t* run_at;    // [max_heavy] where its sorted entries begin in `entries`
  uint32_t* run_n;     // [max_heavy] how many (kRc64Lost: none)
  unsigned long long* entries;  // [entry_cap] {(first point << cbits) | points, mask of the hits}
  uint32_t max_heavy;
  uint32_t entry_cap;
};

// LDS of a wavefront of radius64_coop_count_kernel: doubles {nbd, off0, off1, off2}[POOL], 64-bit words: pool keys
// [POOL], entry keys and values [kRc64MaxEntries] each; 32-bit words: pool refs and key bits [POOL] each.
constexpr uint32_t radius64_coop_lds_bytes(uint32_t pool) { return pool * (4u * 8u + 8u + 2u * 4u) + kRc64MaxEntries * 16u; }

__device__ __forceinline__ uint32_t wave64_sum_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
  return v;
}
// Inclusive prefix sum over the lanes.
__device__ __forceinline__ uint32_t wave64_scan_u32(uint32_t v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)v, d);
    if (lane >= (uint32_t)d) v += o;
  }
  return v;
}

// The count pass (FILL = false) and the fill pass (FILL = true) of a capped call: one query per lane.
template <class M, bool FILL>
__global__ __launch_bounds__(64) void radius64_capped_kernel(
    DevTree64 t, const double* __restrict__ queries, const uint32_t* __restrict__ perm, uint64_t q0, uint64_t nq,
    double radius, double e_inv, uint64_t* __restrict__ counts, const uint64_t* __restrict__ offsets,
    Neighbor64* __restrict__ out, Rec64* __restrict__ stack, uint32_t slots, uint32_t cap,
    const Handover64* __restrict__ ho, uint8_t* __restrict__ flag) {
  const uint64_t i = (uint64_t)xcd_runs(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[q0 + i] : q0 + i;
  Radius64Policy<FILL> pol;
  pol.radius = d_mul(radius, e_inv);  // search_visitor.hpp:265
  pol.e_inv = e_inv;
  pol.count = 0;
  pol.out = FILL ? out + offsets[qi] : nullptr;
  const double* row = queries + qi * t.dim;
  const double x0 = row[0];
  const double x1 = t.dim > 1 ? row[1] : metric64_pad<M>();
  const double x2 = t.dim > 2 ? row[2] : metric64_pad<M>();
  Stack64 st;
  st.init(0, 0, stack, slots);
  Trav64State ts{};
  bool resume = false;
  for (;;) {
    if (traverse64_3<M, Radius64Policy<FILL>, true>(t, x0, x1, x2, pol, st, cap, &ts, resume)) break;
    if constexpr (FILL) {
      if (flag[qi] != 0) return;  // handed over by the count pass: a wavefront writes the rest of the row
    } else {
      const uint32_t h = atomicAdd(&ho->meta[ho->counter], 1u);
      if (h < ho->max_heavy) {
        hand_over64(ho, (uint32_t)qi, h, ts, st, pol.max());
        flag[qi] = 1;
        break;
      }
    }
    cap = 0xFFFFFFFFu;  // (no room in the list: this lane finishes its query itself, in both passes)
    resume = true;
  }
  if constexpr (!FILL) counts[qi] = pol.count;
}

// grid: any number of one-wavefront blocks; block b takes entries b, b + grid, ... of the hand-over list.
// counts[row] holds what the capped lane had counted; the total replaces it.  Rows that could not be finished go to
// redo_list (counted at meta[kMetaRedo]).
template <int POOL, class M>
__global__ __launch_bounds__(64) void radius64_coop_count_kernel(
    DevTree64 t, const double* __restrict__ queries, double radius, double e_inv, uint64_t* __restrict__ counts,
    Handover64 ho, RadiusHeavy64 hv, uint32_t* __restrict__ redo_list, Task64* __restrict__ spill, uint32_t spill_cap) {
  static_assert(POOL >= (int)kMaxTasks, "the pool must hold what a query starts with");
  constexpr int U = 4;
  const Node64* __restrict__ nodes = t.nodes;
  const double* __restrict__ pts = t.pts;
  const uint32_t last = t.n_points - 1;
  Task64* const spill_w = spill + (uint64_t)blockIdx.x * spill_cap;
  const uint32_t lane = threadIdx.x;
  const uint64_t below = (1ull << lane) - 1ull;
  LdsF64* pf = (LdsF64*)ptk_smem;                        // [4][POOL]: nbd, off0, off1, off2
  LdsU64* pkey = (LdsU64*)(pf + 4 * POOL);               // [POOL]
  LdsU64* ekey = pkey + POOL;                            // [kRc64MaxEntries]
  LdsU64* eval = ekey + kRc64MaxEntries;                 // [kRc64MaxEntries]
  LdsU32* pref = (LdsU32*)(eval + kRc64MaxEntries);      // [POOL]
  LdsU32* pkbit = pref + POOL;                           // [POOL]: next key bit | sign: the handed-over form
  const uint32_t listed = ho.meta[ho.counter];
  const uint32_t n_heavy = listed > ho.max_heavy ? ho.max_heavy : listed;
  const double bound = d_mul(radius, e_inv);  // (already scaled by 1 / e for the approximate search, :265)

  for (uint32_t entry = blockIdx.x; entry < n_heavy; entry += gridDim.x) {  // (uniform)
    const uint32_t qi = ho.heavy_list[entry];
    const uint32_t nt = ho.ntasks[entry];
    const Task64* src = ho.tasks + (uint64_t)entry * kMaxTasks;
    const double* qrow = queries + (uint64_t)qi * t.dim;
    const double q0 = qrow[0];
    const double q1 = t.dim > 1 ? qrow[1] : metric64_pad<M>();
    const double q2 = t.dim > 2 ? qrow[2] : metric64_pad<M>();
    const uint32_t own = (uint32_t)counts[qi];
    // (non-monotone box distances do not matter here; kTasksRedo only says so)
    bool lost = nt == kTasksFromRoot || (nt > kMaxTasks && nt != kTasksRedo);
    uint32_t ntasks = lost ? 0u : nt;
    if (nt == kTasksRedo) {  // (hand_over64 wrote the tasks all the same -- how many, it did not say: count again)
      lost = true;
      ntasks = 0u;
    }

    // The handed-over stack, next-to-visit on top; task i gets the key prefix i.
    uint32_t count = ntasks;  // subtrees in the pool (uniform)
    for (uint32_t i = lane; i < ntasks; i += 64u) {
      const Task64 tk = src[i];
      const uint32_t sl = ntasks - 1u - i;
      pref[sl] = tk.ref;
      pf[0 * POOL + sl] = tk.nbd;
      pf[1 * POOL + sl] = tk.off0;
      pf[2 * POOL + sl] = tk.off1;
      pf[3 * POOL + sl] = tk.off2;
      pkey[sl] = (unsigned long long)i << (kRc64KeyTop + 1u);
      pkbit[sl] = kRc64KeyTop | 0x80000000u;  // sign bit: the handed-over form (the parent branch is read first)
    }
    bool busy = false, fresh = false;
    uint32_t ref = 0, spill_n = 0, kbit = 0, n_ent = 0, hits = 0;
    unsigned long long key = 0ull;
    double nbd = 0.0, off0 = 0.0, off1 = 0.0, off2 = 0.0;
    // The piece of a leaf this lane is measuring: first point, points seen, hits among them, its number in the leaf.
    uint32_t l_first = 0, l_pos = 0, l_mask = 0, l_piece = 0;

    for (;;) {
      if (count == 0u && spill_n != 0u) {  // (uniform) a drained pool takes back what was parked in HBM
        const uint32_t m = spill_n < (uint32_t)(POOL / 2) ? spill_n : (uint32_t)(POOL / 2);
        for (uint32_t i = lane; i < m; i += 64u) {
          const Task64 tk = spill_w[spill_n - m + i];
          pref[i] = tk.ref;
          pf[0 * POOL + i] = tk.nbd;
          pf[1 * POOL + i] = tk.off0;
          pf[2 * POOL + i] = tk.off1;
          pf[3 * POOL + i] = tk.off2;
          pkey[i] = (unsigned long long)__double_as_longlong(tk.gmax);  // (the key travels in the word a hand-over uses for gmax)
          pkbit[i] = tk.pad_;
        }
        count = m;
        spill_n -= m;
      }
      // (the ballot is also where the lanes meet after the pool was written)
      const bool want = !busy;
      const uint64_t wmask = __ballot(want);
      if (want) {
        const uint32_t rank = (uint32_t)__popcll(wmask & below);
        if (rank < count) {
          const uint32_t sl = count - 1u - rank;
          ref = pref[sl];
          nbd = pf[0 * POOL + sl];
          off0 = pf[1 * POOL + sl];
          off1 = pf[2 * POOL + sl];
          off2 = pf[3 * POOL + sl];
          key = pkey[sl];
          const uint32_t kb = pkbit[sl];
          kbit = kb & 0x7FFFFFFFu;
          fresh = (kb >> 31) != 0u;
          busy = true;
          if (!fresh && (ref & kLeafBit) != 0u) {  // a leaf begins
            l_first = (ref & 0x7FFFFFFFu) >> t.cbits;
            l_pos = l_mask = l_piece = 0u;
          }
        }
      }
      {
        const uint32_t nw = (uint32_t)__popcll(wmask);
        count -= nw < count ? nw : count;
      }

      // One node per lane.
      bool push = false, emit = false;
      Task64 pt{};
      unsigned long long e_key = 0ull, e_val = 0ull;
      if (busy) {
        const bool is_leaf = !fresh && (ref & kLeafBit) != 0u;
        if (!is_leaf) {
          // A branch, or (fresh) the parent branch of a pending record whose far child is entered as traverse64_3
          // enters it: the same arithmetic with the side given instead of chosen.
          const Node64 nd = nodes[fresh ? (ref & 0x3FFFFFFFu) : ref];
          const double v = sel3d(nd.axis, q0, q1, q2);
          const bool near_left = d_sub(d_sub(d_add(nd.left_max, nd.right_min), v), v) > 0.0;
          const bool go_left = fresh ? (ref & kRecSide) != 0u : near_left;
          const double dv = d_sub(go_left ? nd.right_min : nd.left_max, v);
          const double new_off = M::one(dv);
          const uint32_t far_ref = go_left ? nd.right_ref : nd.left_ref;
          if (fresh) {
            off0 = nd.axis == 0 ? new_off : off0;
            off1 = nd.axis == 1 ? new_off : off1;
            off2 = nd.axis == 2 ? new_off : off2;
            ref = far_ref;
            fresh = false;
          } else {
            const double far_nbd = d_add(d_sub(nbd, sel3d(nd.axis, off0, off1, off2)), new_off);
            if (bound >= far_nbd) {  // the test of kd_tree_search.hpp:99 with the radius visitor's constant max()
              push = true;
              pt.ref = far_ref;
              pt.nbd = far_nbd;
              pt.off0 = nd.axis == 0 ? new_off : off0;
              pt.off1 = nd.axis == 1 ? new_off : off1;
              pt.off2 = nd.axis == 2 ? new_off : off2;
              pt.gmax = __longlong_as_double((long long)(key | (1ull << kbit)));  // visited second
              pt.pad_ = kbit - 1u;
            }
            ref = go_left ? nd.left_ref : nd.right_ref;  // visited first: its bit stays 0
            kbit -= 1u;
          }
          if ((ref & kLeafBit) != 0u) {  // the lane's next node is a leaf
            l_first = (ref & 0x7FFFFFFFu) >> t.cbits;
            l_pos = l_mask = l_piece = 0u;
          }
        } else {
          const uint32_t lv = ref & 0x7FFFFFFFu;
          const uint32_t begin = lv >> t.cbits;
          const uint32_t cnt = lv & t.cmask;
          double px[U], py[U], pz[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {  // every load of the round before the first use
            const uint32_t pu = begin + u <= last ? begin + u : last;  // in range past the leaf's end too
            const double4 a = *reinterpret_cast<const double4*>(pts + (uint64_t)pu * kStride64D3);
            px[u] = a.x;
            py[u] = a.y;
            pz[u] = a.z;
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if ((uint32_t)u < cnt) {
              const double d = d_mul(M::acc(M::acc(M::one(d_sub(q0, px[u])), d_sub(q1, py[u])), d_sub(q2, pz[u])), e_inv);
              l_mask |= (bound > d ? 1u : 0u) << l_pos;  // strict (:141)
              ++l_pos;
            }
          }
          // (a piece closes after kRc64MaskBits = 32 points -- eight steps of four -- or at the leaf's end)
          const bool leaf_done = cnt <= (uint32_t)U;
          if (l_pos == kRc64MaskBits || leaf_done) {
            if (l_mask != 0u) {
              emit = true;
              e_key = key | (unsigned long long)l_piece;
              e_val = (unsigned long long)((l_first << t.cbits) | l_pos) | ((unsigned long long)l_mask << 32);
              hits += (uint32_t)__popcll((unsigned long long)l_mask);
            }
            l_first += l_pos;
            l_pos = l_mask = 0u;
            ++l_piece;
          }
          if (!leaf_done) {
            ref = kLeafBit | ((begin + (uint32_t)U) << t.cbits) | (cnt - (uint32_t)U);
          } else {
            busy = false;
          }
        }
      }

      // The leaves with hits of this step go behind the entries (all lanes take part in the ballot).
      {
        const uint64_t em = __ballot(emit);
        if (emit) {
          const uint32_t at = n_ent + (uint32_t)__popcll(em & below);
          if (at < kRc64MaxEntries) {
            ekey[at] = e_key;
            eval[at] = e_val;
          }
        }
        n_ent += (uint32_t)__popcll(em);
      }

      // Far children kept in this step go onto the pool.
      const uint64_t pmask = __ballot(push);
      if (push) {
        const uint32_t sl = count + (uint32_t)__popcll(pmask & below);
        if (sl < (uint32_t)POOL) {
          pref[sl] = pt.ref;
          pf[0 * POOL + sl] = pt.nbd;
          pf[1 * POOL + sl] = pt.off0;
          pf[2 * POOL + sl] = pt.off1;
          pf[3 * POOL + sl] = pt.off2;
          pkey[sl] = (unsigned long long)__double_as_longlong(pt.gmax);
          pkbit[sl] = pt.pad_;
        } else if (spill_n + (sl - (uint32_t)POOL) < spill_cap) {  // no room in LDS: parked in HBM
          spill_w[spill_n + (sl - (uint32_t)POOL)] = pt;
        }
      }
      count += (uint32_t)__popcll(pmask);
      if (count > (uint32_t)POOL) {
        spill_n += count - (uint32_t)POOL;
        count = (uint32_t)POOL;
        if (spill_n > spill_cap) {  // a subtree was lost: this row is counted again from the root
          lost = true;
          count = 0;
          spill_n = 0;
          busy = false;
        }
      }
      if (__ballot(busy) == 0ull && count == 0u && spill_n == 0u) break;
    }
    if (n_ent > kRc64MaxEntries) lost = true;

    // The entries in the reference's visit order: a bitonic sort by key (n_ent padded to a power of two with keys above
    // every real one).
    uint32_t run = 0;
    if (!lost && n_ent != 0u) {
      uint32_t np = 64u;
      while (np < n_ent) np <<= 1;
      for (uint32_t i = n_ent + lane; i < np; i += 64u) ekey[i] = ~0ull;
      __syncthreads();
      for (uint32_t k = 2u; k <= np; k <<= 1) {        // (uniform)
        for (uint32_t j = k >> 1; j != 0u; j >>= 1) {  // (uniform)
          for (uint32_t i = lane; i < np; i += 64u) {
            const uint32_t o = i ^ j;
            if (o > i) {
              const unsigned long long a = ekey[i], b = ekey[o];
              const bool up = (i & k) == 0u;
              if (up ? a > b : a < b) {
                ekey[i] = b;
                ekey[o] = a;
                const unsigned long long v = eval[i];
                eval[i] = eval[o];
                eval[o] = v;
              }
            }
          }
          __syncthreads();
        }
      }
      // A run of the entry block for them.
      if (lane == 0) run = atomicAdd(&hv.meta[kMetaRc64Entries], n_ent);
      run = (uint32_t)__shfl((int)run, 0);
      if ((uint64_t)run + n_ent > (uint64_t)hv.entry_cap) {
        lost = true;
      } else {
        for (uint32_t i = lane; i < n_ent; i += 64u) hv.entries[run + i] = eval[i];
      }
      __syncthreads();  // (the entries have been read before the next query writes its own)
    }
    const uint32_t found = wave64_sum_u32(hits);
    if (lane == 0) {
      hv.rows[entry] = qi;
      hv.own[entry] = own;
      hv.run_at[entry] = run;
      hv.run_n[entry] = lost ? kRc64Lost : n_ent;
      if (lost) {
        redo_list[atomicAdd(&hv.meta[kMetaRedo], 1u)] = qi;
      } else {
        counts[qi] = (uint64_t)own + found;
      }
    }
  }
}

// The fill pass of the handed-over queries: block b takes hand-overs b, b + grid, ...; the row's first `own` hits are the
// capped lane's, the rest is written here from the sorted entries.  Rows whose entries were lost are listed (over_list,
// counted at meta[kMetaRc64Over]) for radius64_redo_kernel<M, true>.
template <class M>
__global__ __launch_bounds__(64) void radius64_coop_replay_kernel(
    DevTree64 t, const double* __restrict__ queries, double e_inv, RadiusHeavy64 hv, const uint64_t* __restrict__ offsets,
    Neighbor64* __restrict__ out, uint32_t* __restrict__ over_list) {
  const uint32_t lane = threadIdx.x;
  const double* __restrict__ pts = t.pts;
  const uint32_t listed = hv.meta[kMetaHeavy];
  const uint32_t n_heavy = listed > hv.max_heavy ? hv.max_heavy : listed;
  for (uint32_t h = blockIdx.x; h < n_heavy; h += gridDim.x) {  // (uniform)
    const uint32_t qi = hv.rows[h];
    const uint32_t n = hv.run_n[h];
    if (n == kRc64Lost) {
      if (lane == 0) over_list[atomicAdd(&hv.meta[kMetaRc64Over], 1u)] = qi;
      continue;
    }
    const double* qrow = queries + (uint64_t)qi * t.dim;
    const double q0 = qrow[0];
    const double q1 = t.dim > 1 ? qrow[1] : metric64_pad<M>();
    const double q2 = t.dim > 2 ? qrow[2] : metric64_pad<M>();
    const unsigned long long* __restrict__ ent = hv.entries + hv.run_at[h];
    uint64_t at = offsets[qi] + hv.own[h];
    for (uint32_t i0 = 0; i0 < n; i0 += 64u) {  // (uniform)
      const bool have = i0 + lane < n;
      const unsigned long long e = have ? ent[i0 + lane] : 0ull;
      uint32_t mask = (uint32_t)(e >> 32);
      const uint32_t first = ((uint32_t)e & 0x7FFFFFFFu) >> t.cbits;
      const uint32_t c = (uint32_t)__popcll((unsigned long long)mask);
      const uint32_t incl = wave64_scan_u32(c, lane);
      uint64_t w = at + (incl - c);
      while (mask != 0u) {
        const uint32_t b = (uint32_t)__builtin_ctz(mask);
        mask &= mask - 1u;
        const double4 a = *reinterpret_cast<const double4*>(pts + (uint64_t)(first + b) * kStride64D3);
        Neighbor64 nb;
        nb.index = (int32_t)__double_as_longlong(a.w);
        nb.pad_ = 0;
        nb.distance = d_mul(M::acc(M::acc(M::one(d_sub(q0, a.x)), d_sub(q1, a.y)), d_sub(q2, a.z)), e_inv);
        out[w++] = nb;
      }
      at += (uint32_t)__shfl((int)incl, 63);
    }
  }
}

// The reference search from the root, one lane, for the rows the cooperative count could not finish: FILL = false counts
// them again (counts[row]), FILL = true fills them again (the same values at the same places the other writers of such a
// row put them).  `n_word`: the word of `meta` that counts `list`.
template <class M, bool FILL>
__global__ __launch_bounds__(64) void radius64_redo_kernel(
    DevTree64 t, const double* __restrict__ queries, double radius, double e_inv, uint64_t* __restrict__ counts,
    const uint64_t* __restrict__ offsets, Neighbor64* __restrict__ out, const uint32_t* __restrict__ meta, uint32_t n_word,
    const uint32_t* __restrict__ list, Rec64* __restrict__ stack, uint32_t slots) {
  const uint32_t n = meta[n_word];
  for (uint32_t i = blockIdx.x * 64u + threadIdx.x; i < n; i += gridDim.x * 64u) {
    const uint32_t qi = list[i];
    Radius64Policy<FILL> pol;
    pol.radius = d_mul(radius, e_inv);
    pol.e_inv = e_inv;
    pol.count = 0;
    pol.out = FILL ? out + offsets[qi] : nullptr;
    search64<M, true>(t, queries, qi, pol, stack, slots);
    if (!FILL) counts[qi] = pol.count;
  }
}

}  // namespace ptk
