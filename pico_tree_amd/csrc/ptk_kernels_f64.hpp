// ptk_kernels_f64.hpp -- gfx950 device code for kd_trees over DOUBLE points, any dimension.
//
// The reference's kd_tree is generic over the scalar type (point_traits<>::scalar_type) and its
// Python module builds KdTree objects over float64 arrays as readily as over float32 ones
// (/root/reference/src/pyco_tree/pico_tree/_pyco_tree/kd_tree.hpp:383-445, def_core.hpp:17-18).
// This file is that instantiation for the device: the same per-lane replay of the reference's
// visit order as ptk_kernels_nd.hpp -- one query per lane, near child first, far child iff
// `visitor.max() >= node_box_distance` (internal/kd_tree_search.hpp:52-105) -- with every scalar a
// double and one IEEE operation per reference operation (__dadd_rn / __dsub_rn / __dmul_rn: no
// contraction), so indices and distance bits equal the reference compiled with -ffp-contract=off.
//
// Where the per-lane state lives:
//   q[dim], off[dim]  LDS, [axis][lane] (search.hpp:47,111); 16 bytes per axis per lane
//   record stack      the newest S records in an LDS ring ([slot][lane]: a double and a meta word per
//                     slot), older ones spilled to HBM scratch ([block][slot][lane], 16-byte records:
//                     a wave's spill / refill is one coalesced 1 KiB access); 2 * depth + 2 slots
//                     always suffice.
//   k-list            registers for k <= 64 (Knn64RegPolicy, as KnnRegPolicy of ptk_kernels.hpp),
//                     else the caller's output row itself (neighbor<int, double>, 16 bytes)
// Layouts (ptk_backend_f64.hpp, encode64):
//   nodes : 32 B per branch {left_max, right_min, left_ref, right_ref, axis, 0}
//   ref   : bit 31 = leaf; leaf = (begin << cbits) | count; branch = branch index
//   pts   : leaf order, row-major, `stride` doubles per point (dim <= 3: 4 = {x, y, z, original index in the
//           low word of the fourth}, the unused axes zero: they add an exact +0 to every distance; else stride = dim); index[]:
//           original index per position
//   record: x = bit 31 undo | bit 30 (pending: far child is the right one; undo: nbd) |
//               bits 29:0 (pending: branch index; undo_off: axis),  val = double
// dim <= 3 takes traverse64_3: q and off in registers, the coordinates of two leaf points loaded
// before either is used.  Measured steps (BASELINE config 2 in double): DESIGN.md section 4, K9.

#pragma once

#include "ptk_encode.hpp"  // kStride64D3
#include "ptk_kernels_nd.hpp"

namespace ptk {

struct Neighbor64 {  // pico_tree::neighbor<int, double> (core.hpp:24-46): 16 bytes, distance at 8
  int32_t index;
  int32_t pad_;
  double distance;
};
struct Node64 {
  double left_max, right_min;
  uint32_t left_ref, right_ref, axis, pad_;
};
struct Rec64 {
  uint32_t x, pad_;
  double val;
};
static_assert(sizeof(Neighbor64) == 16 && sizeof(Node64) == 32 && sizeof(Rec64) == 16, "f64 records");

struct DevTree64 {
  const Node64* nodes;
  const double* pts;
  const int32_t* index;
  const uint2* ranges;  // per branch: position range of its whole subtree (box search)
  const double2* outer; // per branch {left_min, right_max}: the topological metrics only (else null)
  uint32_t root_ref;
  uint32_t cbits;
  uint32_t cmask;
  uint32_t dim;
  uint32_t stride;    // doubles per point in pts
  uint32_t n_points;
};

constexpr double kDblMax = 1.7976931348623157e308;
typedef PTK_LDS double LdsDouble;

__device__ __forceinline__ double d_add(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ double d_sub(double a, double b) { return __dsub_rn(a, b); }
__device__ __forceinline__ double d_mul(double a, double b) { return __dmul_rn(a, b); }

// metric.hpp:72-150, as MetricL2 / MetricL1 / MetricLInf of ptk_kernels.hpp.
struct Metric64L2 {
  static constexpr bool kMin = false;
  static constexpr bool kTopo = false;
  __device__ __forceinline__ static double one(double x) { return d_mul(x, x); }
  __device__ __forceinline__ static double acc(double d, double diff) { return d_add(d, d_mul(diff, diff)); }
};
struct Metric64L1 {
  static constexpr bool kMin = false;
  static constexpr bool kTopo = false;
  __device__ __forceinline__ static double one(double x) { return fabs(x); }
  __device__ __forceinline__ static double acc(double d, double diff) { return d_add(d, fabs(diff)); }
};
struct Metric64LInf {
  static constexpr bool kMin = false;
  static constexpr bool kTopo = false;
  __device__ __forceinline__ static double one(double x) { return fabs(x); }
  __device__ __forceinline__ static double acc(double d, double diff) {
    const double a = fabs(diff);
    return d < a ? a : d;  // std::max(d, a)
  }
};
struct Metric64LNInf {  // metric_lninf, as MetricLNInf of ptk_kernels.hpp
  static constexpr bool kMin = true;
  static constexpr bool kTopo = false;
  __device__ __forceinline__ static double one(double x) { return fabs(x); }
  __device__ __forceinline__ static double acc(double d, double diff) {
    const double a = fabs(diff);
    return a < d ? a : d;  // std::min(d, a)
  }
};
template <class M>
__device__ __forceinline__ double metric64_init() {
  return M::kMin ? kDblMax : 0.0;
}
template <class M>
__device__ __forceinline__ double metric64_pad() {
  return M::kMin ? __longlong_as_double(0x7FF0000000000000ll) : 0.0;
}

// The reference's topological metrics over double points (metric.hpp:186-257; ptk_kernels_topo.hpp is the float form):
// distance.hpp:26-29 s1_distance, segment.hpp:38-46 segment_r1::distance, :77-99 segment_s1::distance.
__device__ __forceinline__ double s1_distance64(double x, double y) {
  const double d = fabs(d_sub(x, y));
  const double w = d_sub(1.0, d);
  return w < d ? w : d;  // std::min(d, 1 - d)
}
__device__ __forceinline__ double seg_r1_distance64(double mn, double mx, double x) {
  return x < mn ? d_sub(mn, x) : (x > mx ? d_sub(x, mx) : 0.0);
}
__device__ __forceinline__ double seg_s1_distance64(double mn, double mx, double x) {
  const double a = s1_distance64(x, mn), b = s1_distance64(x, mx);
  const double m = b < a ? b : a;  // std::min(a, b)
  if (mn <= mx) return (x < mn || x > mx) ? m : 0.0;
  return (x < mx || x > mn) ? 0.0 : m;
}
struct Topo64SO2 {  // metric_so2, metric.hpp:197-220: dim 1, axis 0 on the circle
  static constexpr bool kMin = false;
  static constexpr bool kTopo = true;
  static constexpr uint32_t kS1Mask = 1u;
  __device__ __forceinline__ static double box(double mn, double mx, double v, uint32_t) {
    return fabs(seg_s1_distance64(mn, mx, v));
  }
  __device__ __forceinline__ static double point(double q0, double, double, double p0, double, double) {
    return s1_distance64(q0, p0);
  }
};
struct Topo64SE2 {  // metric_se2_squared, metric.hpp:228-257: x, y on the line, the angle (axis 2) on the circle
  static constexpr bool kMin = false;
  static constexpr bool kTopo = true;
  static constexpr uint32_t kS1Mask = 4u;
  __device__ __forceinline__ static double box(double mn, double mx, double v, uint32_t axis) {
    const double d = axis < 2u ? seg_r1_distance64(mn, mx, v) : seg_s1_distance64(mn, mx, v);
    return d_mul(d, d);
  }
  __device__ __forceinline__ static double point(double q0, double q1, double q2, double p0, double p1, double p2) {
    const double dx = d_sub(q0, p0), dy = d_sub(q1, p1);
    const double a = s1_distance64(q2, p2);
    return d_add(d_add(d_mul(dx, dx), d_mul(dy, dy)), d_mul(a, a));  // sum over x, y from 0, + squared_s1
  }
};

typedef PTK_LDS uint32_t LdsU32;
#ifndef PTK_RING64
#define PTK_RING64 8
#endif
#ifndef PTK_LEAF64
#define PTK_LEAF64 2
#endif
constexpr int kRing64 = PTK_RING64;  // LDS ring slots per lane (12 bytes each); a power of two
constexpr int kLeaf64 = PTK_LEAF64;  // leaf points per memory round trip (dim <= 3)

// LDS of a block: [vectors: nvec * dim doubles][ring values: S doubles][ring meta: S words], all [slot][lane].
__host__ __device__ constexpr size_t lds64_bytes(uint32_t nvec, uint32_t dim) {
  return ((size_t)nvec * dim * 8 + (size_t)kRing64 * 12) * 64;
}

// Records are numbered 0, 1, 2, ... in push order; [base, top) is resident in the ring (record i at
// slot i mod S), [0, base) has been spilled (record i at ovf[i * 64]).  As Stack of ptk_kernels.hpp.
struct Stack64 {
  static constexpr int S = kRing64;
  static constexpr int kRefill = S / 2;
  LdsDouble* rv;  // this lane's column of values
  LdsU32* rm;     // ... and of meta words
  Rec64* ovf;     // this lane's spill column
  int top, base;
  __device__ __forceinline__ void init(uint32_t nvec, uint32_t dim, Rec64* scratch, uint32_t slots) {
    LdsDouble* ring = (LdsDouble*)ptk_smem + (size_t)nvec * dim * 64;
    rv = ring + threadIdx.x;
    rm = (LdsU32*)(ring + (size_t)S * 64) + threadIdx.x;
    ovf = scratch + (uint64_t)blockIdx.x * slots * 64 + threadIdx.x;
    top = 0;
    base = 0;
  }
  __device__ __forceinline__ bool empty() const { return top == 0; }
  __device__ __forceinline__ void push(uint32_t meta, double val) {
    if (top - base == S) {  // ring full: spill the oldest resident record
      Rec64 r;
      r.x = rm[(base & (S - 1)) * 64];
      r.pad_ = 0;
      r.val = rv[(base & (S - 1)) * 64];
      ovf[(uint64_t)base * 64] = r;
      ++base;
    }
    rm[(top & (S - 1)) * 64] = meta;
    rv[(top & (S - 1)) * 64] = val;
    ++top;
  }
  __device__ __forceinline__ Rec64 pop() {
    if (top == base) {  // ring empty, spilled records remain: bring a batch back
      Rec64 r[kRefill];
#pragma unroll
      for (int i = 0; i < kRefill; ++i) {
        const int idx = base - 1 - i;
        r[i] = ovf[(uint64_t)(idx >= 0 ? idx : 0) * 64];
      }
#pragma unroll
      for (int i = 0; i < kRefill; ++i) {
        const int idx = base - 1 - i;
        if (idx >= 0) {
          rm[(idx & (S - 1)) * 64] = r[i].x;
          rv[(idx & (S - 1)) * 64] = r[i].val;
        }
      }
      base = base > kRefill ? base - kRefill : 0;
    }
    --top;
    Rec64 r;
    r.x = rm[(top & (S - 1)) * 64];
    r.pad_ = 0;
    r.val = rv[(top & (S - 1)) * 64];
    return r;
  }
};

// ---- result policies (search_visitor.hpp), candidates scaled by 1/e as in ptk_kernels.hpp ----
struct Knn64Policy {  // :83-123 / :198-247
  Neighbor64* list;   // the output row
  uint32_t k;
  uint32_t filled;
  double worst;
  double e_inv;
  __device__ __forceinline__ double max() const { return worst; }
  __device__ __forceinline__ void visit(int32_t idx, double d) {
    d = d_mul(d, e_inv);
    if (worst > d) {
      if (filled < k) ++filled;
      uint32_t j = filled - 1;
      while (j > 0) {  // insert_sorted (:24-38): shift while strictly smaller => stable on ties
        const Neighbor64 prev = list[j - 1];
        if (!(d < prev.distance)) break;
        list[j] = prev;
        --j;
      }
      Neighbor64 nb;
      nb.index = idx;
      nb.pad_ = 0;
      nb.distance = d;
      list[j] = nb;
      if (filled == k) worst = list[k - 1].distance;
    }
  }
  __device__ __forceinline__ void end_query() {
    if (filled < k) {  // fewer reachable points than k: the reference's sentinel (:102)
      Neighbor64 nb;
      nb.index = 0;
      nb.pad_ = 0;
      nb.distance = kDblMax;
      list[k - 1] = nb;
    }
  }
};

// Sorted k-list in registers for k <= K <= 64: K branch-free compare / select steps per accepted
// candidate, insert_sorted (:24-38) exactly -- strict `<` keeps a new entry behind equal distances;
// slots start at DBL_MAX (the sentinel of :102).  See KnnRegPolicy in ptk_kernels.hpp.
template <int K>
struct Knn64RegPolicy {
  double ld[K];
  int32_t li[K];
  uint32_t k;
  double worst;
  double e_inv;
  __device__ __forceinline__ void init(uint32_t k_, double e_inv_) {
    k = k_;
    e_inv = e_inv_;
    worst = kDblMax;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      ld[j] = kDblMax;
      li[j] = 0;
    }
  }
  __device__ __forceinline__ double max() const { return worst; }
  __device__ __forceinline__ void visit(int32_t idx, double d) {
    d = d_mul(d, e_inv);
    if (worst > d) {
#pragma unroll
      for (int j = K - 1; j >= 1; --j) {
        const bool shift = d < ld[j - 1];
        const bool here = d < ld[j];
        li[j] = shift ? li[j - 1] : (here ? idx : li[j]);
        ld[j] = shift ? ld[j - 1] : (here ? d : ld[j]);
      }
      if (d < ld[0]) {
        ld[0] = d;
        li[0] = idx;
      }
      double w = ld[0];
#pragma unroll
      for (int j = 1; j < K; ++j) w = (uint32_t)j < k ? ld[j] : w;  // slot k - 1
      worst = w;
    }
  }
  __device__ __forceinline__ void store(Neighbor64* row) const {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if ((uint32_t)j < k) {
        Neighbor64 nb;
        nb.index = li[j];
        nb.pad_ = 0;
        nb.distance = ld[j];
        row[j] = nb;
      }
    }
  }
};

struct Nn64Policy {  // :42-65 / :165-193 -- k == 1 keeps the best in registers
  double best_d;
  int32_t best_i;
  double e_inv;
  __device__ __forceinline__ double max() const { return best_d; }
  __device__ __forceinline__ void visit(int32_t idx, double d) {
    d = d_mul(d, e_inv);
    if (best_d > d) {
      best_d = d;
      best_i = idx;
    }
  }
};

template <bool FILL>
struct Radius64Policy {  // :127-156 / :252-288
  double radius;  // already scaled by 1/e for the approximate search (:265)
  double e_inv;
  uint64_t count;
  Neighbor64* out;
  __device__ __forceinline__ double max() const { return radius; }
  __device__ __forceinline__ void visit(int32_t idx, double d) {
    d = d_mul(d, e_inv);
    if (radius > d) {  // strict
      if (FILL) {
        Neighbor64 nb;
        nb.index = idx;
        nb.pad_ = 0;
        nb.distance = d;
        out[count] = nb;
      }
      ++count;
    }
  }
};

template <class M, class Policy>
__device__ __forceinline__ void traverse64(const DevTree64& t, LdsDouble* q, LdsDouble* off, Policy& pol, Stack64& st) {
  const Node64* __restrict__ nodes = t.nodes;
  const double* __restrict__ pts = t.pts;
  const int32_t* __restrict__ index = t.index;
  const uint32_t dim = t.dim;
  uint32_t ref = t.root_ref;
  double nbd = 0.0;

  for (;;) {
    while (!(ref & kLeafBit)) {
      const Node64 nd = nodes[ref];
      const double v = q[nd.axis * 64];
      const bool go_left = d_sub(d_sub(d_add(nd.left_max, nd.right_min), v), v) > 0.0;  // search.hpp:76
      const double dv = d_sub(go_left ? nd.right_min : nd.left_max, v);
      const double new_off = M::one(dv);                                                  // :80,84
      const double far_nbd = d_add(d_sub(nbd, off[nd.axis * 64]), new_off);               // :94
      if (pol.max() >= far_nbd) st.push(ref | (go_left ? kRecSide : 0u), far_nbd);
      ref = go_left ? nd.left_ref : nd.right_ref;
    }
    {
      const uint32_t lv = ref & 0x7FFFFFFFu;
      const uint32_t begin = lv >> t.cbits;
      const uint32_t count = lv & t.cmask;
      for (uint32_t j = 0; j < count; ++j) {
        const double* p = pts + (uint64_t)(begin + j) * t.stride;
        const int32_t pi = index[begin + j];
        double d = metric64_init<M>();
        // internal::sum (metric.hpp:36-51), kNdBatch coordinates loaded before the first is used; a
        // slot past the last axis contributes diff = 0, an exact no-op (see traverse_nd).
        for (uint32_t a = 0; a < dim; a += kNdBatch) {
          double pc[kNdBatch], qc[kNdBatch];
#pragma unroll
          for (uint32_t u = 0; u < kNdBatch; ++u) {
            const uint32_t au = a + u < dim ? a + u : dim - 1;
            pc[u] = p[au];
            qc[u] = q[au * 64];
          }
#pragma unroll
          for (uint32_t u = 0; u < kNdBatch; ++u) d = M::acc(d, a + u < dim ? d_sub(qc[u], pc[u]) : metric64_pad<M>());
        }
        pol.visit(pi, d);
      }
    }
    for (;;) {
      if (st.empty()) return;
      const Rec64 r = st.pop();
      if (r.x & kRecUndo) {
        if (r.x & kRecSide) {
          nbd = r.val;
        } else {
          off[(r.x & 0x3FFFFFFFu) * 64] = r.val;
        }
        continue;
      }
      if (pol.max() >= r.val) {  // the authoritative test of search.hpp:99
        const uint32_t idx = r.x & 0x3FFFFFFFu;
        const bool far_is_right = (r.x & kRecSide) != 0;
        const Node64 nd = nodes[idx];
        const double dv = d_sub(far_is_right ? nd.right_min : nd.left_max, q[nd.axis * 64]);
        const double new_off = M::one(dv);
        st.push(kRecUndo | nd.axis, off[nd.axis * 64]);
        st.push(kRecUndo | kRecSide, nbd);
        off[nd.axis * 64] = new_off;
        nbd = r.val;
        ref = far_is_right ? nd.right_ref : nd.left_ref;
        break;
      }
    }
  }
}

__device__ __forceinline__ void stage_query64(
    const double* __restrict__ queries, uint32_t dim, uint64_t qi, LdsDouble*& q, LdsDouble*& off) {
  LdsDouble* base = (LdsDouble*)ptk_smem;
  q = base + threadIdx.x;
  off = base + (size_t)dim * 64 + threadIdx.x;
  const double* row = queries + qi * dim;
  for (uint32_t a = 0; a < dim; ++a) {
    q[a * 64] = row[a];
    off[a * 64] = 0.0;  // search.hpp:47
  }
}

// dim <= 3 (32-byte point records): the same walk with q and off in registers.
__device__ __forceinline__ double sel3d(uint32_t axis, double a0, double a1, double a2) {
  return axis == 0 ? a0 : (axis == 1 ? a1 : a2);
}
// A subtree still to be searched, as the cooperative search (ptk_kernels_coop64.hpp) keeps it: Task of ptk_kernels.hpp
// in double.  `gmax` = the largest box distance of a far child on the path down to it; its sign bit marks a subtree
// handed over by a capped traversal in the form of its pending record (`ref` = the record's {parent branch | side}, the
// offset of the split axis still the parent's: entering it reads the parent branch).
struct Task64 {
  uint32_t ref, pad_;
  double nbd, off0, off1, off2, gmax;
};
static_assert(sizeof(Task64) == 48, "Task64");
// Where a capped traversal leaves its unfinished work (Handover of ptk_kernels.hpp; a query that finds the list full
// goes on in its lane).
struct Handover64 {
  uint32_t counter;      // word of `meta` that counts this list
  uint32_t* meta;        // [kMetaWords] counters
  uint32_t* heavy_list;  // [max_heavy] rows of the queries handed over
  uint32_t* ntasks;      // [max_heavy] tasks of list entry h (or kTasksFromRoot / kTasksRedo)
  Task64* tasks;         // [max_heavy][kMaxTasks]
  uint32_t max_heavy;
};

// What a capped traversal that stopped was in the middle of (traverse64_3<.., CAPPED>): the box distance state it would
// unwind with; the far child it was about to enter is back on the stack.
struct Trav64State {
  double nbd, o0, o1, o2;
};
// The hand-over itself (once per long query): every pending far child that can still matter with the state it would
// be entered with, next-to-visit first -- as traverse<.., CAPPED> of ptk_kernels.hpp hands over.
__device__ __forceinline__ void hand_over64(const Handover64* ho, uint32_t row, uint32_t h, Trav64State ts, Stack64 st,
                                            double bound) {
  ho->heavy_list[h] = row;
  Task64* out = ho->tasks + (uint64_t)h * kMaxTasks;
  uint32_t n = 0;
  // `gmax` of a task = the largest box distance of a far child on the path from the root to it.  The far children on
  // the path are the ones the stack still holds an undo record of; while each of them had a box distance >= its
  // parent's (`monotone`: the restored value never exceeds the one it replaces) the largest one is the current one.
  // (Far children entered and left again are on nobody's path: the traversal itself keeps no such flag -- updating one
  // per far child cost the k = 1 kernel 5 % of its time.)
  bool monotone = true;
  while (!st.empty()) {
    const Rec64 u = st.pop();
    if (u.x & kRecUndo) {
      if (u.x & kRecSide) {
        if (u.val > ts.nbd) monotone = false;
        ts.nbd = u.val;
      } else {
        const uint32_t axis = u.x & 0x3FFFFFFFu;
        ts.o0 = axis == 0 ? u.val : ts.o0;
        ts.o1 = axis == 1 ? u.val : ts.o1;
        ts.o2 = axis == 2 ? u.val : ts.o2;
      }
    } else if (bound >= u.val) {
      if (n < kMaxTasks) {
        Task64 k;
        k.ref = u.x;
        k.pad_ = 0;
        k.nbd = u.val;
        k.off0 = ts.o0;
        k.off1 = ts.o1;
        k.off2 = ts.o2;
        // (while `monotone` holds the largest box distance on the path so far is the current one; the sign: the
        // pending-record form)
        k.gmax = __longlong_as_double(__double_as_longlong(ts.nbd < u.val ? u.val : ts.nbd) | (long long)0x8000000000000000ull);
        out[n] = k;
      }
      ++n;
    }
  }
  ho->ntasks[h] = !monotone ? kTasksRedo : (n > kMaxTasks ? kTasksFromRoot : n);
}

// CAPPED (k-NN, exact; r06): a query that has entered more than `cap` far children stops (returns false): the far
// child it was about to enter goes back on the stack, the state it would unwind with to *ts -- the caller hands the
// stack over (hand_over64) or, when the list is full, calls again with `resume` and no cap.  The stop leaves through
// the traversal's one exit (the test for an empty stack): an exit of its own inside the unwind loop -- hand-over and
// return in place -- cost every capped kernel 7-8 % of its time on the bulk of a batch that never stops (knn = 16,
// 7.2 M queries: 11.7 against 10.8 ms; profiles/r06_notes.txt item 12).
template <class M, class Policy, bool CAPPED = false>
__device__ __forceinline__ bool traverse64_3(const DevTree64& t, double q0, double q1, double q2, Policy& pol, Stack64& st,
                                             uint32_t cap = 0, Trav64State* ts = nullptr, bool resume = false) {
  const Node64* __restrict__ nodes = t.nodes;
  const double* __restrict__ pts = t.pts;
  const uint32_t last = t.n_points - 1;
  uint32_t ref = t.root_ref;
  double nbd = 0.0, o0 = 0.0, o1 = 0.0, o2 = 0.0;  // search.hpp:47
  uint32_t entered = 0;
  bool stopped = false;
  if constexpr (CAPPED) {
    if (resume) {  // (an empty leaf: falls through to the unwind)
      ref = kLeafBit;
      nbd = ts->nbd;
      o0 = ts->o0;
      o1 = ts->o1;
      o2 = ts->o2;
    }
  }

  for (;;) {
    while (!(ref & kLeafBit)) {
      const Node64 nd = nodes[ref];
      const double v = sel3d(nd.axis, q0, q1, q2);
      const bool go_left = d_sub(d_sub(d_add(nd.left_max, nd.right_min), v), v) > 0.0;  // search.hpp:76
      const double dv = d_sub(go_left ? nd.right_min : nd.left_max, v);
      const double new_off = M::one(dv);                                                  // :80,84
      const double far_nbd = d_add(d_sub(nbd, sel3d(nd.axis, o0, o1, o2)), new_off);      // :94
      if (pol.max() >= far_nbd) st.push(ref | (go_left ? kRecSide : 0u), far_nbd);
      ref = go_left ? nd.left_ref : nd.right_ref;
    }
    {
      const uint32_t lv = ref & 0x7FFFFFFFu;
      const uint32_t begin = lv >> t.cbits;
      const uint32_t count = lv & t.cmask;
      for (uint32_t j = 0; j < count; j += kLeaf64) {
        double px[kLeaf64], py[kLeaf64], pz[kLeaf64];
        int32_t pi[kLeaf64];
#pragma unroll
        for (int u = 0; u < kLeaf64; ++u) {  // every load of the round before the first use
          const uint32_t pu = begin + j + u <= last ? begin + j + u : last;  // in range past the leaf's end too
          const double4 a = *reinterpret_cast<const double4*>(pts + (uint64_t)pu * kStride64D3);  // {x, y, z, index}
          px[u] = a.x;
          py[u] = a.y;
          pz[u] = a.z;
          pi[u] = (int32_t)__double_as_longlong(a.w);
        }
#pragma unroll
        for (int u = 0; u < kLeaf64; ++u) {
          if (j + u < count) {
            // internal::sum (metric.hpp:36-51) from d = 0: acc(0, x) == one(x) exactly
            pol.visit(pi[u], M::acc(M::acc(M::one(d_sub(q0, px[u])), d_sub(q1, py[u])), d_sub(q2, pz[u])));
          }
        }
      }
    }
    for (;;) {
      if (st.empty() || (CAPPED && stopped)) {
        if constexpr (CAPPED) {
          ts->nbd = nbd;
          ts->o0 = o0;
          ts->o1 = o1;
          ts->o2 = o2;
        }
        return !stopped;
      }
      const Rec64 r = st.pop();
      if (r.x & kRecUndo) {
        if (r.x & kRecSide) {
          nbd = r.val;
        } else {
          const uint32_t axis = r.x & 0x3FFFFFFFu;
          o0 = axis == 0 ? r.val : o0;
          o1 = axis == 1 ? r.val : o1;
          o2 = axis == 2 ? r.val : o2;
        }
        continue;
      }
      if (pol.max() >= r.val) {  // the authoritative test of search.hpp:99
        if constexpr (CAPPED) {
          if (++entered > cap) {
            ++st.top;  // (the record goes back: pop() left it in its ring slot)
            stopped = true;
            continue;
          }
        }
        const uint32_t idx = r.x & 0x3FFFFFFFu;
        const bool far_is_right = (r.x & kRecSide) != 0;
        const Node64 nd = nodes[idx];
        const double dv = d_sub(far_is_right ? nd.right_min : nd.left_max, sel3d(nd.axis, q0, q1, q2));
        const double new_off = M::one(dv);
        st.push(kRecUndo | nd.axis, sel3d(nd.axis, o0, o1, o2));
        st.push(kRecUndo | kRecSide, nbd);
        o0 = nd.axis == 0 ? new_off : o0;
        o1 = nd.axis == 1 ? new_off : o1;
        o2 = nd.axis == 2 ? new_off : o2;
        nbd = r.val;
        ref = far_is_right ? nd.right_ref : nd.left_ref;
        break;
      }
    }
  }
}

// The walk of search_nearest_topological (internal/kd_tree_search.hpp:115-229) over double points: which child is
// nearer and the far child's offset both come from the distance of the query coordinate to the two child intervals
// [left_min, left_max] and [right_min, right_max] (t.outer holds the two bounds the euclidean record lacks).  dim <= 3
// (metric_so2: 1, metric_se2_squared: 3), q and off in registers as traverse64_3; records as there.
template <class T, class Policy>
__device__ __forceinline__ void traverse64_topo(const DevTree64& t, double q0, double q1, double q2, Policy& pol, Stack64& st) {
  const Node64* __restrict__ nodes = t.nodes;
  const double2* __restrict__ outer = t.outer;
  const double* __restrict__ pts = t.pts;
  const uint32_t last = t.n_points - 1;
  uint32_t ref = t.root_ref;
  double nbd = 0.0, o0 = 0.0, o1 = 0.0, o2 = 0.0;

  for (;;) {
    while (!(ref & kLeafBit)) {  // search.hpp:158-193
      const Node64 nd = nodes[ref];
      const double2 ob = outer[ref];  // {left_min, right_max}
      const double v = sel3d(nd.axis, q0, q1, q2);
      const double d1 = T::box(ob.x, nd.left_max, v, nd.axis);
      const double d2 = T::box(nd.right_min, ob.y, v, nd.axis);
      const bool go_left = d1 < d2;
      const double new_off = go_left ? d2 : d1;
      const double far_nbd = d_add(d_sub(nbd, sel3d(nd.axis, o0, o1, o2)), new_off);
      if (pol.max() >= far_nbd) st.push(ref | (go_left ? kRecSide : 0u), far_nbd);
      ref = go_left ? nd.left_ref : nd.right_ref;
    }
    {
      const uint32_t lv = ref & 0x7FFFFFFFu;
      const uint32_t begin = lv >> t.cbits;
      const uint32_t count = lv & t.cmask;
      for (uint32_t j = 0; j < count; j += kLeaf64) {
        double px[kLeaf64], py[kLeaf64], pz[kLeaf64];
        int32_t pi[kLeaf64];
#pragma unroll
        for (int u = 0; u < kLeaf64; ++u) {
          const uint32_t pu = begin + j + u <= last ? begin + j + u : last;
          const double4 a = *reinterpret_cast<const double4*>(pts + (uint64_t)pu * kStride64D3);  // {x, y, z, index}
          px[u] = a.x;
          py[u] = a.y;
          pz[u] = a.z;
          pi[u] = (int32_t)__double_as_longlong(a.w);
        }
#pragma unroll
        for (int u = 0; u < kLeaf64; ++u) {
          if (j + u < count) pol.visit(pi[u], T::point(q0, q1, q2, px[u], py[u], pz[u]));
        }
      }
    }
    for (;;) {
      if (st.empty()) return;
      const Rec64 r = st.pop();
      if (r.x & kRecUndo) {
        if (r.x & kRecSide) {
          nbd = r.val;
        } else {
          const uint32_t axis = r.x & 0x3FFFFFFFu;
          o0 = axis == 0 ? r.val : o0;
          o1 = axis == 1 ? r.val : o1;
          o2 = axis == 2 ? r.val : o2;
        }
        continue;
      }
      if (pol.max() >= r.val) {  // search.hpp:199
        const uint32_t idx = r.x & 0x3FFFFFFFu;
        const bool far_is_right = (r.x & kRecSide) != 0;
        const Node64 nd = nodes[idx];
        const double2 ob = outer[idx];
        const double v = sel3d(nd.axis, q0, q1, q2);
        const double new_off = far_is_right ? T::box(nd.right_min, ob.y, v, nd.axis) : T::box(ob.x, nd.left_max, v, nd.axis);
        st.push(kRecUndo | nd.axis, sel3d(nd.axis, o0, o1, o2));
        st.push(kRecUndo | kRecSide, nbd);
        o0 = nd.axis == 0 ? new_off : o0;
        o1 = nd.axis == 1 ? new_off : o1;
        o2 = nd.axis == 2 ? new_off : o2;
        nbd = r.val;
        ref = far_is_right ? nd.right_ref : nd.left_ref;
        break;
      }
    }
  }
}

// One query's whole search under either layout: D3 (dim <= 3) or the run-time-dim form.
template <class M, bool D3, class Policy>
__device__ __forceinline__ void search64(
    const DevTree64& t, const double* __restrict__ queries, uint64_t qi, Policy& pol, Rec64* stack, uint32_t slots) {
  Stack64 st;
  if constexpr (M::kTopo) {  // (the points' missing axes are zero and no split uses them: a 1-D query is padded alike)
    const double* row = queries + qi * t.dim;
    const double q0 = row[0];
    const double q1 = t.dim > 1 ? row[1] : 0.0;
    const double q2 = t.dim > 2 ? row[2] : 0.0;
    st.init(0, 0, stack, slots);
    traverse64_topo<M>(t, q0, q1, q2, pol, st);
  } else if constexpr (D3) {
    const double* row = queries + qi * t.dim;
    const double q0 = row[0];
    // (a missing axis: zero like the points' for sums and maxima, +inf for the minimum of metric_lninf)
    const double q1 = t.dim > 1 ? row[1] : metric64_pad<M>();
    const double q2 = t.dim > 2 ? row[2] : metric64_pad<M>();
    st.init(0, 0, stack, slots);
    traverse64_3<M>(t, q0, q1, q2, pol, st);
  } else {
    LdsDouble *q, *off;
    stage_query64(queries, t.dim, qi, q, off);
    st.init(2, t.dim, stack, slots);
    traverse64<M>(t, q, off, pol, st);
  }
}

// Launch-order entries [q0, q0 + nq) of the batch (entry j is query perm[j], or j itself without a
// permutation; results always land in the row of the query); block b of the launch owns stack
// columns [b * slots * 64, ...).
template <class M, bool D3>
__global__ __launch_bounds__(64) void knn64_kernel(
    DevTree64 t, const double* __restrict__ queries, const uint32_t* __restrict__ perm, uint64_t q0, uint64_t nq,
    uint32_t k, double e_inv, Neighbor64* __restrict__ out, Rec64* __restrict__ stack, uint32_t slots) {
  const uint64_t i = (uint64_t)xcd_runs(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[q0 + i] : q0 + i;
  if (k == 1) {
    Nn64Policy pol;
    pol.best_d = kDblMax;  // search_visitor.hpp:50
    pol.best_i = 0;
    pol.e_inv = e_inv;
    search64<M, D3>(t, queries, qi, pol, stack, slots);
    Neighbor64 nb;
    nb.index = pol.best_i;
    nb.pad_ = 0;
    nb.distance = pol.best_d;
    out[qi] = nb;
    return;
  }
  Knn64Policy pol;
  pol.list = out + qi * k;
  pol.k = k;
  pol.filled = 0;
  pol.worst = kDblMax;
  pol.e_inv = e_inv;
  search64<M, D3>(t, queries, qi, pol, stack, slots);
  pol.end_query();
}

// 1 < k <= K <= 64, k <= n_points (enforced by the caller, kd_tree.hpp:193: every slot below k ends up real).
template <class M, int K, bool D3>
__global__ __launch_bounds__(64) void knn64_reg_kernel(
    DevTree64 t, const double* __restrict__ queries, const uint32_t* __restrict__ perm, uint64_t q0, uint64_t nq,
    uint32_t k, double e_inv, Neighbor64* __restrict__ out, Rec64* __restrict__ stack, uint32_t slots) {
  const uint64_t i = (uint64_t)xcd_runs(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[q0 + i] : q0 + i;
  Knn64RegPolicy<K> pol;
  pol.init(k, e_inv);
  search64<M, D3>(t, queries, qi, pol, stack, slots);
  pol.store(out + qi * k);
}

template <class M, bool FILL, bool D3>
__global__ __launch_bounds__(64) void radius64_kernel(
    DevTree64 t, const double* __restrict__ queries, const uint32_t* __restrict__ perm, uint64_t q0, uint64_t nq,
    double radius, double e_inv, uint64_t* __restrict__ counts, const uint64_t* __restrict__ offsets,
    Neighbor64* __restrict__ out, Rec64* __restrict__ stack, uint32_t slots) {
  const uint64_t i = (uint64_t)xcd_runs(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[q0 + i] : q0 + i;
  Radius64Policy<FILL> pol;
  pol.radius = d_mul(radius, e_inv);  // search_visitor.hpp:265
  pol.e_inv = e_inv;
  pol.count = 0;
  pol.out = FILL ? out + offsets[qi] : nullptr;
  search64<M, D3>(t, queries, qi, pol, stack, slots);
  if (!FILL) counts[qi] = pol.count;
}

// Morton key of each query inside the root box (first three axes), as morton_kernel of
// ptk_kernels.hpp: neighbouring lanes then walk neighbouring leaves.  Only the launch order depends
// on it, never a result.
struct Morton64Box {
  double lo[3], inv[3];
};
// occ (k-NN searches; r06): one byte per cell of the coarse grid the top kMorton64CellBits bits of the 30-bit key
// address -- 1 = the cell holds a tree point.  A search is long where the query sits in EMPTY space (ptk_kernels.hpp,
// CellTable): with the table the top bit of the order key says "the query's cell holds tree points", so the searches
// that will be long start first and run beside the bulk of the batch instead of behind it.  Only the launch order
// depends on it.
constexpr uint32_t kMorton64CellBits = 18;
PTK_GLOBAL __launch_bounds__(kBlock) void morton64_kernel(
    const double* __restrict__ queries, uint32_t dim, uint64_t nq, Morton64Box box, uint32_t drop,
    uint32_t* __restrict__ keys, uint32_t* __restrict__ ids, const uint8_t* __restrict__ occ = nullptr) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nq) return;
  uint32_t key = 0;
  for (uint32_t a = 0; a < 3 && a < dim; ++a) {
    const double f = fmin(fmax((queries[i * dim + a] - box.lo[a]) * box.inv[a], 0.0), 1023.0);
    key |= spread10((uint32_t)f) << a;
  }
  uint32_t out = key >> drop;
  if (occ != nullptr) {
    const uint32_t cheap = occ[key >> (30u - kMorton64CellBits)] != 0 ? 1u : 0u;
    out = (cheap << (29u - drop)) | (out >> 1);  // (the lowest Morton bit makes room: the key keeps its width)
  }
  keys[i] = out;
  ids[i] = (uint32_t)i;
}

// Rows ascending by distance (kd_tree.hpp:263-265: std::sort, ties in unspecified order; here
// ties go by index so the result is deterministic).  One lane per row, heap sort in place.
PTK_GLOBAL __launch_bounds__(kBlock) void sort_rows64_kernel(
    const uint64_t* __restrict__ offsets, uint64_t nq, Neighbor64* __restrict__ out) {
  const uint64_t qi = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (qi >= nq) return;
  Neighbor64* a = out + offsets[qi];
  const uint64_t n = offsets[qi + 1] - offsets[qi];
  if (n < 2) return;
  auto less = [](const Neighbor64& x, const Neighbor64& y) {
    return x.distance < y.distance || (x.distance == y.distance && x.index < y.index);
  };
  auto sift = [&](uint64_t root, uint64_t end) {
    Neighbor64 v = a[root];
    for (;;) {
      uint64_t c = 2 * root + 1;
      if (c >= end) break;
      if (c + 1 < end && less(a[c], a[c + 1])) ++c;
      if (!less(v, a[c])) break;
      a[root] = a[c];
      root = c;
    }
    a[root] = v;
  };
  for (uint64_t s = n / 2; s-- > 0;) sift(s, n);
  for (uint64_t end = n - 1; end > 0; --end) {
    const Neighbor64 top = a[0];
    a[0] = a[end];
    a[end] = top;
    sift(0, end);
  }
}

// ---- box search (kd_tree_search.hpp:238-381), as box_nd_kernel of ptk_kernels_nd.hpp ----------
// Records: pending-right {branch, val = box max of the axis to restore}, undo {kRecUndo | axis,
// val = box min to restore}.  LDS: query min / max and the running node box min / max.
// TOPO: the tree of a topological metric, as box_kernel<.., TOPO> of ptk_kernels.hpp -- on a circle axis (bit of
// s1_mask) a query interval with min > max wraps through the seam (metric_box_map, box.hpp:300-376), and every axis
// uses the four-bound intersection tests of kd_tree_search.hpp:311-327 (the two outer bounds from t.outer).
template <bool FILL, bool TOPO = false>
__global__ __launch_bounds__(64) void box64_kernel(
    DevTree64 t, const double* __restrict__ root, const double* __restrict__ mins,
    const double* __restrict__ maxs, uint64_t b0, uint64_t nb, uint64_t* __restrict__ counts,
    const uint64_t* __restrict__ offsets, int32_t* __restrict__ out, Rec64* __restrict__ stack, uint32_t slots,
    uint32_t s1_mask = 0) {
  const uint64_t i = (uint64_t)xcd_runs(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  if (i >= nb) return;
  const uint64_t bi = b0 + i;
  const uint32_t dim = t.dim;
  LdsDouble* qn = (LdsDouble*)ptk_smem + threadIdx.x;
  LdsDouble* qx = qn + (size_t)dim * 64;
  LdsDouble* mn = qx + (size_t)dim * 64;
  LdsDouble* mx = mn + (size_t)dim * 64;
  for (uint32_t a = 0; a < dim; ++a) {
    qn[a * 64] = mins[bi * dim + a];
    qx[a * 64] = maxs[bi * dim + a];
    mn[a * 64] = root[a];
    mx[a * 64] = root[dim + a];
  }
  const Node64* __restrict__ nodes = t.nodes;
  uint64_t count = 0;
  int32_t* row = FILL ? out + offsets[bi] : nullptr;
  Stack64 st;
  st.init(4, dim, stack, slots);

  auto inside = [&]() {  // query_.contains(box_): both corners inside the closed query box
    bool in = true;
    for (uint32_t a = 0; a < dim; ++a) {
      const double lo = qn[a * 64], hi = qx[a * 64], bl = mn[a * 64], bh = mx[a * 64];
      if (TOPO) {  // segment_r1 / segment_s1::contains of an interval
        const bool wrap = ((s1_mask >> a) & 1u) != 0u && !(lo <= hi);
        in = in && (wrap ? (bl >= lo || bh <= hi) : (lo <= bl && bh <= hi));
      } else {
        in = in && lo <= bl && bl <= hi && lo <= bh && bh <= hi;
      }
    }
    return in;
  };
  auto report_range = [&](uint32_t begin, uint32_t end) {
    if (FILL) {
      for (uint32_t p = begin; p < end; ++p) row[count + (p - begin)] = t.index[p];
    }
    count += end - begin;
  };
  auto report = [&](uint32_t ref) {
    if (ref & kLeafBit) {
      const uint32_t lv = ref & 0x7FFFFFFFu;
      report_range(lv >> t.cbits, (lv >> t.cbits) + (lv & t.cmask));
    } else {
      const uint2 r = t.ranges[ref];
      report_range(r.x, r.y);
    }
  };
  auto scan_leaf = [&](uint32_t ref) {
    const uint32_t lv = ref & 0x7FFFFFFFu;
    const uint32_t begin = lv >> t.cbits;
    const uint32_t n = lv & t.cmask;
    for (uint32_t j = 0; j < n; ++j) {
      const double* p = t.pts + (uint64_t)(begin + j) * t.stride;
      bool in = true;
      for (uint32_t a = 0; a < dim; ++a) {
        const double lo = qn[a * 64], hi = qx[a * 64];
        const bool wrap = TOPO && ((s1_mask >> a) & 1u) != 0u && !(lo <= hi);
        in = in && (wrap ? (p[a] >= lo || p[a] <= hi) : (lo <= p[a] && p[a] <= hi));
      }
      if (in) {
        if (FILL) row[count] = t.index[begin + j];
        ++count;
      }
    }
  };

  uint32_t ref = t.root_ref;
  bool have = true;
  for (;;) {
    if (have) {
      if (ref & kLeafBit) {
        scan_leaf(ref);
        have = false;
      } else {
        const Node64 nd = nodes[ref];
        st.push(ref, mx[nd.axis * 64]);  // the right child comes later
        mx[nd.axis * 64] = nd.left_max;
        bool enter_left = qn[nd.axis * 64] <= nd.left_max;  // intersects_left
        if (TOPO) enter_left = enter_left || qx[nd.axis * 64] >= t.outer[ref].x;  // || query.max >= left_min
        if (inside()) {
          report(nd.left_ref);
          have = false;
        } else if (enter_left) {
          ref = nd.left_ref;
        } else {
          have = false;
        }
      }
      continue;
    }
    if (st.empty()) break;
    const Rec64 r = st.pop();
    if (r.x & kRecUndo) {
      mn[(r.x & 0x3FFFFFFFu) * 64] = r.val;
      continue;
    }
    // Left side of branch r.x is done: restore max, narrow min, do the right side.
    const Node64 nd = nodes[r.x & 0x3FFFFFFFu];
    mx[nd.axis * 64] = r.val;
    st.push(kRecUndo | nd.axis, mn[nd.axis * 64]);
    mn[nd.axis * 64] = nd.right_min;
    bool enter_right = qx[nd.axis * 64] >= nd.right_min;  // intersects_right
    if (TOPO) enter_right = enter_right || qn[nd.axis * 64] <= t.outer[r.x & 0x3FFFFFFFu].y;  // || query.min <= right_max
    if (inside()) {
      report(nd.right_ref);
    } else if (enter_right) {
      ref = nd.right_ref;
      have = true;
    }
  }
  if (!FILL) counts[bi] = count;
}

}  // namespace ptk
