// ptk_kernels_nd.hpp -- gfx950 device code for trees of any dimension (dim > 3).
//
// Same algorithm and the same per-lane replay of the reference visit order as the 3-D
// kernels of ptk_kernels.hpp (see the header there): one query per lane, near child first,
// far child iff `visitor.max() >= node_box_distance`
// (/root/reference/src/pico_tree/pico_tree/internal/kd_tree_search.hpp:52-105), one LIFO of
// 8-byte records per lane (pending / undo_off / undo_nbd).  What changes with the dimension
// is where the per-lane vectors live: the query q[dim] and the per-axis offsets off[dim]
// (search.hpp:111) cannot sit in registers for a run-time dim, so they are staged in LDS,
// laid out [axis][lane] (a wave's access to one axis is one conflict-free ds_read_b32).
// Point coordinates stream from HBM in leaf order (row-major, `dim` floats per point, no
// index indirection) and are accumulated left to right from d = 0 exactly like
// internal::sum (metric.hpp:36-51): d = (((0 + t0) + t1) + ...) with t = (q - p) * (q - p).
//
// Layouts (ptk_encode.hpp, encode_tree_nd):
//   nodes : 16 B per branch {left_max, right_min, left_ref, right_ref}
//   axes  : 4 B per branch, the split axis
//   ref   : bit 31 = leaf; leaf = (begin << cbits) | count; branch = branch index
//   record: x = bit 31 undo | bit 30 (pending: far child is the right one; undo: nbd) |
//               bits 29:0 (pending: branch index; undo_off: axis),  y = float bits

#pragma once

#include "ptk_kernels.hpp"

namespace ptk {

struct DevTreeND {
  const uint4* nodes;
  const uint32_t* axes;
  const float* pts;     // leaf order, row-major
  const int32_t* index; // leaf order: original index of each point
  uint32_t root_ref;
  uint32_t cbits;
  uint32_t cmask;
  uint32_t dim;
  Record* deep_spill;  // as DevTree::deep_spill
  uint32_t deep_cap;
};

constexpr uint32_t kNdIdxMask = 0x3FFFFFFFu;
constexpr uint32_t kNdBatch = 4;  // point coordinates in flight per lane (leaf scan)

typedef PTK_LDS float LdsFloat;

template <class M = MetricL2, class Policy, class StackT>
__device__ __forceinline__ void traverse_nd(
    const DevTreeND& t, LdsFloat* q, LdsFloat* off, uint32_t stride, Policy& pol, StackT& st) {
  const uint4* __restrict__ nodes = t.nodes;
  const uint32_t* __restrict__ axes = t.axes;
  const float* __restrict__ pts = t.pts;
  const int32_t* __restrict__ index = t.index;
  const uint32_t dim = t.dim;
  uint32_t ref = t.root_ref;
  float nbd = 0.0f;

  for (;;) {
    while (!(ref & kLeafBit)) {
      const uint4 nd = nodes[ref];
      const uint32_t axis = axes[ref];
      const float left_max = __uint_as_float(nd.x);
      const float right_min = __uint_as_float(nd.y);
      const float v = q[axis * stride];
      const bool go_left = f_sub(f_sub(f_add(left_max, right_min), v), v) > 0.0f;  // search.hpp:76
      const float dv = f_sub(go_left ? right_min : left_max, v);
      const float new_off = M::one(dv);                                             // :80,84
      const float far_nbd = f_add(f_sub(nbd, off[axis * stride]), new_off);         // :94
      if (pol.max() >= far_nbd) st.push(ref | (go_left ? kRecSide : 0u), far_nbd);
      ref = go_left ? nd.z : nd.w;
    }
    {
      const uint32_t lv = ref & 0x7FFFFFFFu;
      const uint32_t begin = lv >> t.cbits;
      const uint32_t count = lv & t.cmask;
      for (uint32_t j = 0; j < count; ++j) {
        const float* p = pts + (uint64_t)(begin + j) * dim;
        const int32_t pi = index[begin + j];
        float d = metric_init<M>();
        // kNdBatch coordinates are loaded before the first is used (one coordinate per iteration
        // waits for memory dim times per point).  A slot past the last axis re-reads the last
        // coordinate and contributes diff = 0: acc(d, 0) == d exactly for the three metrics
        // (d >= +0: d + 0 * 0, d + |0|, max(d, |0|)).
        for (uint32_t a = 0; a < dim; a += kNdBatch) {
          float pc[kNdBatch], qc[kNdBatch];
#pragma unroll
          for (uint32_t u = 0; u < kNdBatch; ++u) {
            const uint32_t au = a + u < dim ? a + u : dim - 1;
            pc[u] = p[au];
            qc[u] = q[au * stride];
          }
#pragma unroll
          for (uint32_t u = 0; u < kNdBatch; ++u) d = M::acc(d, a + u < dim ? f_sub(qc[u], pc[u]) : metric_pad<M>());
        }
        pol.visit(pi, d);
      }
    }
    for (;;) {
      if (st.empty()) return;
      const Record r = st.pop();
      const float val = __uint_as_float(r.y);
      if (r.x & kRecUndo) {
        if (r.x & kRecSide) {
          nbd = val;
        } else {
          off[(r.x & kNdIdxMask) * stride] = val;
        }
        continue;
      }
      if (pol.max() >= val) {  // the authoritative test of search.hpp:99
        const uint32_t idx = r.x & kNdIdxMask;
        const bool far_is_right = (r.x & kRecSide) != 0;
        const uint4 nd = nodes[idx];
        const uint32_t axis = axes[idx];
        const float plane = far_is_right ? __uint_as_float(nd.y) : __uint_as_float(nd.x);
        const float dv = f_sub(plane, q[axis * stride]);
        const float new_off = M::one(dv);
        st.push(kRecUndo | axis, off[axis * stride]);
        st.push(kRecUndo | kRecSide, nbd);
        off[axis * stride] = new_off;
        nbd = val;
        ref = far_is_right ? nd.w : nd.z;
        break;
      }
    }
  }
}

// LDS of a block: [record ring S][q dim][off dim][k-list k], all [slot][lane].
template <int S>
__device__ __forceinline__ void stage_query_nd(
    const float* __restrict__ queries, uint32_t dim, uint64_t qi, LdsFloat*& q, LdsFloat*& off) {
  LdsFloat* base = (LdsFloat*)(ptk_smem + (size_t)S * 64 * 8);
  q = base + threadIdx.x;
  off = base + (size_t)dim * 64 + threadIdx.x;
  const float* row = queries + qi * dim;
  for (uint32_t a = 0; a < dim; ++a) {
    q[a * 64] = row[a];
    off[a * 64] = 0.0f;  // search.hpp:47
  }
}

// perm: the launch order of the batch (entry i of the launch is query perm[i]; null = as given).
// Results always land in the row of the query.
template <int S, int OVF, bool LIST_LDS, class M = MetricL2>
__global__ __launch_bounds__(64) void knn_nd_kernel(
    DevTreeND t, const float* __restrict__ queries, const uint32_t* __restrict__ perm, uint64_t nq, uint32_t k,
    float e_inv, Neighbor* __restrict__ out) {
  const uint64_t i = (uint64_t)xcd_runs(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[i] : i;
  LdsFloat *q, *off;
  stage_query_nd<S>(queries, t.dim, qi, q, off);
  PTK_STACK(S, OVF, 64, st, t);
  KnnPolicy<LIST_LDS> pol;
  if constexpr (LIST_LDS) {
    pol.list = (LdsWord*)(ptk_smem + (size_t)S * 64 * 8 + (size_t)t.dim * 64 * 8) + threadIdx.x;
    pol.stride = 64;
  } else {
    pol.list = out + qi * k;
    pol.stride = 1;
  }
  pol.k = k;
  pol.filled = 0;
  pol.worst = 3.402823466e+38f;
  pol.e_inv = e_inv;
  pol.out = out;
  traverse_nd<M>(t, q, off, 64u, pol, st);
  pol.end_query((uint32_t)qi);
}

// k <= K <= 32: the k-list in registers (KnnRegPolicy, ptk_kernels.hpp).
template <int K, int S, int OVF, class M = MetricL2>
__global__ __launch_bounds__(64) void knn_nd_reg_kernel(
    DevTreeND t, const float* __restrict__ queries, const uint32_t* __restrict__ perm, uint64_t nq, uint32_t k,
    float e_inv, Neighbor* __restrict__ out) {
  const uint64_t i = (uint64_t)xcd_runs(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[i] : i;
  LdsFloat *q, *off;
  stage_query_nd<S>(queries, t.dim, qi, q, off);
  Record spill[OVF > 0 ? OVF : 1];
  Stack<S, OVF, 64> st;
  st.init((LdsWord*)ptk_smem, threadIdx.x, spill);
  KnnRegPolicy<K> pol;
  pol.init(k, e_inv);
  traverse_nd<M>(t, q, off, 64u, pol, st);
  pol.store(out + qi * k);
}

// perm / n_dev (fill pass only): the batch is the first *n_dev rows listed in perm -- the rows a
// capture could not hold (see radius_kernel in ptk_kernels.hpp).
template <int S, int OVF, bool FILL, class M = MetricL2>
__global__ __launch_bounds__(64) void radius_nd_kernel(
    DevTreeND t, const float* __restrict__ queries, uint64_t nq, float radius, float e_inv,
    uint64_t* __restrict__ counts, const uint64_t* __restrict__ offsets, Neighbor* __restrict__ out,
    const uint32_t* __restrict__ perm = nullptr, const uint32_t* __restrict__ n_dev = nullptr) {
  if (n_dev != nullptr) nq = *n_dev;
  const uint64_t i = (uint64_t)xcd_runs(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[i] : i;
  LdsFloat *q, *off;
  stage_query_nd<S>(queries, t.dim, qi, q, off);
  PTK_STACK(S, OVF, 64, st, t);
  RadiusPolicy<FILL> pol;
  pol.radius = f_mul(radius, e_inv);  // search_visitor.hpp:265
  pol.e_inv = e_inv;
  pol.count = 0;
  pol.out = FILL ? out + offsets[qi] : nullptr;
  traverse_nd<M>(t, q, off, 64u, pol, st);
  if (!FILL) counts[qi] = pol.count;
}

// The count pass that also captures the rows (RadiusCapture, ptk_kernels.hpp): the cursor of the wavefront's log
// lies behind the staged query.
template <int S, int OVF, class M = MetricL2>
__global__ __launch_bounds__(64) void radius_nd_capture_kernel(
    DevTreeND t, const float* __restrict__ queries, const uint32_t* __restrict__ perm, uint64_t nq, float radius,
    float e_inv, uint64_t* __restrict__ counts, RadiusCapture cap) {
  const uint32_t tile = xcd_runs(blockIdx.x, gridDim.x);
  const uint64_t i = (uint64_t)tile * 64 + threadIdx.x;
  if (i >= nq) {
    cap.qids[i] = kLogEnd;
    return;
  }
  const uint64_t qi = perm ? perm[i] : i;
  cap.qids[i] = (uint32_t)qi;
  LdsFloat *q, *off;
  stage_query_nd<S>(queries, t.dim, qi, q, off);
  Record spill[OVF > 0 ? OVF : 1];
  Stack<S, OVF, 64> st;
  st.init((LdsWord*)ptk_smem, threadIdx.x, spill);
  RadiusPolicy<kRadiusCapture> pol;
  pol.radius = f_mul(radius, e_inv);
  pol.e_inv = e_inv;
  pol.count = 0;
  pol.out = cap.chunks;
  pol.counters = cap.counters;
  pol.cursor = (LdsWord*)(ptk_smem + (size_t)S * 64 * 8 + (size_t)t.dim * 64 * 8);
  pol.sub = (blockIdx.x * 0x9E3779B1u) >> 24;
  pol.sub_cap = cap.sub_cap;
  pol.n_static = cap.n_static;
  pol.lane = threadIdx.x;
  if (threadIdx.x == 0) pol.open_log(tile);
  traverse_nd<M>(t, q, off, 64u, pol, st);
  counts[qi] = pol.count;
#if defined(__HIP_DEVICE_COMPILE__)
  __syncthreads();  // (one wavefront: every lane has appended its last group before the header is closed)
#endif
  cap.captured[tile] = pol.close_log() ? 1 : 0;
}

// ---- box search, any dimension ------------------------------------------------------------
// The walk of box_kernel (ptk_kernels.hpp; kd_tree_search.hpp:238-381) with the four per-lane
// vectors -- query min / max and the running node box min / max -- staged in LDS [axis][lane].
// `root` holds the root box (min[dim], max[dim]).  Records: pending-right {branch, val = box max
// of the axis to restore}, undo {kRecUndo | axis, val = box min to restore}.
template <int S, int OVF, bool FILL>
__global__ __launch_bounds__(64) void box_nd_kernel(
    DevTreeND t, const uint2* __restrict__ ranges, const float* __restrict__ root,
    const float* __restrict__ mins, const float* __restrict__ maxs, uint64_t nb,
    uint64_t* __restrict__ counts, const uint64_t* __restrict__ offsets, int32_t* __restrict__ out,
    const uint32_t* __restrict__ perm = nullptr) {
  const uint64_t li = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (li >= nb) return;
  const uint64_t bi = perm ? perm[li] : li;
  const uint32_t dim = t.dim;
  LdsFloat* base = (LdsFloat*)(ptk_smem + (size_t)S * 64 * 8);
  LdsFloat* qn = base + threadIdx.x;
  LdsFloat* qx = qn + (size_t)dim * 64;
  LdsFloat* mn = qx + (size_t)dim * 64;
  LdsFloat* mx = mn + (size_t)dim * 64;
  for (uint32_t a = 0; a < dim; ++a) {
    qn[a * 64] = mins[bi * dim + a];
    qx[a * 64] = maxs[bi * dim + a];
    mn[a * 64] = root[a];
    mx[a * 64] = root[dim + a];
  }
  const uint4* __restrict__ nodes = t.nodes;
  const uint32_t* __restrict__ axes = t.axes;
  uint64_t count = 0;
  int32_t* row = FILL ? out + offsets[bi] : nullptr;

  PTK_STACK(S, OVF, 64, st, t);

  auto inside = [&]() {  // query_.contains(box_), box.hpp: both corners inside the closed query box
    bool in = true;
    for (uint32_t a = 0; a < dim; ++a) {
      const float lo = qn[a * 64], hi = qx[a * 64], bl = mn[a * 64], bh = mx[a * 64];
      in = in && lo <= bl && bl <= hi && lo <= bh && bh <= hi;
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
      const uint2 r = ranges[ref];
      report_range(r.x, r.y);
    }
  };
  auto scan_leaf = [&](uint32_t ref) {
    const uint32_t lv = ref & 0x7FFFFFFFu;
    const uint32_t begin = lv >> t.cbits;
    const uint32_t n = lv & t.cmask;
    for (uint32_t j = 0; j < n; ++j) {
      const float* p = t.pts + (uint64_t)(begin + j) * dim;
      bool in = true;
      for (uint32_t a = 0; a < dim; ++a) in = in && qn[a * 64] <= p[a] && p[a] <= qx[a * 64];
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
        const uint4 nd = nodes[ref];
        const uint32_t axis = axes[ref];
        const float left_max = __uint_as_float(nd.x);
        st.push(ref, mx[axis * 64]);  // the right child comes later
        mx[axis * 64] = left_max;
        if (inside()) {
          report(nd.z);
          have = false;
        } else if (qn[axis * 64] <= left_max) {  // intersects_left
          ref = nd.z;
        } else {
          have = false;
        }
      }
      continue;
    }
    if (st.empty()) break;
    const Record r = st.pop();
    const float val = __uint_as_float(r.y);
    if (r.x & kRecUndo) {
      mn[(r.x & kNdIdxMask) * 64] = val;
      continue;
    }
    // Left side of branch r.x is done: restore max, narrow min, do the right side.
    const uint32_t idx = r.x & kNdIdxMask;
    const uint4 nd = nodes[idx];
    const uint32_t axis = axes[idx];
    mx[axis * 64] = val;
    const float right_min = __uint_as_float(nd.y);
    st.push(kRecUndo | axis, mn[axis * 64]);
    mn[axis * 64] = right_min;
    if (inside()) {
      report(nd.w);
    } else if (qx[axis * 64] >= right_min) {  // intersects_right
      ref = nd.w;
      have = true;
    }
  }
  if (!FILL) counts[bi] = count;
}

}  // namespace ptk
