// ptk_kernels_topo.hpp -- the searches of the reference's TOPOLOGICAL metrics on the device:
// metric_so2 (points on the circle S1 = [0, 1] / 0 ~ 1, dim 1) and metric_se2_squared (planar
// poses x, y, angle in [0, 1]; dim 3)  --  /root/reference/src/pico_tree/pico_tree/metric.hpp:186-257.
//
// The reference searches these trees with search_nearest_topological
// (internal/kd_tree_search.hpp:115-229): the same depth-first walk and the same incremental box
// distance as the euclidean search, but which child is nearer is decided from the distance of the
// query coordinate to BOTH child intervals -- [left_min, left_max] and [right_min, right_max],
// wrap-around included on a circular axis -- and the far child's offset is that interval
// distance.  Hence the four bounds per branch (kd_tree_node_topological, kd_tree_node.hpp:56-67):
// the device keeps the euclidean 16-byte record and a second array `outer` = {left_min,
// right_max} per branch (DevTree::outer), loaded together with it.
//
// Everything else -- one query per lane, the 8-byte record stack with its LDS ring, the result
// policies, one rounding per reference operation -- is ptk_kernels.hpp's.

#pragma once

#include "ptk_kernels.hpp"

namespace ptk {

// distance.hpp:26-29  s1_distance: d = |x - y|, min(d, 1 - d)
__device__ __forceinline__ float s1_distance(float x, float y) {
  const float d = fabsf(f_sub(x, y));
  const float w = f_sub(1.0f, d);
  return w < d ? w : d;  // std::min(d, 1 - d)
}
// segment.hpp:38-46  segment_r1::distance
__device__ __forceinline__ float seg_r1_distance(float mn, float mx, float x) {
  return x < mn ? f_sub(mn, x) : (x > mx ? f_sub(x, mx) : 0.0f);
}
// segment.hpp:77-99  segment_s1::distance (a segment with min > max wraps through 0 ~ 1)
__device__ __forceinline__ float seg_s1_distance(float mn, float mx, float x) {
  const float a = s1_distance(x, mn), b = s1_distance(x, mx);
  const float m = b < a ? b : a;  // std::min(a, b)
  if (mn <= mx) return (x < mn || x > mx) ? m : 0.0f;
  return (x < mx || x > mn) ? 0.0f : m;
}

struct TopoSO2 {  // metric_so2, metric.hpp:197-220
  // box distance of one child interval along `axis`: metric_(segment distance) = |d|
  __device__ __forceinline__ static float box(float mn, float mx, float v, uint32_t) {
    return fabsf(seg_s1_distance(mn, mx, v));
  }
  __device__ __forceinline__ static float point(float qx, float, float, const float4& p) {
    return s1_distance(qx, p.x);
  }
};
struct TopoSE2 {  // metric_se2_squared, metric.hpp:228-257: axes 0, 1 on the line, axis 2 on the circle
  __device__ __forceinline__ static float box(float mn, float mx, float v, uint32_t axis) {
    const float d = axis < 2u ? seg_r1_distance(mn, mx, v) : seg_s1_distance(mn, mx, v);
    return f_mul(d, d);
  }
  __device__ __forceinline__ static float point(float qx, float qy, float qz, const float4& p) {
    const float dx = f_sub(qx, p.x), dy = f_sub(qy, p.y);
    const float s = s1_distance(qz, p.z);
    return f_add(f_add(f_mul(dx, dx), f_mul(dy, dy)), f_mul(s, s));  // sum over x, y from 0, + squared_s1
  }
};

template <int LEAFB, class T, class Policy, class StackT>
__device__ __forceinline__ void traverse_topo(
    const DevTree& t, float qx, float qy, float qz, Policy& pol, StackT& st) {
  const uint4* __restrict__ nodes = t.nodes;
  const float2* __restrict__ outer = t.outer;
  const float4* __restrict__ pts = t.pts;
  uint32_t ref = t.root_ref;
  float nbd = 0.0f, off0 = 0.0f, off1 = 0.0f, off2 = 0.0f;
  for (;;) {
    while (!(ref & kLeafBit)) {  // search.hpp:158-193
      const uint32_t idx = ref & kBranchIdxMask;
      const uint32_t axis = (ref >> 29) & 3u;
      const uint4 nd = nodes[idx];
      const float2 ob = outer[idx];  // {left_min, right_max}
      const float v = sel3(axis, qx, qy, qz);
      const float d1 = T::box(ob.x, __uint_as_float(nd.x), v, axis);
      const float d2 = T::box(__uint_as_float(nd.y), ob.y, v, axis);
      const bool go_left = d1 < d2;
      const float new_off = go_left ? d2 : d1;
      const float far_nbd = f_add(f_sub(nbd, sel3(axis, off0, off1, off2)), new_off);
      if (pol.max() >= far_nbd) st.push(idx | (axis << 28) | (go_left ? kRecSide : 0u), far_nbd);
      ref = go_left ? nd.z : nd.w;
    }
    {
      const uint32_t lv = ref & 0x7FFFFFFFu;
      const uint32_t begin = lv >> t.cbits;
      const uint32_t count = lv & t.cmask;
      for (uint32_t j = 0; j < count; j += LEAFB) {
        float4 p[LEAFB];
#pragma unroll
        for (int u = 0; u < LEAFB; ++u) p[u] = pts[begin + j + u];
#pragma unroll
        for (int u = 0; u < LEAFB; ++u) {
          if (j + u < count) {
            PTK_KEEP4(p[u]);
            pol.visit(__float_as_int(p[u].w), T::point(qx, qy, qz, p[u]));
          }
        }
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
          const uint32_t axis = (r.x >> 28) & 3u;
          off0 = axis == 0 ? val : off0;
          off1 = axis == 1 ? val : off1;
          off2 = axis == 2 ? val : off2;
        }
        continue;
      }
      if (pol.max() >= val) {  // search.hpp:199
        const uint32_t idx = r.x & kRecIdxMask;
        const uint32_t axis = (r.x >> 28) & 3u;
        const bool far_is_right = (r.x & kRecSide) != 0;
        const uint4 nd = nodes[idx];
        const float2 ob = outer[idx];
        const float v = sel3(axis, qx, qy, qz);
        const float new_off = far_is_right ? T::box(__uint_as_float(nd.y), ob.y, v, axis)
                                           : T::box(ob.x, __uint_as_float(nd.x), v, axis);
        st.push(kRecUndo | (axis << 28), sel3(axis, off0, off1, off2));
        st.push(kRecUndo | kRecSide, nbd);
        off0 = axis == 0 ? new_off : off0;
        off1 = axis == 1 ? new_off : off1;
        off2 = axis == 2 ? new_off : off2;
        nbd = val;
        ref = far_is_right ? nd.w : nd.z;
        break;
      }
    }
  }
}

// k <= K <= 32: the k-list in registers (KnnRegPolicy); larger k: the list in the output row.
template <int K, int S, int OVF, class T>
__global__ __launch_bounds__(64) void knn_topo_reg_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim, const uint32_t* __restrict__ perm, uint64_t nq,
    uint32_t k, float e_inv, Neighbor* __restrict__ out) {
  const uint64_t i = (uint64_t)xcd_runs(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[i] : i;
  float qx, qy, qz;
  load_query(queries, dim, qi, qx, qy, qz);
  PTK_STACK(S, OVF, 64, st, t);
  KnnRegPolicy<K> pol;
  pol.init(k, e_inv);
  traverse_topo<4, T>(t, qx, qy, qz, pol, st);
  pol.store(out + qi * k);
}

template <int S, int OVF, class T>
__global__ __launch_bounds__(64) void knn_topo_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim, const uint32_t* __restrict__ perm, uint64_t nq,
    uint32_t k, float e_inv, Neighbor* __restrict__ out) {
  const uint64_t i = (uint64_t)xcd_runs(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[i] : i;
  float qx, qy, qz;
  load_query(queries, dim, qi, qx, qy, qz);
  PTK_STACK(S, OVF, 64, st, t);
  KnnPolicy<false> pol;
  pol.list = out + qi * k;
  pol.stride = 1;
  pol.k = k;
  pol.filled = 0;
  pol.worst = 3.402823466e+38f;
  pol.e_inv = e_inv;
  pol.out = out;
  traverse_topo<4, T>(t, qx, qy, qz, pol, st);
  pol.end_query((uint32_t)qi);
}

template <int S, int OVF, bool FILL, class T>
__global__ __launch_bounds__(64) void radius_topo_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim, const uint32_t* __restrict__ perm, uint64_t nq,
    float radius, float e_inv, uint64_t* __restrict__ counts, const uint64_t* __restrict__ offsets,
    Neighbor* __restrict__ out) {
  const uint64_t i = (uint64_t)xcd_runs(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[i] : i;
  float qx, qy, qz;
  load_query(queries, dim, qi, qx, qy, qz);
  PTK_STACK(S, OVF, 64, st, t);
  RadiusPolicy<FILL ? kRadiusFill : kRadiusCount> pol;
  pol.radius = f_mul(radius, e_inv);
  pol.e_inv = e_inv;
  pol.count = 0;
  pol.out = FILL ? out + offsets[qi] : nullptr;
  traverse_topo<4, T>(t, qx, qy, qz, pol, st);
  if (!FILL) counts[qi] = pol.count;
}

}  // namespace ptk
