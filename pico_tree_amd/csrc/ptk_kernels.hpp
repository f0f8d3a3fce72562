// ptk_kernels.hpp -- gfx950 device code of the batched k-NN backend.
//
// One query per lane.  Every lane replays, for its own query, exactly the visit
// sequence of the reference's recursive search
// (/root/reference/src/pico_tree/pico_tree/internal/kd_tree_search.hpp:52-105):
// near child first, far child only if `visitor.max() >= node_box_distance`
// (:99), points of a leaf in index order (:54-59), a candidate accepted only if
// strictly closer (search_visitor.hpp:55,107,141).  That order decides which of
// two equidistant points is reported, so lanes never share or reorder visits;
// the 64 lanes of a wavefront only share instruction issue and, thanks to the
// Morton-ordered batch, cache lines.
//
// Arithmetic is IEEE float32, one rounding per operation, no fused multiply-add
// (compiled with -ffp-contract=off and written with __f*_rn), in the reference's
// association:
//   side test   ((left_max + right_min) - v) - v > 0          search.hpp:76
//   new offset  (plane - v) * (plane - v)                      search.hpp:80,84
//   box dist    (nbd - off[axis]) + new_offset                 search.hpp:94
//   distance    ((dx*dx + dy*dy) + dz*dz), from d = 0          metric.hpp:36-51
//
// Per-lane traversal state is {ref, nbd, off[3]} in registers plus one LIFO of
// 8-byte records.  The newest S records live in an LDS ring ([slot][lane], so a
// wave's access is always conflict-free); when the ring is full its OLDEST
// record is spilled to private scratch, and when it runs empty a batch of the
// most recently spilled records is brought back with independent loads.  The
// first descent pushes one record per level and almost all of them are discarded
// unseen at the end of the query, so the shallow levels are what gets spilled:
//
//   pending  {meta = far-side | axis | parent branch, val = far box distance}
//   undo_off {meta = UNDO | axis,                     val = previous off[axis]}
//   undo_nbd {meta = UNDO | NBD,                      val = previous nbd}
//
// A pending record is pushed when a branch is passed and the far child could
// still matter (`max() >= far distance`, which is safe because max() never
// grows).  When it is popped the registers hold the state of the branch that
// pushed it again (the undo records above it have been consumed), so the stored
// distance IS the reference's node_box_distance for the far child and the pop
// test is the authoritative `max() >= node_box_distance`.  Entering a far child
// re-reads the parent branch (one 16-byte load, ~2 per query) for the child
// reference and the plane, and pushes the two undo records.

#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptk {

// ---- device tree --------------------------------------------------------------
//
// nodes : one 16-byte record per BRANCH {left_max, right_min, left_ref, right_ref}
// pts   : float4 {x, y, z, bits(original index)} in leaf order (z = 0 for dim < 3)
// ref   : bit 31 = leaf.
//           branch ref: bits 30:29 = split axis of the child, bits 28:0 = branch index
//           leaf   ref: bits 30:0  = (begin << cbits) | count
struct DevTree {
  const uint4* nodes;
  const float4* pts;
  uint32_t root_ref;
  uint32_t cbits;
  uint32_t cmask;
  uint32_t n_points;
};

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kBranchIdxMask = 0x1FFFFFFFu;
constexpr uint32_t kRecUndo = 0x80000000u;
constexpr uint32_t kRecSide = 0x40000000u;  // pending: far child is the right one; undo: nbd
constexpr uint32_t kRecIdxMask = 0x0FFFFFFFu;
constexpr int kBlock = 256;     // helper kernels (keys, row sort)
constexpr int kLeafPad = 8;     // readable records past the last point (batched leaf loads)

struct Neighbor {
  int32_t index;
  float distance;
};

__device__ __forceinline__ float f_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float f_sub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float f_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float sel3(uint32_t axis, float a0, float a1, float a2) {
  return axis == 0 ? a0 : (axis == 1 ? a1 : a2);
}

// ---- record stack: ring of S slots in LDS + OVF spill slots in private scratch ----
// Records are numbered 0, 1, 2, ... in push order.  [base, top) is resident in the
// ring (record i at slot i mod S); [0, base) has been spilled (record i at
// ovf[i]).  S is a power of two.
template <int S, int OVF, int BLOCK>
struct Stack {
  static_assert((S & (S - 1)) == 0 && S >= 4, "S must be a power of two");
  static constexpr int kRefill = S / 2 < 8 ? S / 2 : 8;
  uint2* lds;  // this lane's column: slot i at lds[i * BLOCK]
  uint2 ovf[OVF > 0 ? OVF : 1];
  int top;
  int base;
  __device__ __forceinline__ void init(uint2* block_base, int tid) {
    lds = block_base + tid;
    top = 0;
    base = 0;
  }
  __device__ __forceinline__ bool empty() const { return top == 0; }
  __device__ __forceinline__ void push(uint32_t meta, float val) {
    if (top - base == S) {  // ring full: spill the oldest resident record
      if (OVF > 0) ovf[base] = lds[(base & (S - 1)) * BLOCK];
      ++base;
    }
    lds[(top & (S - 1)) * BLOCK] = make_uint2(meta, __float_as_uint(val));
    ++top;
  }
  __device__ __forceinline__ uint2 pop() {
    if (top == base) {  // ring empty, spilled records remain: refill a batch
      if (OVF > 0) {
        uint2 r[kRefill];
#pragma unroll
        for (int i = 0; i < kRefill; ++i) {
          const int idx = base - 1 - i;
          r[i] = ovf[idx >= 0 ? idx : 0];
        }
#pragma unroll
        for (int i = 0; i < kRefill; ++i) {
          const int idx = base - 1 - i;
          if (idx >= 0) lds[(idx & (S - 1)) * BLOCK] = r[i];
        }
      }
      base = base > kRefill ? base - kRefill : 0;
    }
    --top;
    return lds[(top & (S - 1)) * BLOCK];
  }
};

// ---- result policies ------------------------------------------------------------
// max()  : current pruning distance            visit(): one measured point
//
// Every candidate distance is multiplied by e_inv = 1/e before it is compared or
// stored, as the reference's approximate visitors do (search_visitor.hpp:173,216,
// 265).  The exact searches pass e_inv = 1.0f: an IEEE multiplication by one is
// the identity on every float, so one code path serves both bit-exactly.

struct NnPolicy {  // search_visitor.hpp:42-65 / :165-193
  float best_d;
  int32_t best_i;
  float e_inv;
  __device__ __forceinline__ float max() const { return best_d; }
  __device__ __forceinline__ void visit(int32_t idx, float d) {
    d = f_mul(d, e_inv);
    if (best_d > d) {
      best_d = d;
      best_i = idx;
    }
  }
};

// Sorted k-list, slot j of this lane at list[j * stride]; LDS or global memory.
struct KnnPolicy {  // search_visitor.hpp:83-123 / :198-247
  Neighbor* list;
  uint32_t stride;
  uint32_t k;
  uint32_t filled;
  float worst;  // == max(): FLT_MAX until the list is full, then the k-th distance
  float e_inv;
  __device__ __forceinline__ float max() const { return worst; }
  __device__ __forceinline__ void visit(int32_t idx, float d) {
    d = f_mul(d, e_inv);
    if (worst > d) {
      if (filled < k) ++filled;
      uint32_t j = filled - 1;
      // insert_sorted (:24-38): shift while strictly smaller => stable on ties.
      while (j > 0) {
        Neighbor prev = list[(j - 1) * stride];
        if (!(d < prev.distance)) break;
        list[j * stride] = prev;
        --j;
      }
      Neighbor nb;
      nb.index = idx;
      nb.distance = d;
      list[j * stride] = nb;
      if (filled == k) worst = list[(k - 1) * stride].distance;
    }
  }
};

template <bool FILL>
struct RadiusPolicy {  // search_visitor.hpp:127-156 / :252-288
  float radius;  // already scaled by 1/e for the approximate search (:265)
  float e_inv;
  uint64_t count;
  Neighbor* out;  // FILL: first record of this query's row
  __device__ __forceinline__ float max() const { return radius; }
  __device__ __forceinline__ void visit(int32_t idx, float d) {
    d = f_mul(d, e_inv);
    if (radius > d) {  // strict
      if (FILL) {
        Neighbor nb;
        nb.index = idx;
        nb.distance = d;
        out[count] = nb;
      }
      ++count;
    }
  }
};

// ---- the traversal ----------------------------------------------------------------
template <int LEAFB, class Policy, class StackT>
__device__ __forceinline__ void traverse(
    const DevTree& t, float qx, float qy, float qz, Policy& pol, StackT& st) {
  const uint4* __restrict__ nodes = t.nodes;
  const float4* __restrict__ pts = t.pts;
  uint32_t ref = t.root_ref;
  float nbd = 0.0f, off0 = 0.0f, off1 = 0.0f, off2 = 0.0f;

  for (;;) {
    // Down to a leaf through the nearer children.
    while (!(ref & kLeafBit)) {
      const uint32_t idx = ref & kBranchIdxMask;
      const uint32_t axis = (ref >> 29) & 3u;
      const uint4 nd = nodes[idx];
      const float left_max = __uint_as_float(nd.x);
      const float right_min = __uint_as_float(nd.y);
      const float v = sel3(axis, qx, qy, qz);
      const float s = f_sub(f_sub(f_add(left_max, right_min), v), v);
      const bool go_left = s > 0.0f;
      const float plane = go_left ? right_min : left_max;  // the far child's face
      const float dv = f_sub(plane, v);
      const float new_off = f_mul(dv, dv);
      const float far_nbd = f_add(f_sub(nbd, sel3(axis, off0, off1, off2)), new_off);
      if (pol.max() >= far_nbd) {
        st.push(idx | (axis << 28) | (go_left ? kRecSide : 0u), far_nbd);
      }
      ref = go_left ? nd.z : nd.w;
    }

    // Measure the leaf: LEAFB points are fetched per round trip with independent
    // loads (the array is padded, so reading past the leaf is harmless), then
    // visited strictly in index order.
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
            const float dx = f_sub(qx, p[u].x);
            const float dy = f_sub(qy, p[u].y);
            const float dz = f_sub(qz, p[u].z);
            const float d = f_add(f_add(f_mul(dx, dx), f_mul(dy, dy)), f_mul(dz, dz));
            pol.visit(__float_as_int(p[u].w), d);
          }
        }
      }
    }

    // Back up to the next far child still worth entering.
    for (;;) {
      if (st.empty()) return;
      const uint2 r = st.pop();
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
      if (pol.max() >= val) {
        const uint32_t idx = r.x & kRecIdxMask;
        const uint32_t axis = (r.x >> 28) & 3u;
        const bool far_is_right = (r.x & kRecSide) != 0;
        const uint4 nd = nodes[idx];
        const float plane = far_is_right ? __uint_as_float(nd.y) : __uint_as_float(nd.x);
        const float dv = f_sub(plane, sel3(axis, qx, qy, qz));
        const float new_off = f_mul(dv, dv);
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

// Blocks are dealt round-robin to the 8 XCDs; give each XCD one contiguous
// eighth of the (spatially sorted) batch so its private L2 only ever sees one
// region of the tree.  Pure performance: any mapping is correct.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t nb) {
  const uint32_t per = nb >> 3;  // blocks per XCD in the evenly divisible part
  const uint32_t even = per << 3;
  if (b >= even) return b;  // tail blocks keep their index
  return (b & 7u) * per + (b >> 3);
}

__device__ __forceinline__ void load_query(
    const float* __restrict__ q, uint32_t dim, uint64_t qi, float& x, float& y, float& z) {
  const float* p = q + qi * dim;
  x = p[0];
  y = dim > 1 ? p[1] : 0.0f;
  z = dim > 2 ? p[2] : 0.0f;
}

extern __shared__ __attribute__((aligned(16))) unsigned char ptk_smem[];

// ---- k = 1 ---------------------------------------------------------------------------
template <int S, int OVF, int BLOCK, int LEAFB>
__global__ __launch_bounds__(BLOCK) void knn1_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim,
    const uint32_t* __restrict__ perm, uint64_t nq, float e_inv, Neighbor* __restrict__ out) {
  const uint32_t tile = xcd_tile(blockIdx.x, gridDim.x);
  const uint64_t i = (uint64_t)tile * BLOCK + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[i] : i;
  float qx, qy, qz;
  load_query(queries, dim, qi, qx, qy, qz);

  Stack<S, OVF, BLOCK> st;
  st.init(reinterpret_cast<uint2*>(ptk_smem), threadIdx.x);
  NnPolicy pol;
  pol.best_d = 3.402823466e+38f;
  pol.best_i = 0;
  pol.e_inv = e_inv;
  traverse<LEAFB>(t, qx, qy, qz, pol, st);

  Neighbor nb;
  nb.index = pol.best_i;
  nb.distance = pol.best_d;
  out[qi] = nb;
}

// ---- general k -------------------------------------------------------------------------
// LIST_LDS: the k-list lives in LDS behind the stack ([slot][lane]) and is copied
// to the output row at the end; otherwise the output row itself is the list.
template <int S, int OVF, int BLOCK, int LEAFB, bool LIST_LDS>
__global__ __launch_bounds__(BLOCK) void knn_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim,
    const uint32_t* __restrict__ perm, uint64_t nq, uint32_t k, float e_inv,
    Neighbor* __restrict__ out) {
  const uint32_t tile = xcd_tile(blockIdx.x, gridDim.x);
  const uint64_t i = (uint64_t)tile * BLOCK + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[i] : i;
  float qx, qy, qz;
  load_query(queries, dim, qi, qx, qy, qz);

  Stack<S, OVF, BLOCK> st;
  st.init(reinterpret_cast<uint2*>(ptk_smem), threadIdx.x);
  KnnPolicy pol;
  if (LIST_LDS) {
    pol.list = reinterpret_cast<Neighbor*>(ptk_smem + (size_t)S * BLOCK * 8) + threadIdx.x;
    pol.stride = BLOCK;
  } else {
    pol.list = out + qi * k;
    pol.stride = 1;
  }
  pol.k = k;
  pol.filled = 0;
  pol.worst = 3.402823466e+38f;
  pol.e_inv = e_inv;
  traverse<LEAFB>(t, qx, qy, qz, pol, st);

  if (LIST_LDS) {
    Neighbor* row = out + qi * k;
    for (uint32_t j = 0; j < pol.filled; ++j) row[j] = pol.list[j * BLOCK];
  }
  if (pol.filled < k) {  // k > reachable points: mirror the reference's sentinel (:102)
    Neighbor nb;
    nb.index = 0;
    nb.distance = 3.402823466e+38f;
    out[qi * k + (k - 1)] = nb;
  }
}

// ---- radius: count pass and fill pass ------------------------------------------------------
template <int S, int OVF, int BLOCK, int LEAFB, bool FILL>
__global__ __launch_bounds__(BLOCK) void radius_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim,
    const uint32_t* __restrict__ perm, uint64_t nq, float radius, float e_inv,
    uint64_t* __restrict__ counts, const uint64_t* __restrict__ offsets,
    Neighbor* __restrict__ out) {
  const uint32_t tile = xcd_tile(blockIdx.x, gridDim.x);
  const uint64_t i = (uint64_t)tile * BLOCK + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[i] : i;
  float qx, qy, qz;
  load_query(queries, dim, qi, qx, qy, qz);

  Stack<S, OVF, BLOCK> st;
  st.init(reinterpret_cast<uint2*>(ptk_smem), threadIdx.x);
  RadiusPolicy<FILL> pol;
  pol.radius = f_mul(radius, e_inv);  // search_visitor.hpp:265
  pol.e_inv = e_inv;
  pol.count = 0;
  pol.out = FILL ? out + offsets[qi] : nullptr;
  traverse<LEAFB>(t, qx, qy, qz, pol, st);
  if (!FILL) counts[qi] = pol.count;
}

// Sorts every row ascending by distance (heap sort, in place, one row per lane).
// std::sort in the reference is unstable, so the order among equal distances is
// unspecified on both sides.
__global__ __launch_bounds__(kBlock) void sort_rows_kernel(
    uint64_t nq, const uint64_t* __restrict__ offsets, Neighbor* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nq) return;
  Neighbor* a = out + offsets[i];
  const uint64_t n = offsets[i + 1] - offsets[i];
  if (n < 2) return;
  auto sift = [&](uint64_t root, uint64_t end) {
    Neighbor v = a[root];
    for (;;) {
      uint64_t c = 2 * root + 1;
      if (c >= end) break;
      if (c + 1 < end && a[c].distance < a[c + 1].distance) ++c;
      if (!(v.distance < a[c].distance)) break;
      a[root] = a[c];
      root = c;
    }
    a[root] = v;
  };
  for (uint64_t s = n / 2; s-- > 0;) sift(s, n);
  for (uint64_t e = n - 1; e > 0; --e) {
    Neighbor top = a[0];
    a[0] = a[e];
    a[e] = top;
    sift(0, e);
  }
}

// ---- batch ordering ------------------------------------------------------------------------
// 30-bit Morton key of each query inside the tree's root box (clamped), plus the
// identity permutation to be sorted along with it.
__device__ __forceinline__ uint32_t spread10(uint32_t x) {
  x &= 0x3FFu;
  x = (x | (x << 16)) & 0x030000FFu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}

__global__ __launch_bounds__(kBlock) void morton_kernel(
    const float* __restrict__ queries, uint32_t dim, uint64_t nq, float3 lo, float3 inv,
    uint32_t* __restrict__ keys, uint32_t* __restrict__ ids) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nq) return;
  float x, y, z;
  load_query(queries, dim, i, x, y, z);
  const float fx = fminf(fmaxf((x - lo.x) * inv.x, 0.0f), 1023.0f);
  const float fy = fminf(fmaxf((y - lo.y) * inv.y, 0.0f), 1023.0f);
  const float fz = fminf(fmaxf((z - lo.z) * inv.z, 0.0f), 1023.0f);
  keys[i] = spread10((uint32_t)fx) | (spread10((uint32_t)fy) << 1) | (spread10((uint32_t)fz) << 2);
  ids[i] = (uint32_t)i;
}

// Inclusive-to-exclusive helper for the radius offsets: offsets[0] = 0 is written
// by the host-side scan call; see ptk_backend.hip.

}  // namespace ptk
