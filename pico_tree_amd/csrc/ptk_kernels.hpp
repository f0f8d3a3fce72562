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

#include <type_traits>
#include <utility>

// A kernel that is not a template is defined in a header several translation units include (ptk_backend_core.hpp):
// internal linkage, so every unit that launches it has its own and the units that do not emit nothing.
#ifndef PTK_GLOBAL
#define PTK_GLOBAL static __global__
#endif

namespace ptk {

// ---- device tree --------------------------------------------------------------
//
// nodes : one 16-byte record per BRANCH {left_max, right_min, left_ref, right_ref}
// pts   : float4 {x, y, z, bits(original index)} in leaf order (z = 0 for dim < 3)
// ref   : bit 31 = leaf.
//           branch ref: bits 30:29 = split axis of the child, bits 28:0 = branch index
//           leaf   ref: bits 30:0  = (begin << cbits) | count
struct Record;
struct DevTree {
  const uint4* nodes;
  const float4* pts;
  uint32_t root_ref;
  uint32_t cbits;
  uint32_t cmask;
  uint32_t n_points;
  // Trees deeper than the private spill classes hold (thousands of coincident points peel one
  // level each): kernels instantiated with OVF < 0 spill to this HBM block, deep_cap records per
  // lane of the launch (set per launch by the backend; null otherwise).
  Record* deep_spill;
  uint32_t deep_cap;
  // Topological metrics only (ptk_kernels_topo.hpp): {left_min, right_max} per branch, the two
  // bounds the 16-byte record does not hold; null otherwise.
  const float2* outer;
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
struct alignas(16) RowPair {  // two neighbours of a row, one 16-byte store
  unsigned long long a, b;
};

// "All four words of this float4 are used here" (no instruction): keeps a 16-byte record load whole.
#if defined(__HIP_DEVICE_COMPILE__)
#define PTK_KEEP4(P) asm volatile("" ::"v"((P).x), "v"((P).y), "v"((P).z), "v"((P).w))
#define PTK_SCALAR(X) asm volatile("" : "+v"(X))
#else
#define PTK_KEEP4(P) ((void)0)
#define PTK_SCALAR(X) ((void)0)
#endif

// LDS pointers carry their address space explicitly so that every stack / k-list access
// is a ds_read / ds_write; a generic pointer makes hipcc fall back to flat_* accesses,
// which go through the vector-memory pipe and cost hundreds of cycles each.
#define PTK_LDS __attribute__((address_space(3)))
// (C++ copy-assignment of class types is not defined across address spaces, so LDS
// slots are plain 64-bit words: one ds_write_b64 / ds_read_b64 each.)
typedef PTK_LDS unsigned long long LdsWord;
struct Record {  // one stack entry: {meta, float bits}
  uint32_t x;
  uint32_t y;
};
__device__ __forceinline__ unsigned long long pack_record(Record r) {
  return (unsigned long long)r.x | ((unsigned long long)r.y << 32);
}
__device__ __forceinline__ Record unpack_record(unsigned long long w) {
  Record r;
  r.x = (uint32_t)w;
  r.y = (uint32_t)(w >> 32);
  return r;
}
__device__ __forceinline__ unsigned long long pack_neighbor(Neighbor n) {
  return (unsigned long long)(uint32_t)n.index | ((unsigned long long)__float_as_uint(n.distance) << 32);
}
__device__ __forceinline__ Neighbor unpack_neighbor(unsigned long long w) {
  Neighbor n;
  n.index = (int32_t)(uint32_t)w;
  n.distance = __uint_as_float((uint32_t)(w >> 32));
  return n;
}

// min() on a 32-bit LDS word shared by the lanes of a wavefront (ds_min_u32).
#if defined(__HIPCC__)
__device__ __forceinline__ void lds_min_u32(PTK_LDS uint32_t* p, uint32_t v) {
  __hip_atomic_fetch_min((uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#else
inline void lds_min_u32(uint32_t* p, uint32_t v) {
  if (v < *p) *p = v;
}
#endif

// += on a 32-bit LDS word shared by the threads of a block (ds_add_u32).
#if defined(__HIPCC__)
__device__ __forceinline__ void lds_add_u32(PTK_LDS uint32_t* p, uint32_t v) {
  __hip_atomic_fetch_add((uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#else
inline void lds_add_u32(uint32_t* p, uint32_t v) { *p += v; }
#endif

__device__ __forceinline__ float f_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float f_sub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float f_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float sel3(uint32_t axis, float a0, float a1, float a2) {
  return axis == 0 ? a0 : (axis == 1 ? a1 : a2);
}

// ---- metrics ---------------------------------------------------------------------------------
// The reference's euclidean search is generic over the metric (metric.hpp:72-150): `one(x)` is the
// one-dimensional form used for the box offsets (search.hpp:80,84), `acc` adds one coordinate to a
// point distance that starts at 0 (internal::sum, metric.hpp:36-51; metric_lpinf: std::max from 0).
// `init` is what a point distance starts from and `pad` the coordinate difference of an axis the
// space does not have (dim < 3 in the 3-D kernels, the tail of a batch of coordinates elsewhere):
// it must leave the distance as it is.
struct MetricL2 {  // metric_l2_squared
  static constexpr bool kMin = false;
  __device__ __forceinline__ static float one(float x) { return f_mul(x, x); }
  __device__ __forceinline__ static float acc(float d, float diff) { return f_add(d, f_mul(diff, diff)); }
};
struct MetricL1 {  // metric_l1
  static constexpr bool kMin = false;
  __device__ __forceinline__ static float one(float x) { return fabsf(x); }
  __device__ __forceinline__ static float acc(float d, float diff) { return f_add(d, fabsf(diff)); }
};
struct MetricLInf {  // metric_lpinf
  static constexpr bool kMin = false;
  __device__ __forceinline__ static float one(float x) { return fabsf(x); }
  __device__ __forceinline__ static float acc(float d, float diff) {
    const float a = fabsf(diff);
    return d < a ? a : d;  // std::max(d, a)
  }
};
// metric_lninf (metric.hpp:157-186): d = std::min(d, |x - y|) from the largest float, so the first
// coordinate's |x - y| is `one`; an axis that does not exist must contribute +inf (kMin: the
// kernels hand such a query coordinate over as +inf, see pad_query()).
struct MetricLNInf {
  static constexpr bool kMin = true;
  __device__ __forceinline__ static float one(float x) { return fabsf(x); }
  __device__ __forceinline__ static float acc(float d, float diff) {
    const float a = fabsf(diff);
    return a < d ? a : d;  // std::min(d, a)
  }
};
template <class M>
__device__ __forceinline__ float metric_init() {  // what internal::sum / the min and max loops start from
  return M::kMin ? 3.402823466e+38f : 0.0f;
}
template <class M>
__device__ __forceinline__ float metric_pad() {  // difference of a coordinate that does not exist
  return M::kMin ? __uint_as_float(0x7F800000u) : 0.0f;
}
// Three coordinates at once: acc(acc(acc(0, dx), dy), dz) without the exact no-op 0 + x.
template <class M>
__device__ __forceinline__ float point_distance3(float dx, float dy, float dz) {
  return M::acc(M::acc(M::one(dx), dy), dz);
}

// ---- record stack: ring of S slots in LDS + OVF spill slots in private scratch ----
// Records are numbered 0, 1, 2, ... in push order.  [base, top) is resident in the
// ring (record i at slot i mod S); [0, base) has been spilled (record i at
// ovf[i]).  S is a power of two.  OVF > 0: that many private slots; OVF == 0: no spill (the ring
// always suffices); OVF < 0: the spill slots are in HBM (PTK_STACK, DevTree::deep_spill).
template <int S, int OVF, int BLOCK>
struct Stack {
  static_assert(S >= 4, "ring too small");
  static constexpr int kRefill = S / 2 < 8 ? S / 2 : 8;
  // Ring slot of record i (a power of two is a mask; other sizes a multiply-shift).
  static __device__ __forceinline__ int slot(int i) {
    return (S & (S - 1)) == 0 ? (i & (S - 1)) : (int)((uint32_t)i % (uint32_t)S);
  }
  // The spill array is owned by the kernel body and only referenced here: if it were
  // a member, the whole struct (top and base included) would live in scratch memory.
  LdsWord* lds;  // this lane's column: slot i at lds[i * BLOCK]
  Record* ovf;   // OVF private-scratch slots
  int top;
  int base;
  __device__ __forceinline__ void init(LdsWord* block_base, int tid, Record* spill) {
    lds = block_base + tid;
    ovf = spill;
    top = 0;
    base = 0;
  }
  __device__ __forceinline__ bool empty() const { return top == 0; }
  __device__ __forceinline__ void push(uint32_t meta, float val) {
    if (top - base == S) {  // ring full: spill the oldest resident record
      if (OVF != 0) ovf[base] = unpack_record(lds[slot(base) * BLOCK]);
      ++base;
    }
    Record rec;
    rec.x = meta;
    rec.y = __float_as_uint(val);
    lds[slot(top) * BLOCK] = pack_record(rec);
    ++top;
  }
  static constexpr int kUnwind = 8;  // records examined per turn (4 / 12: slower, profiles/r03_notes.txt item 13)
  // The newest records, rr[0] the newest; returns how many are valid (>= 1 unless empty).  They
  // stay on the stack until drop().
  __device__ __forceinline__ int peek(Record (&rr)[kUnwind]) {
    if (top == base) {  // ring empty, spilled records remain: bring a batch back
      if (OVF != 0) {
        Record r[kRefill];
#pragma unroll
        for (int i = 0; i < kRefill; ++i) {
          const int idx = base - 1 - i;
          r[i] = ovf[idx >= 0 ? idx : 0];
        }
#pragma unroll
        for (int i = 0; i < kRefill; ++i) {
          const int idx = base - 1 - i;
          if (idx >= 0) lds[slot(idx) * BLOCK] = pack_record(r[i]);
        }
      }
      base = base > kRefill ? base - kRefill : 0;
    }
#pragma unroll
    for (int i = 0; i < kUnwind; ++i) rr[i] = unpack_record(lds[slot(top - 1 - i) * BLOCK]);
    const int resident = top - base;
    return resident < kUnwind ? resident : kUnwind;
  }
  __device__ __forceinline__ void drop(int n) { top -= n; }
  __device__ __forceinline__ Record pop() {
    if (top == base) {  // ring empty, spilled records remain: refill a batch
      if (OVF != 0) {
        Record r[kRefill];
#pragma unroll
        for (int i = 0; i < kRefill; ++i) {
          const int idx = base - 1 - i;
          r[i] = ovf[idx >= 0 ? idx : 0];
        }
#pragma unroll
        for (int i = 0; i < kRefill; ++i) {
          const int idx = base - 1 - i;
          if (idx >= 0) lds[slot(idx) * BLOCK] = pack_record(r[i]);
        }
      }
      base = base > kRefill ? base - kRefill : 0;
    }
    --top;
    return unpack_record(lds[slot(top) * BLOCK]);
  }
};

// Declares the record stack `st` of a kernel: private spill slots, or (OVF < 0) this lane's run of
// TREE.deep_cap records in the launch's HBM spill block.
#define PTK_STACK(S, OVF, BLOCK, st, TREE)                                                                  \
  Record spill_[(OVF) > 0 ? (OVF) : 1];                                                                     \
  Stack<S, OVF, BLOCK> st;                                                                                  \
  st.init((LdsWord*)ptk_smem, threadIdx.x,                                                                  \
          (OVF) < 0 ? (TREE).deep_spill + ((uint64_t)blockIdx.x * (BLOCK) + threadIdx.x) * (TREE).deep_cap : spill_)

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
  Neighbor* out;  // result rows, indexed by original query (end_query)
  __device__ __forceinline__ void begin_query(uint32_t) {
    best_d = 3.402823466e+38f;
    best_i = 0;
  }
  __device__ __forceinline__ void end_query(uint32_t qi) {
    Neighbor nb;
    nb.index = best_i;
    nb.distance = best_d;
    out[qi] = nb;
  }
  __device__ __forceinline__ float max() const { return best_d; }
  // No later point can be accepted (a candidate must be STRICTLY nearer, search_visitor.hpp:55, and no distance is
  // below zero): the search may stop here, its answer is final.
  __device__ __forceinline__ bool settled() const { return best_d == 0.0f; }
  __device__ __forceinline__ void visit(int32_t idx, float d) {
    d = f_mul(d, e_inv);
    if (best_d > d) {
      best_d = d;
      best_i = idx;
    }
  }
};

// Sorted k-list, slot j of this lane at list[j * stride]; in LDS, or the output row itself.
template <bool LIST_LDS>
struct KnnPolicy {  // search_visitor.hpp:83-123 / :198-247
  using ListPtr = typename std::conditional<LIST_LDS, LdsWord*, Neighbor*>::type;
  ListPtr list;
  __device__ __forceinline__ Neighbor get(uint32_t j) const {
    if constexpr (LIST_LDS) {
      return unpack_neighbor(list[j * stride]);
    } else {
      return list[j * stride];
    }
  }
  __device__ __forceinline__ void put(uint32_t j, Neighbor n) {
    if constexpr (LIST_LDS) {
      list[j * stride] = pack_neighbor(n);
    } else {
      list[j * stride] = n;
    }
  }
  uint32_t stride;
  uint32_t k;
  uint32_t filled;
  float worst;  // == max(): FLT_MAX until the list is full, then the k-th distance
  float e_inv;
  Neighbor* out;      // all result rows (end_query)
  __device__ __forceinline__ void begin_query(uint32_t qi) {
    filled = 0;
    worst = 3.402823466e+38f;
    if constexpr (!LIST_LDS) list = out + (uint64_t)qi * k;
  }
  __device__ __forceinline__ void end_query(uint32_t qi) {
    Neighbor* row = out + (uint64_t)qi * k;
    if constexpr (LIST_LDS) {
      for (uint32_t j = 0; j < filled; ++j) row[j] = get(j);
    }
    if (filled < k) {  // fewer reachable points than k: the reference's sentinel (:102)
      Neighbor nb;
      nb.index = 0;
      nb.distance = 3.402823466e+38f;
      row[k - 1] = nb;
    }
  }
  __device__ __forceinline__ float max() const { return worst; }
  __device__ __forceinline__ void visit(int32_t idx, float d) {
    d = f_mul(d, e_inv);
    if (worst > d) {
      if (filled < k) ++filled;
      uint32_t j = filled - 1;
      // insert_sorted (:24-38): shift while strictly smaller => stable on ties.
      while (j > 0) {
        Neighbor prev = get(j - 1);
        if (!(d < prev.distance)) break;
        put(j, prev);
        --j;
      }
      Neighbor nb;
      nb.index = idx;
      nb.distance = d;
      put(j, nb);
      if (filled == k) worst = get(k - 1).distance;
    }
  }
};

// Sorted k-list in REGISTERS for k <= K <= 32 (profiles/r01f_config3_*: the LDS list above
// makes knn = 16 seven times slower per query than knn = 1 -- every accepted candidate walks a
// divergent shift loop of ds_read / ds_write pairs).  Here an insertion is K branch-free steps on registers:
//   new[j] = d < old[j-1] ? old[j-1] : (d < old[j] ? d : old[j])
// which is insert_sorted (search_visitor.hpp:24-38) exactly: strict `<` keeps the new entry
// BEHIND equal distances.  Slots start at FLT_MAX (the reference's sentinel, :102), so max() needs no fill
// counter; k <= n_points is enforced by the caller (kd_tree.hpp:193), hence every slot ends up a real point.
//
// The kernel that uses this is bound by VALU issue, not by memory (profiles/r03l_knn16_pmc.txt: 22 k vector
// instructions per wavefront at knn = 16, 75 % of the kernel's duration, two thirds of them this chain, which the whole
// wavefront executes whenever ONE lane has a candidate).  So the chain is as short as it gets:
//   - the k entries live in the LAST k of the K slots (the slots below hold -inf, which nothing is smaller than:
//     they never move), so that max() is always slot K - 1 -- no scan for slot k - 1 after every insertion;
//   - the new distance of slot j is the MEDIAN of old[j-1], old[j] and d (old[j-1] <= old[j]: d below both -> old[j-1],
//     between -> d, above -> old[j]): one v_med3_f32 instead of two selects; the indices follow the K comparisons
//     d < old[j], two selects each.
__device__ __forceinline__ float f_med3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fmed3f(a, b, c);
#else
  const float lo = a < b ? a : b, hi = a < b ? b : a;
  return c < lo ? lo : (c < hi ? c : hi);
#endif
}

template <int K>
struct KnnRegPolicy {
  float ld[K];
  int32_t li[K];
  uint32_t k;
  float e_inv;
  __device__ __forceinline__ void init(uint32_t k_, float e_inv_) {
    k = k_;
    e_inv = e_inv_;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      ld[j] = (uint32_t)j + k >= (uint32_t)K ? 3.402823466e+38f : __uint_as_float(0xFF800000u);  // unused slots: -inf
      li[j] = 0;
    }
  }
  __device__ __forceinline__ float max() const { return ld[K - 1]; }
  __device__ __forceinline__ bool settled() const { return ld[K - 1] == 0.0f; }  // k points AT the query: see NnPolicy
  __device__ __forceinline__ void visit(int32_t idx, float d) {
    d = f_mul(d, e_inv);
    if (ld[K - 1] > d) {
      bool below[K];
#pragma unroll
      for (int j = 0; j < K; ++j) below[j] = d < ld[j];
#pragma unroll
      for (int j = K - 1; j >= 1; --j) {
        li[j] = below[j - 1] ? li[j - 1] : (below[j] ? idx : li[j]);
        ld[j] = f_med3(ld[j - 1], ld[j], d);
      }
      li[0] = below[0] ? idx : li[0];
      ld[0] = below[0] ? d : ld[0];
    }
  }
  __device__ __forceinline__ void store(Neighbor* row) const {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if ((uint32_t)j + k >= (uint32_t)K) {
        Neighbor nb;
        nb.index = li[j];
        nb.distance = ld[j];
        row[(uint32_t)j + k - (uint32_t)K] = nb;
      }
    }
  }
};

// Rows captured during the count pass.  The radius search is a count pass, a scan and a fill pass; the fill pass
// would repeat the whole traversal only to learn where each hit goes.  With a capture the count pass also writes
// every hit to a LOG OF ITS WAVEFRONT, and the fill pass becomes a copy (radius_log_scatter_kernel).
//
// Why a log per wavefront.  r01/r02 kept a chain of 256-byte chunks per ROW, every lane storing its own hits 16
// bytes at a time: the 16 bytes of one lane are followed by the next 16 of the same 128-byte line hundreds of
// microseconds later, 330 k such lines are open at once (more than the L2s hold), so each store went to HBM on its
// own as a 32-byte write -- 13.9 GB of WRITE_SIZE for 6.06 GB of rows on BASELINE config 3 (profiles/r02y_pmc.txt,
// r03m_c3_traffic.json), 4.7 of the 9.7 ms of the pass.  Here the hits a wavefront finds in ONE visit step (ballot
// of `radius > d` over the lanes executing that step) leave as one contiguous store of 8 bytes per hit at the
// wavefront's cursor, and the next step's store continues where this one ended: the lines of a log fill within
// a few hundred cycles and go to HBM once, whole.
//
// Format.  The log is a chain of chunks of kLogChunk 8-byte slots.  Slot 0 is the header {next chunk, groups in
// this chunk}; entries grow upwards from slot 1; the 64-bit lane mask of each group (which lane owns which entry:
// the entries of a group are in lane order) grows downwards from the last slot, so that the copy can fetch 64 masks
// with one coalesced load.  A group never straddles chunks.  Chunk `w` is the first chunk of wavefront `w` of the
// launch; later ones come from kCapSubPools bump allocators (one atomic per chunk, the counters on separate cache
// lines, a wavefront staying with one pool).  The cursor {next slot, groups, chunk} of the wavefront is two LDS
// words behind the record stack, read and written by the first lane of each group only.  A wavefront whose chain
// could not grow is marked and its 64 rows are searched again by the ordinary fill kernel: a small pool costs time,
// never correctness.  The order of the hits of one row is the order its lane found them in: the visit order of the
// reference (search_visitor.hpp:127-156).
constexpr uint32_t kLogChunk = 1024;  // 8-byte slots per chunk (8 KB), header and masks included
constexpr uint32_t kLogEnd = 0xFFFFFFFFu;    // header: no next chunk / cursor: the capture has failed
constexpr uint32_t kCapSubPools = 256;
constexpr uint32_t kCapCounterStride = 16;   // words between counters: one 64-byte line each

struct RadiusCapture {
  Neighbor* chunks;     // (n_static + kCapSubPools * sub_cap) * kLogChunk slots
  uint32_t* counters;   // kCapSubPools * kCapCounterStride words, zero before the count pass
  uint8_t* captured;    // per wavefront of the launch: 1 = all of its rows are in its log
  uint32_t* qids;       // [n_static * 64] the query each lane of each wavefront searched (kLogEnd: none)
  uint32_t n_static;    // wavefronts of the launch
  uint32_t sub_cap;     // chunks per sub-pool
  // The capture as LEAF LISTS (ptk_kernels_lists.hpp): the chunks then hold list entries, [slot][lane].
  uint32_t* lens;       // [n_static * 64] entries listed by each lane
  uint32_t* tables;     // [n_static * kListMaxChunks] the chunks of each wavefront, in order
};

// What a radius batch keeps of its handed-over queries from the count pass to the fill pass (inside the capture block; ptk_kernels_coopr.hpp).
struct RadiusHeavy {
  uint32_t* meta;        // [kMetaWords] counters: kMetaHeavy = queries handed over, kMetaRedo = rows recounted from the
                         // root, kMetaRcEntries = entries handed out
  uint32_t* rows;        // [max_heavy] the row (query index) of hand-over h
  uint32_t* own;         // [max_heavy] hits its lane had found before it stopped (they are in the lane's list)
  uint32_t* run_at;      // [max_heavy] where its sorted entries begin in `entries`
  uint32_t* run_n;       // [max_heavy] how many (kRcLost: none -- the row is searched again)
  unsigned long long* entries;  // [entry_cap] {(first point << cbits) | points, mask of the hits}, as the lane lists hold them
  uint32_t max_heavy;
  uint32_t entry_cap;
};

constexpr int kRadiusCount = 0, kRadiusFill = 1, kRadiusCapture = 2;

// An entry of a capture log: written once, read once by another kernel -- it need not displace the tree in the L2
// (non-temporal store: capture kernel 7.20 -> 6.86 ms on BASELINE config 3).
__device__ __forceinline__ void store_entry(Neighbor* p, Neighbor nb) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_nontemporal_store(pack_neighbor(nb), reinterpret_cast<unsigned long long*>(p));
#else
  *p = nb;
#endif
}

// Lanes below `lane` set in m.
__device__ __forceinline__ uint32_t lanes_below(uint64_t m, uint32_t lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  (void)lane;
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#else
  return (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
#endif
}

// Policies whose result can become final before the traversal ends have settled().
template <class P, class = void>
struct can_settle : std::false_type {};
template <class P>
struct can_settle<P, std::void_t<decltype(std::declval<const P&>().settled())>> : std::true_type {};

// Policies that take the points of a leaf round together (visit_round<N>) say so with kRoundVisit.
template <class P, class = void>
struct takes_rounds : std::false_type {};
template <class P>
struct takes_rounds<P, std::void_t<decltype(P::kRoundVisit)>> : std::integral_constant<bool, P::kRoundVisit> {};

// Policies that want to know where a leaf begins and ends (leaf_begin(ref, tree) before its first point, leaf_end()
// after its last) say so with kLeafHooks.
template <class P, class = void>
struct takes_leaf_hooks : std::false_type {};
template <class P>
struct takes_leaf_hooks<P, std::void_t<decltype(P::kLeafHooks)>> : std::integral_constant<bool, P::kLeafHooks> {};

template <int MODE>
struct RadiusPolicy {  // search_visitor.hpp:127-156 / :252-288
  static constexpr bool kRoundVisit = MODE == kRadiusCapture;
  float radius;  // already scaled by 1/e for the approximate search (:265)
  float e_inv;
  uint64_t count;
  Neighbor* out;  // fill: first record of this query's row; capture: the chunk array
  // capture only
  uint32_t* counters;
  LdsWord* cursor;  // [0] = next slot | groups << 32, [1] = chunk (kLogEnd: capture failed)
  uint32_t sub, sub_cap, n_static, lane;

  // capture: the cursor of a wavefront that starts on its static chunk (one lane; LDS operations of a wavefront
  // are executed in order, so no barrier is needed before the first group reads it)
  __device__ __forceinline__ void open_log(uint32_t wave) {
    cursor[0] = 1ull;
    cursor[1] = wave;
  }
  // capture, after the traversal: the header of the last chunk; true = the log holds every hit of the wavefront
  __device__ __forceinline__ bool close_log() {
    const uint32_t chunk = (uint32_t)cursor[1];
    if (chunk == kLogEnd) return false;
    const uint32_t ng = (uint32_t)(cursor[0] >> 32);
    Neighbor h;
    h.index = (int32_t)kLogEnd;
    h.distance = __uint_as_float(ng);
    out[(uint64_t)chunk * kLogChunk] = h;
    return true;
  }
  // The first lane of a group: room for n entries and their mask; returns the slot of the first entry (0 = none).
  __device__ __forceinline__ uint32_t append_group(uint64_t mask, uint32_t n) {
    const unsigned long long w = cursor[0];
    uint32_t chunk = (uint32_t)cursor[1];
    if (chunk == kLogEnd) return 0u;
    uint32_t pos = (uint32_t)w, ng = (uint32_t)(w >> 32);
    if (pos + n + ng + 1u > kLogChunk) {  // entries [1, pos) and masks [kLogChunk - ng, kLogChunk) would meet
      const uint32_t nx = atomicAdd(&counters[sub * kCapCounterStride], 1u);
      Neighbor h;
      h.index = (int32_t)(nx < sub_cap ? n_static + sub * sub_cap + nx : kLogEnd);
      h.distance = __uint_as_float(ng);
      out[(uint64_t)chunk * kLogChunk] = h;
      chunk = (uint32_t)h.index;
      cursor[1] = chunk;
      if (chunk == kLogEnd) return 0u;
      pos = 1u;
      ng = 0u;
    }
    Neighbor mk;
    mk.index = (int32_t)(uint32_t)mask;
    mk.distance = __uint_as_float((uint32_t)(mask >> 32));
    out[(uint64_t)chunk * kLogChunk + (kLogChunk - 1u - ng)] = mk;
    cursor[0] = (unsigned long long)(pos + n) | ((unsigned long long)(ng + 1u) << 32);
    return chunk * kLogChunk + pos;
  }
  __device__ __forceinline__ float max() const { return radius; }
  // capture: the N points of one round of a leaf scan at once (traverse<> hands them over together when the policy
  // has this member): one exchange with the cursor for up to N groups instead of N -- the read-modify-write of the
  // cursor is a chain of two LDS latencies that every lane of the wavefront waits for.
  template <int N>
  __device__ __forceinline__ void visit_round(const int32_t* idx, const float* dist, uint32_t nvalid) {
    float d[N];
    bool hit[N];
    uint64_t m[N];
    bool any = false;
#pragma unroll
    for (int u = 0; u < N; ++u) {
      d[u] = f_mul(dist[u], e_inv);
      hit[u] = (uint32_t)u < nvalid && radius > d[u];  // strict
#if defined(__HIP_DEVICE_COMPILE__)
      m[u] = __ballot(hit[u]);
#else
      m[u] = hit[u] ? 1ull << lane : 0ull;
#endif
      any = any || hit[u];
    }
    if (any) {
      uint64_t all = 0ull;
#pragma unroll
      for (int u = 0; u < N; ++u) all |= m[u];
      const uint32_t leader = (uint32_t)__builtin_ctzll(all);
      uint32_t base = 0u;
      if (lane == leader) base = append_groups<N>(m);
#if defined(__HIP_DEVICE_COMPILE__)
      base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
#endif
#pragma unroll
      for (int u = 0; u < N; ++u) {
        if (hit[u]) {
          if (base != 0u) {
            Neighbor nb;
            nb.index = idx[u];
            nb.distance = d[u];
            store_entry(out + base + lanes_below(m[u], lane), nb);
          }
          ++count;
        }
        base += base != 0u ? (uint32_t)__popcll(m[u]) : 0u;
      }
    }
  }
  // One lane: room for the groups of a round (the non-empty masks among m[0..N)); the slot of the first entry.
  template <int N>
  __device__ __forceinline__ uint32_t append_groups(const uint64_t* m) {
    const unsigned long long w = cursor[0];
    uint32_t chunk = (uint32_t)cursor[1];
    if (chunk == kLogEnd) return 0u;
    uint32_t n = 0u, groups = 0u;
#pragma unroll
    for (int u = 0; u < N; ++u) {
      n += (uint32_t)__popcll(m[u]);
      groups += m[u] != 0ull ? 1u : 0u;
    }
    uint32_t pos = (uint32_t)w, ng = (uint32_t)(w >> 32);
    if (pos + n + ng + groups > kLogChunk) {
      const uint32_t nx = atomicAdd(&counters[sub * kCapCounterStride], 1u);
      Neighbor h;
      h.index = (int32_t)(nx < sub_cap ? n_static + sub * sub_cap + nx : kLogEnd);
      h.distance = __uint_as_float(ng);
      out[(uint64_t)chunk * kLogChunk] = h;
      chunk = (uint32_t)h.index;
      cursor[1] = chunk;
      if (chunk == kLogEnd) return 0u;
      pos = 1u;
      ng = 0u;
    }
    cursor[0] = (unsigned long long)(pos + n) | ((unsigned long long)(ng + groups) << 32);
#pragma unroll
    for (int u = 0; u < N; ++u) {
      if (m[u] != 0ull) {
        Neighbor mk;
        mk.index = (int32_t)(uint32_t)m[u];
        mk.distance = __uint_as_float((uint32_t)(m[u] >> 32));
        out[(uint64_t)chunk * kLogChunk + (kLogChunk - 1u - ng)] = mk;
        ++ng;
      }
    }
    return chunk * kLogChunk + pos;
  }
  __device__ __forceinline__ void visit(int32_t idx, float d) {
    d = f_mul(d, e_inv);
    const bool hit = radius > d;  // strict
    if (MODE == kRadiusCapture) {
      // The lanes executing this visit together and their hits.  (The CPU emulation of the test tier runs one lane
      // at a time: every hit is a group of its own there.)
#if defined(__HIP_DEVICE_COMPILE__)
      const uint64_t m = __ballot(hit);
#else
      const uint64_t m = hit ? 1ull << lane : 0ull;
#endif
      if (hit) {
        const uint32_t rank = lanes_below(m, lane);
        uint32_t base = 0u;
        if (rank == 0u) base = append_group(m, (uint32_t)__popcll(m));
#if defined(__HIP_DEVICE_COMPILE__)
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(m));
#endif
        if (base != 0u) {
          Neighbor nb;
          nb.index = idx;
          nb.distance = d;
          store_entry(out + base + rank, nb);
        }
        ++count;
      }
    } else if (hit) {
      if (MODE == kRadiusFill) {
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
// RESUME: the stack already holds the pending records of a finished first descent (phase 2
// of the two-phase k = 1 search): start by unwinding instead of descending from the root.
// A subtree still to be searched, as the cooperative search (knn1_coop_kernel) keeps it: the state
// the reference would enter it with.  `gmax` = the largest box distance of a far child on the
// path from the root down to it; its sign bit marks a subtree handed over by a capped traversal
// in the form of its pending record: `ref` is then the record's {parent branch | axis | side},
// the offset of the split axis still the parent's (entering it reads the parent branch).
struct Task {
  uint32_t ref;
  float nbd, off0, off1, off2, gmax;
};
// Counters in Cont::meta (zeroed by knn1_phase_meta_kernel): queries phase 2 gave up on, the next
// entry of that list to hand to a group, queries the cooperative search could not certify.
constexpr uint32_t kMetaHeavy = 24, kMetaRedo = 26, kMetaRanked = 3;
constexpr uint32_t kMetaHeavyRest = 27;  // k > 1: the list of the second of two capped launches side by side
// Words of the counters block (Handover::meta) the cooperative search adds to: why a query went to the redo list
// (pool / spill overflow or a hand-over it cannot start from; more equal distances than the second sweep holds; a box
// distance above the k-th distance on the way to a neighbour; a k-th distance outside [1e-30, 1e30]) and how many
// queries took the second sweep.
constexpr uint32_t kKnnWhyPool = 8, kKnnWhyTie = 9, kKnnWhyBox = 10, kKnnWhyRange = 11, kKnnTieSweeps = 12;
constexpr uint32_t kMaxTasks = 64;               // tasks a capped traversal can hand over per query
constexpr uint32_t kTasksFromRoot = 0xFFFFFFFFu;  // more than that (or no room): search again from the root
constexpr uint32_t kTasksRedo = 0xFFFFFFFEu;      // non-monotone box distances met: only the reference order will do

// Where a capped traversal leaves its unfinished work.
struct Handover {
  uint32_t counter;      // word of Cont::meta that counts this list (kMetaHeavy)
  uint32_t* meta;        // Cont::meta
  uint32_t* heavy_list;  // [nq] slots of the queries handed over
  uint32_t* ntasks;      // [nq] tasks of list entry h (or kTasksFromRoot / kTasksRedo)
  Task* tasks;           // [max_heavy][kMaxTasks]
  uint32_t max_heavy;
  uint32_t slot;         // this query
  uint32_t full_keeps;   // != 0: a query that finds the list full goes on in its lane (the consumer takes
                         // min(count, max_heavy) entries); 0: it is listed without tasks (kTasksFromRoot)
};

// CAPPED: give up (return false) when more than `cap` far children have been entered: what is
// still on the stack goes to `ho` -- every pending far child that can still matter together with
// the state it would be entered with, next-to-visit first.
// KEEPS (CAPPED only; Handover::full_keeps set): a query that finds the hand-over list full goes on in its lane.
template <int LEAFB, bool RESUME = false, class M = MetricL2, bool CAPPED = false, bool KEEPS = false, class Policy,
          class StackT>
__device__ __forceinline__ bool traverse(
    const DevTree& t, float qx, float qy, float qz, Policy& pol, StackT& st, uint32_t cap = 0,
    const Handover* ho = nullptr) {
  const uint4* __restrict__ nodes = t.nodes;
  const float4* __restrict__ pts = t.pts;
  uint32_t ref = RESUME ? kLeafBit : t.root_ref;  // RESUME: an empty leaf, falls through to the unwind
  float nbd = 0.0f, off0 = 0.0f, off1 = 0.0f, off2 = 0.0f;
  uint32_t entered = 0;
  bool on_its_own = false;  // KEEPS: the list was full when this query reached its cap

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
      const float new_off = M::one(dv);
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
      if constexpr (takes_leaf_hooks<Policy>::value) {
        if (count != 0u) pol.leaf_begin(ref, t);
      }
      for (uint32_t j = 0; j < count; j += LEAFB) {
        float4 p[LEAFB];
#pragma unroll
        for (int u = 0; u < LEAFB; ++u) p[u] = pts[begin + j + u];
        if constexpr (takes_rounds<Policy>::value) {
          int32_t ids[LEAFB];
          float ds[LEAFB];
#pragma unroll
          for (int u = 0; u < LEAFB; ++u) {
            PTK_KEEP4(p[u]);
            ids[u] = __float_as_int(p[u].w);
            // (each difference pinned to a register of its own: left alone, the compiler pairs the arithmetic of two
            // points into packed instructions and spends more on moving operands into pairs than it saves -- capture
            // kernel 7.17 -> 6.84 ms without the pairing)
            float dx = f_sub(qx, p[u].x), dy = f_sub(qy, p[u].y), dz = f_sub(qz, p[u].z);
            PTK_SCALAR(dx);
            PTK_SCALAR(dy);
            PTK_SCALAR(dz);
            float dsum = M::one(dx);
            PTK_SCALAR(dsum);
            dsum = M::acc(dsum, dy);
            PTK_SCALAR(dsum);
            ds[u] = M::acc(dsum, dz);
          }
          pol.template visit_round<LEAFB>(ids, ds, count - j);
          continue;
        }
#pragma unroll
        for (int u = 0; u < LEAFB; ++u) {
          if (j + u < count) {
            // All four words of the record are needed HERE: without this the compiler narrows the
            // 16-byte load to 12 bytes and fetches the index with a second, dependent load inside
            // the branch that stores a hit (radius fill pass: +10 ms of 27 on BASELINE config 3).
            PTK_KEEP4(p[u]);
            float dx = f_sub(qx, p[u].x);
            float dy = f_sub(qy, p[u].y);
            float dz = f_sub(qz, p[u].z);
            PTK_SCALAR(dx);
            PTK_SCALAR(dy);
            PTK_SCALAR(dz);
            pol.visit(__float_as_int(p[u].w), point_distance3<M>(dx, dy, dz));
          }
        }
      }
      if constexpr (takes_leaf_hooks<Policy>::value) {
        if (count != 0u) pol.leaf_end();
      }
    }

    // Back up to the next far child still worth entering.  The records are consumed kUnwind at a
    // time: that many independent LDS reads are issued together and then examined in order from
    // registers (measured: popping them one by one, each ds_read waiting for the branch on the
    // previous record, was the largest part of a round of the expensive queries, ~2.5 k cycles).
    uint32_t enter_meta = 0;
    float enter_val = 0.0f;
    for (;;) {
      if (st.empty()) return true;
      if constexpr (can_settle<Policy>::value) {
        if (pol.settled()) return true;  // (every record left would be popped for nothing)
      }
      Record rr[StackT::kUnwind];
      const int got = st.peek(rr);  // 1 .. kUnwind records, newest first (refills the ring if needed)
      int used = 0;
      bool enter = false;
#pragma unroll
      for (int i = 0; i < StackT::kUnwind; ++i) {
        if (!enter && i < got) {
          used = i + 1;
          const float val = __uint_as_float(rr[i].y);
          if (rr[i].x & kRecUndo) {
            if (rr[i].x & kRecSide) {
              nbd = val;
            } else {
              const uint32_t axis = (rr[i].x >> 28) & 3u;
              off0 = axis == 0 ? val : off0;
              off1 = axis == 1 ? val : off1;
              off2 = axis == 2 ? val : off2;
            }
          } else if (pol.max() >= val) {  // the authoritative test of search.hpp:99
            enter = true;
            enter_meta = rr[i].x;
            enter_val = val;
          }
        }
      }
      st.drop(used);
      // ONE batch of records per turn of the outer loop: a lane with a long way back up --
      // above all the last unwind of a query, which pops what is left of the home path, some thirty records nearly
      // all rejected -- no longer holds the wavefront in this loop while the other lanes have leaves to scan (an
      // empty leaf brings it back here next turn).  knn = 16: 5.28 -> 4.86 ms on cloud L, 5.80 -> 5.16 on cloud U;
      // bounding the descent (4 steps per turn) or the leaf scan (one round per turn) the same way loses
      // (profiles/r03_notes.txt item 13).  Phase 2 of the k = 1 search (RESUME, CAPPED): 1.330 -> 1.299 ms of
      // traversal kernels on cloud L, 1.332 -> 1.297 on cloud U.
      if (!enter) {
        if (st.empty()) return true;
        ref = kLeafBit;
        break;
      }
      if (enter) {
        bool hand_over = CAPPED && !(KEEPS && on_its_own) && ++entered > cap;
        uint32_t h = 0;
        if (hand_over) {
          h = atomicAdd(&ho->meta[ho->counter], 1u);
          if (KEEPS && h >= ho->max_heavy) {  // no room: this lane finishes its query itself
            hand_over = false;
            on_its_own = true;
          }
        }
        if (hand_over) {
          ho->heavy_list[h] = ho->slot;
          Task* out = h < ho->max_heavy ? ho->tasks + (uint64_t)h * kMaxTasks : nullptr;
          uint32_t n = 0;
          auto emit = [&](uint32_t meta, float val) {
            if (out != nullptr && n < kMaxTasks) {
              Task k;
              k.ref = meta;
              k.nbd = val;
              k.off0 = off0;
              k.off1 = off1;
              k.off2 = off2;
              // While `monotone` holds the largest box distance on the path so far is the current one.
              k.gmax = __uint_as_float(__float_as_uint(nbd < val ? val : nbd) | 0x80000000u);
              out[n] = k;
            }
            ++n;
          };
          // `gmax` of a task = the largest box distance of a far child on the path from the root to it.  The far
          // children on that path are the ones the stack still holds an undo record of; while each of them had a box
          // distance >= its parent's (`monotone`: a restored value never exceeds the one it replaces) the largest is the
          // current one.  Far children entered and left again are on nobody's path, so the traversal keeps no flag
          // (r06: one updated per far child cost the double k = 1 kernel 5 % of its time, profiles/r06_notes.txt item 12).
          bool monotone = true;
          emit(enter_meta, enter_val);
          while (!st.empty()) {
            const Record r = st.pop();
            const float val = __uint_as_float(r.y);
            if (r.x & kRecUndo) {
              if (r.x & kRecSide) {
                if (val > nbd) monotone = false;
                nbd = val;
              } else {
                const uint32_t axis = (r.x >> 28) & 3u;
                off0 = axis == 0 ? val : off0;
                off1 = axis == 1 ? val : off1;
                off2 = axis == 2 ? val : off2;
              }
            } else if (pol.max() >= val) {
              emit(r.x, val);
            }
          }
          ho->ntasks[h] = !monotone ? kTasksRedo : (out == nullptr || n > kMaxTasks ? kTasksFromRoot : n);
          return false;
        }
        const float val = enter_val;
        const uint32_t idx = enter_meta & kRecIdxMask;
        const uint32_t axis = (enter_meta >> 28) & 3u;
        const bool far_is_right = (enter_meta & kRecSide) != 0;
        const uint4 nd = nodes[idx];
        const float plane = far_is_right ? __uint_as_float(nd.y) : __uint_as_float(nd.x);
        const float dv = f_sub(plane, sel3(axis, qx, qy, qz));
        const float new_off = M::one(dv);
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

// Blocks are dealt round-robin to the 8 XCDs, each with an L2 of its own: a wavefront's neighbours in the
// (spatially sorted) batch should run on the same XCD.  Pure performance: any mapping is correct.
// Blocks are taken in groups of 8 << run_log2; inside a group XCD x gets the run of
// 1 << run_log2 consecutive tiles number x.  Every XCD then moves through the batch at the same pace while its
// L2 still sees runs of neighbouring queries.  A batch whose cost varies along the order must not leave one XCD
// with the expensive eighth: the radius search of BASELINE config 3 on the scan-like cloud took 32 ms with
// contiguous eighths (the XCD that got the dense middle of the scan ran on alone) and 16.7 ms in runs of 8 - 32
// wavefronts; knn = 16 7.65 -> 6.67 ms (profiles/r02_notes.txt item 21).
constexpr uint32_t kXcdRunLog2 = 4;
// The general 3-D kernels (one launch, expensive queries first) take longer runs: knn = 16 with runs of 8 / 16 / 32 / 64
// wavefronts 4.69 / 4.73 / 4.65 / 4.62 ms on cloud L, 5.06 / 5.01 / 4.94 / 4.88 on cloud U; the radius capture does not care.
constexpr uint32_t kXcdRunGeneral = 6;
__device__ __forceinline__ uint32_t xcd_runs(uint32_t b, uint32_t nb, uint32_t run_log2 = kXcdRunLog2) {
  const uint32_t group = 8u << run_log2;
  const uint32_t g = b / group;
  if ((g + 1u) * group > nb) return b;  // the incomplete last group keeps its order
  const uint32_t r = b - g * group;
  return g * group + ((r & 7u) << run_log2) + (r >> 3);
}

__device__ __forceinline__ void load_query(
    const float* __restrict__ q, uint32_t dim, uint64_t qi, float& x, float& y, float& z) {
  if (dim == 3u) {  // (uniform) the row with ONE 12-byte load: a gathered row is one request instead of three
    const float3 v = reinterpret_cast<const float3*>(q)[qi];
    x = v.x;
    y = v.y;
    z = v.z;
    return;
  }
  const float* p = q + qi * dim;
  x = p[0];
  y = dim > 1 ? p[1] : 0.0f;
  z = dim > 2 ? p[2] : 0.0f;
}

// The 3-D kernels give a space of fewer dimensions zero coordinates for the missing axes -- exact
// for sums and maxima, wrong for a minimum: there the query's missing coordinates become +inf
// (|inf - 0| = inf never is the minimum; no split ever uses those axes).
template <class M>
__device__ __forceinline__ void pad_query(uint32_t dim, float& y, float& z) {
  if (M::kMin) {
    const float inf = __uint_as_float(0x7F800000u);
    y = dim > 1 ? y : inf;
    z = dim > 2 ? z : inf;
  }
}

extern __shared__ __attribute__((aligned(16))) unsigned char ptk_smem[];

// Experiment builds only (PTK_EXTRA_FLAGS=-DPTK_WAVE_TRACE, tools/wave_trace.py): when and where every wavefront of
// the general kernels ran -- {start, end} of the 100 MHz wall clock and the hardware id -- to tell a kernel that is
// bound by its throughput from one that waits for a tail of slow wavefronts.  The shipped library has none of this.
#if defined(PTK_WAVE_TRACE)
__device__ unsigned long long* g_wave_trace = nullptr;  // [4 * blocks of the launch]
__device__ int g_wave_trace_sel = 0;                    // which of the k = 1 kernels records (0: none of them)
#endif
#if defined(PTK_WAVE_TRACE) && defined(__HIP_DEVICE_COMPILE__)
#define PTK_TRACE_BEGIN() const unsigned long long trace_t0_ = wall_clock64(); const unsigned long long trace_c0_ = clock64()
#define PTK_TRACE_END()                                                                                  \
  do {                                                                                                   \
    if (g_wave_trace != nullptr && g_wave_trace_sel == 0 && threadIdx.x == 0) {                          \
      g_wave_trace[4ull * blockIdx.x + 0] = trace_t0_;                                                   \
      g_wave_trace[4ull * blockIdx.x + 1] = wall_clock64();                                              \
      g_wave_trace[4ull * blockIdx.x + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |  \
          ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); /* HW_ID | XCC_ID << 32 */ \
      g_wave_trace[4ull * blockIdx.x + 3] = clock64() - trace_c0_;                                         \
    }                                                                                                    \
  } while (0)
// The k = 1 kernels (their lanes leave one by one): recorded only when g_wave_trace_sel names the kernel (1 = phase 1,
// 2 = phase 2, 3 = the direct cooperative search, 4 = the cooperative search behind phase 2); the end is the last lane's.
#define PTK_TRACE_BEGIN_SEL(ID)                                                      \
  const bool trace_on_ = g_wave_trace != nullptr && g_wave_trace_sel == (ID);        \
  const unsigned long long trace_t0_ = wall_clock64()
#define PTK_TRACE_END_ANY()                                                                                   \
  do {                                                                                                        \
    if (trace_on_) {                                                                                          \
      g_wave_trace[4ull * blockIdx.x + 0] = trace_t0_;                                                        \
      atomicMax(&g_wave_trace[4ull * blockIdx.x + 1], (unsigned long long)wall_clock64());                    \
      g_wave_trace[4ull * blockIdx.x + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |   \
          ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);                             \
    }                                                                                                         \
  } while (0)
#else
#define PTK_TRACE_BEGIN() ((void)0)
#define PTK_TRACE_END() ((void)0)
#define PTK_TRACE_BEGIN_SEL(ID) ((void)0)
#define PTK_TRACE_END_ANY() ((void)0)
#endif

// ---- general k -------------------------------------------------------------------------
// LIST_LDS: the k-list lives in LDS behind the stack ([slot][lane]) and is copied
// to the output row at the end; otherwise the output row itself is the list.
template <int S, int OVF, int BLOCK, int LEAFB, bool LIST_LDS, class M = MetricL2>
__global__ __launch_bounds__(BLOCK) void knn_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim,
    const uint32_t* __restrict__ perm, uint64_t nq, uint32_t k, float e_inv,
    Neighbor* __restrict__ out) {
  const uint32_t tile = xcd_runs(blockIdx.x, gridDim.x, kXcdRunGeneral);
  const uint64_t i = (uint64_t)tile * BLOCK + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[i] : i;
  float qx, qy, qz;
  load_query(queries, dim, qi, qx, qy, qz);
  pad_query<M>(dim, qy, qz);

  PTK_STACK(S, OVF, BLOCK, st, t);
  KnnPolicy<LIST_LDS> pol;
  if constexpr (LIST_LDS) {
    pol.list = (LdsWord*)(ptk_smem + (size_t)S * BLOCK * 8) + threadIdx.x;
    pol.stride = BLOCK;
  } else {
    pol.list = out + qi * k;
    pol.stride = 1;
  }
  pol.k = k;
  pol.filled = 0;
  pol.worst = 3.402823466e+38f;
  pol.e_inv = e_inv;
  traverse<LEAFB, false, M>(t, qx, qy, qz, pol, st);

  if (LIST_LDS) {
    Neighbor* row = out + qi * k;
    for (uint32_t j = 0; j < pol.filled; ++j) row[j] = pol.get(j);
  }
  if (pol.filled < k) {  // k > reachable points: mirror the reference's sentinel (:102)
    Neighbor nb;
    nb.index = 0;
    nb.distance = 3.402823466e+38f;
    out[qi * k + (k - 1)] = nb;
  }
}

// (experiments: -DPTK_KNN_WAVES=n asks the compiler to fit n wavefronts per SIMD)
#if defined(PTK_KNN_WAVES) && defined(__HIP_DEVICE_COMPILE__)
#define PTK_KNN_REG_WAVES __attribute__((amdgpu_waves_per_eu(PTK_KNN_WAVES, PTK_KNN_WAVES)))
#else
#define PTK_KNN_REG_WAVES
#endif
// CAPPED: a query that has entered more than `cap` far children stops; its list (in its row, as always) and its
// stack (`ho`) go to knn_coop_kernel (ptk_kernels_coopk.hpp), which finishes it with a whole wavefront.
template <int K, int S, int OVF, int BLOCK, int LEAFB, class M = MetricL2, bool CAPPED = false>
__global__ __launch_bounds__(BLOCK) PTK_KNN_REG_WAVES void knn_reg_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim,
    const uint32_t* __restrict__ perm, uint64_t nq, uint32_t k, float e_inv,
    Neighbor* __restrict__ out, uint32_t cap = 0, Handover ho = Handover{}) {
  PTK_TRACE_BEGIN();
  const uint32_t tile = xcd_runs(blockIdx.x, gridDim.x, kXcdRunGeneral);
  const uint64_t i = (uint64_t)tile * BLOCK + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[i] : i;
  float qx, qy, qz;
  load_query(queries, dim, qi, qx, qy, qz);
  pad_query<M>(dim, qy, qz);
  Record spill[OVF > 0 ? OVF : 1];
  Stack<S, OVF, BLOCK> st;
  st.init((LdsWord*)ptk_smem, threadIdx.x, spill);
  KnnRegPolicy<K> pol;
  pol.init(k, e_inv);
  if constexpr (CAPPED) {
    ho.slot = (uint32_t)qi;
    traverse<LEAFB, false, M, true, true>(t, qx, qy, qz, pol, st, cap, &ho);
  } else {
    traverse<LEAFB, false, M>(t, qx, qy, qz, pol, st);
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // Full lists of a full wavefront leave through the LDS the stack no longer needs: a lane's K entries are one
  // row of K x 8 bytes, and K lanes write it with ONE store (whole 64-byte sectors) instead of each lane writing
  // 8 bytes of its own row K times over (64 partial lines per store: 1.83 GB of WRITE_SIZE for 0.92 GB of rows at
  // knn = 16).  Slot (lane, j) sits at lane * K + ((j + lane) % K): the writes and the reads are conflict-free.
  if constexpr (BLOCK == 64 && K <= S && (K & (K - 1)) == 0 && K >= 2) {
    if (k == (uint32_t)K && (uint64_t)tile * 64u + 63u < nq) {  // (uniform)
      LdsWord* rows = (LdsWord*)ptk_smem;
      const uint32_t lane = threadIdx.x;
      __syncthreads();  // (one wavefront: every lane is out of the traversal before its stack slots are reused)
#pragma unroll
      for (int j = 0; j < K; ++j) {
        Neighbor nb;
        nb.index = pol.li[j];
        nb.distance = pol.ld[j];
        rows[lane * K + (((uint32_t)j + lane) & (K - 1))] = pack_neighbor(nb);
      }
      unsigned long long* __restrict__ dst = reinterpret_cast<unsigned long long*>(out);
      if ((reinterpret_cast<uintptr_t>(out) & 15u) == 0u) {  // (uniform)
        // Two entries -- 16 bytes -- per lane, K / 2 lanes to a row: a store instruction writes 128 / K whole rows.
        // (With 8 bytes per lane the rows left as twice their size in WRITE_SIZE: 1.83 GB for the 0.92 GB of config 3.)
        constexpr uint32_t kLanesPerRow = K / 2, kRowsPerStore = 64u / kLanesPerRow;
        const uint32_t e2 = lane % kLanesPerRow, sub = lane / kLanesPerRow;
#pragma unroll
        for (uint32_t r0 = 0; r0 < 64u; r0 += kRowsPerStore) {
          const uint32_t r = r0 + sub;  // the lane whose row this is
          const uint32_t q_r = (uint32_t)__shfl((int)(uint32_t)qi, (int)r);
          RowPair two;
          two.a = rows[r * K + ((2u * e2 + r) & (K - 1))];
          two.b = rows[r * K + ((2u * e2 + 1u + r) & (K - 1))];
          *reinterpret_cast<RowPair*>(dst + (uint64_t)q_r * K + 2u * e2) = two;
        }
      } else {
        constexpr uint32_t kRowsPerStore = 64u / K;
        const uint32_t e = lane & (K - 1), sub = lane / K;
#pragma unroll
        for (uint32_t r0 = 0; r0 < 64u; r0 += kRowsPerStore) {
          const uint32_t r = r0 + sub;  // the lane whose row this is
          const uint32_t q_r = (uint32_t)__shfl((int)(uint32_t)qi, (int)r);
          dst[(uint64_t)q_r * K + e] = rows[r * K + ((e + r) & (K - 1))];
        }
      }
      PTK_TRACE_END();
      return;
    }
  }
#endif
  pol.store(out + qi * k);
}

// ---- radius: count pass and fill pass ------------------------------------------------------
// n_dev (fill pass only): the batch is the first *n_dev entries of perm -- the rows a capture
// could not hold, listed on the device (no host round trip to size the launch).
template <int S, int OVF, int BLOCK, int LEAFB, bool FILL, class M = MetricL2>
__global__ __launch_bounds__(BLOCK) void radius_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim,
    const uint32_t* __restrict__ perm, uint64_t nq, float radius, float e_inv,
    uint64_t* __restrict__ counts, const uint64_t* __restrict__ offsets,
    Neighbor* __restrict__ out, const uint32_t* __restrict__ n_dev = nullptr) {
  if (n_dev != nullptr) nq = *n_dev;
  const uint32_t tile = xcd_runs(blockIdx.x, gridDim.x, kXcdRunGeneral);
  const uint64_t i = (uint64_t)tile * BLOCK + threadIdx.x;
  if (i >= nq) return;
  const uint64_t qi = perm ? perm[i] : i;
  float qx, qy, qz;
  load_query(queries, dim, qi, qx, qy, qz);
  pad_query<M>(dim, qy, qz);

  PTK_STACK(S, OVF, BLOCK, st, t);
  RadiusPolicy<FILL ? kRadiusFill : kRadiusCount> pol;
  pol.radius = f_mul(radius, e_inv);  // search_visitor.hpp:265
  pol.e_inv = e_inv;
  pol.count = 0;
  pol.out = FILL ? out + offsets[qi] : nullptr;
  traverse<LEAFB, false, M>(t, qx, qy, qz, pol, st);
  if (!FILL) counts[qi] = pol.count;
}

// The count pass that also captures the rows (see RadiusCapture).  One wavefront per block: the cursor of its log
// is the two LDS words behind the record stack.
template <int S, int OVF, int BLOCK, int LEAFB, class M = MetricL2>
__global__ __launch_bounds__(BLOCK) void radius_capture_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim,
    const uint32_t* __restrict__ perm, uint64_t nq, float radius, float e_inv,
    uint64_t* __restrict__ counts, RadiusCapture cap) {
  static_assert(BLOCK == 64, "one log per block");
  const uint32_t tile = xcd_runs(blockIdx.x, gridDim.x, kXcdRunGeneral);
  const uint64_t i = (uint64_t)tile * BLOCK + threadIdx.x;
  if (i >= nq) {
    cap.qids[i] = kLogEnd;
    return;
  }
  const uint64_t qi = perm ? perm[i] : i;
  cap.qids[i] = (uint32_t)qi;
  float qx, qy, qz;
  load_query(queries, dim, qi, qx, qy, qz);
  pad_query<M>(dim, qy, qz);

  Record spill[OVF > 0 ? OVF : 1];
  Stack<S, OVF, BLOCK> st;
  st.init((LdsWord*)ptk_smem, threadIdx.x, spill);
  RadiusPolicy<kRadiusCapture> pol;
  pol.radius = f_mul(radius, e_inv);
  pol.e_inv = e_inv;
  pol.count = 0;
  pol.out = cap.chunks;
  pol.counters = cap.counters;
  pol.cursor = (LdsWord*)(ptk_smem + (size_t)S * BLOCK * 8);
  pol.sub = (blockIdx.x * 0x9E3779B1u) >> 24;  // kCapSubPools = 256: the top byte of a hash of the block
  pol.sub_cap = cap.sub_cap;
  pol.n_static = cap.n_static;
  pol.lane = threadIdx.x;
  if (threadIdx.x == 0) pol.open_log(tile);
  traverse<LEAFB, false, M>(t, qx, qy, qz, pol, st);
  counts[qi] = pol.count;
#if defined(__HIP_DEVICE_COMPILE__)
  __syncthreads();  // (one wavefront: every lane has appended its last group before the header is closed)
#endif
  cap.captured[tile] = pol.close_log() ? 1 : 0;  // (every lane writes the same two values)
}

// The fill pass of a captured batch: one wavefront copies the log of one wavefront of the count pass into the 64
// rows it belongs to, a chunk at a time:
//   stage   the chunk is brought into LDS whole -- loads of 512 contiguous bytes, all in flight together, and the
//           loads of the NEXT chunk (its number is in this chunk's header) are issued before this one is taken
//           apart: one memory latency per chunk;
//   count   the masks of up to 64 groups are read with one LDS access (lane g holds group g's); every lane counts
//           the groups it has an entry in, and a scan over the lanes gives each row its place in the sorted chunk;
//   move    group by group (the slot of a group's first entry is a running sum on the scalar unit) the owners move
//           their entries, LDS to LDS, behind one another: the chunk is now ordered by row, each row's entries in
//           log order -- the order its lane found them in;
//   write   the sorted chunk leaves with consecutive lanes on consecutive entries; the entries of a row lie at
//           consecutive addresses, so each run of them is one request.  A run stops at the last 32-byte boundary
//           of its row it reaches; the up to three entries behind it wait (in registers) for the next chunk, so no
//           32-byte sector of a row is written twice (without this: 10.4 GB of WRITE_SIZE for 6.06 GB of rows).
// (Appending entry by entry from the log to 64 rows -- the first form of this kernel -- is 758 M uncoalesced 8-byte
// stores on BASELINE config 3, each a request of its own and a 32-byte write in HBM: 8.7 ms, 13.0 GB of WRITE_SIZE
// for 6.06 GB of rows, 90 % of the wavefront cycles waiting to issue; profiles/r03r_c3_pmc.txt.)
// Wavefronts the capture could not hold are listed, row by row, for radius_kernel<FILL>.
constexpr int kLogUnroll = 4;
// LDS bytes per wavefront: staged chunk, sorted chunk (+ 3 entries kept back per row), row tables, owner of each slot
constexpr uint32_t kLogKeep = 3;  // entries a row may hold back: its runs end on boundaries of (kLogKeep + 1) x 8 bytes
constexpr uint32_t kLogCarry = 64u * kLogKeep;  // room for them in the sorted chunk (rounded up below)
constexpr uint32_t kLogSortedPad = (kLogCarry + 63u) & ~63u;
constexpr uint32_t kLogScatterLds = kLogChunk * 8u * 2u + kLogSortedPad * 8u + 64u * 12u + kLogChunk + kLogSortedPad;
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void radius_log_scatter_kernel(
    RadiusCapture cap, const uint64_t* __restrict__ offsets, Neighbor* __restrict__ out,
    uint32_t* __restrict__ over_list, uint32_t* __restrict__ n_over) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t tile = blockIdx.x * WAVES + threadIdx.x / 64u;
  if (tile >= cap.n_static) return;
  const uint32_t qi = cap.qids[(uint64_t)tile * 64 + lane];
  const bool valid = qi != kLogEnd;
  if (!cap.captured[tile]) {
    if (valid) over_list[atomicAdd(n_over, 1u)] = qi;
    return;
  }
  uint32_t chunk = tile;
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr uint32_t kPer = kLogChunk / 64u;  // slots of a chunk per lane
  uint64_t rowpos = valid ? offsets[qi] : 0ull;  // where this lane's next entry goes
  unsigned char PTK_LDS* mine_lds = (unsigned char PTK_LDS*)ptk_smem + (size_t)(threadIdx.x / 64u) * kLogScatterLds;
  LdsWord* stage = (LdsWord*)mine_lds;
  LdsWord* sorted = stage + kLogChunk;
  LdsWord* adj = sorted + kLogChunk + kLogSortedPad;
  uint32_t PTK_LDS* lim = (uint32_t PTK_LDS*)(adj + 64);
  unsigned char PTK_LDS* own = (unsigned char PTK_LDS*)(lim + 64);
  unsigned long long* __restrict__ dst = reinterpret_cast<unsigned long long*>(out);
  // (two slots per lane and load: 16 bytes per lane, 1 KB per instruction)
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const u64x2* __restrict__ pairs = reinterpret_cast<const u64x2*>(cap.chunks);
  u64x2 in[kPer / 2];
#pragma unroll
  for (uint32_t j = 0; j < kPer / 2; ++j) in[j] = __builtin_nontemporal_load(pairs + (uint64_t)chunk * (kLogChunk / 2) + j * 64u + lane);
  unsigned long long kept[kLogKeep] = {};  // entries of this lane's row still to be written
  uint32_t n_kept = 0u;
  for (;;) {
#pragma unroll
    for (uint32_t j = 0; j < kPer / 2; ++j) {
      stage[2u * (j * 64u + lane)] = in[j].x;
      stage[2u * (j * 64u + lane) + 1u] = in[j].y;
    }
    const unsigned long long head = stage[0];
    const uint32_t next = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)head);
    const uint32_t ng = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(head >> 32));
    if (next != kLogEnd) {
#pragma unroll
      for (uint32_t j = 0; j < kPer / 2; ++j) in[j] = __builtin_nontemporal_load(pairs + (uint64_t)next * (kLogChunk / 2) + j * 64u + lane);
    }
    // count
    uint32_t n_mine = 0u;
    for (uint32_t g0 = 0; g0 < ng; g0 += 64u) {
      const uint32_t left = ng - g0 < 64u ? ng - g0 : 64u;
      const unsigned long long mk = lane < left ? stage[kLogChunk - 1u - g0 - lane] : 0ull;  // (no group: no owner)
      for (uint32_t g = 0; g < left; ++g) {
        const uint64_t m = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mk, (int)g) |
                           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mk >> 32), (int)g) << 32);
        n_mine += (uint32_t)((m >> lane) & 1ull);
      }
    }
    // (entries held back from the last chunk come first, see `write`)
    const uint32_t n_row = n_mine + n_kept;
    uint32_t incl = n_row;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, d);
      incl += lane >= (uint32_t)d ? v : 0u;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t first = incl - n_row;
    uint32_t c = first;
    adj[lane] = rowpos - (uint64_t)first;  // (row position of sorted slot f of this lane = adj + f; wraps are harmless)
    // What leaves now: up to the last 32-byte boundary of the row this chunk reaches (everything with the last chunk).
    const uint64_t row_end = rowpos + n_row;
    const uint64_t stop = next == kLogEnd ? row_end : (row_end & ~(uint64_t)kLogKeep);
    const uint32_t n_out = stop > rowpos ? (uint32_t)(stop - rowpos) : 0u;
    lim[lane] = first + n_out;
    // move
#pragma unroll
    for (uint32_t j = 0; j < kLogKeep; ++j) {
      if (j < n_kept) {
        sorted[c] = kept[j];
        own[c] = (unsigned char)lane;
        ++c;
      }
    }
    uint32_t pos = 1u;
    for (uint32_t g0 = 0; g0 < ng; g0 += 64u) {
      const uint32_t left = ng - g0 < 64u ? ng - g0 : 64u;
      const unsigned long long mk = lane < left ? stage[kLogChunk - 1u - g0 - lane] : 0ull;
      for (uint32_t g = 0; g < left; g += kLogUnroll) {
        unsigned long long e[kLogUnroll];
        bool mine[kLogUnroll];
#pragma unroll
        for (int u = 0; u < kLogUnroll; ++u) {
          const int src = (int)((g + u) & 63u);
          const uint64_t m = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mk, src) |
                             ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mk >> 32), src) << 32);
          const uint64_t mm = g + u < left ? m : 0ull;
          mine[u] = ((mm >> lane) & 1ull) != 0ull;
          if (mine[u]) e[u] = stage[pos + lanes_below(mm, lane)];
          pos += (uint32_t)__popcll(mm);
        }
#pragma unroll
        for (int u = 0; u < kLogUnroll; ++u) {
          if (mine[u]) {
            sorted[c] = e[u];
            own[c] = (unsigned char)lane;
            ++c;
          }
        }
      }
    }
    // write
    for (uint32_t f = lane; f < total; f += 64u) {
      const uint32_t o = own[f];
      if (f < lim[o]) dst[adj[o] + f] = sorted[f];  // (not a non-temporal store: the runs of a row merge in the L2; 4.4 vs 5.2 ms)
    }
    rowpos += n_out;
    n_kept = n_row - n_out;  // <= kLogKeep
#pragma unroll
    for (uint32_t j = 0; j < kLogKeep; ++j) {
      if (j < n_kept) kept[j] = sorted[first + n_out + j];
    }
    if (next == kLogEnd) break;
  }
#else
  Neighbor* row = out + (valid ? offsets[qi] : 0ull);
  for (;;) {  // one lane at a time: the same walk without the wavefront
    const Neighbor* base = cap.chunks + (uint64_t)chunk * kLogChunk;
    const Neighbor head = base[0];
    const uint32_t ng = __float_as_uint(head.distance);
    uint32_t pos = 1u;
    for (uint32_t g = 0; g < ng; ++g) {
      const Neighbor mk = base[kLogChunk - 1u - g];
      const uint64_t m = (uint64_t)(uint32_t)mk.index | ((uint64_t)__float_as_uint(mk.distance) << 32);
      if ((m >> lane) & 1ull) *row++ = base[pos + lanes_below(m, lane)];
      pos += (uint32_t)__popcll(m);
    }
    if ((uint32_t)head.index == kLogEnd) break;
    chunk = (uint32_t)head.index;
  }
#endif
}

// ---- two-phase k = 1 search ---------------------------------------------------------------
//
// Measured on BASELINE config 2 (profiles/r01b_*): in the single kernel above a wavefront
// spends two thirds of its memory round trips in far-child work that only a few of its 64
// lanes need (the average query enters 2.3 far leaves, the worst lane of a wave 8-15), and
// a few "monster" queries -- e.g. at the centre of the scanner's blind disc, 600 leaves --
// hold a whole wave for milliseconds.  The first descent, by contrast, is the same ~22
// steps for every lane.  So the k = 1 search is split where the work stops being uniform:
//
//   phase 1  every query: root -> home leaf, best = nearest point of that leaf.  The far
//            children passed on the way that can still matter (box distance <= best; the
//            pop-time test of the reference can only reject more) are kept, at most kContSlots
//            records.  The reference's next steps are taken as well while they need no stack
//            (the deepest record's far child is a leaf: typically the home leaf's sibling).
//            What then remains is written out as a CONTINUATION, typically 1-3 records; none
//            for about half of the queries, whose answer is then already final.
//   sort     continuations are ordered by class (= record count), each class keeping its Morton order: a
//            counting sort over the three class bits for exact searches (phase 2 is capped, see below); for
//            approximate ones, which phase 2 runs to their end, by the full 16-bit key (make_cont_key) that
//            also ranks the classes holding every expensive query by how far their home-leaf best is (which
//            predicts the cost).  The heaviest start first, in three tiers (knn1_phase2_kernel).
//   phase 2  the records are pushed back and the reference traversal resumes exactly where
//            it left off (state: box distance 0, all offsets 0, best = home-leaf best).
//
// The visit order of each query is unchanged, so results stay bit-identical.
constexpr int kContSlots = 6;           // records a continuation can carry
constexpr uint32_t kContOverflow = 7;   // class of a query with more: redone from the root
constexpr uint32_t kHeavyClass = 4;     // classes >= this are dealt across wavefronts

// Sort key of a continuation (16 bits, ascending = processed first):
//   classes >= kRankedClass ("ranked"): 0 .. 0x1FFF, smaller for a LARGER home-leaf best distance.
//     tools/analyse_cost.py: these 2 % of the queries hold every expensive one, and how far the
//     home-leaf best is predicts the cost (every query with > 500 phase-2 steps is in the top half);
//   classes 4 .. 1: (7 - class) << 13, low bits 0 so the stable sort keeps their Morton order;
//   class 0 (final after phase 1): 7 << 13, sorts behind everything.
typedef uint16_t ContKey;
constexpr uint32_t kRankedClass = 5;
__device__ __forceinline__ ContKey make_cont_key(uint32_t cls, float best_d) {
  if (cls >= kRankedClass) return (ContKey)(0x1FFFu - ((__float_as_uint(best_d) >> 18) & 0x1FFFu));
  return (ContKey)((7u - cls) << 13);
}
__device__ __forceinline__ bool cont_key_is_final(ContKey k) { return (k >> 13) == 7u; }

// Continuations are indexed by the query's slot in the (Morton-ordered) packed batch, so
// phase 1 needs no allocation and no atomics; the class sort doubles as the compaction.
struct Cont {
  uint4* best;          // [nq]  {best index, bits(best distance), class, 0} after phase 1
  Record* rec;          // [kContSlots][nq]  record s of slot i at rec[s * nq + i] (a wave stores and
                        //                   loads one record index as one coalesced run), shallowest first
  ContKey* key;         // [nq]  sort key of the slot, see make_cont_key()
  uint32_t* ids;        // [nq]  slot index (sorted together with key)
  uint32_t* meta;       // [kMetaWords], written by knn1_phase_meta_kernel (layout there)
  uint64_t nq;          // slots (= queries of the batch)
  __device__ __forceinline__ Record& record(uint32_t slot, uint32_t s) const { return rec[(uint64_t)s * nq + slot]; }
};

// Phase 1, with a wave-uniform prefix.  The batch is spatially sorted, so the 64
// queries of a wavefront walk the SAME branches for most of the way down (measured: they part
// ways 4-6 levels above the leaves).  profiles/r01d: phase 1 is bound by the rate at which the
// vector-memory pipe takes divergent 16-byte loads (59 per wave, ~48 cycles each per CU), so while
// all lanes agree the node is fetched ONCE, through the scalar cache (s_load_dwordx4 on a
// readfirstlane'd index), and only the per-lane arithmetic stays on the vector unit.  A ballot
// after each step tells whether the lanes still agree.  One descent, no LDS: the far children that can still
// matter once the home leaf has given a bound are chosen from the few with the smallest box distances, kept in
// registers on the way down.
__device__ __forceinline__ uint32_t uniform_value(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// The launch-order query records {x, y, z, bits(row)} of phase 2 are made here as well (gather through `perm`,
// or the identity if it is null, + one coalesced 16-byte store) instead of by a packing pass of their own.
// M: metric_l2_squared, or (r06) metric_l1 -- the per-axis term of a split offset and of a point distance is the
// metric's, everything else (the tree, `nbd - old + new`, the strict compares) is metric-free.  The cooperative search
// and its certificate need a box distance that is a lower bound of the point distances of the subtree: true of the two
// sums, NOT of metric_lpinf / metric_lninf (the reference sums the per-axis offsets for every Lp metric,
// kd_tree_search.hpp:91-92, while their point distance is a maximum / minimum): those keep the general kernel.
template <int LEAFB, class M = MetricL2>
__global__ __launch_bounds__(64) void knn1_phase1u_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim, const uint32_t* __restrict__ perm, uint64_t nq,
    float e_inv, Neighbor* __restrict__ out, Cont cont, float4* __restrict__ qs_out,
    uint32_t* __restrict__ tile_counts = nullptr, uint32_t count_stride = 0,
    const uint32_t* __restrict__ as_given = nullptr) {
  PTK_TRACE_BEGIN_SEL(1);
  const uint64_t i0 = (uint64_t)xcd_runs(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  const bool valid = i0 < nq;
  const uint64_t i = valid ? i0 : nq - 1;  // idle lanes shadow the last query: ballots stay full-width
  const uint4* __restrict__ nodes = t.nodes;
  const float4* __restrict__ pts = t.pts;
  float qx, qy, qz;
  // (as_given: the verdict of the coherence sample -- 1 = the batch came in a coherent order, the sort left at once
  // and `perm` holds nothing)
  if (as_given != nullptr && *as_given != 0u) perm = nullptr;
  const uint32_t qi = perm ? perm[i] : (uint32_t)i;
  load_query(queries, dim, qi, qx, qy, qz);
  if (valid) qs_out[i] = make_float4(qx, qy, qz, __uint_as_float(qi));

  NnPolicy pol;
  pol.e_inv = e_inv;
  pol.out = out;
  pol.begin_query(qi);

  // ---- the descent to the home leaf ----
  // Which far children can still matter is only known once the home leaf has given a bound, and a wave cannot
  // afford a record per level (24+ levels x 8 bytes x 64 lanes of LDS, or a second descent: 12 more divergent
  // 16-byte gathers per query on the unit that bounds this kernel, profiles/r01d).  But a continuation carries at
  // most kContSlots records, and the far children that pass `best >= box distance` are those with the SMALLEST
  // box distances: the kContSlots + 1 smallest seen so far are kept in registers, sorted (an insertion is seven
  // compare / select steps).  If even the last of them passes, more than kContSlots may: the overflow class.
  constexpr int kCand = kContSlots + 1;
  float cand_d[kCand];
  uint32_t cand_m[kCand];
#pragma unroll
  for (int j = 0; j < kCand; ++j) {
    cand_d[j] = __uint_as_float(0x7F800000u);
    cand_m[j] = 0xFFFFFFFFu;
  }
  auto candidate = [&](uint32_t idx, uint32_t axis, float left_max, float right_min, bool go_left) {
    const float v = sel3(axis, qx, qy, qz);
    const float dv = f_sub(go_left ? right_min : left_max, v);
    // Descent state: box distance 0, offsets 0 => (0 - 0) + new_off, as the reference computes it.
    const float d = f_add(f_sub(0.0f, 0.0f), M::one(dv));
    const uint32_t m = idx | (axis << 28) | (go_left ? kRecSide : 0u);
#pragma unroll
    for (int j = kCand - 1; j >= 1; --j) {
      const bool shift = d < cand_d[j - 1];
      const bool here = d < cand_d[j];
      cand_m[j] = shift ? cand_m[j - 1] : (here ? m : cand_m[j]);
      cand_d[j] = shift ? cand_d[j - 1] : (here ? d : cand_d[j]);
    }
    if (d < cand_d[0]) {
      cand_d[0] = d;
      cand_m[0] = m;
    }
  };
  uint32_t ref = t.root_ref;
  {
    bool together = true;
    while (together && !(ref & kLeafBit)) {
      const uint32_t uref = uniform_value(ref);
      const uint32_t idx = uref & kBranchIdxMask;
      const uint4 nd = nodes[idx];
      const uint32_t axis = (uref >> 29) & 3u;
      const float left_max = __uint_as_float(nd.x);
      const float right_min = __uint_as_float(nd.y);
      const float v = sel3(axis, qx, qy, qz);
      const bool go_left = f_sub(f_sub(f_add(left_max, right_min), v), v) > 0.0f;
      candidate(idx, axis, left_max, right_min, go_left);
      ref = go_left ? nd.z : nd.w;
      const uint64_t b = __ballot(go_left);
      together = b == 0ull || b == ~0ull;
    }
    while (!(ref & kLeafBit)) {
      const uint32_t idx = ref & kBranchIdxMask;
      const uint4 nd = nodes[idx];
      const uint32_t axis = (ref >> 29) & 3u;
      const float left_max = __uint_as_float(nd.x);
      const float right_min = __uint_as_float(nd.y);
      const float v = sel3(axis, qx, qy, qz);
      const bool go_left = f_sub(f_sub(f_add(left_max, right_min), v), v) > 0.0f;
      candidate(idx, axis, left_max, right_min, go_left);
      ref = go_left ? nd.z : nd.w;
    }
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
          float dx = f_sub(qx, p[u].x);
          float dy = f_sub(qy, p[u].y);
          float dz = f_sub(qz, p[u].z);
          PTK_SCALAR(dx);
          PTK_SCALAR(dy);
          PTK_SCALAR(dz);
          pol.visit(__float_as_int(p[u].w), point_distance3<M>(dx, dy, dz));
        }
      }
    }
  }

  // ---- the far children that can still matter, shallowest first ----
  // The candidates are sorted by box distance, so those that pass are a prefix; along a root-to-leaf path the
  // branch numbers of the depth-first layout increase, so sorting the survivors by branch number is sorting them
  // by depth (a 12-comparator network; the slots behind the survivors sort to the end).
  Record keep[kContSlots];
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < kCand; ++j) c += pol.max() >= cand_d[j] ? 1u : 0u;
  // A point AT the query: nothing can be strictly nearer, so whatever the reference still visits (every subtree that
  // touches the query -- on data snapped to a grid a pile of hundreds of coincident points) cannot change its answer.
  c = pol.settled() ? 0u : c;
  {
    uint32_t key[kContSlots];
#pragma unroll
    for (int j = 0; j < kContSlots; ++j) {
      keep[j].x = cand_m[j];
      keep[j].y = __float_as_uint(cand_d[j]);
      key[j] = (uint32_t)j < c ? (cand_m[j] & kRecIdxMask) : 0xFFFFFFFFu;
    }
    auto order = [&](int a, int b) {
      const bool swap = key[a] > key[b];
      const uint32_t ka = key[a], kb = key[b];
      const Record ra = keep[a], rb = keep[b];
      key[a] = swap ? kb : ka;
      key[b] = swap ? ka : kb;
      keep[a].x = swap ? rb.x : ra.x;
      keep[a].y = swap ? rb.y : ra.y;
      keep[b].x = swap ? ra.x : rb.x;
      keep[b].y = swap ? ra.y : rb.y;
    };
    static_assert(kContSlots == 6, "the sorting network below is the one for six");
    order(0, 5); order(1, 3); order(2, 4);
    order(1, 2); order(3, 4);
    order(0, 3); order(2, 5);
    order(0, 1); order(2, 3); order(4, 5);
    order(1, 2); order(3, 4);
  }
  // ---- the reference's next steps while they are leaves ----
  // The reference now pops the deepest record, tests it against the best so far and enters the far child.
  // While that child is a LEAF (the sibling of the home leaf, more often than not) the step needs no stack:
  // it is taken here, in the reference's order, and the better best it leaves behind drops records that the
  // reference would skip when it pops them (the best only shrinks) -- for many queries all of them.
  if (c <= (uint32_t)kContSlots) {
    while (c > 0u) {
      Record r = keep[0];
#pragma unroll
      for (int s = 1; s < kContSlots; ++s) {
        if (c - 1u == (uint32_t)s) r = keep[s];
      }
      if (pol.max() >= __uint_as_float(r.y)) {
        const uint4 nd = nodes[r.x & kRecIdxMask];
        const uint32_t far = (r.x & kRecSide) ? nd.w : nd.z;
        if (!(far & kLeafBit)) break;
        const uint32_t lv = far & 0x7FFFFFFFu;
        const uint32_t begin = lv >> t.cbits;
        const uint32_t count = lv & t.cmask;
        for (uint32_t j = 0; j < count; j += LEAFB) {
          float4 p[LEAFB];
#pragma unroll
          for (int u = 0; u < LEAFB; ++u) p[u] = pts[begin + j + u];
#pragma unroll
          for (int u = 0; u < LEAFB; ++u) {
            if (j + u < count) {
              float dx = f_sub(qx, p[u].x);
              float dy = f_sub(qy, p[u].y);
              float dz = f_sub(qz, p[u].z);
              PTK_SCALAR(dx);
              PTK_SCALAR(dy);
              PTK_SCALAR(dz);
              pol.visit(__float_as_int(p[u].w), point_distance3<M>(dx, dy, dz));
            }
          }
        }
      }
      --c;
    }
    // What is left, still shallowest first, without the records the best has overtaken.
    uint32_t kept = 0;
#pragma unroll
    for (int s = 0; s < kContSlots; ++s) {
      const Record r = keep[s];
      if ((uint32_t)s < c && pol.max() >= __uint_as_float(r.y)) {
#pragma unroll
        for (int d = 0; d <= s; ++d) {
          if (kept == (uint32_t)d) keep[d] = r;
        }
        ++kept;
      }
    }
    c = kept;
  }
  const uint32_t cls = c > (uint32_t)kContSlots ? kContOverflow : c;
  const ContKey ckey = make_cont_key(cls, pol.best_d);
  if (tile_counts != nullptr) {
    // The class order is a counting sort over the three class bits of the key (class_order_kernel below): this
    // wavefront's 64 slots are one tile of it, counted here (lane b stores bucket b).
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t b = 0; b < 8; ++b) {
      const uint32_t n = (uint32_t)__popcll(__ballot(valid && (uint32_t)(ckey >> 13) == b));
      mine = threadIdx.x == b ? n : mine;
    }
    if (threadIdx.x < 8u) tile_counts[(uint64_t)threadIdx.x * count_stride + (uint32_t)(i0 / 64u)] = mine;
  }
  if (!valid) return;
  const uint32_t e = (uint32_t)i;
  cont.key[e] = ckey;
  if (cont.ids != nullptr) cont.ids[e] = e;  // (the counting sort numbers the slots itself)
  if (cls == 0) {
    pol.end_query(qi);  // nothing else can be nearer: the home-leaf best is the answer
  } else {
    cont.best[e] = make_uint4((uint32_t)pol.best_i, __float_as_uint(pol.best_d), cls, 0u);
  }
  if (cls != 0 && cls != kContOverflow) {
#pragma unroll
    for (int s = 0; s < kContSlots; ++s) {
      if ((uint32_t)s < c) cont.record(e, (uint32_t)s) = keep[s];
    }
  }
  PTK_TRACE_END_ANY();
}

// One thread, after the sort: where the tiers of the sorted list end.
//   meta[0] n2      continuations (everything before class 0)
//   meta[1] heavy   end of the dealt tier (classes >= kHeavyClass)
//   meta[2] waves of the dealt tier
//   meta[3] entries of the ranked classes (the head of the list)
//   meta[4] end of the narrow tiers     meta[5] their waves in total
//   meta[8 + 4 i ..]  narrow tier i: {first entry, end entry, lanes per wave, first wave}
// The narrow tiers cut the head of the ranked classes at cumulative per-mille marks.
//   meta[24] queries phase 2 gave up on (cap reached; listed for knn1_coop_kernel)
//   meta[26] queries the cooperative search could not certify
constexpr uint32_t kMaxTiers = 4;
constexpr uint32_t kMetaWords = 32;  // size of Cont::meta (8 fixed words + 4 per narrow tier + 3 counters, rounded up)
struct TierSpec {
  uint32_t permille[kMaxTiers];  // cumulative share of the ranked group where tier i ends (0 = unused)
  uint32_t lanes[kMaxTiers];
};

// The tier table of phase 2 from the three boundaries of the class-sorted list.
// skip_ranked: the ranked classes (>= kRankedClass, the head of the list) are not phase 2's -- the cooperative
// search takes them straight from phase 1 (knn1_coop_kernel<.., DIRECT>): the dealt tier begins behind them.
__device__ inline void write_phase_meta(const Cont& cont, uint32_t n2, uint32_t ranked, uint32_t heavy,
                                        const TierSpec& tiers, uint32_t max_narrow_waves, bool skip_ranked = false) {
  uint32_t begin = skip_ranked ? ranked : 0u, wave = 0;
  for (uint32_t i = 0; i < kMaxTiers; ++i) {
    uint32_t end = (uint32_t)(((uint64_t)ranked * tiers.permille[i]) / 1000u);
    if (end > heavy) end = heavy;
    if (end < begin) end = begin;
    uint32_t hl = tiers.lanes[i] < 1u || tiers.lanes[i] > 64u ? 64u : tiers.lanes[i];
    uint32_t waves = (end - begin + hl - 1u) / hl;
    if (wave + waves > max_narrow_waves) {  // the launch has no room: this tier ends early
      waves = max_narrow_waves - wave;
      end = begin + waves * hl;
    }
    cont.meta[8 + 4 * i + 0] = begin;
    cont.meta[8 + 4 * i + 1] = end;
    cont.meta[8 + 4 * i + 2] = hl;
    cont.meta[8 + 4 * i + 3] = wave;
    begin = end;
    wave += waves;
  }
  cont.meta[0] = n2;
  cont.meta[1] = heavy;
  cont.meta[2] = (heavy - begin + 63u) / 64u;
  cont.meta[kMetaRanked] = ranked;
  cont.meta[4] = begin;
  cont.meta[5] = wave;
  cont.meta[kMetaHeavy] = 0;
  cont.meta[kMetaRedo] = 0;
}

// Behind a full 16-bit radix sort of the keys (searches without the cap): the boundaries by binary search.
static_assert(kHeavyClass < kRankedClass && kHeavyClass >= 1u, "the dealt tier begins inside the unranked classes");
PTK_GLOBAL void knn1_phase_meta_kernel(const ContKey* __restrict__ sorted_key, uint32_t nq, Cont cont,
                                       TierSpec tiers, uint32_t max_narrow_waves) {
  auto first_at_least = [&](uint32_t k) {  // sorted_key is ascending
    uint32_t lo = 0, hi = nq;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (sorted_key[mid] < k) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const uint32_t n2 = first_at_least(7u << 13);
  const uint32_t ranked = first_at_least(1u << 13);  // classes >= kRankedClass
  const uint32_t heavy = first_at_least((7u - kHeavyClass + 1u) << 13);
  write_phase_meta(cont, n2, ranked, heavy, tiers, max_narrow_waves);
}

// ---- the class order as a counting sort ----------------------------------------------------------------
// With the cap only the three class bits of a key matter, i.e. 8 buckets: a stable counting sort over the 2-byte
// keys -- count per tile, scan of the 8 rows of counters, stable scatter of the slot numbers -- instead of a general
// radix sort of (key, value) pairs with its histogram, look-back state and their memsets; and the tier table comes
// from the scanned counters (where a bucket starts) instead of binary searches.  Everything wave-synchronous
// (ballots), no LDS.
constexpr uint32_t kClassBuckets = 8;

// ---- the class order from the tile counts of phase 1 (the shipped form) ---------------------------------------
// Phase 1 leaves counts[b * stride + tile] = slots of bucket b in tile `tile` (64 consecutive slots).  Two launches
// instead of count + library scan (two launches of its own) + scatter:
//   class_scan_kernel     the rows are cut into segments of `seg` tiles (a multiple of 4), one wavefront each:
//                         exclusive prefix inside the segment, in place; segment total -> seg_totals[b * segs + s]
//   class_order_kernel    chunk of `per` slots per wavefront, as class_scatter_kernel; where bucket b of the chunk
//                         goes = (everything in earlier buckets and earlier segments: a prefix over the flattened
//                         bucket-major table of segment totals, at most 128 words) + the prefix of its first tile
constexpr uint32_t kClassMaxSegs = 16;

PTK_GLOBAL __launch_bounds__(64) void class_scan_kernel(uint32_t* __restrict__ counts, uint32_t ntiles, uint32_t stride,
                                                        uint32_t seg, uint32_t* __restrict__ seg_totals) {
  const uint32_t lane = threadIdx.x;
  const uint32_t segs = gridDim.x / kClassBuckets;
  const uint32_t b = blockIdx.x / segs, sg = blockIdx.x % segs;
  const uint32_t t0 = sg * seg;                                    // first tile of the segment (a multiple of 4)
  const uint32_t t1 = t0 + seg < ntiles ? t0 + seg : ntiles;       // one past its last tile
  uint4* row = reinterpret_cast<uint4*>(counts + (uint64_t)b * stride + t0);
  const uint32_t n4 = (t1 - t0 + 3u) / 4u;                         // (stride is a multiple of 4: the row has room)
  uint32_t carry = 0;
  for (uint32_t c0 = 0; c0 < n4; c0 += 256u) {
    uint4 v[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const uint32_t i4 = c0 + lane * 4u + u;
      v[u] = i4 < n4 ? row[i4] : make_uint4(0u, 0u, 0u, 0u);
      const uint32_t e = t0 + i4 * 4u;  // (what lies behind the last tile was never written)
      v[u].x = e + 0u < t1 ? v[u].x : 0u;
      v[u].y = e + 1u < t1 ? v[u].y : 0u;
      v[u].z = e + 2u < t1 ? v[u].z : 0u;
      v[u].w = e + 3u < t1 ? v[u].w : 0u;
    }
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) sum += v[u].x + v[u].y + v[u].z + v[u].w;
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
      if ((int)lane >= d) incl += up;
    }
    uint32_t run = carry + incl - sum;
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      uint4 o;
      o.x = run;
      o.y = o.x + v[u].x;
      o.z = o.y + v[u].y;
      o.w = o.z + v[u].z;
      run = o.w + v[u].w;
      const uint32_t i4 = c0 + lane * 4u + u;
      if (i4 < n4) row[i4] = o;
    }
    carry += (uint32_t)__shfl((int)incl, 63);
  }
  if (lane == 0u) seg_totals[blockIdx.x] = carry;
}

// `prefix` = the counts after class_scan_kernel.  per: slots per chunk, a multiple of 64.
PTK_GLOBAL __launch_bounds__(64) void class_order_kernel(const ContKey* __restrict__ keys, uint32_t nq, uint32_t per,
                                                         const uint32_t* __restrict__ prefix, uint32_t stride,
                                                         uint32_t seg, uint32_t segs,
                                                         const uint32_t* __restrict__ seg_totals,
                                                         uint32_t* __restrict__ sorted_ids, Cont cont, TierSpec tiers,
                                                         uint32_t max_narrow_waves, uint32_t skip_ranked) {
  const uint32_t chunk = blockIdx.x, lane = threadIdx.x;
  const uint32_t lo = chunk * per;
  const uint32_t hi = lo + per < nq ? lo + per : nq;
  const uint64_t below = (1ull << lane) - 1ull;
  const uint32_t tile0 = lo / 64u, seg0 = tile0 / seg;
  // Exclusive prefix over the flattened table seg_totals[b * segs + s] (at most 8 x 16 words: two per lane).
  const uint32_t n_tab = kClassBuckets * segs;
  const uint32_t a0 = 2u * lane < n_tab ? seg_totals[2u * lane] : 0u;
  const uint32_t a1 = 2u * lane + 1u < n_tab ? seg_totals[2u * lane + 1u] : 0u;
  uint32_t incl = a0 + a1;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
    if ((int)lane >= d) incl += up;
  }
  const uint32_t ex0 = incl - a0 - a1, ex1 = ex0 + a0;  // exclusive prefixes of table entries 2 lane, 2 lane + 1
  uint32_t base[kClassBuckets];   // next free position of every bucket for this chunk
  uint32_t start[kClassBuckets];  // where every bucket starts in the list
#pragma unroll
  for (uint32_t b = 0; b < kClassBuckets; ++b) {
    // (f and f0 are the same in every lane: the holder of table entry f is lane f / 2, the parity picks its word)
    const uint32_t f = b * segs + seg0, f0 = b * segs;
    base[b] = (uint32_t)__shfl((int)((f & 1u) ? ex1 : ex0), (int)(f >> 1)) + prefix[(uint64_t)b * stride + tile0];
    start[b] = (uint32_t)__shfl((int)((f0 & 1u) ? ex1 : ex0), (int)(f0 >> 1));
  }
  if (chunk == 0 && lane == 0) {
    // n2 = everything before class 0, ranked = classes >= kRankedClass, heavy = classes >= kHeavyClass
    write_phase_meta(cont, start[7], start[1], start[7u - kHeavyClass + 1u], tiers, max_narrow_waves, skip_ranked != 0u);
  }
  for (uint32_t i = lo; i < hi; i += 256u) {
    uint32_t k[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const uint32_t j = i + u * 64u + lane;
      k[u] = j < hi ? (uint32_t)(keys[j] >> 13) : kClassBuckets;
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {  // tiles in slot order: the sort is stable
      uint32_t pos = 0;
#pragma unroll
      for (uint32_t b = 0; b < kClassBuckets; ++b) {
        const uint64_t m = __ballot(k[u] == b);
        if (k[u] == b) pos = base[b] + (uint32_t)__popcll(m & below);
        base[b] += (uint32_t)__popcll(m);
      }
      const uint32_t j = i + u * 64u + lane;
      if (j < hi) sorted_ids[pos] = j;
    }
  }
}

// Phase 2: one continuation per lane, taken from the class-sorted entry list.
// cap > 0 (exact searches only): a query that has entered `cap` far children and still has work
// left stops here -- its best so far goes back to cont.best and its slot onto heavy_list, for the
// cooperative search below.  The few thousand queries this concerns are dependent chains of
// hundreds of leaf visits which used to decide the duration of the whole launch.
template <int S, int OVF, int LEAFB, class M = MetricL2>
__global__ __launch_bounds__(64) void knn1_phase2_kernel(
    DevTree t, const float4* __restrict__ qs, float e_inv, Neighbor* __restrict__ out, Cont cont,
    const uint32_t* __restrict__ sorted_ids, uint32_t cap = 0, Handover ho = Handover{}) {
  PTK_TRACE_BEGIN_SEL(2);
  const uint32_t n2 = cont.meta[0];
  const uint32_t heavy = cont.meta[1];
  const uint32_t heavy_waves = cont.meta[2];
  const uint32_t top = cont.meta[4];
  const uint32_t top_waves = cont.meta[5];
  const uint32_t wave = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  // Three tiers of the sorted list, most expensive first (blocks are dispatched in order):
  //   narrow the head of the ranked classes in up to kMaxTiers tiers of few lanes per wavefront:
  //          these queries are long dependent chains (hundreds of leaves); a wave's round costs
  //          the maximum over its lanes, so fewer lanes = shorter rounds on what is the critical
  //          path of the launch (empty tiers have first wave == the next tier's first wave);
  //   dealt  the other heavy classes: lane l of wave w takes entry l * waves + w, so neighbours in
  //          the list (similar cost, spatially adjacent) land in different wavefronts;
  //   light  64 consecutive entries per wavefront (Morton order within a class).
  uint32_t s;
  bool valid;
  if (wave < top_waves) {
    uint32_t tier = 0;
#pragma unroll
    for (uint32_t i = 1; i < kMaxTiers; ++i) {
      if (wave >= cont.meta[8 + 4 * i + 3]) tier = i;
    }
    const uint32_t hl = cont.meta[8 + 4 * tier + 2];
    s = cont.meta[8 + 4 * tier + 0] + (wave - cont.meta[8 + 4 * tier + 3]) * hl + lane;
    valid = lane < hl && s < cont.meta[8 + 4 * tier + 1];
  } else if (wave < top_waves + heavy_waves) {
    s = top + lane * heavy_waves + (wave - top_waves);
    valid = s < heavy;
  } else {
    s = heavy + (wave - top_waves - heavy_waves) * 64u + lane;
    valid = s < n2;
  }
  if (!valid) return;
  const uint32_t e = sorted_ids[s];
  const float4 qrec = qs[e];
  const float qx = qrec.x, qy = qrec.y, qz = qrec.z;
  const uint32_t qi = __float_as_uint(qrec.w);
  const uint4 start = cont.best[e];
  const uint32_t cls = start.z;

  Record spill[OVF > 0 ? OVF : 1];
  Stack<S, OVF, 64> st;
  st.init((LdsWord*)ptk_smem, threadIdx.x, spill);
  NnPolicy pol;
  pol.e_inv = e_inv;
  pol.out = out;
  bool finished;
  ho.slot = e;
  if (cls == kContOverflow) {
    pol.begin_query(qi);
    finished = cap ? traverse<LEAFB, false, M, true>(t, qx, qy, qz, pol, st, cap, &ho)
                   : traverse<LEAFB, false, M>(t, qx, qy, qz, pol, st);
  } else {
    pol.best_i = (int32_t)start.x;
    pol.best_d = __uint_as_float(start.y);
    for (uint32_t j = 0; j < cls; ++j) {
      const Record r = cont.record(e, j);
      st.push(r.x, __uint_as_float(r.y));
    }
    finished = cap ? traverse<LEAFB, true, M, true>(t, qx, qy, qz, pol, st, cap, &ho)
                   : traverse<LEAFB, true, M>(t, qx, qy, qz, pol, st);
  }
  if (finished) {
    pol.end_query(qi);
  } else {  // listed by the traversal; the cooperative search starts from the best so far
    cont.best[e] = make_uint4((uint32_t)pol.best_i, __float_as_uint(pol.best_d), cls, 0u);
  }
  PTK_TRACE_END_ANY();
}

// ---- cooperative search of the queries phase 2 gave up on ---------------------------------------
//
// What is left after the cap are dependent chains: the reference's depth-first search of such a
// query visits hundreds of leaves one after the other (its bound tightens slowly -- the query sits
// in an empty region with a ring of points all about equally far away), and no lane-per-query
// schedule can shorten a chain.  Here G lanes work on ONE query: they share a LIFO pool of
// subtrees still to be searched (it starts with what the capped traversal handed over) and the
// best distance found so far, and every lane advances its own subtree by one node per step (a
// branch: keep the far child for later if it can still matter, go near; a leaf: measure four
// points), so a step is one memory round trip for up to G nodes of the same query.
//
// This is NOT the reference's visit order, so what makes the result the reference's?  The
// reference (kd_tree_search.hpp:52-105 + search_visitor.hpp:42-65) reports, of the points it
// visits, the first in its depth-first order that attains the smallest distance.  Let d* be the
// smallest float distance over ALL points of the tree and p* the point at d* that comes first in
// that depth-first order (near child before far child, a leaf in index order).  If every far child
// on the path from the root to p*'s leaf has a box distance <= d* -- the reference enters a far
// child when `max() >= box distance`, and its max() is never below d* -- then the reference
// reaches p*'s leaf whatever it visited before, accepts p* (a strict improvement on anything else
// it can hold then) and keeps it: its answer is exactly (p*, d*).  The condition holds unless a
// float box distance exceeds the float distance of a point inside the box (rounding); when it
// fails the query is listed for knn1_redo_kernel, which replays the reference search.
//
// So the cooperative search has to find the float minimum over all points, and among several
// points at the minimum the first in depth-first order (dfs_before(), a descent from the root
// that only runs on exact ties; more than a few ties per query -> redo).  It prunes a subtree
// only when its box distance exceeds best * (1 + 2^-10): the float box distance (a chain of at
// most 2 roundings per level) exceeds the exact one by a relative 2^-12 at most for trees less
// than ~2000 levels deep, and a float point distance is within 5 roundings of the exact one, so a
// pruned subtree holds no point at a float distance <= best -- neither the minimum nor a tie for
// it.  Every subtree carries the largest box distance met on its path (`gmax`), which is what the
// condition above compares with d*.  Searches with a best below 1e-30 or above 1e30 (products
// underflow or overflow; the error bounds assume neither) are redone as well.  The points the
// capped traversal had visited before it stopped need no second look: the reference has visited
// them too, so if one of them is p* it is the best handed over, and it wins ties against
// anything found here (it came first).
//
// Group i of the launch takes entries i, i + groups, ... of the list.

// True if, in the reference's depth-first order for query q, the point at record position `a`
// comes before the one at `b` (a != b): they part ways at some branch, where the nearer child is
// searched first; inside one leaf the lower position is visited first.
__device__ inline bool dfs_before(const DevTree& t, const uint2* __restrict__ ranges, float qx, float qy, float qz,
                                  uint32_t a, uint32_t b) {
  uint32_t ref = t.root_ref;
  for (;;) {
    if (ref & kLeafBit) return a < b;
    const uint32_t axis = (ref >> 29) & 3u;
    const uint4 nd = t.nodes[ref & kBranchIdxMask];
    uint32_t mid;  // first record of the right child
    if (nd.w & kLeafBit) {
      mid = (nd.w & 0x7FFFFFFFu) >> t.cbits;
    } else {
      mid = ranges[nd.w & kBranchIdxMask].x;
    }
    const bool a_left = a < mid, b_left = b < mid;
    if (a_left == b_left) {
      ref = a_left ? nd.z : nd.w;
      continue;
    }
    const float v = sel3(axis, qx, qy, qz);
    const bool go_left = f_sub(f_sub(f_add(__uint_as_float(nd.x), __uint_as_float(nd.y)), v), v) > 0.0f;
    return go_left == a_left;
  }
}

constexpr uint32_t kCoopTieBudget = 6;  // exact ties a lane resolves per query before it asks for a redo

// range: 0 = the whole list, 1 = what the heavy tiers listed, 2 = what the light tier added.
// DIRECT: the list is the head of the class-sorted entry list -- the ranked classes, which hold every expensive
// query (tools/analyse_cost.py) -- and a query's subtrees are its continuation records as phase 1 left them (the
// pending-record form a capped traversal hands over: offsets 0, box distance 0 above them).  On a small batch the
// capped traversal of such a query and its cooperative search afterwards are two chains of dependent rounds, one
// after the other, on a GPU that is mostly idle; taken straight from phase 1 the query is searched once, beside
// phase 2 (which then runs without the ranked classes) on a second stream.
//
// A group's pool of subtrees lives in LDS (POOL tasks); with SPILL what does not fit goes to the group's run of
// `spill_cap` tasks in HBM and comes back when the pool has drained (the order of the visits is free, see above).  A search
// that starts from the home-leaf bound only (DIRECT) keeps many more subtrees alive than one that a capped
// traversal has tightened first: without the spill 39 of 12 139 ranked queries of a 900 k-query shard overflowed
// a pool of 96 and their single-lane replays took 1.3 ms (profiles/r03_notes.txt item 3).
template <int G, int POOL, bool DIRECT = false, bool SPILL = DIRECT, class M = MetricL2>
__global__ __launch_bounds__(64) void knn1_coop_kernel(
    DevTree t, const uint2* __restrict__ ranges, const float4* __restrict__ qs, Neighbor* __restrict__ out, Cont cont,
    Handover ho, uint32_t* __restrict__ redo_list, const uint32_t* __restrict__ sorted_ids = nullptr,
    Task* __restrict__ spill = nullptr, uint32_t spill_cap = 0) {
  static_assert(G == 8 || G == 16 || G == 32 || G == 64, "lanes per query");
  static_assert(DIRECT ? POOL >= 2 * kContSlots : POOL >= (int)kMaxTasks, "the pool must hold what a query starts with");
  PTK_TRACE_BEGIN_SEL(DIRECT ? 3 : 4);
  constexpr int NG = 64 / G;
  typedef PTK_LDS uint32_t LdsU32;
  const uint4* __restrict__ nodes = t.nodes;
  const float4* __restrict__ pts = t.pts;
  const uint32_t lane = threadIdx.x;
  const uint32_t g = lane / G, gl = lane % G;
  const uint64_t gmask = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << (g * G);
  const uint64_t below = gmask & ((1ull << lane) - 1ull);  // lanes of this group before this one
  LdsU32* pool = (LdsU32*)ptk_smem + g * (6 * POOL);       // [field][slot] of this group
  LdsU32* gbest = (LdsU32*)ptk_smem + NG * (6 * POOL) + g;  // bits of the group's best distance
  LdsU32* gsecond = gbest + NG;                             // (when a query ends) the nearest of all OTHER points
  const uint32_t n_heavy = DIRECT ? cont.meta[kMetaRanked] : cont.meta[ho.counter];
  Task* const spill_g = spill + (uint64_t)(blockIdx.x * (uint32_t)NG + g) * spill_cap;  // this group's run
  uint32_t spill_n = 0;  // tasks in it (the same value in every lane of the group)

  bool have = false, exhausted = false, busy = false, failed = false;
  uint32_t count = 0;  // subtrees in the pool (the same value in every lane of the group)
  float qx = 0.0f, qy = 0.0f, qz = 0.0f;
  uint32_t e = 0, qi = 0;
  uint32_t ref = 0;
  float nbd = 0.0f, off0 = 0.0f, off1 = 0.0f, off2 = 0.0f, gmax = 0.0f;
  // This lane's best: distance, record position, "gmax <= cd"; ties it may still resolve.
  float cd = 3.402823466e+38f;
  uint32_t cpos = 0;
  bool cok = false;
  float cg = 0.0f;                  // the largest box distance on the way to this lane's best (what cok compared)
  float cd2 = 3.402823466e+38f;     // the nearest point this lane has measured other than its best
  uint32_t tie_budget = 0;
  float start_d = 0.0f;  // the best handed over (phase 2's own, already the reference's)
  uint32_t start_i = 0;
  uint32_t next_idx = blockIdx.x * (uint32_t)NG + g;  // this group's next entry of the list

  for (;;) {
    // Groups without a query take the next one.
    const bool need = !have && !exhausted;
    if (__ballot(need) != 0ull) {
      if (need) {
        // Entry `group`, `group + groups`, ... of the list: a shared counter would hand the queries
        // out more evenly, but one atomic per query on one address is what bounds the launch then
        // (measured: ~11 ns per query whatever the number of waves).
        const uint32_t idx = next_idx;
        next_idx += gridDim.x * (uint32_t)NG;
        if (idx < n_heavy) {
          e = DIRECT ? sorted_ids[idx] : ho.heavy_list[idx];
          const float4 qrec = qs[e];
          qx = qrec.x;
          qy = qrec.y;
          qz = qrec.z;
          qi = __float_as_uint(qrec.w);
          const uint4 st = cont.best[e];
          start_i = st.x;
          start_d = __uint_as_float(st.y);
          const uint32_t nt = DIRECT ? (st.z == kContOverflow ? kTasksFromRoot : st.z) : ho.ntasks[idx];
          have = true;
          spill_n = 0;
          busy = false;
          // A best of exactly 0 cannot be improved on: nothing to search.
          failed = (start_d != 0.0f && !(start_d >= 1e-30f && start_d <= 1e30f)) || nt == kTasksRedo;
          cd = 3.402823466e+38f;
          cpos = 0;
          cok = false;
          cg = 0.0f;
          cd2 = 3.402823466e+38f;
          tie_budget = kCoopTieBudget;
          if (failed || start_d == 0.0f) {
            count = 0;
          } else if (nt == kTasksFromRoot) {  // everything again, the points already visited included
            count = 1;
            if (gl == 0) {
              pool[0 * POOL] = t.root_ref;
              pool[1 * POOL] = 0u;  // box distance 0, offsets 0, no far child above
              pool[2 * POOL] = 0u;
              pool[3 * POOL] = 0u;
              pool[4 * POOL] = 0u;
              pool[5 * POOL] = 0u;
            }
          } else if (DIRECT) {  // the continuation records, shallowest first: the deepest is the next to visit
            count = nt;
            for (uint32_t i = gl; i < nt; i += G) {
              const Record r = cont.record(e, i);
              pool[0 * POOL + i] = r.x;
              pool[1 * POOL + i] = r.y;
              pool[2 * POOL + i] = 0u;
              pool[3 * POOL + i] = 0u;
              pool[4 * POOL + i] = 0u;
              pool[5 * POOL + i] = r.y | 0x80000000u;  // largest box distance so far = its own; pending-record form
            }
          } else {  // the capped traversal's stack, next-to-visit on top
            count = nt;
            const Task* src = ho.tasks + (uint64_t)idx * kMaxTasks;
            for (uint32_t i = gl; i < nt; i += G) {
              const Task k = src[i];
              const uint32_t sl = nt - 1u - i;
              pool[0 * POOL + sl] = k.ref;
              pool[1 * POOL + sl] = __float_as_uint(k.nbd);
              pool[2 * POOL + sl] = __float_as_uint(k.off0);
              pool[3 * POOL + sl] = __float_as_uint(k.off1);
              pool[4 * POOL + sl] = __float_as_uint(k.off2);
              pool[5 * POOL + sl] = __float_as_uint(k.gmax);
            }
          }
          if (gl == 0) *gbest = st.y;
        } else {
          exhausted = true;
        }
      }
    }
    // A drained pool takes back what had to be parked in HBM (the newest first, up to half a pool).
    if (SPILL && have && count == 0u && spill_n != 0u) {
      const uint32_t m = spill_n < (uint32_t)(POOL / 2) ? spill_n : (uint32_t)(POOL / 2);
      for (uint32_t i = gl; i < m; i += G) {
        const Task k = spill_g[spill_n - m + i];
        pool[0 * POOL + i] = k.ref;
        pool[1 * POOL + i] = __float_as_uint(k.nbd);
        pool[2 * POOL + i] = __float_as_uint(k.off0);
        pool[3 * POOL + i] = __float_as_uint(k.off1);
        pool[4 * POOL + i] = __float_as_uint(k.off2);
        pool[5 * POOL + i] = __float_as_uint(k.gmax);
      }
      count = m;
      spill_n -= m;
    }
    if (__ballot(have) == 0ull) break;

    const float best = __uint_as_float(*gbest);
    const float bm = f_add(best, f_mul(best, 0.0009765625f));  // best * (1 + 2^-10), see above

    // Idle lanes take subtrees off the top of the pool.
    const bool want = have && !busy;
    const uint64_t wmask = __ballot(want) & gmask;
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
    if (busy) {
      // Every lane's first 16 bytes come from ONE load instruction, whatever it holds -- a pending
      // record (the parent branch), a branch, or a leaf (its first point; the other three follow at
      // once) -- so that a step waits for memory once, not once per kind of node.
      const bool is_leaf = !fresh && (ref & kLeafBit) != 0u;
      const uint32_t lv = ref & 0x7FFFFFFFu;
      const uint32_t begin = lv >> t.cbits;
      const uint32_t cnt = lv & t.cmask;
      const uint4* src = is_leaf ? reinterpret_cast<const uint4*>(pts + begin)
                                 : nodes + (fresh ? (ref & kRecIdxMask) : (ref & kBranchIdxMask));
      const uint4 w0 = *src;
      float4 p[4];
      p[0] = make_float4(__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w));
      if (is_leaf) {
#pragma unroll
        for (int u = 1; u < 4; ++u) p[u] = pts[begin + u];
      }
      if (!is_leaf) {
        // A branch, or (fresh) the parent branch of a pending record whose far child is entered as traverse()
        // enters it: the same arithmetic with the side given instead of chosen, and nothing kept.
        const uint32_t axis = fresh ? (ref >> 28) & 3u : (ref >> 29) & 3u;
        const float left_max = __uint_as_float(w0.x);
        const float right_min = __uint_as_float(w0.y);
        const float v = sel3(axis, qx, qy, qz);
        const bool near_left = f_sub(f_sub(f_add(left_max, right_min), v), v) > 0.0f;
        const bool go_left = fresh ? (ref & kRecSide) != 0u : near_left;  // near side left = far child right (the record's side bit)
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
          }
          ref = go_left ? w0.z : w0.w;
        }
      } else {
        // The nearest of the (up to) four points, the first of them on a tie (a leaf is visited in index
        // order), without a branch per point; then one comparison with what this lane holds.
        float d = 3.402823466e+38f, d_second = 3.402823466e+38f;  // the batch's nearest and its runner-up
        uint32_t du = 0;
#pragma unroll
        for (int u = 3; u >= 0; --u) {
          float dx = f_sub(qx, p[u].x);
          float dy = f_sub(qy, p[u].y);
          float dz = f_sub(qz, p[u].z);
          PTK_SCALAR(dx);
          PTK_SCALAR(dy);
          PTK_SCALAR(dz);
          const float du_d = (uint32_t)u < cnt ? point_distance3<M>(dx, dy, dz) : 3.402823466e+38f;
          const bool take = (uint32_t)u < cnt && du_d <= d;
          d_second = take ? d : (du_d < d_second ? du_d : d_second);
          d = take ? du_d : d;
          du = take ? (uint32_t)u : du;
        }
        // (d stays FLT_MAX only for an empty batch, which does not occur: cnt >= 1.)
        if (d < cd) {
          cd2 = cd < d_second ? cd : d_second;  // the old best and the rest of the batch are "other points" now
          cd = d;
          cpos = begin + du;
          cok = gmax <= d;
          cg = gmax;
        } else {
          cd2 = d < cd2 ? d : cd2;
          if (d == cd && d <= best) {  // an exact tie that can still matter: the one the reference visits first
            if (tie_budget == 0u) {
              cok = false;
              cg = 3.402823466e+38f;
            } else {
              --tie_budget;
              if (!dfs_before(t, ranges, qx, qy, qz, cpos, begin + du)) {
                cpos = begin + du;
                cok = gmax <= d;
                cg = gmax;
              }
            }
          }
        }
        if (cnt > 4u) {
          ref = kLeafBit | ((begin + 4u) << t.cbits) | (cnt - 4u);
        } else {
          busy = false;
        }
        if (cd < best) lds_min_u32(gbest, __float_as_uint(cd));
      }
    }

    // Far children kept in this step go onto the pool.
    const uint64_t pmask = __ballot(push) & gmask;
    if (push) {
      const uint32_t sl = count + (uint32_t)__popcll(pmask & below);
      if (sl < (uint32_t)POOL) {
        pool[0 * POOL + sl] = p_ref;
        pool[1 * POOL + sl] = __float_as_uint(p_nbd);
        pool[2 * POOL + sl] = __float_as_uint(p_off0);
        pool[3 * POOL + sl] = __float_as_uint(p_off1);
        pool[4 * POOL + sl] = __float_as_uint(p_off2);
        pool[5 * POOL + sl] = __float_as_uint(p_gmax);
      } else if (SPILL && spill_n + (sl - (uint32_t)POOL) < spill_cap) {  // no room in LDS: parked in HBM
        Task k;
        k.ref = p_ref;
        k.nbd = p_nbd;
        k.off0 = p_off0;
        k.off1 = p_off1;
        k.off2 = p_off2;
        k.gmax = p_gmax;
        spill_g[spill_n + (sl - (uint32_t)POOL)] = k;
      }
    }
    count += (uint32_t)__popcll(pmask);
    if (count > (uint32_t)POOL) {
      if (SPILL) {
        spill_n += count - (uint32_t)POOL;
        count = (uint32_t)POOL;
      }
      if (!SPILL || spill_n > spill_cap) {  // a subtree was lost: this query cannot be certified here
        failed = true;
        count = 0;
        spill_n = 0;
        busy = false;
      }
    }

    // A query is done when its pool is empty and no lane of the group holds a subtree.
    const uint64_t bmask = __ballot(busy) & gmask;
    const bool done = have && count == 0u && (!SPILL || spill_n == 0u) && bmask == 0ull;
    if (__ballot(done) != 0ull) {
      const float dstar = __uint_as_float(*gbest);
      // (A minimum of exactly 0 is safe -- every term of such a distance is 0, so is every box
      // distance above the point, and nothing is pruned at 0 -- but not one in the denormal range.)
      if (dstar != 0.0f && dstar < 1e-30f) failed = true;
      // Lanes holding a point at the minimum; several: keep the one the reference visits first.
      bool mine = done && !failed && cd == dstar;
      for (;;) {
        const uint64_t mm = __ballot(mine) & gmask;
        const bool several = done && __popcll(mm) > 1;
        if (__ballot(several) == 0ull) break;
        const int la = several ? (int)__builtin_ctzll(mm) : (int)lane;
        const int lb = several ? (int)__builtin_ctzll(mm & (mm - 1ull)) : (int)lane;
        const uint32_t pos_b = (uint32_t)__shfl((int)cpos, lb);
        int a_first = 1;
        if (several && (int)lane == la) a_first = dfs_before(t, ranges, qx, qy, qz, cpos, pos_b) ? 1 : 0;
        a_first = __shfl(a_first, la);
        if (several && (int)lane == (a_first ? lb : la)) mine = false;
      }
      // The certificate, widened by one case.  `cok` says every far child on the way to the winner has a box distance
      // <= d*.  On data with coordinates on a grid the nearest point often lies ON the planes that bound its subtrees
      // and the incrementally updated box distance ends one rounding ABOVE the point's distance: cok fails, though the
      // reference does reach the point -- it enters a far child when its best so far is >= the box distance, and its
      // best so far is the distance of some OTHER point.  So: if every other point is at least as far as the largest
      // box distance g on the winner's path, the reference enters every far child on that path, accepts the winner
      // (its best then is >= g > d*, strictly) and keeps it.  "Every other point" = the nearest of what the lanes have
      // measured except the winner (their bests, the winner's runner-up) and the best that was handed over; what was
      // pruned is farther than d* (1 + 2^-11) (see above), hence g <= d* (1 + 2^-12) is required as well.
      if (done && gl == 0) *gsecond = __float_as_uint(start_d);
      const uint64_t mm = __ballot(mine) & gmask;
      if (done && !failed) lds_min_u32(gsecond, __float_as_uint(mine ? cd2 : cd));
      const uint64_t strict = __ballot(mine && cok) & gmask;
      const float d_other = __uint_as_float(*gsecond);
      const bool wide = mine && !cok && cg <= d_other && cg <= f_add(dstar, f_mul(dstar, 0.000244140625f));
      const uint64_t good = strict | (__ballot(wide) & gmask);
      if (done) {
        if (failed) {
          if (gl == 0) redo_list[atomicAdd(&cont.meta[kMetaRedo], 1u)] = e;
        } else if (!(dstar < start_d)) {  // nothing closer than what the reference had already found
          if (gl == 0) {
            Neighbor nb;
            nb.index = (int32_t)start_i;
            nb.distance = start_d;
            out[qi] = nb;
          }
        } else if (__popcll(mm) == 1 && good == mm) {
          if (mine) {
            Neighbor nb;
            nb.index = __float_as_int(pts[cpos].w);
            nb.distance = cd;
            out[qi] = nb;
          }
        } else if (gl == 0) {
          redo_list[atomicAdd(&cont.meta[kMetaRedo], 1u)] = e;
        }
        have = false;
      }
    }
  }
  PTK_TRACE_END_ANY();
}

// The reference search from the root for the queries the cooperative search listed.
template <int S, int OVF, int LEAFB, class M = MetricL2>
__global__ __launch_bounds__(64) void knn1_redo_kernel(
    DevTree t, const float4* __restrict__ qs, float e_inv, Neighbor* __restrict__ out, Cont cont,
    const uint32_t* __restrict__ redo_list) {
  const uint32_t n = cont.meta[kMetaRedo];
  Record spill[OVF > 0 ? OVF : 1];
  for (uint32_t i = blockIdx.x * 64u + threadIdx.x; i < n; i += gridDim.x * 64u) {
    const float4 qrec = qs[redo_list[i]];
    const uint32_t qi = __float_as_uint(qrec.w);
    Stack<S, OVF, 64> st;
    st.init((LdsWord*)ptk_smem, threadIdx.x, spill);
    NnPolicy pol;
    pol.e_inv = e_inv;
    pol.out = out;
    pol.begin_query(qi);
    traverse<LEAFB, false, M>(t, qrec.x, qrec.y, qrec.z, pol, st);
    pol.end_query(qi);
  }
}

// ---- box search ---------------------------------------------------------------------------
// search_box (/root/reference/src/pico_tree/pico_tree/internal/kd_tree_search.hpp:238-381): a
// left-then-right depth-first walk with a RUNNING node box; a child whose box lies inside the
// query box is reported wholesale (its whole index range, :321-337), a child that merely
// intersects is entered, a leaf is filtered point by point (closed interval test, box.hpp:31-42).
// One query box per lane.  Records on the lane's LIFO:
//   right  {node | axis << 28, val = max[axis] to restore}  the right child of `node` is still to do
//   undo   {kRecUndo | axis << 28, val = min[axis] to restore}
// `ranges[branch] = {begin, end}` is the index range of the whole subtree (what report_left /
// report_right find by walking to the outermost leaves, :339-353).  Unused axes (dim < 3) carry
// the box [0, 0] against a query of (-inf, +inf).  COUNT pass: counts[i]; FILL pass: indices.
struct BoxState {
  float mn0, mn1, mn2, mx0, mx1, mx2;
};
__device__ __forceinline__ void set_axis(float& a0, float& a1, float& a2, uint32_t axis, float v) {
  a0 = axis == 0 ? v : a0;
  a1 = axis == 1 ? v : a1;
  a2 = axis == 2 ? v : a2;
}

// TOPO: the tree of a topological metric (metric_so2: axis 0 a circle; metric_se2_squared: axis 2), searched as the
// reference's search_box does with a metric_box_map query (kd_tree_search.hpp:238-381, box.hpp:300-376, segment.hpp):
// on a circle axis (bit of s1_mask) a query interval with min > max wraps around the seam -- it contains x iff
// x >= min || x <= max and a node interval iff node.min >= min || node.max <= max; and EVERY axis of such a tree uses
// the four-bound intersection tests (`query.min <= left_max || query.max >= left_min`, :311-327), for which the two
// outer bounds of the branch come from t.outer.
template <int S, int OVF, bool FILL, bool TOPO = false>
__global__ __launch_bounds__(64) void box_kernel(
    DevTree t, const uint2* __restrict__ ranges, BoxState root, const float* __restrict__ mins,
    const float* __restrict__ maxs, uint32_t dim, uint64_t nb, uint64_t* __restrict__ counts,
    const uint64_t* __restrict__ offsets, int32_t* __restrict__ out, const uint32_t* __restrict__ perm = nullptr,
    uint32_t s1_mask = 0) {
  const uint64_t li = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (li >= nb) return;
  const uint64_t bi = perm ? perm[li] : li;  // launch order only (boxes sorted by their min corner)
  const float inf = __uint_as_float(0x7F800000u);
  float qn0, qn1, qn2, qx0, qx1, qx2;
  load_query(mins, dim, bi, qn0, qn1, qn2);
  load_query(maxs, dim, bi, qx0, qx1, qx2);
  if (dim < 2) { qn1 = -inf; qx1 = inf; }
  if (dim < 3) { qn2 = -inf; qx2 = inf; }
  BoxState b = root;
  const uint4* __restrict__ nodes = t.nodes;
  const float4* __restrict__ pts = t.pts;
  uint64_t count = 0;
  int32_t* row = FILL ? out + offsets[bi] : nullptr;

  PTK_STACK(S, OVF, 64, st, t);

  // One axis of metric_box_map::contains (segment_r1 / segment_s1::contains of an interval and of a coordinate).
  auto seg_in = [&](uint32_t axis, float qn, float qx, float mn, float mx) {
    const bool wrap = ((s1_mask >> axis) & 1u) != 0u && !(qn <= qx);
    return wrap ? (mn >= qn || mx <= qx) : (qn <= mn && mx <= qx);
  };
  auto pt_in = [&](uint32_t axis, float qn, float qx, float x) {
    const bool wrap = ((s1_mask >> axis) & 1u) != 0u && !(qn <= qx);
    return wrap ? (x >= qn || x <= qx) : (qn <= x && x <= qx);
  };
  auto inside = [&]() {  // query_.contains(box_): both corners inside the closed query box
    if (TOPO)
      return seg_in(0, qn0, qx0, b.mn0, b.mx0) && seg_in(1, qn1, qx1, b.mn1, b.mx1) && seg_in(2, qn2, qx2, b.mn2, b.mx2);
    return qn0 <= b.mn0 && b.mn0 <= qx0 && qn0 <= b.mx0 && b.mx0 <= qx0 &&
           qn1 <= b.mn1 && b.mn1 <= qx1 && qn1 <= b.mx1 && b.mx1 <= qx1 &&
           qn2 <= b.mn2 && b.mn2 <= qx2 && qn2 <= b.mx2 && b.mx2 <= qx2;
  };
  auto report_range = [&](uint32_t begin, uint32_t end) {
    if (FILL) {
      for (uint32_t p = begin; p < end; ++p) row[count + (p - begin)] = __float_as_int(pts[p].w);
    }
    count += end - begin;
  };
  auto report = [&](uint32_t ref) {  // report_node: the whole subtree
    if (ref & kLeafBit) {
      const uint32_t lv = ref & 0x7FFFFFFFu;
      report_range(lv >> t.cbits, (lv >> t.cbits) + (lv & t.cmask));
    } else {
      const uint2 r = ranges[ref & kBranchIdxMask];
      report_range(r.x, r.y);
    }
  };
  auto scan_leaf = [&](uint32_t ref) {
    const uint32_t lv = ref & 0x7FFFFFFFu;
    const uint32_t begin = lv >> t.cbits;
    const uint32_t n = lv & t.cmask;
    for (uint32_t j = 0; j < n; ++j) {
      const float4 p = pts[begin + j];
      PTK_KEEP4(p);  // one 16-byte load: the index must not become a dependent load inside the branch
      const bool hit = TOPO ? (pt_in(0, qn0, qx0, p.x) && pt_in(1, qn1, qx1, p.y) && pt_in(2, qn2, qx2, p.z))
                            : (qn0 <= p.x && p.x <= qx0 && qn1 <= p.y && p.y <= qx1 && qn2 <= p.z && p.z <= qx2);
      if (hit) {
        if (FILL) row[count] = __float_as_int(p.w);
        ++count;
      }
    }
  };

  uint32_t ref = t.root_ref;  // the subtree to walk next (kLeafBit | 0x7FFFFFFF... is never produced)
  bool have = true;
  for (;;) {
    if (have) {
      if (ref & kLeafBit) {
        scan_leaf(ref);
        have = false;
      } else {
        const uint32_t idx = ref & kBranchIdxMask;
        const uint32_t axis = (ref >> 29) & 3u;
        const uint4 nd = nodes[idx];
        const float left_max = __uint_as_float(nd.x);
        st.push(idx | (axis << 28), sel3(axis, b.mx0, b.mx1, b.mx2));  // the right child comes later
        set_axis(b.mx0, b.mx1, b.mx2, axis, left_max);
        bool enter_left = sel3(axis, qn0, qn1, qn2) <= left_max;  // intersects_left
        if (TOPO) enter_left = enter_left || sel3(axis, qx0, qx1, qx2) >= t.outer[idx].x;  // || query.max >= left_min
        if (inside()) {
          report(nd.z);
          have = false;
        } else if (enter_left) {
          ref = nd.z;
        } else {
          have = false;
        }
      }
      continue;
    }
    if (st.empty()) break;
    const Record r = st.pop();
    const uint32_t axis = (r.x >> 28) & 3u;
    const float val = __uint_as_float(r.y);
    if (r.x & kRecUndo) {
      set_axis(b.mn0, b.mn1, b.mn2, axis, val);
      continue;
    }
    // Left side of node r.x is done: restore max, narrow min, do the right side.
    set_axis(b.mx0, b.mx1, b.mx2, axis, val);
    const uint4 nd = nodes[r.x & kRecIdxMask];
    const float right_min = __uint_as_float(nd.y);
    st.push(kRecUndo | (axis << 28), sel3(axis, b.mn0, b.mn1, b.mn2));
    set_axis(b.mn0, b.mn1, b.mn2, axis, right_min);
    bool enter_right = sel3(axis, qx0, qx1, qx2) >= right_min;  // intersects_right
    if (TOPO) enter_right = enter_right || sel3(axis, qn0, qn1, qn2) <= t.outer[r.x & kRecIdxMask].y;  // || query.min <= right_max
    if (inside()) {
      report(nd.w);
    } else if (enter_right) {
      ref = nd.w;
      have = true;
    }
  }
  if (!FILL) counts[bi] = count;
}

// Sorts every row ascending by distance (heap sort, in place, one row per lane).
// std::sort in the reference is unstable, so the order among equal distances is
// unspecified on both sides.
PTK_GLOBAL __launch_bounds__(kBlock) void sort_rows_kernel(
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

// Point records of the device tree: pts[pos] = {point indices[pos], bits(indices[pos])} in leaf
// order, the kLeafPad records behind the last point repeat it (see ptk_encode.hpp).
PTK_GLOBAL __launch_bounds__(kBlock) void encode_points_kernel(
    const float* __restrict__ points, uint32_t dim, const int32_t* __restrict__ indices, uint64_t n,
    float4* __restrict__ pts) {
  const uint64_t pos = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (pos >= n + kLeafPad) return;
  const int32_t idx = indices[pos < n ? pos : n - 1];
  float x, y, z;
  load_query(points, dim, (uint64_t)idx, x, y, z);
  pts[pos] = make_float4(x, y, z, __int_as_float(idx));
}

// ---- batch ordering ------------------------------------------------------------------------
// Morton key of each query inside the tree's root box (clamped), plus the identity permutation
// to be sorted along with it.  spread10 serves the float64 kernel (ten bits per axis).
// ---- piles (ptk_piles.hpp): the rows of a k = 1 search on the view of a tree with piles ------------------------
// A row whose index is the stand-in of a pile gets the index the reference reports: the pile's first-visited point
// for the side of the query on each axis (the side test of traverse<> with both bounds at the pile's coordinate).
struct DevPileRecord {  // == ptk::PileRecord
  float c[3];
  uint32_t count;
  int32_t first[8];
};
struct DevPiles {
  const uint32_t* of_point;  // [n_points] 1 + pile of the point that stands for it, else 0
  const DevPileRecord* recs;
  uint32_t n_points;
};
PTK_GLOBAL __launch_bounds__(kBlock) void resolve_piles_kernel(const float* __restrict__ queries, uint32_t dim, uint64_t nq,
                                                                DevPiles piles, Neighbor* __restrict__ rows) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nq) return;
  const Neighbor nb = rows[i];
  if ((uint32_t)nb.index >= piles.n_points) return;
  const uint32_t p = piles.of_point[nb.index];
  if (p == 0u) return;
  const DevPileRecord& rec = piles.recs[p - 1u];
  float q[3];
  load_query(queries, dim, i, q[0], q[1], q[2]);
  uint32_t bits = 0u;
#pragma unroll
  for (uint32_t a = 0; a < 3; ++a) {
    const float s = f_sub(f_sub(f_add(rec.c[a], rec.c[a]), q[a]), q[a]);
    if (a < dim && s > 0.0f) bits |= 1u << a;
  }
  rows[i].index = rec.first[bits];
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {
  x &= 0x3FFu;
  x = (x | (x << 16)) & 0x030000FFu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}

// The float32 key has an uneven number of bits per axis (bx + by + bz <= 30; the backend derives them from the
// tree: how often a root-to-leaf path splits on each axis).  A cloud that is flat along one axis -- a LiDAR
// scan is mostly floor -- then spends its key bits where its leaves actually divide space.  Bits are
// interleaved from the most significant level down; an axis joins in at the level its own bits begin.
__device__ __forceinline__ uint32_t morton_key(float x, float y, float z, float3 lo, float3 inv, uint3 bits) {
  const uint32_t cx = (uint32_t)fminf(fmaxf((x - lo.x) * inv.x, 0.0f), (float)((1u << bits.x) - 1u));
  const uint32_t cy = (uint32_t)fminf(fmaxf((y - lo.y) * inv.y, 0.0f), (float)((1u << bits.y) - 1u));
  const uint32_t cz = (uint32_t)fminf(fmaxf((z - lo.z) * inv.z, 0.0f), (float)((1u << bits.z) - 1u));
  const uint32_t top = bits.x > bits.y ? (bits.x > bits.z ? bits.x : bits.z) : (bits.y > bits.z ? bits.y : bits.z);
  uint32_t key = 0;
  for (uint32_t level = top; level-- > 0;) {
    if (bits.z > level) key = (key << 1) | ((cz >> level) & 1u);
    if (bits.y > level) key = (key << 1) | ((cy >> level) & 1u);
    if (bits.x > level) key = (key << 1) | ((cx >> level) & 1u);
  }
  return key;
}

// Which queries will be expensive?  A kd-tree search is long where the query sits in EMPTY space: its bound stays
// loose and it walks the ring of points around the hole (the scanner's blind disc: 600 leaves at knn = 16 where the
// mean is 12).  The tree's handle keeps one byte per cell of a coarse Morton grid over the root box (about 32 tree
// points per cell on average: 2^18 cells for BASELINE config 2): 1 = the cell holds a tree point.  On that cloud the
// queries in empty cells are 3.2 % of the batch and include EVERY query with more than 99 leaf visits (knn = 16;
// more than 42 at knn = 1), profiles/r03_notes.txt item 8.  The kernels that run every query to its end in its lane
// sort them to the FRONT of the launch: their long dependent chains then start at once and run beside the bulk
// of the batch instead of behind it.
struct CellTable {
  const uint8_t* occ = nullptr;  // [1 << (bits.x + bits.y + bits.z)]: 0 = no tree point in the cell, else
                                 // 1 + floor(log2(points in it)); null = no table (the key is the plain Morton key)
  float3 inv = make_float3(0.0f, 0.0f, 0.0f);
  uint3 bits = make_uint3(0u, 0u, 0u);
  uint32_t key_bits = 0;         // width of the order key
  uint32_t mode = 0;             // kCellsEmptyFirst / kCellsDenseFirst: what the top bit(s) of the key say
};
// kCellsEmptyFirst  (k nearest neighbours): top bit = "the query's cell holds tree points": the searches that will be
//                   long start first.
// kCellsDenseFirst  (radius search): the cost of a query is its number of hits, i.e. the density around it.  A batch of
//                   config 3 ends with the queries on the floor under the scanner (thousands of hits each; 225 k
//                   queries take 5.5 ms, 7.2 M 15.4: a tail of ~4.5 ms whatever the size, profiles/r03_notes.txt item
//                   9): two top bits = density class, the densest cells first.
constexpr uint32_t kCellsEmptyFirst = 1, kCellsDenseFirst = 2;
__device__ __forceinline__ uint32_t order_key(float x, float y, float z, float3 lo, float3 inv, uint3 bits,
                                              const CellTable& cells) {
  const uint32_t key = morton_key(x, y, z, lo, inv, bits);
  if (cells.occ == nullptr) return key;
  const uint32_t v = cells.occ[morton_key(x, y, z, lo, cells.inv, cells.bits)];
  if (cells.mode == kCellsEmptyFirst) {
    const uint32_t cheap = v != 0u ? 1u : 0u;
    return (cheap << (cells.key_bits - 1u)) | (key >> 1);  // (the lowest Morton bit makes room)
  }
  // >= 2048 / 512 / 128 points in the cell (the average is ~32), the rest
  const uint32_t cls = v >= 12u ? 0u : (v >= 10u ? 1u : (v >= 8u ? 2u : 3u));
  return (cls << (cells.key_bits - 2u)) | (key >> 2);
}

// Points per cell of the coarse grid (at creation), then its class byte.  The points come in leaf order, so the
// lanes of a wavefront hold runs of the same cell: one atomic per run (its first lane adds the run's length) instead of
// one per point (7.7 M atomics on 262 k words took 4.6 ms, the dense cells serialising).
PTK_GLOBAL __launch_bounds__(kBlock) void cell_count_kernel(const float4* __restrict__ pts, uint64_t n, float3 lo, float3 inv,
                                                            uint3 bits, uint32_t* __restrict__ counts) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < n;
  const float4 p = pts[valid ? i : n - 1];
  const uint32_t cell = valid ? morton_key(p.x, p.y, p.z, lo, inv, bits) : 0xFFFFFFFFu;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t prev = (uint32_t)__shfl_up((int)cell, 1);
  const bool head = lane == 0u || prev != cell;
  const uint64_t heads = __ballot(head);
  if (head && valid) {
    const uint64_t above = lane == 63u ? 0ull : heads >> (lane + 1u);   // the next run's head, if any
    const uint32_t len = above != 0ull ? (uint32_t)__builtin_ctzll(above) + 1u : 64u - lane;
    // (an invalid tail of the last wavefront is a run of its own: `len` of the last valid run stops before it)
    atomicAdd(&counts[cell], len);
  }
}
PTK_GLOBAL __launch_bounds__(kBlock) void cell_class_kernel(const uint32_t* __restrict__ counts, uint64_t n_cells,
                                                            uint8_t* __restrict__ occ) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_cells) return;
  const uint32_t c = counts[i];
  uint32_t v = 0;
  while ((c >> v) != 0u) ++v;  // 0 for an empty cell, else 1 + floor(log2(c))
  occ[i] = (uint8_t)v;
}

PTK_GLOBAL __launch_bounds__(kBlock) void morton_kernel(
    const float* __restrict__ queries, uint32_t dim, uint64_t nq, float3 lo, float3 inv, uint3 bits,
    uint32_t* __restrict__ keys, uint32_t* __restrict__ ids, CellTable cells = CellTable{}, uint64_t begin = 0) {
  const uint64_t i = begin + (uint64_t)blockIdx.x * kBlock + threadIdx.x;  // rows [begin, nq) of the batch
  if (i >= nq) return;
  float x, y, z;
  load_query(queries, dim, i, x, y, z);
  keys[i] = order_key(x, y, z, lo, inv, bits, cells);
  ids[i] = (uint32_t)i;
}

// Is the batch already in a coherent order?  The reference walks the rows in the caller's order
// (_pyco_tree/kd_tree.hpp:128-134) and real scans arrive in scan order: sorting such a batch again buys nothing.
// kCoherenceWindows windows of 64 consecutive rows, evenly spread over the batch, one wavefront each:
// fail[w] = 1 if the rows of window w spread over more than 2^max_log2 cells of the order-key grid (the product of
// the per-axis extents of their bounding box, each rounded up to a power of two) -- 64 neighbours of a sorted batch of
// nq rows cover about 2^(key bits) x 64 / nq cells.
// The verdict stays ON THE DEVICE (the entry points that take device buffers only enqueue: ptk.h): every window adds
// {1 << 16 | its failure} to state[0]; the window that finds the other kCoherenceWindows - 1 counted writes
// state[1] = 1 if at most 15 % of the windows failed (a window may straddle a cell boundary), else 0, and puts
// state[0] back to 0 for the next batch.  The kernels of the sort and phase 1 of the search read state[1]: a coherent
// batch leaves the sort's kernels at their first instruction and is searched in the caller's order.
constexpr uint32_t kCoherenceWindows = 256;
PTK_GLOBAL __launch_bounds__(64) void coherence_sample_kernel(const float* __restrict__ queries, uint32_t dim, uint64_t nq,
                                                              float3 lo, float3 inv, uint3 bits, uint32_t max_log2,
                                                              uint8_t* __restrict__ fail, uint32_t* __restrict__ state = nullptr) {
  const uint32_t w = blockIdx.x, lane = threadIdx.x;
  const uint64_t start = gridDim.x > 1u ? (nq - 64u) * w / (gridDim.x - 1u) : 0u;  // (nq >= 64)
  float x, y, z;
  load_query(queries, dim, start + lane, x, y, z);
  const uint32_t cx = (uint32_t)fminf(fmaxf((x - lo.x) * inv.x, 0.0f), (float)((1u << bits.x) - 1u));
  const uint32_t cy = (uint32_t)fminf(fmaxf((y - lo.y) * inv.y, 0.0f), (float)((1u << bits.y) - 1u));
  const uint32_t cz = (uint32_t)fminf(fmaxf((z - lo.z) * inv.z, 0.0f), (float)((1u << bits.z) - 1u));
  uint32_t mn[3] = {cx, cy, cz}, mx[3] = {cx, cy, cz};
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const uint32_t lo_o = (uint32_t)__shfl_xor((int)mn[a], d), hi_o = (uint32_t)__shfl_xor((int)mx[a], d);
      mn[a] = lo_o < mn[a] ? lo_o : mn[a];
      mx[a] = hi_o > mx[a] ? hi_o : mx[a];
    }
  }
  if (lane == 0u) {
    uint32_t log2_cells = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const uint32_t ext = mx[a] - mn[a];  // cells spanned - 1
      log2_cells += ext == 0u ? 0u : 32u - (uint32_t)__builtin_clz(ext);
    }
    const uint32_t failed = log2_cells > max_log2 ? 1u : 0u;
    fail[w] = (uint8_t)failed;
    if (state != nullptr) {
      const uint32_t before = atomicAdd(&state[0], (1u << 16) | failed);
      if ((before >> 16) == gridDim.x - 1u) {  // the last window in
        const uint32_t failing = (before & 0xFFFFu) + failed;
        state[1] = failing * 100u <= gridDim.x * 15u ? 1u : 0u;
        state[0] = 0u;
      }
    }
  }
}

// Inclusive-to-exclusive helper for the radius offsets: offsets[0] = 0 is written
// by the host-side scan call; see ptk_backend.hip.

}  // namespace ptk
