// ptk_forest.hpp -- gfx950 device code of the randomised kd-forest search (BASELINE config 5).
//
// What it computes (reference: /root/reference/examples/pico_understory/pico_understory/
// kd_forest.hpp:70-115, internal/kd_tree_priority_search.hpp:24-142,
// internal/rkd_tree_hh_data.hpp:38-93): T trees, each built over a Householder-reflected copy of
// the points (y = x - 2 (r.x) r); a query visits the trees in order, reflects itself the same way,
// and runs a best-bin-first search per tree -- a min-queue of (box distance, node), each pop
// descends near-first to one leaf, measures it and, ON THE WAY BACK UP (priority_search:118-128,
// i.e. with the visitor already updated by that leaf), queues the far children that are still
// closer than visitor.max() -- until `max_leaves_visited` leaves were measured in this tree or the
// closest queued node is farther than visitor.max().  One visitor is shared by all trees.
//
// Two deliberate differences from the reference, both prescribed by SURVEY.md 8(a) row A13:
//   * the k-list is DEDUPLICATED by point index (the reference's plain search_knn visitor receives
//     the same point once per tree and fills the list with copies of it);
//   * point distances are measured in the ORIGINAL space (a reflection is an isometry, so this is
//     the same quantity without the per-tree rounding noise) -- which also means only ONE copy of
//     the points is kept in HBM (512 MB for SIFT-1M) instead of one rotated copy per tree (4.1 GB):
//     the trees keep their split bounds in reflected coordinates plus an index permutation.
// Ties between queued nodes of equal distance are broken by the node reference (branches before
// leaves, then depth-first order; the reference compares node addresses there, which is unspecified).
// The reference draws its reflection vectors from std::random_device; here they are given by the
// caller (libptk derives them from a seed), so a forest is reproducible.
//
// Mapping: ONE QUERY PER WAVEFRONT.  10 000 queries of 128 dimensions cannot feed one query per
// lane (157 waves, a 10 KB queue per lane); the work is the leaf scans (32 points x 128 floats,
// 512 leaves per query), so the 64 lanes measure one leaf together -- rows of 128 floats are read
// by the whole wave, one coalesced 512-byte read per row, 16 rows in flight, and summed with a
// transposing butterfly; other dimensions: lane j streams the row of point j -- while the
// descent, the queue and the k-list are wave-uniform:
//   * node records come through the scalar cache (uniform index -> s_load_dwordx8);
//   * the queue lives in LDS; extract-min is a strided scan + a 6-step wave reduction;
//   * the sorted k-list lives in REGISTERS, entry j in lane j (k <= 64): a candidate's rank is a
//     ballot popcount, the shift is one __shfl_up, the duplicate test one ballot.
// No MFMA: a leaf scan is a 32 x 128 matrix-vector product read once -- HBM-bound.

#pragma once

#include "ptk_forest_host.hpp"
#include "ptk_kernels.hpp"

namespace ptk {

struct ForestTreeDev {
  const ForestNode* nodes;
  const int32_t* indices;   // leaf-ordered permutation
  const float* rotation;    // dim floats, unit length
  uint32_t root_ref;
  uint32_t cbits;
  uint32_t cmask;
  uint32_t pad;
};

struct ForestDev {
  const ForestTreeDev* trees;
  const float* points;      // n x dim, ORIGINAL order, row-major
  uint32_t n_trees;
  uint32_t dim;
};

constexpr float kFltMax = 3.402823466e+38f;

__device__ __forceinline__ float wave_shfl(float v, int lane) { return __shfl(v, lane); }
// Value of `v` in lane `src`, for a wave-uniform `src`: one v_readlane instead of a trip through the
// LDS crossbar (ds_bpermute).
__device__ __forceinline__ int32_t lane_value(int32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float lane_value(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// Order of two queued nodes at EQUAL distance: branches before leaves, then depth-first stream
// order (leaf references grow with it; branch records carry their stream position, fetched only
// in this rare case).
__device__ __forceinline__ bool queued_before(const ForestNode* nodes, uint32_t a, uint32_t b) {
  if ((a | b) & kLeafBit) return a < b;
  return a != b && nodes[a].stream_id < nodes[b].stream_id;
}

// LDS of a block: q[dim] | qr[dim] | queue_d[Q] | queue_ref[Q] | path_d[P] | path_ref[P]
template <int KMAX>
__global__ __launch_bounds__(64) void forest_knn_kernel(
    ForestDev f, const float* __restrict__ queries, uint64_t nq, uint32_t k, uint32_t max_leaves,
    Neighbor* __restrict__ out, uint32_t* __restrict__ dropped) {
  const uint64_t qi = blockIdx.x;
  if (qi >= nq) return;
  const uint32_t lane = threadIdx.x;
  const uint64_t lanes_below = (1ull << lane) - 1ull;
  const uint32_t dim = f.dim;
  PTK_LDS float* q = (PTK_LDS float*)ptk_smem;
  PTK_LDS float* qr = q + dim;
  PTK_LDS float* queue_d = qr + dim;
  PTK_LDS uint32_t* queue_ref = (PTK_LDS uint32_t*)(queue_d + kForestQueue);
  PTK_LDS float* path_d = (PTK_LDS float*)(queue_ref + kForestQueue);
  PTK_LDS uint32_t* path_ref = (PTK_LDS uint32_t*)(path_d + kForestPath);

  for (uint32_t a = lane; a < dim; a += 64) q[a] = queries[qi * dim + a];
  __syncthreads();

  // The k-list: entry j lives in lane j.
  float ld = kFltMax;
  int32_t li = -1;
  uint32_t filled = 0;
  float worst = kFltMax;  // visitor.max(): FLT_MAX until the list is full

  for (uint32_t ti = 0; ti < f.n_trees; ++ti) {
    const ForestTreeDev t = f.trees[ti];
    // Reflect the query (rkd_tree_hh_data.hpp:80-90): the dot product is a left-to-right sum.
    float dot = 0.0f;
    for (uint32_t a = 0; a < dim; ++a) dot = f_add(dot, f_mul(t.rotation[a], q[a]));
    dot = f_mul(dot, 2.0f);
    for (uint32_t a = lane; a < dim; a += 64) qr[a] = f_sub(q[a], f_mul(dot, t.rotation[a]));
    __syncthreads();

    uint32_t qn = 1;  // queue size
    if (lane == 0) {
      queue_d[0] = 0.0f;
      queue_ref[0] = t.root_ref;
    }
    __syncthreads();
    uint32_t leaves_visited = 0;

    while (qn > 0) {
      // ---- extract-min over (distance, reference) ----
      float bd = kFltMax;
      uint32_t bref = 0xFFFFFFFFu, bpos = 0xFFFFFFFFu;
      for (uint32_t p = lane; p < qn; p += 64) {
        const float d = queue_d[p];
        const uint32_t r = queue_ref[p];
        if (d < bd || (d == bd && (bpos == 0xFFFFFFFFu || queued_before(t.nodes, r, bref)))) {
          bd = d;
          bref = r;
          bpos = p;
        }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const float od = __shfl_xor(bd, off);
        const uint32_t oref = (uint32_t)__shfl_xor((int)bref, off);
        const uint32_t opos = (uint32_t)__shfl_xor((int)bpos, off);
        if (opos != 0xFFFFFFFFu &&
            (od < bd || (od == bd && (bpos == 0xFFFFFFFFu || queued_before(t.nodes, oref, bref))))) {
          bd = od;
          bref = oref;
          bpos = opos;
        }
      }
      if (leaves_visited >= max_leaves || worst < bd) break;  // priority_search:55-58
      uint32_t ref = bref;
      float nbd = bd;
      __syncthreads();
      --qn;
      if (lane == 0 && bpos != qn) {  // fill the hole with the last entry
        queue_d[bpos] = queue_d[qn];
        queue_ref[bpos] = queue_ref[qn];
      }
      __syncthreads();

      // ---- near-first descent to one leaf, recording the far children ----
      uint32_t depth = 0;
      while (!(ref & kLeafBit)) {
        // One fetch per BLOCK of three levels (ptk_forest_host.hpp): lane s holds slot s, the steps
        // inside the block read their record with v_readlane.
        const uint32_t blk = uniform_value(ref) >> 3;
        const ForestNode mine = t.nodes[blk * 8u + (lane & 7u)];
        while (!(ref & kLeafBit) && (ref >> 3) == blk) {
          const int slot = (int)(ref & 7u);
          ForestNode nd;
          nd.left_min = lane_value(mine.left_min, slot);
          nd.left_max = lane_value(mine.left_max, slot);
          nd.right_min = lane_value(mine.right_min, slot);
          nd.right_max = lane_value(mine.right_max, slot);
          nd.left_ref = (uint32_t)lane_value((int32_t)mine.left_ref, slot);
          nd.right_ref = (uint32_t)lane_value((int32_t)mine.right_ref, slot);
          nd.split_dim = (uint32_t)lane_value((int32_t)mine.split_dim, slot);
          const float v = qr[nd.split_dim];
          float old_off, new_off;
          uint32_t far_ref;
          if (f_sub(f_sub(f_add(nd.left_max, nd.right_min), v), v) > 0.0f) {  // priority_search:97
            far_ref = nd.right_ref;
            const float a = f_sub(nd.left_min, v);
            old_off = v > nd.left_min ? 0.0f : f_mul(a, a);
            const float b = f_sub(nd.right_min, v);
            new_off = f_mul(b, b);
            ref = nd.left_ref;
          } else {
            far_ref = nd.left_ref;
            const float a = f_sub(nd.right_max, v);
            old_off = v < nd.right_max ? 0.0f : f_mul(a, a);
            const float b = f_sub(nd.left_max, v);
            new_off = f_mul(b, b);
            ref = nd.right_ref;
          }
          if (depth < kForestPath && lane == 0) {
            path_d[depth] = f_add(f_sub(nbd, old_off), new_off);  // :123
            path_ref[depth] = far_ref;
          }
          ++depth;
        }
      }

      // ---- the leaf: lane j measures point j, candidates enter the list in leaf order ----
      {
        const uint32_t lv = ref & 0x7FFFFFFFu;
        const uint32_t begin = lv >> t.cbits;
        const uint32_t count = lv & t.cmask;
        if ((dim & 127u) == 0u) {
          // Rows of 128 * M floats: the WAVE reads one row at a time, lane l the two floats
          // 2l, 2l + 1 of every 128-float segment (one coalesced 512-byte read per segment), 16
          // rows in flight.  The 64 partial sums of a row are added in a fixed tree -- lanes
          // pair up by bit 5 of the lane number, then bit 4, ... bit 0 -- which for 16 rows at
          // once is a transposing butterfly: after the steps for bits 5..2 lane l holds the
          // partial of row (l >> 2) & 15 only, so 17 shuffles serve 16 rows.  (The oracle adds in
          // the same tree; see ptk_oracle.cpp, forest_l2sq.)
          const uint32_t segs = dim >> 7;
          for (uint32_t base = 0; base < count; base += 16) {
            const uint32_t rows = count - base < 16u ? count - base : 16u;
            const int32_t my_idx = lane < rows ? t.indices[begin + base + lane] : -1;
            float part[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) part[r] = 0.0f;
            for (uint32_t m = 0; m < segs; ++m) {
              const float qa = q[128u * m + 2u * lane], qb = q[128u * m + 2u * lane + 1u];
              float2 p[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int32_t ir = __shfl(my_idx, r);
                p[r] = ir >= 0 ? *reinterpret_cast<const float2*>(f.points + (uint64_t)ir * dim + 128u * m + 2u * lane)
                               : make_float2(qa, qb);
              }
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const float d0 = f_sub(qa, p[r].x), d1 = f_sub(qb, p[r].y);
                part[r] = f_add(part[r], f_mul(d0, d0));
                part[r] = f_add(part[r], f_mul(d1, d1));
              }
            }
            const bool b5 = (lane & 32u) != 0, b4 = (lane & 16u) != 0, b3 = (lane & 8u) != 0, b2 = (lane & 4u) != 0;
            float v8[8], v4[4], v2[2];
#pragma unroll
            for (int j = 0; j < 8; ++j)
              v8[j] = f_add(b5 ? part[j + 8] : part[j], __shfl_xor(b5 ? part[j] : part[j + 8], 32));
#pragma unroll
            for (int j = 0; j < 4; ++j) v4[j] = f_add(b4 ? v8[j + 4] : v8[j], __shfl_xor(b4 ? v8[j] : v8[j + 4], 16));
#pragma unroll
            for (int j = 0; j < 2; ++j) v2[j] = f_add(b3 ? v4[j + 2] : v4[j], __shfl_xor(b3 ? v4[j] : v4[j + 2], 8));
            float dist = f_add(b2 ? v2[1] : v2[0], __shfl_xor(b2 ? v2[0] : v2[1], 4));
            dist = f_add(dist, __shfl_xor(dist, 2));
            dist = f_add(dist, __shfl_xor(dist, 1));
            // Row r of the chunk is now in lanes 4r .. 4r + 3; candidates enter in leaf order.  Only
            // the rows that still beat max() cost an iteration (the test is re-evaluated after every
            // insertion, as the reference's visitor does point by point).
            bool pending = (lane & 3u) == 0u && (lane >> 2) < rows;
            for (;;) {
              const uint64_t m = __ballot(pending && worst > dist);  // search_visitor.hpp:107
              if (m == 0ull) break;
              const int src = __builtin_ctzll(m);
              const float cd = lane_value(dist, src);
              const int32_t ci = lane_value(my_idx, src >> 2);
              if ((int)lane == src) pending = false;
              if (__ballot(lane < filled && li == ci) != 0ull) continue;  // already listed
              const uint32_t pos = (uint32_t)__popcll(__ballot(lane < filled && !(cd < ld)));
              const float up_d = __shfl_up(ld, 1);
              const int32_t up_i = __shfl_up(li, 1);
              if (filled < k) ++filled;
              if (lane > pos && lane < filled) {
                ld = up_d;
                li = up_i;
              } else if (lane == pos && lane < filled) {
                ld = cd;
                li = ci;
              }
              worst = filled == k ? lane_value(ld, (int)k - 1) : kFltMax;
            }
          }
        } else
        for (uint32_t base = 0; base < count; base += 64) {
          const bool has = base + lane < count;
          int32_t idx = -1;
          float d = kFltMax;
          if (has) {
            idx = t.indices[begin + base + lane];
            const float4* row = reinterpret_cast<const float4*>(f.points + (uint64_t)idx * dim);
            float acc = 0.0f;
            uint32_t a = 0;
            if ((dim & 31u) == 0u) {
              // 8 x 16 bytes in flight per lane (a row is a stream of independent loads; issued one
              // at a time each would pay the full memory latency).  Summation stays left to right.
              for (; a < dim; a += 32) {
                float4 p[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = row[(a >> 2) + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                  const uint32_t b = a + 4u * u;
                  const float d0 = f_sub(q[b], p[u].x), d1 = f_sub(q[b + 1], p[u].y);
                  const float d2 = f_sub(q[b + 2], p[u].z), d3 = f_sub(q[b + 3], p[u].w);
                  acc = f_add(acc, f_mul(d0, d0));
                  acc = f_add(acc, f_mul(d1, d1));
                  acc = f_add(acc, f_mul(d2, d2));
                  acc = f_add(acc, f_mul(d3, d3));
                }
              }
            } else if ((dim & 3u) == 0u) {
              for (; a < dim; a += 4) {
                const float4 p = row[a >> 2];
                const float d0 = f_sub(q[a], p.x), d1 = f_sub(q[a + 1], p.y);
                const float d2 = f_sub(q[a + 2], p.z), d3 = f_sub(q[a + 3], p.w);
                acc = f_add(acc, f_mul(d0, d0));
                acc = f_add(acc, f_mul(d1, d1));
                acc = f_add(acc, f_mul(d2, d2));
                acc = f_add(acc, f_mul(d3, d3));
              }
            } else {
              const float* r1 = f.points + (uint64_t)idx * dim;
              for (; a < dim; ++a) {
                const float d0 = f_sub(q[a], r1[a]);
                acc = f_add(acc, f_mul(d0, d0));
              }
            }
            d = acc;
          }
          bool pending = has;
          for (;;) {
            const uint64_t m = __ballot(pending && worst > d);  // search_visitor.hpp:107
            if (m == 0ull) break;
            const int src = __builtin_ctzll(m);
            const float cd = lane_value(d, src);
            const int32_t ci = lane_value(idx, src);
            if ((int)lane == src) pending = false;
            if (__ballot(lane < filled && li == ci) != 0ull) continue;  // already listed
            // insert_sorted (search_visitor.hpp:24-38): behind every entry that is not larger.
            const uint32_t pos = (uint32_t)__popcll(__ballot(lane < filled && !(cd < ld)));
            const float up_d = __shfl_up(ld, 1);
            const int32_t up_i = __shfl_up(li, 1);
            if (filled < k) ++filled;
            if (lane > pos && lane < filled) {
              ld = up_d;
              li = up_i;
            } else if (lane == pos && lane < filled) {
              ld = cd;
              li = ci;
            }
            worst = filled == k ? lane_value(ld, (int)k - 1) : kFltMax;
          }
        }
      }
      ++leaves_visited;

      // ---- back up: queue the far children that are still closer than max() ----
      if (depth > kForestPath) depth = kForestPath;  // deeper levels were not recorded (never in practice)
      __syncthreads();
      for (uint32_t l = depth; l-- > 0;) {
        const float d = path_d[l];
        if (worst > d) {  // priority_search:126
          if (qn < kForestQueue) {
            if (lane == 0) {
              queue_d[qn] = d;
              queue_ref[qn] = path_ref[l];
            }
            ++qn;
          } else if (lane == 0 && dropped) {
            atomicAdd(dropped, 1u);
          }
        }
      }
      __syncthreads();
    }
    __syncthreads();
  }

  if (lane < k) {
    Neighbor nb;
    nb.index = lane < filled ? li : -1;
    nb.distance = lane < filled ? ld : kFltMax;
    out[qi * k + lane] = nb;
  }
  (void)lanes_below;
}

}  // namespace ptk
