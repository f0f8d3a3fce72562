// ptk_encode.hpp -- host-side re-encoding of the flat DFS tree into the layout
// the gfx950 kernels traverse (plain C++17, no HIP types).
//
// Input : the ptk_node stream of include/ptk.h (DFS pre-order, leaves are nodes,
//         the reference's on-disk order: internal/kd_tree_data.hpp:109-135),
//         the leaf-ordered `indices` permutation and the points in original order.
// Output: * one 16-byte record per BRANCH only:
//             {left_max, right_min, left_ref, right_ref}
//           Leaves disappear as nodes: a child reference that points at a leaf
//           carries the leaf's [begin, count) itself, so reaching a leaf costs no
//           extra dependent load.
//         * points copied into LEAF ORDER as 16-byte records {x, y, z, index}
//           (z = 0 when dim < 3; y = z = 0 when dim == 1), so a leaf is one
//           contiguous run of 16-byte loads and needs no index indirection;
//           `begin` counts records of this array (leaves may be padded apart,
//           see kEncLeafAlign), not positions of `indices`.
//         Reference encoding (32 bits):
//             bit 31 = 1: leaf,   bits 30:0  = (begin << cbits) | count
//             bit 31 = 0: branch, bits 30:29 = split axis OF THE CHILD,
//                                 bits 28:0  = branch index
//         Zero-padding unused axes is exact: the padded terms contribute
//         (0 - 0)^2 = +0 to every distance and x + 0 == x bit-for-bit for the
//         non-negative partial sums involved.

#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <utility>
#include <thread>
#include <vector>

#include "ptk.h"

namespace ptk {

struct EncNode {
  uint32_t left_max_bits;
  uint32_t right_min_bits;
  uint32_t left_ref;
  uint32_t right_ref;
};
struct EncPoint {
  float x, y, z;
  int32_t index;
};
static_assert(sizeof(EncNode) == 16 && sizeof(EncPoint) == 16, "device record sizes");

constexpr uint32_t kEncLeafBit = 0x80000000u;
constexpr uint32_t kEncLeafPad = 8;    // readable records past the last point (== kLeafPad)
// Leaves start on multiples of this many 16-byte records.  8 (one leaf = one 128-byte line) was
// measured on BASELINE config 2 and is NOT worth it: +25 % point bytes, traversal 3 % slower
// (profiles/r01e_notes.txt) -- the divergent-gather rate, not line straddling, bounds the leaf scan.
constexpr uint64_t kEncLeafAlign = 1;

struct TreeStats {
  uint64_t n_leaves = 0;
  uint32_t max_leaf_count = 0;
  uint32_t max_depth = 0;
};

struct EncRange {
  uint32_t begin, end;
};

struct EncodedTree {
  std::vector<EncNode> nodes;   // at least one element (dummy when the root is a leaf)
  std::vector<EncRange> ranges; // per branch: record range [begin, end) of its whole subtree (box search)
  std::vector<EncPoint> points; // n_points + kEncLeafPad (the tail repeats the last point)
  uint32_t root_ref = 0;
  uint32_t cbits = 0;
};

inline uint32_t bits_for(uint64_t v) {  // bits needed to represent 0..v
  uint32_t b = 0;
  while (v) {
    ++b;
    v >>= 1;
  }
  return b;
}

// Validates the DFS stream and gathers statistics; optionally numbers branches.
// Returns an empty string on success, an error message otherwise.
inline std::string analyse_stream(
    uint32_t dim, uint64_t n_points, const ptk_node* nodes, uint64_t n_nodes, TreeStats& st,
    std::vector<uint32_t>* branch_id) {
  if (n_nodes == 0) return "empty node stream";
  if (branch_id) branch_id->assign(n_nodes, 0);
  st = TreeStats{};
  uint32_t n_branch = 0;
  std::vector<std::pair<uint32_t, uint32_t>> pending;  // (right child index, its depth)
  uint32_t depth = 0;
  // The leaves must tile [0, n_points) in stream order (as the reference builds them,
  // kd_tree_builder.hpp:352-396): the device layout places the points of a leaf by a running sum of
  // the leaf sizes, so overlapping or out-of-order ranges would silently search the wrong points.
  uint64_t next_begin = 0;
  for (uint64_t i = 0; i < n_nodes; ++i) {
    if (!pending.empty() && pending.back().first == i) {
      depth = pending.back().second;
      pending.pop_back();
    }
    if (depth > st.max_depth) st.max_depth = depth;
    const ptk_node& nd = nodes[i];
    if (nd.right == PTK_LEAF) {
      int32_t b, e;
      std::memcpy(&b, &nd.a, 4);
      std::memcpy(&e, &nd.b, 4);
      if (b < 0 || e < b || (uint64_t)e > n_points)
        return "leaf " + std::to_string(i) + " has a bad index range";
      if ((uint64_t)b != next_begin)
        return "leaf " + std::to_string(i) + " does not start where the previous leaf ended";
      next_begin = (uint64_t)e;
      ++st.n_leaves;
      if ((uint32_t)(e - b) > st.max_leaf_count) st.max_leaf_count = (uint32_t)(e - b);
    } else {
      if (nd.right <= i + 1 || nd.right >= n_nodes)
        return "branch " + std::to_string(i) + " has a bad right child";
      if (nd.split_dim >= dim) return "branch " + std::to_string(i) + " splits on an axis >= dim";
      if (branch_id) (*branch_id)[i] = n_branch;
      ++n_branch;
      pending.emplace_back(nd.right, depth + 1);
      ++depth;  // the left child follows immediately
    }
  }
  if (!pending.empty()) return "node stream is truncated";
  if (st.n_leaves != (uint64_t)n_branch + 1) return "node stream is not a binary tree";
  if (next_begin != n_points) return "the leaves do not cover all points";
  return std::string();
}

// Re-encodes for the device.  `unsupported` is set when the tree is valid but
// does not fit the 32-bit reference encoding.
inline std::string encode_tree(
    uint32_t dim, uint64_t n_points, const float* points, const ptk_node* nodes, uint64_t n_nodes,
    const int32_t* indices, TreeStats& st, EncodedTree& out, bool& unsupported, bool with_points = true,
    const uint8_t* leaf_single = nullptr, uint32_t force_cbits = 0) {
  // with_points = false: the caller fills the point records itself (the backend gathers them on
  // the device: encode_points_kernel); `points` may then be null.
  // leaf_single (per node; the k = 1 view of a tree with piles, ptk_piles.hpp): leaves marked 1 keep their range --
  // the point records are the full tree's -- but are referenced with a count of one; force_cbits: the count bits of
  // the full tree, whose point array the view shares.
  unsupported = false;
  if (dim == 0 || dim > 3) {
    unsupported = true;
    return "only dim <= 3 is encoded for the device";
  }
  std::vector<uint32_t> branch_id;
  std::string err = analyse_stream(dim, n_points, nodes, n_nodes, st, &branch_id);
  if (!err.empty()) return err;

  // Device positions of the leaves (kEncLeafAlign records apart at least; gaps, if any, repeat
  // the leaf's last point and are never visited: the count bounds every scan).
  std::vector<uint64_t> leaf_pos(n_nodes, 0);
  uint64_t n_slots = 0;
  for (uint64_t i = 0; i < n_nodes; ++i) {
    const ptk_node& nd = nodes[i];
    if (nd.right != PTK_LEAF) continue;
    leaf_pos[i] = n_slots;
    n_slots += ((uint64_t)(nd.b - nd.a) + kEncLeafAlign - 1) / kEncLeafAlign * kEncLeafAlign;
  }
  const uint64_t n_branch = n_nodes - st.n_leaves;
  if (leaf_single != nullptr) {  // (the statistics are those of what a search visits)
    st.max_leaf_count = 0;
    for (uint64_t i = 0; i < n_nodes; ++i) {
      if (nodes[i].right != PTK_LEAF) continue;
      const uint32_t c = leaf_single[i] ? 1u : nodes[i].b - nodes[i].a;
      if (c > st.max_leaf_count) st.max_leaf_count = c;
    }
  }
  const uint32_t cbits = force_cbits ? force_cbits : bits_for(st.max_leaf_count);
  if (bits_for(st.max_leaf_count) > cbits) return "a leaf does not fit the count bits asked for";
  const uint32_t bbits = bits_for(n_slots);
  if (cbits + bbits > 31) {
    unsupported = true;
    return "leaf reference needs " + std::to_string(bbits) + " + " + std::to_string(cbits) +
           " bits (> 31): n_points x max leaf size too large";
  }
  if (n_branch >= (1ull << 28)) {
    unsupported = true;
    return "more than 2^28 branch nodes";
  }

  auto ref_of = [&](uint64_t i) -> uint32_t {
    const ptk_node& nd = nodes[i];
    if (nd.right == PTK_LEAF)
      return kEncLeafBit | ((uint32_t)leaf_pos[i] << cbits) | (leaf_single != nullptr && leaf_single[i] ? 1u : nd.b - nd.a);
    return (nd.split_dim << 29) | branch_id[i];
  };

  out.nodes.assign(n_branch > 0 ? n_branch : 1, EncNode{0, 0, 0, 0});
  for (uint64_t i = 0; i < n_nodes; ++i) {
    const ptk_node& nd = nodes[i];
    if (nd.right == PTK_LEAF) continue;
    EncNode r;
    r.left_max_bits = nd.a;
    r.right_min_bits = nd.b;
    r.left_ref = ref_of(i + 1);
    r.right_ref = ref_of(nd.right);
    out.nodes[branch_id[i]] = r;
  }
  out.points.clear();
  if (with_points) out.points.assign(n_slots + kEncLeafPad, EncPoint{0.0f, 0.0f, 0.0f, 0});
  for (uint64_t i = 0; i < n_nodes && with_points; ++i) {
    const ptk_node& nd = nodes[i];
    if (nd.right != PTK_LEAF) continue;
    const uint64_t count = nd.b - nd.a;
    const uint64_t slots = (count + kEncLeafAlign - 1) / kEncLeafAlign * kEncLeafAlign;
    EncPoint v{0.0f, 0.0f, 0.0f, 0};
    for (uint64_t j = 0; j < slots; ++j) {
      if (j < count) {
        const int32_t idx = indices[nd.a + j];
        if (idx < 0 || (uint64_t)idx >= n_points) return "index out of range in the permutation";
        const float* p = points + (uint64_t)idx * dim;
        v.x = p[0];
        v.y = dim > 1 ? p[1] : 0.0f;
        v.z = dim > 2 ? p[2] : 0.0f;
        v.index = idx;
      }
      out.points[leaf_pos[i] + j] = v;
    }
  }
  for (uint32_t i = 0; i < kEncLeafPad && with_points; ++i) out.points[n_slots + i] = out.points[n_slots ? n_slots - 1 : 0];
  // Subtree ranges: children come later in the stream, so one backward pass suffices.
  static_assert(kEncLeafAlign == 1, "subtree ranges assume leaves are packed without gaps");
  {
    std::vector<EncRange> of_node(n_nodes);
    for (uint64_t i = n_nodes; i-- > 0;) {
      const ptk_node& nd = nodes[i];
      if (nd.right == PTK_LEAF) {
        of_node[i] = EncRange{(uint32_t)leaf_pos[i], (uint32_t)leaf_pos[i] + (nd.b - nd.a)};
      } else {
        of_node[i] = EncRange{of_node[i + 1].begin, of_node[nd.right].end};
      }
    }
    out.ranges.assign(n_branch > 0 ? n_branch : 1, EncRange{0, 0});
    for (uint64_t i = 0; i < n_nodes; ++i)
      if (nodes[i].right != PTK_LEAF) out.ranges[branch_id[i]] = of_node[i];
  }
  out.root_ref = ref_of(0);
  out.cbits = cbits;
  return std::string();
}

// fn(lo, hi, chunk) over [0, n) cut into at most `threads` chunks of at least `grain` items, one thread each.
template <typename Fn_>
inline void parallel_chunks(size_t n, unsigned threads, size_t grain, Fn_&& fn) {
  const size_t want = grain > 0 ? (n + grain - 1) / grain : 1;
  const unsigned chunks = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(threads ? threads : 1, 64), want));
  const size_t per = (n + chunks - 1) / chunks;
  if (chunks == 1) {
    fn(size_t(0), n, 0u);
    return;
  }
  std::vector<std::thread> pool;
  for (unsigned c = 1; c < chunks; ++c) pool.emplace_back([&fn, c, per, n] { fn(std::min(n, per * c), std::min(n, per * (c + 1)), c); });
  fn(size_t(0), std::min(n, per), 0u);
  for (auto& t : pool) t.join();
}

// encode_tree() without the point records for a stream the library's own builder has just made: nothing to validate,
// the statistics are the builder's (`st`: n_leaves, max_leaf_count, max_depth), and every pass runs on `threads`
// threads.  The leaves of such a stream tile [0, n_points) in order, so a leaf's device position is its `a`.
inline std::string encode_tree_of_builder(
    uint32_t dim, uint64_t n_points, const ptk_node* nodes, uint64_t n_nodes, const TreeStats& st, EncodedTree& out,
    bool& unsupported, unsigned threads) {
  static_assert(kEncLeafAlign == 1, "leaf positions assume packed leaves");
  unsupported = false;
  if (dim == 0 || dim > 3) {
    unsupported = true;
    return "only dim <= 3 is encoded for the device";
  }
  const uint64_t n_branch = n_nodes - st.n_leaves;
  const uint32_t cbits = bits_for(st.max_leaf_count);
  const uint32_t bbits = bits_for(n_points);
  if (cbits + bbits > 31) {
    unsupported = true;
    return "leaf reference needs " + std::to_string(bbits) + " + " + std::to_string(cbits) +
           " bits (> 31): n_points x max leaf size too large";
  }
  if (n_branch >= (1ull << 28)) {
    unsupported = true;
    return "more than 2^28 branch nodes";
  }
  // Branch numbers: branches per chunk, their running sum, then the numbers.
  std::vector<uint32_t> branch_id(n_nodes);
  std::vector<uint64_t> chunk_count(65, 0);
  const size_t grain = 1u << 15;
  parallel_chunks(n_nodes, threads, grain, [&](size_t lo, size_t hi, unsigned c) {
    uint64_t count = 0;
    for (size_t i = lo; i < hi; ++i) count += nodes[i].right != PTK_LEAF ? 1 : 0;
    chunk_count[c + 1] = count;
  });
  for (size_t c = 0; c + 1 < chunk_count.size(); ++c) chunk_count[c + 1] += chunk_count[c];
  parallel_chunks(n_nodes, threads, grain, [&](size_t lo, size_t hi, unsigned c) {
    uint32_t next = (uint32_t)chunk_count[c];
    for (size_t i = lo; i < hi; ++i) branch_id[i] = nodes[i].right != PTK_LEAF ? next++ : 0u;
  });
  auto ref_of = [&](uint64_t i) -> uint32_t {
    const ptk_node& nd = nodes[i];
    if (nd.right == PTK_LEAF) return kEncLeafBit | (nd.a << cbits) | (nd.b - nd.a);
    return (nd.split_dim << 29) | branch_id[i];
  };
  out.nodes.assign(n_branch > 0 ? n_branch : 1, EncNode{0, 0, 0, 0});
  out.ranges.assign(n_branch > 0 ? n_branch : 1, EncRange{0, 0});
  out.points.clear();
  parallel_chunks(n_nodes, threads, grain, [&](size_t lo, size_t hi, unsigned) {
    for (size_t i = lo; i < hi; ++i) {
      const ptk_node& nd = nodes[i];
      if (nd.right == PTK_LEAF) continue;
      EncNode r;
      r.left_max_bits = nd.a;
      r.right_min_bits = nd.b;
      r.left_ref = ref_of(i + 1);
      r.right_ref = ref_of(nd.right);
      out.nodes[branch_id[i]] = r;
      // The records of the subtree: from the first point of its leftmost leaf (left children follow their parent
      // in the stream) to the last of its rightmost one.
      size_t l = i + 1, rr = nd.right;
      while (nodes[l].right != PTK_LEAF) ++l;
      while (nodes[rr].right != PTK_LEAF) rr = nodes[rr].right;
      out.ranges[branch_id[i]] = EncRange{nodes[l].a, nodes[rr].b};
    }
  });
  out.root_ref = ref_of(0);
  out.cbits = cbits;
  return std::string();
}

// ---- any dimension (dim > 3): see ptk_kernels_nd.hpp -----------------------------------------
struct EncodedTreeND {
  std::vector<EncNode> nodes;     // per branch; refs carry no axis
  std::vector<uint32_t> axes;     // per branch
  std::vector<float> points;      // leaf order, row-major (n_points x dim)
  std::vector<int32_t> index;     // leaf order: original index
  std::vector<EncRange> ranges;   // per branch: position range [begin, end) of its whole subtree (box search)
  uint32_t root_ref = 0;
  uint32_t cbits = 0;
};

inline std::string encode_tree_nd(
    uint32_t dim, uint64_t n_points, const float* points, const ptk_node* nodes, uint64_t n_nodes,
    const int32_t* indices, TreeStats& st, EncodedTreeND& out, bool& unsupported) {
  unsupported = false;
  std::vector<uint32_t> branch_id;
  std::string err = analyse_stream(dim, n_points, nodes, n_nodes, st, &branch_id);
  if (!err.empty()) return err;
  const uint64_t n_branch = n_nodes - st.n_leaves;
  const uint32_t cbits = bits_for(st.max_leaf_count);
  if (cbits + bits_for(n_points) > 31) {
    unsupported = true;
    return "leaf reference does not fit 31 bits: n_points x max leaf size too large";
  }
  if (n_branch >= (1ull << 30)) {
    unsupported = true;
    return "more than 2^30 branch nodes";
  }
  auto ref_of = [&](uint64_t i) -> uint32_t {
    const ptk_node& nd = nodes[i];
    if (nd.right == PTK_LEAF) return kEncLeafBit | (nd.a << cbits) | (nd.b - nd.a);
    return branch_id[i];
  };
  out.nodes.assign(n_branch > 0 ? n_branch : 1, EncNode{0, 0, 0, 0});
  out.axes.assign(n_branch > 0 ? n_branch : 1, 0u);
  for (uint64_t i = 0; i < n_nodes; ++i) {
    const ptk_node& nd = nodes[i];
    if (nd.right == PTK_LEAF) continue;
    out.nodes[branch_id[i]] = EncNode{nd.a, nd.b, ref_of(i + 1), ref_of(nd.right)};
    out.axes[branch_id[i]] = nd.split_dim;
  }
  out.points.resize(n_points * dim);
  out.index.resize(n_points);
  for (uint64_t pos = 0; pos < n_points; ++pos) {
    const int32_t idx = indices[pos];
    if (idx < 0 || (uint64_t)idx >= n_points) return "index out of range in the permutation";
    std::memcpy(&out.points[pos * dim], points + (uint64_t)idx * dim, dim * sizeof(float));
    out.index[pos] = idx;
  }
  {  // subtree ranges: children come later in the stream, so one backward pass suffices
    std::vector<EncRange> of_node(n_nodes);
    for (uint64_t i = n_nodes; i-- > 0;) {
      const ptk_node& nd = nodes[i];
      of_node[i] = nd.right == PTK_LEAF ? EncRange{nd.a, nd.b} : EncRange{of_node[i + 1].begin, of_node[nd.right].end};
    }
    out.ranges.assign(n_branch > 0 ? n_branch : 1, EncRange{0, 0});
    for (uint64_t i = 0; i < n_nodes; ++i)
      if (nodes[i].right != PTK_LEAF) out.ranges[branch_id[i]] = of_node[i];
  }
  out.root_ref = ref_of(0);
  out.cbits = cbits;
  return std::string();
}

// ---- double points, any dimension: see ptk_kernels_f64.hpp -------------------------------------
struct EncNode64 {  // == ptk::Node64
  double left_max, right_min;
  uint32_t left_ref, right_ref, axis, pad_;
};
static_assert(sizeof(EncNode64) == 32, "device record size");

constexpr uint32_t kStride64D3 = 4;  // doubles per point record of a double tree with dim <= 3: {x, y, z, index}
struct EncodedTree64 {
  std::vector<EncNode64> nodes;   // per branch
  std::vector<EncRange> ranges;   // per branch: position range [begin, end) of its whole subtree
  std::vector<double> points;     // leaf order, row-major (n_points x stride)
  uint32_t stride = 0;            // doubles per point: 4 for dim <= 3 (unused axes zero, the index in the fourth), else dim
  uint32_t root_ref = 0;
  uint32_t cbits = 0;
};

// `fn`: the host builder's DFS pre-order nodes (pico_tree::internal::flat_node<int, double>: members
// is_leaf(), begin / end, left_max / right_min, right, split_dim).  encode = false only validates
// the stream and gathers the statistics.
template <class NodeT>
inline std::string encode_tree64(
    uint32_t dim, uint64_t n_points, const double* points, const NodeT* fn, uint64_t n_nodes, const int32_t* indices,
    TreeStats& st, EncodedTree64& out, bool& unsupported, bool encode = true) {
  unsupported = false;
  // analyse_stream works on ptk_node records: hand it the structure (child links, leaf ranges,
  // split axes); the planes are not part of the check.
  std::vector<ptk_node> shape(n_nodes);
  for (uint64_t i = 0; i < n_nodes; ++i) {
    if (fn[i].is_leaf()) {
      shape[i] = ptk_node{(uint32_t)fn[i].begin, (uint32_t)fn[i].end, PTK_LEAF, 0};
    } else {
      shape[i] = ptk_node{0, 0, fn[i].right, fn[i].split_dim};
    }
  }
  std::vector<uint32_t> branch_id;
  std::string err = analyse_stream(dim, n_points, shape.data(), n_nodes, st, &branch_id);
  if (!err.empty() || !encode) return err;
  const uint64_t n_branch = n_nodes - st.n_leaves;
  const uint32_t cbits = bits_for(st.max_leaf_count);
  if (cbits + bits_for(n_points) > 31) {
    unsupported = true;
    return "leaf reference does not fit 31 bits: n_points x max leaf size too large";
  }
  if (n_branch >= (1ull << 30)) {
    unsupported = true;
    return "more than 2^30 branch nodes";
  }
  auto ref_of = [&](uint64_t i) -> uint32_t {
    if (fn[i].is_leaf()) return kEncLeafBit | ((uint32_t)fn[i].begin << cbits) | (uint32_t)(fn[i].end - fn[i].begin);
    return branch_id[i];
  };
  out.nodes.assign(n_branch > 0 ? n_branch : 1, EncNode64{0, 0, 0, 0, 0, 0});
  out.ranges.assign(out.nodes.size(), EncRange{0, 0});
  {
    std::vector<EncRange> of_node(n_nodes);  // children come later in the stream: one backward pass
    for (uint64_t i = n_nodes; i-- > 0;) {
      of_node[i] = fn[i].is_leaf() ? EncRange{(uint32_t)fn[i].begin, (uint32_t)fn[i].end}
                                   : EncRange{of_node[i + 1].begin, of_node[fn[i].right].end};
    }
    for (uint64_t i = 0; i < n_nodes; ++i) {
      if (fn[i].is_leaf()) continue;
      out.nodes[branch_id[i]] =
          EncNode64{fn[i].left_max, fn[i].right_min, ref_of(i + 1), ref_of(fn[i].right), fn[i].split_dim, 0};
      out.ranges[branch_id[i]] = of_node[i];
    }
  }
  // dim <= 3: a point is one 32-byte record {x, y, z, original index} -- a lane fetches it with one aligned gather
  // instead of 24 bytes that straddle sectors plus a word of the index array (r06).  The index sits in the low 32 bits
  // of the fourth double; the kernels never read that double as a number.
  out.stride = dim <= 3 ? kStride64D3 : dim;
  out.points.assign((size_t)n_points * out.stride, 0.0);
  for (uint64_t pos = 0; pos < n_points; ++pos) {
    const int32_t idx = indices[pos];
    if (idx < 0 || (uint64_t)idx >= n_points) return "index out of range in the permutation";
    std::memcpy(&out.points[pos * out.stride], points + (uint64_t)idx * dim, dim * sizeof(double));
    if (dim <= 3) {
      const int64_t tag = idx;
      std::memcpy(&out.points[pos * out.stride + 3], &tag, sizeof(tag));
    }
  }
  out.root_ref = ref_of(0);
  out.cbits = cbits;
  return std::string();
}

// The two outer bounds of every branch -- {left_min, right_max}, per stream node in `outer` -- in the branch order of
// encode_tree64's records (what the topological metrics read beside them).
template <class NodeT, class PairT>
inline std::string encode_outer64(uint32_t dim, uint64_t n_points, const NodeT* fn, uint64_t n_nodes, const PairT* outer,
                                  std::vector<double>& out) {
  std::vector<ptk_node> shape(n_nodes);
  for (uint64_t i = 0; i < n_nodes; ++i) {
    if (fn[i].is_leaf()) {
      shape[i] = ptk_node{(uint32_t)fn[i].begin, (uint32_t)fn[i].end, PTK_LEAF, 0};
    } else {
      shape[i] = ptk_node{0, 0, fn[i].right, fn[i].split_dim};
    }
  }
  TreeStats st;
  std::vector<uint32_t> branch_id;
  std::string err = analyse_stream(dim, n_points, shape.data(), n_nodes, st, &branch_id);
  if (!err.empty()) return err;
  const uint64_t n_branch = n_nodes - st.n_leaves;
  out.assign(2 * (n_branch > 0 ? n_branch : 1), 0.0);
  for (uint64_t i = 0; i < n_nodes; ++i) {
    if (fn[i].is_leaf()) continue;
    out[2 * (size_t)branch_id[i]] = outer[i][0];
    out[2 * (size_t)branch_id[i] + 1] = outer[i][1];
  }
  return std::string();
}

}  // namespace ptk
