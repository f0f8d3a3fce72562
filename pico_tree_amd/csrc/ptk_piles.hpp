// ptk_piles.hpp -- the k = 1 view of a tree that holds PILES: subtrees all of whose points are the same point
// (plain C++17, no HIP types: compiled into the backend and into the CPU test tier).
//
// The reference splits a node of coincident points like any other (kd_tree_builder.hpp:255-275: the plane slides to
// the one coordinate there is and one point is peeled off per level), so a pile of n identical points is a chain n
// levels deep whose leaves hold one point each -- coordinates snapped to a grid of 1.0 give BASELINE config 2 55 k
// piles of up to 615 points and a tree 627 levels deep, 28 without them.  A nearest-neighbour search
// (search_visitor.hpp:42-65) that reaches such a subtree measures the same distance n times; only the FIRST point it
// visits can become the answer (`if (max > d)` is strict, :55), and which one that is follows from the traversal
// alone: the nearer child first (kd_tree_search.hpp:76-92), where inside a pile every plane of an axis is the pile's
// coordinate c on that axis (left_max == right_min == c: the bounds are the points' own; +0 and -0 alike) and the test
// `((c + c) - v) - v > 0` gives the same answer at every level -- three bits per query.
//
// The k = 1 view replaces every maximal pile by a leaf of ONE point (the first record of the pile's range, whose
// index stands for the whole pile) and keeps everything else: same branch records above, same planes, same point
// records (the view shares the point array of the tree; a pile leaf has its full range but a count of one).  The
// k = 1 kernels run on the view unchanged; a pass over the rows then gives every row whose index is a pile's
// stand-in the index the reference reports: first[bits] of that pile, found at creation by walking the pile's chain
// once per bit pattern.  Everything that depends on visiting the points of a pile one by one (k > 1: the order of
// equal distances; radius and box rows) keeps the full tree.
#pragma once

#include <atomic>
#include <cstdint>
#include <cstring>
#include <vector>

#include "ptk.h"
#include "ptk_encode.hpp"

namespace ptk {

struct PileRecord {   // 48 bytes
  float c[3];         // the pile's point (0 for axes the space does not have)
  uint32_t count;     // points in the pile
  int32_t first[8];   // the index the reference reports, by bit a = (query goes left on axis a)
};

struct PileView {
  std::vector<ptk_node> nodes;        // the stream with every maximal pile replaced by one leaf
  std::vector<uint8_t> single;        // per node of `nodes`: 1 = pile leaf (encoded with a count of one)
  std::vector<PileRecord> piles;
  std::vector<uint32_t> pile_of_point;  // [n_points] 1 + pile of the point that stands for it, else 0
  uint64_t pile_points = 0;           // points the piles hold
  bool empty() const { return piles.empty(); }
};

namespace piles_detail {
inline int32_t leaf_a(const ptk_node& nd) {
  int32_t v;
  std::memcpy(&v, &nd.a, 4);
  return v;
}
inline int32_t leaf_b(const ptk_node& nd) {
  int32_t v;
  std::memcpy(&v, &nd.b, 4);
  return v;
}
}  // namespace piles_detail

// Finds the piles of a validated stream and builds the view; an empty view when there are none.  A tree of distinct
// points costs one threaded pass over the branch records (and a look at two points where the bounds of a branch meet).
inline void build_pile_view(uint32_t dim, uint64_t n_points, const float* points, const ptk_node* nodes,
                            uint64_t n_nodes, const int32_t* indices, PileView& out, unsigned threads = 1) {
  using piles_detail::leaf_a;
  using piles_detail::leaf_b;
  out = PileView{};
  // "The same" is the equality the searches see: +0 and -0 are one coordinate (they give the same differences, the
  // same side tests), a NaN is none.
  auto bounds_meet = [&](const ptk_node& nd) {
    float lm, rm;
    std::memcpy(&lm, &nd.a, 4);
    std::memcpy(&rm, &nd.b, 4);
    return lm == rm;
  };
  auto same_point = [&](int32_t p, int32_t q) {
    const float* x = points + (uint64_t)p * dim;
    const float* y = points + (uint64_t)q * dim;
    for (uint32_t d = 0; d < dim; ++d)
      if (!(x[d] == y[d])) return false;
    return true;
  };
  auto first_point = [&](uint64_t i) {  // of the subtree at node i: its leftmost leaf is the next leaf of the stream
    while (nodes[i].right != PTK_LEAF) ++i;
    return indices[leaf_a(nodes[i])];
  };
  // The one pass every tree pays, on `threads` threads: is there a branch whose bounds meet AND whose two sides begin
  // with the same point?  (Bounds that meet are common -- the points of a floor share a coordinate --, sides that begin
  // with the same point are a pile or next to one.)
  std::atomic<bool> candidate{false};
  parallel_chunks(n_nodes, threads, size_t(1) << 17, [&](size_t lo, size_t hi, unsigned) {
    for (size_t i = lo; i < hi; ++i) {
      if (nodes[i].right != PTK_LEAF && bounds_meet(nodes[i]) && same_point(first_point(i + 1), first_point(nodes[i].right))) {
        candidate.store(true, std::memory_order_relaxed);
        return;
      }
    }
  });
  if (!candidate.load()) return;

  // Bottom-up (children follow their parent in the stream): 1 = every point below is the same point, rep = one of them.
  // Leaves are looked at only below a branch that could be a pile.
  std::vector<uint8_t> same(n_nodes, 0);
  std::vector<int32_t> rep(n_nodes, 0);
  std::vector<uint64_t> end(n_nodes, 0);  // one past the subtree in the stream
  auto leaf_same = [&](uint64_t i) {
    const int32_t a = leaf_a(nodes[i]), b = leaf_b(nodes[i]);
    rep[i] = indices[a];
    for (int32_t j = a + 1; j < b; ++j)
      if (!same_point(indices[j], rep[i])) return false;
    return b > a;
  };
  for (uint64_t i = n_nodes; i-- > 0;) {
    const ptk_node& nd = nodes[i];
    if (nd.right == PTK_LEAF) {
      end[i] = i + 1;
      continue;  // (same[] of a leaf is worked out by its parent, if that parent asks)
    }
    end[i] = end[nd.right];
    if (!bounds_meet(nd)) continue;
    const uint64_t l = i + 1, r = nd.right;
    const bool ls = nodes[l].right == PTK_LEAF ? leaf_same(l) : same[l] != 0;
    if (!ls) continue;
    const bool rs = nodes[r].right == PTK_LEAF ? leaf_same(r) : same[r] != 0;
    if (!rs || !same_point(rep[l], rep[r])) continue;
    // The planes must meet AT the pile: the side test of the searches is `left_max + right_min` against the query
    // (kd_tree_search.hpp:76), the pile pass decides it from the pile's coordinate.  The reference's builder always
    // puts them there (they are the tightened boxes of the two sides); a stream loaded from elsewhere need not.
    {
      float lm;
      std::memcpy(&lm, &nd.a, 4);
      const uint32_t axis = nd.split_dim;
      if (axis >= dim || !(lm == points[(uint64_t)rep[l] * dim + axis])) continue;
    }
    same[i] = 1;
    rep[i] = rep[l];
  }

  // Top-down: the maximal piles, the new position of every node that stays.
  std::vector<uint64_t> new_index(n_nodes, 0);
  uint64_t kept = 0;
  for (uint64_t i = 0; i < n_nodes;) {
    new_index[i] = kept++;
    i = nodes[i].right != PTK_LEAF && same[i] ? end[i] : i + 1;
  }
  if (kept == n_nodes) return;  // (branches with equal bounds, none of them a pile)
  out.nodes.reserve(kept);
  out.single.reserve(kept);
  out.pile_of_point.assign(n_points, 0u);
  for (uint64_t i = 0; i < n_nodes;) {
    const ptk_node& nd = nodes[i];
    if (nd.right == PTK_LEAF) {
      out.nodes.push_back(nd);
      out.single.push_back(0);
      ++i;
    } else if (!same[i]) {
      ptk_node keep = nd;
      keep.right = (uint32_t)new_index[nd.right];
      out.nodes.push_back(keep);
      out.single.push_back(0);
      ++i;
    } else {
      // The leaves below tile a range of the permutation: the first leaf of the subtree is the next node that is a
      // leaf, the last one is the node before end[i].
      uint64_t fl = i;
      while (nodes[fl].right != PTK_LEAF) ++fl;
      const int32_t a = leaf_a(nodes[fl]), b = leaf_b(nodes[end[i] - 1]);
      ptk_node leaf{};
      std::memcpy(&leaf.a, &a, 4);
      std::memcpy(&leaf.b, &b, 4);
      leaf.right = PTK_LEAF;
      leaf.split_dim = 0;
      out.nodes.push_back(leaf);
      out.single.push_back(1);
      PileRecord rec{};
      const float* p = points + (uint64_t)indices[a] * dim;
      for (uint32_t d = 0; d < 3; ++d) rec.c[d] = d < dim ? p[d] : 0.0f;
      rec.count = (uint32_t)(b - a);
      for (uint32_t bits = 0; bits < 8; ++bits) {  // the nearer child first, all the way down
        uint64_t at = i;
        while (nodes[at].right != PTK_LEAF) at = ((bits >> nodes[at].split_dim) & 1u) ? at + 1 : nodes[at].right;
        rec.first[bits] = indices[leaf_a(nodes[at])];
      }
      out.pile_of_point[(uint64_t)indices[a]] = (uint32_t)out.piles.size() + 1u;
      out.piles.push_back(rec);
      out.pile_points += rec.count;
      i = end[i];
    }
  }
}

}  // namespace ptk
