// ptk_forest_host.hpp -- host side of the kd-forest: reflection vectors, per-tree build and
// encoding (plain C++17, no HIP types; shared by ptk_backend.hip and the CPU kernel emulator).

#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <random>
#include <string>
#include <vector>

#include "pico_tree/internal/flat_tree.hpp"
#include "pico_tree/map.hpp"
#include "ptk_encode.hpp"

namespace ptk {

// One node of a forest tree, 32 bytes.  Branch: bounds of both children on the split axis
// (the reference's kd_tree_node_topological, internal/kd_tree_node.hpp:56-67).
struct ForestNode {
  float left_min, left_max, right_min, right_max;
  uint32_t left_ref, right_ref;  // bit 31 = leaf: (begin << cbits) | count; else block * 8 + slot
  uint32_t split_dim;
  uint32_t stream_id;            // position in the depth-first stream (orders equal queue distances)
};
static_assert(sizeof(ForestNode) == 32, "forest node layout");

constexpr uint32_t kForestQueue = 1024;   // queued nodes per query and tree
constexpr uint32_t kForestPath = 96;      // deepest descent recorded (tree depth limit)

struct ForestTreeHost {
  std::vector<ForestNode> nodes;
  std::vector<int32_t> indices;
  uint32_t root_ref = 0;
  uint32_t cbits = 0;
  uint32_t max_depth = 0;
};

// Unit vector with independent N(0,1) components before normalisation: std::mt19937 words
// through Box-Muller in double (the reference: std::normal_distribution on a
// std::random_device seed, rkd_tree_hh_data.hpp:14-29).
inline void reflection_vector(uint64_t seed, uint32_t tree, uint32_t dim, float* out) {
  std::mt19937 gen(static_cast<uint32_t>(seed * 1000003ull + tree * 7919ull + 12345ull));
  double norm2 = 0.0;
  std::vector<double> v(dim);
  for (uint32_t i = 0; i < dim; ++i) {
    const double u1 = ((gen() >> 8) + 0.5) * (1.0 / 16777216.0);
    const double u2 = ((gen() >> 8) + 0.5) * (1.0 / 16777216.0);
    v[i] = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    norm2 += v[i] * v[i];
  }
  const double inv = 1.0 / std::sqrt(norm2 > 0 ? norm2 : 1.0);
  for (uint32_t i = 0; i < dim; ++i) out[i] = static_cast<float>(v[i] * inv);
}

// Builds one tree over the points reflected by r (`rotated` is scratch of n * dim floats).
// Returns an empty string or an error message.
inline std::string build_forest_tree(const float* points, uint64_t n, uint32_t dim, uint64_t max_leaf_size,
                                     const float* r, std::vector<float>& rotated, ForestTreeHost& out,
                                     unsigned threads = 1) {
  using namespace pico_tree;
  using space_t = space_map<point_map<float const, dynamic_extent>>;
  rotated.resize(n * dim);
  // y = x - (2 (r.x)) r, the dot product summed left to right (rkd_tree_hh_data.hpp:80-90).
  for (uint64_t i = 0; i < n; ++i) {
    const float* x = points + i * dim;
    float* y = rotated.data() + i * dim;
    float dot = 0.0f;
    for (uint32_t a = 0; a < dim; ++a) dot += r[a] * x[a];
    dot *= 2.0f;
    for (uint32_t a = 0; a < dim; ++a) y[a] = x[a] - (dot * r[a]);
  }
  space_t space(rotated.data(), n, dim);
  internal::space_view<space_t> view(space);
  auto flat = internal::build_flat_tree<int>(view, max_leaf_size_t(max_leaf_size), bounds_from_space,
                                             sliding_midpoint_max_side, /*keep_outer_bounds=*/true, threads);
  if (flat.max_depth >= kForestPath)
    return "forest tree is " + std::to_string(flat.max_depth) + " levels deep (limit " +
           std::to_string(kForestPath - 1) + ")";
  // Encode: branches only, children as references (a leaf's range is packed into the reference).
  // Branch records are laid out in 256-byte BLOCKS of three tree levels (slot 0 the block's root,
  // 1-2 its children, 3-6 its grandchildren, 7 unused; blocks in depth-first order of the block
  // tree): the search fetches a block with ONE memory round trip and then walks up to three levels
  // out of registers -- its descents are chains of dependent fetches that miss the L2 (the leaf
  // rows stream through it), so this cuts the chain by up to 3x.
  const size_t n_nodes = flat.nodes.size();
  auto is_branch = [&](size_t i) { return flat.nodes[i].right != internal::flat_leaf_tag; };
  std::vector<uint32_t> branch_id(n_nodes, 0);
  uint32_t max_count = 0;
  for (size_t i = 0; i < n_nodes; ++i)
    if (!is_branch(i)) max_count = std::max<uint32_t>(max_count, (uint32_t)(flat.nodes[i].end - flat.nodes[i].begin));
  uint64_t n_blocks = 0;
  if (is_branch(0)) {
    std::vector<size_t> roots{0};  // LIFO: the leftmost block root is placed next
    while (!roots.empty()) {
      const size_t r = roots.back();
      roots.pop_back();
      const uint64_t base = n_blocks++ * 8;
      size_t level[7];
      bool used[7] = {true, false, false, false, false, false, false};
      level[0] = r;
      for (int sl = 0; sl < 3; ++sl) {  // slots 0..2 have their children in slots 2 sl + 1, 2 sl + 2
        if (!used[sl]) continue;
        const size_t kids[2] = {level[sl] + 1, (size_t)flat.nodes[level[sl]].right};
        for (int c = 0; c < 2; ++c)
          if (is_branch(kids[c])) {
            level[2 * sl + 1 + c] = kids[c];
            used[2 * sl + 1 + c] = true;
          }
      }
      for (int sl = 0; sl < 7; ++sl)
        if (used[sl]) branch_id[level[sl]] = (uint32_t)(base + sl);
      for (int sl = 6; sl >= 3; --sl) {  // the grandchildren's branch children root the next blocks
        if (!used[sl]) continue;
        const size_t kids[2] = {level[sl] + 1, (size_t)flat.nodes[level[sl]].right};
        for (int c = 1; c >= 0; --c)
          if (is_branch(kids[c])) roots.push_back(kids[c]);
      }
    }
  }
  if (n_blocks * 8 >= (1ull << 31)) return "forest tree has too many branch blocks";
  const uint32_t cbits = bits_for(max_count);
  if (cbits + bits_for(n) > 31) return "forest leaf reference does not fit 31 bits";
  auto ref_of = [&](size_t i) -> uint32_t {
    const auto& nd = flat.nodes[i];
    if (nd.right == internal::flat_leaf_tag)
      return kEncLeafBit | ((uint32_t)nd.begin << cbits) | (uint32_t)(nd.end - nd.begin);
    return branch_id[i];
  };
  out.nodes.assign(std::max<uint64_t>(n_blocks, 1) * 8, ForestNode{});
  for (size_t i = 0; i < n_nodes; ++i) {
    const auto& nd = flat.nodes[i];
    if (nd.right == internal::flat_leaf_tag) continue;
    ForestNode o;
    o.left_min = flat.outer_bounds[i][0];
    o.left_max = nd.left_max;
    o.right_min = nd.right_min;
    o.right_max = flat.outer_bounds[i][1];
    o.left_ref = ref_of(i + 1);
    o.right_ref = ref_of(nd.right);
    o.split_dim = nd.split_dim;
    o.stream_id = (uint32_t)i;
    out.nodes[branch_id[i]] = o;
  }
  out.indices.assign(flat.indices.begin(), flat.indices.end());
  out.root_ref = ref_of(0);
  out.cbits = cbits;
  out.max_depth = flat.max_depth;
  return std::string();
}

}  // namespace ptk
