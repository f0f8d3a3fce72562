#pragma once
//! \file stream.hpp
//! \brief Reads and writes a flat_tree in the reference's binary tree format.
//! \details Byte-compatible with kd_tree::save / kd_tree::load of the reference
//! (kd_tree.hpp:336-370, internal/kd_tree_data.hpp:43-58,90-135,
//! internal/stream_wrapper.hpp:25-105): native endianness, no validation.
//!
//!   size_t sdim
//!   size_t n, Index indices[n]
//!   Scalar root_min[sdim], Scalar root_max[sdim]
//!   nodes in depth-first pre-order, each
//!     bool is_leaf (1 byte), then
//!     leaf   { Index begin_idx; Index end_idx; }  or
//!     branch { int split_dim; Scalar left_max; Scalar right_min; }
//!
//! Both records are written as whole structs (kd_tree_data.hpp:113,116), so for Scalar = double
//! the branch carries the 4 bytes of padding the compiler puts after split_dim (24 bytes, not
//! 20).  The reference writes whatever those bytes hold; this writer zeroes them.
//!
//! The flat node array is already in that order, so saving is one linear pass
//! and loading rebuilds only the right-child indices.

#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "flat_tree.hpp"

namespace pico_tree::internal {

inline std::fstream open_stream(std::string const& filename, std::ios_base::openmode mode) {
  std::fstream stream(filename, mode);
  if (!stream.is_open()) {
    throw std::runtime_error("Unable to open file: " + filename);
  }
  return stream;
}

template <typename T_>
inline void put_pod(std::ostream& s, T_ const& v) {
  s.write(reinterpret_cast<char const*>(&v), sizeof(T_));
}
template <typename T_>
inline void get_pod(std::istream& s, T_& v) {
  s.read(reinterpret_cast<char*>(&v), sizeof(T_));
}

// kd_tree_branch_single<Scalar_> (kd_tree_node.hpp:43-50), the layout of a branch on disk.
template <typename Scalar_>
struct stream_branch {
  int split_dim;
  Scalar_ left_max;
  Scalar_ right_min;
};

// kd_tree_branch_double<Scalar_> (kd_tree_node.hpp:52-67): the branch of a tree over a topological
// space (flat_tree::keep_outer_bounds).
template <typename Scalar_>
struct stream_branch_double {
  int split_dim;
  Scalar_ left_min;
  Scalar_ left_max;
  Scalar_ right_min;
  Scalar_ right_max;
};

template <typename Tree_>
inline void write_flat_tree(Tree_ const& tree, std::ostream& s) {
  using index = typename Tree_::index_type;
  using scalar = typename Tree_::scalar_type;
  size_t const sdim = tree.root_box.size();
  put_pod(s, sdim);
  size_t const n = tree.indices.size();
  put_pod(s, n);
  s.write(reinterpret_cast<char const*>(tree.indices.data()),
          static_cast<std::streamsize>(n * sizeof(index)));
  s.write(reinterpret_cast<char const*>(tree.root_box.min()),
          static_cast<std::streamsize>(sdim * sizeof(scalar)));
  s.write(reinterpret_cast<char const*>(tree.root_box.max()),
          static_cast<std::streamsize>(sdim * sizeof(scalar)));
  for (size_t ni = 0; ni < tree.nodes.size(); ++ni) {
    auto const& nd = tree.nodes[ni];
    bool const leaf = nd.is_leaf();
    put_pod(s, leaf);
    if (leaf) {
      put_pod(s, nd.begin);
      put_pod(s, nd.end);
    } else if (tree.keep_outer_bounds) {
      stream_branch_double<scalar> b;
      std::memset(&b, 0, sizeof(b));
      b.split_dim = static_cast<int>(nd.split_dim);
      b.left_min = tree.outer_bounds[ni][0];
      b.left_max = nd.left_max;
      b.right_min = nd.right_min;
      b.right_max = tree.outer_bounds[ni][1];
      put_pod(s, b);
    } else {
      stream_branch<scalar> b;
      std::memset(&b, 0, sizeof(b));  // padding included
      b.split_dim = static_cast<int>(nd.split_dim);
      b.left_max = nd.left_max;
      b.right_min = nd.right_min;
      put_pod(s, b);
    }
  }
}

//! \param expect_sdim, expect_n  when non-zero: what the caller's space says; a stream that
//! disagrees is rejected BEFORE anything is sized from its (unchecked) header fields.
template <typename Tree_>
inline Tree_ read_flat_tree(std::istream& s, bool outer_bounds = false, size_t expect_sdim = 0,
                            size_t expect_n = 0) {
  using index = typename Tree_::index_type;
  using scalar = typename Tree_::scalar_type;
  size_t sdim = 0;
  get_pod(s, sdim);
  if (!s) throw std::runtime_error("kd_tree stream ended early");
  if (expect_sdim != 0 && sdim != expect_sdim)
    throw std::runtime_error("kd_tree stream has another spatial dimension than the space");
  if (sdim == 0 || sdim > (size_t(1) << 20)) throw std::runtime_error("kd_tree stream has an implausible dimension");
  Tree_ tree(sdim);
  tree.keep_outer_bounds = outer_bounds;
  size_t n = 0;
  get_pod(s, n);
  if (!s) throw std::runtime_error("kd_tree stream ended early");
  if (expect_n != 0 && n != expect_n)
    throw std::runtime_error("kd_tree stream indexes another number of points than the space holds");
  tree.indices.resize(n);
  s.read(reinterpret_cast<char*>(tree.indices.data()),
         static_cast<std::streamsize>(n * sizeof(index)));
  s.read(reinterpret_cast<char*>(tree.root_box.min()),
         static_cast<std::streamsize>(sdim * sizeof(scalar)));
  s.read(reinterpret_cast<char*>(tree.root_box.max()),
         static_cast<std::streamsize>(sdim * sizeof(scalar)));

  // Pre-order stream: a branch is followed by its whole left subtree, then its
  // right subtree.  `open` holds branches whose left subtree is being read.
  struct pending {
    std::uint32_t node;
    std::uint32_t depth;
    bool left_done;
  };
  std::vector<pending> open;
  std::uint32_t depth = 0;
  bool more = true;
  while (more) {
    if (!s) throw std::runtime_error("kd_tree stream ended early");
    std::uint32_t const self = static_cast<std::uint32_t>(tree.nodes.size());
    tree.nodes.emplace_back();
    if (outer_bounds) tree.outer_bounds.push_back({scalar(0), scalar(0)});
    if (depth > tree.max_depth) tree.max_depth = depth;
    bool leaf = false;
    get_pod(s, leaf);
    if (!leaf) {
      auto& b = tree.nodes[self];
      if (outer_bounds) {
        stream_branch_double<scalar> rec{};
        get_pod(s, rec);
        b.split_dim = static_cast<std::uint32_t>(rec.split_dim);
        b.left_max = rec.left_max;
        b.right_min = rec.right_min;
        tree.outer_bounds[self] = {rec.left_min, rec.right_max};
      } else {
        stream_branch<scalar> rec{};
        get_pod(s, rec);
        b.split_dim = static_cast<std::uint32_t>(rec.split_dim);
        b.left_max = rec.left_max;
        b.right_min = rec.right_min;
      }
      b.right = 0;
      open.push_back(pending{self, depth, false});
      ++depth;  // next node is the left child
      continue;
    }
    auto& l = tree.nodes[self];
    get_pod(s, l.begin);
    get_pod(s, l.end);
    l.right = flat_leaf_tag;
    l.split_dim = 0;
    ++tree.leaf_count;
    tree.max_leaf_points =
        std::max(tree.max_leaf_points, static_cast<size_t>(l.end - l.begin));
    // A finished subtree: the next node is the right child of the innermost
    // branch still waiting for one.
    for (;;) {
      if (open.empty()) {
        more = false;
        break;
      }
      pending& p = open.back();
      if (!p.left_done) {
        p.left_done = true;
        tree.nodes[p.node].right = static_cast<std::uint32_t>(tree.nodes.size());
        depth = p.depth + 1;
        break;
      }
      open.pop_back();
    }
  }
  return tree;
}

}  // namespace pico_tree::internal
