#pragma once
//! \file flat_tree.hpp
//! \brief The kd-tree as one flat array: build parameters, storage, builder.
//! \details The reference stores pointer-linked nodes in a chunk allocator
//! (internal/kd_tree_node.hpp:7-27, internal/memory.hpp:20-127).  Here the tree
//! is flat from the start: nodes live in one std::vector in depth-first
//! pre-order, the left child of a branch is the next element and the right
//! child is stored as an index.  That order is also the reference's on-disk
//! order (internal/kd_tree_data.hpp:109-135) and, after re-encoding, the layout
//! the HIP backend keeps in HBM.
//!
//! The builder reproduces the reference tree exactly (same split planes, same
//! index permutation, same tightened left_max / right_min):
//!   build_kd_tree_impl::create_node    internal/kd_tree_builder.hpp:352-396
//!   sliding midpoint / midpoint / median splitters   :144-280
//!   stop conditions                    :410-425
//!   start bounds                       :496-511
//! It calls std::partition / std::nth_element with the same predicates in the
//! same sequence, because their permutation decides the visit order inside a
//! leaf and therefore which of two equidistant points a query reports.

#include <algorithm>
#include <array>
#include <atomic>
#include <cassert>
#include <chrono>
#include <cstdint>
#include <future>
#include <exception>
#include <limits>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "../core.hpp"
#include "access.hpp"

namespace pico_tree {

// ---- build parameters (public names kept from the reference) ---------------

template <typename Derived_>
struct splitter_rule_t {
  Derived_ const& derived() const { return *static_cast<Derived_ const*>(this); }

 protected:
  constexpr explicit splitter_rule_t() = default;
  constexpr explicit splitter_rule_t(splitter_rule_t const&) = default;
  constexpr explicit splitter_rule_t(splitter_rule_t&&) = default;
};

//! Split on the median along the longest side of the node box.
struct median_max_side_t : splitter_rule_t<median_max_side_t> {
  constexpr explicit median_max_side_t() = default;
};
//! Split the node box in the middle of its longest side; may leave empty leaves.
struct midpoint_max_side_t : splitter_rule_t<midpoint_max_side_t> {
  constexpr explicit midpoint_max_side_t() = default;
};
//! Midpoint split that slides to the nearest point when one side would be empty.
struct sliding_midpoint_max_side_t : splitter_rule_t<sliding_midpoint_max_side_t> {
  constexpr explicit sliding_midpoint_max_side_t() = default;
};

inline constexpr median_max_side_t median_max_side{};
inline constexpr midpoint_max_side_t midpoint_max_side{};
inline constexpr sliding_midpoint_max_side_t sliding_midpoint_max_side{};

template <typename Derived_>
struct splitter_stop_condition_t {
  Derived_ const& derived() const { return *static_cast<Derived_ const*>(this); }

 protected:
  constexpr explicit splitter_stop_condition_t() = default;
  constexpr explicit splitter_stop_condition_t(splitter_stop_condition_t const&) = default;
  constexpr explicit splitter_stop_condition_t(splitter_stop_condition_t&&) = default;
};

//! A node with at most this many points becomes a leaf.
struct max_leaf_size_t : splitter_stop_condition_t<max_leaf_size_t> {
  constexpr max_leaf_size_t(size_t v) : value(v) { assert(value > 0); }
  size_t value;
};

namespace internal {

//! std::partition's result, computed by several threads.
//! \details libstdc++'s std::partition on bidirectional (and random access) iterators is the
//! two-pointer scheme: the k-th element from the left that fails the predicate is swapped with
//! the k-th element from the right that passes it, until the pointers meet at position
//! m = number of passing elements.  So the permutation is determined by two lists -- the
//! misplaced positions below m in increasing order and the misplaced positions from m on in
//! decreasing order -- which threads can build chunk by chunk and swap pairwise.  The result is
//! element-for-element what the serial call leaves behind (tests/test_cabi.py compares trees
//! built with 1 and many threads; the oracle pins them to the reference).
template <typename Index_, typename Pred_>
Index_* parallel_partition(Index_* begin, Index_* end, Pred_ pred, unsigned threads) {
  std::ptrdiff_t const n = end - begin;
  if (threads < 2 || n < (std::ptrdiff_t(1) << 16)) return std::partition(begin, end, pred);
  unsigned const t_count = threads > 64 ? 64 : threads;
  std::ptrdiff_t const per = (n + t_count - 1) / t_count;
  std::vector<unsigned char> pass(static_cast<size_t>(n));
  std::vector<std::ptrdiff_t> count(t_count + 1, 0);
  auto for_chunks = [&](auto&& body) {
    std::vector<std::future<void>> jobs;
    for (unsigned t = 1; t < t_count; ++t)
      jobs.push_back(std::async(std::launch::async, [&body, t] { body(t); }));
    body(0u);
    for (auto& j : jobs) j.get();
  };
  for_chunks([&](unsigned t) {
    std::ptrdiff_t const lo = std::min<std::ptrdiff_t>(n, per * t), hi = std::min<std::ptrdiff_t>(n, lo + per);
    std::ptrdiff_t c = 0;
    for (std::ptrdiff_t i = lo; i < hi; ++i) {
      bool const ok = pred(begin[i]);
      pass[static_cast<size_t>(i)] = ok ? 1 : 0;
      c += ok ? 1 : 0;
    }
    count[t + 1] = c;
  });
  for (unsigned t = 0; t < t_count; ++t) count[t + 1] += count[t];
  std::ptrdiff_t const m = count[t_count];
  if (m == 0 || m == n) return begin + m;  // nothing to move
  // Misplaced positions per chunk (increasing), then their global ranks.
  std::vector<std::vector<std::ptrdiff_t>> left(t_count), right(t_count);
  for_chunks([&](unsigned t) {
    std::ptrdiff_t const lo = std::min<std::ptrdiff_t>(n, per * t), hi = std::min<std::ptrdiff_t>(n, lo + per);
    for (std::ptrdiff_t i = lo; i < hi; ++i) {
      if (i < m) {
        if (!pass[static_cast<size_t>(i)]) left[t].push_back(i);
      } else if (pass[static_cast<size_t>(i)]) {
        right[t].push_back(i);
      }
    }
  });
  std::vector<std::ptrdiff_t> lbase(t_count + 1, 0), rbase(t_count + 1, 0);
  for (unsigned t = 0; t < t_count; ++t) {
    lbase[t + 1] = lbase[t] + static_cast<std::ptrdiff_t>(left[t].size());
    rbase[t + 1] = rbase[t] + static_cast<std::ptrdiff_t>(right[t].size());
  }
  std::ptrdiff_t const swaps = lbase[t_count];  // == rbase[t_count]
  std::vector<std::ptrdiff_t> rflat(static_cast<size_t>(swaps));
  for_chunks([&](unsigned t) {
    for (size_t k = 0; k < right[t].size(); ++k) rflat[static_cast<size_t>(rbase[t]) + k] = right[t][k];
  });
  // The k-th misplaced position from the left meets the k-th from the right (rflat read backwards).
  for_chunks([&](unsigned t) {
    for (size_t k = 0; k < left[t].size(); ++k) {
      std::ptrdiff_t const rank = lbase[t] + static_cast<std::ptrdiff_t>(k);
      std::swap(begin[left[t][k]], begin[rflat[static_cast<size_t>(swaps - 1 - rank)]]);
    }
  });
  return begin + m;
}

}  // namespace internal

//! A node at this depth becomes a leaf (depth 0 = the root).
struct max_leaf_depth_t : splitter_stop_condition_t<max_leaf_depth_t> {
  constexpr max_leaf_depth_t(size_t v) : value(v) {}
  size_t value;
};

template <typename Derived_>
struct splitter_start_bounds_t {
  Derived_ const& derived() const { return *static_cast<Derived_ const*>(this); }

 protected:
  constexpr explicit splitter_start_bounds_t() = default;
  constexpr explicit splitter_start_bounds_t(splitter_start_bounds_t const&) = default;
  constexpr explicit splitter_start_bounds_t(splitter_start_bounds_t&&) = default;
};

//! Start from the bounding box of the point set.
struct bounds_from_space_t : splitter_start_bounds_t<bounds_from_space_t> {
  constexpr explicit bounds_from_space_t() = default;
};
inline constexpr bounds_from_space_t bounds_from_space{};

//! Start from a caller-supplied box.
template <typename Point_>
struct bounds_t : splitter_start_bounds_t<bounds_t<Point_>> {
  constexpr explicit bounds_t(Point_ const& min, Point_ const& max)
      : min_(min), max_(max) {}
  constexpr Point_ const& min() const { return min_; }
  constexpr Point_ const& max() const { return max_; }

 private:
  Point_ min_;
  Point_ max_;
};

namespace internal {

inline constexpr std::uint32_t flat_leaf_tag = 0xFFFFFFFFu;

//! One node.  For <int, float> this is 16 bytes and layout-identical to the
//! C-ABI's ptk_node {a, b, right, split_dim} (include/ptk.h).
template <typename Index_, typename Scalar_>
struct flat_node {
  union {
    Scalar_ left_max;  //!< branch: max of the (tightened) left box on split_dim
    Index_ begin;      //!< leaf: first position in indices
  };
  union {
    Scalar_ right_min;  //!< branch: min of the (tightened) right box
    Index_ end;         //!< leaf: one past the last position in indices
  };
  std::uint32_t right;      //!< branch: index of the right child; leaf: flat_leaf_tag
  std::uint32_t split_dim;  //!< branch: split axis

  bool is_leaf() const { return right == flat_leaf_tag; }
};

//! Axis-aligned box with compile-time or run-time dimension: min[d], max[d].
template <typename Scalar_, size_t Dim_>
class aabb {
  using storage = std::conditional_t<
      Dim_ == dynamic_extent,
      std::vector<Scalar_>,
      std::array<Scalar_, (Dim_ == dynamic_extent ? 1 : 2 * Dim_)>>;

 public:
  explicit aabb(size_t d) : d_(d) {
    if constexpr (Dim_ == dynamic_extent) c_.resize(2 * d);
  }
  size_t size() const { return d_; }
  Scalar_* min() { return c_.data(); }
  Scalar_* max() { return c_.data() + d_; }
  Scalar_ const* min() const { return c_.data(); }
  Scalar_ const* max() const { return c_.data() + d_; }
  Scalar_& min(size_t i) { return c_[i]; }
  Scalar_& max(size_t i) { return c_[d_ + i]; }
  Scalar_ min(size_t i) const { return c_[i]; }
  Scalar_ max(size_t i) const { return c_[d_ + i]; }

  //! Inverted box that any fit() will shrink-wrap (box.hpp:54-59).
  void invert() {
    for (size_t i = 0; i < d_; ++i) {
      min(i) = std::numeric_limits<Scalar_>::max();
      max(i) = std::numeric_limits<Scalar_>::lowest();
    }
  }
  void fit(Scalar_ const* x) {
    for (size_t i = 0; i < d_; ++i) {
      if (x[i] < min(i)) min(i) = x[i];
      if (x[i] > max(i)) max(i) = x[i];
    }
  }
  void fit(aabb const& o) {
    for (size_t i = 0; i < d_; ++i) {
      if (o.min(i) < min(i)) min(i) = o.min(i);
      if (o.max(i) > max(i)) max(i) = o.max(i);
    }
  }
  //! First axis with the strictly greatest extent (box.hpp:71-82).
  void longest_side(size_t& axis, Scalar_& extent) const {
    extent = std::numeric_limits<Scalar_>::lowest();
    for (size_t i = 0; i < d_; ++i) {
      Scalar_ const delta = max(i) - min(i);
      if (delta > extent) {
        axis = i;
        extent = delta;
      }
    }
  }
  //! Closed containment test (box.hpp:31-47).
  bool contains(Scalar_ const* x) const {
    for (size_t i = 0; i < d_; ++i) {
      if (min(i) > x[i] || max(i) < x[i]) return false;
    }
    return true;
  }
  bool contains(aabb const& o) const { return contains(o.min()) && contains(o.max()); }

 private:
  size_t d_;
  storage c_;
};

//! The whole index structure.  Points are not stored (the space owns them).
template <typename Index_, typename Scalar_, size_t Dim_>
struct flat_tree {
  using index_type = Index_;
  using scalar_type = Scalar_;
  using node_type = flat_node<Index_, Scalar_>;
  using box_type = aabb<Scalar_, Dim_>;
  static constexpr size_t dim = Dim_;

  explicit flat_tree(size_t sdim) : root_box(sdim) {}

  std::vector<Index_> indices;   //!< leaf-ordered permutation of [0, n)
  box_type root_box;             //!< start bounds of the build
  std::vector<node_type> nodes;  //!< DFS pre-order; nodes[0] is the root
  std::uint32_t max_depth = 0;   //!< depth of the deepest node (root = 0)
  size_t leaf_count = 0;
  size_t max_leaf_points = 0;
  //! Optional (keep_outer_bounds): per node, for branches, {left_min, right_max} on the
  //! split axis -- with left_max / right_min these are the four bounds of the reference's
  //! kd_tree_node_topological (internal/kd_tree_node.hpp:56-67, :99-117), which the
  //! kd_forest's priority search needs.
  bool keep_outer_bounds = false;
  std::vector<std::array<Scalar_, 2>> outer_bounds;
  //! Per axis: the number of splits on that axis on the way from the root to a leaf, summed over the points
  //! (integers held in doubles: exact in any order of summation).  Divided by the number of points this is how finely
  //! a root-to-leaf path cuts each axis -- what a space-filling order of queries should spend its bits on.
  std::vector<double> axis_weight;
};

//! Builds a flat_tree over a space_view.
template <typename SpaceView_, typename Index_, typename Rule_, typename Stop_>
class flat_builder {
 public:
  using scalar_type = typename SpaceView_::scalar_type;
  static constexpr size_t dim = SpaceView_::dim;
  using tree_type = flat_tree<Index_, scalar_type, dim>;
  using box_type = typename tree_type::box_type;

  flat_builder(SpaceView_ const& space, size_t stop_value, tree_type& out)
      : space_(space), stop_(static_cast<Index_>(stop_value)), tree_(out) {}

  //! \param threads > 1: the two subtrees of the top levels are built concurrently (each into
  //! its own node array, spliced back in depth-first order).  The result does not depend on it:
  //! every std::partition / std::nth_element call sees the same range in the same state.
  void run(box_type const& start_bounds, unsigned threads = 1) {
    size_t const n = space_.size();
    assert(n > 0);
    tree_.indices.resize(n);
    for (size_t i = 0; i < n; ++i) tree_.indices[i] = static_cast<Index_>(i);
    tree_.root_box = start_bounds;
    tree_.nodes.clear();
    // A binary tree with L leaves has 2L - 1 nodes; reserve for ~n / (stop/2).
    tree_.nodes.reserve(std::is_same_v<Stop_, max_leaf_size_t>
                            ? 4 * n / static_cast<size_t>(stop_ > 0 ? stop_ : 1) + 16
                            : 1024);
    box_type work = start_bounds;
    index_base_ = tree_.indices.data();
    path_.assign(tree_.root_box.size(), 0);
    tree_.axis_weight.assign(tree_.root_box.size(), 0.0);
    task_pool pool;
    pool.idle.store(static_cast<int>(threads) - 1);  // this thread is the first worker
    pool.threads = threads;
    pool.n_total = n;
    pool_ = threads > 1 ? &pool : nullptr;
    grow(0, index_base_, index_base_ + n, work);
  }

  //! The subtree over [begin, end) of a permutation that starts at `base`, for a node at `depth` whose (loose)
  //! box is `box`: exactly what grow() does when the recursion arrives there.  `box` leaves tightened.  Nodes
  //! are numbered from 0 in this builder's tree (see append()).  Serial.
  void run_range(std::uint32_t depth, Index_* base, Index_* begin, Index_* end, box_type& box,
                 std::vector<std::uint32_t> const& splits_above) {
    index_base_ = base;
    pool_ = nullptr;
    path_ = splits_above;  // splits per axis between the root and this node
    tree_.axis_weight.assign(tree_.root_box.size(), 0.0);
    grow(depth, begin, end, box);
  }

  //! Appends a separately built subtree in depth-first position (its links are relative to its first node).
  void append(tree_type const& sub) { splice(sub); }

 private:
  bool stops(std::uint32_t depth, Index_ const* begin, Index_ const* end) const {
    if constexpr (std::is_same_v<Stop_, max_leaf_size_t>) {
      return (end - begin) <= stop_;
    } else {
      return (static_cast<Index_>(depth) == stop_) || ((end - begin) <= 1);
    }
  }

  void split(
      Index_* begin,
      Index_* end,
      box_type const& box,
      Index_*& cut,
      size_t& axis,
      scalar_type& plane,
      unsigned threads = 1) const {
    scalar_type extent;
    box.longest_side(axis, extent);
    size_t const a = axis;
    auto const by_axis = [this, a](Index_ const i, Index_ const j) -> bool {
      return space_[i][a] < space_[j][a];
    };

    if constexpr (std::is_same_v<Rule_, median_max_side_t>) {
      cut = begin + (end - begin) / 2;
      std::nth_element(begin, cut, end, by_axis);
      plane = space_[*cut][a];
    } else {
      if constexpr (std::is_same_v<Rule_, midpoint_max_side_t>) {
        plane = extent * scalar_type(0.5) + box.min(a);
      } else {
        plane = extent / scalar_type(2.0) + box.min(a);
      }
      scalar_type const p = plane;
      cut = parallel_partition(begin, end, [this, a, p](Index_ const i) -> bool {
        return space_[i][a] < p;
      }, threads);
      if constexpr (std::is_same_v<Rule_, sliding_midpoint_max_side_t>) {
        if (cut == end) {  // nothing on the right: slide the largest point over
          --cut;
          std::nth_element(begin, cut, end, by_axis);
          plane = space_[*cut][a];
        } else if (cut == begin) {  // nothing on the left
          ++cut;
          std::nth_element(begin, cut, end, by_axis);
          plane = space_[*cut][a];
        }
      }
    }
  }

  //! Workers of a threaded build.  Tasks are spawned by SIZE, wherever both children of a node are
  //! worth one and a worker is idle -- a fixed number of top levels is not enough: on a LiDAR
  //! cloud the sliding midpoint splits 77 / 23 again and again, and one of 64 equal-depth subtrees
  //! held a fifth of the points.  A thread that waits for a child it spawned counts as idle.
  struct task_pool {
    std::atomic<int> idle{0};
    unsigned threads = 1;
    size_t n_total = 0;
    bool try_acquire() {
      int v = idle.load(std::memory_order_relaxed);
      while (v > 0) {
        if (idle.compare_exchange_weak(v, v - 1, std::memory_order_acq_rel)) return true;
      }
      return false;
    }
    void release() { idle.fetch_add(1, std::memory_order_acq_rel); }
    void reclaim() { idle.fetch_sub(1, std::memory_order_acq_rel); }  // may dip below zero for a moment
  };

  //! Ranges below this many points are not worth a task.
  static constexpr std::ptrdiff_t kParallelMin = 20000;
  static constexpr std::uint32_t kMaxBuildDepth = 8192;

  //! Appends the nodes of a separately built subtree; its right-child links are relative to its
  //! own first node.
  void splice(tree_type const& sub) {
    std::uint32_t const base = static_cast<std::uint32_t>(tree_.nodes.size());
    size_t const count = sub.nodes.size();
    tree_.nodes.resize(base + count);
    auto* dst = tree_.nodes.data() + base;
    auto const* src = sub.nodes.data();
    for (size_t i = 0; i < count; ++i) {
      auto nd = src[i];
      if (nd.right != flat_leaf_tag) nd.right += base;
      dst[i] = nd;
    }
    if (tree_.keep_outer_bounds)
      tree_.outer_bounds.insert(tree_.outer_bounds.end(), sub.outer_bounds.begin(), sub.outer_bounds.end());
    tree_.max_depth = std::max(tree_.max_depth, sub.max_depth);
    tree_.leaf_count += sub.leaf_count;
    tree_.max_leaf_points = std::max(tree_.max_leaf_points, sub.max_leaf_points);
    for (size_t a = 0; a < sub.axis_weight.size() && a < tree_.axis_weight.size(); ++a) tree_.axis_weight[a] += sub.axis_weight[a];
  }

  std::uint32_t grow(
      std::uint32_t depth, Index_* begin, Index_* end, box_type& box) {
    // Degenerate input (thousands of identical points with a small leaf size) makes the sliding
    // midpoint peel off one point per level; the reference recurses until its stack overflows.
    if (depth > kMaxBuildDepth)
      throw std::length_error("kd_tree: the tree is deeper than 8192 levels (degenerate point set?)");
    std::uint32_t const self = static_cast<std::uint32_t>(tree_.nodes.size());
    tree_.nodes.emplace_back();
    if (tree_.keep_outer_bounds) tree_.outer_bounds.push_back({scalar_type(0), scalar_type(0)});
    if (depth > tree_.max_depth) tree_.max_depth = depth;

    if (stops(depth, begin, end)) {
      auto& leaf = tree_.nodes[self];
      leaf.begin = static_cast<Index_>(begin - index_base_);
      leaf.end = static_cast<Index_>(end - index_base_);
      leaf.right = flat_leaf_tag;
      leaf.split_dim = 0;
      ++tree_.leaf_count;
      tree_.max_leaf_points =
          std::max(tree_.max_leaf_points, static_cast<size_t>(end - begin));
      for (size_t a = 0; a < path_.size(); ++a) tree_.axis_weight[a] += static_cast<double>(path_[a]) * static_cast<double>(end - begin);
      // Shrink the box to the leaf's points; an empty leaf (midpoint rule only)
      // keeps the box it was given.
      if (begin < end || !std::is_same_v<Rule_, midpoint_max_side_t>) {
        box.invert();
        for (Index_* it = begin; it < end; ++it) box.fit(space_[*it]);
      }
      return self;
    }

    Index_* cut = begin;
    size_t axis = 0;
    scalar_type plane = scalar_type(0);
    // A node's partition gets the share of the workers its share of the points is worth.
    unsigned split_threads = 1;
    if (pool_ != nullptr) {
      size_t const share = static_cast<size_t>(end - begin) * pool_->threads / pool_->n_total;
      split_threads = share > 64 ? 64u : (share < 1 ? 1u : static_cast<unsigned>(share));
    }
    split(begin, end, box, cut, axis, plane, split_threads);

    box_type right = box;
    box.max(axis) = plane;    // `box` now bounds the left child
    right.min(axis) = plane;
    ++path_[axis];

    std::uint32_t r;
    if (pool_ != nullptr && (cut - begin) >= kParallelMin && (end - cut) >= kParallelMin && pool_->try_acquire()) {
      // Left subtree in another thread, right subtree here, each into its own tree object.
      size_t const sdim = tree_.root_box.size();
      tree_type lt(sdim), rt(sdim);
      lt.keep_outer_bounds = rt.keep_outer_bounds = tree_.keep_outer_bounds;
      flat_builder lb(space_, static_cast<size_t>(stop_), lt), rb(space_, static_cast<size_t>(stop_), rt);
      lb.index_base_ = rb.index_base_ = index_base_;
      lb.pool_ = rb.pool_ = pool_;
      lb.path_ = rb.path_ = path_;
      lt.axis_weight.assign(sdim, 0.0);
      rt.axis_weight.assign(sdim, 0.0);
      task_pool* const pool = pool_;
      auto left_done = std::async(std::launch::async, [&lb, &box, pool, depth, begin, cut] {
        struct on_exit {  // the worker goes idle again, also when the subtree throws
          task_pool* p;
          ~on_exit() { p->release(); }
        } guard{pool};
        lb.grow(depth + 1, begin, cut, box);
      });
      try {
        rb.grow(depth + 1, cut, end, right);
      } catch (...) {
        left_done.wait();
        throw;
      }
      pool->release();  // waiting is not working
      try {
        left_done.get();
      } catch (...) {
        pool->reclaim();
        throw;
      }
      pool->reclaim();
      splice(lt);  // lands at self + 1
      r = static_cast<std::uint32_t>(tree_.nodes.size());
      splice(rt);
    } else {
      grow(depth + 1, begin, cut, box);  // lands at self + 1
      r = grow(depth + 1, cut, end, right);
    }

    --path_[axis];
    auto& branch = tree_.nodes[self];  // taken after the recursion: vector may grow
    branch.left_max = box.max(axis);   // both tightened by the children
    branch.right_min = right.min(axis);
    branch.right = r;
    branch.split_dim = static_cast<std::uint32_t>(axis);
    if (tree_.keep_outer_bounds) tree_.outer_bounds[self] = {box.min(axis), right.max(axis)};

    box.fit(right);  // parent box = union of the children
    return self;
  }

  SpaceView_ const& space_;
  Index_ stop_;
  tree_type& tree_;
  Index_* index_base_ = nullptr;  //!< first element of the (shared) index permutation
  std::vector<std::uint32_t> path_;  //!< splits per axis between the root and the node grow() is at
  task_pool* pool_ = nullptr;     //!< shared by the builders of one threaded build
};

//! Entry point: build with the given parameters.
template <typename Index_, typename SpaceView_, typename Stop_, typename Bounds_, typename Rule_>
flat_tree<Index_, typename SpaceView_::scalar_type, SpaceView_::dim> build_flat_tree(
    SpaceView_ const& space,
    splitter_stop_condition_t<Stop_> const& stop,
    splitter_start_bounds_t<Bounds_> const& bounds,
    splitter_rule_t<Rule_> const&,
    bool keep_outer_bounds = false,
    unsigned threads = 1) {
  using scalar_type = typename SpaceView_::scalar_type;
  using tree_type = flat_tree<Index_, scalar_type, SpaceView_::dim>;
  using box_type = typename tree_type::box_type;

  size_t const sdim = space.sdim();
  box_type start(sdim);
  start.invert();
  if constexpr (std::is_same_v<Bounds_, bounds_from_space_t>) {
    size_t const n = space.size();
    if (threads > 1 && n >= (size_t(1) << 18)) {  // min / max are exact: any grouping gives the same box
      unsigned const t_count = threads > 64 ? 64 : threads;
      std::vector<box_type> part(t_count, box_type(sdim));
      std::vector<std::future<void>> jobs;
      auto body = [&](unsigned t) {
        part[t].invert();
        size_t const per = (n + t_count - 1) / t_count, lo = std::min(n, per * t), hi = std::min(n, lo + per);
        for (size_t i = lo; i < hi; ++i) part[t].fit(space[i]);
      };
      for (unsigned t = 1; t < t_count; ++t) jobs.push_back(std::async(std::launch::async, [&body, t] { body(t); }));
      body(0u);
      for (auto& j : jobs) j.get();
      for (unsigned t = 0; t < t_count; ++t) start.fit(part[t]);
    } else {
      for (size_t i = 0; i < n; ++i) start.fit(space[i]);
    }
  } else {
    using bound_point = std::decay_t<decltype(bounds.derived().min())>;
    start.fit(point_view<bound_point>(bounds.derived().min()).data());
    start.fit(point_view<bound_point>(bounds.derived().max()).data());
  }

  tree_type tree(sdim);
  tree.keep_outer_bounds = keep_outer_bounds;
  flat_builder<SpaceView_, Index_, Rule_, Stop_>(space, stop.derived().value, tree).run(start, threads);
  return tree;
}

//! A branch of the top of a tree whose partitions were made elsewhere (by the device: ptk_build.hpp): the
//! split the builder would have chosen for its box, and where std::partition left the cut.
template <typename Scalar_>
struct top_branch {
  std::uint32_t axis;
  Scalar_ plane;
  std::int32_t left;   //!< >= 0: top_branch; < 0: ~(index into the frontier ranges)
  std::int32_t right;
};

//! One node of a level of the top of the tree, for the partitioner of split_top_levels(): the points of
//! [begin, end) with coordinate `axis` below `plane` go first, as std::partition leaves them; `cut` (out) is the
//! position of the first point of the second group.
template <typename Scalar_>
struct top_segment {
  size_t begin, end;
  std::uint32_t axis;
  Scalar_ plane;
  size_t cut;
};

//! The top of a sliding-midpoint tree, level by level: every node of more than `threshold` points (>= the leaf
//! size) is split the way flat_builder::split() splits it -- longest side of the box handed down, plane at its
//! middle -- with the partitions of one level made together by `partition_level(std::vector<top_segment>&)`
//! (false = it failed).  When a partition leaves one side empty the rule slides the plane to the nearest point:
//! `slide(top_segment&, nth)` has to leave [begin, end) as std::nth_element(begin, begin + nth, end) by the
//! coordinate of the axis leaves it and set `plane` to the coordinate of the point at begin + nth (false = it
//! failed).  Returns false, leaving the caller to build the whole tree the ordinary way, when a callback fails.
//! Output: see build_flat_tree_below().
template <typename Scalar_, size_t Dim_, typename PartitionLevel_, typename Slide_>
bool split_top_levels(
    aabb<Scalar_, Dim_> const& root_box,
    size_t n,
    size_t threshold,
    PartitionLevel_&& partition_level,
    Slide_&& slide,
    std::vector<top_branch<Scalar_>>& top,
    std::vector<std::pair<size_t, size_t>>& frontier) {
  using box_type = aabb<Scalar_, Dim_>;
  struct open_node {
    std::int32_t id;  // its top_branch
    size_t begin, end;
    box_type box;
  };
  top.clear();
  frontier.clear();
  if (n <= threshold) {
    frontier.emplace_back(size_t(0), n);
    return true;
  }
  std::vector<open_node> level;
  top.emplace_back();
  level.push_back(open_node{0, 0, n, root_box});
  std::vector<top_segment<Scalar_>> segments;
  while (!level.empty()) {
    segments.clear();
    for (auto const& node : level) {
      size_t axis = 0;
      Scalar_ extent;
      node.box.longest_side(axis, extent);
      Scalar_ const plane = extent / Scalar_(2.0) + node.box.min(axis);  // flat_builder::split(), sliding midpoint
      segments.push_back(top_segment<Scalar_>{node.begin, node.end, static_cast<std::uint32_t>(axis), plane, node.begin});
    }
    if (!partition_level(segments)) return false;
    std::vector<open_node> next;
    for (size_t i = 0; i < level.size(); ++i) {
      auto& seg = segments[i];
      if (seg.cut >= seg.end) {  // nothing on the right: slide the largest point over (flat_builder::split())
        if (!slide(seg, seg.end - seg.begin - 1)) return false;
        seg.cut = seg.end - 1;
      } else if (seg.cut <= seg.begin) {  // nothing on the left
        if (!slide(seg, size_t(1))) return false;
        seg.cut = seg.begin + 1;
      }
      box_type left = level[i].box, right = level[i].box;
      left.max(seg.axis) = seg.plane;
      right.min(seg.axis) = seg.plane;
      auto child = [&](size_t begin, size_t end, box_type const& box) -> std::int32_t {
        if (end - begin > threshold) {
          std::int32_t const id = static_cast<std::int32_t>(top.size());
          top.emplace_back();
          next.push_back(open_node{id, begin, end, box});
          return id;
        }
        frontier.emplace_back(begin, end);
        return ~static_cast<std::int32_t>(frontier.size() - 1);
      };
      std::int32_t const l = child(seg.begin, seg.cut, left);
      std::int32_t const r = child(seg.cut, seg.end, right);
      auto& branch = top[static_cast<size_t>(level[i].id)];
      branch.axis = seg.axis;
      branch.plane = seg.plane;
      branch.left = l;
      branch.right = r;
    }
    level.swap(next);
  }
  return true;
}

//! Finishes a sliding-midpoint / midpoint build whose top levels are given: `indices` already holds the
//! permutation every std::partition of those levels leaves behind, `top` the branches (top[0] = the root; empty:
//! the root is frontier range 0) and `frontier` the [begin, end) ranges the builder has not split yet.  The
//! subtrees below the frontier are built by `threads` workers, each exactly as grow() builds it, and the whole
//! is spliced in depth-first order: the result is the tree build_flat_tree() returns for the same input.
template <typename Index_, typename SpaceView_, typename Stop_, typename Rule_>
flat_tree<Index_, typename SpaceView_::scalar_type, SpaceView_::dim> build_flat_tree_below(
    SpaceView_ const& space,
    splitter_stop_condition_t<Stop_> const& stop,
    splitter_rule_t<Rule_> const&,
    typename flat_tree<Index_, typename SpaceView_::scalar_type, SpaceView_::dim>::box_type const& root_box,
    std::vector<Index_>&& indices,
    std::vector<top_branch<typename SpaceView_::scalar_type>> const& top,
    std::vector<std::pair<size_t, size_t>> const& frontier,
    bool keep_outer_bounds,
    unsigned threads,
    double* phase_ms = nullptr) {
  using scalar_type = typename SpaceView_::scalar_type;
  using tree_type = flat_tree<Index_, scalar_type, SpaceView_::dim>;
  using box_type = typename tree_type::box_type;
  using builder_type = flat_builder<SpaceView_, Index_, Rule_, Stop_>;
  size_t const sdim = space.sdim();
  auto const t_begin = std::chrono::steady_clock::now();

  tree_type tree(sdim);
  tree.keep_outer_bounds = keep_outer_bounds;
  tree.indices = std::move(indices);
  tree.root_box = root_box;

  // The depth and the (loose) box each frontier range is reached with.
  struct start {
    std::uint32_t depth;
    box_type box;
    std::vector<std::uint32_t> splits;  // per axis, between the root and the range
  };
  std::vector<start> starts(frontier.size(), start{0, box_type(sdim), {}});
  {
    struct frame {
      std::int32_t id;
      std::uint32_t depth;
      box_type box;
      std::vector<std::uint32_t> splits;
    };
    std::vector<frame> todo;
    todo.push_back(frame{top.empty() ? ~std::int32_t(0) : 0, 0, root_box, std::vector<std::uint32_t>(sdim, 0)});
    while (!todo.empty()) {
      frame f = std::move(todo.back());
      todo.pop_back();
      if (f.id < 0) {
        starts[static_cast<size_t>(~f.id)] = start{f.depth, f.box, f.splits};
        continue;
      }
      auto const& b = top[static_cast<size_t>(f.id)];
      box_type right = f.box;
      f.box.max(b.axis) = b.plane;
      right.min(b.axis) = b.plane;
      ++f.splits[b.axis];
      todo.push_back(frame{b.right, f.depth + 1, right, f.splits});
      todo.push_back(frame{b.left, f.depth + 1, f.box, f.splits});
    }
  }

  // The subtrees, largest first, by a pool of workers.
  std::vector<tree_type> subs(frontier.size(), tree_type(sdim));
  std::vector<size_t> order(frontier.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) {
    return frontier[a].second - frontier[a].first > frontier[b].second - frontier[b].first;
  });
  std::atomic<size_t> next{0};
  std::exception_ptr failure;
  std::mutex failure_lock;
  Index_* const base = tree.indices.data();
  auto work = [&] {
    for (;;) {
      size_t const k = next.fetch_add(1);
      if (k >= order.size()) return;
      size_t const i = order[k];
      try {
        subs[i].keep_outer_bounds = keep_outer_bounds;
        subs[i].nodes.reserve(4 * (frontier[i].second - frontier[i].first) / (stop.derived().value > 0 ? stop.derived().value : 1) + 16);
        builder_type(space, stop.derived().value, subs[i])
            .run_range(starts[i].depth, base, base + frontier[i].first, base + frontier[i].second, starts[i].box,
                       starts[i].splits);
      } catch (...) {
        std::lock_guard<std::mutex> hold(failure_lock);
        if (!failure) failure = std::current_exception();
      }
    }
  };
  {
    unsigned const workers = std::max(1u, std::min<unsigned>(threads, static_cast<unsigned>(order.size())));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < workers; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
  }
  if (failure) std::rethrow_exception(failure);
  auto const t_built = std::chrono::steady_clock::now();
  if (phase_ms != nullptr) phase_ms[0] = std::chrono::duration<double, std::milli>(t_built - t_begin).count();

  // Where everything goes in depth-first order: a walk over the top gives every branch and every subtree its
  // first node; the subtrees are then copied into place by the workers (right-child links shifted by the base, as
  // flat_builder::splice() does) and the top branches filled in on the way back up, boxes tightening as in grow().
  std::vector<std::uint32_t> sub_base(frontier.size(), 0), top_self(top.size(), 0);
  {
    std::uint32_t cursor = 0;
    std::vector<std::int32_t> todo;
    todo.push_back(top.empty() ? ~std::int32_t(0) : 0);
    while (!todo.empty()) {  // pre-order: node, left subtree, right subtree
      std::int32_t const id = todo.back();
      todo.pop_back();
      if (id < 0) {
        sub_base[static_cast<size_t>(~id)] = cursor;
        cursor += static_cast<std::uint32_t>(subs[static_cast<size_t>(~id)].nodes.size());
      } else {
        top_self[static_cast<size_t>(id)] = cursor++;
        todo.push_back(top[static_cast<size_t>(id)].right);
        todo.push_back(top[static_cast<size_t>(id)].left);
      }
    }
    tree.nodes.resize(cursor);
    if (keep_outer_bounds) tree.outer_bounds.resize(cursor);
  }
  next.store(0);
  auto copy = [&] {
    for (;;) {
      size_t const k = next.fetch_add(1);
      if (k >= order.size()) return;
      size_t const i = order[k];
      std::uint32_t const base_node = sub_base[i];
      auto const& sub = subs[i];
      auto* dst = tree.nodes.data() + base_node;
      for (size_t j = 0; j < sub.nodes.size(); ++j) {
        auto nd = sub.nodes[j];
        if (nd.right != flat_leaf_tag) nd.right += base_node;
        dst[j] = nd;
      }
      if (keep_outer_bounds) std::copy(sub.outer_bounds.begin(), sub.outer_bounds.end(), tree.outer_bounds.begin() + base_node);
    }
  };
  {
    unsigned const workers = std::max(1u, std::min<unsigned>(threads, static_cast<unsigned>(order.size())));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < workers; ++t) pool.emplace_back(copy);
    copy();
    for (auto& t : pool) t.join();
  }
  tree.axis_weight.assign(sdim, 0.0);
  for (auto const& sub : subs) {
    tree.max_depth = std::max(tree.max_depth, sub.max_depth);
    tree.leaf_count += sub.leaf_count;
    tree.max_leaf_points = std::max(tree.max_leaf_points, sub.max_leaf_points);
    for (size_t a = 0; a < sdim; ++a) tree.axis_weight[a] += sub.axis_weight[a];
  }
  struct placer {
    tree_type& tree;
    std::vector<top_branch<scalar_type>> const& top;
    std::vector<start>& starts;
    std::vector<std::uint32_t> const& sub_base;
    std::vector<std::uint32_t> const& top_self;
    std::uint32_t place(std::int32_t id, std::uint32_t depth, box_type& box) {
      if (id < 0) {
        box = starts[static_cast<size_t>(~id)].box;  // (tightened by run_range())
        return sub_base[static_cast<size_t>(~id)];
      }
      auto const& b = top[static_cast<size_t>(id)];
      std::uint32_t const self = top_self[static_cast<size_t>(id)];
      if (depth > tree.max_depth) tree.max_depth = depth;
      box_type right = box;
      box.max(b.axis) = b.plane;
      right.min(b.axis) = b.plane;
      place(b.left, depth + 1, box);
      std::uint32_t const r = place(b.right, depth + 1, right);
      auto& branch = tree.nodes[self];
      branch.left_max = box.max(b.axis);
      branch.right_min = right.min(b.axis);
      branch.right = r;
      branch.split_dim = b.axis;
      if (tree.keep_outer_bounds) tree.outer_bounds[self] = {box.min(b.axis), right.max(b.axis)};
      box.fit(right);
      return self;
    }
  } p{tree, top, starts, sub_base, top_self};
  box_type work_box = root_box;
  p.place(top.empty() ? ~std::int32_t(0) : 0, 0, work_box);
  if (phase_ms != nullptr) phase_ms[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_built).count();
  return tree;
}

}  // namespace internal
}  // namespace pico_tree
