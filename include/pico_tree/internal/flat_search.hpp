#pragma once
//! \file flat_search.hpp
//! \brief Single-query traversals over a flat_tree on the host.
//! \details These serve the per-query members of kd_tree (custom visitors,
//! custom metrics, double precision): the batched members of the accelerated
//! instantiation never come here, they go to the HIP backend.
//!
//! nearest_search is the reference's near-first depth-first search with
//! incremental box distances (internal/kd_tree_search.hpp:24-113, after Arya &
//! Mount) turned into a loop over one LIFO of small records -- the same scheme
//! the gfx950 kernels run per lane:
//!
//!   state     : node, box distance `nbd`, per-axis offsets `off[dim]`
//!   pending   : {far child, axis, new_offset}   pushed when a branch is passed
//!   undo_off  : {axis, previous off[axis]}      pushed when a far child is entered
//!   undo_nbd  : {previous nbd}                  idem
//!
//! Popping a pending record recomputes nbd' = nbd - off[axis] + new_offset with
//! the state of the node that pushed it (the undo records below it have already
//! restored that state), applies the reference's test `visitor.max() >= nbd'`
//! (:99) and, if it passes, enters the far child.  Visit order, distance bits
//! and tie-breaking are therefore those of the recursive reference.

#include <cstdint>
#include <vector>

#include <algorithm>

#include "../core.hpp"
#include "../distance.hpp"
#include "flat_tree.hpp"

namespace pico_tree::internal {

template <typename Scalar_>
struct search_record {
  enum kind_t : std::uint32_t { pending = 0, undo_off = 1, undo_nbd = 2 };
  std::uint32_t kind;
  std::uint32_t node;  //!< pending: far child
  std::uint32_t axis;  //!< pending, undo_off
  Scalar_ value;       //!< pending: new_offset; undo_*: value to restore
};

template <typename Tree_, typename SpaceView_, typename Metric_, typename PointView_, typename Visitor_>
inline void nearest_search(
    Tree_ const& tree,
    SpaceView_ const& space,
    Metric_ const& metric,
    PointView_ const& query,
    Visitor_& visitor) {
  using scalar = typename Tree_::scalar_type;
  using record = search_record<scalar>;
  auto const* const nodes = tree.nodes.data();
  auto const* const indices = tree.indices.data();

  // Offsets: on the stack for small fixed dimensions, heap otherwise.
  constexpr size_t kInline = 8;
  scalar inline_off[kInline] = {};
  std::vector<scalar> heap_off;
  size_t const sdim = space.sdim();
  scalar* off = inline_off;
  if (sdim > kInline) {
    heap_off.assign(sdim, scalar(0));
    off = heap_off.data();
  }

  constexpr size_t kInlineRecords = 3 * 64;
  record inline_records[kInlineRecords];
  std::vector<record> heap_records;
  record* stack = inline_records;
  size_t capacity = kInlineRecords;
  size_t top = 0;
  auto push = [&](record const& r) {
    if (top == capacity) {  // rare: very deep tree
      if (stack == inline_records) heap_records.assign(stack, stack + top);
      heap_records.resize(capacity * 2);
      capacity *= 2;
      stack = heap_records.data();
    }
    stack[top++] = r;
  };

  std::uint32_t node = 0;
  scalar nbd = scalar(0);

  for (;;) {
    // Walk down to a leaf, always into the child nearer to the query.
    while (!nodes[node].is_leaf()) {
      auto const& b = nodes[node];
      size_t const axis = b.split_dim;
      scalar const v = query[axis];
      std::uint32_t near_child, far_child;
      scalar new_offset;
      if ((b.left_max + b.right_min - v - v) > 0) {
        near_child = node + 1;
        far_child = b.right;
        new_offset = metric(b.right_min - v);
      } else {
        near_child = b.right;
        far_child = node + 1;
        new_offset = metric(b.left_max - v);
      }
      push(record{record::pending, far_child, static_cast<std::uint32_t>(axis), new_offset});
      node = near_child;
    }

    // Measure the leaf's points in index order.
    {
      auto const& leaf = nodes[node];
      for (auto i = leaf.begin; i < leaf.end; ++i) {
        auto const idx = indices[i];
        visitor(idx, metric(query.begin(), query.end(), space[idx]));
      }
    }

    // Unwind to the next far child worth entering.
    for (;;) {
      if (top == 0) return;
      record const r = stack[--top];
      if (r.kind == record::undo_nbd) {
        nbd = r.value;
      } else if (r.kind == record::undo_off) {
        off[r.axis] = r.value;
      } else {
        scalar const old_offset = off[r.axis];
        scalar const far_nbd = nbd - old_offset + r.value;
        if (visitor.max() >= far_nbd) {
          push(record{record::undo_off, 0, r.axis, old_offset});
          push(record{record::undo_nbd, 0, 0, nbd});
          off[r.axis] = r.value;
          nbd = far_nbd;
          node = r.node;
          break;
        }
      }
    }
  }
}

//! Distance from x to the segment [min, max] of the real line (internal/segment.hpp:21-48).
template <typename Scalar_>
constexpr Scalar_ segment_distance(Scalar_ min, Scalar_ max, Scalar_ x, one_space_r1) {
  if (x < min) return min - x;
  if (x > max) return x - max;
  return Scalar_(0);
}

//! ... and to the arc from min to max of the unit circle [0, 1] / 0 ~ 1, which wraps through 0
//! when min > max (internal/segment.hpp:50-104).
template <typename Scalar_>
constexpr Scalar_ segment_distance(Scalar_ min, Scalar_ max, Scalar_ x, one_space_s1) {
  if (min <= max) {  // linear
    if (x < min || x > max) return std::min(s1_distance(x, min), s1_distance(x, max));
    return Scalar_(0);
  }
  if (x < max || x > min) return Scalar_(0);
  return std::min(s1_distance(x, min), s1_distance(x, max));
}

//! The reference's search for topological spaces (internal/kd_tree_search.hpp:115-229): both boxes
//! of a branch are measured on the split axis with the metric's notion of that axis (line or
//! circle: apply_dim_space), the nearer child is visited first (ties: the right one, `d1 < d2`),
//! the farther iff `visitor.max() >= node_box_distance` with the same incremental update as the
//! euclidean search.  Needs the four bounds per branch (flat_tree::outer_bounds).  Recursive: this
//! is a host-only path.
template <typename Tree_, typename SpaceView_, typename Metric_, typename PointView_, typename Visitor_>
class nearest_search_topological {
 public:
  using scalar = typename Tree_::scalar_type;
  nearest_search_topological(
      Tree_ const& tree, SpaceView_ const& space, Metric_ const& metric, PointView_ const& query, Visitor_& visitor)
      : tree_(tree), space_(space), metric_(metric), query_(query), visitor_(visitor), off_(space.sdim(), scalar(0)) {}

  void operator()() { descend(0, scalar(0)); }

 private:
  scalar box_distance(scalar min, scalar max, scalar v, int dim) const {
    scalar d{};
    metric_.apply_dim_space(dim, [&](auto one_space) { d = segment_distance(min, max, v, one_space); });
    return metric_(d);
  }

  void descend(std::uint32_t node, scalar node_box_distance) {
    auto const& nd = tree_.nodes[node];
    if (nd.is_leaf()) {
      for (auto i = nd.begin; i < nd.end; ++i) {
        auto const idx = tree_.indices[static_cast<size_t>(i)];
        visitor_(idx, metric_(query_.begin(), query_.end(), space_[idx]));
      }
      return;
    }
    size_t const axis = nd.split_dim;
    scalar const v = query_[axis];
    auto const& outer = tree_.outer_bounds[node];  // {left_min, right_max}
    scalar const d1 = box_distance(outer[0], nd.left_max, v, static_cast<int>(axis));
    scalar const d2 = box_distance(nd.right_min, outer[1], v, static_cast<int>(axis));
    std::uint32_t first, second;
    scalar new_offset;
    if (d1 < d2) {
      first = node + 1;
      second = nd.right;
      new_offset = d2;
    } else {
      first = nd.right;
      second = node + 1;
      new_offset = d1;
    }
    descend(first, node_box_distance);
    scalar const old_offset = off_[axis];
    node_box_distance = node_box_distance - old_offset + new_offset;
    if (visitor_.max() >= node_box_distance) {
      off_[axis] = new_offset;
      descend(second, node_box_distance);
      off_[axis] = old_offset;
    }
  }

  Tree_ const& tree_;
  SpaceView_ const& space_;
  Metric_ const& metric_;
  PointView_ const& query_;
  Visitor_& visitor_;
  std::vector<scalar> off_;
};

//! Which axes of a metric's space are circles (apply_dim_space of the topological metrics,
//! reference metric.hpp:216-219, :249-256); a euclidean metric has none.
struct no_circle_axes {
  constexpr bool operator()(size_t) const { return false; }
};
template <typename Metric_>
struct circle_axes_of {
  Metric_ const& metric;
  bool operator()(size_t axis) const {
    bool circle = false;
    metric.apply_dim_space(static_cast<int>(axis), [&](auto one_space) {
      circle = std::is_same_v<decltype(one_space), one_space_s1>;
    });
    return circle;
  }
};

//! All indices inside the closed box [qmin, qmax], in the reference's report
//! order (internal/kd_tree_search.hpp:238-381): a node whose running box is
//! fully inside the query is reported wholesale, a partially covered node is
//! descended, left before right.
//!
//! Topological_ (a tree over a topological space: flat_tree::outer_bounds): the query is the
//! reference's metric_box_map (box.hpp:300-376) -- on a circle axis (`is_circle(axis)`) an
//! interval with min > max runs through the seam 0 ~ 1 and contains x iff x >= min || x <= max
//! (segment_s1::contains, segment.hpp:61-75) -- and every axis takes the four-bound
//! intersection tests of kd_tree_search.hpp:311-327.
template <bool Topological_ = false, typename Tree_, typename SpaceView_, typename Index_,
          typename CircleAxes_ = no_circle_axes>
inline void box_search(
    Tree_ const& tree,
    SpaceView_ const& space,
    typename Tree_::scalar_type const* qmin,
    typename Tree_::scalar_type const* qmax,
    std::vector<Index_>& out,
    CircleAxes_ const& is_circle = CircleAxes_{}) {
  using scalar = typename Tree_::scalar_type;
  using box_type = typename Tree_::box_type;
  auto const* const nodes = tree.nodes.data();
  auto const* const indices = tree.indices.data();
  size_t const sdim = space.sdim();

  // The query box with the containment tests of box_map (euclidean) or metric_box_map.
  struct query_box {
    box_type b;
    std::vector<char> wraps;  // per axis: a circle axis whose interval runs through the seam
    scalar& min(size_t i) { return b.min(i); }
    scalar& max(size_t i) { return b.max(i); }
    scalar min(size_t i) const { return b.min(i); }
    scalar max(size_t i) const { return b.max(i); }
    bool contains(scalar const* p) const {
      if constexpr (!Topological_) {
        return b.contains(p);
      } else {
        for (size_t i = 0; i < wraps.size(); ++i) {
          scalar const x = p[i];
          if (wraps[i] ? !(x >= b.min(i) || x <= b.max(i)) : !(b.min(i) <= x && x <= b.max(i))) return false;
        }
        return true;
      }
    }
    bool contains(box_type const& x) const {
      if constexpr (!Topological_) {
        return b.contains(x);
      } else {
        for (size_t i = 0; i < wraps.size(); ++i) {
          if (wraps[i] ? !(x.min(i) >= b.min(i) || x.max(i) <= b.max(i))
                       : !(b.min(i) <= x.min(i) && x.max(i) <= b.max(i)))
            return false;
        }
        return true;
      }
    }
  };
  query_box query{box_type(sdim), std::vector<char>(Topological_ ? sdim : 0, 0)};
  for (size_t i = 0; i < sdim; ++i) {
    query.min(i) = qmin[i];
    query.max(i) = qmax[i];
    if constexpr (Topological_) query.wraps[i] = is_circle(i) && !(qmin[i] <= qmax[i]);
  }
  box_type box = tree.root_box;

  // Range of index positions covered by a subtree: its left-most leaf's begin
  // and right-most leaf's end.
  auto const first_pos = [&](std::uint32_t n) {
    while (!nodes[n].is_leaf()) n = n + 1;
    return nodes[n].begin;
  };
  auto const last_pos = [&](std::uint32_t n) {
    while (!nodes[n].is_leaf()) n = nodes[n].right;
    return nodes[n].end;
  };
  auto const report = [&](std::uint32_t n) {
    for (auto i = first_pos(n); i < last_pos(n); ++i) out.push_back(indices[i]);
  };

  // Explicit stack of {node, phase}; phase 0 = enter, 1 = left done, 2 = right done.
  struct frame {
    std::uint32_t node;
    std::uint32_t phase;
    scalar saved;
  };
  std::vector<frame> stack;
  stack.push_back(frame{0, 0, scalar(0)});
  while (!stack.empty()) {
    frame& f = stack.back();
    auto const& nd = nodes[f.node];
    if (nd.is_leaf()) {
      for (auto i = nd.begin; i < nd.end; ++i) {
        if (query.contains(space[indices[i]])) out.push_back(indices[i]);
      }
      stack.pop_back();
      continue;
    }
    size_t const axis = nd.split_dim;
    if (f.phase == 0) {
      f.phase = 1;
      f.saved = box.max(axis);
      box.max(axis) = nd.left_max;
      bool enter = query.min(axis) <= nd.left_max;  // intersects_left
      if constexpr (Topological_) enter = enter || query.max(axis) >= tree.outer_bounds[f.node][0];  // || max >= left_min
      if (query.contains(box)) {
        report(f.node + 1);
      } else if (enter) {
        stack.push_back(frame{f.node + 1, 0, scalar(0)});
      }
    } else if (f.phase == 1) {
      f.phase = 2;
      box.max(axis) = f.saved;
      f.saved = box.min(axis);
      box.min(axis) = nd.right_min;
      bool enter = query.max(axis) >= nd.right_min;  // intersects_right
      if constexpr (Topological_) enter = enter || query.min(axis) <= tree.outer_bounds[f.node][1];  // || min <= right_max
      if (query.contains(box)) {
        report(nd.right);
      } else if (enter) {
        std::uint32_t const r = nd.right;
        stack.push_back(frame{r, 0, scalar(0)});
      }
    } else {
      box.min(axis) = f.saved;
      stack.pop_back();
    }
  }
}

}  // namespace pico_tree::internal
