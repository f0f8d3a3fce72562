#pragma once
//! \file kd_tree.hpp
//! \brief pico_tree::kd_tree -- the public search structure.
//! \details Source-compatible with the reference class
//! (/root/reference/src/pico_tree/pico_tree/kd_tree.hpp:19-435): same template
//! parameters, constructor, per-query members, tags, CTAD guide and
//! make_kd_tree.  Two things differ underneath:
//!
//!  1. The tree is a flat depth-first array (internal/flat_tree.hpp) instead of
//!     pointer-linked nodes.
//!  2. A family of BATCHED members takes a whole query space at once -- the
//!     shape of the reference's only batched caller, the OpenMP loops of its
//!     Python binding (src/pyco_tree/pico_tree/_pyco_tree/kd_tree.hpp:117-268).
//!     For kd_tree<Space, metric_l2_squared | metric_l1 | metric_lpinf, int> over
//!     float they run on the MI355X through the C ABI of include/ptk.h (-lptk).  They have
//!     no host fallback: without a usable backend they throw.
//!
//! The per-query members (search_nn(x, nn), search_knn(x, k, knn), custom
//! visitors, other metrics, double) run on the host, as in the reference.

#include <algorithm>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <type_traits>
#include <vector>

#include "core.hpp"
#include "internal/access.hpp"
#include "internal/backend.hpp"
#include "internal/flat_search.hpp"
#include "internal/flat_tree.hpp"
#include "internal/stream.hpp"
#include "internal/visitors.hpp"
#include "metric.hpp"

namespace pico_tree {

//! Off by default: a batched search the device refuses for a valid tree (PTK_ERR_UNSUPPORTED) throws.  With
//! allow_host_loop(true) such a call is served by the loop of per-query host members the reference runs for every call
//! (_pyco_tree/kd_tree.hpp:128), after one message on stderr.  Process-wide.
inline void allow_host_loop(bool on) { internal::host_loop_flag().store(on); }

template <typename Space_, typename Metric_ = metric_l2_squared, typename Index_ = int>
class kd_tree {
  static_assert(
      std::is_same_v<std::remove_cv_t<Space_>, Space_>,
      "SPACE_TYPE_MUST_BE_NON-CONST_NON-VOLATILE");
  //! Topological metrics (metric_so2, metric_se2_squared): the tree keeps four bounds per branch
  //! (the reference's kd_tree_node_topological) and the per-query members run
  //! internal::nearest_search_topological on the host.
  static constexpr bool topological =
      !std::is_same_v<typename Metric_::space_category, euclidean_space_tag>;

  using plain_space_type = internal::unwrap_ref_t<Space_>;
  using space_view_type = internal::space_view<plain_space_type>;

 public:
  using size_type = size_t;
  using index_type = Index_;
  using scalar_type = typename space_view_type::scalar_type;
  static constexpr size_type dim = space_view_type::dim;
  using space_type = Space_;
  using metric_type = Metric_;
  using neighbor_type = neighbor<index_type, scalar_type>;

 private:
  using tree_type = internal::flat_tree<index_type, scalar_type, dim>;
  static constexpr bool accelerated =
      internal::is_accelerated_v<Metric_, scalar_type, Index_>;
  using api = internal::ptk_api<std::conditional_t<accelerated, scalar_type, float>>;

 public:
  //! Index positions of one leaf: a [begin, end) range into the tree's indices.
  class leaf_range_type {
   public:
    using iterator_type = typename std::vector<index_type>::const_iterator;
    leaf_range_type(iterator_type b, iterator_type e) : b_(b), e_(e) {}
    iterator_type begin() const { return b_; }
    iterator_type end() const { return e_; }

   private:
    iterator_type b_;
    iterator_type e_;
  };

  //! Builds the tree.  \p space is taken by value: move it in, or use
  //! std::reference_wrapper<Space> as the space type to borrow it.
  template <
      typename Stop_,
      typename Bounds_ = bounds_from_space_t,
      typename Rule_ = sliding_midpoint_max_side_t>
  kd_tree(
      space_type space,
      splitter_stop_condition_t<Stop_> const& stop_condition,
      splitter_start_bounds_t<Bounds_> const& start_bounds = Bounds_{},
      splitter_rule_t<Rule_> const& rule = Rule_{})
      : space_(std::move(space)),
        metric_(),
        tree_(internal::build_flat_tree<index_type>(
            view(), stop_condition, start_bounds, rule, topological)) {}

  kd_tree(kd_tree const&) = delete;
  kd_tree(kd_tree&&) = default;
  kd_tree& operator=(kd_tree const&) = delete;
  kd_tree& operator=(kd_tree&&) = default;

  // ---- per-query searches (host) -------------------------------------------

  //! Generic traversal: \p visitor sees every measured point as
  //! visitor(index, distance) and prunes with visitor.max().
  template <typename P_, typename V_>
  inline void search_nearest(P_ const& x, V_& visitor) const {
    using query_view = internal::point_view<P_>;
    static_assert(
        std::is_same_v<scalar_type, typename query_view::scalar_type>,
        "POINT_AND_TREE_SCALAR_TYPES_DIFFER");
    static_assert(
        dim == query_view::dim || dim == dynamic_extent ||
            query_view::dim == dynamic_extent,
        "POINT_AND_TREE_DIMS_DIFFER");
    query_view q(x);
    space_view_type s = view();
    if constexpr (topological) {
      internal::nearest_search_topological<tree_type, space_view_type, metric_type, query_view, V_>(
          tree_, s, metric_, q, visitor)();
    } else {
      internal::nearest_search(tree_, s, metric_, q, visitor);
    }
  }

  template <typename P_>
  inline void search_nn(P_ const& x, neighbor_type& nn) const {
    internal::nn_visitor<neighbor_type> v(nn);
    search_nearest(x, v);
  }

  //! Approximate nn: may return a neighbour up to a factor \p e (in metric
  //! units) farther than the true one; the reported distance is scaled by 1/e.
  template <typename P_>
  inline void search_nn(P_ const& x, scalar_type const e, neighbor_type& nn) const {
    internal::nn_visitor<neighbor_type, true> v(nn, e);
    search_nearest(x, v);
  }

  template <typename P_, typename RandomAccessIterator_>
  inline void search_knn(
      P_ const& x, RandomAccessIterator_ begin, RandomAccessIterator_ end) const {
    static_assert(
        std::is_same_v<
            typename std::iterator_traits<RandomAccessIterator_>::value_type,
            neighbor_type>,
        "ITERATOR_VALUE_TYPE_DOES_NOT_EQUAL_NEIGHBOR_TYPE");
    internal::knn_visitor<RandomAccessIterator_> v(begin, end);
    search_nearest(x, v);
  }

  //! \p knn is resized to min(k, number of points).
  template <typename P_>
  inline void search_knn(
      P_ const& x, size_type const k, std::vector<neighbor_type>& knn) const {
    knn.resize(std::min(k, view().size()));
    search_knn(x, knn.begin(), knn.end());
  }

  template <typename P_, typename RandomAccessIterator_>
  inline void search_knn(
      P_ const& x,
      scalar_type const e,
      RandomAccessIterator_ begin,
      RandomAccessIterator_ end) const {
    static_assert(
        std::is_same_v<
            typename std::iterator_traits<RandomAccessIterator_>::value_type,
            neighbor_type>,
        "ITERATOR_VALUE_TYPE_DOES_NOT_EQUAL_NEIGHBOR_TYPE");
    internal::knn_visitor<RandomAccessIterator_, true> v(begin, end, e);
    search_nearest(x, v);
  }

  template <typename P_>
  inline void search_knn(
      P_ const& x,
      size_type const k,
      scalar_type const e,
      std::vector<neighbor_type>& knn) const {
    knn.resize(std::min(k, view().size()));
    search_knn(x, e, knn.begin(), knn.end());
  }

  //! All points with distance < \p radius (metric units: squared for the
  //! default metric), in traversal order unless \p sort.
  template <typename P_>
  inline void search_radius(
      P_ const& x,
      scalar_type const radius,
      std::vector<neighbor_type>& n,
      bool const sort = false) const {
    internal::radius_visitor<neighbor_type> v(radius, n);
    search_nearest(x, v);
    if (sort) v.sort();
  }

  template <typename P_>
  inline void search_radius(
      P_ const& x,
      scalar_type const radius,
      scalar_type const e,
      std::vector<neighbor_type>& n,
      bool const sort = false) const {
    internal::radius_visitor<neighbor_type, true> v(radius, n, e);
    search_nearest(x, v);
    if (sort) v.sort();
  }

  //! All indices inside the closed box [min, max].
  template <typename P_>
  inline void search_box(
      P_ const& min, P_ const& max, std::vector<index_type>& idxs) const {
    idxs.clear();
    space_view_type s = view();
    if constexpr (topological) {  // metric_box_map: intervals through the seam of a circle axis, four-bound tests
      internal::box_search<true>(
          tree_,
          s,
          internal::point_view<P_>(min).data(),
          internal::point_view<P_>(max).data(),
          idxs,
          internal::circle_axes_of<metric_type>{metric_});
    } else {
      internal::box_search(
          tree_,
          s,
          internal::point_view<P_>(min).data(),
          internal::point_view<P_>(max).data(),
          idxs);
    }
  }

  // ---- batched searches (MI355X) -------------------------------------------
  // QuerySpace_ is anything with space_traits<>.  Row i of every output belongs
  // to query i.  Only available for the accelerated instantiation.

  //! out[i] = nearest neighbour of query i.  out must hold queries.size().
  template <typename QuerySpace_>
  inline void search_nn(QuerySpace_ const& queries, neighbor_type* out) const {
    search_knn(queries, size_type(1), out);
  }

  //! out[i * k + j] = j-th nearest neighbour of query i (ascending).  Unlike the
  //! vector overload k is NOT clamped: k <= number of points is required.
  template <typename QuerySpace_>
  inline void search_knn(
      QuerySpace_ const& queries, size_type const k, neighbor_type* out) const {
    batched_knn(queries, k, scalar_type(1.0), out);
  }

  template <typename QuerySpace_>
  inline void search_knn(
      QuerySpace_ const& queries,
      size_type const k,
      scalar_type const e,
      neighbor_type* out) const {
    batched_knn(queries, k, e, out);
  }

  //! out[i] = all neighbours of query i within \p radius.
  template <typename QuerySpace_>
  inline void search_radius(
      QuerySpace_ const& queries,
      scalar_type const radius,
      std::vector<std::vector<neighbor_type>>& out,
      bool const sort = false) const {
    batched_radius(queries, radius, scalar_type(1.0), out, sort);
  }

  template <typename QuerySpace_>
  inline void search_radius(
      QuerySpace_ const& queries,
      scalar_type const radius,
      scalar_type const e,
      std::vector<std::vector<neighbor_type>>& out,
      bool const sort = false) const {
    batched_radius(queries, radius, e, out, sort);
  }

  //! Flat ragged form: row i is flat[offsets[i] .. offsets[i + 1]).
  template <typename QuerySpace_>
  inline void search_radius(
      QuerySpace_ const& queries,
      scalar_type const radius,
      std::vector<std::uint64_t>& offsets,
      std::vector<neighbor_type>& flat,
      bool const sort = false) const {
    static_assert(accelerated, "BATCHED_SEARCH_NEEDS_A_BACKEND_METRIC_FLOAT_OR_DOUBLE_INT");
    internal::dense_rows<internal::unwrap_ref_t<QuerySpace_>> q(unwrap(queries));
    check_query_dim(q.cols());
    offsets.assign(q.rows() + 1, 0);
    typename api::neighbor* rows = nullptr;
    try {
      internal::ptk_check(
          api::radius(
              device(), q.data(), q.rows(), radius, scalar_type(1), sort ? 1 : 0,
              offsets.data(), &rows),
          "ptk_search_radius");
    } catch (internal::ptk_unsupported const& refused) {
      if (!internal::host_loop_flag().load()) throw;  // (the batched path has no CPU fallback unless asked for)
      std::vector<std::vector<neighbor_type>> per_row;
      host_radius(q, radius, scalar_type(1), per_row, sort, refused.what());
      for (size_type i = 0; i < q.rows(); ++i) offsets[i + 1] = offsets[i] + per_row[i].size();
      flat.resize(offsets.back());
      for (size_type i = 0; i < q.rows(); ++i) std::copy(per_row[i].begin(), per_row[i].end(), flat.data() + offsets[i]);
      return;
    }
    library_rows keep(rows);  // freed even if the copy below throws
    flat.resize(offsets.back());
    auto const* src = reinterpret_cast<neighbor_type const*>(rows);
    std::copy(src, src + flat.size(), flat.data());
  }

  //! Batched box search: row i (flat[offsets[i] .. offsets[i + 1])) lists the indices inside the
  //! closed box [mins[i], maxs[i]], in the traversal order of the per-query search_box.
  template <typename BoxSpace_>
  inline void search_box(
      BoxSpace_ const& mins,
      BoxSpace_ const& maxs,
      std::vector<std::uint64_t>& offsets,
      std::vector<index_type>& flat) const {
    static_assert(accelerated, "BATCHED_SEARCH_NEEDS_A_BACKEND_METRIC_FLOAT_OR_DOUBLE_INT");
    internal::dense_rows<internal::unwrap_ref_t<BoxSpace_>> lo(unwrap(mins)), hi(unwrap(maxs));
    check_query_dim(lo.cols());
    check_query_dim(hi.cols());
    if (lo.rows() != hi.rows()) throw std::invalid_argument("query min and max don't have equal size");
    offsets.assign(lo.rows() + 1, 0);
    std::int32_t* rows = nullptr;
    try {
      internal::ptk_check(
          api::box(device(), lo.data(), hi.data(), lo.rows(), offsets.data(), &rows), "ptk_search_box");
    } catch (internal::ptk_unsupported const& refused) {
      if (!internal::host_loop_flag().load()) throw;  // (the batched path has no CPU fallback unless asked for)
      internal::warn_host_loop(refused.what());
      std::vector<std::vector<index_type>> per_row(lo.rows());
      space_view_type s = view();
      internal::host_rows_loop(lo.rows(), [&](size_type i) {
        if constexpr (topological)
          internal::box_search<true>(tree_, s, lo.data() + i * lo.cols(), hi.data() + i * hi.cols(), per_row[i],
                                     internal::circle_axes_of<metric_type>{metric_});
        else
          internal::box_search(tree_, s, lo.data() + i * lo.cols(), hi.data() + i * hi.cols(), per_row[i]);
      });
      for (size_type i = 0; i < lo.rows(); ++i) offsets[i + 1] = offsets[i] + per_row[i].size();
      flat.resize(offsets.back());
      for (size_type i = 0; i < lo.rows(); ++i) std::copy(per_row[i].begin(), per_row[i].end(), flat.data() + offsets[i]);
      return;
    }
    library_rows keep(rows);
    flat.assign(rows, rows + offsets.back());
  }

  //! Uploads the tree to the device now instead of at the first batched call.
  inline void prepare_device() const {
    static_assert(accelerated, "BATCHED_SEARCH_NEEDS_A_BACKEND_METRIC_FLOAT_OR_DOUBLE_INT");
    (void)device();
  }

  // ---- introspection --------------------------------------------------------

  //! Index range of every non-empty leaf, in depth-first order.
  inline std::vector<leaf_range_type> leaf_ranges() const {
    std::vector<leaf_range_type> ranges;
    for (auto const& nd : tree_.nodes) {
      if (nd.is_leaf() && nd.begin != nd.end) {
        ranges.emplace_back(
            tree_.indices.cbegin() + nd.begin, tree_.indices.cbegin() + nd.end);
      }
    }
    return ranges;
  }

  inline space_type const& space() const { return space_; }
  inline metric_type const& metric() const { return metric_; }

  // ---- persistence (reference-compatible byte stream) ------------------------

  static kd_tree load(space_type space, std::string const& filename) {
    std::fstream stream =
        internal::open_stream(filename, std::ios::in | std::ios::binary);
    return load(std::move(space), stream);
  }
  static kd_tree load(space_type space, std::iostream& stream) {
    return kd_tree(std::move(space), stream);
  }
  static void save(kd_tree const& tree, std::string const& filename) {
    std::fstream stream =
        internal::open_stream(filename, std::ios::out | std::ios::binary);
    save(tree, stream);
  }
  static void save(kd_tree const& tree, std::iostream& stream) {
    internal::write_flat_tree(tree.tree_, stream);
  }

 private:
  kd_tree(space_type space, std::iostream& stream)
      : space_(std::move(space)),
        metric_(),
        tree_(internal::read_flat_tree<tree_type>(
            stream, topological, space_view_type(unwrap(space_)).sdim(), space_view_type(unwrap(space_)).size())) {}

  //! A result buffer allocated by libptk: released with ptk_free on every path out of the scope.
  struct ptk_free_fn {
    void operator()(void* p) const { ptk_free(p); }
  };
  using library_rows = std::unique_ptr<void, ptk_free_fn>;

  template <typename T_>
  static T_ const& unwrap(T_ const& s) {
    return s;
  }
  template <typename T_>
  static T_ const& unwrap(std::reference_wrapper<T_> const& s) {
    return s.get();
  }

  space_view_type view() const {
    return space_view_type(unwrap(space_));
  }

  typename api::tree* device() const {
    return device_.get(tree_, view(), accelerated ? internal::ptk_metric_v<Metric_> : 0);
  }

  void check_query_dim(size_type qdim) const {
    if (qdim != view().sdim()) {
      throw std::invalid_argument("pico_tree: query dimension differs from the tree's");
    }
  }

  template <typename QuerySpace_>
  void batched_knn(
      QuerySpace_ const& queries, size_type k, scalar_type e, neighbor_type* out) const {
    static_assert(accelerated, "BATCHED_SEARCH_NEEDS_A_BACKEND_METRIC_FLOAT_OR_DOUBLE_INT");
    static_assert(sizeof(neighbor_type) == sizeof(typename api::neighbor), "neighbor layout");
    internal::dense_rows<internal::unwrap_ref_t<QuerySpace_>> q(unwrap(queries));
    check_query_dim(q.cols());
    try {
      internal::ptk_check(
          api::knn(
              device(), q.data(), q.rows(), static_cast<std::uint32_t>(k), e,
              reinterpret_cast<typename api::neighbor*>(out)),
          "ptk_search_knn");
    } catch (internal::ptk_unsupported const& refused) {
      if (!internal::host_loop_flag().load()) throw;  // (the batched path has no CPU fallback unless asked for)
      // The reference's loop (_pyco_tree/kd_tree.hpp:128-134): row i into out[i * k .. i * k + k).
      internal::warn_host_loop(refused.what());
      using row_point = point_map<scalar_type const, dim>;
      internal::host_rows_loop(q.rows(), [&](size_type i) {
        row_point x = make_row(q.data() + i * q.cols(), q.cols());
        if (e == scalar_type(1)) {
          search_knn(x, out + i * k, out + (i + 1) * k);
        } else {
          search_knn(x, e, out + i * k, out + (i + 1) * k);
        }
      });
    }
  }

  //! Row i of a dense query matrix as a point.
  static point_map<scalar_type const, dim> make_row(scalar_type const* p, size_type sdim) {
    if constexpr (dim == dynamic_extent) {
      return point_map<scalar_type const, dim>(p, sdim);
    } else {
      (void)sdim;
      return point_map<scalar_type const, dim>(p);
    }
  }

  template <typename Rows_>
  void host_radius(
      Rows_ const& q,
      scalar_type radius,
      scalar_type e,
      std::vector<std::vector<neighbor_type>>& out,
      bool sort,
      char const* why) const {
    internal::warn_host_loop(why);
    out.resize(q.rows());
    internal::host_rows_loop(q.rows(), [&](size_type i) {
      auto x = make_row(q.data() + i * q.cols(), q.cols());
      if (e == scalar_type(1)) {
        search_radius(x, radius, out[i], sort);
      } else {
        search_radius(x, radius, e, out[i], sort);
      }
    });
  }

  template <typename QuerySpace_>
  void batched_radius(
      QuerySpace_ const& queries,
      scalar_type radius,
      scalar_type e,
      std::vector<std::vector<neighbor_type>>& out,
      bool sort) const {
    static_assert(accelerated, "BATCHED_SEARCH_NEEDS_A_BACKEND_METRIC_FLOAT_OR_DOUBLE_INT");
    internal::dense_rows<internal::unwrap_ref_t<QuerySpace_>> q(unwrap(queries));
    check_query_dim(q.cols());
    std::vector<std::uint64_t> offsets(q.rows() + 1, 0);
    typename api::neighbor* rows = nullptr;
    try {
      internal::ptk_check(
          api::radius(
              device(), q.data(), q.rows(), radius, e, sort ? 1 : 0, offsets.data(), &rows),
          "ptk_search_radius");
    } catch (internal::ptk_unsupported const& refused) {
      if (!internal::host_loop_flag().load()) throw;  // (the batched path has no CPU fallback unless asked for)
      host_radius(q, radius, e, out, sort, refused.what());
      return;
    }
    library_rows keep(rows);
    out.resize(q.rows());
    auto const* src = reinterpret_cast<neighbor_type const*>(rows);
    for (size_type i = 0; i < q.rows(); ++i) {
      out[i].assign(src + offsets[i], src + offsets[i + 1]);
    }
  }

  space_type space_;
  metric_type metric_;
  tree_type tree_;
  internal::device_tree<std::conditional_t<accelerated, scalar_type, float>> device_;
};

template <typename Space_, typename... Args>
kd_tree(Space_, Args...) -> kd_tree<Space_, metric_l2_squared, int>;

template <
    typename Metric_ = metric_l2_squared,
    typename Index_ = int,
    typename Bounds_ = bounds_from_space_t,
    typename Rule_ = sliding_midpoint_max_side_t,
    typename Space_,
    typename Stop_>
kd_tree<std::decay_t<Space_>, Metric_, Index_> make_kd_tree(
    Space_&& space,
    splitter_stop_condition_t<Stop_> const& stop_condition,
    splitter_start_bounds_t<Bounds_> const& start_bounds = Bounds_{},
    splitter_rule_t<Rule_> const& rule = Rule_{}) {
  return kd_tree<std::decay_t<Space_>, Metric_, Index_>(
      std::forward<Space_>(space), stop_condition, start_bounds, rule);
}

}  // namespace pico_tree
