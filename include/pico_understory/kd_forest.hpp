#pragma once
//! \file kd_forest.hpp
//! \brief pico_tree::kd_forest -- randomised kd-forest for approximate nearest neighbours in
//! high dimensions, searched on the MI355X.
//! \details Public shape of the reference's pico_understory class
//! (/root/reference/examples/pico_understory/pico_understory/kd_forest.hpp:12-123): same
//! template parameters, same constructor `(space, max_leaf_size, forest_size)`, same
//! `search_nn(x, max_leaves_visited, nn)`.  The search itself lives behind the C ABI
//! (ptk_forest_* in ptk.h, device code in pico_tree_amd/csrc/ptk_forest.hpp); there is no host
//! search path in this header, so every call needs the backend and throws without it.
//!
//! Added, with no reference counterpart (the reference is one query at a time): batched members
//! `search_nn(QuerySpace, max_leaves, neighbor*)` and `search_knn(QuerySpace, k, max_leaves,
//! neighbor*)`.  Differences from the reference that the backend documents: the k-list is
//! de-duplicated by index, distances are measured in the original space, and the Householder
//! reflections derive from a seed (4th constructor argument) instead of std::random_device.
//! Only `Metric_ = metric_l2_squared`, `float` scalars and `Index_ = int` are built.

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../ptk.h"
#include "../pico_tree/core.hpp"
#include "../pico_tree/internal/access.hpp"
#include "../pico_tree/internal/backend.hpp"
#include "../pico_tree/metric.hpp"

namespace pico_tree {

template <typename Space_, typename Metric_ = metric_l2_squared, typename Index_ = int>
class kd_forest {
  using space_view_type = internal::space_view<Space_>;

 public:
  using size_type = size_t;
  using index_type = Index_;
  using scalar_type = typename space_view_type::scalar_type;
  static size_type constexpr dim = space_view_type::dim;
  using space_type = Space_;
  using metric_type = Metric_;
  using neighbor_type = neighbor<index_type, scalar_type>;

  static_assert(
      internal::is_accelerated_v<Metric_, scalar_type, Index_> && std::is_same_v<Metric_, metric_l2_squared>,
      "kd_forest is built for metric_l2_squared over float points with int indices");

  kd_forest(space_type space, size_type max_leaf_size, size_type forest_size, std::uint64_t seed = 0)
      : space_(std::move(space)), metric_() {
    internal::dense_rows<Space_> rows(space_);
    ptk_forest* h = nullptr;
    internal::ptk_check(
        ptk_forest_create(
            rows.data(), rows.rows(), static_cast<std::uint32_t>(rows.cols()), max_leaf_size,
            static_cast<std::uint32_t>(forest_size), seed, PTK_DEVICE_CURRENT, &h),
        "ptk_forest_create");
    handle_ = std::shared_ptr<ptk_forest>(h, &ptk_forest_destroy);
  }

  kd_forest(kd_forest const&) = delete;
  kd_forest(kd_forest&&) = default;
  kd_forest& operator=(kd_forest const&) = delete;
  kd_forest& operator=(kd_forest&&) = default;

  //! Nearest neighbour of the single point \p x (kd_forest.hpp:78-85).
  template <typename P_>
  inline void search_nn(P_ const& x, size_type max_leaves_visited, neighbor_type& nn) const {
    internal::point_view<P_> p(x);
    ptk_neighbor out{};
    internal::ptk_check(
        ptk_forest_search_knn(handle_.get(), p.data(), 1, 1, max_leaves_visited, &out),
        "ptk_forest_search_knn");
    nn.index = out.index;
    nn.distance = out.distance;
  }

  //! Batched: out[i] is the nearest neighbour of query i.
  template <typename QuerySpace_, typename = std::enable_if_t<!std::is_same_v<QuerySpace_, neighbor_type>>>
  inline void search_nn(QuerySpace_ const& queries, size_type max_leaves_visited, neighbor_type* out) const {
    search_knn(queries, size_type(1), max_leaves_visited, out);
  }

  //! Batched: out is nq x k row-major, row i ascending; slots beyond the distinct points found
  //! hold {-1, FLT_MAX}.
  template <typename QuerySpace_>
  inline void search_knn(
      QuerySpace_ const& queries, size_type k, size_type max_leaves_visited, neighbor_type* out) const {
    static_assert(sizeof(neighbor_type) == sizeof(ptk_neighbor), "neighbor layout");
    internal::dense_rows<QuerySpace_> rows(queries);
    internal::ptk_check(
        ptk_forest_search_knn(
            handle_.get(), rows.data(), rows.rows(), static_cast<std::uint32_t>(k), max_leaves_visited,
            reinterpret_cast<ptk_neighbor*>(out)),
        "ptk_forest_search_knn");
  }

  space_type const& space() const { return space_; }
  metric_type const& metric() const { return metric_; }

 private:
  space_type space_;
  metric_type metric_;
  std::shared_ptr<ptk_forest> handle_;
};

template <typename Space_>
kd_forest(Space_, size_t, size_t) -> kd_forest<Space_, metric_l2_squared, int>;

}  // namespace pico_tree
