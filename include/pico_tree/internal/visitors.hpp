#pragma once
//! \file visitors.hpp
//! \brief Result policies of a nearest-neighbour traversal.
//! \details A visitor is called as v(index, distance) for every point the
//! traversal measures and exposes max(), the current pruning distance (user
//! visitors follow the same concept, see the reference's
//! examples/kd_tree/kd_tree_custom_search_visitor.cpp:9-45).  The semantics the
//! GPU kernels replicate are fixed here:
//!   * nn:     replace only when strictly closer       (search_visitor.hpp:54-58)
//!   * knn:    sorted k-list, stable among equal distances, k-th slot starts at
//!             FLT_MAX                                  (:24-38, :98-118)
//!   * radius: keep when strictly inside, traversal order, optional std::sort
//!                                                      (:133-151)
//!   * approximate variants scale every candidate by 1/e before comparing and
//!     storing                                          (:165-288)

#include <algorithm>
#include <iterator>
#include <limits>
#include <vector>

#include "../core.hpp"

namespace pico_tree::internal {

//! Inserts \p item into the sorted range [begin, end), dropping the last
//! element.  Elements are shifted while item < previous, so an item never
//! overtakes an equal one: first come, first kept.
template <
    typename RandomAccessIterator_,
    typename Compare_ = std::less<
        typename std::iterator_traits<RandomAccessIterator_>::value_type>>
inline void insert_sorted(
    RandomAccessIterator_ begin,
    RandomAccessIterator_ end,
    typename std::iterator_traits<RandomAccessIterator_>::value_type item,
    Compare_ comp = Compare_()) {
  RandomAccessIterator_ slot = end - 1;
  while (slot > begin && comp(item, *(slot - 1))) {
    *slot = std::move(*(slot - 1));
    --slot;
  }
  *slot = std::move(item);
}

//! Scale applied to candidate distances: identity for the exact searches.
template <typename Scalar_, bool Approximate_>
struct candidate_scale {
  constexpr explicit candidate_scale(Scalar_) {}
  constexpr Scalar_ operator()(Scalar_ d) const { return d; }
};
template <typename Scalar_>
struct candidate_scale<Scalar_, true> {
  constexpr explicit candidate_scale(Scalar_ e) : inv_(Scalar_(1.0) / e) {}
  constexpr Scalar_ operator()(Scalar_ d) const { return d * inv_; }
  Scalar_ inv_;
};

template <typename Neighbor_, bool Approximate_ = false>
class nn_visitor {
 public:
  using neighbor_type = Neighbor_;
  using index_type = typename Neighbor_::index_type;
  using scalar_type = typename Neighbor_::scalar_type;

  explicit nn_visitor(neighbor_type& nn, scalar_type e = scalar_type(1.0))
      : scale_(e), nn_(nn) {
    nn_.distance = std::numeric_limits<scalar_type>::max();
  }

  void operator()(index_type idx, scalar_type dst) const {
    dst = scale_(dst);
    if (max() > dst) nn_ = {idx, dst};
  }
  scalar_type max() const { return nn_.distance; }

 private:
  candidate_scale<scalar_type, Approximate_> scale_;
  neighbor_type& nn_;
};

template <typename RandomAccessIterator_, bool Approximate_ = false>
class knn_visitor {
 public:
  using neighbor_type =
      typename std::iterator_traits<RandomAccessIterator_>::value_type;
  using index_type = typename neighbor_type::index_type;
  using scalar_type = typename neighbor_type::scalar_type;

  static_assert(
      std::is_base_of_v<
          std::random_access_iterator_tag,
          typename std::iterator_traits<RandomAccessIterator_>::iterator_category>,
      "EXPECTED_RANDOM_ACCESS_ITERATOR");

  knn_visitor(
      RandomAccessIterator_ begin,
      RandomAccessIterator_ end,
      scalar_type e = scalar_type(1.0))
      : scale_(e), begin_(begin), end_(end), filled_(begin) {
    (end_ - 1)->distance = std::numeric_limits<scalar_type>::max();
  }

  void operator()(index_type idx, scalar_type dst) {
    dst = scale_(dst);
    if (max() > dst) {
      if (filled_ < end_) ++filled_;
      insert_sorted(begin_, filled_, neighbor_type{idx, dst});
    }
  }
  scalar_type max() const { return (end_ - 1)->distance; }

 private:
  candidate_scale<scalar_type, Approximate_> scale_;
  RandomAccessIterator_ begin_;
  RandomAccessIterator_ end_;
  RandomAccessIterator_ filled_;
};

template <typename Neighbor_, bool Approximate_ = false>
class radius_visitor {
 public:
  using neighbor_type = Neighbor_;
  using index_type = typename Neighbor_::index_type;
  using scalar_type = typename Neighbor_::scalar_type;

  radius_visitor(
      scalar_type radius,
      std::vector<neighbor_type>& out,
      scalar_type e = scalar_type(1.0))
      : scale_(e), radius_(scale_(radius)), out_(out) {
    out_.clear();
  }

  void operator()(index_type idx, scalar_type dst) const {
    dst = scale_(dst);
    if (max() > dst) out_.push_back({idx, dst});
  }
  void sort() const { std::sort(out_.begin(), out_.end()); }
  scalar_type max() const { return radius_; }

 private:
  candidate_scale<scalar_type, Approximate_> scale_;
  scalar_type radius_;
  std::vector<neighbor_type>& out_;
};

}  // namespace pico_tree::internal
