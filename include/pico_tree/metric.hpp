#pragma once
//! \file metric.hpp
//! \brief Metrics usable with kd_tree (reference: metric.hpp:53-186).
//! \details A metric provides (i) space_category, (ii) a point-to-point call
//! operator()(begin1, end1, begin2) and (iii) a one-dimensional call
//! operator()(x) that is the un-rooted per-axis term, so that a box distance is
//! a sum of per-axis terms.  Only metric_l2_squared on float is accelerated by
//! the HIP backend; the others run the host traversal.  The topological metrics
//! of the reference (metric_so2, metric_se2_squared) are out of scope here.

#include <cmath>
#include <iterator>
#include <limits>

#include "core.hpp"
#include "distance.hpp"

namespace pico_tree {

class topological_space_tag {};
class euclidean_space_tag : public topological_space_tag {};

namespace internal {

//! Left-to-right accumulation d = 0; d += term(a_i, b_i).  The order and the
//! absence of fused multiply-add are part of the results contract (the GPU
//! kernels reproduce exactly this sequence).
template <typename It1_, typename End1_, typename It2_, typename Term_>
constexpr auto accumulate_terms(It1_ a, End1_ a_end, It2_ b, Term_ term) {
  using scalar = typename std::iterator_traits<It1_>::value_type;
  scalar d{};
  for (; a != a_end; ++a, ++b) {
    d += term(*a, *b);
  }
  return d;
}

}  // namespace internal

struct metric_l1 {
  using space_category = euclidean_space_tag;

  template <typename It1_, typename End1_, typename It2_>
  constexpr auto operator()(It1_ a, End1_ a_end, It2_ b) const {
    return internal::accumulate_terms(
        a, a_end, b, [](auto x, auto y) { return r1_distance(x, y); });
  }

  template <typename S_>
  constexpr S_ operator()(S_ x) const {
    return std::abs(x);
  }
};

struct metric_l2_squared {
  using space_category = euclidean_space_tag;

  template <typename It1_, typename End1_, typename It2_>
  constexpr auto operator()(It1_ a, End1_ a_end, It2_ b) const {
    return internal::accumulate_terms(
        a, a_end, b, [](auto x, auto y) { return squared_r1_distance(x, y); });
  }

  template <typename S_>
  constexpr S_ operator()(S_ x) const {
    return squared(x);
  }
};

struct metric_lpinf {
  using space_category = euclidean_space_tag;

  template <typename It1_, typename End1_, typename It2_>
  constexpr auto operator()(It1_ a, End1_ a_end, It2_ b) const {
    using scalar = typename std::iterator_traits<It1_>::value_type;
    scalar d{};
    for (; a != a_end; ++a, ++b) d = std::max(d, r1_distance(*a, *b));
    return d;
  }

  template <typename S_>
  constexpr S_ operator()(S_ x) const {
    return std::abs(x);
  }
};

struct metric_lninf {
  using space_category = euclidean_space_tag;

  template <typename It1_, typename End1_, typename It2_>
  constexpr auto operator()(It1_ a, End1_ a_end, It2_ b) const {
    using scalar = typename std::iterator_traits<It1_>::value_type;
    scalar d = std::numeric_limits<scalar>::max();
    for (; a != a_end; ++a, ++b) d = std::min(d, r1_distance(*a, *b));
    return d;
  }

  template <typename S_>
  constexpr S_ operator()(S_ x) const {
    return std::abs(x);
  }
};

}  // namespace pico_tree
