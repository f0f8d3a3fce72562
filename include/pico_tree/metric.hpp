#pragma once
//! \file metric.hpp
//! \brief Metrics usable with kd_tree (reference: metric.hpp:53-186).
//! \details A metric provides (i) space_category, (ii) a point-to-point call
//! operator()(begin1, end1, begin2) and (iii) a one-dimensional call
//! operator()(x) that is the un-rooted per-axis term, so that a box distance is
//! a sum of per-axis terms.  metric_l2_squared, metric_l1 and metric_lpinf are
//! accelerated by the HIP backend (float and double points); the others run the host
//! traversal: metric_lninf and the topological metrics metric_so2 / metric_se2_squared
//! (reference metric.hpp:186-257), which also provide apply_dim_space(dim, f) to tell
//! the search which axes wrap around (internal/flat_search.hpp, nearest_search_topological).

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

//! Distances on the unit circle S1 = [0, 1] / 0 ~ 1 (reference metric.hpp:186-220).
struct metric_so2 {
  using space_category = topological_space_tag;

  template <typename It1_, typename End1_, typename It2_>
  constexpr auto operator()(It1_ a, End1_, It2_ b) const {
    return s1_distance(*a, *b);
  }

  template <typename S_>
  constexpr S_ operator()(S_ x) const {
    return std::abs(x);
  }

  template <typename F_>
  void apply_dim_space(int, F_ f) const {
    f(one_space_s1{});
  }
};

//! Squared distances between Euclidean motions of the plane: (x, y) in R2 and an angle in
//! [0, 1] / 0 ~ 1 (reference metric.hpp:222-257).
struct metric_se2_squared {
  using space_category = topological_space_tag;

  template <typename It1_, typename End1_, typename It2_>
  constexpr auto operator()(It1_ a, End1_, It2_ b) const {
    return internal::accumulate_terms(
               a, a + 2, b, [](auto x, auto y) { return squared_r1_distance(x, y); }) +
           squared_s1_distance(*(a + 2), *(b + 2));
  }

  template <typename S_>
  constexpr S_ operator()(S_ x) const {
    return squared(x);
  }

  template <typename F_>
  void apply_dim_space(int dim, F_ f) const {
    if (dim < 2) {
      f(one_space_r1{});
    } else {
      f(one_space_s1{});
    }
  }
};

}  // namespace pico_tree
