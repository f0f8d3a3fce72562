#pragma once
//! \file core.hpp
//! \brief Vocabulary types shared by every pico_tree header of this repository.
//! \details Written from scratch for the MI355X build; keeps the public names of
//! the reference (/root/reference/src/pico_tree/pico_tree/core.hpp:13-54) so
//! user code compiles unchanged: pico_tree::size_t, dynamic_extent, neighbor<>.

#include <cstddef>
#include <type_traits>

namespace pico_tree {

using size_t = std::size_t;

//! Marks a spatial dimension that is only known at run time.
inline constexpr size_t dynamic_extent = static_cast<size_t>(-1);

//! One search result: the index of a point and its distance to the query.
//! Trivial and exactly {Index_, Scalar_} so that neighbor<int, float> is
//! layout-identical to the C-ABI's ptk_neighbor (include/ptk.h).
template <typename Index_, typename Scalar_>
struct neighbor {
  static_assert(std::is_integral_v<Index_>, "INDEX_NOT_AN_INTEGRAL_TYPE");
  static_assert(std::is_arithmetic_v<Scalar_>, "SCALAR_NOT_AN_ARITHMETIC_TYPE");

  using index_type = Index_;
  using scalar_type = Scalar_;

  constexpr neighbor() = default;
  constexpr neighbor(index_type i, scalar_type d) noexcept
      : index(i), distance(d) {}

  index_type index;
  scalar_type distance;
};

//! Neighbors order by distance only (ties are unordered).
template <typename I_, typename S_>
constexpr bool operator<(
    neighbor<I_, S_> const& a, neighbor<I_, S_> const& b) noexcept {
  return a.distance < b.distance;
}

}  // namespace pico_tree
