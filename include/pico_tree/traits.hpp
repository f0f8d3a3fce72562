#pragma once
//! \file traits.hpp
//! \brief The two customisation points of the library and their stock
//! specialisations.
//! \details Same contracts as the reference (point_traits.hpp:13,
//! space_traits.hpp:12-30, array_traits.hpp:12-44, vector_traits.hpp:16-43):
//!
//!   point_traits<P>: scalar_type, dim, data(p) -> scalar const*, size(p)
//!   space_traits<S>: space_type, point_type, scalar_type, dim,
//!                    point_at(s, i), size(s), sdim(s)
//!
//! User code specialises these for its own types exactly as with the reference
//! (examples/kd_tree/kd_tree_custom_point_type.cpp:16-31,
//! kd_tree_custom_space_type.cpp:12-37).

#include <array>
#include <functional>
#include <type_traits>
#include <vector>

#include "core.hpp"

namespace pico_tree {

template <typename Point_>
struct point_traits;

template <typename Space_>
struct space_traits;

// ---- points ----------------------------------------------------------------

namespace internal {

//! Shared body for fixed-size, contiguous point types.
template <typename Point_, typename Scalar_, std::size_t Dim_>
struct fixed_point_traits {
  using point_type = Point_;
  using scalar_type = Scalar_;
  using size_type = size_t;
  static constexpr size_type dim = static_cast<size_type>(Dim_);
  static constexpr size_type size(point_type const&) { return dim; }
};

}  // namespace internal

//! Scalar_[Dim_]
template <typename Scalar_, std::size_t Dim_>
struct point_traits<Scalar_[Dim_]>
    : internal::fixed_point_traits<Scalar_[Dim_], Scalar_, Dim_> {
  static constexpr Scalar_ const* data(Scalar_ const (&p)[Dim_]) { return p; }
};

//! std::array<Scalar_, Dim_>
template <typename Scalar_, std::size_t Dim_>
struct point_traits<std::array<Scalar_, Dim_>>
    : internal::fixed_point_traits<std::array<Scalar_, Dim_>, Scalar_, Dim_> {
  static constexpr Scalar_ const* data(std::array<Scalar_, Dim_> const& p) {
    return p.data();
  }
};

// ---- spaces ----------------------------------------------------------------

//! std::vector of any point type with a compile-time dimension.
template <typename Point_, typename Allocator_>
struct space_traits<std::vector<Point_, Allocator_>> {
  using space_type = std::vector<Point_, Allocator_>;
  using point_type = Point_;
  using scalar_type = typename point_traits<Point_>::scalar_type;
  using size_type = size_t;
  static constexpr size_type dim = point_traits<Point_>::dim;

  static_assert(
      dim != dynamic_extent, "VECTOR_OF_POINT_DOES_NOT_SUPPORT_DYNAMIC_DIM");

  template <typename Index_>
  static Point_ const& point_at(space_type const& s, Index_ i) {
    return s[static_cast<size_type>(i)];
  }
  static size_type size(space_type const& s) { return s.size(); }
  static constexpr size_type sdim(space_type const&) { return dim; }
};

//! std::reference_wrapper<Space_>: lets a kd_tree borrow instead of own.
template <typename Space_>
struct space_traits<std::reference_wrapper<Space_>>
    : space_traits<std::remove_const_t<Space_>> {
  using space_type = std::reference_wrapper<Space_>;
};

}  // namespace pico_tree
