#pragma once
//! \file map.hpp
//! \brief Non-owning views over raw coordinate memory: point_map and space_map.
//! \details Same public surface as the reference (map.hpp:91-169,
//! map_traits.hpp:13-44): point_map<Scalar_, Dim_> views Dim_ (or a run-time
//! number of) contiguous scalars; space_map<Point_> views an array of points;
//! space_map<point_map<Scalar_, Dim_>> views a row-major matrix of scalars.
//! These are what a binding uses to hand a foreign buffer to kd_tree, and what
//! the batched query members accept as a query space.

#include <type_traits>

#include "core.hpp"
#include "traits.hpp"

namespace pico_tree {

namespace internal {

//! A length that is either a compile-time constant or a stored value.
template <size_t Extent_>
struct extent_holder {
  constexpr explicit extent_holder(size_t) noexcept {}
  static constexpr size_t value() noexcept { return Extent_; }
};

template <>
struct extent_holder<dynamic_extent> {
  constexpr explicit extent_holder(size_t v) noexcept : v_(v) {}
  constexpr size_t value() const noexcept { return v_; }
  size_t v_;
};

}  // namespace internal

template <typename Scalar_, size_t Dim_>
class point_map {
  static_assert(std::is_arithmetic_v<Scalar_>, "SCALAR_NOT_AN_ARITHMETIC_TYPE");
  static_assert(Dim_ == dynamic_extent || Dim_ > 0, "DIM_MUST_BE_DYNAMIC_OR_>_0");

 public:
  using element_type = Scalar_;
  using scalar_type = std::remove_cv_t<Scalar_>;
  using size_type = size_t;
  static constexpr size_type dim = Dim_;

  explicit constexpr point_map(element_type* data) noexcept
      : data_(data), size_(Dim_) {}

  constexpr point_map(element_type* data, size_type size) noexcept
      : data_(data), size_(size) {}

  template <typename It_>
  constexpr point_map(It_ begin, It_ end) noexcept
      : data_(&(*begin)), size_(static_cast<size_type>(end - begin)) {}

  constexpr element_type& operator[](size_type i) const { return data_[i]; }
  constexpr element_type* data() const noexcept { return data_; }
  constexpr size_type size() const noexcept { return size_.value(); }

 private:
  element_type* data_;
  internal::extent_holder<Dim_> size_;
};

//! View over an array of Point_ objects (any type with point_traits and a
//! compile-time dimension).
template <typename Point_>
class space_map {
 public:
  using point_element_type = Point_;
  using point_type = std::remove_cv_t<Point_>;
  using scalar_type = typename point_traits<point_type>::scalar_type;
  using size_type = size_t;
  static constexpr size_type dim = point_traits<point_type>::dim;

  static_assert(
      dim != dynamic_extent, "SPACE_MAP_OF_POINT_DOES_NOT_SUPPORT_DYNAMIC_DIM");

  constexpr space_map(point_element_type* data, size_type size) noexcept
      : data_(data), size_(size) {}

  constexpr point_element_type& operator[](size_type i) const {
    return data_[i];
  }
  constexpr point_element_type* data() const noexcept { return data_; }
  constexpr size_type size() const noexcept { return size_; }
  constexpr size_type sdim() const noexcept { return dim; }

 private:
  point_element_type* data_;
  size_type size_;
};

//! View over a row-major (size x sdim) matrix of scalars.
template <typename Scalar_, size_t Dim_>
class space_map<point_map<Scalar_, Dim_>> {
 public:
  using point_type = point_map<Scalar_, Dim_>;
  using scalar_type = typename point_type::scalar_type;
  using scalar_element_type = typename point_type::element_type;
  using size_type = size_t;
  static constexpr size_type dim = Dim_;

  constexpr space_map(scalar_element_type* data, size_type size) noexcept
      : data_(data), size_(size), sdim_(Dim_) {}

  constexpr space_map(
      scalar_element_type* data, size_type size, size_type sdim) noexcept
      : data_(data), size_(size), sdim_(sdim) {}

  constexpr point_type operator[](size_type i) const noexcept {
    return point_type(data(i), sdim());
  }
  constexpr scalar_element_type* data() const noexcept { return data_; }
  constexpr scalar_element_type* data(size_type i) const noexcept {
    return data_ + i * sdim();
  }
  constexpr size_type size() const noexcept { return size_; }
  constexpr size_type sdim() const noexcept { return sdim_.value(); }

 private:
  scalar_element_type* data_;
  size_type size_;
  internal::extent_holder<Dim_> sdim_;
};

template <typename Scalar_, size_t Dim_>
struct point_traits<point_map<Scalar_, Dim_>> {
  using point_type = point_map<Scalar_, Dim_>;
  using scalar_type = typename point_type::scalar_type;
  using size_type = size_t;
  static constexpr size_type dim = Dim_;

  static scalar_type const* data(point_type const& p) { return p.data(); }
  static size_type size(point_type const& p) { return p.size(); }
};

template <typename Point_>
struct space_traits<space_map<Point_>> {
  using space_type = space_map<Point_>;
  using point_type = typename space_type::point_type;
  using scalar_type = typename space_type::scalar_type;
  using size_type = size_t;
  static constexpr size_type dim = space_type::dim;

  template <typename Index_>
  static decltype(auto) point_at(space_type const& s, Index_ i) {
    return s[static_cast<size_type>(i)];
  }
  static size_type size(space_type const& s) { return s.size(); }
  static size_type sdim(space_type const& s) { return s.sdim(); }
};

}  // namespace pico_tree
