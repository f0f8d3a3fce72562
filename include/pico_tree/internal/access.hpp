#pragma once
//! \file access.hpp
//! \brief Uniform access to user point / space types through their traits.
//! \details The algorithms in this directory never touch a user type directly;
//! they see a point as `scalar const*` + length and a space as an indexable set
//! of such points.  Same role as the reference's point_wrapper.hpp:15-48 and
//! space_wrapper.hpp:16-54, written independently.

#include <functional>
#include <type_traits>

#include "../core.hpp"
#include "../traits.hpp"

namespace pico_tree::internal {

template <typename T_>
struct unwrap_ref {
  using type = T_;
};
template <typename T_>
struct unwrap_ref<std::reference_wrapper<T_>> {
  using type = std::remove_cv_t<T_>;
};
template <typename T_>
using unwrap_ref_t = typename unwrap_ref<T_>::type;

//! Read-only view of one point.
template <typename Point_>
class point_view {
  using traits = point_traits<Point_>;

 public:
  using scalar_type = typename traits::scalar_type;
  using size_type = size_t;
  static constexpr size_type dim = traits::dim;

  explicit point_view(Point_ const& p) : p_(p) {}

  scalar_type const* data() const { return traits::data(p_); }
  size_type size() const {
    if constexpr (dim != dynamic_extent) {
      return dim;
    } else {
      return traits::size(p_);
    }
  }
  scalar_type const* begin() const { return data(); }
  scalar_type const* end() const { return data() + size(); }
  scalar_type const& operator[](size_type i) const { return data()[i]; }

 private:
  Point_ const& p_;
};

//! Read-only view of a space (Space_ is the unwrapped space type).
template <typename Space_>
class space_view {
  using traits = space_traits<Space_>;
  using point_type = typename traits::point_type;

 public:
  using scalar_type = typename traits::scalar_type;
  using size_type = size_t;
  static constexpr size_type dim = traits::dim;

  explicit space_view(Space_ const& s) : s_(s) {}

  //! Coordinates of point \p i.
  template <typename Index_>
  scalar_type const* operator[](Index_ i) const {
    return point_traits<point_type>::data(traits::point_at(s_, i));
  }
  size_type size() const { return traits::size(s_); }
  size_type sdim() const {
    if constexpr (dim != dynamic_extent) {
      return dim;
    } else {
      return traits::sdim(s_);
    }
  }

 private:
  Space_ const& s_;
};

}  // namespace pico_tree::internal
