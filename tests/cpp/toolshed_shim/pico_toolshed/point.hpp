#pragma once
// TEST-ONLY stand-in for the reference's examples/pico_toolshed/pico_toolshed/point.hpp, so
// that the reference's example programs can be compiled IN PLACE against include/pico_tree
// (tests/test_cpp_api.py::test_reference_examples_compile_unchanged).  Written from scratch:
// a fixed-size point with the members those examples use, its point_traits, the point_Nf /
// point_Nd aliases and a seeded generate_random_n.

#include <array>
#include <cstddef>
#include <ostream>
#include <random>
#include <vector>

#include <pico_tree/core.hpp>
#include <pico_tree/traits.hpp>

namespace pico_tree {

template <typename Scalar_, std::size_t Dim_>
struct point {
  using scalar_type = Scalar_;
  using size_type = std::size_t;
  static constexpr size_type dim = Dim_;

  std::array<Scalar_, Dim_> elems_;

  constexpr Scalar_& operator[](size_type i) { return elems_[i]; }
  constexpr Scalar_ const& operator[](size_type i) const { return elems_[i]; }
  constexpr Scalar_ const* data() const { return elems_.data(); }
  constexpr Scalar_* data() { return elems_.data(); }
  constexpr size_type size() const { return Dim_; }
  void fill(Scalar_ v) { elems_.fill(v); }

  point& operator+=(Scalar_ v) {
    for (auto& e : elems_) e += v;
    return *this;
  }
  point operator+(Scalar_ v) const {
    point p = *this;
    return p += v;
  }
  point& operator-=(Scalar_ v) {
    for (auto& e : elems_) e -= v;
    return *this;
  }
  point operator-(Scalar_ v) const {
    point p = *this;
    return p -= v;
  }
};

template <typename Scalar_, std::size_t Dim_>
struct point_traits<point<Scalar_, Dim_>> {
  using point_type = point<Scalar_, Dim_>;
  using scalar_type = Scalar_;
  using size_type = std::size_t;
  static constexpr size_type dim = Dim_;
  static Scalar_ const* data(point_type const& p) { return p.data(); }
  static constexpr size_type size(point_type const&) { return Dim_; }
};

template <typename Scalar_, std::size_t Dim_>
std::ostream& operator<<(std::ostream& s, point<Scalar_, Dim_> const& p) {
  for (std::size_t i = 0; i < Dim_; ++i) s << (i ? " " : "") << p[i];
  return s;
}

using point_1f = point<float, 1>;
using point_2f = point<float, 2>;
using point_3f = point<float, 3>;
using point_1d = point<double, 1>;
using point_2d = point<double, 2>;
using point_3d = point<double, 3>;

template <typename Point_>
std::vector<Point_> generate_random_n(
    std::size_t n, typename Point_::scalar_type min, typename Point_::scalar_type max) {
  std::mt19937 gen(12345);
  std::uniform_real_distribution<typename Point_::scalar_type> dist(min, max);
  std::vector<Point_> out(n);
  for (auto& p : out)
    for (std::size_t d = 0; d < Point_::dim; ++d) p[d] = dist(gen);
  return out;
}

template <typename Point_>
std::vector<Point_> generate_random_n(std::size_t n, typename Point_::scalar_type size) {
  return generate_random_n<Point_>(n, typename Point_::scalar_type(0), size);
}

}  // namespace pico_tree
