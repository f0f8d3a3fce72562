#pragma once
//! \file distance.hpp
//! \brief Scalar distance helpers (reference: distance.hpp:19-48).

#include <algorithm>
#include <cmath>

namespace pico_tree {

class one_space_r1 {};
class one_space_s1 {};

template <typename S_>
constexpr S_ squared(S_ x) {
  return x * x;
}

template <typename S_>
constexpr S_ r1_distance(S_ x, S_ y) {
  return std::abs(x - y);
}

template <typename S_>
constexpr S_ squared_r1_distance(S_ x, S_ y) {
  return squared(x - y);
}

//! Distance on the unit circle [0, 1] / 0 ~ 1.
template <typename S_>
constexpr S_ s1_distance(S_ x, S_ y) {
  S_ const d = std::abs(x - y);
  return std::min(d, S_(1) - d);
}

template <typename S_>
constexpr S_ squared_s1_distance(S_ x, S_ y) {
  return squared(s1_distance(x, y));
}

}  // namespace pico_tree
