#pragma once
// TEST-ONLY stand-in for the reference's pico_toolshed/scoped_timer.hpp (see point.hpp here).

#include <chrono>
#include <cstddef>
#include <iostream>
#include <string>

namespace pico_tree {

class scoped_timer {
 public:
  explicit scoped_timer(std::string name, std::size_t runs = 1)
      : name_(std::move(name)), runs_(runs), start_(std::chrono::steady_clock::now()) {}
  ~scoped_timer() {
    std::chrono::duration<double, std::milli> ms = std::chrono::steady_clock::now() - start_;
    std::cout << name_ << ": " << ms.count() / static_cast<double>(runs_) << " ms\n";
  }

 private:
  std::string name_;
  std::size_t runs_;
  std::chrono::steady_clock::time_point start_;
};

}  // namespace pico_tree
