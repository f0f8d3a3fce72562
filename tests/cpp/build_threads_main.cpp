// tests/cpp/build_threads_main.cpp -- TEST INFRASTRUCTURE (tests/test_sanitizers.py).
// Builds the flat tree of the product's header-only builder over a seeded cloud with 1 thread and
// with T threads (task-parallel subtrees + parallel_partition) and compares the two node for node.
// Meant to be compiled with -fsanitize=thread (and with address,undefined): the result is the
// process exit code, the sanitizer's report goes to stderr.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pico_tree/internal/flat_tree.hpp"
#include "pico_tree/map.hpp"

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 300000;
  const unsigned threads = argc > 2 ? (unsigned)std::atoi(argv[2]) : 8u;
  std::vector<float> pts(n * 3);
  uint64_t state = 0x9E3779B97F4A7C15ull;
  auto next = [&]() {  // SplitMix64
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  for (size_t i = 0; i < n; ++i) {
    // a lopsided cloud (two dense sheets + noise): the sliding midpoint splits it unevenly, so the
    // builder spawns tasks at many depths; every 16th point repeats its predecessor (ties)
    const float u = (float)(next() >> 40) * 0x1p-24f, v = (float)(next() >> 40) * 0x1p-24f,
                w = (float)(next() >> 40) * 0x1p-24f;
    pts[3 * i + 0] = u * 100.0f;
    pts[3 * i + 1] = v * 100.0f;
    pts[3 * i + 2] = (i % 3 == 0) ? w * 0.01f : (i % 3 == 1 ? 15.0f + w * 0.01f : w * 15.0f);
    if (i % 16 == 15) std::memcpy(&pts[3 * i], &pts[3 * (i - 1)], 12);
  }
  using namespace pico_tree;
  using space_t = space_map<point_map<float const, dynamic_extent>>;
  space_t space(pts.data(), n, 3);
  internal::space_view<space_t> view(space);
  auto one = internal::build_flat_tree<int>(view, max_leaf_size_t(10), bounds_from_space, sliding_midpoint_max_side,
                                            false, 1);
  auto many = internal::build_flat_tree<int>(view, max_leaf_size_t(10), bounds_from_space, sliding_midpoint_max_side,
                                             false, threads);
  if (one.nodes.size() != many.nodes.size() || one.indices != many.indices ||
      std::memcmp(one.nodes.data(), many.nodes.data(), one.nodes.size() * sizeof(one.nodes[0])) != 0) {
    std::fprintf(stderr, "trees differ: %zu vs %zu nodes\n", one.nodes.size(), many.nodes.size());
    return 1;
  }
  std::printf("%zu nodes, depth %u, identical with 1 and %u threads\n", one.nodes.size(), (unsigned)one.max_depth, threads);
  return 0;
}
