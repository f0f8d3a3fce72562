// tests/cpp/build_threads_main.cpp -- TEST INFRASTRUCTURE (tests/test_sanitizers.py).
// Builds the flat tree of the product's header-only builder over a seeded cloud with 1 thread and
// with T threads (task-parallel subtrees + parallel_partition) and compares the two node for node.
// Meant to be compiled with -fsanitize=thread (and with address,undefined): the result is the
// process exit code, the sanitizer's report goes to stderr.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <utility>
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
  pts[3 * 7 + 0] = 900.0f;  // two far outliers: the planes of the first levels slide
  pts[3 * 9 + 1] = -700.0f;
  using namespace pico_tree;
  using space_t = space_map<point_map<float const, dynamic_extent>>;
  space_t space(pts.data(), n, 3);
  internal::space_view<space_t> view(space);
  auto one = internal::build_flat_tree<int>(view, max_leaf_size_t(10), bounds_from_space, sliding_midpoint_max_side,
                                            false, 1);
  auto many = internal::build_flat_tree<int>(view, max_leaf_size_t(10), bounds_from_space, sliding_midpoint_max_side,
                                             false, threads);
  if (one.nodes.size() != many.nodes.size() || one.indices != many.indices || one.axis_weight != many.axis_weight ||
      std::memcmp(one.nodes.data(), many.nodes.data(), one.nodes.size() * sizeof(one.nodes[0])) != 0) {
    std::fprintf(stderr, "trees differ: %zu vs %zu nodes\n", one.nodes.size(), many.nodes.size());
    return 1;
  }
  // The top levels split level by level (as the device does it for ptk_tree_create_from_points, with std::partition
  // standing in for the kernels) and the subtrees below built by a pool of workers: the same tree again, outer bounds
  // included.
  for (size_t threshold : {n / 37 + 11, n / 3, n + 1, size_t(10)}) {
    auto ref = internal::build_flat_tree<int>(view, max_leaf_size_t(10), bounds_from_space, sliding_midpoint_max_side,
                                              true, 1);
    std::vector<int> indices(n);
    for (size_t i = 0; i < n; ++i) indices[i] = (int)i;
    std::vector<internal::top_branch<float>> top;
    std::vector<std::pair<size_t, size_t>> frontier;
    auto on_host = [&](std::vector<internal::top_segment<float>>& segments) {
      for (auto& s : segments) {
        const uint32_t a = s.axis;
        const float p = s.plane;
        s.cut = (size_t)(std::partition(indices.begin() + s.begin, indices.begin() + s.end,
                                        [&](int i) { return pts[3 * (size_t)i + a] < p; }) - indices.begin());
      }
      return true;
    };
    size_t slides = 0;
    auto slide = [&](internal::top_segment<float>& s, size_t nth) {
      const uint32_t a = s.axis;
      std::nth_element(indices.begin() + s.begin, indices.begin() + s.begin + nth, indices.begin() + s.end,
                       [&](int i, int j) { return pts[3 * (size_t)i + a] < pts[3 * (size_t)j + a]; });
      s.plane = pts[3 * (size_t)indices[s.begin + nth] + a];
      ++slides;
      return true;
    };
    if (!internal::split_top_levels(ref.root_box, n, threshold, on_host, slide, top, frontier)) {
      std::fprintf(stderr, "threshold %zu: split_top_levels failed\n", threshold);
      return 1;
    }
    auto below = internal::build_flat_tree_below<int>(view, max_leaf_size_t(10), sliding_midpoint_max_side, ref.root_box,
                                                      std::move(indices), top, frontier, true, threads);
    if (ref.nodes.size() != below.nodes.size() || ref.indices != below.indices ||
        std::memcmp(ref.nodes.data(), below.nodes.data(), ref.nodes.size() * sizeof(ref.nodes[0])) != 0 ||
        ref.outer_bounds != below.outer_bounds || ref.max_depth != below.max_depth || ref.leaf_count != below.leaf_count ||
        ref.axis_weight != below.axis_weight || ref.axis_weight.size() != 3 || !(ref.axis_weight[0] > 0.0) ||
        ref.max_leaf_points != below.max_leaf_points) {
      std::fprintf(stderr, "threshold %zu: the tree built below %zu top branches differs\n", threshold, top.size());
      return 1;
    }
    std::printf("threshold %zu: %zu top branches (%zu planes slid), %zu subtrees, identical\n", threshold, top.size(), slides,
                frontier.size());
  }
  std::printf("%zu nodes, depth %u, identical with 1 and %u threads\n", one.nodes.size(), (unsigned)one.max_depth, threads);
  return 0;
}
