// tests/cpp/forest_main.cpp -- the C++ kd_forest (include/pico_understory/kd_forest.hpp) driven
// like /root/reference/examples/kd_forest/kd_forest.cpp drives the reference's: build a forest
// over a std::vector of points, query it one point at a time with search_nn and in one batch.
//   forest_main <dir>   reads <dir>/points.bin, <dir>/queries.bin (float32, dim 16), writes results.
#include <array>
#include <cstdio>
#include <string>
#include <vector>

#include <pico_tree/array_traits.hpp>
#include <pico_tree/vector_traits.hpp>
#include <pico_understory/kd_forest.hpp>

using point_t = std::array<float, 16>;

static std::vector<point_t> load(std::string const& path) {
  std::vector<point_t> v;
  if (FILE* f = std::fopen(path.c_str(), "rb")) {
    point_t p;
    while (std::fread(p.data(), sizeof(float), 16, f) == 16) v.push_back(p);
    std::fclose(f);
  }
  return v;
}

template <typename T>
static void dump(std::string const& path, std::vector<T> const& v) {
  FILE* f = std::fopen(path.c_str(), "wb");
  std::fwrite(v.data(), sizeof(T), v.size(), f);
  std::fclose(f);
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::string const dir = argv[1];
  std::vector<point_t> pts = load(dir + "/points.bin");
  std::vector<point_t> qs = load(dir + "/queries.bin");
  using neighbor_t = pico_tree::neighbor<int, float>;
  pico_tree::kd_forest<std::reference_wrapper<std::vector<point_t>>> forest(std::ref(pts), 8, 4, 11);
  std::size_t const leaves = 10;
  std::vector<neighbor_t> one(qs.size()), batch(qs.size()), knn(qs.size() * 5);
  for (std::size_t i = 0; i < qs.size(); ++i) forest.search_nn(qs[i], leaves, one[i]);  // per query
  forest.search_nn(qs, leaves, batch.data());                                            // one batch
  forest.search_knn(qs, 5, leaves, knn.data());
  dump(dir + "/f_one.bin", one);
  dump(dir + "/f_batch.bin", batch);
  dump(dir + "/f_knn.bin", knn);
  return 0;
}
