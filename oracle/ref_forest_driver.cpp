// oracle/ref_forest_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// extern "C" shim around the *actual reference* kd_forest
// (/root/reference/examples/pico_understory/pico_understory/kd_forest.hpp), compiled where it
// lies (-I/root/reference/src/pico_tree -I/root/reference/examples/pico_understory); nothing is
// copied.  Goes into oracle/_ref/libptk_ref_forest.so.  The reference forest draws its
// reflections from std::random_device, so its results are not reproducible: it serves as the
// RECALL yardstick for the product's forest (tests/test_forest.py) and as the CPU baseline of
// tools/bench_forest.py -- never for bit comparisons.
//
//   kd_forest ctor               kd_forest.hpp:42-51  (max_leaf_size, forest_size)
//   kd_forest::search_nn         kd_forest.hpp:78-85
//   kd_forest::search_nearest    kd_forest.hpp:70-76 with a search_knn visitor
//                                (what examples/kd_forest/kd_forest.cpp does for k > 1)

#include <omp.h>

#include <cstdint>
#include <vector>

#include <pico_tree/kd_tree.hpp>
#include <pico_tree/map_traits.hpp>
#include <pico_understory/kd_forest.hpp>

namespace {
using neighbor_t = pico_tree::neighbor<int, float>;
using space_t = pico_tree::space_map<pico_tree::point_map<float const, pico_tree::dynamic_extent>>;
using point_t = pico_tree::point_map<float const, pico_tree::dynamic_extent>;
using forest_t = pico_tree::kd_forest<space_t>;
}  // namespace

extern "C" {

void* ptkref_forest_create(float const* pts, size_t n, size_t dim, size_t max_leaf, size_t forest_size) {
  if (n == 0 || dim == 0 || max_leaf == 0 || forest_size == 0) return nullptr;
  return new forest_t(space_t(pts, n, dim), max_leaf, forest_size);
}

void ptkref_forest_destroy(void* f) { delete static_cast<forest_t*>(f); }

// Row i: the k-list of query i exactly as the reference's shared search_knn visitor leaves it
// (duplicates of one index included -- SURVEY.md 8a row A13).  OpenMP over queries.
void ptkref_forest_search_knn(void* handle, float const* q, size_t nq, size_t dim, size_t k, size_t max_leaves,
                              neighbor_t* out) {
  auto* f = static_cast<forest_t*>(handle);
  long long const count = static_cast<long long>(nq);
#pragma omp parallel for schedule(dynamic, 16)
  for (long long i = 0; i < count; ++i) {
    point_t x(q + static_cast<size_t>(i) * dim, dim);
    neighbor_t* row = out + static_cast<size_t>(i) * k;
    if (k == 1) {
      f->search_nn(x, max_leaves, row[0]);
    } else {
      pico_tree::internal::search_knn<neighbor_t*> v(row, row + k);
      f->search_nearest(x, max_leaves, v);
    }
  }
}

}  // extern "C"
