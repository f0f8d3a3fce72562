// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A thin extern "C" shim around the *actual reference* pico_tree headers, which
// are compiled where they lie (-I/root/reference/src/pico_tree); no reference
// source is copied into this repository.  The result, oracle/_ref/libptk_ref.so,
// is (i) the ground truth the restatement in oracle/ptk_oracle.cpp is pinned
// against and (ii) the "reference" CPU baseline timed by bench.py.  It is only
// loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
//
// Reference entry points exercised (all paths relative to /root/reference):
//   kd_tree ctor                    src/pico_tree/pico_tree/kd_tree.hpp:76-88
//   kd_tree::search_nn              kd_tree.hpp:126-129
//   kd_tree::search_knn             kd_tree.hpp:169-181 (+ approximate :205-218)
//   kd_tree::search_radius          kd_tree.hpp:257-268 (+ approximate :278-290)
//   kd_tree::search_box             kd_tree.hpp:296-318
//   kd_tree::save                   kd_tree.hpp:367-370 / internal/kd_tree_data.hpp:109-135
//   batch loop shape                src/pyco_tree/pico_tree/_pyco_tree/kd_tree.hpp:117-135
//                                   (#pragma omp parallel for schedule(dynamic, 128))
//   sliding midpoint splitter       internal/kd_tree_builder.hpp:220-280
//
// Build flags are the canonical oracle flags of SURVEY.md section 8(c):
//   g++ -std=c++17 -O3 -ffp-contract=off -fopenmp

#include <omp.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include <pico_tree/kd_tree.hpp>
#include <pico_tree/map_traits.hpp>
#include <pico_tree/vector_traits.hpp>
#include <pico_tree/array_traits.hpp>

// Scalar type of the instantiation: _ref/libptk_ref.so is kd_tree over float points,
// _ref/libptk_ref64.so (-DPTKREF_DOUBLE) the same reference headers over double points.
#ifdef PTKREF_DOUBLE
typedef double scalar_t;
#else
typedef float scalar_t;
#endif

namespace {

using neighbor_t = pico_tree::neighbor<int, scalar_t>;
static_assert(sizeof(neighbor_t) == (sizeof(scalar_t) == 4 ? 8 : 16), "neighbor layout");

constexpr int kChunk = 128;  // _pyco_tree/kd_tree.hpp:94

// A metric identical to metric_l2_squared that counts its two call kinds: the
// scalar overload is called exactly once per visited branch
// (kd_tree_search.hpp:80,84) and the range overload once per visited point
// (kd_tree_search.hpp:58).  Used to pin the oracle's visit counters.
struct counting_l2 {
  using space_category = pico_tree::euclidean_space_tag;
  static thread_local std::uint64_t n_branch;
  static thread_local std::uint64_t n_pts;

  template <typename I1, typename S1, typename I2>
  auto operator()(I1 b1, S1 e1, I2 b2) const {
    ++n_pts;
    return pico_tree::metric_l2_squared()(b1, e1, b2);
  }
  template <typename S>
  S operator()(S x) const {
    ++n_branch;
    return pico_tree::metric_l2_squared()(x);
  }
};
thread_local std::uint64_t counting_l2::n_branch = 0;
thread_local std::uint64_t counting_l2::n_pts = 0;

struct tree_base {
  virtual ~tree_base() = default;
  virtual std::string save() const = 0;
  virtual void knn(scalar_t const* q, size_t nq, size_t k, scalar_t e, bool approx,
                   neighbor_t* out) const = 0;
  virtual void nn(scalar_t const* q, size_t nq, neighbor_t* out) const = 0;
  virtual void radius(scalar_t const* q, size_t nq, scalar_t r, scalar_t e, bool approx,
                      bool sort,
                      std::vector<std::vector<neighbor_t>>& out) const = 0;
  virtual void box(scalar_t const* mins, scalar_t const* maxs, size_t nb,
                   std::vector<std::vector<int>>& out) const = 0;
  size_t dim = 0;
  size_t n = 0;
};

template <size_t Dim_, typename Metric_ = pico_tree::metric_l2_squared>
struct tree_impl final : tree_base {
  using point_t = pico_tree::point_map<scalar_t const, Dim_>;
  using space_t = pico_tree::space_map<point_t>;
  using kd_t = pico_tree::kd_tree<space_t, Metric_>;

  std::vector<scalar_t> pts;  // the driver owns a copy so callers may free theirs
  std::unique_ptr<kd_t> tree;

  space_t make_space(scalar_t const* p, size_t count) const {
    if constexpr (Dim_ == pico_tree::dynamic_extent) {
      return space_t(p, count, dim);
    } else {
      return space_t(p, count);
    }
  }

  tree_impl(scalar_t const* p, size_t count, size_t d, size_t max_leaf) {
    dim = d;
    n = count;
    pts.assign(p, p + count * d);
    tree = std::make_unique<kd_t>(
        make_space(pts.data(), count), pico_tree::max_leaf_size_t(max_leaf));
  }

  std::string save() const override {
    std::stringstream ss(std::ios::in | std::ios::out | std::ios::binary);
    kd_t::save(*tree, ss);
    return ss.str();
  }

  void nn(scalar_t const* q, size_t nq, neighbor_t* out) const override {
    auto query = make_space(q, nq);
    std::ptrdiff_t const count = static_cast<std::ptrdiff_t>(nq);
#pragma omp parallel for schedule(dynamic, kChunk)
    for (std::ptrdiff_t i = 0; i < count; ++i) {
      tree->search_nn(query[static_cast<size_t>(i)], out[i]);
    }
  }

  void knn(scalar_t const* q, size_t nq, size_t k, scalar_t e, bool approx,
           neighbor_t* out) const override {
    auto query = make_space(q, nq);
    std::ptrdiff_t const count = static_cast<std::ptrdiff_t>(nq);
#pragma omp parallel for schedule(dynamic, kChunk)
    for (std::ptrdiff_t i = 0; i < count; ++i) {
      size_t const ui = static_cast<size_t>(i);
      if (approx) {
        tree->search_knn(query[ui], e, out + ui * k, out + ui * k + k);
      } else {
        tree->search_knn(query[ui], out + ui * k, out + ui * k + k);
      }
    }
  }

  void radius(scalar_t const* q, size_t nq, scalar_t r, scalar_t e, bool approx,
              bool sort,
              std::vector<std::vector<neighbor_t>>& out) const override {
    auto query = make_space(q, nq);
    out.resize(nq);
    std::ptrdiff_t const count = static_cast<std::ptrdiff_t>(nq);
#pragma omp parallel for schedule(dynamic, kChunk)
    for (std::ptrdiff_t i = 0; i < count; ++i) {
      size_t const ui = static_cast<size_t>(i);
      if (approx) {
        tree->search_radius(query[ui], r, e, out[ui], sort);
      } else {
        tree->search_radius(query[ui], r, out[ui], sort);
      }
    }
  }

  void box(scalar_t const* mins, scalar_t const* maxs, size_t nb,
           std::vector<std::vector<int>>& out) const override {
    auto qmin = make_space(mins, nb);
    auto qmax = make_space(maxs, nb);
    out.resize(nb);
    std::ptrdiff_t const count = static_cast<std::ptrdiff_t>(nb);
#pragma omp parallel for schedule(dynamic, kChunk)
    for (std::ptrdiff_t i = 0; i < count; ++i) {
      size_t const ui = static_cast<size_t>(i);
      tree->search_box(qmin[ui], qmax[ui], out[ui]);
    }
  }
};

template <typename F>
auto dispatch_dim(size_t dim, F&& f) {
  // Same dispatch as the reference's Python binding: compile-time 2 and 3,
  // run-time otherwise (_pyco_tree/kd_tree.hpp:383-445).
  if (dim == 2) return f(std::integral_constant<size_t, 2>());
  if (dim == 3) return f(std::integral_constant<size_t, 3>());
  return f(std::integral_constant<size_t, pico_tree::dynamic_extent>());
}

struct radius_result {
  std::vector<std::vector<neighbor_t>> rows;
};
struct box_result {
  std::vector<std::vector<int>> rows;
};

}  // namespace

extern "C" {

void* ptkref_create(scalar_t const* pts, size_t n, size_t dim, size_t max_leaf) {
  if (n == 0 || dim == 0 || max_leaf == 0) return nullptr;
  return dispatch_dim(dim, [&](auto d) -> void* {
    return static_cast<tree_base*>(
        new tree_impl<decltype(d)::value>(pts, n, dim, max_leaf));
  });
}

// The same tree searched under another of the reference's metrics
// (0 metric_l2_squared, 1 metric_l1, 2 metric_lpinf; metric.hpp:78-152) or built over a topological
// space (3 metric_so2: dim 1, coordinates in [0, 1]; 4 metric_se2_squared: dim 3, the third
// coordinate in [0, 1]; metric.hpp:186-257 -- kd_tree_node_topological, search_nearest_topological).
void* ptkref_create_metric(scalar_t const* pts, size_t n, size_t dim,
                           size_t max_leaf, int metric) {
  if (n == 0 || dim == 0 || max_leaf == 0) return nullptr;
  if (metric == 0) return ptkref_create(pts, n, dim, max_leaf);
  return dispatch_dim(dim, [&](auto d) -> void* {
    constexpr size_t kDim = decltype(d)::value;
    if (metric == 1)
      return static_cast<tree_base*>(
          new tree_impl<kDim, pico_tree::metric_l1>(pts, n, dim, max_leaf));
    if (metric == 2)
      return static_cast<tree_base*>(
          new tree_impl<kDim, pico_tree::metric_lpinf>(pts, n, dim, max_leaf));
    if (metric == 5)
      return static_cast<tree_base*>(
          new tree_impl<kDim, pico_tree::metric_lninf>(pts, n, dim, max_leaf));
    if (metric == 3 && dim == 1)
      return static_cast<tree_base*>(
          new tree_impl<kDim, pico_tree::metric_so2>(pts, n, dim, max_leaf));
    if (metric == 4 && dim == 3)
      return static_cast<tree_base*>(
          new tree_impl<kDim, pico_tree::metric_se2_squared>(pts, n, dim, max_leaf));
    return nullptr;
  });
}

void ptkref_destroy(void* t) { delete static_cast<tree_base*>(t); }

// Serialises with the reference's own kd_tree::save; returns the byte count.
// Call with buf == nullptr to query the size.
size_t ptkref_save(void* t, unsigned char* buf, size_t cap) {
  std::string s = static_cast<tree_base*>(t)->save();
  if (buf != nullptr && cap >= s.size()) std::memcpy(buf, s.data(), s.size());
  return s.size();
}

void ptkref_set_threads(int threads) {
  if (threads > 0) omp_set_num_threads(threads);
}

int ptkref_max_threads() { return omp_get_max_threads(); }

void ptkref_search_nn(void* t, scalar_t const* q, size_t nq, void* out) {
  static_cast<tree_base*>(t)->nn(q, nq, static_cast<neighbor_t*>(out));
}

void ptkref_search_knn(void* t, scalar_t const* q, size_t nq, size_t k,
                       void* out) {
  static_cast<tree_base*>(t)->knn(q, nq, k, 1.0f, false,
                                  static_cast<neighbor_t*>(out));
}

void ptkref_search_knn_approx(void* t, scalar_t const* q, size_t nq, size_t k,
                              scalar_t e, void* out) {
  static_cast<tree_base*>(t)->knn(q, nq, k, e, true,
                                  static_cast<neighbor_t*>(out));
}

// Ragged radius search. offsets has nq + 1 entries. Returns a handle holding
// the rows; copy them out with ptkref_radius_copy and release it.
void* ptkref_search_radius(void* t, scalar_t const* q, size_t nq, scalar_t radius,
                           int sort, int approx, scalar_t e,
                           std::uint64_t* offsets) {
  auto* r = new radius_result;
  static_cast<tree_base*>(t)->radius(q, nq, radius, e, approx != 0, sort != 0,
                                     r->rows);
  std::uint64_t acc = 0;
  for (size_t i = 0; i < nq; ++i) {
    offsets[i] = acc;
    acc += r->rows[i].size();
  }
  offsets[nq] = acc;
  return r;
}

void ptkref_radius_copy(void* h, void* out) {
  auto* r = static_cast<radius_result*>(h);
  auto* o = static_cast<neighbor_t*>(out);
  for (auto const& row : r->rows) {
    if (!row.empty()) std::memcpy(o, row.data(), row.size() * sizeof(neighbor_t));
    o += row.size();
  }
}

void ptkref_radius_free(void* h) { delete static_cast<radius_result*>(h); }

void* ptkref_search_box(void* t, scalar_t const* mins, scalar_t const* maxs,
                        size_t nb, std::uint64_t* offsets) {
  auto* r = new box_result;
  static_cast<tree_base*>(t)->box(mins, maxs, nb, r->rows);
  std::uint64_t acc = 0;
  for (size_t i = 0; i < nb; ++i) {
    offsets[i] = acc;
    acc += r->rows[i].size();
  }
  offsets[nb] = acc;
  return r;
}

void ptkref_box_copy(void* h, int* out) {
  auto* r = static_cast<box_result*>(h);
  for (auto const& row : r->rows) {
    if (!row.empty()) std::memcpy(out, row.data(), row.size() * sizeof(int));
    out += row.size();
  }
}

void ptkref_box_free(void* h) { delete static_cast<box_result*>(h); }

// Visit counters of the reference traversal for search_knn(k) (k == 1 uses the
// search_knn visitor with a one-element range, which visits exactly what
// search_nn visits): per query, the number of branch nodes expanded and points
// measured.  Single-threaded; builds its own tree with the counting metric.
void ptkref_count_visits(scalar_t const* pts, size_t n, size_t dim,
                         size_t max_leaf, scalar_t const* q, size_t nq, size_t k,
                         std::uint32_t* n_branch, std::uint32_t* n_pts) {
  dispatch_dim(dim, [&](auto d) {
    constexpr size_t D = decltype(d)::value;
    using point_t = pico_tree::point_map<scalar_t const, D>;
    using space_t = pico_tree::space_map<point_t>;
    using kd_t = pico_tree::kd_tree<space_t, counting_l2>;
    auto mk = [&](scalar_t const* p, size_t c) {
      if constexpr (D == pico_tree::dynamic_extent) {
        return space_t(p, c, dim);
      } else {
        return space_t(p, c);
      }
    };
    kd_t tree(mk(pts, n), pico_tree::max_leaf_size_t(max_leaf));
    auto query = mk(q, nq);
    std::vector<neighbor_t> knn(k);
    for (size_t i = 0; i < nq; ++i) {
      counting_l2::n_branch = 0;
      counting_l2::n_pts = 0;
      tree.search_knn(query[i], knn.begin(), knn.end());
      n_branch[i] = static_cast<std::uint32_t>(counting_l2::n_branch);
      n_pts[i] = static_cast<std::uint32_t>(counting_l2::n_pts);
    }
    return 0;
  });
}

// Known-answer hook for the reference's own splitter test
// (test/pico_tree/kd_tree_builder_test.cpp:134-197): runs
// splitter_sliding_midpoint_max_side on 2-D points with the given box.
void ptkref_sliding_midpoint_2d(scalar_t const* pts, size_t n, int* indices,
                                scalar_t const* box_min, scalar_t const* box_max,
                                size_t* split_offset, size_t* split_dim,
                                scalar_t* split_val) {
  using point_t = pico_tree::point_map<scalar_t const, 2>;
  using space_t = pico_tree::space_map<point_t>;
  using wrapper_t = pico_tree::internal::space_wrapper<space_t>;
  using splitter_t =
      pico_tree::internal::splitter_sliding_midpoint_max_side<wrapper_t>;
  space_t space(pts, n);
  wrapper_t wrapper(space);
  splitter_t splitter(wrapper);
  pico_tree::internal::box<scalar_t, 2> box(2);
  for (size_t i = 0; i < 2; ++i) {
    box.min(i) = box_min[i];
    box.max(i) = box_max[i];
  }
  std::vector<int> idx(indices, indices + n);
  std::vector<int>::iterator split;
  pico_tree::size_t sd = 0;
  scalar_t sv = 0;
  splitter(0, idx.begin(), idx.end(), box, split, sd, sv);
  *split_offset = static_cast<size_t>(split - idx.begin());
  *split_dim = sd;
  *split_val = sv;
  std::memcpy(indices, idx.data(), n * sizeof(int));
}

}  // extern "C"
