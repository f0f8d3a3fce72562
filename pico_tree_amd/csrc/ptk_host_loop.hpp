// ptk_host_loop.hpp -- ptk_host_search_*: the reference's own batch loop on the host, for the calls the DEVICE search
// refuses (PTK_ERR_UNSUPPORTED: a topological tree deeper than the device stack, a dimension beyond the LDS staging).
//
// This is NOT a fallback of ptk_search_*: those entry points never come here -- without a usable device, or on any
// HIP failure, they fail (PTK_ERR_DEVICE) and the caller sees it.  The wrappers (pico_tree_amd.KdTree; the C++ batched
// members do the same with their own per-query members) call these entry points by name, and only after a device search
// has answered PTK_ERR_UNSUPPORTED for a valid tree, with a warning: that call is then served the way the reference
// serves every call -- a loop of per-query searches over the rows, `schedule(dynamic, 128)`
// (_pyco_tree/kd_tree.hpp:117-135, :179-200, :245-268) -- by the header-only host traversal of
// include/pico_tree/internal/flat_search.hpp on the flat tree the handle keeps.  The points are the caller's: a handle
// holds them on the device only.
//
// Included by ptk_backend.hip (needs ptk_tree and fail()).
#pragma once

#include <atomic>
#include <exception>
#include <system_error>
#include <thread>
#include <vector>

#include "pico_tree/internal/flat_search.hpp"
#include "pico_tree/internal/visitors.hpp"
#include "pico_tree/metric.hpp"

namespace ptk_host {

using namespace pico_tree;
using flat_t = internal::flat_tree<int, float, dynamic_extent>;
using space_t = space_map<point_map<float const, dynamic_extent>>;
using view_t = internal::space_view<space_t>;
using query_t = internal::point_view<point_map<float const, dynamic_extent>>;
using neighbor_t = neighbor<int, float>;

inline flat_t make_flat(const ptk_tree* t) {
  static_assert(sizeof(flat_t::node_type) == sizeof(ptk_node), "node layout");
  flat_t flat(t->dim);
  flat.indices.assign(t->indices.begin(), t->indices.end());
  flat.nodes.resize(t->nodes.size());
  std::memcpy(static_cast<void*>(flat.nodes.data()), t->nodes.data(), t->nodes.size() * sizeof(ptk_node));
  for (uint32_t d = 0; d < t->dim; ++d) {
    flat.root_box.min(d) = t->root_min[d];
    flat.root_box.max(d) = t->root_max[d];
  }
  flat.max_depth = t->max_depth;
  if (t->outer.size() == 2 * t->nodes.size()) {
    flat.keep_outer_bounds = true;
    flat.outer_bounds.resize(t->nodes.size());
    std::memcpy(static_cast<void*>(flat.outer_bounds.data()), t->outer.data(), t->outer.size() * sizeof(float));
  }
  return flat;
}
// The handle's flat view: built by the first host search, shared by the later ones (and by concurrent ones).
inline std::shared_ptr<const flat_t> flat_of(const ptk_tree* t) {
  std::lock_guard<std::mutex> lock(t->host_flat_mutex);
  if (!t->host_flat) t->host_flat = std::make_shared<const flat_t>(make_flat(t));
  return std::static_pointer_cast<const flat_t>(t->host_flat);
}

// fn(i) for every row, rows handed out 128 at a time.
template <class Fn>
inline void rows_loop(uint64_t n, Fn fn) {
  constexpr uint64_t chunk = 128;
  unsigned workers = std::thread::hardware_concurrency();
  workers = workers == 0 ? 1u : workers;
  workers = (unsigned)std::min<uint64_t>(workers, (n + chunk - 1) / chunk);
  std::atomic<uint64_t> next{0};
  std::exception_ptr failure;
  std::mutex failure_mutex;
  auto work = [&] {
    try {
      for (;;) {
        const uint64_t lo = next.fetch_add(chunk);
        if (lo >= n) break;
        const uint64_t hi = std::min(n, lo + chunk);
        for (uint64_t i = lo; i < hi; ++i) fn(i);
      }
    } catch (...) {
      std::lock_guard<std::mutex> lock(failure_mutex);
      if (!failure) failure = std::current_exception();
      next.store(n);
    }
  };
  // (a thread that cannot be started -- a limit of the process -- is simply one worker fewer: the rows are dealt out in
  // chunks, whoever is there takes them; and the threads that did start are joined on every way out)
  std::vector<std::thread> pool;
  struct join_all {
    std::vector<std::thread>& threads;
    ~join_all() {
      for (auto& th : threads)
        if (th.joinable()) th.join();
    }
  } joiner{pool};
  try {
    pool.reserve(workers);
    for (unsigned w = 1; w < workers; ++w) pool.emplace_back(work);
  } catch (const std::system_error&) {
  }
  work();
  for (auto& th : pool) th.join();
  if (failure) std::rethrow_exception(failure);
}

// One traversal of query row x with visitor v under the handle's metric.
template <class Visitor>
inline void search_one(const ptk_tree* t, const flat_t& flat, const view_t& view, const float* x, Visitor& v) {
  const point_map<float const, dynamic_extent> row(x, t->dim);  // (the view refers to it)
  query_t q(row);
  switch (t->metric.load()) {
    case PTK_METRIC_L1: internal::nearest_search(flat, view, metric_l1(), q, v); break;
    case PTK_METRIC_LPINF: internal::nearest_search(flat, view, metric_lpinf(), q, v); break;
    case PTK_METRIC_LNINF: internal::nearest_search(flat, view, metric_lninf(), q, v); break;
    case PTK_METRIC_SO2: {
      metric_so2 m;
      internal::nearest_search_topological<flat_t, view_t, metric_so2, query_t, Visitor>(flat, view, m, q, v)();
    } break;
    case PTK_METRIC_SE2_SQUARED: {
      metric_se2_squared m;
      internal::nearest_search_topological<flat_t, view_t, metric_se2_squared, query_t, Visitor>(flat, view, m, q, v)();
    } break;
    default: internal::nearest_search(flat, view, metric_l2_squared(), q, v); break;
  }
}

inline bool topological_without_bounds(const ptk_tree* t, const flat_t& flat) {
  const int m = t->metric.load();
  return (m == PTK_METRIC_SO2 || m == PTK_METRIC_SE2_SQUARED) && !flat.keep_outer_bounds;
}

}  // namespace ptk_host

extern "C" {

int ptk_host_search_knn(const ptk_tree* t, const float* points, const float* q, uint64_t nq, uint32_t k, float e,
                        ptk_neighbor* out) {
  if (t == nullptr || points == nullptr || (nq > 0 && (q == nullptr || out == nullptr)))
    return fail(PTK_ERR_INVALID, "null argument");
  if (k == 0 || !(e > 0.0f)) return fail(PTK_ERR_INVALID, "k must be >= 1 and e > 0");
  try {
    using namespace ptk_host;
    const std::shared_ptr<const flat_t> flat_holder = flat_of(t);
    const flat_t& flat = *flat_holder;
    if (topological_without_bounds(t, flat)) return fail(PTK_ERR_INVALID, "this tree has no outer bounds (ptk_tree_set_outer_bounds)");
    space_t space(points, t->n_points, t->dim);
    view_t view(space);
    auto* rows = reinterpret_cast<neighbor_t*>(out);
    rows_loop(nq, [&](uint64_t i) {
      neighbor_t* b = rows + i * k;
      if (e == 1.0f) {
        internal::knn_visitor<neighbor_t*> v(b, b + k);
        search_one(t, flat, view, q + i * t->dim, v);
      } else {
        internal::knn_visitor<neighbor_t*, true> v(b, b + k, e);
        search_one(t, flat, view, q + i * t->dim, v);
      }
    });
  } catch (const std::bad_alloc&) {
    return fail(PTK_ERR_NOMEM, "out of host memory");
  } catch (const std::exception& ex) {
    return fail(PTK_ERR_INVALID, "host search failed: %s", ex.what());
  }
  return PTK_OK;
}

int ptk_host_search_radius(const ptk_tree* t, const float* points, const float* q, uint64_t nq, float radius, float e,
                           int sort, uint64_t* offsets, ptk_neighbor** out) {
  if (t == nullptr || points == nullptr || offsets == nullptr || out == nullptr || (nq > 0 && q == nullptr))
    return fail(PTK_ERR_INVALID, "null argument");
  if (!(e > 0.0f)) return fail(PTK_ERR_INVALID, "approximation ratio e must be > 0");
  *out = nullptr;
  try {
    using namespace ptk_host;
    const std::shared_ptr<const flat_t> flat_holder = flat_of(t);
    const flat_t& flat = *flat_holder;
    if (topological_without_bounds(t, flat)) return fail(PTK_ERR_INVALID, "this tree has no outer bounds (ptk_tree_set_outer_bounds)");
    space_t space(points, t->n_points, t->dim);
    view_t view(space);
    std::vector<std::vector<neighbor_t>> per_row(nq);
    rows_loop(nq, [&](uint64_t i) {
      if (e == 1.0f) {
        internal::radius_visitor<neighbor_t> v(radius, per_row[i]);
        search_one(t, flat, view, q + i * t->dim, v);
        if (sort) v.sort();
      } else {
        internal::radius_visitor<neighbor_t, true> v(radius, per_row[i], e);
        search_one(t, flat, view, q + i * t->dim, v);
        if (sort) v.sort();
      }
    });
    offsets[0] = 0;
    for (uint64_t i = 0; i < nq; ++i) offsets[i + 1] = offsets[i] + per_row[i].size();
    auto* rows = static_cast<neighbor_t*>(std::malloc(std::max<size_t>(offsets[nq], 1) * sizeof(neighbor_t)));
    if (rows == nullptr) return fail(PTK_ERR_NOMEM, "out of host memory");
    for (uint64_t i = 0; i < nq; ++i) std::copy(per_row[i].begin(), per_row[i].end(), rows + offsets[i]);
    *out = reinterpret_cast<ptk_neighbor*>(rows);
  } catch (const std::bad_alloc&) {
    return fail(PTK_ERR_NOMEM, "out of host memory");
  } catch (const std::exception& ex) {
    return fail(PTK_ERR_INVALID, "host search failed: %s", ex.what());
  }
  return PTK_OK;
}

int ptk_host_search_box(const ptk_tree* t, const float* points, const float* mins, const float* maxs, uint64_t nb,
                        uint64_t* offsets, int32_t** out) {
  if (t == nullptr || points == nullptr || offsets == nullptr || out == nullptr || (nb > 0 && (mins == nullptr || maxs == nullptr)))
    return fail(PTK_ERR_INVALID, "null argument");
  *out = nullptr;
  try {
    using namespace ptk_host;
    const std::shared_ptr<const flat_t> flat_holder = flat_of(t);
    const flat_t& flat = *flat_holder;
    space_t space(points, t->n_points, t->dim);
    view_t view(space);
    std::vector<std::vector<int>> per_row(nb);
    const int metric = t->metric.load();
    if (metric == PTK_METRIC_SO2 || metric == PTK_METRIC_SE2_SQUARED) {
      // a tree over a topological space: the metric_box_map query of the reference (box.hpp:300-376) -- intervals
      // through the seam of the circle axis, the four-bound intersection tests
      if (topological_without_bounds(t, flat)) return fail(PTK_ERR_INVALID, "this tree has no outer bounds (ptk_tree_set_outer_bounds)");
      const metric_so2 so2;
      const metric_se2_squared se2;
      rows_loop(nb, [&](uint64_t i) {
        if (metric == PTK_METRIC_SO2)
          internal::box_search<true>(flat, view, mins + i * t->dim, maxs + i * t->dim, per_row[i],
                                     internal::circle_axes_of<metric_so2>{so2});
        else
          internal::box_search<true>(flat, view, mins + i * t->dim, maxs + i * t->dim, per_row[i],
                                     internal::circle_axes_of<metric_se2_squared>{se2});
      });
    } else {
      rows_loop(nb, [&](uint64_t i) { internal::box_search(flat, view, mins + i * t->dim, maxs + i * t->dim, per_row[i]); });
    }
    offsets[0] = 0;
    for (uint64_t i = 0; i < nb; ++i) offsets[i + 1] = offsets[i] + per_row[i].size();
    auto* rows = static_cast<int32_t*>(std::malloc(std::max<size_t>(offsets[nb], 1) * sizeof(int32_t)));
    if (rows == nullptr) return fail(PTK_ERR_NOMEM, "out of host memory");
    for (uint64_t i = 0; i < nb; ++i) std::copy(per_row[i].begin(), per_row[i].end(), rows + offsets[i]);
    *out = rows;
  } catch (const std::bad_alloc&) {
    return fail(PTK_ERR_NOMEM, "out of host memory");
  } catch (const std::exception& ex) {
    return fail(PTK_ERR_INVALID, "host search failed: %s", ex.what());
  }
  return PTK_OK;
}

}  // extern "C"
