#pragma once
//! \file backend.hpp
//! \brief Glue between the header-only host API and the C-ABI HIP backend.
//! \details Everything that crosses into libptk.so goes through the plain C
//! functions of include/ptk.h.  A non-zero status becomes a std::runtime_error
//! carrying ptk_last_error().  There is deliberately NO host fallback here: if
//! the backend cannot be used (library built without a device, no gfx950 GPU)
//! the batched calls throw.  (One case is served on the host: a call the device
//! search REFUSES for a valid tree, PTK_ERR_UNSUPPORTED -- see ptk_unsupported.)

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../ptk.h"
#include "../core.hpp"
#include "../map.hpp"
#include "../metric.hpp"
#include "access.hpp"
#include "flat_tree.hpp"
#include "stream.hpp"

namespace pico_tree::internal {

//! The device search cannot take this tree or batch (PTK_ERR_UNSUPPORTED: a topological tree deeper than the device
//! stack, a dimension beyond the LDS staging).  The batched members THROW it like every other failure of the backend
//! (no device, HIP error, out of memory): the batched path has no CPU fallback.  A caller who wants such a call served
//! the way the reference serves every call -- a loop of the per-query host members over the rows
//! (_pyco_tree/kd_tree.hpp:128) -- says so once per process with pico_tree::allow_host_loop(true); the members then
//! catch the refusal, print one message and run that loop.  Off by default.
struct ptk_unsupported : std::runtime_error {
  using std::runtime_error::runtime_error;
};

inline void ptk_check(int status, char const* what) {
  if (status == PTK_ERR_UNSUPPORTED) {
    throw ptk_unsupported(std::string("pico_tree backend: ") + what + ": " + ptk_last_error());
  }
  if (status != PTK_OK) {
    throw std::runtime_error(
        std::string("pico_tree backend: ") + what + ": " + ptk_last_error());
  }
}

inline std::atomic<bool>& host_loop_flag() {
  static std::atomic<bool> allowed{false};
  return allowed;
}

inline void warn_host_loop(char const* why) {
  static std::once_flag once;
  std::call_once(once, [why] {
    std::fprintf(
        stderr,
        "pico_tree: the device search refused this call (%s); it runs as a loop of the per-query host members "
        "instead (further refusals are served the same way without this message)\n",
        why);
  });
}

//! fn(i) for every row i in [0, n), rows handed out in chunks of 128 (the reference's schedule(dynamic, 128)).
template <typename Fn_>
inline void host_rows_loop(std::size_t n, Fn_ fn) {
  constexpr std::size_t chunk = 128;
  unsigned workers = std::thread::hardware_concurrency();
  workers = workers == 0 ? 1u : workers;
  workers = static_cast<unsigned>(std::min<std::size_t>(workers, (n + chunk - 1) / chunk));
  std::atomic<std::size_t> next{0};
  std::exception_ptr failure;
  std::mutex failure_mutex;
  auto work = [&] {
    try {
      for (;;) {
        std::size_t const lo = next.fetch_add(chunk);
        if (lo >= n) break;
        std::size_t const hi = std::min(n, lo + chunk);
        for (std::size_t i = lo; i < hi; ++i) fn(i);
      }
    } catch (...) {
      std::lock_guard<std::mutex> lock(failure_mutex);
      if (!failure) failure = std::current_exception();
      next.store(n);
    }
  };
  // (a thread that cannot be started is one worker fewer; the threads that did start are joined on every way out)
  std::vector<std::thread> pool;
  struct join_all {
    std::vector<std::thread>& threads;
    ~join_all() {
      for (auto& t : threads)
        if (t.joinable()) t.join();
    }
  } joiner{pool};
  try {
    pool.reserve(workers);
    for (unsigned w = 1; w < workers; ++w) pool.emplace_back(work);
  } catch (std::system_error const&) {
  }
  work();
  for (auto& t : pool) t.join();
  if (failure) std::rethrow_exception(failure);
}

}  // namespace pico_tree::internal

namespace pico_tree {

//! \brief An array of page-locked host memory (ptk_host_alloc) for the query and result arrays of the batched
//! members: `tree.search_knn(queries, k, rows.data())` then moves the rows straight into it, without staging and
//! without the first touch of fresh pages a new std::vector pays on every call.  Keep it and reuse it.
template <typename T_>
class pinned_buffer {
 public:
  pinned_buffer() = default;
  explicit pinned_buffer(std::size_t count) { resize(count); }
  pinned_buffer(pinned_buffer const&) = delete;
  pinned_buffer& operator=(pinned_buffer const&) = delete;
  pinned_buffer(pinned_buffer&& o) noexcept : data_(o.data_), size_(o.size_), capacity_(o.capacity_) {
    o.data_ = nullptr;
    o.size_ = o.capacity_ = 0;
  }
  pinned_buffer& operator=(pinned_buffer&& o) noexcept {
    if (this != &o) {
      ptk_host_free(data_);
      data_ = o.data_, size_ = o.size_, capacity_ = o.capacity_;
      o.data_ = nullptr;
      o.size_ = o.capacity_ = 0;
    }
    return *this;
  }
  ~pinned_buffer() { ptk_host_free(data_); }
  //! Contents are NOT preserved when the buffer has to grow.
  void resize(std::size_t count) {
    static_assert(std::is_trivially_copyable_v<T_>, "plain records only");
    if (count > capacity_) {
      ptk_host_free(data_);
      data_ = nullptr;
      capacity_ = 0;
      void* p = nullptr;
      if (ptk_host_alloc(count * sizeof(T_), &p) != PTK_OK)
        throw std::runtime_error(std::string("pico_tree backend: ptk_host_alloc: ") + ptk_last_error());
      data_ = static_cast<T_*>(p);
      capacity_ = count;
    }
    size_ = count;
  }
  T_* data() { return data_; }
  T_ const* data() const { return data_; }
  std::size_t size() const { return size_; }
  T_& operator[](std::size_t i) { return data_[i]; }
  T_ const& operator[](std::size_t i) const { return data_[i]; }
  T_* begin() { return data_; }
  T_* end() { return data_ + size_; }

 private:
  T_* data_ = nullptr;
  std::size_t size_ = 0, capacity_ = 0;
};

}  // namespace pico_tree

namespace pico_tree::internal {

//! PTK_METRIC_* of a metric type the backend knows; -1 otherwise.
template <typename Metric_>
inline constexpr int ptk_metric_v = std::is_same_v<Metric_, metric_l2_squared> ? PTK_METRIC_L2_SQUARED
                                    : std::is_same_v<Metric_, metric_l1>       ? PTK_METRIC_L1
                                    : std::is_same_v<Metric_, metric_lpinf>    ? PTK_METRIC_LPINF
                                    : std::is_same_v<Metric_, metric_lninf>    ? PTK_METRIC_LNINF
                                    : std::is_same_v<Metric_, metric_so2>      ? PTK_METRIC_SO2
                                    : std::is_same_v<Metric_, metric_se2_squared> ? PTK_METRIC_SE2_SQUARED
                                                                               : -1;
//! The topological metrics (the tree carries four bounds per branch).
template <typename Metric_>
inline constexpr bool ptk_topological_v =
    std::is_same_v<Metric_, metric_so2> || std::is_same_v<Metric_, metric_se2_squared>;

//! Which kd_tree instantiations run on the GPU: float points through ptk_* and double points
//! through ptk_tree64_* / ptk_search64_* (include/ptk.h).
template <typename Metric_, typename Scalar_, typename Index_>
inline constexpr bool is_accelerated_v =
    ptk_metric_v<Metric_> >= 0 &&
    (std::is_same_v<Scalar_, float> || std::is_same_v<Scalar_, double>) &&
    std::is_same_v<Index_, int> && sizeof(int) == 4;

//! The C entry points of one scalar type under one set of names.
template <typename Scalar_>
struct ptk_api;
template <>
struct ptk_api<float> {
  using tree = ptk_tree;
  using neighbor = ptk_neighbor;
  static void destroy(tree* t) { ptk_tree_destroy(t); }
  static int set_metric(tree* t, int m) { return ptk_tree_set_metric(t, m); }
  static int knn(tree const* t, float const* q, std::uint64_t nq, std::uint32_t k, float e, neighbor* out) {
    return ptk_search_knn(t, q, nq, k, e, out);
  }
  static int radius(tree const* t, float const* q, std::uint64_t nq, float r, float e, int sort,
                    std::uint64_t* offsets, neighbor** out) {
    return ptk_search_radius(t, q, nq, r, e, sort, offsets, out);
  }
  static int box(tree const* t, float const* lo, float const* hi, std::uint64_t nb, std::uint64_t* offsets,
                 std::int32_t** out) {
    return ptk_search_box(t, lo, hi, nb, offsets, out);
  }
};
template <>
struct ptk_api<double> {
  using tree = ptk_tree64;
  using neighbor = ptk_neighbor64;
  static void destroy(tree* t) { ptk_tree64_destroy(t); }
  static int set_metric(tree* t, int m) { return ptk_tree64_set_metric(t, m); }
  static int knn(tree const* t, double const* q, std::uint64_t nq, std::uint32_t k, double e, neighbor* out) {
    return ptk_search64_knn(t, q, nq, k, e, out);
  }
  static int radius(tree const* t, double const* q, std::uint64_t nq, double r, double e, int sort,
                    std::uint64_t* offsets, neighbor** out) {
    return ptk_search64_radius(t, q, nq, r, e, sort, offsets, out);
  }
  static int box(tree const* t, double const* lo, double const* hi, std::uint64_t nb, std::uint64_t* offsets,
                 std::int32_t** out) {
    return ptk_search64_box(t, lo, hi, nb, offsets, out);
  }
};

//! True for space types whose points are known to be one contiguous row-major
//! float matrix, so a batch can be handed over without a gather copy.
template <typename Space_>
struct is_dense_space : std::false_type {};
template <typename Scalar_, size_t Dim_>
struct is_dense_space<space_map<point_map<Scalar_, Dim_>>> : std::true_type {};
template <typename Scalar_, std::size_t Dim_, typename Alloc_>
struct is_dense_space<std::vector<std::array<Scalar_, Dim_>, Alloc_>> : std::true_type {};

//! Row-major matrix view of any space: zero-copy when dense, gathered otherwise.
template <typename Space_>
class dense_rows {
 public:
  using scalar = typename space_view<Space_>::scalar_type;
  explicit dense_rows(Space_ const& space) {
    space_view<Space_> view(space);
    n_ = view.size();
    dim_ = view.sdim();
    if constexpr (is_dense_space<Space_>::value) {
      data_ = n_ > 0 ? view[size_t(0)] : nullptr;
    } else {
      owned_.resize(n_ * dim_);
      for (size_t i = 0; i < n_; ++i) {
        scalar const* p = view[i];
        for (size_t d = 0; d < dim_; ++d) owned_[i * dim_ + d] = p[d];
      }
      data_ = owned_.data();
    }
  }
  scalar const* data() const { return data_; }
  size_t rows() const { return n_; }
  size_t cols() const { return dim_; }

 private:
  std::vector<scalar> owned_;
  scalar const* data_ = nullptr;
  size_t n_ = 0;
  size_t dim_ = 0;
};

//! Owns the device-side replica of one flat tree; created lazily, shared by
//! moves of the owning kd_tree.
template <typename Scalar_>
class device_tree {
 public:
  using api = ptk_api<Scalar_>;
  using handle_type = typename api::tree;
  device_tree() : state_(std::make_shared<state>()) {}

  template <typename Tree_, typename SpaceView_>
  handle_type* get(Tree_ const& tree, SpaceView_ const& space, int metric = PTK_METRIC_L2_SQUARED) const {
    std::lock_guard<std::mutex> lock(state_->mutex);
    if (state_->handle == nullptr) {
      size_t const n = space.size();
      size_t const dim = space.sdim();
      std::vector<Scalar_> pts(n * dim);
      for (size_t i = 0; i < n; ++i) {
        Scalar_ const* p = space[i];
        for (size_t d = 0; d < dim; ++d) pts[i * dim + d] = p[d];
      }
      handle_type* h = nullptr;
      if constexpr (std::is_same_v<Scalar_, float>) {
        static_assert(sizeof(typename Tree_::node_type) == sizeof(ptk_node), "node layout");
        ptk_tree_desc desc{};
        desc.dim = static_cast<std::uint32_t>(dim);
        desc.n_points = n;
        desc.points = pts.data();
        desc.n_nodes = tree.nodes.size();
        desc.nodes = reinterpret_cast<ptk_node const*>(tree.nodes.data());
        desc.indices = tree.indices.data();
        desc.root_min = tree.root_box.min();
        desc.root_max = tree.root_box.max();
        desc.max_depth = tree.max_depth;
        desc.device = PTK_DEVICE_CURRENT;
        ptk_check(ptk_tree_create(&desc, &h), "ptk_tree_create");
        if (metric == PTK_METRIC_SO2 || metric == PTK_METRIC_SE2_SQUARED) {  // the other two bounds of every branch
          static_assert(sizeof(tree.outer_bounds[0]) == 2 * sizeof(float), "outer bounds layout");
          int const rc = tree.outer_bounds.size() == tree.nodes.size()
                             ? ptk_tree_set_outer_bounds(h, tree.outer_bounds[0].data(), tree.nodes.size())
                             : PTK_ERR_INVALID;
          if (rc != PTK_OK) ptk_tree_destroy(h);
          ptk_check(rc, "ptk_tree_set_outer_bounds");
        }
      } else {
        // The already built tree crosses the boundary in the reference's own stream format.
        std::ostringstream os(std::ios::out | std::ios::binary);
        write_flat_tree(tree, os);
        std::string const bytes = os.str();
        // (a tree over a topological space writes four bounds per branch: write_flat_tree, keep_outer_bounds)
        if (tree.keep_outer_bounds) {
          ptk_check(
              ptk_tree64_create_from_topological_stream(
                  pts.data(), n, static_cast<std::uint32_t>(dim), bytes.data(), bytes.size(), PTK_DEVICE_CURRENT, &h),
              "ptk_tree64_create_from_topological_stream");
        } else {
          ptk_check(
              ptk_tree64_create_from_stream(
                  pts.data(), n, static_cast<std::uint32_t>(dim), bytes.data(), bytes.size(), PTK_DEVICE_CURRENT, &h),
              "ptk_tree64_create_from_stream");
        }
      }
      if (metric != PTK_METRIC_L2_SQUARED) {
        int const rc = api::set_metric(h, metric);
        if (rc != PTK_OK) api::destroy(h);
        ptk_check(rc, "ptk_tree_set_metric");
      }
      state_->handle = h;
      state_->destroy = &api::destroy;
    }
    return state_->handle;
  }

 private:
  struct state {
    std::mutex mutex;
    handle_type* handle = nullptr;
    // Set where the handle is made, so that a translation unit that never issues
    // a batched call does not reference (and need not link) libptk.
    void (*destroy)(handle_type*) = nullptr;
    ~state() {
      if (handle != nullptr && destroy != nullptr) destroy(handle);
    }
  };
  std::shared_ptr<state> state_;
};

}  // namespace pico_tree::internal
