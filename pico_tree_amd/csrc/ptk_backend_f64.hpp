// ptk_backend_f64.hpp -- host side of the double-precision entry points of include/ptk.h
// (ptk_tree64_* / ptk_search64_*).  Included at the end of ptk_backend.hip: one translation
// unit, so it shares the error channel, the HIP helpers and the host builder set-up.
//
// Same responsibilities as the float32 side: build (or load) the flat tree on the host with the
// product's own builder instantiated over double, re-encode it for the kernels of
// ptk_kernels_f64.hpp, keep it in HBM, launch, move results.  No CPU search path.

#pragma once

#include "ptk_kernels_f64.hpp"
#include "ptk_kernels_coop64.hpp"

static_assert(sizeof(ptk_neighbor64) == 16 && offsetof(ptk_neighbor64, distance) == 8, "neighbor<int, double> layout");

struct ptk_tree64 {
  using flat_t = pico_tree::internal::flat_tree<int, double, pico_tree::dynamic_extent>;
  uint32_t dim = 0;
  uint64_t n_points = 0;
  flat_t flat{1};  // host copy (DFS pre-order stream); its own outer_bounds stay empty:
  std::vector<std::array<double, 2>> outer;  // per stream node {left_min, right_max} (topological metrics); may be empty
  uint64_t n_leaves = 0;
  uint32_t max_depth = 0;
  uint32_t max_leaf_count = 0;

  int device = kDeviceNone;
  ptk::DevTree64 dev{};
  void* d_nodes = nullptr;
  void* d_pts = nullptr;
  void* d_index = nullptr;
  void* d_ranges = nullptr;
  void* d_root = nullptr;  // root box: min[dim], max[dim]
  void* d_outer = nullptr; // topological metrics only: double2 per branch (made by ptk_tree64_set_metric)
  void* d_occ = nullptr;   // which cells of the coarse Morton grid hold tree points (morton64_kernel): k-NN batches start
                           // their long searches first
  uint64_t device_bytes = 0;
  uint32_t slots = 0;      // stack records a lane may need: 2 * depth + 4
  int cus = 256;           // compute units of the device
  // counters of the last capped k-NN call (ptk_tree64_debug_knn_counts): they live in the stack block
  mutable uint32_t* last_meta = nullptr;
  std::atomic<int> metric{PTK_METRIC_L2_SQUARED};

  // The record stacks of a launch live in one grow-only HBM block; calls on one handle enqueue under
  // `mutex` and the block is reused in stream order (`done` orders a call on another stream).
  mutable std::mutex mutex;
  mutable char* stack = nullptr;
  mutable size_t stack_capacity = 0;
  mutable hipEvent_t done = nullptr;
  mutable hipStream_t last_stream = nullptr;
  mutable bool has_work = false;

  // The device copies of a HOST-buffer call's queries / counts / offsets (block 0) and rows (block 1): kept on the
  // handle up to kIoKeepBytes each (a call of 2 000 queries spent a third of its time in hipMalloc / hipFree), allocated
  // per call beyond that.  `io_mutex` is held for the whole of such a call.
  mutable std::mutex io_mutex;
  mutable char* io[2] = {nullptr, nullptr};
  mutable size_t io_capacity[2] = {0, 0};
};

namespace {

// Stack block of one launch; larger batches go through in pieces (PTK_STACK64_MB shrinks it for tests).
// Default 8 GiB: BASELINE config 2 in one launch (7.2 M queries x 68 slots x 16 B = 7.8 GB; every
// launch ends with the tail of its slowest queries, so pieces cost time: DESIGN.md section 4, K9).  The block is allocated at the size the largest batch so far needed, not up front.
size_t stack64_bytes() { return (size_t)std::max(1, env_int("PTK_STACK64_MB", 8192)) << 20; }

int encode64(ptk_tree64& t, const double* points) {
  ptk::TreeStats st;
  ptk::EncodedTree64 enc;
  bool unsupported = false;
  std::string err = ptk::encode_tree64(t.dim, t.n_points, points, t.flat.nodes.data(), t.flat.nodes.size(),
                                       t.flat.indices.data(), st, enc, unsupported, t.device != kDeviceNone);
  if (!err.empty()) return fail(unsupported ? PTK_ERR_UNSUPPORTED : PTK_ERR_INVALID, "%s", err.c_str());
  t.n_leaves = st.n_leaves;
  t.max_depth = st.max_depth;
  t.max_leaf_count = st.max_leaf_count;
  t.slots = 2 * st.max_depth + 4;
  if (t.device == kDeviceNone) return PTK_OK;

  static_assert(sizeof(ptk::EncNode64) == sizeof(ptk::Node64), "records");
  std::vector<double> root(2 * (size_t)t.dim);
  std::memcpy(root.data(), t.flat.root_box.min(), t.dim * sizeof(double));
  std::memcpy(root.data() + t.dim, t.flat.root_box.max(), t.dim * sizeof(double));
  const size_t nb = enc.nodes.size() * sizeof(ptk::Node64), pb = enc.points.size() * sizeof(double),
               ib = t.flat.indices.size() * sizeof(int32_t), rb = enc.ranges.size() * sizeof(ptk::EncRange),
               bb = root.size() * sizeof(double);
  PTK_HIP(hipMalloc(&t.d_nodes, nb));
  PTK_HIP(hipMalloc(&t.d_pts, pb));
  PTK_HIP(hipMalloc(&t.d_index, ib));
  PTK_HIP(hipMalloc(&t.d_ranges, rb));
  PTK_HIP(hipMalloc(&t.d_root, bb));
  PTK_HIP(hipMemcpy(t.d_nodes, enc.nodes.data(), nb, hipMemcpyHostToDevice));
  PTK_HIP(hipMemcpy(t.d_pts, enc.points.data(), pb, hipMemcpyHostToDevice));
  PTK_HIP(hipMemcpy(t.d_index, t.flat.indices.data(), ib, hipMemcpyHostToDevice));
  PTK_HIP(hipMemcpy(t.d_ranges, enc.ranges.data(), rb, hipMemcpyHostToDevice));
  PTK_HIP(hipMemcpy(t.d_root, root.data(), bb, hipMemcpyHostToDevice));
  {  // the occupancy of the coarse grid (the key arithmetic of morton64_kernel, on the host: once per tree)
    std::vector<uint8_t> occ(size_t(1) << ptk::kMorton64CellBits, 0);
    double lo[3] = {0, 0, 0}, inv[3] = {0, 0, 0};
    for (uint32_t d = 0; d < t.dim && d < 3; ++d) {
      lo[d] = t.flat.root_box.min()[d];
      const double ext = t.flat.root_box.max()[d] - t.flat.root_box.min()[d];
      inv[d] = ext > 0 ? 1024.0 / ext : 0.0;
    }
    auto spread = [](uint32_t v) {  // spread10 of ptk_kernels.hpp
      v &= 0x3FFu;
      v = (v | (v << 16)) & 0x030000FFu;
      v = (v | (v << 8)) & 0x0300F00Fu;
      v = (v | (v << 4)) & 0x030C30C3u;
      v = (v | (v << 2)) & 0x09249249u;
      return v;
    };
    for (uint64_t i = 0; i < t.n_points; ++i) {
      uint32_t key = 0;
      for (uint32_t a = 0; a < 3 && a < t.dim; ++a) {
        const double f = std::fmin(std::fmax((points[i * t.dim + a] - lo[a]) * inv[a], 0.0), 1023.0);
        key |= spread((uint32_t)f) << a;
      }
      occ[key >> (30u - ptk::kMorton64CellBits)] = 1;
    }
    PTK_HIP(hipMalloc(&t.d_occ, occ.size()));
    PTK_HIP(hipMemcpy(t.d_occ, occ.data(), occ.size(), hipMemcpyHostToDevice));
  }
  t.device_bytes = nb + pb + ib + rb + bb + (size_t(1) << ptk::kMorton64CellBits);
  t.dev.nodes = static_cast<const ptk::Node64*>(t.d_nodes);
  t.dev.pts = static_cast<const double*>(t.d_pts);
  t.dev.index = static_cast<const int32_t*>(t.d_index);
  t.dev.ranges = static_cast<const uint2*>(t.d_ranges);
  t.dev.root_ref = enc.root_ref;
  t.dev.cbits = enc.cbits;
  t.dev.cmask = (1u << enc.cbits) - 1u;
  t.dev.dim = t.dim;
  t.dev.stride = enc.stride;
  t.dev.n_points = (uint32_t)t.n_points;
  return PTK_OK;
}

int finish_create64(ptk_tree64* t, const double* points, int32_t device, ptk_tree64** out) {
  int dev = kDeviceNone;
  if (device != kDeviceNone) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
      delete t;
      return fail(PTK_ERR_DEVICE, "no HIP device is visible");
    }
    dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev >= count) {
      delete t;
      return fail(PTK_ERR_INVALID, "device %d out of range (%d visible)", dev, count);
    }
  }
  t->device = dev;
  if (dev != kDeviceNone) {
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n_cu > 0) t->cus = n_cu;
  }
  int rc;
  if (dev == kDeviceNone) {
    rc = encode64(*t, points);
  } else {
    DeviceGuard guard(dev);
    rc = guard.ok ? encode64(*t, points) : fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", dev);
  }
  if (rc != PTK_OK) {
    ptk_tree64_destroy(t);
    return rc;
  }
  *out = t;
  return PTK_OK;
}

int check_search64(const ptk_tree64* t, const void* q, uint64_t nq) {
  if (t == nullptr) return fail(PTK_ERR_INVALID, "null tree");
  if (nq > 0 && q == nullptr) return fail(PTK_ERR_INVALID, "null query buffer");
  if (t->device == kDeviceNone) return fail(PTK_ERR_DEVICE, "this handle has no device replica");
  return PTK_OK;
}

// A device buffer of a host-buffer call: block `which` of the handle when it fits what a handle keeps, else an allocation
// of the call's own (freed by the destructor).  The caller holds t->io_mutex.
constexpr size_t kIoKeepBytes = size_t(64) << 20;
struct IoBuffer {
  char* p = nullptr;
  bool own = false;
  IoBuffer() = default;
  IoBuffer(const IoBuffer&) = delete;
  IoBuffer& operator=(const IoBuffer&) = delete;
  ~IoBuffer() {
    if (own && p) (void)hipFree(p);
  }
  hipError_t get(const ptk_tree64* t, int which, size_t bytes) {
    bytes = std::max<size_t>(bytes, 256);
    if (bytes <= kIoKeepBytes) {
      if (bytes > t->io_capacity[which]) {
        if (t->io[which]) (void)hipFree(t->io[which]);
        t->io[which] = nullptr;
        t->io_capacity[which] = 0;
        const size_t want = std::min(kIoKeepBytes, std::max(bytes * 2, size_t(1) << 20));
        const hipError_t he = hipMalloc((void**)&t->io[which], want);
        if (he != hipSuccess) return he;
        t->io_capacity[which] = want;
      }
      p = t->io[which];
      own = false;
      return hipSuccess;
    }
    own = true;
    return hipMalloc((void**)&p, bytes);
  }
};

// Pieces of a batch that fit the stack block; *stack is valid on `s` after the call.
struct Stack64Lease {
  const ptk_tree64* t;
  std::unique_lock<std::mutex> lock;
  hipStream_t s;
  bool armed = false;
  Stack64Lease(const ptk_tree64* tree, hipStream_t stream) : t(tree), lock(tree->mutex), s(stream) {}
  uint64_t piece = 0;  // queries per launch
  char* aux = nullptr; // aux_bytes of scratch for the caller (the launch-order permutation)
  ptk::Rec64* stack = nullptr;
  int acquire(uint64_t n, size_t aux_bytes = 0) {
    t->last_meta = nullptr;  // (the aux block is about to be written again: the last capped call's counters go with it)
    const size_t per_block = (size_t)t->slots * 64 * sizeof(ptk::Rec64);
    uint64_t blocks = (n + 63) / 64;
    const uint64_t max_blocks = std::max<uint64_t>(1, stack64_bytes() / per_block);
    if (blocks > max_blocks) blocks = max_blocks;
    aux_bytes = (aux_bytes + 255) & ~size_t(255);
    const size_t bytes = blocks * per_block + aux_bytes;
    if (bytes > t->stack_capacity) {
      if (t->has_work) (void)hipEventSynchronize(t->done);
      if (t->stack) (void)hipFree(t->stack);
      t->stack = nullptr;
      t->last_meta = nullptr;
      t->stack_capacity = 0;
      t->has_work = false;
      if (hipMalloc((void**)&t->stack, bytes) != hipSuccess) {
        (void)hipGetLastError();
        t->stack = nullptr;
        return fail(PTK_ERR_NOMEM, "out of device memory (%zu bytes of traversal stacks)", bytes);
      }
      t->stack_capacity = bytes;
    }
    if (t->has_work && t->last_stream != s) PTK_HIP(hipStreamWaitEvent(s, t->done, 0));
    piece = blocks * 64;
    aux = t->stack;
    stack = reinterpret_cast<ptk::Rec64*>(t->stack + aux_bytes);
    armed = true;
    return PTK_OK;
  }
  ~Stack64Lease() {
    if (!armed) return;
    if (t->done == nullptr && hipEventCreateWithFlags(&t->done, hipEventDisableTiming) != hipSuccess) {
      t->done = nullptr;
      (void)hipStreamSynchronize(s);
      t->has_work = false;
      return;
    }
    (void)hipEventRecord(t->done, s);
    t->last_stream = s;
    t->has_work = true;
  }
};

bool want_reorder64(uint64_t nq) { return nq >= 8192 && nq < (1ull << 32); }
size_t permutation64_bytes(uint64_t nq) { return want_reorder64(nq) ? ptkf::permutation_scratch_bytes(nq) + 5 * 256 : 0; }

// Device-side Morton ordering of a batch (make_permutation of the float32 side): *perm lists the
// query rows in launch order; it lives in the lease's aux block.
// long_first (k-NN): the queries in empty cells of the tree's occupancy grid to the front of the launch.
int make_permutation64(const ptk_tree64* t, const double* d_q, uint64_t nq, hipStream_t s, Stack64Lease& lease,
                       const uint32_t** perm, bool long_first = false) {
  *perm = nullptr;
  if (!want_reorder64(nq)) return PTK_OK;
  const int bits = ptkf::morton_bits(nq);
  size_t tmp_bytes = ptkf::sort_tmp_bytes(nq, bits);
  auto align = [](size_t v) { return (v + 255) & ~size_t(255); };
  char* p = lease.aux;
  uint32_t* keys = reinterpret_cast<uint32_t*>(p);
  p += align(nq * 4);
  uint32_t* keys_out = reinterpret_cast<uint32_t*>(p);
  p += align(nq * 4);
  uint32_t* ids = reinterpret_cast<uint32_t*>(p);
  p += align(nq * 4);
  uint32_t* ids_out = reinterpret_cast<uint32_t*>(p);
  p += align(nq * 4);
  void* tmp = p;
  ptk::Morton64Box box{};
  for (uint32_t d = 0; d < t->dim && d < 3; ++d) {
    box.lo[d] = t->flat.root_box.min()[d];
    const double ext = t->flat.root_box.max()[d] - t->flat.root_box.min()[d];
    box.inv[d] = ext > 0 ? 1024.0 / ext : 0.0;
  }
  hipLaunchKernelGGL(ptk::morton64_kernel, dim3((uint32_t)((nq + ptk::kBlock - 1) / ptk::kBlock)), dim3(ptk::kBlock), 0, s,
                     d_q, t->dim, nq, box, (uint32_t)(30 - bits), keys, ids,
                     long_first ? static_cast<const uint8_t*>(t->d_occ) : nullptr);
  const int rc_sort = ptkf::sort_pairs_u32(tmp, tmp_bytes, keys, keys_out, ids, ids_out, nq, bits, s);
  if (rc_sort != PTK_OK) return rc_sort;
  *perm = ids_out;
  return PTK_OK;
}

#define PTK_WITH_METRIC64(CALL)                         \
  do {                                                  \
    const int metric_ = t->metric.load();               \
    if (metric_ == PTK_METRIC_L1) {                     \
      using M = ptk::Metric64L1;                        \
      CALL;                                             \
    } else if (metric_ == PTK_METRIC_LPINF) {           \
      using M = ptk::Metric64LInf;                      \
      CALL;                                             \
    } else if (metric_ == PTK_METRIC_LNINF) {           \
      using M = ptk::Metric64LNInf;                     \
      CALL;                                             \
    } else if (metric_ == PTK_METRIC_SO2) {             \
      using M = ptk::Topo64SO2;                         \
      CALL;                                             \
    } else if (metric_ == PTK_METRIC_SE2_SQUARED) {     \
      using M = ptk::Topo64SE2;                         \
      CALL;                                             \
    } else {                                            \
      using M = ptk::Metric64L2;                        \
      CALL;                                             \
    }                                                   \
  } while (0)

template <class M>
int launch_knn64(const ptk_tree64* t, const double* d_q, const uint32_t* perm, uint64_t nq, uint32_t k, double e,
                 ptk::Neighbor64* d_out, hipStream_t s, Stack64Lease& lease, bool short_tree = false) {
  const bool d3 = t->dim <= 3;  // q / off in registers: the LDS holds the record ring only
  const size_t smem = ptk::lds64_bytes(d3 ? 0 : 2, t->dim);
  if (smem > 160 * 1024) return fail(PTK_ERR_UNSUPPORTED, "dim %u needs %zu bytes of LDS per wavefront (> 160 KiB)", t->dim, smem);
  int rc = PTK_OK;
  // k-list in registers for 1 < k <= 64 (it assumes every slot gets filled: not for k > n_points).
  const int reg = (k > 1 && k <= 64 && !short_tree) ? (k <= 4 ? 4 : (k <= 8 ? 8 : (k <= 16 ? 16 : (k <= 32 ? 32 : 64)))) : 0;
#define PTK_LAUNCH64(KERNEL)                                                                                          \
  do {                                                                                                                \
    rc = allow_lds(KERNEL, smem);                                                                                     \
    if (rc != PTK_OK) return rc;                                                                                      \
    for (uint64_t q0 = 0; q0 < nq; q0 += lease.piece) {                                                               \
      const uint64_t n = std::min(lease.piece, nq - q0);                                                              \
      hipLaunchKernelGGL(KERNEL, dim3((uint32_t)((n + 63) / 64)), dim3(64), smem, s, t->dev, d_q, perm, q0, n, k,   \
                         1.0 / e, d_out, lease.stack, t->slots);                                                     \
    }                                                                                                                 \
  } while (0)
  if (d3) {
    if (reg == 4) PTK_LAUNCH64((ptk::knn64_reg_kernel<M, 4, true>));
    else if (reg == 8) PTK_LAUNCH64((ptk::knn64_reg_kernel<M, 8, true>));
    else if (reg == 16) PTK_LAUNCH64((ptk::knn64_reg_kernel<M, 16, true>));
    else if (reg == 32) PTK_LAUNCH64((ptk::knn64_reg_kernel<M, 32, true>));
    else if (reg == 64) PTK_LAUNCH64((ptk::knn64_reg_kernel<M, 64, true>));
    else PTK_LAUNCH64((ptk::knn64_kernel<M, true>));
  } else if constexpr (!M::kTopo) {  // (the topological metrics: dim 1 or 3)
    if (reg == 4) PTK_LAUNCH64((ptk::knn64_reg_kernel<M, 4, false>));
    else if (reg == 8) PTK_LAUNCH64((ptk::knn64_reg_kernel<M, 8, false>));
    else if (reg == 16) PTK_LAUNCH64((ptk::knn64_reg_kernel<M, 16, false>));
    else if (reg == 32) PTK_LAUNCH64((ptk::knn64_reg_kernel<M, 32, false>));
    else PTK_LAUNCH64((ptk::knn64_kernel<M, false>));  // (64 slots: dim <= 3 only -- 252 VGPRs with q / off in registers)
  }
#undef PTK_LAUNCH64
  PTK_HIP(hipGetLastError());
  return PTK_OK;
}

// ---- the capped k-NN launch + the cooperative search of what it hands over (ptk_kernels_coop64.hpp) ----
// Far children a query may enter before a wavefront takes it over; 0 = every query runs to its end in its lane.  Exact
// searches, dim <= 3, metric_l2_squared / metric_l1 (box distances that are lower bounds: knn_coop_kernel), k <= 32 (the
// cooperative kernel's k-list is three registers per slot and double), trees of k points or more.  As knn_cap of the
// float32 side the cap follows the batch -- a capped launch ends with the lanes that ran to their cap -- and k = 1 is
// capped as well: there is no two-phase search in double.  Test hook knn64_cap: that cap for every batch (0: none).
constexpr int kKnn64CoopPool = 128;
// Tasks a wavefront of the cooperative search can park in HBM (48 bytes each).  k > 16: four times the room and half the
// wavefronts (the K = 32 kernel holds 232 VGPRs: two wavefronts per SIMD) -- with 1 024 a hundredth of the k = 32
// hand-overs of config 2 lost a subtree and were searched again from the root by one lane (24 of 3 684 at 150 k queries).
inline uint32_t knn64_coop_spill(uint32_t k) { return k > 16 ? 4096u : 1024u; }
inline uint32_t knn64_cap(const ptk_tree64* t, double e, uint64_t nq, uint32_t k, bool short_tree) {
  const int m = t->metric.load();
  if (t->dim > 3 || (m != PTK_METRIC_L2_SQUARED && m != PTK_METRIC_L1) || e != 1.0 || k > 32 || short_tree) return 0;
  if (nq < (uint64_t)std::max(1, knob_int("knn_cap_min_nq", 32)) || nq >= (1ull << 32)) return 0;  // (as knn_cap)
  const int forced = knob_int("knn64_cap", -1);
  if (forced >= 0) return (uint32_t)forced;
  // Fitted to tools/sweep_knn64_cap.py on BASELINE config 2's cloud L (profiles/r06_knn64_cap_sweep.jsonl; step ms, best
  // cap against no cap):  k = 1   20 k 0.27 (4) / 1.74, 150 k 0.41 (4) / 2.91, 900 k 0.97 (16) / 3.41, 3.6 M 2.37 (64) / 3.99,
  // 7.2 M 4.21 (128) / 5.03;  k = 4  0.37 (4) / 2.09, 0.59 (8) / 3.48, 1.32 (32) / 3.96, 3.32 (64) / 5.15, 5.99 (128) / 6.78;
  // k = 16  0.69 (8) / 2.50, 1.10 (16) / 4.32, 2.75 (64) / 5.41, 7.14 (128) / 8.06, 11.9 (512) / 10.8;  k = 32 (with the
  // larger spill of knn64_coop_spill)  1.27 (16) / 3.21, 2.10 (32) / 5.28, 5.94 (128) / 6.53, 15.4 (256) / 13.6.  A cap too
  // LOW is a cliff (k = 16, 150 k queries, cap 8: 69 k hand-overs, 3.2 ms), so the floors err upwards; the capped
  // instantiation costs the bulk of a batch 2-7 % (more registers), which the tail it removes no longer pays for at
  // k > 8 on the largest batches and at k > 16 from 1.5 M queries on.
  const double n = (double)nq;
  double cap, lo, hi;
  if (k == 1) {
    cap = n / 56000.0, lo = 4.0, hi = 128.0;
  } else if (k <= 4) {
    cap = n / 56000.0, lo = 8.0, hi = 128.0;
  } else if (k <= 8) {
    cap = n / 28000.0, lo = 12.0, hi = 192.0;
  } else if (k <= 16) {
    if (nq >= 5000000) return 0;
    cap = n / 14000.0, lo = 16.0, hi = 256.0;
  } else {
    if (nq >= 1500000) return 0;
    cap = n / 4700.0, lo = 16.0, hi = 128.0;
  }
  return (uint32_t)std::min(hi, std::max(lo, cap));
}
inline uint64_t knn64_max_handover(uint64_t nq) { return std::max<uint64_t>(nq / 48, std::min<uint64_t>(nq, 24576)); }
inline uint32_t knn64_coop_blocks(const ptk_tree64* t, uint64_t nq, uint32_t k) {
  return (uint32_t)std::min<uint64_t>((uint64_t)t->cus * (k > 16 ? 8u : 16u), std::max<uint64_t>(64, knn64_max_handover(nq)));
}
// Transient arrays of a capped call (behind the permutation in the lease's aux block): counters, the hand-over list
// with its tasks, the redo list, the spill runs.
inline size_t knn64_coop_scratch_bytes(const ptk_tree64* t, uint64_t nq, uint32_t k) {
  const uint64_t mh = knn64_max_handover(nq);
  return ptk::kMetaWords * 4 + 256 + 3 * (mh * 4) + mh * ptk::kMaxTasks * sizeof(ptk::Task64) +
         (size_t)knn64_coop_blocks(t, nq, k) * knn64_coop_spill(k) * sizeof(ptk::Task64) + 2048;
}

template <class M>
int launch_knn64_capped(const ptk_tree64* t, const double* d_q, const uint32_t* perm, uint64_t nq, uint32_t k,
                        uint32_t cap, ptk::Neighbor64* d_out, hipStream_t s, Stack64Lease& lease, char* scratch) {
  auto align = [](size_t v) { return (v + 255) & ~size_t(255); };
  const uint64_t mh = knn64_max_handover(nq);
  const uint32_t coop_blocks = knn64_coop_blocks(t, nq, k);
  const uint32_t spill_cap = knn64_coop_spill(k);
  char* p = scratch;
  uint32_t* meta = reinterpret_cast<uint32_t*>(p);
  p += align(ptk::kMetaWords * 4);
  ptk::Handover64* d_ho = reinterpret_cast<ptk::Handover64*>(p);
  p += align(sizeof(ptk::Handover64));
  uint32_t* heavy_list = reinterpret_cast<uint32_t*>(p);
  p += align(mh * 4);
  uint32_t* ntasks = reinterpret_cast<uint32_t*>(p);
  p += align(mh * 4);
  uint32_t* redo_list = reinterpret_cast<uint32_t*>(p);
  p += align(mh * 4);
  ptk::Task64* tasks = reinterpret_cast<ptk::Task64*>(p);
  p += align(mh * ptk::kMaxTasks * sizeof(ptk::Task64));
  ptk::Task64* spill = reinterpret_cast<ptk::Task64*>(p);
  t->last_meta = meta;
  ptk::Handover64 ho{};
  ho.counter = ptk::kMetaHeavy;
  ho.meta = meta;
  ho.heavy_list = heavy_list;
  ho.ntasks = ntasks;
  ho.tasks = tasks;
  ho.max_heavy = (uint32_t)mh;
  static_assert(ptk::kMetaWords <= 64, "knn64_handover_init_kernel: one wavefront zeroes the counters");
  hipLaunchKernelGGL(ptk::knn64_handover_init_kernel, dim3(1), dim3(64), 0, s, ho, d_ho);
  const size_t smem = ptk::lds64_bytes(0, t->dim);
  const size_t coop_smem = ptk::knn64_coop_lds_bytes(kKnn64CoopPool);
  const uint32_t redo_blocks = (uint32_t)std::min<uint64_t>((uint64_t)t->cus, std::max<uint64_t>(1, lease.piece / 64));
#define PTK_LAUNCH64C(KK)                                                                                             \
  do {                                                                                                                \
    for (uint64_t q0 = 0; q0 < nq; q0 += lease.piece) {                                                               \
      const uint64_t n = std::min(lease.piece, nq - q0);                                                              \
      hipLaunchKernelGGL((ptk::knn64_capped_kernel<M, KK>), dim3((uint32_t)((n + 63) / 64)), dim3(64), smem, s, t->dev, \
                         d_q, perm, q0, n, k, d_out, lease.stack, t->slots, cap, d_ho);                               \
    }                                                                                                                 \
    hipLaunchKernelGGL((ptk::knn64_coop_kernel<KK, kKnn64CoopPool, M>), dim3(coop_blocks), dim3(64), coop_smem, s,     \
                       t->dev, d_q, k, d_out, ho, redo_list, ptk::kMetaRedo, spill, spill_cap);                       \
    hipLaunchKernelGGL((ptk::knn64_redo_kernel<M, KK>), dim3(redo_blocks), dim3(64), smem, s, t->dev, d_q, k, d_out,   \
                       meta, ptk::kMetaRedo, redo_list, lease.stack, t->slots);                                       \
  } while (0)
  if (k == 1) PTK_LAUNCH64C(1);
  else if (k <= 4) PTK_LAUNCH64C(4);
  else if (k <= 8) PTK_LAUNCH64C(8);
  else if (k <= 16) PTK_LAUNCH64C(16);
  else PTK_LAUNCH64C(32);
#undef PTK_LAUNCH64C
  PTK_HIP(hipGetLastError());
  return PTK_OK;
}

// ---- the capped radius search (ptk_kernels_coop64.hpp) ----
// Far children a query of the double radius search may enter before a wavefront takes it over; 0 = every query runs to
// its end in its lane.  dim <= 3, the four non-topological metrics (no certificate is involved: the radius visitor's
// bound never changes), trees no deeper than a key has bits for, leaves of at most 64 pieces of 32 points.  The rule is
// radius_cap's of the float32 side (what the best caps have in common is 8-17 thousand hand-overs).  Test hook
// radius64_cap: that cap for every batch (0: none).
constexpr int kRadius64CoopPool = 64;
constexpr uint32_t kRadius64CoopSpill = 1024;  // tasks a wavefront of the cooperative count can park in HBM (48 KB)
inline uint32_t radius64_cap(const ptk_tree64* t, uint64_t nq) {
  const int m = t->metric.load();
  if (t->dim > 3 || m == PTK_METRIC_SO2 || m == PTK_METRIC_SE2_SQUARED) return 0;
  if (t->max_depth > ptk::kRc64MaxDepth || t->max_leaf_count > 2048u) return 0;
  // (no lower limit, as radius_cap: 8 double queries with a long one among them 0.94 ms uncapped, 0.24 capped)
  if (nq < (uint64_t)std::max(1, knob_int("radius_cap_min_nq", 1)) || nq >= (1ull << 32)) return 0;
  const int forced = knob_int("radius64_cap", -1);
  if (forced >= 0) return (uint32_t)forced;
  if (nq >= 1500000) return 0;
  return (uint32_t)std::min(256.0, std::max(8.0, (double)nq / 2400.0));
}
inline uint64_t radius64_max_handover(uint64_t nq) { return std::max<uint64_t>(nq / 48, std::min<uint64_t>(nq, 24576)); }
inline uint64_t radius64_entry_cap(uint64_t nq) { return std::max<uint64_t>(radius64_max_handover(nq), 64) * 192; }  // entries of all hand-overs together
inline uint32_t radius64_coop_blocks(const ptk_tree64* t, uint64_t nq) {
  return (uint32_t)std::min<uint64_t>((uint64_t)t->cus * 12u, std::max<uint64_t>(64, radius64_max_handover(nq)));
}
// What a capped call carves out of the lease's aux block (behind the permutation).
struct Radius64Scratch {
  uint32_t* meta = nullptr;
  ptk::Handover64* d_ho = nullptr;
  ptk::Handover64 ho{};
  ptk::RadiusHeavy64 hv{};
  uint32_t* redo_list = nullptr;
  uint32_t* over_list = nullptr;
  ptk::Task64* spill = nullptr;
  uint8_t* flag = nullptr;
  uint32_t coop_blocks = 0;
};
inline size_t radius64_coop_scratch_bytes(const ptk_tree64* t, uint64_t nq) {
  const uint64_t mh = radius64_max_handover(nq);
  return ptk::kMetaWords * 4 + 256 + 8 * (mh * 4 + 256) + mh * ptk::kMaxTasks * sizeof(ptk::Task64) + radius64_entry_cap(nq) * 8 +
         (size_t)radius64_coop_blocks(t, nq) * kRadius64CoopSpill * sizeof(ptk::Task64) + nq + 4096;
}
inline Radius64Scratch radius64_carve(const ptk_tree64* t, uint64_t nq, char* p) {
  auto align = [](size_t v) { return (v + 255) & ~size_t(255); };
  const uint64_t mh = radius64_max_handover(nq);
  Radius64Scratch r;
  auto take = [&](size_t bytes) {
    char* at = p;
    p += align(bytes);
    return at;
  };
  r.meta = reinterpret_cast<uint32_t*>(take(ptk::kMetaWords * 4));
  r.d_ho = reinterpret_cast<ptk::Handover64*>(take(sizeof(ptk::Handover64)));
  r.ho.counter = ptk::kMetaHeavy;
  r.ho.meta = r.meta;
  r.ho.heavy_list = reinterpret_cast<uint32_t*>(take(mh * 4));
  r.ho.ntasks = reinterpret_cast<uint32_t*>(take(mh * 4));
  r.ho.tasks = reinterpret_cast<ptk::Task64*>(take(mh * ptk::kMaxTasks * sizeof(ptk::Task64)));
  r.ho.max_heavy = (uint32_t)mh;
  r.hv.meta = r.meta;
  r.hv.rows = reinterpret_cast<uint32_t*>(take(mh * 4));
  r.hv.own = reinterpret_cast<uint32_t*>(take(mh * 4));
  r.hv.run_at = reinterpret_cast<uint32_t*>(take(mh * 4));
  r.hv.run_n = reinterpret_cast<uint32_t*>(take(mh * 4));
  r.hv.entries = reinterpret_cast<unsigned long long*>(take(radius64_entry_cap(nq) * 8));
  r.hv.max_heavy = (uint32_t)mh;
  r.hv.entry_cap = (uint32_t)std::min<uint64_t>(radius64_entry_cap(nq), 0xFFFFFFFFull);
  r.redo_list = reinterpret_cast<uint32_t*>(take(mh * 4));
  r.over_list = reinterpret_cast<uint32_t*>(take(mh * 4));
  r.coop_blocks = radius64_coop_blocks(t, nq);
  r.spill = reinterpret_cast<ptk::Task64*>(take((size_t)r.coop_blocks * kRadius64CoopSpill * sizeof(ptk::Task64)));
  r.flag = reinterpret_cast<uint8_t*>(take(nq));
  return r;
}

// One pass of a capped call: FILL = false the count pass (capped launch, cooperative count, recount of what that lost),
// FILL = true the fill pass (capped launch, cooperative replay, refill of the lost rows).
template <class M, bool FILL>
int launch_radius64_capped(const ptk_tree64* t, const double* d_q, const uint32_t* perm, uint64_t nq, double radius, double e,
                           uint32_t cap, uint64_t* d_counts, const uint64_t* d_offsets, ptk::Neighbor64* d_out, hipStream_t s,
                           Stack64Lease& lease, const Radius64Scratch& r) {
  const size_t smem = ptk::lds64_bytes(0, t->dim);
  const size_t coop_smem = ptk::radius64_coop_lds_bytes(kRadius64CoopPool);
  const uint32_t redo_blocks = (uint32_t)std::min<uint64_t>((uint64_t)t->cus, std::max<uint64_t>(1, lease.piece / 64));
  if (!FILL) {
    hipLaunchKernelGGL(ptk::knn64_handover_init_kernel, dim3(1), dim3(64), 0, s, r.ho, r.d_ho);
    PTK_HIP(hipMemsetAsync(r.flag, 0, nq, s));
  }
  for (uint64_t q0 = 0; q0 < nq; q0 += lease.piece) {
    const uint64_t n = std::min(lease.piece, nq - q0);
    hipLaunchKernelGGL((ptk::radius64_capped_kernel<M, FILL>), dim3((uint32_t)((n + 63) / 64)), dim3(64), smem, s, t->dev, d_q,
                       perm, q0, n, radius, 1.0 / e, d_counts, d_offsets, d_out, lease.stack, t->slots, cap, r.d_ho, r.flag);
  }
  if (!FILL) {
    hipLaunchKernelGGL((ptk::radius64_coop_count_kernel<kRadius64CoopPool, M>), dim3(r.coop_blocks), dim3(64), coop_smem, s,
                       t->dev, d_q, radius, 1.0 / e, d_counts, r.ho, r.hv, r.redo_list, r.spill, kRadius64CoopSpill);
    hipLaunchKernelGGL((ptk::radius64_redo_kernel<M, false>), dim3(redo_blocks), dim3(64), smem, s, t->dev, d_q, radius, 1.0 / e,
                       d_counts, d_offsets, d_out, r.meta, ptk::kMetaRedo, r.redo_list, lease.stack, t->slots);
  } else {
    hipLaunchKernelGGL((ptk::radius64_coop_replay_kernel<M>), dim3(r.coop_blocks), dim3(64), 0, s, t->dev, d_q, 1.0 / e, r.hv,
                       d_offsets, d_out, r.over_list);
    hipLaunchKernelGGL((ptk::radius64_redo_kernel<M, true>), dim3(redo_blocks), dim3(64), smem, s, t->dev, d_q, radius, 1.0 / e,
                       d_counts, d_offsets, d_out, r.meta, ptk::kMetaRc64Over, r.over_list, lease.stack, t->slots);
  }
  PTK_HIP(hipGetLastError());
  return PTK_OK;
}
#define PTK_WITH_EUCLID64(CALL)                      \
  do {                                               \
    const int metric_ = t->metric.load();            \
    if (metric_ == PTK_METRIC_L1) {                  \
      using M = ptk::Metric64L1;                     \
      CALL;                                          \
    } else if (metric_ == PTK_METRIC_LPINF) {        \
      using M = ptk::Metric64LInf;                   \
      CALL;                                          \
    } else if (metric_ == PTK_METRIC_LNINF) {        \
      using M = ptk::Metric64LNInf;                  \
      CALL;                                          \
    } else {                                         \
      using M = ptk::Metric64L2;                     \
      CALL;                                          \
    }                                                \
  } while (0)

template <class M, bool FILL>
int launch_radius64(const ptk_tree64* t, const double* d_q, const uint32_t* perm, uint64_t nq, double radius, double e,
                    uint64_t* d_counts, const uint64_t* d_offsets, ptk::Neighbor64* d_out, hipStream_t s,
                    Stack64Lease& lease) {
  const bool d3 = t->dim <= 3;
  const size_t smem = ptk::lds64_bytes(d3 ? 0 : 2, t->dim);
  if (smem > 160 * 1024) return fail(PTK_ERR_UNSUPPORTED, "dim %u needs %zu bytes of LDS per wavefront (> 160 KiB)", t->dim, smem);
  int rc = allow_lds(ptk::radius64_kernel<M, FILL, M::kTopo>, smem);
  if (rc != PTK_OK) return rc;
  for (uint64_t q0 = 0; q0 < nq; q0 += lease.piece) {
    const uint64_t n = std::min(lease.piece, nq - q0);
    if (d3) {
      hipLaunchKernelGGL((ptk::radius64_kernel<M, FILL, true>), dim3((uint32_t)((n + 63) / 64)), dim3(64), smem, s, t->dev,
                         d_q, perm, q0, n, radius, 1.0 / e, d_counts, d_offsets, d_out, lease.stack, t->slots);
    } else if constexpr (!M::kTopo) {
      hipLaunchKernelGGL((ptk::radius64_kernel<M, FILL, false>), dim3((uint32_t)((n + 63) / 64)), dim3(64), smem, s, t->dev,
                         d_q, perm, q0, n, radius, 1.0 / e, d_counts, d_offsets, d_out, lease.stack, t->slots);
    }
  }
  PTK_HIP(hipGetLastError());
  return PTK_OK;
}

template <bool FILL>
int launch_box64(const ptk_tree64* t, const double* d_mins, const double* d_maxs, uint64_t nb, uint64_t* d_counts,
                 const uint64_t* d_offsets, int32_t* d_out, hipStream_t s, Stack64Lease& lease) {
  const size_t smem = ptk::lds64_bytes(4, t->dim);
  if (smem > 160 * 1024) return fail(PTK_ERR_UNSUPPORTED, "dim %u needs %zu bytes of LDS per wavefront (> 160 KiB)", t->dim, smem);
  // A topological tree: circle axes (metric_so2: axis 0; metric_se2_squared: axis 2), the four-bound tests.
  const int metric = t->metric.load();
  const bool topo = metric == PTK_METRIC_SO2 || metric == PTK_METRIC_SE2_SQUARED;
  const uint32_t s1_mask = !topo ? 0u : (metric == PTK_METRIC_SO2 ? 1u : 4u);
  if (topo && t->dev.outer == nullptr) return fail(PTK_ERR_INVALID, "this topological tree has no outer bounds on the device");
  int rc = topo ? allow_lds(ptk::box64_kernel<FILL, true>, smem) : allow_lds(ptk::box64_kernel<FILL, false>, smem);
  if (rc != PTK_OK) return rc;
  for (uint64_t b0 = 0; b0 < nb; b0 += lease.piece) {
    const uint64_t n = std::min(lease.piece, nb - b0);
    if (topo) {
      hipLaunchKernelGGL((ptk::box64_kernel<FILL, true>), dim3((uint32_t)((n + 63) / 64)), dim3(64), smem, s, t->dev,
                         static_cast<const double*>(t->d_root), d_mins, d_maxs, b0, n, d_counts, d_offsets, d_out,
                         lease.stack, t->slots, s1_mask);
    } else {
      hipLaunchKernelGGL((ptk::box64_kernel<FILL, false>), dim3((uint32_t)((n + 63) / 64)), dim3(64), smem, s, t->dev,
                         static_cast<const double*>(t->d_root), d_mins, d_maxs, b0, n, d_counts, d_offsets, d_out,
                         lease.stack, t->slots, 0u);
    }
  }
  PTK_HIP(hipGetLastError());
  return PTK_OK;
}

// counts (nq + 1, last = 0) -> offsets (nq + 1) on the device, offsets copied to the host.
int scan_counts64(uint64_t* d_c, uint64_t* d_o, uint64_t nq, uint64_t* offsets) {
  size_t tmp_bytes = 0;
  void* tmp = nullptr;
  hipError_t he = rocprim::exclusive_scan(nullptr, tmp_bytes, d_c, d_o, (uint64_t)0, nq + 1, rocprim::plus<uint64_t>(),
                                          (hipStream_t) nullptr);
  if (he == hipSuccess) he = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16);
  if (he == hipSuccess)
    he = rocprim::exclusive_scan(tmp, tmp_bytes, d_c, d_o, (uint64_t)0, nq + 1, rocprim::plus<uint64_t>(),
                                 (hipStream_t) nullptr);
  if (he == hipSuccess) he = hipMemcpy(offsets, d_o, (nq + 1) * 8, hipMemcpyDeviceToHost);
  if (tmp) (void)hipFree(tmp);
  if (he != hipSuccess) return fail(PTK_ERR_DEVICE, "HIP error in the offsets scan: %s", hipGetErrorString(he));
  return PTK_OK;
}

template <class Flat>
void adopt_flat64(ptk_tree64* t, Flat&& flat, uint32_t dim, uint64_t n_points) {
  t->dim = dim;
  t->n_points = n_points;
  t->flat = std::forward<Flat>(flat);
  // The four-bound form lives beside the tree, so that ptk_tree64_serialize keeps writing the euclidean stream.
  t->outer = std::move(t->flat.outer_bounds);
  t->flat.outer_bounds.clear();
  t->flat.keep_outer_bounds = false;
}

// What write_flat_tree reads of a tree, with the outer bounds of the handle (or none).
struct FlatView64 {
  using index_type = int;
  using scalar_type = double;
  const decltype(ptk_tree64::flat_t::root_box)& root_box;
  const decltype(ptk_tree64::flat_t::indices)& indices;
  const decltype(ptk_tree64::flat_t::nodes)& nodes;
  bool keep_outer_bounds;
  const std::vector<std::array<double, 2>>& outer_bounds;
};

int serialize64(const ptk_tree64* t, bool topological, void* buf, uint64_t cap, uint64_t* size) {
  if (t == nullptr || size == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  if (topological && t->outer.size() != t->flat.nodes.size())
    return fail(PTK_ERR_INVALID, "this tree has no outer bounds: it cannot be written as a topological tree");
  try {
    std::ostringstream os(std::ios::out | std::ios::binary);
    const FlatView64 view{t->flat.root_box, t->flat.indices, t->flat.nodes, topological, t->outer};
    pico_tree::internal::write_flat_tree(view, os);
    const std::string bytes = os.str();
    *size = bytes.size();
    if (buf == nullptr) return PTK_OK;
    if (cap < bytes.size()) return fail(PTK_ERR_INVALID, "buffer of %llu bytes, stream needs %llu",
                                        (unsigned long long)cap, (unsigned long long)bytes.size());
    std::memcpy(buf, bytes.data(), bytes.size());
    return PTK_OK;
  } catch (const std::bad_alloc&) {
    return fail(PTK_ERR_NOMEM, "out of memory");
  }
}

int create_from_stream64(const double* points, uint64_t n_points, uint32_t dim, const void* stream, uint64_t stream_bytes,
                         bool topological, int32_t device, ptk_tree64** out) {
  if (out == nullptr) return fail(PTK_ERR_INVALID, "null out pointer");
  *out = nullptr;
  if (points == nullptr || stream == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  if (dim == 0 || n_points == 0) return fail(PTK_ERR_INVALID, "dim and n_points must be positive");
  ptk_tree64* t = nullptr;
  try {
    std::istringstream is(std::string(static_cast<const char*>(stream), stream_bytes), std::ios::in | std::ios::binary);
    ptk_tree64::flat_t flat = pico_tree::internal::read_flat_tree<ptk_tree64::flat_t>(is, topological, dim, n_points);
    if (flat.root_box.size() != dim) return fail(PTK_ERR_INVALID, "stream is %zu-dimensional, points are %u-dimensional",
                                                 (size_t)flat.root_box.size(), dim);
    if (flat.indices.size() != n_points)
      return fail(PTK_ERR_INVALID, "stream indexes %zu points, %llu were given", flat.indices.size(),
                  (unsigned long long)n_points);
    t = new ptk_tree64;
    adopt_flat64(t, std::move(flat), dim, n_points);
  } catch (const std::bad_alloc&) {
    delete t;
    return fail(PTK_ERR_NOMEM, "out of memory");
  } catch (const std::exception& e) {
    delete t;
    return fail(PTK_ERR_INVALID, "bad kd_tree stream: %s", e.what());
  }
  return finish_create64(t, points, device, out);
}

}  // namespace

extern "C" {

int ptk_tree64_create_from_points(const double* points, uint64_t n_points, uint32_t dim, uint64_t max_leaf_size,
                                  int32_t device, ptk_tree64** out) {
  if (out == nullptr) return fail(PTK_ERR_INVALID, "null out pointer");
  *out = nullptr;
  if (points == nullptr) return fail(PTK_ERR_INVALID, "null points");
  if (dim == 0 || n_points == 0 || max_leaf_size == 0)
    return fail(PTK_ERR_INVALID, "dim, n_points and max_leaf_size must be positive");
  if (n_points >= (1ull << 31)) return fail(PTK_ERR_INVALID, "n_points must be < 2^31");
  ptk_tree64* t = new (std::nothrow) ptk_tree64;
  if (t == nullptr) return fail(PTK_ERR_NOMEM, "out of memory");
  try {
    using namespace pico_tree;
    using space_t = space_map<point_map<double const, dynamic_extent>>;
    space_t space(points, n_points, dim);
    internal::space_view<space_t> view(space);
    adopt_flat64(t, internal::build_flat_tree<int>(view, max_leaf_size_t(max_leaf_size), bounds_from_space,
                                                   sliding_midpoint_max_side, true, build_threads()),
                 dim, n_points);
  } catch (const std::bad_alloc&) {
    delete t;
    return fail(PTK_ERR_NOMEM, "out of memory");
  } catch (const std::length_error& err) {  // degenerate point set: see flat_builder::grow
    delete t;
    return fail(PTK_ERR_UNSUPPORTED, "%s", err.what());
  }
  return finish_create64(t, points, device, out);
}

int ptk_tree64_create_from_stream(const double* points, uint64_t n_points, uint32_t dim, const void* stream,
                                  uint64_t stream_bytes, int32_t device, ptk_tree64** out) {
  return create_from_stream64(points, n_points, dim, stream, stream_bytes, false, device, out);
}

int ptk_tree64_create_from_topological_stream(const double* points, uint64_t n_points, uint32_t dim, const void* stream,
                                              uint64_t stream_bytes, int32_t device, ptk_tree64** out) {
  return create_from_stream64(points, n_points, dim, stream, stream_bytes, true, device, out);
}

void ptk_tree64_destroy(ptk_tree64* t) {
  if (t == nullptr) return;
  if (t->device >= 0) {
    DeviceGuard guard(t->device);
    if (t->has_work) (void)hipEventSynchronize(t->done);
    if (t->done) (void)hipEventDestroy(t->done);
    if (t->stack) (void)hipFree(t->stack);
    if (t->d_nodes) (void)hipFree(t->d_nodes);
    if (t->d_pts) (void)hipFree(t->d_pts);
    if (t->d_index) (void)hipFree(t->d_index);
    if (t->d_ranges) (void)hipFree(t->d_ranges);
    if (t->d_root) (void)hipFree(t->d_root);
    if (t->d_outer) (void)hipFree(t->d_outer);
    if (t->d_occ) (void)hipFree(t->d_occ);
    for (char* b : t->io)
      if (b) (void)hipFree(b);
  }
  delete t;
}

int ptk_tree64_get_info(const ptk_tree64* t, ptk_tree_info* info) {
  if (t == nullptr || info == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  info->dim = t->dim;
  info->n_points = t->n_points;
  info->n_nodes = t->flat.nodes.size();
  info->n_leaves = t->n_leaves;
  info->max_depth = t->max_depth;
  info->max_leaf_count = t->max_leaf_count;
  info->device_bytes = t->device_bytes;
  info->device = t->device;
  return PTK_OK;
}

int ptk_tree64_set_metric(ptk_tree64* t, int metric) {
  if (t == nullptr || metric < PTK_METRIC_L2_SQUARED || metric > PTK_METRIC_SE2_SQUARED)
    return fail(PTK_ERR_INVALID, "bad metric");
  if (metric == PTK_METRIC_SO2 || metric == PTK_METRIC_SE2_SQUARED) {
    if (metric == PTK_METRIC_SO2 && t->dim != 1) return fail(PTK_ERR_INVALID, "metric_so2 is a metric of 1-dimensional points");
    if (metric == PTK_METRIC_SE2_SQUARED && t->dim != 3)
      return fail(PTK_ERR_INVALID, "metric_se2_squared is a metric of 3-dimensional points (x, y, angle)");
    if (t->outer.size() != t->flat.nodes.size())
      return fail(PTK_ERR_INVALID, "the topological metrics need the outer bounds of every branch "
                                   "(ptk_tree64_create_from_points / ptk_tree64_create_from_topological_stream)");
    std::lock_guard<std::mutex> lock(t->mutex);
    if (t->device >= 0 && t->d_outer == nullptr) {  // branch order of the device records
      std::vector<double> dev_outer;
      std::string err = ptk::encode_outer64(t->dim, t->n_points, t->flat.nodes.data(), t->flat.nodes.size(),
                                            t->outer.data(), dev_outer);
      if (!err.empty()) return fail(PTK_ERR_INVALID, "%s", err.c_str());
      DeviceGuard guard(t->device);
      if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", t->device);
      PTK_HIP(hipMalloc(&t->d_outer, dev_outer.size() * sizeof(double)));
      PTK_HIP(hipMemcpy(t->d_outer, dev_outer.data(), dev_outer.size() * sizeof(double), hipMemcpyHostToDevice));
      t->dev.outer = static_cast<const double2*>(t->d_outer);
      t->device_bytes += dev_outer.size() * sizeof(double);
    }
  }
  t->metric.store(metric);
  return PTK_OK;
}

int ptk_tree64_serialize(const ptk_tree64* t, void* buf, uint64_t cap, uint64_t* size) {
  return serialize64(t, false, buf, cap, size);
}

int ptk_tree64_serialize_topological(const ptk_tree64* t, void* buf, uint64_t cap, uint64_t* size) {
  return serialize64(t, true, buf, cap, size);
}

int ptk_search64_knn_device(const ptk_tree64* t, const double* d_q, uint64_t nq, uint32_t k, double e,
                            ptk_neighbor64* d_out, void* stream) {
  int rc = check_search64(t, d_q, nq);
  if (rc != PTK_OK) return rc;
  if (k == 0) return fail(PTK_ERR_INVALID, "k must be >= 1");
  if (!(e > 0.0)) return fail(PTK_ERR_INVALID, "approximation ratio e must be > 0");
  if (nq == 0) return PTK_OK;
  if (d_out == nullptr) return fail(PTK_ERR_INVALID, "null output buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  DeviceGuard guard(t->device);
  if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", t->device);
  // k > n_points, as the float32 entry (ptk_search_knn_device): the n_points neighbours in order, the last
  // slot's distance at the DBL_MAX sentinel (search_visitor.hpp:95-110), the slots in between zeroed.
  const bool short_tree = k > t->n_points;
  if (short_tree) PTK_HIP(hipMemsetAsync(d_out, 0, (size_t)nq * k * sizeof(ptk_neighbor64), s));
  Stack64Lease lease(t, s);
  uint32_t cap = knn64_cap(t, e, nq, k, short_tree);
  const size_t perm_bytes = (permutation64_bytes(nq) + 255) & ~size_t(255);
  rc = lease.acquire(nq, perm_bytes + (cap != 0u ? knn64_coop_scratch_bytes(t, nq, k) : 0));
  if (rc == PTK_ERR_NOMEM && cap != 0u) {  // (no room for the hand-over list: the uncapped search needs none)
    cap = 0u;
    rc = lease.acquire(nq, perm_bytes);
  }
  if (rc != PTK_OK) return rc;
  const uint32_t* perm = nullptr;
  rc = make_permutation64(t, d_q, nq, s, lease, &perm, /*long_first=*/true);
  if (rc != PTK_OK) return rc;
  if (cap != 0u) {
    if (t->metric.load() == PTK_METRIC_L1)
      return launch_knn64_capped<ptk::Metric64L1>(t, d_q, perm, nq, k, cap, reinterpret_cast<ptk::Neighbor64*>(d_out), s, lease,
                                                 lease.aux + perm_bytes);
    return launch_knn64_capped<ptk::Metric64L2>(t, d_q, perm, nq, k, cap, reinterpret_cast<ptk::Neighbor64*>(d_out), s, lease,
                                               lease.aux + perm_bytes);
  }
  PTK_WITH_METRIC64(rc = launch_knn64<M>(t, d_q, perm, nq, k, e, reinterpret_cast<ptk::Neighbor64*>(d_out), s, lease,
                                         short_tree));
  return rc;
}

int ptk_tree64_debug_knn_coop_counts(const ptk_tree64* t, uint32_t counts[7]) {
  if (t == nullptr || counts == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  if (t->device == kDeviceNone) return fail(PTK_ERR_DEVICE, "this handle has no device replica");
  DeviceGuard guard(t->device);
  std::lock_guard<std::mutex> lock(t->mutex);
  for (int i = 0; i < 7; ++i) counts[i] = 0;
  if (t->last_meta == nullptr) return PTK_OK;
  uint32_t meta[ptk::kMetaWords];
  PTK_HIP(hipDeviceSynchronize());
  PTK_HIP(hipMemcpy(meta, t->last_meta, sizeof(meta), hipMemcpyDeviceToHost));
  counts[0] = meta[ptk::kMetaHeavy];
  counts[1] = meta[ptk::kMetaRedo];
  counts[2] = meta[ptk::kKnnWhyPool];
  counts[3] = meta[ptk::kKnnWhyTie];
  counts[4] = meta[ptk::kKnnWhyBox];
  counts[5] = meta[ptk::kKnnWhyRange];
  counts[6] = meta[ptk::kKnnTieSweeps];
  return PTK_OK;
}

int ptk_search64_knn(const ptk_tree64* t, const double* q, uint64_t nq, uint32_t k, double e, ptk_neighbor64* out) {
  int rc = check_search64(t, q, nq);
  if (rc != PTK_OK) return rc;
  if (nq == 0) return PTK_OK;
  if (out == nullptr) return fail(PTK_ERR_INVALID, "null output buffer");
  DeviceGuard guard(t->device);
  if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", t->device);
  std::lock_guard<std::mutex> io_lock(t->io_mutex);
  IoBuffer bq, bo;
  const size_t qbytes = (size_t)nq * t->dim * sizeof(double), obytes = (size_t)nq * k * sizeof(ptk_neighbor64);
  hipError_t he = bq.get(t, 0, qbytes);
  if (he == hipSuccess) he = bo.get(t, 1, obytes);
  double* d_q = reinterpret_cast<double*>(bq.p);
  ptk_neighbor64* d_out = reinterpret_cast<ptk_neighbor64*>(bo.p);
  if (he == hipSuccess) he = hipMemset(d_out, 0, obytes ? obytes : 16);  // padding bytes of the records
  if (he == hipSuccess) he = hipMemcpy(d_q, q, qbytes, hipMemcpyHostToDevice);
  if (he == hipSuccess) {
    rc = ptk_search64_knn_device(t, d_q, nq, k, e, d_out, nullptr);
    if (rc == PTK_OK) he = hipMemcpy(out, d_out, obytes, hipMemcpyDeviceToHost);
  }
  if (rc == PTK_OK && he != hipSuccess) rc = fail(PTK_ERR_DEVICE, "HIP error: %s", hipGetErrorString(he));
  return rc;
}

int ptk_search64_radius(const ptk_tree64* t, const double* q, uint64_t nq, double radius, double e, int sort,
                        uint64_t* offsets, ptk_neighbor64** out) {
  if (out == nullptr || offsets == nullptr) return fail(PTK_ERR_INVALID, "null output pointer");
  *out = nullptr;
  int rc = check_search64(t, q, nq);
  if (rc != PTK_OK) return rc;
  if (!(e > 0.0)) return fail(PTK_ERR_INVALID, "approximation ratio e must be > 0");
  offsets[0] = 0;
  if (nq == 0) return PTK_OK;
  DeviceGuard guard(t->device);
  if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", t->device);
  std::lock_guard<std::mutex> io_lock(t->io_mutex);
  IoBuffer bin, bout;
  const size_t qbytes = (size_t)nq * t->dim * sizeof(double);
  const size_t q_room = (qbytes + 255) & ~size_t(255), c_room = ((nq + 1) * 8 + 255) & ~size_t(255);
  hipError_t he = bin.get(t, 0, q_room + 2 * c_room);
  double* d_q = reinterpret_cast<double*>(bin.p);
  uint64_t* d_c = reinterpret_cast<uint64_t*>(bin.p + q_room);
  uint64_t* d_o = reinterpret_cast<uint64_t*>(bin.p + q_room + c_room);
  ptk_neighbor64* d_out = nullptr;
  if (he == hipSuccess) he = hipMemset(d_c, 0, (nq + 1) * 8);
  if (he == hipSuccess) he = hipMemcpy(d_q, q, qbytes, hipMemcpyHostToDevice);
  uint64_t total = 0;
  if (he == hipSuccess) {
    Stack64Lease lease(t, nullptr);
    uint32_t cap = radius64_cap(t, nq);
    const size_t perm_bytes = (permutation64_bytes(nq) + 255) & ~size_t(255);
    rc = lease.acquire(nq, perm_bytes + (cap != 0u ? radius64_coop_scratch_bytes(t, nq) : 0));
    if (rc == PTK_ERR_NOMEM && cap != 0u) {  // (no room for the hand-over list: the uncapped search needs none)
      cap = 0u;
      rc = lease.acquire(nq, perm_bytes);
    }
    const uint32_t* perm = nullptr;
    if (rc == PTK_OK) rc = make_permutation64(t, d_q, nq, nullptr, lease, &perm);
    Radius64Scratch rs;
    if (rc == PTK_OK && cap != 0u) {
      rs = radius64_carve(t, nq, lease.aux + perm_bytes);
      t->last_meta = rs.meta;
      PTK_WITH_EUCLID64(rc = (launch_radius64_capped<M, false>(t, d_q, perm, nq, radius, e, cap, d_c, nullptr, nullptr, nullptr,
                                                              lease, rs)));
    } else if (rc == PTK_OK) {
      PTK_WITH_METRIC64(rc = (launch_radius64<M, false>(t, d_q, perm, nq, radius, e, d_c, nullptr, nullptr, nullptr, lease)));
    }
    if (rc == PTK_OK) rc = scan_counts64(d_c, d_o, nq, offsets);
    if (rc == PTK_OK) {
      total = offsets[nq];
      const size_t obytes = std::max<uint64_t>(total, 1) * sizeof(ptk_neighbor64);
      he = bout.get(t, 1, obytes);
      d_out = reinterpret_cast<ptk_neighbor64*>(bout.p);
      if (he == hipSuccess) he = hipMemset(d_out, 0, obytes);
      if (he == hipSuccess && cap != 0u) {
        PTK_WITH_EUCLID64(rc = (launch_radius64_capped<M, true>(t, d_q, perm, nq, radius, e, cap, nullptr, d_o,
                                                               reinterpret_cast<ptk::Neighbor64*>(d_out), nullptr, lease, rs)));
      } else if (he == hipSuccess) {
        PTK_WITH_METRIC64(rc = (launch_radius64<M, true>(t, d_q, perm, nq, radius, e, nullptr, d_o,
                                                         reinterpret_cast<ptk::Neighbor64*>(d_out), nullptr, lease)));
      }
      if (he == hipSuccess && rc == PTK_OK && sort) {
        hipLaunchKernelGGL(ptk::sort_rows64_kernel, dim3((uint32_t)((nq + ptk::kBlock - 1) / ptk::kBlock)), dim3(ptk::kBlock),
                           0, nullptr, d_o, nq, reinterpret_cast<ptk::Neighbor64*>(d_out));
        he = hipGetLastError();
      }
      if (he == hipSuccess && rc == PTK_OK) {
        *out = static_cast<ptk_neighbor64*>(std::malloc(obytes));
        if (*out == nullptr) {
          rc = fail(PTK_ERR_NOMEM, "out of memory");
        } else {
          he = hipMemcpy(*out, d_out, obytes, hipMemcpyDeviceToHost);
        }
      }
    }
  }
  if (rc == PTK_OK && he != hipSuccess) rc = fail(PTK_ERR_DEVICE, "HIP error: %s", hipGetErrorString(he));
  if (rc != PTK_OK && *out) {
    std::free(*out);
    *out = nullptr;
  }
  return rc;
}

int ptk_search64_box(const ptk_tree64* t, const double* mins, const double* maxs, uint64_t nb, uint64_t* offsets,
                     int32_t** out) {
  if (out == nullptr || offsets == nullptr) return fail(PTK_ERR_INVALID, "null output pointer");
  *out = nullptr;
  int rc = check_search64(t, mins, nb);
  if (rc != PTK_OK) return rc;
  if (nb > 0 && maxs == nullptr) return fail(PTK_ERR_INVALID, "null box buffer");
  offsets[0] = 0;
  if (nb == 0) return PTK_OK;
  DeviceGuard guard(t->device);
  if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", t->device);
  double *d_mn = nullptr, *d_mx = nullptr;
  uint64_t *d_c = nullptr, *d_o = nullptr;
  int32_t* d_out = nullptr;
  const size_t bbytes = (size_t)nb * t->dim * sizeof(double);
  hipError_t he = hipMalloc((void**)&d_mn, bbytes);
  if (he == hipSuccess) he = hipMalloc((void**)&d_mx, bbytes);
  if (he == hipSuccess) he = hipMalloc((void**)&d_c, (nb + 1) * 8);
  if (he == hipSuccess) he = hipMalloc((void**)&d_o, (nb + 1) * 8);
  if (he == hipSuccess) he = hipMemset(d_c, 0, (nb + 1) * 8);
  if (he == hipSuccess) he = hipMemcpy(d_mn, mins, bbytes, hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(d_mx, maxs, bbytes, hipMemcpyHostToDevice);
  if (he == hipSuccess) {
    Stack64Lease lease(t, nullptr);
    rc = lease.acquire(nb);
    if (rc == PTK_OK) rc = launch_box64<false>(t, d_mn, d_mx, nb, d_c, nullptr, nullptr, nullptr, lease);
    if (rc == PTK_OK) rc = scan_counts64(d_c, d_o, nb, offsets);
    if (rc == PTK_OK) {
      const size_t obytes = std::max<uint64_t>(offsets[nb], 1) * sizeof(int32_t);
      he = hipMalloc((void**)&d_out, obytes);
      if (he == hipSuccess) rc = launch_box64<true>(t, d_mn, d_mx, nb, nullptr, d_o, d_out, nullptr, lease);
      if (he == hipSuccess && rc == PTK_OK) {
        *out = static_cast<int32_t*>(std::malloc(obytes));
        if (*out == nullptr) {
          rc = fail(PTK_ERR_NOMEM, "out of memory");
        } else {
          he = hipMemcpy(*out, d_out, obytes, hipMemcpyDeviceToHost);
        }
      }
    }
  }
  if (d_mn) (void)hipFree(d_mn);
  if (d_mx) (void)hipFree(d_mx);
  if (d_c) (void)hipFree(d_c);
  if (d_o) (void)hipFree(d_o);
  if (d_out) (void)hipFree(d_out);
  if (rc == PTK_OK && he != hipSuccess) rc = fail(PTK_ERR_DEVICE, "HIP error: %s", hipGetErrorString(he));
  if (rc != PTK_OK && *out) {
    std::free(*out);
    *out = nullptr;
  }
  return rc;
}

}  // extern "C"
