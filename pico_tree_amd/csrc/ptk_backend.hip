// ptk_backend.hip -- host side of libptk.so: the C ABI of include/ptk.h.
//
// Responsibilities: validate arguments, build (optionally) and re-encode the flat
// tree for the device, keep it resident in HBM, order query batches, launch the
// gfx950 kernels of ptk_kernels.hpp, and move results.  There is no CPU search
// path in this file: without a usable device every search entry point fails with
// PTK_ERR_DEVICE.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <random>
#include <sstream>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "ptk.h"
#include "ptk_hostio.hpp"
#include "ptk_encode.hpp"
#include "ptk_kernels.hpp"
#include "ptk_kernels_lists.hpp"
#include "ptk_kernels_coopk.hpp"
#include "ptk_piles.hpp"
// Geometry of the launches (measured optima, profiles/r02_notes.txt item 23, r03_notes.txt item 13):
constexpr int kP2Ring = 12;   // LDS ring of the capped phase 2 (records per lane; 8 / 10 / 16: 1.335 / 1.329 / 1.308 vs 1.224 ms)
#ifndef PTK_GEN_RING
#define PTK_GEN_RING 16
#endif
constexpr int kGenRing = PTK_GEN_RING;  // LDS ring of the general searches (k > 1, radius)
constexpr int kGenLeafB = 5;  // points per leaf round of the general searches (4 / 5 / 6: knn = 16 4.85 / 4.71 / 4.68 ms, radius capture 7.21 / 7.16 / 7.69)
#include "ptk_build.hpp"
#include "ptk_sort.hpp"
#include "ptk_kernels_nd.hpp"
#include "ptk_kernels_topo.hpp"
#include "ptk_forest.hpp"

// Host-side builder: the product's own header-only flat-tree builder.
#include "pico_tree/internal/flat_tree.hpp"
#include "pico_tree/internal/stream.hpp"
#include "pico_tree/map.hpp"

static_assert(sizeof(ptk_neighbor) == 8 && sizeof(ptk::Neighbor) == 8, "neighbor layout");
static_assert(sizeof(ptk_node) == 16, "node layout");
static_assert(
    sizeof(pico_tree::internal::flat_node<int, float>) == sizeof(ptk_node), "flat node layout");

namespace {

thread_local std::string g_error = "";

int fail(int status, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return status;
}

#define PTK_HIP(expr)                                                              \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      return fail(PTK_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    }                                                                              \
  } while (0)

constexpr int kDeviceNone = -2;  // handle without a device replica (host tools, CPU-only tests)

struct PendingEvent {
  hipEvent_t a, b;
  int kind;  // 0 search, 1 reorder, 2 other, 3 search (continuation of the same launch)
  uint64_t queries;
  bool keep_b;  // `b` is also the `a` of the next section (Timer::next), which recycles it
};

struct Profile {
  std::mutex mutex;
  std::atomic<bool> enabled{false};
  ptk_profile acc{};
  std::vector<PendingEvent> pending;  // recorded, not yet read back
  std::vector<hipEvent_t> idle;       // events ready for reuse (hipEventCreate is slow)
};

// Device scratch of a handle: ONE grow-only HBM block, bump-allocated per search call.
// A batch of BASELINE config 2 needs ~0.8 GB of transient arrays (packed queries, sort
// double buffers, continuation records); asking the runtime for them on every call left
// the GPU idle for ~0.6 ms per step (profiles/r01c_two_phase_timeline.txt), so they are
// kept.  Calls on one handle enqueue under `mutex`; the block is reused in stream order,
// and a call that arrives on a DIFFERENT stream first waits for everything the last stream
// holds: the event for that is recorded on the last stream when the switch happens, not at
// the end of every call (an event between two calls costs the GPU ~4 us of a 0.3 ms
// search).  A stream searches were issued on has to be synchronised before it is destroyed.
struct Workspace {
  std::mutex mutex;
  char* base = nullptr;
  size_t capacity = 0;
  size_t used = 0;
  hipStream_t last_stream = nullptr;
  hipEvent_t done = nullptr;
  bool has_work = false;
  const uint32_t* last_meta = nullptr;  // Cont::meta of the last two-phase k = 1 search (ptk_debug_knn1_counts)
  // Second stream of a small k = 1 batch: the cooperative search of the ranked classes runs beside phase 2
  // (launch_knn1_two_phase); forked and joined with events, so the caller's stream still orders everything.
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  // The coherence sample of a batch (ptk::coherence_sample_kernel): its state and verdict stay on the device.
  uint32_t* d_sample = nullptr;             // device: {windows counted << 16 | windows failed, verdict}, zero between batches
  const uint32_t* last_verdict = nullptr;   // the verdict word of the last batch on this block, if it was sampled
  int last_order = 0;  // the last batch on this block: 0 = taken as it came, 1 = sorted on the device (2 = found coherent
                       // and left alone is only known on the device: last_verdict, ptk_debug_batch_order)

  // Rows captured by the last radius count pass (ptk::RadiusCapture): a block of its own, because
  // it must survive until the fill pass of the same batch while other searches reuse `base`.
  // The key is what the fill pass must repeat to be served from it.
  char* cap_base = nullptr;
  size_t cap_capacity = 0;
  bool cap_valid = false;
  bool cap_lists = false;  // the capture holds leaf lists (ptk_kernels_lists.hpp), not a log of hits
  ptk::RadiusCapture cap{};
  const float* cap_q = nullptr;
  uint64_t cap_nq = 0;
  float cap_radius = 0.0f, cap_e = 0.0f;
  int cap_metric = 0;
  hipStream_t cap_stream = nullptr;
};

// Staging of the host-buffer entry points (ptk_search_knn with host pointers, see ptk_hostio.hpp): device blocks
// for a whole batch, rings of pinned host pieces, streams, events and the copy threads, kept with the handle so that
// a call costs copies and searches, not allocations (hipFree synchronises the device; pinning memory takes
// milliseconds).  Calls that use it are serialised by `mutex`.
struct HostIo {
  static constexpr int kRing = 3;  // pinned pieces per direction
  std::mutex mutex;
  hipStream_t up = nullptr, down = nullptr;
  hipStream_t search[2] = {nullptr, nullptr};
  hipEvent_t up_done[kRing] = {}, down_done[kRing] = {};
  std::vector<hipEvent_t> searched;  // one per piece of the batch in flight
  char* d_in = nullptr;
  char* d_out = nullptr;
  size_t in_capacity = 0, out_capacity = 0;
  char* h_in[kRing] = {};
  char* h_out[kRing] = {};
  size_t h_in_capacity = 0, h_out_capacity = 0;  // bytes per ring slot
  std::unique_ptr<CopyPool> pool;
};

}  // namespace

struct ptk_tree {
  // host copy of the flat tree (DFS stream as handed in / built)
  uint32_t dim = 0;
  uint64_t n_points = 0;
  std::vector<ptk_node> nodes;
  std::vector<int32_t> indices;
  std::vector<float> root_min, root_max;
  std::vector<float> outer;  // per node {left_min, right_max} (topological metrics); may be empty
  // The flat-tree view the host loop searches (ptk_host_loop.hpp), made on its first call and kept: the node and
  // index arrays are copied ONCE per handle, not once per ptk_host_search_* call.  Dropped when `outer` changes.
  mutable std::shared_ptr<const void> host_flat;
  mutable std::mutex host_flat_mutex;
  double axis_splits[3] = {0, 0, 0};  // mean number of splits per axis on a root-to-leaf path (point-weighted)
  bool builder_made = false;  // nodes / indices come from the library's own builder: n_leaves, max_leaf_count, max_depth and
                              // axis_splits are set and the stream needs no validation
  uint32_t max_depth = 0;
  uint64_t n_leaves = 0;
  uint32_t max_leaf_count = 0;
  double create_ms[3] = {0, 0, 0};  // host build | re-encoding + checks | upload + point gather (ptk_debug_create_phases)

  // device replica
  int device = kDeviceNone;
  ptk::DevTree dev{};
  void* d_nodes = nullptr;
  void* d_pts = nullptr;
  void* d_ranges = nullptr; // dim <= 3: subtree ranges for the box search
  void* d_axes = nullptr;   // dim > 3 only
  void* d_index = nullptr;  // dim > 3 only
  void* d_outer = nullptr;  // topological metrics only: float2 per branch
  // What the device is (hipDeviceProp_t at creation): launches are sized from this, not from "an MI355X has 256 CUs
  // of 160 KiB" -- a partitioned device (CPX: 32 CUs per logical GPU) or another part must not be oversubscribed.
  int cus = 256;                      // compute units
  size_t lds_per_cu = 160 * 1024;     // LDS of one CU
  size_t lds_per_block = 160 * 1024;  // most dynamic LDS one workgroup may ask for
  size_t hbm_bytes = 0;               // device memory in total
  // The k = 1 view of a tree that holds piles (ptk_piles.hpp; dim <= 3): branch records and subtree ranges of its
  // own, the point array of the tree; null / zero when the tree has no pile.
  ptk::DevTree dev1{};
  void* d_nodes1 = nullptr;
  void* d_ranges1 = nullptr;
  void* d_pile_of_point = nullptr;
  void* d_pile_recs = nullptr;
  uint32_t n_piles = 0;
  uint64_t pile_points = 0;
  uint32_t max_depth1 = 0;
  void* d_cells = nullptr;  // dim <= 3: which cells of a coarse Morton grid hold tree points (ptk::CellTable)
  ptk::CellTable cells{};
  ptk::DevTreeND dev_nd{};
  uint64_t device_bytes = 0;
  bool gpu_layout = false;

  std::atomic<int> reorder{PTK_REORDER_AUTO};
  std::atomic<int> metric{PTK_METRIC_L2_SQUARED};
  mutable Profile profile;
  mutable Workspace ws;
  // k-NN calls arriving on other HIP streams get a scratch block of their own (up to kExtraWs
  // streams per handle; further ones share `ws` in stream order), so that batches issued on several
  // streams overlap: the tail of one batch's phase 2 is a few long dependent chains with the machine
  // mostly idle (profiles/r01n_streams.jsonl).  slot_stream[i] is the stream slot i belongs to.
  static constexpr int kExtraWs = 3;
  mutable Workspace extra_ws[kExtraWs];
  mutable std::mutex slot_mutex;
  mutable hipStream_t slot_stream[1 + kExtraWs] = {};
  mutable bool slot_taken[1 + kExtraWs] = {};
  mutable HostIo io;
};

namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

int env_int(const char* name, int fallback);  // defined with the launch helpers below
void axis_bits(const ptk_tree* t, int bits, uint32_t b[3]);

// Host threads for the tree build (the result does not depend on it): PTK_BUILD_THREADS, else the
// hardware concurrency capped at 32.
unsigned build_threads() {
  const int v = env_int("PTK_BUILD_THREADS", 0);
  if (v > 0) return (unsigned)v;
  const unsigned hc = std::thread::hardware_concurrency();
  return hc == 0 ? 1u : (hc > 32u ? 32u : hc);
}

// PTK_CREATE_TIMING=1: the phases of a tree creation on stderr (tools/time_build.py).
struct CreateClock {
  bool on;
  double* sink;  // ptk_tree::create_ms (ptk_debug_create_phases), or null
  std::chrono::steady_clock::time_point t0;
  explicit CreateClock(double* phases = nullptr)
      : on(env_int("PTK_CREATE_TIMING", 0) != 0), sink(phases), t0(std::chrono::steady_clock::now()) {}
  // slot: 0 host build, 1 re-encoding for the device (+ stream checks), 2 upload + point gather on the device
  void lap(const char* what, int slot) {
    const auto t1 = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (sink != nullptr && slot >= 0) sink[slot] += ms;
    if (on) std::fprintf(stderr, "[ptk create] %-28s %8.2f ms\n", what, ms);
    t0 = t1;
  }
};

int analyse(ptk_tree& t) {
  if (t.builder_made) return PTK_OK;
  ptk::TreeStats st;
  std::string err = ptk::analyse_stream(t.dim, t.n_points, t.nodes.data(), t.nodes.size(), st, nullptr);
  if (!err.empty()) return fail(PTK_ERR_INVALID, "%s", err.c_str());
  t.n_leaves = st.n_leaves;
  t.max_leaf_count = st.max_leaf_count;
  t.max_depth = st.max_depth;
  if (t.dim <= 3) {  // how a root-to-leaf path divides space, per axis (weights: points per leaf)
    struct Frame {
      uint32_t right;
      uint32_t cnt[3];
    };
    std::vector<Frame> stack;
    uint32_t cnt[3] = {0, 0, 0};
    double sum[3] = {0, 0, 0};
    for (size_t i = 0; i < t.nodes.size(); ++i) {
      if (!stack.empty() && stack.back().right == i) {
        std::memcpy(cnt, stack.back().cnt, sizeof(cnt));
        stack.pop_back();
      }
      const ptk_node& nd = t.nodes[i];
      if (nd.right == PTK_LEAF) {
        int32_t b, e;
        std::memcpy(&b, &nd.a, 4);
        std::memcpy(&e, &nd.b, 4);
        for (int a = 0; a < 3; ++a) sum[a] += (double)cnt[a] * (e - b);
      } else {
        ++cnt[nd.split_dim];
        Frame f;
        f.right = nd.right;
        std::memcpy(f.cnt, cnt, sizeof(cnt));
        stack.push_back(f);
      }
    }
    for (int a = 0; a < 3; ++a) t.axis_splits[a] = t.n_points ? sum[a] / (double)t.n_points : 0.0;
  }
  return PTK_OK;
}

int upload(ptk_tree& t, const float* points) {
  CreateClock clock(t.create_ms);
  int rc = analyse(t);
  if (rc != PTK_OK) return rc;
  clock.lap("analyse stream", 1);
  if (clock.on && t.dim <= 3)
    std::fprintf(stderr, "[ptk create] splits per root-to-leaf path: x %.2f  y %.2f  z %.2f (depth %u)\n", t.axis_splits[0],
                 t.axis_splits[1], t.axis_splits[2], t.max_depth);
  if (t.dim > 3) {  // any-dimension layout (ptk_kernels_nd.hpp)
    ptk::TreeStats st;
    ptk::EncodedTreeND enc;
    bool unsupported = false;
    std::string err = ptk::encode_tree_nd(t.dim, t.n_points, points, t.nodes.data(), t.nodes.size(),
                                          t.indices.data(), st, enc, unsupported);
    if (!err.empty()) return fail(unsupported ? PTK_ERR_UNSUPPORTED : PTK_ERR_INVALID, "%s", err.c_str());
    const size_t nb = enc.nodes.size() * sizeof(uint4), ab = enc.axes.size() * 4, pb = enc.points.size() * 4,
                 ib = enc.index.size() * 4;
    PTK_HIP(hipMalloc(&t.d_nodes, nb));
    PTK_HIP(hipMalloc(&t.d_axes, ab));
    PTK_HIP(hipMalloc(&t.d_pts, pb));
    PTK_HIP(hipMalloc(&t.d_index, ib));
    PTK_HIP(hipMemcpy(t.d_nodes, enc.nodes.data(), nb, hipMemcpyHostToDevice));
    PTK_HIP(hipMemcpy(t.d_axes, enc.axes.data(), ab, hipMemcpyHostToDevice));
    PTK_HIP(hipMemcpy(t.d_pts, enc.points.data(), pb, hipMemcpyHostToDevice));
    PTK_HIP(hipMemcpy(t.d_index, enc.index.data(), ib, hipMemcpyHostToDevice));
    const size_t rb = enc.ranges.size() * sizeof(ptk::EncRange);
    PTK_HIP(hipMalloc(&t.d_ranges, rb));
    PTK_HIP(hipMemcpy(t.d_ranges, enc.ranges.data(), rb, hipMemcpyHostToDevice));
    t.device_bytes = nb + ab + pb + ib + rb;
    t.dev_nd.nodes = static_cast<const uint4*>(t.d_nodes);
    t.dev_nd.axes = static_cast<const uint32_t*>(t.d_axes);
    t.dev_nd.pts = static_cast<const float*>(t.d_pts);
    t.dev_nd.index = static_cast<const int32_t*>(t.d_index);
    t.dev_nd.root_ref = enc.root_ref;
    t.dev_nd.cbits = enc.cbits;
    t.dev_nd.cmask = (1u << enc.cbits) - 1u;
    t.dev_nd.dim = t.dim;
    t.gpu_layout = true;
    return PTK_OK;
  }
  ptk::TreeStats st;
  ptk::EncodedTree enc;
  bool unsupported = false;
  // Branch records and references on the host; the 16-byte point records are gathered on the
  // device from the raw points and the leaf-order permutation (no host pass over the points).
  static_assert(ptk::kEncLeafAlign == 1, "encode_points_kernel assumes packed leaves");
  std::string err;
  if (t.builder_made) {
    st.n_leaves = t.n_leaves;
    st.max_leaf_count = t.max_leaf_count;
    st.max_depth = t.max_depth;
    err = ptk::encode_tree_of_builder(t.dim, t.n_points, t.nodes.data(), t.nodes.size(), st, enc, unsupported, build_threads());
  } else {
    err = ptk::encode_tree(t.dim, t.n_points, nullptr, t.nodes.data(), t.nodes.size(), t.indices.data(), st, enc, unsupported,
                           /*with_points=*/false);
  }
  if (!err.empty()) return fail(unsupported ? PTK_ERR_UNSUPPORTED : PTK_ERR_INVALID, "%s", err.c_str());
  clock.lap("encode branch records", 1);
  if (!t.builder_made) {
    for (int32_t idx : t.indices)
      if (idx < 0 || (uint64_t)idx >= t.n_points) return fail(PTK_ERR_INVALID, "index out of range in the permutation");
    clock.lap("check permutation", 1);
  }

  static_assert(sizeof(ptk::EncNode) == sizeof(uint4) && sizeof(ptk::EncPoint) == sizeof(float4), "records");
  const size_t n_records = t.n_points + ptk::kEncLeafPad;
  PTK_HIP(hipMalloc(&t.d_nodes, enc.nodes.size() * sizeof(uint4)));
  PTK_HIP(hipMalloc(&t.d_pts, n_records * sizeof(float4)));
  PTK_HIP(hipMemcpy(t.d_nodes, enc.nodes.data(), enc.nodes.size() * sizeof(uint4), hipMemcpyHostToDevice));
  {
    float* d_raw = nullptr;
    int32_t* d_idx = nullptr;
    const size_t raw_bytes = (size_t)t.n_points * t.dim * sizeof(float);
    hipError_t he = hipMalloc((void**)&d_raw, raw_bytes);
    if (he == hipSuccess) he = hipMalloc((void**)&d_idx, t.n_points * sizeof(int32_t));
    if (he == hipSuccess) he = hipMemcpy(d_raw, points, raw_bytes, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(d_idx, t.indices.data(), t.n_points * sizeof(int32_t), hipMemcpyHostToDevice);
    if (he == hipSuccess) {
      const uint32_t blocks = (uint32_t)((n_records + ptk::kBlock - 1) / ptk::kBlock);
      hipLaunchKernelGGL(ptk::encode_points_kernel, dim3(blocks), dim3(ptk::kBlock), 0, nullptr, d_raw, t.dim, d_idx,
                         t.n_points, static_cast<float4*>(t.d_pts));
      he = hipDeviceSynchronize();
    }
    if (d_raw) (void)hipFree(d_raw);
    if (d_idx) (void)hipFree(d_idx);
    if (he != hipSuccess) return fail(PTK_ERR_DEVICE, "HIP error while encoding the points: %s", hipGetErrorString(he));
  }
  size_t cell_bytes = 0;
  {
    // The coarse grid of occupied cells (ptk::CellTable): about 32 tree points per cell on average, the cell bits
    // spread over the axes like the bits of the order key.  PTK_CELL_TABLE=0: none.
    int cb = 0;
    while ((64ull << cb) <= t.n_points) ++cb;  // floor(log2(n / 32))
    cb = std::min(std::max(cb, 6), 22);
    if (env_int("PTK_CELL_TABLE", 1) != 0) {
      uint32_t b[3];
      axis_bits(&t, cb, b);
      float lo[3] = {0, 0, 0}, inv[3] = {0, 0, 0};
      for (uint32_t d = 0; d < t.dim && d < 3; ++d) {
        lo[d] = t.root_min[d];
        const float ext = t.root_max[d] - t.root_min[d];
        inv[d] = ext > 0 ? (float)(1u << b[d]) / ext : 0.0f;
      }
      const size_t n_cells = (size_t)1 << (b[0] + b[1] + b[2]);
      cell_bytes = n_cells;
      uint32_t* d_counts = nullptr;
      PTK_HIP(hipMalloc(&t.d_cells, n_cells));
      PTK_HIP(hipMalloc((void**)&d_counts, n_cells * 4));
      {
        const hipError_t he = hipMemsetAsync(d_counts, 0, n_cells * 4, nullptr);
        if (he != hipSuccess) {
          (void)hipFree(d_counts);
          return fail(PTK_ERR_DEVICE, "hipMemsetAsync failed: %s", hipGetErrorString(he));
        }
      }
      const uint32_t blocks = (uint32_t)((t.n_points + ptk::kBlock - 1) / ptk::kBlock);
      hipLaunchKernelGGL(ptk::cell_count_kernel, dim3(blocks), dim3(ptk::kBlock), 0, nullptr,
                         static_cast<const float4*>(t.d_pts), t.n_points, make_float3(lo[0], lo[1], lo[2]),
                         make_float3(inv[0], inv[1], inv[2]), make_uint3(b[0], b[1], b[2]), d_counts);
      hipLaunchKernelGGL(ptk::cell_class_kernel, dim3((uint32_t)((n_cells + ptk::kBlock - 1) / ptk::kBlock)),
                         dim3(ptk::kBlock), 0, nullptr, d_counts, n_cells, static_cast<uint8_t*>(t.d_cells));
      hipError_t he = hipGetLastError();  // (a launch that failed)
      const hipError_t he_sync = hipDeviceSynchronize();
      if (he == hipSuccess) he = he_sync;
      (void)hipFree(d_counts);
      if (he != hipSuccess) return fail(PTK_ERR_DEVICE, "HIP error while counting the grid cells: %s", hipGetErrorString(he));
      t.cells.occ = static_cast<const uint8_t*>(t.d_cells);
      t.cells.inv = make_float3(inv[0], inv[1], inv[2]);
      t.cells.bits = make_uint3(b[0], b[1], b[2]);
    }
  }
  clock.lap("upload + gather points", 2);
  PTK_HIP(hipMalloc(&t.d_ranges, enc.ranges.size() * sizeof(ptk::EncRange)));
  PTK_HIP(hipMemcpy(t.d_ranges, enc.ranges.data(), enc.ranges.size() * sizeof(ptk::EncRange), hipMemcpyHostToDevice));
  t.device_bytes = enc.nodes.size() * sizeof(uint4) + n_records * sizeof(float4) +
                   enc.ranges.size() * sizeof(ptk::EncRange) + cell_bytes;
  t.dev.nodes = static_cast<const uint4*>(t.d_nodes);
  t.dev.pts = static_cast<const float4*>(t.d_pts);
  t.dev.root_ref = enc.root_ref;
  t.dev.cbits = enc.cbits;
  t.dev.cmask = (1u << enc.cbits) - 1u;
  t.dev.n_points = (uint32_t)t.n_points;
  t.gpu_layout = true;
  // Piles -- subtrees of one point many times over: the k = 1 search gets a view in which each is a leaf of one point
  // (ptk_piles.hpp).  A tree of points in general position pays one pass over its branch records here.
  if (env_int("PTK_PILE_VIEW", 1) != 0) {
    ptk::PileView view;
    ptk::build_pile_view(t.dim, t.n_points, points, t.nodes.data(), t.nodes.size(), t.indices.data(), view, build_threads());
    if (!view.empty()) {
      ptk::TreeStats st1;
      ptk::EncodedTree enc1;
      bool unsup1 = false;
      const std::string err1 = ptk::encode_tree(t.dim, t.n_points, nullptr, view.nodes.data(), view.nodes.size(), t.indices.data(),
                                                st1, enc1, unsup1, /*with_points=*/false, view.single.data(), enc.cbits);
      if (err1.empty()) {
        static_assert(sizeof(ptk::PileRecord) == sizeof(ptk::DevPileRecord) && sizeof(ptk::PileRecord) == 48, "pile records");
        const size_t nb = enc1.nodes.size() * sizeof(uint4), rb = enc1.ranges.size() * sizeof(ptk::EncRange),
                     ob = view.pile_of_point.size() * 4, pb = view.piles.size() * sizeof(ptk::PileRecord);
        PTK_HIP(hipMalloc(&t.d_nodes1, nb));
        PTK_HIP(hipMalloc(&t.d_ranges1, rb));
        PTK_HIP(hipMalloc(&t.d_pile_of_point, ob));
        PTK_HIP(hipMalloc(&t.d_pile_recs, pb));
        PTK_HIP(hipMemcpy(t.d_nodes1, enc1.nodes.data(), nb, hipMemcpyHostToDevice));
        PTK_HIP(hipMemcpy(t.d_ranges1, enc1.ranges.data(), rb, hipMemcpyHostToDevice));
        PTK_HIP(hipMemcpy(t.d_pile_of_point, view.pile_of_point.data(), ob, hipMemcpyHostToDevice));
        PTK_HIP(hipMemcpy(t.d_pile_recs, view.piles.data(), pb, hipMemcpyHostToDevice));
        t.dev1 = t.dev;
        t.dev1.nodes = static_cast<const uint4*>(t.d_nodes1);
        t.dev1.root_ref = enc1.root_ref;
        t.n_piles = (uint32_t)view.piles.size();
        t.pile_points = view.pile_points;
        t.max_depth1 = st1.max_depth;
        t.device_bytes += nb + rb + ob + pb;
      }  // (a view that cannot be encoded is not needed: the full tree serves every search)
      if (clock.on)
        std::fprintf(stderr, "[ptk create] piles: %zu holding %llu points, view of %zu nodes, depth %u%s%s\n", view.piles.size(),
                     (unsigned long long)view.pile_points, view.nodes.size(), st1.max_depth, err1.empty() ? "" : " -- not used: ",
                     err1.c_str());
    }
    clock.lap("pile view", 1);
  }
  return PTK_OK;
}

// The first kernel launch on a device loads libptk's code object for it (~10 MB: 0.17 s on the bench box,
// profiles/r02_notes.txt item 13).  That load can be started ahead of time on a thread of the library -- by
// ptk_warmup(device), or by the first creation for the device -- so that it runs beside whatever the caller does before
// its first tree (reading the points) and beside the host part of that creation.  One load per device and process;
// wait(device) before the first launch of the calling thread.  PTK_EAGER_WARMUP=0: no thread, the first launch loads.
// (A query about devices -- ptk_device_count() -- starts nothing: in a job of one process per GPU every rank counts the
// devices before it picks its own, and none of them wants a context on device 0.)
__global__ void ptk_warm_kernel() {}
struct ProcessWarmup {
  static constexpr int kMaxDevices = 64;
  std::mutex lock;
  std::thread threads[kMaxDevices];
  bool started[kMaxDevices] = {};
  void start(int32_t device) {
    if (device == kDeviceNone) return;
    const char* sw = std::getenv("PTK_EAGER_WARMUP");
    if (sw != nullptr && std::atoi(sw) == 0) return;
    int dev = device;  // (the device the CALLER is on, not the new thread's default)
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return;
    if (dev < 0 || dev >= kMaxDevices) return;
    std::lock_guard<std::mutex> hold(lock);
    if (started[dev]) return;
    started[dev] = true;
    threads[dev] = std::thread([dev] {
      if (hipSetDevice(dev) != hipSuccess) {
        (void)hipGetLastError();
        return;
      }
      hipLaunchKernelGGL(ptk_warm_kernel, dim3(1), dim3(1), 0, nullptr);
      (void)hipDeviceSynchronize();
      (void)hipGetLastError();
    });
  }
  void wait(int32_t device) {
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return;
    if (dev < 0 || dev >= kMaxDevices) return;
    std::lock_guard<std::mutex> hold(lock);
    if (threads[dev].joinable()) threads[dev].join();
  }
  ~ProcessWarmup() {
    for (auto& th : threads)
      if (th.joinable()) th.join();
  }
};
ProcessWarmup g_warmup;

int finish_create(ptk_tree* t, const float* points, int32_t device, ptk_tree** out) {
  if (t->root_min.empty()) {  // start bounds not supplied: bounding box of the points
    t->root_min.assign(t->dim, std::numeric_limits<float>::max());
    t->root_max.assign(t->dim, std::numeric_limits<float>::lowest());
    for (uint64_t i = 0; i < t->n_points; ++i)
      for (uint32_t d = 0; d < t->dim; ++d) {
        const float v = points[i * t->dim + d];
        t->root_min[d] = std::min(t->root_min[d], v);
        t->root_max[d] = std::max(t->root_max[d], v);
      }
  }
  if (device == kDeviceNone) {
    int rc = analyse(*t);
    if (rc != PTK_OK) {
      delete t;
      return rc;
    }
    t->device = kDeviceNone;
    *out = t;
    return PTK_OK;
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    delete t;
    return fail(PTK_ERR_DEVICE, "no HIP device is visible");
  }
  int dev = device;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev >= count) {
    delete t;
    return fail(PTK_ERR_INVALID, "device %d out of range (%d visible)", dev, count);
  }
  t->device = dev;
  DeviceGuard guard(dev);
  if (!guard.ok) {
    delete t;
    return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", dev);
  }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
      if (prop.multiProcessorCount > 0) t->cus = prop.multiProcessorCount;
      if (prop.maxSharedMemoryPerMultiProcessor > 0) t->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
      const size_t optin = prop.sharedMemPerBlockOptin > 0 ? (size_t)prop.sharedMemPerBlockOptin : (size_t)prop.sharedMemPerBlock;
      if (optin > 0) t->lds_per_block = std::min(optin, t->lds_per_cu);
      t->hbm_bytes = prop.totalGlobalMem;
    } else {
      (void)hipGetLastError();
    }
  }
  int rc = upload(*t, points);
  if (rc != PTK_OK) {
    ptk_tree_destroy(t);
    return rc;
  }
  *out = t;
  return PTK_OK;
}

// ---- launch helpers ---------------------------------------------------------------

// Optional HIP-event bracket around one kernel.  Recording is asynchronous: the
// pair is queued on the handle and only read back (with a synchronisation) by
// ptk_profile_get, so profiling may stay enabled inside a timed region.
struct Timer {
  const ptk_tree* t;
  hipStream_t s;
  hipEvent_t a = nullptr, b = nullptr;
  bool on;
  Timer(const ptk_tree* tree, hipStream_t stream) : t(tree), s(stream) {
    on = tree->profile.enabled;
    if (on) {
      {
        std::lock_guard<std::mutex> lock(t->profile.mutex);
        std::vector<hipEvent_t>& idle = t->profile.idle;
        if (idle.size() >= 2) {
          a = idle.back();
          idle.pop_back();
          b = idle.back();
          idle.pop_back();
        }
      }
      // Timing only: no system-scope fence when the event completes (the default event makes the device write
      // its caches back, which costs the kernels around it: 0.042 ms per k = 1 search with ten such events).
      if (a == nullptr)
        on = hipEventCreateWithFlags(&a, hipEventDisableSystemFence) == hipSuccess &&
             hipEventCreateWithFlags(&b, hipEventDisableSystemFence) == hipSuccess;
      if (on) (void)hipEventRecord(a, s);
    }
  }
  void stop(int kind, uint64_t queries) {
    if (!on) return;
    (void)hipEventRecord(b, s);
    std::lock_guard<std::mutex> lock(t->profile.mutex);
    t->profile.pending.push_back(PendingEvent{a, b, kind, queries, false});
    a = b = nullptr;
    on = false;
  }
  // Ends a section and begins the next at the same instant: ONE event between two kernels instead of two
  // (an event between dependent launches costs ~3 us of idle device).
  void next(int kind, uint64_t queries) {
    if (!on) return;
    (void)hipEventRecord(b, s);
    hipEvent_t fresh = nullptr;
    {
      std::lock_guard<std::mutex> lock(t->profile.mutex);
      t->profile.pending.push_back(PendingEvent{a, b, kind, queries, true});
      if (!t->profile.idle.empty()) {
        fresh = t->profile.idle.back();
        t->profile.idle.pop_back();
      }
    }
    a = b;  // ours now: the section that ended does not recycle it
    b = fresh;
    if (b == nullptr && hipEventCreateWithFlags(&b, hipEventDisableSystemFence) != hipSuccess) {
      b = nullptr;
      on = false;  // (the destructor drops `a`; the ended section only reads it before that if it is resolved first)
    }
  }
  ~Timer() {  // only reached with events in hand when a search failed half-way
    if (a) {
      // After next() `a` is also the end of the section before (kept there with keep_b): that section takes it
      // over instead of being left with a destroyed event.
      std::lock_guard<std::mutex> lock(t->profile.mutex);
      for (auto it = t->profile.pending.rbegin(); it != t->profile.pending.rend(); ++it)
        if (it->b == a && it->keep_b) {
          it->keep_b = false;
          a = nullptr;
          break;
        }
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
  }
};

// One search call's view of the handle's scratch block (see Workspace).
// The scratch block a call on stream `s` uses: the handle's main one, or -- k-NN calls only (the
// radius capture lives in the main block) -- the one assigned to that stream.
inline Workspace& workspace_for(const ptk_tree* t, hipStream_t s, bool per_stream) {
  if (!per_stream) return t->ws;
  std::lock_guard<std::mutex> lock(t->slot_mutex);
  for (int i = 0; i <= ptk_tree::kExtraWs; ++i)
    if (t->slot_taken[i] && t->slot_stream[i] == s) return i == 0 ? t->ws : t->extra_ws[i - 1];
  for (int i = 0; i <= ptk_tree::kExtraWs; ++i)
    if (!t->slot_taken[i]) {
      t->slot_taken[i] = true;
      t->slot_stream[i] = s;
      return i == 0 ? t->ws : t->extra_ws[i - 1];
    }
  return t->ws;
}

// Everything enqueued for the block's last user has finished (host wait).
inline void drain_workspace(Workspace& ws) {
  if (!ws.has_work) return;
  if (hipStreamSynchronize(ws.last_stream) != hipSuccess) {  // (the stream is gone: whatever it held is waited for)
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();
  }
  ws.has_work = false;
}

class Scratch {
 public:
  Scratch(const ptk_tree* t, hipStream_t s, bool per_stream = false)
      : ws_(workspace_for(t, s, per_stream)), lock_(ws_.mutex), s_(s) {}
  ~Scratch() {
    if (!reserved_) return;
    ws_.last_stream = s_;
    ws_.has_work = true;
  }
  // Room for `bytes` in total over all take() calls of this search; orders the call
  // after the previous user of the block.
  int reserve(size_t bytes) {
    bytes += 64 * kAlign;  // alignment slack of the individual arrays
    if (bytes > ws_.capacity) {
      drain_workspace(ws_);
      if (ws_.base) (void)hipFree(ws_.base);
      ws_.base = nullptr;
      ws_.capacity = 0;
      ws_.has_work = false;
      const size_t want = (bytes + (size_t(32) << 20)) & ~((size_t(32) << 20) - 1);
      if (hipMalloc((void**)&ws_.base, want) != hipSuccess) {
        (void)hipGetLastError();  // not sticky: the next launch must not report this again
        ws_.base = nullptr;
        return fail(PTK_ERR_NOMEM, "out of device memory (%zu bytes of search scratch)", want);
      }
      ws_.capacity = want;
    }
    if (ws_.has_work && ws_.last_stream != s_) {  // the block changes streams: behind all the last one holds
      bool ordered = ws_.done != nullptr || hipEventCreateWithFlags(&ws_.done, hipEventDisableTiming) == hipSuccess;
      ordered = ordered && hipEventRecord(ws_.done, ws_.last_stream) == hipSuccess &&
                hipStreamWaitEvent(s_, ws_.done, 0) == hipSuccess;
      if (!ordered) {
        (void)hipGetLastError();
        drain_workspace(ws_);
      }
    }
    ws_.used = 0;
    ws_.last_meta = nullptr;  // whatever the last k = 1 search left in the block is about to be overwritten (or freed)
    ws_.last_order = 0;
    ws_.last_verdict = nullptr;
    reserved_ = true;
    return PTK_OK;
  }
  void note_meta(const uint32_t* meta) { ws_.last_meta = meta; }
  void note_order(int how) { ws_.last_order = how; }
  // The second stream of this scratch block and its two events (made on first use); false if they cannot be had.
  bool side_stream(hipStream_t* side, hipEvent_t* fork, hipEvent_t* join) {
    if (ws_.side == nullptr) {
      // (at the device's highest stream priority instead: measured, no different -- shard 0.228 / 0.228 ms, the
      // headline 1.23 / 1.23, knn = 16 3.76 / 3.76: profiles/r05_notes.txt item 18)
      if (hipStreamCreateWithFlags(&ws_.side, hipStreamNonBlocking) != hipSuccess) {
        ws_.side = nullptr;
        (void)hipGetLastError();
        return false;
      }
    }
    if (ws_.fork == nullptr && hipEventCreateWithFlags(&ws_.fork, hipEventDisableTiming) != hipSuccess) ws_.fork = nullptr;
    if (ws_.join == nullptr && hipEventCreateWithFlags(&ws_.join, hipEventDisableTiming) != hipSuccess) ws_.join = nullptr;
    if (ws_.fork == nullptr || ws_.join == nullptr) {
      (void)hipGetLastError();
      return false;
    }
    *side = ws_.side;
    *fork = ws_.fork;
    *join = ws_.join;
    return true;
  }
  // The two device words of the coherence sample of this block's batches (made and zeroed on first use).
  uint32_t* sample_state() {
    if (ws_.d_sample == nullptr) {
      if (hipMalloc((void**)&ws_.d_sample, 256) != hipSuccess || hipMemset(ws_.d_sample, 0, 256) != hipSuccess) {
        (void)hipGetLastError();
        if (ws_.d_sample) (void)hipFree(ws_.d_sample);
        ws_.d_sample = nullptr;
      }
    }
    return ws_.d_sample;
  }
  void note_verdict(const uint32_t* verdict) { ws_.last_verdict = verdict; }
  const uint32_t* batch_verdict() const { return ws_.last_verdict; }
  template <class T>
  T* take(size_t count) {
    const size_t bytes = (count * sizeof(T) + kAlign - 1) & ~(kAlign - 1);
    if (!reserved_ || ws_.used + bytes > ws_.capacity) return nullptr;  // reserve() was too small: a bug
    T* p = reinterpret_cast<T*>(ws_.base + ws_.used);
    ws_.used += bytes;
    return p;
  }
  static constexpr size_t kAlign = 256;

 private:
  Workspace& ws_;
  std::unique_lock<std::mutex> lock_;
  hipStream_t s_;
  bool reserved_ = false;
};

// Stack geometry: the newest S records of a lane live in an LDS ring, older ones
// spill to OVF private-scratch slots.  A traversal holds, per level of the current
// root path, either one pending record (went near, far child unexplored) or two
// undo records (went far), so 2 * depth + 2 slots always suffice.
constexpr int kDeepClass = 3;  // deeper than the private classes: spill to HBM, generic kernels only
// What a k = 1 search of the default metric traverses: the view without the piles if the tree has any.
const ptk::DevTree& knn1_tree(const ptk_tree* t) { return t->n_piles ? t->dev1 : t->dev; }
const uint2* knn1_ranges(const ptk_tree* t) { return static_cast<const uint2*>(t->n_piles ? t->d_ranges1 : t->d_ranges); }
uint32_t knn1_depth(const ptk_tree* t) { return t->n_piles ? t->max_depth1 : t->max_depth; }

int ovf_class_of(uint32_t depth, int s_lds);
int ovf_class(const ptk_tree* t, int s_lds) { return ovf_class_of(t->max_depth, s_lds); }
int ovf_class_of(uint32_t depth, int s_lds) {
  const uint32_t need = 2 * depth + 2;
  if (need <= (uint32_t)s_lds + 64) return 0;
  if (need <= (uint32_t)s_lds + 256) return 1;
  if (need <= (uint32_t)s_lds + 2048) return 2;
  return kDeepClass;
}
bool deep_tree(const ptk_tree* t) { return ovf_class(t, 16) == kDeepClass; }

// A deep tree's launches: `cap` spill records per lane, `piece` queries per launch so that the
// block stays within PTK_DEEP_SPILL_MB (default 2048).
struct DeepPlan {
  uint32_t cap;
  uint64_t piece;
  size_t bytes() const { return (size_t)piece * cap * sizeof(ptk::Record); }
};
DeepPlan deep_plan(const ptk_tree* t, uint64_t n) {
  DeepPlan p;
  p.cap = 2 * t->max_depth + 2;
  const size_t budget = (size_t)std::max(1, env_int("PTK_DEEP_SPILL_MB", 2048)) << 20;
  uint64_t piece = (budget / ((size_t)p.cap * sizeof(ptk::Record))) & ~(uint64_t)63;
  if (piece < 64) piece = 64;
  const uint64_t all = (n + 63) & ~(uint64_t)63;
  p.piece = piece < all ? piece : all;
  return p;
}

// Dynamic LDS above 64 KiB must be opted into per kernel.
template <typename K>
int allow_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    PTK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  }
  return PTK_OK;
}

// An integer from the environment (the memory caps and test switches listed in INTEGRATION.md section 6).
int env_int(const char* name, int fallback) {
  const char* v = std::getenv(name);
  return v ? std::atoi(v) : fallback;
}

bool want_reorder(const ptk_tree* t, uint64_t nq) {
  const int mode = t->reorder.load();
  if (mode == PTK_REORDER_ON) return nq > 1;
  if (mode == PTK_REORDER_OFF) return false;
  return nq >= 8192;
}

// Bits of the Morton key the batch is sorted by.  The search only needs neighbouring lanes to walk neighbouring
// leaves: 24 bits -- three 8-bit radix passes -- spread over the axes the way the tree itself divides space
// (axis_bits below) order the batch as well as 30 bits spent evenly do, for one pass less
// (profiles/r02_notes.txt items 18 and 20).  A small batch is thin in space anyway and pays ~35 us per pass in fixed
// costs: 16 bits (two passes) below 1 M queries (one eighth of BASELINE config 2: 0.537 vs 0.560 ms per step; at
// 1.8 M queries the two are equal, at 3.6 M 24 bits win by 6 %; item 29).
int morton_bits(uint64_t nq) { return nq < (1ull << 20) ? 16 : 24; }

// `bits` key bits over the three axes in proportion to how often a root-to-leaf path splits on each (at most 15
// per axis).  A cloud that is flat along one axis -- most of a LiDAR scan is floor -- gets few bits there and finer
// cells in the plane; a uniform cube gets bits / 3 each.
void axis_bits(const ptk_tree* t, int bits, uint32_t b[3]) {
  b[0] = b[1] = b[2] = 0;
  const uint32_t axes = t->dim < 3 ? t->dim : 3;
  double want[3] = {0, 0, 0};
  const double total = t->axis_splits[0] + t->axis_splits[1] + t->axis_splits[2];
  for (uint32_t a = 0; a < axes; ++a) want[a] = total > 0.0 ? bits * t->axis_splits[a] / total : (double)bits / axes;
  for (int given = 0; given < bits; ++given) {  // largest remaining share first
    int best = -1;
    for (uint32_t a = 0; a < axes; ++a)
      if (b[a] < 15 && (best < 0 || want[a] - b[a] > want[best] - b[best])) best = (int)a;
    if (best < 0) break;
    ++b[best];
  }
}

// rocprim switches from the onesweep radix sort to a merge sort below 1 M items by default -- 24 launches
// and 0.16 ms for the 900 k queries of one eighth of BASELINE config 2 (a shard of configs[3]), where three
// onesweep passes take 0.05 ms (profiles/r02_notes.txt item 17).  The limit is lowered to 128 k items.
using MortonSortConfig =
    rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 131072>;

size_t sort_tmp_bytes(uint64_t nq, int bits) {
  size_t tmp_bytes = 0;
  uint32_t* k32 = nullptr;
  (void)rocprim::radix_sort_pairs<MortonSortConfig>(nullptr, tmp_bytes, k32, k32, k32, k32, nq, 0, bits,
                                                    (hipStream_t) nullptr);
  return tmp_bytes + 256;
}

// The library's own radix sort (ptk_sort.hpp): tiles of `tile` items, one wavefront each.  At most 4096 tiles
// (a row of the digit-by-tile histogram is scanned by one wavefront), at least 512 items per tile (256 / 512 / 1024:
// 68 / 58 / 60 us for 900 k queries; PTK_SORT_TILE for experiments).
uint32_t sort_tile(uint64_t nq) {
  const uint64_t t = ((nq + 4095) / 4096 + 63) & ~(uint64_t)63;
  return (uint32_t)std::max<uint64_t>(t, (uint64_t)std::max(64, env_int("PTK_SORT_TILE", 512)) & ~(uint64_t)63);
}
uint32_t sort_tiles(uint64_t nq) { return (uint32_t)((nq + sort_tile(nq) - 1) / sort_tile(nq)); }
uint32_t sort_stride(uint64_t nq) { return (sort_tiles(nq) + 3u) & ~3u;  }  // row of the histogram: 16-byte steps
size_t own_sort_bytes(uint64_t nq) {  // (the histogram of whichever form has more tiles)
  const size_t stride = std::max<size_t>(sort_stride(nq), ((nq + ptk::kSortTile - 1) / ptk::kSortTile + 3) & ~(size_t)3);
  return ((size_t)ptk::kRadixBins * stride + ptk::kRadixBins) * 4 + 2 * nq * sizeof(uint2) + 1024;
}

// Which sort orders the batch: the library's own (ptk_sort.hpp) -- its passes with one wavefront per tile below 0.75 M
// rows (six launches and 58 us for a 900 k-query shard against rocprim's ten launches and 96 us), with blocks of eight
// wavefronts on tiles of 4 096 items from there on (reorder ms, one wavefront per tile / blocks / rocprim's onesweep:
// 900 k rows 0.063 / 0.056 / 0.110, 2 M 0.144 / 0.093 / 0.158, 7.2 M 0.319 / 0.192 / 0.275; profiles/r04_notes.txt
// item 13).  PTK_SORT = 0 forces rocprim's, PTK_SORT_BLOCK = 0 / 1 either form of the own.
bool block_sort(uint64_t nq) {
  const int mode = env_int("PTK_SORT_BLOCK", -1);
  return mode < 0 ? nq >= (3ull << 18) : mode != 0 && nq >= ptk::kSortTile;
}
bool own_sort(uint64_t nq) {
  if (nq >= (1ull << 31)) return false;
  return env_int("PTK_SORT", 1) != 0;
}

size_t permutation_scratch_bytes(uint64_t nq) { return 4 * (nq * 4) + sort_tmp_bytes(nq, 30) + own_sort_bytes(nq) + 1024; }

// Device-side Morton ordering of a batch: *perm (device, nq uint32, in `scratch`) lists the
// query rows in launch order.
// heavy_first (ptk::kCellsEmptyFirst / kCellsDenseFirst, 0 = plain Morton order): the queries that will be expensive
// -- by the tree's coarse grid of cell occupancies, ptk::CellTable -- go to the front of the order: for the kernels that
// run every query to its end in its lane.
// may_skip: the batch is sampled first (ptk::coherence_sample_kernel: 256 windows of 64 consecutive rows); if it is
// already in a coherent order -- a scan in scan order, a batch the caller sorted -- the kernels of the sort leave at
// their first instruction and phase 1 of the search takes the rows in the caller's order, as the reference does
// (_pyco_tree/kd_tree.hpp:128-134).  The verdict never leaves the device (two words of the workspace): the entry points
// that take device buffers only enqueue, whatever the batch looks like (r04 waited for the verdict on the host).
const uint32_t* sample_batch(const ptk_tree* t, const float* d_q, uint64_t nq, hipStream_t s, Scratch& scratch, int bits,
                             const float3& lo3, const float3& inv3, const uint3& b3) {
  if (nq < 8192 || env_int("PTK_COHERENCE_CHECK", 1) == 0) return nullptr;
  uint8_t* d_fail = scratch.take<uint8_t>(ptk::kCoherenceWindows);
  uint32_t* state = scratch.sample_state();
  if (d_fail == nullptr || state == nullptr) return nullptr;
  // 64 neighbours of a sorted batch cover about 2^bits x 64 / nq cells; five more bits of slack (the box of a window is
  // rounded up per axis, and a window may sit across a cell boundary).
  uint32_t lg = 0;
  while ((128ull << lg) <= nq) ++lg;  // floor(log2(nq / 64))
  const uint32_t max_log2 = (uint32_t)std::min(bits, std::max(bits - (int)lg, 0) + 5);
  hipLaunchKernelGGL(ptk::coherence_sample_kernel, dim3(ptk::kCoherenceWindows), dim3(64), 0, s, d_q, t->dim, nq, lo3, inv3,
                     b3, max_log2, d_fail, state);
  return state + 1;
}

int make_permutation(const ptk_tree* t, const float* d_q, uint64_t nq, hipStream_t s, Scratch& scratch,
                     uint32_t** perm, uint32_t heavy_first = 0, bool may_skip = false) {
  *perm = nullptr;
  if (nq >= (1ull << 32)) return fail(PTK_ERR_UNSUPPORTED, "batches of 2^32 or more queries are not supported");
  Timer timer(t, s);
  const int bits = morton_bits(nq);
  size_t tmp_bytes = sort_tmp_bytes(nq, bits);
  uint32_t* keys = scratch.take<uint32_t>(nq);
  uint32_t* keys_out = scratch.take<uint32_t>(nq);
  uint32_t* ids = scratch.take<uint32_t>(nq);
  uint32_t* ids_out = scratch.take<uint32_t>(nq);
  void* tmp = scratch.take<char>(tmp_bytes);
  if (!keys || !keys_out || !ids || !ids_out || !tmp) return fail(PTK_ERR_NOMEM, "scratch block too small");
  uint32_t b[3];
  axis_bits(t, bits, b);
  float lo[3] = {0, 0, 0}, inv[3] = {0, 0, 0};
  for (uint32_t d = 0; d < t->dim && d < 3; ++d) {
    lo[d] = t->root_min[d];
    const float ext = t->root_max[d] - t->root_min[d];
    inv[d] = ext > 0 ? (float)(1u << b[d]) / ext : 0.0f;
  }
  ptk::CellTable cells{};
  if (heavy_first != 0u && t->cells.occ != nullptr && env_int("PTK_HEAVY_FIRST", 1) != 0) {
    cells = t->cells;
    cells.key_bits = (uint32_t)bits;
    cells.mode = heavy_first;
  }
  // (only the library's own sort can leave early: the batch is sampled only when that sort runs)
  const uint32_t* as_given = may_skip && own_sort(nq)
                                 ? sample_batch(t, d_q, nq, s, scratch, bits, make_float3(lo[0], lo[1], lo[2]),
                                                make_float3(inv[0], inv[1], inv[2]), make_uint3(b[0], b[1], b[2]))
                                 : nullptr;
  scratch.note_verdict(as_given);
  if (own_sort(nq)) {
    // Key + histogram kernel, then per 8-bit pass: scan of the digit-by-tile histogram, stable scatter (and the
    // histogram of the next digit): 3 launches per pass - 1... nothing to clear, no look-back (ptk_sort.hpp).
    // keys -> pairs A -> [pairs B ->] permutation; the arrays of the rocprim path serve (keys | keys_out + ids = A).
    const bool blocks = block_sort(nq);
    const uint32_t tile = blocks ? ptk::kSortTile : sort_tile(nq);
    const uint32_t tiles = blocks ? (uint32_t)((nq + tile - 1) / tile) : sort_tiles(nq);
    const uint32_t stride = blocks ? (tiles + 3u) & ~3u : sort_stride(nq);
    uint32_t* hist = scratch.take<uint32_t>((size_t)ptk::kRadixBins * stride);
    uint32_t* totals = scratch.take<uint32_t>(ptk::kRadixBins);
    uint2* pairs_a = scratch.take<uint2>(nq);
    uint2* pairs_b = bits > 16 ? scratch.take<uint2>(nq) : pairs_a;
    if (!hist || !totals || !pairs_a || !pairs_b) return fail(PTK_ERR_NOMEM, "scratch block too small");
    const float3 lo3 = make_float3(lo[0], lo[1], lo[2]), inv3 = make_float3(inv[0], inv[1], inv[2]);
    const uint3 b3 = make_uint3(b[0], b[1], b[2]);
    const int passes = (bits + 7) / 8;
    const size_t smem = ptk::kRadixBins * 4;
    const uint2* in = nullptr;
    for (int p = 0; p < passes; ++p) {
      const uint32_t shift = 8u * (uint32_t)p;
      const bool first = p == 0, last = p + 1 == passes;
      uint2* out = in == pairs_a ? pairs_b : pairs_a;
      if (blocks && first)
        hipLaunchKernelGGL((ptk::radix_block_hist_kernel<true>), dim3(tiles), dim3(ptk::kSortBlock), smem, s, d_q, t->dim,
                           (uint32_t)nq, lo3, inv3, b3, keys, in, shift, stride, hist, cells, as_given);
      else if (blocks)
        hipLaunchKernelGGL((ptk::radix_block_hist_kernel<false>), dim3(tiles), dim3(ptk::kSortBlock), smem, s, d_q, t->dim,
                           (uint32_t)nq, lo3, inv3, b3, keys, in, shift, stride, hist, ptk::CellTable{}, as_given);
      else if (first)
        hipLaunchKernelGGL((ptk::radix_hist_kernel<true>), dim3(tiles), dim3(64), smem, s, d_q, t->dim, (uint32_t)nq, lo3,
                           inv3, b3, keys, in, shift, tile, stride, hist, cells, as_given);
      else
        hipLaunchKernelGGL((ptk::radix_hist_kernel<false>), dim3(tiles), dim3(64), smem, s, d_q, t->dim, (uint32_t)nq, lo3,
                           inv3, b3, keys, in, shift, tile, stride, hist, ptk::CellTable{}, as_given);
      // (the scan runs whatever the verdict: it reads `tiles` counters of every digit, whatever they hold)
      hipLaunchKernelGGL(ptk::radix_scan_kernel, dim3(ptk::kRadixBins), dim3(64), 0, s, hist, tiles, stride, totals);
#define PTK_SCATTER(F, L)                                                                                              \
  if (blocks)                                                                                                          \
    hipLaunchKernelGGL((ptk::radix_block_scatter_kernel<F, L>), dim3(tiles), dim3(ptk::kSortBlock),                    \
                       ptk::kSortScatterLds, s, keys, in, out, ids_out, (uint32_t)nq, shift, stride, hist, totals,     \
                       as_given);                                                                                      \
  else                                                                                                                 \
    hipLaunchKernelGGL((ptk::radix_scatter_kernel<F, L>), dim3(tiles), dim3(64), smem, s, keys, in, out, ids_out,      \
                       (uint32_t)nq, shift, tile, stride, hist, totals, as_given)
      if (first && last) PTK_SCATTER(true, true);
      else if (first) PTK_SCATTER(true, false);
      else if (last) PTK_SCATTER(false, true);
      else PTK_SCATTER(false, false);
#undef PTK_SCATTER
      in = out;
    }
    PTK_HIP(hipGetLastError());
    *perm = ids_out;
    scratch.note_order(1);
    timer.stop(1, 0);
    return PTK_OK;
  }
  {
    const uint32_t blocks = (uint32_t)((nq + ptk::kBlock - 1) / ptk::kBlock);
    hipLaunchKernelGGL(ptk::morton_kernel, dim3(blocks), dim3(ptk::kBlock), 0, s, d_q, t->dim, nq,
                       make_float3(lo[0], lo[1], lo[2]), make_float3(inv[0], inv[1], inv[2]),
                       make_uint3(b[0], b[1], b[2]), keys, ids, cells, (uint64_t)0);
  }
  PTK_HIP(rocprim::radix_sort_pairs<MortonSortConfig>(tmp, tmp_bytes, keys, keys_out, ids, ids_out, nq, 0, bits, s));
  *perm = ids_out;
  scratch.note_order(1);
  timer.stop(1, 0);
  return PTK_OK;
}

int check_search(const ptk_tree* t, const void* q, uint64_t nq) {
  if (t == nullptr) return fail(PTK_ERR_INVALID, "null tree");
  if (nq > 0 && q == nullptr) return fail(PTK_ERR_INVALID, "null query buffer");
  if (t->device == kDeviceNone) return fail(PTK_ERR_DEVICE, "this handle has no device replica");
  if (!t->gpu_layout) return fail(PTK_ERR_DEVICE, "this handle has no device replica");
  return PTK_OK;
}

float inv_ratio(float e) { return 1.0f / e; }

template <int S, int OVF, int BLOCK, int LEAFB, class M = ptk::MetricL2>
int launch_knn(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, uint32_t k, float e,
               ptk::Neighbor* d_out, hipStream_t s) {
  const uint32_t blocks = (uint32_t)((nq + BLOCK - 1) / BLOCK);
  const size_t stack_bytes = (size_t)S * BLOCK * 8;
  const size_t list_bytes = (size_t)k * BLOCK * 8;
  // The k-list goes to LDS while a wavefront's block stays under 48 KiB (k <= 80); beyond that the output row itself is
  // the list.  (Kernel ms on 900 k queries of config 3, list in LDS / in the row: knn = 65 47 / 71, knn = 100 138 / 138,
  // knn = 200 881 / 409 -- a list that leaves a CU two wavefronts loses to one in HBM.  PTK_KNN_LIST_LDS_KB: the limit,
  // for experiments.  Both forms are insert_sorted as a loop per lane: k beyond 64 wants a design of its own.)
  const bool list_lds = stack_bytes + list_bytes <= (size_t)std::min(156, std::max(0, env_int("PTK_KNN_LIST_LDS_KB", 48))) * 1024;
  Timer timer(t, s);
  if (list_lds) {
    const int lds_rc = allow_lds(ptk::knn_kernel<S, OVF, BLOCK, LEAFB, true, M>, stack_bytes + list_bytes);
    if (lds_rc != PTK_OK) return lds_rc;
    hipLaunchKernelGGL((ptk::knn_kernel<S, OVF, BLOCK, LEAFB, true, M>), dim3(blocks), dim3(BLOCK),
                       stack_bytes + list_bytes, s, t->dev, d_q, t->dim, perm, nq, k, inv_ratio(e), d_out);
  } else {
    hipLaunchKernelGGL((ptk::knn_kernel<S, OVF, BLOCK, LEAFB, false, M>), dim3(blocks), dim3(BLOCK), stack_bytes, s,
                       t->dev, d_q, t->dim, perm, nq, k, inv_ratio(e), d_out);
  }
  PTK_HIP(hipGetLastError());
  timer.stop(0, nq);
  return PTK_OK;
}

// Far children a query of the general k-NN kernel may enter before it is handed to the cooperative search
// (ptk_kernels_coopk.hpp; PTK_KNN_CAP, 0 = every query runs to its end in its lane).  Exact searches with the default
// metric only: the argument that makes the merged result the reference's needs e = 1 and the error bounds of a sum of
// squares.
constexpr int kKnnCoopPool = 128;
constexpr uint32_t kKnnCoopSpill = 2048;  // tasks a wavefront of the cooperative search can park in HBM
uint32_t knn_coop_blocks(const ptk_tree* t) { return (uint32_t)t->cus * (uint32_t)std::max(1, env_int("PTK_KNN_COOP_WAVES", 32)); }
// The cap follows the batch: a capped launch ends with the lanes that ran to their cap -- cap x ~7 us, a lonely lane's
// price per far child -- however small the batch, so a batch the chip gets through in less than that wants a lower cap
// and more hand-overs (at a fixed 256 ANY batch of knn = 16 took 1.9 ms).  Fitted to a sweep of twelve caps at eight
// batch sizes and four k (tools/sweep_knn_cap.sh, profiles/r05_knn_cap_sweep.jsonl; notes r05 item 13): the best cap
// is linear in the batch -- it keeps the hand-overs at 7-15 thousand, what the cooperative search gets through beside
// the capped launch's end -- with a slope that follows k (a query's far children grow with its k), steeper for the
// largest batches of k = 8 / 16 (two launches side by side there, see launch_knn_reg: the tail of the front hides
// behind the rest, so fewer hand-overs win), between a floor and a top per k:
//   k <= 4  nq / 37 500                                  8 .. 256      k <= 16  nq / 16 000      16 .. 448
//   k <= 8  max(nq / 30 000, (nq - 1.2 M) / 20 000)      12 .. 320     k <= 32  nq / 9 400       32 .. 512
// (a cap that lets more queries through than the hand-over list holds is a cliff -- those queries finish alone in
// their lanes -- so the slopes err towards the higher cap: knn = 8 at 900 k queries, caps 24 / 32: 0.94 / 0.63 ms)
// PTK_KNN_CAP = n: that cap for every batch (0: no cap).
uint32_t knn_cap(float e, uint64_t nq, uint32_t k) {
  // (below a few wavefronts of queries the two extra launches cost more than the tail: kernel ms with / without the
  // cap at 64 / 500 / 3 000 queries, knn = 16 0.13 / 0.27 / 0.30 against 0.11 / 0.66 / 0.90.  PTK_KNN_CAP_MIN_NQ: tests)
  if (e != 1.0f || nq < (uint64_t)std::max(1, env_int("PTK_KNN_CAP_MIN_NQ", 256))) return 0;
  const int forced = env_int("PTK_KNN_CAP", -1);
  if (forced >= 0) return (uint32_t)forced;
  const double n = (double)nq;
  double cap, lo, hi;
  if (k <= 4) {
    cap = n / 37500.0, lo = 8.0, hi = 256.0;
  } else if (k <= 8) {
    cap = std::max(n / 30000.0, (n - 1.2e6) / 20000.0), lo = 12.0, hi = 320.0;
  } else if (k <= 16) {
    cap = n / 16000.0, lo = 16.0, hi = 448.0;
  } else if (k <= 32) {
    cap = n / 9400.0, lo = 32.0, hi = 512.0;
  } else if (k <= 56) {
    // (33 .. 56, lists of 64 slots: the cooperative kernel of that size fits one wavefront per SIMD, so it should see
    // few queries -- kernel ms, rule / uncapped: knn = 40 at 150 k 2.0 / 5.5, 900 k 4.4 / 4.8, 7.2 M 15.9 / 15.6)
    cap = n / 3500.0, lo = 64.0, hi = 768.0;
  } else {
    // (57 .. 64: the second sweep ranks at most 64 points, k of them are the handed-over entries -- nearly every tie
    // would be redone by one lane, milliseconds each: these run uncapped)
    return 0;
  }
  return (uint32_t)std::min(hi, std::max(lo, cap));
}
// Entries of the hand-over list (64 tasks of 24 bytes each): a query that finds it full goes on in its lane.
uint64_t knn_max_handover(uint64_t nq) { return std::max<uint64_t>(nq / 48, std::min<uint64_t>(nq, 24576)); }
size_t knn_coop_scratch_bytes(const ptk_tree* t, uint64_t nq) {
  return 3 * (nq * 4) + (knn_max_handover(nq) + 24576) * ptk::kMaxTasks * sizeof(ptk::Task) + ptk::kMetaWords * 4 +
         (size_t)knn_coop_blocks(t) * kKnnCoopSpill * sizeof(ptk::Task) + 1024;
}

// The largest k whose list lives in registers (3-D kernels, every metric): 64 slots (a list of 40 in LDS took 74 ms on
// BASELINE config 3 where 32 in registers take 7: insert_sorted through LDS is a loop per lane).
inline uint32_t knn_reg_max(bool) { return 64u; }
// k <= 64: the k-list in registers (K = 4 / 8 / 16 / 32 / 64 slots compiled).
template <int S, int OVF, int BLOCK, int LEAFB, class M = ptk::MetricL2>
int launch_knn_reg(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, uint32_t k, float e,
                   ptk::Neighbor* d_out, hipStream_t s, Scratch* scratch = nullptr) {
  const uint32_t blocks = (uint32_t)((nq + BLOCK - 1) / BLOCK);
  const size_t smem = (size_t)S * BLOCK * 8;
  Timer timer(t, s);
  if constexpr (std::is_same<M, ptk::MetricL2>::value && BLOCK == 64) {
    const uint32_t cap = scratch != nullptr ? knn_cap(e, nq, k) : 0u;
    if (cap != 0u) {
      // The capped launch, the cooperative search of what it handed over, the reference search of what that could
      // not certify: the counts stay on the device.  A batch of four million queries or more goes through as TWO capped
      // launches side by side -- the front of the launch order (the expensive rows: `perm` puts them first) on a second
      // stream, the rest on the caller's -- so that the cooperative search of what the front handed over runs BESIDE the
      // rest instead of behind it (PTK_KNN_OVERLAP_PCT: the front's share of the rows, 0 = one launch).
      const uint32_t coop_blocks = knn_coop_blocks(t);
      uint64_t n_front = 0;
      hipStream_t side = nullptr;
      hipEvent_t fork = nullptr, join = nullptr;
      // (kernel ms, two launches / one: 7.2 M queries knn = 4 / 8 / 16 / 32 2.15 / 2.67 / 3.78 / 7.07 against 2.20 / 2.71 /
      // 3.86 / 7.11, 4.8 M 1.58 / 2.01 / 2.75 / 5.00 against 1.59 / 1.95 / 2.86 / 5.23; at 2.4 M and below, and for
      // knn = 2, the second launch costs more than the overlap returns: 1.82 against 1.67 at knn = 16)
      if (perm != nullptr && nq >= (1ull << 22) && k > 2) {
        const uint64_t pct = (uint64_t)std::min(90, std::max(0, env_int("PTK_KNN_OVERLAP_PCT", 20)));
        n_front = (nq * pct / 100) / BLOCK * BLOCK;
        if (n_front != 0 && !scratch->side_stream(&side, &fork, &join)) n_front = 0;
      }
      uint32_t* meta = scratch->take<uint32_t>(ptk::kMetaWords);
      uint32_t* heavy_list = scratch->take<uint32_t>(nq);
      uint32_t* ntasks = scratch->take<uint32_t>(nq);
      const uint32_t cap_front = (uint32_t)knn_max_handover(n_front), cap_rest = (uint32_t)knn_max_handover(nq - n_front);
      ptk::Task* tasks = scratch->take<ptk::Task>(((size_t)(n_front ? cap_front : 0) + cap_rest) * ptk::kMaxTasks);
      uint32_t* redo_list = scratch->take<uint32_t>(nq);
      ptk::Task* spill = scratch->take<ptk::Task>((size_t)coop_blocks * kKnnCoopSpill);
      if (!meta || !heavy_list || !ntasks || !tasks || !redo_list || !spill)
        return fail(PTK_ERR_NOMEM, "scratch block too small");
      scratch->note_meta(meta);
      PTK_HIP(hipMemsetAsync(meta, 0, ptk::kMetaWords * 4, s));
      const size_t coop_smem = (size_t)ptk::knn_coop_lds_words(kKnnCoopPool, k > 32 ? 64u : 32u) * 4;
      const uint2* ranges = static_cast<const uint2*>(t->d_ranges);
      // One capped launch over launch-order rows [lo, lo + n) and the cooperative search of its hand-overs, on `st`
      // (`word`: the counter of its list; `cb` wavefronts from `first_block` of the spill block).
      auto part = [&](uint64_t lo, uint64_t n, uint32_t word, uint32_t max_heavy, ptk::Task* part_tasks, uint32_t cb,
                      uint32_t first_block, hipStream_t st) {
        ptk::Handover ho{};
        ho.counter = word;
        ho.meta = meta;
        ho.heavy_list = heavy_list + lo;
        ho.ntasks = ntasks + lo;
        ho.max_heavy = max_heavy;
        ho.full_keeps = 1u;
        ho.tasks = part_tasks;
        const uint32_t nb = (uint32_t)((n + BLOCK - 1) / BLOCK);
        const uint32_t cap_n = cap;  // (of the whole batch: the two launches share the chip)
        ptk::Task* sp = spill + (size_t)first_block * kKnnCoopSpill;
#define PTK_LAUNCH_REG(KK)                                                                                              \
  do {                                                                                                                  \
    hipLaunchKernelGGL((ptk::knn_reg_kernel<KK, S, OVF, BLOCK, LEAFB, M, true>), dim3(nb), dim3(BLOCK), smem, st,        \
                       t->dev, d_q, t->dim, perm ? perm + lo : nullptr, n, k, inv_ratio(e), d_out, cap_n, ho);          \
    hipLaunchKernelGGL((ptk::knn_coop_kernel<KK, kKnnCoopPool>), dim3(cb), dim3(64), coop_smem, st, t->dev, ranges,      \
                       d_q, t->dim, k, d_out, ho, redo_list, ptk::kMetaRedo, sp, kKnnCoopSpill);                        \
  } while (0)
        if (k <= 4) PTK_LAUNCH_REG(4);
        else if (k <= 8) PTK_LAUNCH_REG(8);
        else if (k <= 16) PTK_LAUNCH_REG(16);
        else if (k <= 32) PTK_LAUNCH_REG(32);
        else PTK_LAUNCH_REG(64);
#undef PTK_LAUNCH_REG
      };
      if (n_front != 0) {
        // (a failure between fork and join must not leave the second stream working on a scratch block the next call reuses)
        struct SideGuard {
          hipStream_t side = nullptr;
          ~SideGuard() {
            if (side) (void)hipStreamSynchronize(side);
          }
        } guard;
        guard.side = side;
        PTK_HIP(hipEventRecord(fork, s));
        PTK_HIP(hipStreamWaitEvent(side, fork, 0));
        const uint32_t half = std::max(1u, coop_blocks / 2);
        part(0, n_front, ptk::kMetaHeavy, cap_front, tasks, half, 0u, side);
        PTK_HIP(hipEventRecord(join, side));
        part(n_front, nq - n_front, ptk::kMetaHeavyRest, cap_rest, tasks + (size_t)cap_front * ptk::kMaxTasks,
             coop_blocks - half, half, s);
        PTK_HIP(hipStreamWaitEvent(s, join, 0));
        guard.side = nullptr;
      } else {
        part(0, nq, ptk::kMetaHeavy, cap_rest, tasks, coop_blocks, 0u, s);
      }
#define PTK_LAUNCH_REDO(KK)                                                                                             \
  hipLaunchKernelGGL((ptk::knn_redo_kernel<KK, S, OVF, LEAFB, M>), dim3(t->cus), dim3(64), smem, s, t->dev, d_q, t->dim, \
                     k, inv_ratio(e), d_out, meta, ptk::kMetaRedo, redo_list)
      if (k <= 4) PTK_LAUNCH_REDO(4);
      else if (k <= 8) PTK_LAUNCH_REDO(8);
      else if (k <= 16) PTK_LAUNCH_REDO(16);
      else if (k <= 32) PTK_LAUNCH_REDO(32);
      else PTK_LAUNCH_REDO(64);
#undef PTK_LAUNCH_REDO
      PTK_HIP(hipGetLastError());
      timer.stop(0, nq);
      return PTK_OK;
    }
  }
#define PTK_LAUNCH_REG(KK)                                                                                          \
  hipLaunchKernelGGL((ptk::knn_reg_kernel<KK, S, OVF, BLOCK, LEAFB, M>), dim3(blocks), dim3(BLOCK), smem, s, t->dev, d_q, \
                     t->dim, perm, nq, k, inv_ratio(e), d_out, 0u, ptk::Handover{})
  if (k <= 4) PTK_LAUNCH_REG(4);
  else if (k <= 8) PTK_LAUNCH_REG(8);
  else if (k <= 16) PTK_LAUNCH_REG(16);
  else if (k <= 32) PTK_LAUNCH_REG(32);
  else PTK_LAUNCH_REG(64);  // (33 .. 64: knn_reg_max)
#undef PTK_LAUNCH_REG
  PTK_HIP(hipGetLastError());
  timer.stop(0, nq);
  return PTK_OK;
}

template <int S, int OVF, int BLOCK, int LEAFB, class M = ptk::MetricL2>
int launch_radius(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float radius, float e,
                  bool fill, uint64_t* d_counts, const uint64_t* d_offsets, ptk::Neighbor* d_out,
                  hipStream_t s, const uint32_t* n_dev = nullptr) {
  const uint32_t blocks = (uint32_t)((nq + BLOCK - 1) / BLOCK);
  const size_t smem = (size_t)S * BLOCK * 8;
  Timer timer(t, s);
  if (!fill) {
    hipLaunchKernelGGL((ptk::radius_kernel<S, OVF, BLOCK, LEAFB, false, M>), dim3(blocks), dim3(BLOCK), smem, s,
                       t->dev, d_q, t->dim, perm, nq, radius, inv_ratio(e), d_counts, d_offsets, d_out, n_dev);
  } else {
    hipLaunchKernelGGL((ptk::radius_kernel<S, OVF, BLOCK, LEAFB, true, M>), dim3(blocks), dim3(BLOCK), smem, s,
                       t->dev, d_q, t->dim, perm, nq, radius, inv_ratio(e), d_counts, d_offsets, d_out, n_dev);
  }
  PTK_HIP(hipGetLastError());
  timer.stop(0, n_dev ? 0 : nq);
  return PTK_OK;
}

// The count pass that also captures the rows.
template <int S, int OVF, int BLOCK, int LEAFB, class M = ptk::MetricL2>
int launch_radius_capture(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float radius,
                          float e, uint64_t* d_counts, const ptk::RadiusCapture& cap, hipStream_t s) {
  const uint32_t blocks = (uint32_t)((nq + BLOCK - 1) / BLOCK);
  const size_t smem = (size_t)S * BLOCK * 8 + 16;  // + the cursor of the wavefront's log
  Timer timer(t, s);
  PTK_HIP(hipMemsetAsync(cap.counters, 0, ptk::kCapSubPools * ptk::kCapCounterStride * 4, s));
  hipLaunchKernelGGL((ptk::radius_capture_kernel<S, OVF, BLOCK, LEAFB, M>), dim3(blocks), dim3(BLOCK), smem, s,
                     t->dev, d_q, t->dim, perm, nq, radius, inv_ratio(e), d_counts, cap);
  PTK_HIP(hipGetLastError());
  timer.stop(0, nq);
  return PTK_OK;
}

// The radius search of a 3-D tree with the rows made from leaf lists (ptk_kernels_lists.hpp): the count pass ...
template <int S, int OVF, int LEAFB, class M = ptk::MetricL2>
int launch_radius_list(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float radius, float e,
                       uint64_t* d_counts, const ptk::RadiusCapture& cap, hipStream_t s) {
  const size_t smem = (size_t)S * 64 * 8 + ptk::kListLds;  // + the group buffers and the chunk table of the wavefront
  Timer timer(t, s);
  PTK_HIP(hipMemsetAsync(cap.counters, 0, ptk::kCapSubPools * ptk::kCapCounterStride * 4, s));
  // (leaves of more than kListMaskBits points -- a count of 32 needs six bits -- / an approximate search: see RadiusListPolicy)
  const bool big = t->dev.cmask >= ptk::kListMaskBits, exact = e == 1.0f;
#define PTK_LAUNCH_LIST(BIG, EXACT)                                                                                  \
  hipLaunchKernelGGL((ptk::radius_list_kernel<S, OVF, LEAFB, M, BIG, EXACT>), dim3(cap.n_static), dim3(64), smem, s, \
                     t->dev, d_q, t->dim, perm, nq, radius, inv_ratio(e), d_counts, cap)
  if (big && exact) PTK_LAUNCH_LIST(true, true);
  else if (big) PTK_LAUNCH_LIST(true, false);
  else if (exact) PTK_LAUNCH_LIST(false, true);
  else PTK_LAUNCH_LIST(false, false);
#undef PTK_LAUNCH_LIST
  PTK_HIP(hipGetLastError());
  timer.stop(0, nq);
  return PTK_OK;
}

// ... and the fill pass.  n_over is zeroed here; the queries of wavefronts whose lists were lost are listed for
// radius_kernel<FILL>.
// Hits fetched together / entries a lane can hold back: (5, 16) 4.72 ms, (4, 16) 4.73, (5, 32) 4.49, (8, 32) 4.38 on
// BASELINE config 3 -- fewer, larger rounds win although the ring of 32 halves the wavefronts per CU.
constexpr int kReplayHits = 8, kReplayRing = 32;
template <class M = ptk::MetricL2>
int launch_radius_replay(const ptk_tree* t, const float* d_q, float e, const ptk::RadiusCapture& cap,
                         const uint64_t* d_offsets, ptk::Neighbor* d_out, uint32_t* over_list, uint32_t* n_over,
                         hipStream_t s) {
  Timer timer(t, s);
  PTK_HIP(hipMemsetAsync(n_over, 0, 4, s));
  hipLaunchKernelGGL((ptk::radius_replay_kernel<kReplayHits, kReplayRing, M>), dim3(cap.n_static), dim3(64),
                     ptk::replay_lds(kReplayRing), s, t->dev, d_q, t->dim, inv_ratio(e), cap, d_offsets, d_out, over_list,
                     n_over);
  PTK_HIP(hipGetLastError());
  timer.stop(0, 0);
  return PTK_OK;
}

// PTK_RADIUS_CAPTURE_MB: the most device memory the captured rows of a radius batch may take
// (default 16384; 0 switches the capture off and every fill pass repeats the traversal).
size_t capture_budget_bytes(const ptk_tree* t) {
  // default: 16 GiB, but no more than a quarter of the device's memory (a partitioned or smaller device)
  const size_t quarter_mb = t->hbm_bytes ? (t->hbm_bytes >> 22) : 16384;
  const int mb = env_int("PTK_RADIUS_CAPTURE_MB", (int)std::min<size_t>(16384, std::max<size_t>(quarter_mb, 64)));
  return mb <= 0 ? 0 : (size_t)mb << 20;
}

// Sizes (and if needed allocates) the capture block for a batch of nq rows; false = no capture.
// Layout: counters | captured flags (one per wavefront) | query of every lane | chunks.  Every wavefront of the
// launch owns one static chunk; the dynamic pool is sized for 1024 hits per row when the budget allows (the
// scan-like cloud of BASELINE config 3 averages 105); PTK_RADIUS_CAPTURE_CHUNKS overrides chunks per sub-pool (tests).
bool prepare_capture(const ptk_tree* t, uint64_t nq, Workspace& ws) {
  const size_t budget = capture_budget_bytes(t);
  if (budget == 0 || nq == 0 || nq >= (1ull << 31)) return false;
  const size_t waves = (size_t)((nq + 63) / 64);
  const size_t chunk_bytes = (size_t)ptk::kLogChunk * sizeof(ptk::Neighbor);
  const size_t flags_at = (size_t)ptk::kCapSubPools * ptk::kCapCounterStride * 4;
  const size_t qids_at = flags_at + ((waves + 255) & ~(size_t)255);
  const size_t lens_at = qids_at + waves * 64 * 4;
  const size_t tables_at = lens_at + waves * 64 * 4;
  const size_t head = (tables_at + waves * ptk::kListMaxChunks * 4 + 4095) & ~(size_t)4095;
  if (head + waves * chunk_bytes > budget) return false;
  const size_t dyn = std::min<size_t>((budget - head - waves * chunk_bytes) / chunk_bytes, waves * 64 * 2);
  const int forced = env_int("PTK_RADIUS_CAPTURE_CHUNKS", -1);
  size_t sub_cap = forced >= 0 ? (size_t)forced : dyn / ptk::kCapSubPools;
  // (a slot of the log is addressed with 32 bits)
  const size_t max_chunks = ((1ull << 32) - 1) / ptk::kLogChunk;
  if (waves >= max_chunks) return false;
  if (waves + sub_cap * ptk::kCapSubPools > max_chunks) sub_cap = (max_chunks - waves) / ptk::kCapSubPools;
  const size_t bytes = head + (waves + sub_cap * ptk::kCapSubPools) * chunk_bytes;
  if (bytes > ws.cap_capacity) {
    drain_workspace(ws);
    if (ws.cap_base) (void)hipFree(ws.cap_base);
    ws.cap_base = nullptr;
    ws.cap_capacity = 0;
    ws.cap_valid = false;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes > free_b / 2) return false;
    if (hipMalloc((void**)&ws.cap_base, bytes) != hipSuccess) {
      (void)hipGetLastError();
      ws.cap_base = nullptr;
      return false;
    }
    ws.cap_capacity = bytes;
  }
  ws.cap.counters = reinterpret_cast<uint32_t*>(ws.cap_base);
  ws.cap.captured = reinterpret_cast<uint8_t*>(ws.cap_base + flags_at);
  ws.cap.qids = reinterpret_cast<uint32_t*>(ws.cap_base + qids_at);
  ws.cap.lens = reinterpret_cast<uint32_t*>(ws.cap_base + lens_at);
  ws.cap.tables = reinterpret_cast<uint32_t*>(ws.cap_base + tables_at);
  ws.cap.chunks = reinterpret_cast<ptk::Neighbor*>(ws.cap_base + head);
  ws.cap.n_static = (uint32_t)waves;
  ws.cap.sub_cap = (uint32_t)sub_cap;
  return true;
}

// Narrow tiers of phase 2 (cumulative per-mille marks of the ranked classes, lanes per wavefront).  A search that
// runs every query to its end (e != 1: no cap, see phase2_cap) starts the 6 % most expensive continuations four to
// a wavefront: they are the critical path of the launch (profiles/r01e_notes.txt).  With the cap the long chains go
// to the cooperative search and there is no narrow tier.
ptk::TierSpec phase2_tiers(uint32_t cap) {
  ptk::TierSpec t{};
  if (cap == 0) {
    t.permille[0] = 60;
    t.lanes[0] = 4;
  }
  return t;
}

size_t class_sort_tmp_bytes(uint64_t nq) {
  size_t tmp_bytes = 0;
  ptk::ContKey* k16 = nullptr;
  uint32_t* v32 = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, tmp_bytes, k16, k16, v32, v32, nq, 0, 16, (hipStream_t) nullptr);
  return tmp_bytes + 256;
}

// Queries whose unfinished stack phase 2 can hand to the cooperative search (kMaxTasks records
// each); further ones are searched again from the root.
uint64_t max_handover(uint64_t nq) { return std::max<uint64_t>(nq / 32, std::min<uint64_t>(nq, 16384)); }

// The class order as a counting sort (ptk_kernels.hpp, class_scan_kernel / class_order_kernel): phase 1 counts its
// own tile of 64 slots; the rows of counters are scanned in at most kClassMaxSegs segments of a multiple of 1024
// tiles; the scatter takes chunks of `per` slots, one wavefront each (short chains for a small batch).
struct ClassPlan {
  uint32_t ntiles, stride, seg, segs, per, chunks;
};
ClassPlan class_plan(uint64_t nq) {
  ClassPlan p;
  p.ntiles = (uint32_t)((nq + 63) / 64);
  p.stride = (p.ntiles + 3u) & ~3u;
  p.seg = 1024u * std::max<uint32_t>(1u, (p.ntiles + 1024u * ptk::kClassMaxSegs - 1u) / (1024u * ptk::kClassMaxSegs));
  p.segs = (p.ntiles + p.seg - 1u) / p.seg;
  p.per = (uint32_t)std::max(64, env_int("PTK_CLASS_PER", nq < (2ull << 20) ? 512 : 1792)) & ~63u;
  p.chunks = (uint32_t)((nq + p.per - 1) / p.per);
  return p;
}

// The cooperative search: 16 lanes per query, a pool of 96 subtrees per group (8 / 32 / 64 lanes were measured:
// 1.91 / 1.90 / slower vs 1.72 ms of traversal kernels, profiles/r02_notes.txt items 6, 12).
constexpr int kCoopLanes = 16, kCoopPool = 96;
// Tasks a group of the cooperative search may park in HBM when its LDS pool is full.
constexpr uint32_t kCoopSpill = 256;
inline int coop_waves(const ptk_tree* t) {
  constexpr size_t smem = (size_t)(64 / kCoopLanes) * (6 * kCoopPool + 2) * 4;
  // As many waves as can be resident at once (LDS-bound: CUs x LDS per CU of the device), each group working
  // through its share of the list: a second round of blocks would start when most of the work is done.
  return t->cus * (int)std::max<size_t>(1, std::min<size_t>(24, t->lds_per_cu / (smem + 512)));
}
size_t coop_spill_bytes(const ptk_tree* t) {
  return (size_t)coop_waves(t) * (64 / kCoopLanes) * kCoopSpill * sizeof(ptk::Task);
}

size_t two_phase_scratch_bytes(const ptk_tree* t, uint64_t nq) {
  return nq * sizeof(float4) + nq * ptk::kContSlots * sizeof(ptk::Record) + nq * sizeof(uint4) + 4 * nq +
         2 * (nq * 4) + 3 * (nq * 4) + max_handover(nq) * ptk::kMaxTasks * sizeof(ptk::Task) + 64 + 2 * coop_spill_bytes(t) +
         class_sort_tmp_bytes(nq) + (size_t)ptk::kClassBuckets * (class_plan(nq).stride + ptk::kClassMaxSegs) * 4 + 2048;
}

// Far children a query may enter in phase 2 before it is handed to the cooperative search
// (PTK_P2_CAP; 0 = phase 2 runs every query to its end).  Exact searches only: the argument that
// makes the cooperative result the reference's (ptk_kernels.hpp, knn1_coop_kernel) needs e = 1.
// The cap is a chain length (cap x ~10 us of dependent rounds in the dealt tier): what a big batch hides behind
// its light tier is exposed on a small one, e.g. one shard of BASELINE configs[3] -- 900 k queries: 0.556 ms per
// step with a cap of 8, 0.604 with 16, 0.612 with 4 (profiles/r02_notes.txt items 4, 17).  With the ranked classes
// taken out of phase 2 (coop_direct_mode): 900 k queries 6 / 8 / 12 / 16 = 0.247 / 0.245 / 0.265 / 0.285 ms of traversal
// kernels, 7.2 M queries 8 / 12 / 16 / 24 / 32 = 1.287 / 1.268 / 1.255 / 1.238 / 1.271 (profiles/r03_notes.txt item 3).
uint32_t phase2_cap(float e, uint64_t nq) {
  if (e != 1.0f) return 0;
  const int cap = env_int("PTK_P2_CAP", nq >= (4ull << 20) ? 24 : 8);
  return cap < 0 ? 0u : (uint32_t)cap;
}

// direct_ids == nullptr: the list is `ho`'s.  Otherwise it is the ranked head of the class-sorted entries, searched
// straight from the continuation records of phase 1 (knn1_coop_kernel<.., DIRECT>), `lanes` lanes per query.
template <int G>
int launch_knn1_coop_direct(const ptk_tree* t, const float4* qs, ptk::Neighbor* d_out, const ptk::Cont& cont,
                            const ptk::Handover& ho, uint32_t* redo_list, hipStream_t s, ptk::Task* spill,
                            const uint32_t* direct_ids) {
  constexpr size_t smem = (size_t)(64 / G) * (6 * kCoopPool + 2) * 4;
  // (the spill block is sized for coop_waves(t) x 64 / kCoopLanes groups: a wider group count would not fit)
  static_assert(G >= kCoopLanes, "the spill block is sized for groups of kCoopLanes lanes");
  const int resident = t->cus * (int)std::max<size_t>(1, std::min<size_t>(32, t->lds_per_cu / (smem + 512)));
  const int waves = std::min(resident, coop_waves(t) * (G / kCoopLanes));
  hipLaunchKernelGGL((ptk::knn1_coop_kernel<G, kCoopPool, true>), dim3(waves), dim3(64), smem, s, knn1_tree(t),
                     knn1_ranges(t), qs, d_out, cont, ho, redo_list, direct_ids, spill, kCoopSpill);
  PTK_HIP(hipGetLastError());
  return PTK_OK;
}

int launch_knn1_coop(const ptk_tree* t, const float4* qs, ptk::Neighbor* d_out, const ptk::Cont& cont,
                     const ptk::Handover& ho, uint32_t* redo_list, hipStream_t s, ptk::Task* spill,
                     const uint32_t* direct_ids = nullptr) {
  constexpr size_t smem = (size_t)(64 / kCoopLanes) * (6 * kCoopPool + 2) * 4;
  const int waves = coop_waves(t);
  const uint32_t spill_cap = spill ? kCoopSpill : 0u;
  if (direct_ids != nullptr) {
    switch (env_int("PTK_COOP_DIRECT_LANES", 32)) {
      case 32: return launch_knn1_coop_direct<32>(t, qs, d_out, cont, ho, redo_list, s, spill, direct_ids);
      case 64: return launch_knn1_coop_direct<64>(t, qs, d_out, cont, ho, redo_list, s, spill, direct_ids);
      default: return launch_knn1_coop_direct<16>(t, qs, d_out, cont, ho, redo_list, s, spill, direct_ids);
    }
  }
  // What phase 2 hands over has been tightened by its first far children: a pool of 96 holds it (0 of 152 k queries of
  // BASELINE config 2 overflow; the spill costs the step loop 5 %).  PTK_COOP_SPILL=1: with the spill all the same
  // (tie-prone data whose replays would be long chains).
  if (spill_cap != 0u && env_int("PTK_COOP_SPILL", 0) != 0)
    hipLaunchKernelGGL((ptk::knn1_coop_kernel<kCoopLanes, kCoopPool, false, true>), dim3(waves), dim3(64), smem, s, knn1_tree(t),
                       knn1_ranges(t), qs, d_out, cont, ho, redo_list, nullptr, spill, spill_cap);
  else
    hipLaunchKernelGGL((ptk::knn1_coop_kernel<kCoopLanes, kCoopPool, false, false>), dim3(waves), dim3(64), smem, s, knn1_tree(t),
                       knn1_ranges(t), qs, d_out, cont, ho, redo_list, nullptr, spill, 0u);
  PTK_HIP(hipGetLastError());
  return PTK_OK;
}

// How the ranked classes of an exact k = 1 batch -- 2 % of the queries, every expensive one among them -- are
// searched (PTK_COOP_DIRECT overrides):
//   0  by phase 2 up to the cap like everything else, what is left cooperatively afterwards
//   1  cooperatively straight from phase 1, on the same stream, before phase 2
//   2  the same on a second stream BESIDE phase 2: the capped traversal of such a query and its cooperative search
//      afterwards are two chains of dependent rounds; taken straight from phase 1 it is searched once, while phase 2
//      works on the rest.  Scan-like cloud (profiles/r03_notes.txt item 3): 900 k queries 0.348 -> 0.245 ms of
//      traversal kernels, 7.2 M queries 1.35 -> 1.23 ms.  A uniform cloud has no expensive queries: there the
//      cooperative search of the ranked classes is merely the dearer way (7.2 M queries 1.31 -> 1.35 ms; 900 k queries
//      0.255 -> 0.242 ms all the same, the GPU being mostly idle).
// The sign of a cloud with expensive queries is a tree much deeper than a balanced one (the sliding midpoint peels
// dense regions level by level: scan-like cloud 33 levels for 2^20 leaves, uniform cloud 23).
int coop_direct_mode(const ptk_tree* t, uint64_t nq) {
  uint32_t balanced = 0;
  while ((1ull << balanced) < t->n_leaves) ++balanced;
  const bool deep = t->max_depth >= balanced + 6u;
  return env_int("PTK_COOP_DIRECT", deep || nq < (1ull << 20) ? 2 : 0);
}

// The k = 1 search under the default metric (ptk_kernels.hpp, "the two-phase k = 1 search"): phase 1 (which also
// packs the launch-order records), the class order of the continuations, phase 2, and for exact searches the
// cooperative search of what phase 2 handed over plus the replay of what that could not certify.
template <int OVF>
int launch_knn1_two_phase(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float e,
                          ptk::Neighbor* d_out, hipStream_t s, Scratch& scratch) {
  constexpr int LEAFB = 4;  // points fetched per round trip
  if (nq >= (1ull << 32)) return fail(PTK_ERR_UNSUPPORTED, "batches of 2^32 or more queries are not supported");
  float4* qs = scratch.take<float4>(nq);
  if (qs == nullptr) return fail(PTK_ERR_NOMEM, "scratch block too small");
  ptk::Cont cont{};
  cont.nq = nq;
  size_t tmp_bytes = class_sort_tmp_bytes(nq);
  cont.rec = scratch.take<ptk::Record>(nq * ptk::kContSlots);
  cont.best = scratch.take<uint4>(nq);
  cont.key = scratch.take<ptk::ContKey>(nq);
  ptk::ContKey* key_out = scratch.take<ptk::ContKey>(nq);
  cont.ids = scratch.take<uint32_t>(nq);
  uint32_t* ids_out = scratch.take<uint32_t>(nq);
  cont.meta = scratch.take<uint32_t>(ptk::kMetaWords);
  void* tmp = scratch.take<char>(tmp_bytes);
  if (!cont.rec || !cont.best || !cont.key || !key_out || !cont.ids || !ids_out || !cont.meta || !tmp)
    return fail(PTK_ERR_NOMEM, "scratch block too small");
  ptk::Handover ho{};
  ho.counter = ptk::kMetaHeavy;
  ho.meta = cont.meta;
  ho.heavy_list = scratch.take<uint32_t>(nq);
  ho.ntasks = scratch.take<uint32_t>(nq);
  ho.max_heavy = (uint32_t)max_handover(nq);
  ho.tasks = scratch.take<ptk::Task>((size_t)ho.max_heavy * ptk::kMaxTasks);
  uint32_t* redo_list = scratch.take<uint32_t>(nq);
  if (!ho.heavy_list || !ho.ntasks || !ho.tasks || !redo_list) return fail(PTK_ERR_NOMEM, "scratch block too small");
  // Where the groups of the cooperative search park subtrees their LDS pool has no room for (one block per launch
  // that may be in flight: the direct one on the second stream, the one behind phase 2).
  ptk::Task* spill_a = reinterpret_cast<ptk::Task*>(scratch.take<char>(coop_spill_bytes(t)));
  ptk::Task* spill_b = reinterpret_cast<ptk::Task*>(scratch.take<char>(coop_spill_bytes(t)));
  if (!spill_a || !spill_b) return fail(PTK_ERR_NOMEM, "scratch block too small");
  scratch.note_meta(cont.meta);
  const uint32_t blocks = (uint32_t)((nq + 63) / 64);
  const float e_inv = inv_ratio(e);
  const uint32_t cap = phase2_cap(e, nq);
  uint32_t* const slot_ids = cont.ids;
  if (cap) cont.ids = nullptr;  // phase 1 need not write slot numbers: the counting sort produces them
  // The grid has room for nq / 64 extra waves in the narrow tiers; the meta kernel cuts the tiers to what fits.
  const ptk::TierSpec tiers = phase2_tiers(cap);
  const uint32_t extra_waves = tiers.permille[0] == 0 ? 0u : (uint32_t)(nq / 64) + 2u;
  // The ranked classes straight to the cooperative search (exact searches only), beside phase 2 if a second stream
  // can be had.
  int direct = cap ? coop_direct_mode(t, nq) : 0;
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  if (direct == 2 && !scratch.side_stream(&side, &fork, &join)) direct = 1;
  // Exact searches: phase 1 counts the class buckets of its tile for the counting sort below.
  const ClassPlan cp = class_plan(nq);
  uint32_t* tile_counts = nullptr;
  uint32_t* seg_totals = nullptr;
  if (cap) {
    tile_counts = scratch.take<uint32_t>((size_t)ptk::kClassBuckets * cp.stride);
    seg_totals = scratch.take<uint32_t>((size_t)ptk::kClassBuckets * ptk::kClassMaxSegs);
    if (!tile_counts || !seg_totals) return fail(PTK_ERR_NOMEM, "scratch block too small");
  }
  // One chain of sections: search (phase 1) | other (class order) | search (phase 2, cooperative search, replay).
  Timer timer(t, s);
  hipLaunchKernelGGL((ptk::knn1_phase1u_kernel<LEAFB>), dim3(blocks), dim3(64), 0, s, knn1_tree(t), d_q, t->dim, perm, nq,
                     e_inv, d_out, cont, qs, tile_counts, cp.stride, scratch.batch_verdict());
  timer.next(0, nq);
  if (cap) {
    // With the cap the order inside the heavy classes does not matter (no query runs long), only the three class
    // bits do: 8 buckets, counted per tile by phase 1 -- scan of the counters, stable scatter, which also writes
    // the tier table (ptk_kernels.hpp, "the class order from the tile counts of phase 1").
    hipLaunchKernelGGL(ptk::class_scan_kernel, dim3(ptk::kClassBuckets * cp.segs), dim3(64), 0, s, tile_counts, cp.ntiles,
                       cp.stride, cp.seg, seg_totals);
    hipLaunchKernelGGL(ptk::class_order_kernel, dim3(cp.chunks), dim3(64), 0, s, cont.key, (uint32_t)nq, cp.per,
                       tile_counts, cp.stride, cp.seg, cp.segs, seg_totals, ids_out, cont, tiers, extra_waves,
                       direct ? 1u : 0u);
    PTK_HIP(hipGetLastError());
  } else {
    // Every query runs to its end in phase 2: the full 16-bit key (the ranked classes by how far their
    // home-leaf best is), so that the most expensive continuations start first.
    PTK_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, cont.key, key_out, slot_ids, ids_out, nq, 0, 16, s));
    hipLaunchKernelGGL(ptk::knn1_phase_meta_kernel, dim3(1), dim3(1), 0, s, key_out, (uint32_t)nq, cont, tiers,
                       extra_waves);
  }
  timer.next(2, 0);
  // (a failure between fork and join must not leave the second stream working on a scratch block the next call reuses)
  struct SideGuard {
    hipStream_t side = nullptr;
    ~SideGuard() {
      if (side) (void)hipStreamSynchronize(side);
    }
  } side_guard;
  if (direct == 2) {  // fork: the ranked classes, cooperatively, while phase 2 takes the rest
    side_guard.side = side;
    PTK_HIP(hipEventRecord(fork, s));
    PTK_HIP(hipStreamWaitEvent(side, fork, 0));
    int rc = launch_knn1_coop(t, qs, d_out, cont, ho, redo_list, side, spill_a, ids_out);
    if (rc != PTK_OK) return rc;
    PTK_HIP(hipEventRecord(join, side));
  } else if (direct == 1) {
    int rc = launch_knn1_coop(t, qs, d_out, cont, ho, redo_list, s, spill_a, ids_out);
    if (rc != PTK_OK) return rc;
  }
  const dim3 p2_grid(blocks + 1 + extra_waves);
  // LDS ring of phase 2.  With the cap no stack grows deep: 12 slots = 6 KB per wave = 26 waves per CU beat
  // 16 (20 waves) and 8 (40 waves) on both clouds (profiles/r02_notes.txt items 10, 23); without it 16 slots
  // (r01l_notes item 8).
  if (cap) {
    hipLaunchKernelGGL((ptk::knn1_phase2_kernel<kP2Ring, OVF, LEAFB>), p2_grid, dim3(64), (size_t)kP2Ring * 64 * 8, s, knn1_tree(t), qs,
                       e_inv, d_out, cont, ids_out, cap, ho);
  } else {
    hipLaunchKernelGGL((ptk::knn1_phase2_kernel<16, OVF, LEAFB>), p2_grid, dim3(64), (size_t)16 * 64 * 8, s, knn1_tree(t), qs,
                       e_inv, d_out, cont, ids_out, 0u, ho);
  }
  PTK_HIP(hipGetLastError());
  if (cap) {  // the queries phase 2 gave up on, then whatever the cooperative search could not certify
    // (the two cooperative launches may run side by side: each has its own spill block, and they append to the one
    // redo list through one atomic counter)
    int rc = launch_knn1_coop(t, qs, d_out, cont, ho, redo_list, s, spill_b);
    if (rc != PTK_OK) return rc;
    if (direct == 2) {  // join: the replay needs both lists complete
      PTK_HIP(hipStreamWaitEvent(s, join, 0));
      side_guard.side = nullptr;
    }
    hipLaunchKernelGGL((ptk::knn1_redo_kernel<16, OVF, LEAFB>), dim3(t->cus), dim3(64), (size_t)16 * 64 * 8, s, knn1_tree(t), qs,
                       e_inv, d_out, cont, redo_list);
    PTK_HIP(hipGetLastError());
  }
  if (t->n_piles) {  // rows that name the stand-in of a pile get the point of it the reference reports (ptk_piles.hpp)
    ptk::DevPiles piles;
    piles.of_point = static_cast<const uint32_t*>(t->d_pile_of_point);
    piles.recs = static_cast<const ptk::DevPileRecord*>(t->d_pile_recs);
    piles.n_points = (uint32_t)t->n_points;
    hipLaunchKernelGGL(ptk::resolve_piles_kernel, dim3((uint32_t)((nq + ptk::kBlock - 1) / ptk::kBlock)), dim3(ptk::kBlock), 0, s,
                       d_q, t->dim, nq, piles, d_out);
    PTK_HIP(hipGetLastError());
  }
  timer.stop(3, 0);
  return PTK_OK;
}

// ---- any dimension (dim > 3) -----------------------------------------------------------------
// LDS per 64-lane block: record ring + q[dim] + off[dim] (+ the k-list while it fits).
// (the most dynamic LDS a block may ask for is the handle's lds_per_block, from the device's properties)

template <int OVF, class M = ptk::MetricL2>
int launch_knn_nd(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, uint32_t k, float e,
                  ptk::Neighbor* d_out, hipStream_t s, bool no_register_list = false) {
  constexpr int S = 16;
  const uint32_t blocks = (uint32_t)((nq + 63) / 64);
  const size_t base = (size_t)S * 64 * 8 + (size_t)t->dim * 64 * 8;
  if (base > t->lds_per_block)
    return fail(PTK_ERR_UNSUPPORTED, "dimension %u does not fit the LDS staging of the device search", t->dim);
  if (k <= 64 && !no_register_list) {  // k-list in registers (K = 4 / 8 / 16 / 32 / 64 slots compiled)
    Timer timer(t, s);
    int rc = PTK_OK;
#define PTK_LAUNCH_ND_REG(KK)                                                                                       \
  do {                                                                                                              \
    rc = allow_lds(ptk::knn_nd_reg_kernel<KK, S, OVF, M>, base);                                                    \
    if (rc == PTK_OK)                                                                                               \
      hipLaunchKernelGGL((ptk::knn_nd_reg_kernel<KK, S, OVF, M>), dim3(blocks), dim3(64), base, s, t->dev_nd, d_q,  \
                         perm, nq, k, inv_ratio(e), d_out);                                                         \
  } while (0)
    if (k <= 4) PTK_LAUNCH_ND_REG(4);
    else if (k <= 8) PTK_LAUNCH_ND_REG(8);
    else if (k <= 16) PTK_LAUNCH_ND_REG(16);
    else if (k <= 32) PTK_LAUNCH_ND_REG(32);
    else PTK_LAUNCH_ND_REG(64);
#undef PTK_LAUNCH_ND_REG
    if (rc != PTK_OK) return rc;
    PTK_HIP(hipGetLastError());
    timer.stop(0, nq);
    return PTK_OK;
  }
  const size_t list_bytes = (size_t)k * 64 * 8;
  const bool list_lds = base + list_bytes <= 64 * 1024;
  const size_t smem = base + (list_lds ? list_bytes : 0);
  Timer timer(t, s);
  if (list_lds) {
    hipLaunchKernelGGL((ptk::knn_nd_kernel<S, OVF, true, M>), dim3(blocks), dim3(64), smem, s, t->dev_nd, d_q, perm, nq, k,
                       inv_ratio(e), d_out);
  } else {
    int rc = allow_lds(ptk::knn_nd_kernel<S, OVF, false, M>, smem);
    if (rc != PTK_OK) return rc;
    hipLaunchKernelGGL((ptk::knn_nd_kernel<S, OVF, false, M>), dim3(blocks), dim3(64), smem, s, t->dev_nd, d_q, perm, nq, k,
                       inv_ratio(e), d_out);
  }
  PTK_HIP(hipGetLastError());
  timer.stop(0, nq);
  return PTK_OK;
}

template <int OVF, class M = ptk::MetricL2>
int launch_radius_nd(const ptk_tree* t, const float* d_q, uint64_t nq, float radius, float e, bool fill,
                     uint64_t* d_counts, const uint64_t* d_offsets, ptk::Neighbor* d_out, hipStream_t s,
                     const uint32_t* perm = nullptr, const uint32_t* n_dev = nullptr) {
  constexpr int S = 16;
  const uint32_t blocks = (uint32_t)((nq + 63) / 64);
  const size_t smem = (size_t)S * 64 * 8 + (size_t)t->dim * 64 * 8;
  if (smem > t->lds_per_block)
    return fail(PTK_ERR_UNSUPPORTED, "dimension %u does not fit the LDS staging of the device search", t->dim);
  Timer timer(t, s);
  if (!fill) {
    int rc = allow_lds(ptk::radius_nd_kernel<S, OVF, false, M>, smem);
    if (rc != PTK_OK) return rc;
    hipLaunchKernelGGL((ptk::radius_nd_kernel<S, OVF, false, M>), dim3(blocks), dim3(64), smem, s, t->dev_nd, d_q, nq,
                       radius, inv_ratio(e), d_counts, d_offsets, d_out, perm, nullptr);
  } else {
    int rc = allow_lds(ptk::radius_nd_kernel<S, OVF, true, M>, smem);
    if (rc != PTK_OK) return rc;
    hipLaunchKernelGGL((ptk::radius_nd_kernel<S, OVF, true, M>), dim3(blocks), dim3(64), smem, s, t->dev_nd, d_q, nq,
                       radius, inv_ratio(e), d_counts, d_offsets, d_out, perm, n_dev);
  }
  PTK_HIP(hipGetLastError());
  timer.stop(0, n_dev ? 0 : nq);
  return PTK_OK;
}

template <int OVF, class M = ptk::MetricL2>
int launch_radius_nd_capture(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float radius,
                             float e, uint64_t* d_counts, const ptk::RadiusCapture& cap, hipStream_t s) {
  constexpr int S = 16;
  const uint32_t blocks = (uint32_t)((nq + 63) / 64);
  const size_t smem = (size_t)S * 64 * 8 + (size_t)t->dim * 64 * 8 + 16;  // + the cursor of the wavefront's log
  if (smem > t->lds_per_block)
    return fail(PTK_ERR_UNSUPPORTED, "dimension %u does not fit the LDS staging of the device search", t->dim);
  Timer timer(t, s);
  int rc = allow_lds(ptk::radius_nd_capture_kernel<S, OVF, M>, smem);
  if (rc != PTK_OK) return rc;
  PTK_HIP(hipMemsetAsync(cap.counters, 0, ptk::kCapSubPools * ptk::kCapCounterStride * 4, s));
  hipLaunchKernelGGL((ptk::radius_nd_capture_kernel<S, OVF, M>), dim3(blocks), dim3(64), smem, s, t->dev_nd, d_q, perm,
                     nq, radius, inv_ratio(e), d_counts, cap);
  PTK_HIP(hipGetLastError());
  timer.stop(0, nq);
  return PTK_OK;
}

// Runs CALL with OVF bound to the spill capacity the tree's depth needs.
#define PTK_WITH_OVF(SLDS, CALL)                                                                            \
  switch (ovf_class(t, SLDS)) {                                                                             \
    case 0: { constexpr int OVF = 64; rc = CALL; } break;                                                   \
    case 1: { constexpr int OVF = 256; rc = CALL; } break;                                                  \
    case 2: { constexpr int OVF = 2048; rc = CALL; } break;                                                 \
    default: rc = fail(PTK_ERR_UNSUPPORTED, "tree depth %u is too deep for the device stack", t->max_depth); \
  }

// Runs CALL with M bound to the metric of the handle (other than L2 squared).
#define PTK_WITH_METRIC(CALL)                                              \
  switch (t->metric.load()) {                                             \
    case PTK_METRIC_L1: { using M = ptk::MetricL1; CALL; } break;         \
    case PTK_METRIC_LPINF: { using M = ptk::MetricLInf; CALL; } break;    \
    case PTK_METRIC_LNINF: { using M = ptk::MetricLNInf; CALL; } break;   \
    default: { using M = ptk::MetricL2; CALL; } break;                    \
  }

// ---- topological metrics (ptk_kernels_topo.hpp) ------------------------------------------------
bool topological(const ptk_tree* t) {
  const int m = t->metric.load();
  return m == PTK_METRIC_SO2 || m == PTK_METRIC_SE2_SQUARED;
}
#define PTK_WITH_TOPO(CALL)                                         \
  if (t->metric.load() == PTK_METRIC_SO2) {                         \
    using T = ptk::TopoSO2;                                         \
    CALL;                                                           \
  } else {                                                          \
    using T = ptk::TopoSE2;                                         \
    CALL;                                                           \
  }

template <int OVF>
int launch_knn_topo(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, uint32_t k, float e,
                    ptk::Neighbor* d_out, hipStream_t s, bool short_tree) {
  const uint32_t blocks = (uint32_t)((nq + 63) / 64);
  const size_t smem = (size_t)16 * 64 * 8;
  Timer timer(t, s);
#define PTK_LAUNCH_TOPO_REG(KK)                                                                                         \
  PTK_WITH_TOPO({ hipLaunchKernelGGL((ptk::knn_topo_reg_kernel<KK, 16, OVF, T>), dim3(blocks), dim3(64), smem, s, t->dev, \
                                     d_q, t->dim, perm, nq, k, inv_ratio(e), d_out); })
  if (k <= 64 && !short_tree) {
    if (k <= 4) { PTK_LAUNCH_TOPO_REG(4); }
    else if (k <= 8) { PTK_LAUNCH_TOPO_REG(8); }
    else if (k <= 16) { PTK_LAUNCH_TOPO_REG(16); }
    else if (k <= 32) { PTK_LAUNCH_TOPO_REG(32); }
    else { PTK_LAUNCH_TOPO_REG(64); }
  } else {
    PTK_WITH_TOPO({ hipLaunchKernelGGL((ptk::knn_topo_kernel<16, OVF, T>), dim3(blocks), dim3(64), smem, s, t->dev, d_q,
                                       t->dim, perm, nq, k, inv_ratio(e), d_out); });
  }
#undef PTK_LAUNCH_TOPO_REG
  PTK_HIP(hipGetLastError());
  timer.stop(0, nq);
  return PTK_OK;
}

template <int OVF>
int launch_radius_topo(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float radius, float e,
                       bool fill, uint64_t* d_counts, const uint64_t* d_offsets, ptk::Neighbor* d_out, hipStream_t s) {
  const uint32_t blocks = (uint32_t)((nq + 63) / 64);
  const size_t smem = (size_t)16 * 64 * 8;
  Timer timer(t, s);
  if (fill) {
    PTK_WITH_TOPO({ hipLaunchKernelGGL((ptk::radius_topo_kernel<16, OVF, true, T>), dim3(blocks), dim3(64), smem, s, t->dev,
                                       d_q, t->dim, perm, nq, radius, inv_ratio(e), d_counts, d_offsets, d_out); });
  } else {
    PTK_WITH_TOPO({ hipLaunchKernelGGL((ptk::radius_topo_kernel<16, OVF, false, T>), dim3(blocks), dim3(64), smem, s, t->dev,
                                       d_q, t->dim, perm, nq, radius, inv_ratio(e), d_counts, d_offsets, d_out); });
  }
  PTK_HIP(hipGetLastError());
  timer.stop(0, fill ? 0 : nq);
  return PTK_OK;
}

int dispatch_knn1(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float e,
                  ptk::Neighbor* d_out, hipStream_t s, Scratch& scratch) {
  int rc = PTK_OK;
  switch (ovf_class_of(knn1_depth(t), 16)) {  // (the depth of what is traversed: the view without the piles if there is one)
    case 0: rc = launch_knn1_two_phase<64>(t, d_q, perm, nq, e, d_out, s, scratch); break;
    case 1: rc = launch_knn1_two_phase<256>(t, d_q, perm, nq, e, d_out, s, scratch); break;
    case 2: rc = launch_knn1_two_phase<2048>(t, d_q, perm, nq, e, d_out, s, scratch); break;
    default: rc = fail(PTK_ERR_UNSUPPORTED, "tree depth %u is too deep for the device stack", knn1_depth(t));
  }
  return rc;
}

}  // namespace

// =====================================================================================
extern "C" {

int ptk_version(void) { return PTK_VERSION; }

const char* ptk_last_error(void) { return g_error.c_str(); }

int ptk_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return -1;
  return count;
}

int ptk_warmup(int32_t device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(PTK_ERR_DEVICE, "no HIP device is visible");
  if (device >= count) return fail(PTK_ERR_INVALID, "device %d out of range (%d visible)", device, count);
  g_warmup.start(device);
  return PTK_OK;
}

int ptk_tree_create(const ptk_tree_desc* d, ptk_tree** out) {
  if (out == nullptr) return fail(PTK_ERR_INVALID, "null out pointer");
  *out = nullptr;
  if (d == nullptr || d->points == nullptr || d->nodes == nullptr || d->indices == nullptr)
    return fail(PTK_ERR_INVALID, "null descriptor field");
  if (d->dim == 0 || d->n_points == 0 || d->n_nodes == 0)
    return fail(PTK_ERR_INVALID, "dim, n_points and n_nodes must be positive");
  if (d->n_points >= (1ull << 31)) return fail(PTK_ERR_INVALID, "n_points must be < 2^31");
  if (d->n_nodes >= (1ull << 32) - 1) return fail(PTK_ERR_INVALID, "n_nodes must be < 2^32 - 1");
  ptk_tree* t = new (std::nothrow) ptk_tree;
  if (t == nullptr) return fail(PTK_ERR_NOMEM, "out of memory");
  try {
    t->dim = d->dim;
    t->n_points = d->n_points;
    t->nodes.assign(d->nodes, d->nodes + d->n_nodes);
    t->indices.assign(d->indices, d->indices + d->n_points);
    if (d->root_min != nullptr && d->root_max != nullptr) {
      t->root_min.assign(d->root_min, d->root_min + d->dim);
      t->root_max.assign(d->root_max, d->root_max + d->dim);
    }
  } catch (const std::bad_alloc&) {
    delete t;
    return fail(PTK_ERR_NOMEM, "out of memory");
  }
  return finish_create(t, d->points, d->device, out);
}

int ptk_tree_create_from_points(const float* points, uint64_t n_points, uint32_t dim, uint64_t max_leaf_size,
                                int32_t device, ptk_tree** out) {
  if (out == nullptr) return fail(PTK_ERR_INVALID, "null out pointer");
  *out = nullptr;
  if (points == nullptr) return fail(PTK_ERR_INVALID, "null points");
  if (dim == 0 || n_points == 0 || max_leaf_size == 0)
    return fail(PTK_ERR_INVALID, "dim, n_points and max_leaf_size must be positive");
  if (n_points >= (1ull << 31)) return fail(PTK_ERR_INVALID, "n_points must be < 2^31");
  ptk_tree* t = new (std::nothrow) ptk_tree;
  if (t == nullptr) return fail(PTK_ERR_NOMEM, "out of memory");
  g_warmup.start(device);
  try {
    using namespace pico_tree;
    using space_t = space_map<point_map<float const, dynamic_extent>>;
    space_t space(points, n_points, dim);
    internal::space_view<space_t> view(space);
    // (the two outer bounds per branch come for free while the child boxes are at hand; only the
    // topological metrics ever read them)
    CreateClock clock(t->create_ms);
    // Large clouds bound for a device: the partitions of the top levels are made there (ptk_build.hpp), the subtrees
    // below by the host's workers -- the same tree.  PTK_DEVICE_BUILD=0: everything on the host.
    internal::flat_tree<int, float, dynamic_extent> flat(dim);
    bool built = false;
    if (device != kDeviceNone && n_points >= (1ull << 18) && env_int("PTK_DEVICE_BUILD", 1) != 0) {
      int count = 0, dev = device;
      if (hipGetDeviceCount(&count) == hipSuccess && count > 0 && (dev >= 0 || hipGetDevice(&dev) == hipSuccess) && dev < count) {
        DeviceGuard guard(dev);
        g_warmup.wait(dev);
        double ms[2] = {0, 0};
        const char* why = "hipSetDevice failed";
        if (guard.ok) built = ptk::device_top_build(points, n_points, dim, (size_t)max_leaf_size, build_threads(), view, flat, ms, &why);
        if (!built && clock.on) std::fprintf(stderr, "[ptk create] top levels not on the device: %s\n", why);
        if (built && clock.on)
          std::fprintf(stderr, "[ptk create] top levels on the device     %8.2f ms\n[ptk create] subtrees on the host          %8.2f ms\n",
                       ms[0], ms[1]);
      }
      (void)hipGetLastError();
    }
    if (!built)
      flat = internal::build_flat_tree<int>(view, max_leaf_size_t(max_leaf_size), bounds_from_space,
                                            sliding_midpoint_max_side, true, build_threads());
    clock.lap(built ? "device + host build" : "host build", 0);
    t->dim = dim;
    t->n_points = n_points;
    t->nodes.resize(flat.nodes.size());
    t->outer.resize(flat.outer_bounds.size() * 2);
    ptk::parallel_chunks(flat.nodes.size(), build_threads(), 1u << 16, [&](size_t lo, size_t hi, unsigned) {
      std::memcpy(t->nodes.data() + lo, flat.nodes.data() + lo, (hi - lo) * sizeof(ptk_node));
      if (!flat.outer_bounds.empty()) std::memcpy(t->outer.data() + 2 * lo, flat.outer_bounds.data() + lo, (hi - lo) * 2 * sizeof(float));
    });
    t->indices = std::move(flat.indices);
    t->builder_made = true;
    t->n_leaves = flat.leaf_count;
    t->max_leaf_count = (uint32_t)flat.max_leaf_points;
    t->max_depth = flat.max_depth;
    for (uint32_t a = 0; a < 3; ++a) t->axis_splits[a] = a < dim && a < flat.axis_weight.size() ? flat.axis_weight[a] / (double)n_points : 0.0;
    t->root_min.assign(flat.root_box.min(), flat.root_box.min() + dim);
    t->root_max.assign(flat.root_box.max(), flat.root_box.max() + dim);
    clock.lap("copy into the handle", 1);
  } catch (const std::bad_alloc&) {
    delete t;
    return fail(PTK_ERR_NOMEM, "out of memory");
  } catch (const std::length_error& err) {  // degenerate point set: see flat_builder::grow
    delete t;
    return fail(PTK_ERR_UNSUPPORTED, "%s", err.what());
  }
  return finish_create(t, points, device, out);
}

void ptk_tree_destroy(ptk_tree* t) {
  if (t == nullptr) return;
  if (t->device >= 0) {
    DeviceGuard guard(t->device);
    for (PendingEvent& p : t->profile.pending) {
      (void)hipEventDestroy(p.a);
      if (!p.keep_b) (void)hipEventDestroy(p.b);
    }
    for (hipEvent_t e : t->profile.idle) (void)hipEventDestroy(e);
    auto drop_side = [](Workspace& w) {
      if (w.side) (void)hipStreamSynchronize(w.side);
      if (w.fork) (void)hipEventDestroy(w.fork);
      if (w.join) (void)hipEventDestroy(w.join);
      if (w.side) (void)hipStreamDestroy(w.side);
      if (w.d_sample) (void)hipFree(w.d_sample);
    };
    drain_workspace(t->ws);
    drop_side(t->ws);
    if (t->ws.done) (void)hipEventDestroy(t->ws.done);
    if (t->ws.base) (void)hipFree(t->ws.base);
    if (t->ws.cap_base) (void)hipFree(t->ws.cap_base);
    for (Workspace& w : t->extra_ws) {
      drain_workspace(w);
      drop_side(w);
      if (w.done) (void)hipEventDestroy(w.done);
      if (w.base) (void)hipFree(w.base);
    }
    if (t->io.d_in) (void)hipFree(t->io.d_in);
    if (t->io.d_out) (void)hipFree(t->io.d_out);
    for (int i = 0; i < HostIo::kRing; ++i) {
      if (t->io.h_in[i]) (void)hipHostFree(t->io.h_in[i]);
      if (t->io.h_out[i]) (void)hipHostFree(t->io.h_out[i]);
      if (t->io.up_done[i]) (void)hipEventDestroy(t->io.up_done[i]);
      if (t->io.down_done[i]) (void)hipEventDestroy(t->io.down_done[i]);
    }
    for (hipStream_t st : {t->io.up, t->io.down, t->io.search[0], t->io.search[1]})
      if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : t->io.searched)
      if (ev) (void)hipEventDestroy(ev);
    if (t->d_nodes) (void)hipFree(t->d_nodes);
    if (t->d_pts) (void)hipFree(t->d_pts);
    if (t->d_ranges) (void)hipFree(t->d_ranges);
    if (t->d_axes) (void)hipFree(t->d_axes);
    if (t->d_index) (void)hipFree(t->d_index);
    if (t->d_outer) (void)hipFree(t->d_outer);
    if (t->d_cells) (void)hipFree(t->d_cells);
    if (t->d_nodes1) (void)hipFree(t->d_nodes1);
    if (t->d_ranges1) (void)hipFree(t->d_ranges1);
    if (t->d_pile_of_point) (void)hipFree(t->d_pile_of_point);
    if (t->d_pile_recs) (void)hipFree(t->d_pile_recs);
  }
  delete t;
}

int ptk_tree_get_info(const ptk_tree* t, ptk_tree_info* info) {
  if (t == nullptr || info == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  info->dim = t->dim;
  info->n_points = t->n_points;
  info->n_nodes = t->nodes.size();
  info->n_leaves = t->n_leaves;
  info->max_depth = t->max_depth;
  info->max_leaf_count = t->max_leaf_count;
  info->device_bytes = t->device_bytes;
  info->device = t->device;
  return PTK_OK;
}

int ptk_tree_get_flat(const ptk_tree* t, ptk_node* nodes, int32_t* indices, float* root_min, float* root_max) {
  if (t == nullptr) return fail(PTK_ERR_INVALID, "null tree");
  if (nodes) std::memcpy(nodes, t->nodes.data(), t->nodes.size() * sizeof(ptk_node));
  if (indices) std::memcpy(indices, t->indices.data(), t->indices.size() * sizeof(int32_t));
  if (root_min) std::memcpy(root_min, t->root_min.data(), t->dim * sizeof(float));
  if (root_max) std::memcpy(root_max, t->root_max.data(), t->dim * sizeof(float));
  return PTK_OK;
}

static int serialize_tree(const ptk_tree* t, bool topological, void* buf, uint64_t cap, uint64_t* size) {
  if (t == nullptr || size == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  if (topological && t->outer.size() != 2 * t->nodes.size())
    return fail(PTK_ERR_INVALID, "this tree has no outer bounds (ptk_tree_set_outer_bounds): it cannot be written as a topological tree");
  try {
    using tree_t = pico_tree::internal::flat_tree<int, float, pico_tree::dynamic_extent>;
    tree_t flat(t->dim);
    flat.indices.assign(t->indices.begin(), t->indices.end());
    std::memcpy(flat.root_box.min(), t->root_min.data(), t->dim * sizeof(float));
    std::memcpy(flat.root_box.max(), t->root_max.data(), t->dim * sizeof(float));
    flat.nodes.resize(t->nodes.size());
    std::memcpy(static_cast<void*>(flat.nodes.data()), t->nodes.data(), t->nodes.size() * sizeof(ptk_node));
    if (topological) {  // the four bounds of kd_tree_branch_double (kd_tree_node.hpp:52-67)
      flat.keep_outer_bounds = true;
      flat.outer_bounds.resize(t->nodes.size());
      std::memcpy(static_cast<void*>(flat.outer_bounds.data()), t->outer.data(), t->outer.size() * sizeof(float));
    }
    std::ostringstream os(std::ios::out | std::ios::binary);
    pico_tree::internal::write_flat_tree(flat, os);
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

int ptk_tree_serialize(const ptk_tree* t, void* buf, uint64_t cap, uint64_t* size) {
  return serialize_tree(t, false, buf, cap, size);
}

int ptk_tree_serialize_topological(const ptk_tree* t, void* buf, uint64_t cap, uint64_t* size) {
  return serialize_tree(t, true, buf, cap, size);
}

static int create_from_stream(const float* points, uint64_t n_points, uint32_t dim, const void* stream,
                              uint64_t stream_bytes, bool topological, int32_t device, ptk_tree** out) {
  if (out == nullptr) return fail(PTK_ERR_INVALID, "null out pointer");
  *out = nullptr;
  if (points == nullptr || stream == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  if (dim == 0 || n_points == 0) return fail(PTK_ERR_INVALID, "dim and n_points must be positive");
  ptk_tree* t = nullptr;
  try {
    using tree_t = pico_tree::internal::flat_tree<int, float, pico_tree::dynamic_extent>;
    std::istringstream is(std::string(static_cast<const char*>(stream), stream_bytes), std::ios::in | std::ios::binary);
    tree_t flat = pico_tree::internal::read_flat_tree<tree_t>(is, topological, dim, n_points);
    if (flat.root_box.size() != dim) return fail(PTK_ERR_INVALID, "stream is %zu-dimensional, points are %u-dimensional",
                                                 (size_t)flat.root_box.size(), dim);
    if (flat.indices.size() != n_points)
      return fail(PTK_ERR_INVALID, "stream indexes %zu points, %llu were given", flat.indices.size(),
                  (unsigned long long)n_points);
    t = new ptk_tree;
    t->dim = dim;
    t->n_points = n_points;
    t->nodes.resize(flat.nodes.size());
    std::memcpy(t->nodes.data(), flat.nodes.data(), flat.nodes.size() * sizeof(ptk_node));
    if (topological) {  // {left_min, right_max} per node: what the topological metrics need besides the 16-byte record
      t->outer.resize(flat.outer_bounds.size() * 2);
      if (!flat.outer_bounds.empty()) std::memcpy(t->outer.data(), flat.outer_bounds.data(), t->outer.size() * sizeof(float));
    }
    t->indices.assign(flat.indices.begin(), flat.indices.end());
    t->root_min.assign(flat.root_box.min(), flat.root_box.min() + dim);
    t->root_max.assign(flat.root_box.max(), flat.root_box.max() + dim);
  } catch (const std::bad_alloc&) {
    delete t;
    return fail(PTK_ERR_NOMEM, "out of memory");
  } catch (const std::exception& e) {
    delete t;
    return fail(PTK_ERR_INVALID, "bad kd_tree stream: %s", e.what());
  }
  return finish_create(t, points, device, out);
}

int ptk_tree_create_from_stream(const float* points, uint64_t n_points, uint32_t dim, const void* stream,
                                uint64_t stream_bytes, int32_t device, ptk_tree** out) {
  return create_from_stream(points, n_points, dim, stream, stream_bytes, false, device, out);
}

int ptk_tree_create_from_topological_stream(const float* points, uint64_t n_points, uint32_t dim, const void* stream,
                                            uint64_t stream_bytes, int32_t device, ptk_tree** out) {
  return create_from_stream(points, n_points, dim, stream, stream_bytes, true, device, out);
}

int ptk_tree_set_outer_bounds(ptk_tree* t, const float* outer, uint64_t n_nodes) {
  if (t == nullptr || outer == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  if (n_nodes != t->nodes.size()) return fail(PTK_ERR_INVALID, "outer bounds for %llu nodes, the tree has %zu",
                                              (unsigned long long)n_nodes, t->nodes.size());
  try {
    t->outer.assign(outer, outer + 2 * n_nodes);
    {
      std::lock_guard<std::mutex> lock(t->host_flat_mutex);
      t->host_flat.reset();
    }
  } catch (const std::bad_alloc&) {
    return fail(PTK_ERR_NOMEM, "out of memory");
  }
  return PTK_OK;
}

int ptk_tree_get_outer_bounds(const ptk_tree* t, float* outer) {
  if (t == nullptr || outer == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  if (t->outer.size() != 2 * t->nodes.size()) return fail(PTK_ERR_INVALID, "this tree has no outer bounds");
  std::memcpy(outer, t->outer.data(), t->outer.size() * sizeof(float));
  return PTK_OK;
}

int ptk_tree_set_metric(ptk_tree* t, int metric) {
  if (t == nullptr || metric < PTK_METRIC_L2_SQUARED || metric > PTK_METRIC_SE2_SQUARED)
    return fail(PTK_ERR_INVALID, "bad metric");
  if (metric == PTK_METRIC_SO2 || metric == PTK_METRIC_SE2_SQUARED) {
    if (metric == PTK_METRIC_SO2 && t->dim != 1) return fail(PTK_ERR_INVALID, "metric_so2 is a metric of 1-dimensional points");
    if (metric == PTK_METRIC_SE2_SQUARED && t->dim != 3)
      return fail(PTK_ERR_INVALID, "metric_se2_squared is a metric of 3-dimensional points (x, y, angle)");
    if (t->outer.size() != 2 * t->nodes.size())
      return fail(PTK_ERR_INVALID, "the topological metrics need the outer bounds of every branch (ptk_tree_set_outer_bounds)");
    if (t->device >= 0 && t->gpu_layout && t->d_outer == nullptr) {  // branch order of the device records
      ptk::TreeStats st;
      std::vector<uint32_t> branch_id;
      std::string err = ptk::analyse_stream(t->dim, t->n_points, t->nodes.data(), t->nodes.size(), st, &branch_id);
      if (!err.empty()) return fail(PTK_ERR_INVALID, "%s", err.c_str());
      std::vector<float> dev_outer(2 * std::max<size_t>(t->nodes.size() - st.n_leaves, 1), 0.0f);
      for (size_t i = 0; i < t->nodes.size(); ++i) {
        if (t->nodes[i].right == PTK_LEAF) continue;
        dev_outer[2 * (size_t)branch_id[i]] = t->outer[2 * i];
        dev_outer[2 * (size_t)branch_id[i] + 1] = t->outer[2 * i + 1];
      }
      DeviceGuard guard(t->device);
      PTK_HIP(hipMalloc(&t->d_outer, dev_outer.size() * sizeof(float)));
      PTK_HIP(hipMemcpy(t->d_outer, dev_outer.data(), dev_outer.size() * sizeof(float), hipMemcpyHostToDevice));
      t->dev.outer = static_cast<const float2*>(t->d_outer);
      t->device_bytes += dev_outer.size() * sizeof(float);
    }
  }
  t->metric.store(metric);
  return PTK_OK;
}

int ptk_tree_set_reorder(ptk_tree* t, int mode) {
  if (t == nullptr || mode < PTK_REORDER_AUTO || mode > PTK_REORDER_OFF)
    return fail(PTK_ERR_INVALID, "bad reorder mode");
  t->reorder.store(mode);
  return PTK_OK;
}

// ---- knn --------------------------------------------------------------------------------

int ptk_search_knn_device(const ptk_tree* t, const float* d_q, uint64_t nq, uint32_t k, float e,
                          ptk_neighbor* d_out, void* stream) {
  int rc = check_search(t, d_q, nq);
  if (rc != PTK_OK) return rc;
  if (k == 0) return fail(PTK_ERR_INVALID, "k must be >= 1");
  if (!(e > 0.0f)) return fail(PTK_ERR_INVALID, "approximation ratio e must be > 0");
  if (nq == 0) return PTK_OK;
  if (d_out == nullptr) return fail(PTK_ERR_INVALID, "null output buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // k > n_points: what the reference's iterator-range search_knn does with a range longer than the
  // tree (search_visitor.hpp:95-110; its Python binding passes k through unclamped): the n_points
  // neighbours in order, the last slot's distance left at the FLT_MAX sentinel.  The slots in
  // between are the caller's in the reference; here they are zeroed.  (The register k-list assumes
  // every slot gets filled, so these rows take the list-in-the-row kernels.)
  const bool short_tree = k > t->n_points;
  // Very large batches go through in pieces of at most 2^25 queries: the scratch of a piece stays
  // at a few GB and every 32-bit index in the kernels holds (PTK_MAX_BATCH shrinks it for tests).
  const uint64_t piece = (uint64_t)std::max(1, env_int("PTK_MAX_BATCH", 1 << 25));
  if (nq > piece) {
    for (uint64_t done = 0; done < nq; done += piece) {
      const uint64_t n = std::min(piece, nq - done);
      rc = ptk_search_knn_device(t, d_q + done * t->dim, n, k, e, d_out + done * k, stream);
      if (rc != PTK_OK) return rc;
    }
    return PTK_OK;
  }
  DeviceGuard guard(t->device);
  if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", t->device);
  // (on the tree's device, once per piece: a null stream is the null stream of THAT device)
  if (short_tree) PTK_HIP(hipMemsetAsync(d_out, 0, (size_t)nq * k * sizeof(ptk_neighbor), s));
  const bool l2 = t->metric.load() == PTK_METRIC_L2_SQUARED;
  if (topological(t)) {
    if (deep_tree(t)) return fail(PTK_ERR_UNSUPPORTED, "tree depth %u is too deep for the device stack", t->max_depth);
    const bool reorder = want_reorder(t, nq);
    Scratch scratch(t, s, /*per_stream=*/true);
    rc = scratch.reserve(reorder ? permutation_scratch_bytes(nq) : 0);
    if (rc != PTK_OK) return rc;
    uint32_t* perm = nullptr;
    if (reorder) {
      rc = make_permutation(t, d_q, nq, s, scratch, &perm);
      if (rc != PTK_OK) return rc;
    }
    PTK_WITH_OVF(16, (launch_knn_topo<OVF>(t, d_q, perm, nq, k, e, reinterpret_cast<ptk::Neighbor*>(d_out), s, short_tree)));
    return rc;
  }
  const bool knn1_view = k == 1 && l2 && t->dim <= 3 && t->n_piles != 0 && ovf_class_of(knn1_depth(t), 16) != kDeepClass;
  if (deep_tree(t) && !knn1_view) {  // a few queries at a time, the record stacks spilling to HBM (any k, any metric)
    const DeepPlan plan = deep_plan(t, nq);
    Scratch scratch(t, s, /*per_stream=*/true);
    rc = scratch.reserve(plan.bytes());
    if (rc != PTK_OK) return rc;
    ptk::Record* spill = scratch.take<ptk::Record>((size_t)plan.piece * plan.cap);
    if (spill == nullptr) return fail(PTK_ERR_NOMEM, "scratch block too small");
    auto* o = reinterpret_cast<ptk::Neighbor*>(d_out);
    Timer timer(t, s);
    for (uint64_t lo = 0; lo < nq; lo += plan.piece) {
      const uint64_t n = std::min<uint64_t>(plan.piece, nq - lo);
      const uint32_t blocks = (uint32_t)((n + 63) / 64);
      if (t->dim > 3) {
        ptk::DevTreeND dev = t->dev_nd;
        dev.deep_spill = spill;
        dev.deep_cap = plan.cap;
        const size_t smem = (size_t)16 * 64 * 8 + (size_t)t->dim * 64 * 8;
        if (smem > t->lds_per_block)
          return fail(PTK_ERR_UNSUPPORTED, "dimension %u does not fit the LDS staging of the device search", t->dim);
        PTK_WITH_METRIC({
          rc = allow_lds(ptk::knn_nd_kernel<16, -1, false, M>, smem);
          if (rc == PTK_OK)
            hipLaunchKernelGGL((ptk::knn_nd_kernel<16, -1, false, M>), dim3(blocks), dim3(64), smem, s, dev,
                               d_q + lo * t->dim, nullptr, n, k, inv_ratio(e), o + lo * k);
        });
        if (rc != PTK_OK) return rc;
      } else {
        ptk::DevTree dev = t->dev;
        dev.deep_spill = spill;
        dev.deep_cap = plan.cap;
        PTK_WITH_METRIC({
          hipLaunchKernelGGL((ptk::knn_kernel<16, -1, 64, 4, false, M>), dim3(blocks), dim3(64), (size_t)16 * 64 * 8, s,
                             dev, d_q + lo * t->dim, t->dim, nullptr, n, k, inv_ratio(e), o + lo * k);
        });
      }
      PTK_HIP(hipGetLastError());
    }
    timer.stop(0, nq);
    return PTK_OK;
  }
  const bool reorder = want_reorder(t, nq);
  Scratch scratch(t, s, /*per_stream=*/true);
  rc = scratch.reserve((reorder ? permutation_scratch_bytes(nq) : 0) +
                       (k == 1 && l2 && t->dim <= 3 ? two_phase_scratch_bytes(t, nq) : 0) +
                       (k > 1 && k <= 64 && l2 && t->dim <= 3 && knn_cap(e, nq, k) != 0u ? knn_coop_scratch_bytes(t, nq) : 0));
  if (rc != PTK_OK) return rc;
  uint32_t* perm = nullptr;
  if (reorder) {  // Morton order along the first three axes, whatever the dimension
    // (the general kernels run every query to its end in its lane: the expensive queries to the front of the launch)
    // (the two-phase k = 1 search orders its own continuations: a batch that arrives coherent is not sorted again --
    // REORDER_AUTO only; the general kernels want the expensive queries in front whatever the order)
    const bool may_skip = k == 1 && l2 && t->dim <= 3 && t->reorder.load() == PTK_REORDER_AUTO;
    rc = make_permutation(t, d_q, nq, s, scratch, &perm, t->dim <= 3 && !(k == 1 && l2) ? ptk::kCellsEmptyFirst : 0u, may_skip);
    if (rc != PTK_OK) return rc;
  }
  if (t->dim > 3) {
    PTK_WITH_METRIC(PTK_WITH_OVF(16, (launch_knn_nd<OVF, M>(t, d_q, perm, nq, k, e, reinterpret_cast<ptk::Neighbor*>(d_out), s, short_tree))));
    return rc;
  }
  if (k == 1 && l2) {  // the two-phase search is built for the default metric
    rc = dispatch_knn1(t, d_q, perm, nq, e, reinterpret_cast<ptk::Neighbor*>(d_out), s, scratch);
  } else if (k <= knn_reg_max(l2) && !short_tree) {
    PTK_WITH_METRIC(PTK_WITH_OVF(kGenRing, (launch_knn_reg<kGenRing, OVF, 64, kGenLeafB, M>(t, d_q, perm, nq, k, e, reinterpret_cast<ptk::Neighbor*>(d_out), s, &scratch))));
  } else {
    PTK_WITH_METRIC(PTK_WITH_OVF(16, (launch_knn<16, OVF, 64, 4, M>(t, d_q, perm, nq, k, e, reinterpret_cast<ptk::Neighbor*>(d_out), s))));
  }
  return rc;
}

// Grow-only device block of the host-buffer entry points (rounded up to 1 MiB; the old contents are dropped).
static int grow_device_block(char** p, size_t* capacity, size_t bytes) {
  if (bytes <= *capacity) return PTK_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *capacity = 0;
  const size_t want = (bytes + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
  if (hipMalloc((void**)p, want) != hipSuccess) {
    (void)hipGetLastError();
    *p = nullptr;
    return fail(PTK_ERR_NOMEM, "out of device memory (%zu bytes)", want);
  }
  *capacity = want;
  return PTK_OK;
}

// Grow-only ring of pinned host pieces (the old contents are dropped): the first `slots` entries, `bytes` each.
static int grow_pinned_ring(char* (&ring)[HostIo::kRing], size_t* capacity, size_t bytes, int slots) {
  bool have = bytes <= *capacity;
  for (int i = 0; i < slots && have; ++i) have = ring[i] != nullptr;
  if (have) return PTK_OK;
  const size_t want = std::max(*capacity, (bytes + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1));
  if (want > *capacity) {  // (slots of another size are of no use)
    for (char*& p : ring) {
      if (p) (void)hipHostFree(p);
      p = nullptr;
    }
    *capacity = 0;
  }
  for (int i = 0; i < slots; ++i) {
    if (ring[i] != nullptr) continue;
    if (hipHostMalloc((void**)&ring[i], want, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      ring[i] = nullptr;
      return fail(PTK_ERR_NOMEM, "out of pinned host memory (%zu bytes)", want);
    }
  }
  *capacity = want;
  return PTK_OK;
}

// Is [p, p + bytes) page-locked host memory the runtime knows (ptk_host_alloc, hipHostMalloc, hipHostRegister)?  Copies
// to and from such memory need no staging.
static bool is_pinned_host(const void* p, size_t bytes) {
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  if (a.type != hipMemoryTypeHost) return false;
  if (bytes > 1) {
    hipPointerAttribute_t z{};
    if (hipPointerGetAttributes(&z, static_cast<const char*>(p) + bytes - 1) != hipSuccess || z.type != hipMemoryTypeHost) {
      (void)hipGetLastError();
      return false;
    }
  }
  return true;
}

// The pieces a host-buffer batch goes through in: first rows of every piece (and nq at the end).
//   k = 1   eight equal pieces, none below 256 k queries (the pipeline fills and drains by one piece on either side;
//           a search of fewer queries is all fixed cost).  On this platform the copy engine takes uploads and
//           downloads one after the other (48 and 56 GB/s alone, 53 GB/s together, profiles/r03a_pcie.json), so
//           the 144 MB of BASELINE config 2 cannot pass in less than 2.7 ms; ms per batch on one box
//           (profiles/r03_notes.txt item 7): one piece 8.5-22 (pageable copies, nothing overlapped), eight pieces 3.5,
//           four 3.6, two 3.9; a small first and last piece around large ones (the downloads then queue behind long
//           uploads) 3.8-4.3; uploads read by a kernel instead of the copy engine 5.2.
//   k > 1   the general kernels end with the tail of their slowest queries (3.4 ms for ANY piece of config 3 at
//           knn = 16), so a piece costs its tail: at most three pieces of at most 256 MB of rows.  Where the long
//           queries are handed to the cooperative search (`capped`: exact, metric_l2_squared, k <= 32, with a cap that
//           follows the piece -- knn_cap) a piece costs what its size costs, and the rows of the first one can leave
//           a millisecond after the call: eight pieces as for k = 1 (ms per batch of BASELINE config 3, three pieces /
//           eight: knn = 16 20.8 / 19.3 -- the 1 008 MB of the call at the 53 GB/s the link gives both directions
//           together are 19.0 --, knn = 8 15.0 / 10.5, knn = 4 9.9 / 7.0; profiles/r05_notes.txt item 13).
// PTK_HOST_PIECE = n: equal pieces of n queries (experiments).
static std::vector<uint64_t> host_pieces(uint64_t nq, uint32_t k, bool two_phase, bool capped) {
  std::vector<uint64_t> first;
  const int forced = env_int("PTK_HOST_PIECE", 0);
  if (forced > 0) {
    for (uint64_t lo = 0; lo < nq; lo += (uint64_t)forced) first.push_back(lo);
  } else if (!two_phase && !capped) {
    const size_t obytes = (size_t)nq * k * sizeof(ptk_neighbor);
    const uint64_t pieces = std::min<uint64_t>(std::max<uint64_t>(obytes / (size_t(256) << 20), 1), 3);
    const uint64_t per = (nq + pieces - 1) / pieces;
    for (uint64_t lo = 0; lo < nq; lo += per) first.push_back(lo);
  } else {
    const uint64_t per = std::max<uint64_t>((nq + 7) / 8, uint64_t(1) << 18);
    for (uint64_t lo = 0; lo < nq; lo += per) first.push_back(lo);
  }
  first.push_back(nq);
  return first;
}

static int search_knn_host(const ptk_tree* t, const float* q, uint64_t nq, uint32_t k, float e, ptk_neighbor* out);
int ptk_search_knn(const ptk_tree* t, const float* q, uint64_t nq, uint32_t k, float e, ptk_neighbor* out) {
  try {  // (containers and threads of the pipeline may throw: nothing leaves through the C boundary)
    return search_knn_host(t, q, nq, k, e, out);
  } catch (const std::bad_alloc&) {
    return fail(PTK_ERR_NOMEM, "out of host memory in the host-buffer search");
  } catch (const std::exception& ex) {
    return fail(PTK_ERR_DEVICE, "host-buffer search failed: %s", ex.what());
  }
}
static int search_knn_host(const ptk_tree* t, const float* q, uint64_t nq, uint32_t k, float e, ptk_neighbor* out) {
  int rc = check_search(t, q, nq);
  if (rc != PTK_OK) return rc;
  if (k == 0) return fail(PTK_ERR_INVALID, "k must be >= 1");
  if (nq == 0) return PTK_OK;
  if (out == nullptr) return fail(PTK_ERR_INVALID, "null output buffer");
  DeviceGuard guard(t->device);
  if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", t->device);
  const size_t row_in = (size_t)t->dim * sizeof(float), row_out = (size_t)k * sizeof(ptk_neighbor);
  HostIo& io = t->io;
  std::lock_guard<std::mutex> lock(io.mutex);
  for (hipStream_t* st : {&io.up, &io.down, &io.search[0], &io.search[1]})
    if (*st == nullptr) PTK_HIP(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
  for (int i = 0; i < HostIo::kRing; ++i) {
    if (io.up_done[i] == nullptr) PTK_HIP(hipEventCreateWithFlags(&io.up_done[i], hipEventDisableTiming));
    if (io.down_done[i] == nullptr) PTK_HIP(hipEventCreateWithFlags(&io.down_done[i], hipEventDisableTiming));
  }
  const bool two_phase = k == 1 && t->dim <= 3 && t->metric.load() == PTK_METRIC_L2_SQUARED &&
                         ovf_class_of(knn1_depth(t), 16) != kDeepClass;
  const bool capped = !two_phase && k > 1 && k <= 64 && t->dim <= 3 && t->metric.load() == PTK_METRIC_L2_SQUARED &&
                      knn_cap(e, std::max<uint64_t>((nq + 7) / 8, uint64_t(1) << 18), k) != 0u;
  const std::vector<uint64_t> first = host_pieces(nq, k, two_phase, capped);
  const uint64_t pieces = first.size() - 1;
  uint64_t piece = 0;  // the largest piece: the size of a ring slot
  for (uint64_t i = 0; i < pieces; ++i) piece = std::max(piece, first[i + 1] - first[i]);
  while (io.searched.size() < pieces) {
    hipEvent_t ev = nullptr;
    PTK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    io.searched.push_back(ev);
  }
  rc = grow_device_block(&io.d_in, &io.in_capacity, (size_t)nq * row_in);
  if (rc == PTK_OK) rc = grow_device_block(&io.d_out, &io.out_capacity, (size_t)nq * row_out);
  if (rc != PTK_OK) return rc;
  const int n_search_streams = env_int("PTK_HOST_STREAMS", 2);
  // Arrays that are page-locked already (ptk_host_alloc: what the Python wrapper returns its rows in) are copied to
  // and from directly; pageable ones go through the handle's pinned rings.
  const bool in_pinned = env_int("PTK_HOST_DIRECT", 1) != 0 && is_pinned_host(q, (size_t)nq * row_in);
  const bool out_pinned = env_int("PTK_HOST_DIRECT", 1) != 0 && is_pinned_host(out, (size_t)nq * row_out);
  if (!in_pinned) rc = grow_pinned_ring(io.h_in, &io.h_in_capacity, (size_t)piece * row_in, (int)std::min<uint64_t>(pieces, HostIo::kRing));
  // The rows of a piece come down in chunks of at most 32 MB, each copied into the caller's array while the next
  // is on the link (a piece of knn = 16 rows is 300 MB: one copy per piece left the host copy exposed).
  const uint64_t out_chunk = std::max<uint64_t>(std::min<uint64_t>(piece, (size_t(32) << 20) / row_out), 1);
  struct Chunk {
    uint64_t piece, lo, n;
  };
  std::vector<Chunk> chunks;
  for (uint64_t pi = 0; pi < pieces; ++pi)
    for (uint64_t lo = first[pi]; lo < first[pi + 1]; lo += out_chunk)
      chunks.push_back(Chunk{pi, lo, std::min(out_chunk, first[pi + 1] - lo)});
  if (rc == PTK_OK && !out_pinned)
    rc = grow_pinned_ring(io.h_out, &io.h_out_capacity, (size_t)out_chunk * row_out,
                          (int)std::min<uint64_t>(chunks.size(), HostIo::kRing));
  if (rc == PTK_ERR_NOMEM) {
    // No pinned memory to be had: the batch goes through piece by piece with plain (pageable) copies -- slower, not a failure.
    for (uint64_t i = 0; i < pieces; ++i) {
      const uint64_t lo = first[i], n = first[i + 1] - lo;
      PTK_HIP(hipMemcpy(io.d_in + lo * row_in, reinterpret_cast<const char*>(q) + lo * row_in, (size_t)n * row_in, hipMemcpyHostToDevice));
      rc = ptk_search_knn_device(t, reinterpret_cast<float*>(io.d_in) + lo * t->dim, n, k, e,
                                 reinterpret_cast<ptk_neighbor*>(io.d_out) + lo * k, io.search[0]);
      if (rc != PTK_OK) return rc;
      PTK_HIP(hipStreamSynchronize(io.search[0]));
      PTK_HIP(hipMemcpy(reinterpret_cast<char*>(out) + lo * row_out, io.d_out + lo * row_out, (size_t)n * row_out, hipMemcpyDeviceToHost));
    }
    return PTK_OK;
  }
  if (rc != PTK_OK) return rc;
  if (io.pool == nullptr) {
    const unsigned hc = std::thread::hardware_concurrency();
    const int want = env_int("PTK_IO_THREADS", hc >= 16 ? 8 : (hc >= 4 ? (int)hc / 2 : 1));
    io.pool.reset(new CopyPool((unsigned)std::max(0, want - 1)));  // (the calling threads copy too)
  }
  float* d_q = reinterpret_cast<float*>(io.d_in);
  ptk_neighbor* d_out = reinterpret_cast<ptk_neighbor*>(io.d_out);
  const char* src = reinterpret_cast<const char*>(q);
  char* dst = reinterpret_cast<char*>(out);

  const bool trace = env_int("PTK_HOST_TRACE", 0) != 0;
  const auto t_begin = std::chrono::steady_clock::now();
  auto stamp = [&](const char* what, uint64_t i) {
    if (!trace) return;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count();
    std::fprintf(stderr, "[ptk host] %9.1f us  %-22s piece %llu\n", us, what, (unsigned long long)i);
  };
  // What the two threads tell each other: pieces whose search has been enqueued, and whether anything failed.
  std::mutex m;
  std::condition_variable cv;
  uint64_t issued = 0;
  bool failed = false;
  std::string down_error;
  const int device = t->device;

  std::thread downloader([&] {
    if (hipSetDevice(device) != hipSuccess) {
      std::lock_guard<std::mutex> l(m);
      failed = true;
      down_error = "hipSetDevice failed in the download thread";
      return;
    }
    auto bail = [&](hipError_t he) {
      std::lock_guard<std::mutex> l(m);
      failed = true;
      down_error = std::string("HIP error in the download thread: ") + hipGetErrorString(he);
    };
    // `e` chunks have their D2H enqueued, `c` are copied out (c <= e <= c + kRing).  A D2H is enqueued only when the
    // search of its piece HAS finished (see the header of ptk_hostio.hpp); while a search is still running the
    // thread copies out what has arrived.
    const uint64_t n_chunks = chunks.size();
    uint64_t e_ = 0, c_ = 0, searched_pieces = 0;  // pieces [0, searched_pieces) are known to be done
    while (c_ < n_chunks) {
      bool can_enqueue = false;
      if (e_ < n_chunks && e_ < c_ + (uint64_t)HostIo::kRing) {
        const uint64_t pi = chunks[e_].piece;
        can_enqueue = pi < searched_pieces;
        if (!can_enqueue) {
          {
            std::unique_lock<std::mutex> l(m);
            if (e_ == c_) cv.wait(l, [&] { return issued > pi || failed; });  // nothing to copy out meanwhile
            if (failed) return;
            can_enqueue = issued > pi;
          }
          if (can_enqueue) {
            const hipError_t q_ = e_ == c_ ? hipEventSynchronize(io.searched[pi]) : hipEventQuery(io.searched[pi]);
            if (q_ == hipErrorNotReady) {
              (void)hipGetLastError();
              can_enqueue = false;
            } else if (q_ != hipSuccess) {
              return bail(q_);
            } else {
              searched_pieces = pi + 1;
            }
          }
        }
      }
      if (can_enqueue) {
        const Chunk& ch = chunks[e_];
        const int slot = (int)(e_ % HostIo::kRing);
        hipError_t he = hipMemcpyAsync(out_pinned ? dst + ch.lo * row_out : io.h_out[slot], io.d_out + ch.lo * row_out,
                                       (size_t)ch.n * row_out, hipMemcpyDeviceToHost, io.down);
        if (he == hipSuccess) he = hipEventRecord(io.down_done[slot], io.down);
        if (he != hipSuccess) return bail(he);
        stamp("down: D2H enqueued", e_);
        ++e_;
        continue;
      }
      // (c_ < e_ here: the search of the next piece is still running, or the ring is full)
      const Chunk& ch = chunks[c_];
      const int slot = (int)(c_ % HostIo::kRing);
      const hipError_t he = hipEventSynchronize(io.down_done[slot]);
      if (he != hipSuccess) return bail(he);
      stamp("down: copy out", c_);
      if (!out_pinned) io.pool->copy(dst + ch.lo * row_out, io.h_out[slot], (size_t)ch.n * row_out);
      stamp("down: copied", c_);
      ++c_;
    }
  });

  // (whatever throws from here on: the download thread is told to stop and joined before the frame goes)
  struct Joiner {
    std::thread& th;
    std::mutex& m;
    std::condition_variable& cv;
    bool& failed;
    ~Joiner() {
      if (!th.joinable()) return;
      {
        std::lock_guard<std::mutex> l(m);
        failed = true;
      }
      cv.notify_all();
      th.join();
    }
  } joiner{downloader, m, cv, failed};
  hipError_t he = hipSuccess;
  // The copy of piece i + 1 into its ring slot runs on the pool while piece i is issued.
  auto start_copy_in = [&](uint64_t i) {
    const uint64_t lo = first[i], n = first[i + 1] - lo;
    if (in_pinned) return std::shared_ptr<CopyPool::Job>();
    return io.pool->start(io.h_in[i % HostIo::kRing], src + lo * row_in, (size_t)n * row_in);
  };
  std::shared_ptr<CopyPool::Job> copy_in = start_copy_in(0);
  for (uint64_t i = 0; i < pieces && he == hipSuccess && rc == PTK_OK; ++i) {
    const uint64_t lo = first[i], n = first[i + 1] - lo;
    const int slot = (int)(i % HostIo::kRing);
    if (copy_in) io.pool->finish(copy_in);
    stamp("up: copied", i);
    if (i + 1 < pieces && !in_pinned) {
      // (slot of piece i + 1: its last upload, of piece i + 1 - kRing, must have left it)
      if (i + 1 >= (uint64_t)HostIo::kRing) he = hipEventSynchronize(io.up_done[(i + 1) % HostIo::kRing]);
      if (he != hipSuccess) break;
      copy_in = start_copy_in(i + 1);
    }
    he = hipMemcpyAsync(io.d_in + lo * row_in, in_pinned ? src + lo * row_in : io.h_in[slot], (size_t)n * row_in,
                        hipMemcpyHostToDevice, io.up);
    if (he == hipSuccess) he = hipEventRecord(io.up_done[slot], io.up);
    hipStream_t ss = io.search[n_search_streams > 1 ? (i & 1) : 0];  // consecutive pieces on two streams: the tail of one overlaps the next
    if (he == hipSuccess) he = hipStreamWaitEvent(ss, io.up_done[slot], 0);
    if (he != hipSuccess) break;
    rc = ptk_search_knn_device(t, d_q + lo * t->dim, n, k, e, d_out + lo * k, ss);
    if (rc != PTK_OK) break;
    he = hipEventRecord(io.searched[i], ss);
    if (he != hipSuccess) break;
    stamp("up: search enqueued", i);
    {
      std::lock_guard<std::mutex> l(m);
      issued = i + 1;
      if (failed) break;
    }
    cv.notify_all();
  }
  if (copy_in) io.pool->finish(copy_in);  // (a piece copied ahead when the loop was left early: the caller's array is still being read)
  {
    std::lock_guard<std::mutex> l(m);
    if (he != hipSuccess || rc != PTK_OK) failed = true;
  }
  cv.notify_all();
  downloader.join();
  // Nothing of this call may be in flight when the caller's arrays (and, on failure, the ring) are touched again.
  hipError_t hs = hipSuccess;
  for (hipStream_t st : {io.up, io.search[0], io.search[1], io.down}) {
    const hipError_t r = hipStreamSynchronize(st);
    if (hs == hipSuccess) hs = r;
  }
  if (rc != PTK_OK) return rc;
  if (he != hipSuccess) return fail(PTK_ERR_DEVICE, "HIP error: %s", hipGetErrorString(he));
  if (!down_error.empty()) return fail(PTK_ERR_DEVICE, "%s", down_error.c_str());
  if (hs != hipSuccess) return fail(PTK_ERR_DEVICE, "HIP error: %s", hipGetErrorString(hs));
  return PTK_OK;
}

// ---- radius -------------------------------------------------------------------------------

static int radius_pass_device(const ptk_tree* t, const float* d_q, uint64_t nq, float radius, float e, bool fill,
                              uint64_t* d_counts, const uint64_t* d_offsets, ptk_neighbor* d_out, int sort,
                              hipStream_t s) {
  int rc = check_search(t, d_q, nq);
  if (rc != PTK_OK) return rc;
  if (!(e > 0.0f)) return fail(PTK_ERR_INVALID, "approximation ratio e must be > 0");
  if (nq == 0) return PTK_OK;
  DeviceGuard guard(t->device);
  if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", t->device);
  const bool nd = t->dim > 3;
  const bool reorder = want_reorder(t, nq);  // Morton order along the first three axes, whatever the dimension
  const int metric = t->metric.load();
  Scratch scratch(t, s);
  Workspace& ws = t->ws;  // locked by `scratch` for the duration of this call
  // A fill pass that repeats the arguments of the last count pass is served from its capture.
  const bool from_capture = fill && ws.cap_valid && ws.cap_q == d_q && ws.cap_nq == nq && ws.cap_radius == radius &&
                            ws.cap_e == e && ws.cap_metric == metric && ws.cap_stream == s;
  if (from_capture) {
    rc = scratch.reserve(nq * 4 + 256);
    if (rc != PTK_OK) return rc;
    uint32_t* over_list = scratch.take<uint32_t>(nq);
    uint32_t* n_over = scratch.take<uint32_t>(1);
    if (over_list == nullptr || n_over == nullptr) return fail(PTK_ERR_NOMEM, "scratch block too small");
    if (ws.cap_lists) {  // (3-D trees: the rows are made from the leaf lists of the count pass)
      PTK_WITH_METRIC((rc = launch_radius_replay<M>(t, d_q, e, ws.cap, d_offsets, reinterpret_cast<ptk::Neighbor*>(d_out),
                                                    over_list, n_over, s)));
      if (rc != PTK_OK) return rc;
    } else {
      Timer timer(t, s);
      PTK_HIP(hipMemsetAsync(n_over, 0, 4, s));
      constexpr int W = 1;  // (one wavefront per block: the LDS of a CU divides evenly)
      const size_t hold = (size_t)W * ptk::kLogScatterLds;  // a staged and a sorted chunk per wavefront
      rc = allow_lds(ptk::radius_log_scatter_kernel<W>, hold);
      if (rc != PTK_OK) return rc;
      hipLaunchKernelGGL((ptk::radius_log_scatter_kernel<W>), dim3((ws.cap.n_static + W - 1) / W), dim3(64 * W), hold, s,
                         ws.cap, d_offsets, reinterpret_cast<ptk::Neighbor*>(d_out), over_list, n_over);
      PTK_HIP(hipGetLastError());
      timer.stop(0, 0);
    }
    // Rows the capture could not hold (possibly none: the blocks then leave at once).
    if (nd) {
      PTK_WITH_METRIC(PTK_WITH_OVF(16, (launch_radius_nd<OVF, M>(t, d_q, nq, radius, e, true, nullptr, d_offsets,
                                                                 reinterpret_cast<ptk::Neighbor*>(d_out), s, over_list,
                                                                 n_over))));
    } else {
      PTK_WITH_METRIC(PTK_WITH_OVF(16, (launch_radius<16, OVF, 64, kGenLeafB, M>(t, d_q, over_list, nq, radius, e, true, nullptr,
                                                                         d_offsets, reinterpret_cast<ptk::Neighbor*>(d_out),
                                                                         s, n_over))));
    }
  } else if (topological(t)) {  // count pass and fill pass both traverse (no capture)
    if (deep_tree(t)) return fail(PTK_ERR_UNSUPPORTED, "tree depth %u is too deep for the device stack", t->max_depth);
    if (!fill) ws.cap_valid = false;
    rc = scratch.reserve(reorder ? permutation_scratch_bytes(nq) : 0);
    if (rc != PTK_OK) return rc;
    uint32_t* perm = nullptr;
    if (reorder) {
      rc = make_permutation(t, d_q, nq, s, scratch, &perm);
      if (rc != PTK_OK) return rc;
    }
    PTK_WITH_OVF(16, (launch_radius_topo<OVF>(t, d_q, perm, nq, radius, e, fill, d_counts, d_offsets,
                                              reinterpret_cast<ptk::Neighbor*>(d_out), s)));
  } else if (deep_tree(t)) {  // record stacks spilling to HBM, a few queries per launch, no capture
    if (!fill) ws.cap_valid = false;
    const DeepPlan plan = deep_plan(t, nq);
    rc = scratch.reserve(plan.bytes());
    if (rc != PTK_OK) return rc;
    ptk::Record* spill = scratch.take<ptk::Record>((size_t)plan.piece * plan.cap);
    if (spill == nullptr) return fail(PTK_ERR_NOMEM, "scratch block too small");
    auto* o = reinterpret_cast<ptk::Neighbor*>(d_out);
    Timer timer(t, s);
    for (uint64_t lo = 0; lo < nq; lo += plan.piece) {
      const uint64_t n = std::min<uint64_t>(plan.piece, nq - lo);
      const uint32_t blocks = (uint32_t)((n + 63) / 64);
      uint64_t* c = fill ? nullptr : d_counts + lo;
      const uint64_t* of = fill ? d_offsets + lo : nullptr;
      if (nd) {
        ptk::DevTreeND dev = t->dev_nd;
        dev.deep_spill = spill;
        dev.deep_cap = plan.cap;
        const size_t smem = (size_t)16 * 64 * 8 + (size_t)t->dim * 64 * 8;
        if (smem > t->lds_per_block)
          return fail(PTK_ERR_UNSUPPORTED, "dimension %u does not fit the LDS staging of the device search", t->dim);
        PTK_WITH_METRIC({
          if (fill) {
            rc = allow_lds(ptk::radius_nd_kernel<16, -1, true, M>, smem);
            if (rc == PTK_OK)
              hipLaunchKernelGGL((ptk::radius_nd_kernel<16, -1, true, M>), dim3(blocks), dim3(64), smem, s, dev,
                                 d_q + lo * t->dim, n, radius, inv_ratio(e), c, of, o, nullptr, nullptr);
          } else {
            rc = allow_lds(ptk::radius_nd_kernel<16, -1, false, M>, smem);
            if (rc == PTK_OK)
              hipLaunchKernelGGL((ptk::radius_nd_kernel<16, -1, false, M>), dim3(blocks), dim3(64), smem, s, dev,
                                 d_q + lo * t->dim, n, radius, inv_ratio(e), c, of, o, nullptr, nullptr);
          }
        });
        if (rc != PTK_OK) return rc;
      } else {
        ptk::DevTree dev = t->dev;
        dev.deep_spill = spill;
        dev.deep_cap = plan.cap;
        PTK_WITH_METRIC({
          if (fill)
            hipLaunchKernelGGL((ptk::radius_kernel<16, -1, 64, 4, true, M>), dim3(blocks), dim3(64), (size_t)16 * 64 * 8, s,
                               dev, d_q + lo * t->dim, t->dim, nullptr, n, radius, inv_ratio(e), c, of, o, nullptr);
          else
            hipLaunchKernelGGL((ptk::radius_kernel<16, -1, 64, 4, false, M>), dim3(blocks), dim3(64), (size_t)16 * 64 * 8, s,
                               dev, d_q + lo * t->dim, t->dim, nullptr, n, radius, inv_ratio(e), c, of, o, nullptr);
        });
      }
      PTK_HIP(hipGetLastError());
    }
    timer.stop(0, fill ? 0 : nq);
  } else {
    const bool capture = !fill && prepare_capture(t, nq, ws);
    const bool lists = capture && !nd && env_int("PTK_RADIUS_LISTS", 1) != 0;
    if (!fill) ws.cap_valid = false;
    rc = scratch.reserve(reorder ? permutation_scratch_bytes(nq) : 0);
    if (rc != PTK_OK) return rc;
    uint32_t* perm = nullptr;
    if (reorder) {  // (a query costs what it finds: the densest cells to the front of the launch)
      rc = make_permutation(t, d_q, nq, s, scratch, &perm, nd ? 0u : ptk::kCellsDenseFirst);
      if (rc != PTK_OK) return rc;
    }
    if (capture) {
      if (nd) {
        PTK_WITH_METRIC(PTK_WITH_OVF(16, (launch_radius_nd_capture<OVF, M>(t, d_q, perm, nq, radius, e, d_counts, ws.cap, s))));
      } else if (lists) {  // the count pass lists the leaves with hits for the fill pass
        PTK_WITH_METRIC(PTK_WITH_OVF(kGenRing, (launch_radius_list<kGenRing, OVF, kGenLeafB, M>(t, d_q, perm, nq, radius, e, d_counts,
                                                                                ws.cap, s))));
      } else {
        PTK_WITH_METRIC(PTK_WITH_OVF(kGenRing, (launch_radius_capture<kGenRing, OVF, 64, kGenLeafB, M>(t, d_q, perm, nq, radius, e, d_counts,
                                                                                   ws.cap, s))));
      }
      if (rc == PTK_OK) {
        ws.cap_valid = true;
        ws.cap_lists = lists;
        ws.cap_q = d_q;
        ws.cap_nq = nq;
        ws.cap_radius = radius;
        ws.cap_e = e;
        ws.cap_metric = metric;
        ws.cap_stream = s;
      }
    } else if (nd) {
      PTK_WITH_METRIC(PTK_WITH_OVF(16, (launch_radius_nd<OVF, M>(t, d_q, nq, radius, e, fill, d_counts, d_offsets,
                                                                 reinterpret_cast<ptk::Neighbor*>(d_out), s, perm))));
    } else {
      PTK_WITH_METRIC(PTK_WITH_OVF(16, (launch_radius<16, OVF, 64, kGenLeafB, M>(t, d_q, perm, nq, radius, e, fill, d_counts,
                                                                         d_offsets, reinterpret_cast<ptk::Neighbor*>(d_out), s))));
    }
  }
  if (rc == PTK_OK && fill && sort) {
    Timer timer(t, s);
    const uint32_t blocks = (uint32_t)((nq + ptk::kBlock - 1) / ptk::kBlock);
    hipLaunchKernelGGL(ptk::sort_rows_kernel, dim3(blocks), dim3(ptk::kBlock), 0, s, nq, d_offsets,
                       reinterpret_cast<ptk::Neighbor*>(d_out));
    PTK_HIP(hipGetLastError());
    timer.stop(2, 0);
  }
  return rc;
}

int ptk_search_radius_count_device(const ptk_tree* t, const float* d_q, uint64_t nq, float radius, float e,
                                   uint64_t* d_counts, void* stream) {
  if (nq > 0 && d_counts == nullptr) return fail(PTK_ERR_INVALID, "null counts buffer");
  return radius_pass_device(t, d_q, nq, radius, e, false, d_counts, nullptr, nullptr, 0,
                            static_cast<hipStream_t>(stream));
}

int ptk_search_radius_fill_device(const ptk_tree* t, const float* d_q, uint64_t nq, float radius, float e,
                                  const uint64_t* d_offsets, ptk_neighbor* d_out, int sort, void* stream) {
  if (nq > 0 && d_offsets == nullptr) return fail(PTK_ERR_INVALID, "null offsets buffer");
  return radius_pass_device(t, d_q, nq, radius, e, true, nullptr, d_offsets, d_out, sort,
                            static_cast<hipStream_t>(stream));
}

// The capture of a radius count pass is keyed on the device address of the query batch.  The host
// forms below own that buffer for one call only: once it is freed the address may be handed out
// again for ANOTHER batch, so the capture must not outlive the call.
static void drop_radius_capture(const ptk_tree* t) {
  std::lock_guard<std::mutex> lock(t->ws.mutex);
  t->ws.cap_valid = false;
}

int ptk_search_radius_count(const ptk_tree* t, const float* q, uint64_t nq, float radius, float e,
                            uint64_t* counts) {
  int rc = check_search(t, q, nq);
  if (rc != PTK_OK) return rc;
  if (nq == 0) return PTK_OK;
  if (counts == nullptr) return fail(PTK_ERR_INVALID, "null counts buffer");
  DeviceGuard guard(t->device);
  if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", t->device);
  float* d_q = nullptr;
  uint64_t* d_c = nullptr;
  const size_t qbytes = (size_t)nq * t->dim * sizeof(float);
  hipError_t he = hipMalloc((void**)&d_q, qbytes);
  if (he == hipSuccess) he = hipMalloc((void**)&d_c, nq * 8);
  if (he == hipSuccess) he = hipMemcpy(d_q, q, qbytes, hipMemcpyHostToDevice);
  if (he == hipSuccess) {
    rc = ptk_search_radius_count_device(t, d_q, nq, radius, e, d_c, nullptr);
    if (rc == PTK_OK) he = hipMemcpy(counts, d_c, nq * 8, hipMemcpyDeviceToHost);
  }
  drop_radius_capture(t);
  if (d_q) (void)hipFree(d_q);
  if (d_c) (void)hipFree(d_c);
  if (rc != PTK_OK) return rc;
  if (he != hipSuccess) return fail(PTK_ERR_DEVICE, "HIP error: %s", hipGetErrorString(he));
  return PTK_OK;
}

int ptk_search_radius_fill(const ptk_tree* t, const float* q, uint64_t nq, float radius, float e,
                           const uint64_t* offsets, ptk_neighbor* out, int sort) {
  int rc = check_search(t, q, nq);
  if (rc != PTK_OK) return rc;
  if (nq == 0) return PTK_OK;
  if (offsets == nullptr) return fail(PTK_ERR_INVALID, "null offsets buffer");
  const uint64_t total = offsets[nq];
  if (total > 0 && out == nullptr) return fail(PTK_ERR_INVALID, "null output buffer");
  DeviceGuard guard(t->device);
  if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", t->device);
  float* d_q = nullptr;
  uint64_t* d_o = nullptr;
  ptk_neighbor* d_out = nullptr;
  const size_t qbytes = (size_t)nq * t->dim * sizeof(float);
  hipError_t he = hipMalloc((void**)&d_q, qbytes);
  if (he == hipSuccess) he = hipMalloc((void**)&d_o, (nq + 1) * 8);
  if (he == hipSuccess) he = hipMalloc((void**)&d_out, std::max<uint64_t>(total, 1) * 8);
  if (he == hipSuccess) he = hipMemcpy(d_q, q, qbytes, hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(d_o, offsets, (nq + 1) * 8, hipMemcpyHostToDevice);
  if (he == hipSuccess) {
    drop_radius_capture(t);  // d_q is this call's own copy: whatever was captured belongs to another buffer
    rc = ptk_search_radius_fill_device(t, d_q, nq, radius, e, d_o, d_out, sort, nullptr);
    if (rc == PTK_OK && total > 0) he = hipMemcpy(out, d_out, total * 8, hipMemcpyDeviceToHost);
  }
  if (d_q) (void)hipFree(d_q);
  if (d_o) (void)hipFree(d_o);
  if (d_out) (void)hipFree(d_out);
  if (rc != PTK_OK) return rc;
  if (he != hipSuccess) return fail(PTK_ERR_DEVICE, "HIP error: %s", hipGetErrorString(he));
  return PTK_OK;
}

int ptk_search_radius(const ptk_tree* t, const float* q, uint64_t nq, float radius, float e, int sort,
                      uint64_t* offsets, ptk_neighbor** out) {
  if (out == nullptr || offsets == nullptr) return fail(PTK_ERR_INVALID, "null output pointer");
  *out = nullptr;
  int rc = check_search(t, q, nq);
  if (rc != PTK_OK) return rc;
  offsets[0] = 0;
  if (nq == 0) return PTK_OK;
  DeviceGuard guard(t->device);
  if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", t->device);
  float* d_q = nullptr;
  uint64_t *d_c = nullptr, *d_o = nullptr;
  ptk_neighbor* d_out = nullptr;
  void* tmp = nullptr;
  const size_t qbytes = (size_t)nq * t->dim * sizeof(float);
  hipError_t he = hipMalloc((void**)&d_q, qbytes);
  if (he == hipSuccess) he = hipMalloc((void**)&d_c, (nq + 1) * 8);
  if (he == hipSuccess) he = hipMalloc((void**)&d_o, (nq + 1) * 8);
  if (he == hipSuccess) he = hipMemset(d_c, 0, (nq + 1) * 8);
  if (he == hipSuccess) he = hipMemcpy(d_q, q, qbytes, hipMemcpyHostToDevice);
  uint64_t total = 0;
  if (he == hipSuccess) {
    rc = ptk_search_radius_count_device(t, d_q, nq, radius, e, d_c, nullptr);
    if (rc == PTK_OK) {
      size_t tmp_bytes = 0;
      he = rocprim::exclusive_scan(nullptr, tmp_bytes, d_c, d_o, (uint64_t)0, nq + 1, rocprim::plus<uint64_t>(),
                                   (hipStream_t) nullptr);
      if (he == hipSuccess) he = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16);
      if (he == hipSuccess)
        he = rocprim::exclusive_scan(tmp, tmp_bytes, d_c, d_o, (uint64_t)0, nq + 1, rocprim::plus<uint64_t>(),
                                     (hipStream_t) nullptr);
      if (he == hipSuccess) he = hipMemcpy(offsets, d_o, (nq + 1) * 8, hipMemcpyDeviceToHost);
      if (he == hipSuccess) {
        total = offsets[nq];
        he = hipMalloc((void**)&d_out, std::max<uint64_t>(total, 1) * 8);
      }
      if (he == hipSuccess) rc = ptk_search_radius_fill_device(t, d_q, nq, radius, e, d_o, d_out, sort, nullptr);
      if (he == hipSuccess && rc == PTK_OK) {
        *out = static_cast<ptk_neighbor*>(std::malloc(std::max<uint64_t>(total, 1) * 8));
        if (*out == nullptr) {
          rc = fail(PTK_ERR_NOMEM, "out of memory");
        } else if (total > 0) {
          he = hipMemcpy(*out, d_out, total * 8, hipMemcpyDeviceToHost);
        }
      }
    }
  }
  drop_radius_capture(t);
  if (tmp) (void)hipFree(tmp);
  if (d_q) (void)hipFree(d_q);
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

// One pass of the box search over device buffers: count (fill == false: d_counts[i] = hits of box i)
// or fill (d_offsets = exclusive scan of the counts; row i of d_out in reference traversal order).
static int box_pass_device(const ptk_tree* t, const float* d_mn, const float* d_mx, uint64_t nb, bool fill,
                           uint64_t* d_counts, const uint64_t* d_offsets, int32_t* d_out, hipStream_t s) {
  int rc = check_search(t, d_mn, nb);
  if (rc != PTK_OK) return rc;
  if (nb > 0 && d_mx == nullptr) return fail(PTK_ERR_INVALID, "null box buffer");
  const size_t nd_smem = (size_t)16 * 64 * 8 + (size_t)4 * t->dim * 64 * 4;
  if (t->dim > 3 && nd_smem > t->lds_per_block)
    return fail(PTK_ERR_UNSUPPORTED, "dimension %u does not fit the LDS staging of the device box search", t->dim);
  if (nb == 0) return PTK_OK;
  if (nb >= (1ull << 32)) return fail(PTK_ERR_UNSUPPORTED, "batches of 2^32 or more boxes are not supported");
  DeviceGuard guard(t->device);
  if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", t->device);
  ptk::BoxState root{0, 0, 0, 0, 0, 0};
  {
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (uint32_t d = 0; d < t->dim && d < 3; ++d) {  // dim > 3 hands its root box over in scratch
      mn[d] = t->root_min[d];
      mx[d] = t->root_max[d];
    }
    root = ptk::BoxState{mn[0], mn[1], mn[2], mx[0], mx[1], mx[2]};
  }
  // A topological tree: circle axes (metric_so2: axis 0; metric_se2_squared: axis 2), the four-bound tests.
  const bool topo = topological(t);
  const uint32_t s1_mask = !topo ? 0u : (t->metric.load() == PTK_METRIC_SO2 ? 1u : 4u);
  if (topo && (deep_tree(t) || t->dev.outer == nullptr))
    return fail(PTK_ERR_UNSUPPORTED, "the box search of this topological tree runs on the host members (kd_tree::search_box)");
  // Boxes in Morton order of their min corners (launch order only; rows stay in the caller's order).
  const bool deep = deep_tree(t);
  const bool reorder = !deep && want_reorder(t, nb);
  const DeepPlan plan = deep ? deep_plan(t, nb) : DeepPlan{0, 0};
  Scratch scratch(t, s);
  rc = scratch.reserve((reorder ? permutation_scratch_bytes(nb) : 0) + (size_t)2 * t->dim * sizeof(float) + 512 +
                       (deep ? plan.bytes() : 0));
  if (rc != PTK_OK) return rc;
  float* d_root = nullptr;
  if (t->dim > 3) {  // the root box of the any-dimension kernel: min[dim], max[dim]
    d_root = scratch.take<float>((size_t)2 * t->dim);
    if (d_root == nullptr) return fail(PTK_ERR_NOMEM, "scratch block too small");
    PTK_HIP(hipMemcpyAsync(d_root, t->root_min.data(), t->dim * sizeof(float), hipMemcpyHostToDevice, s));
    PTK_HIP(hipMemcpyAsync(d_root + t->dim, t->root_max.data(), t->dim * sizeof(float), hipMemcpyHostToDevice, s));
  }
  uint32_t* perm = nullptr;
  if (reorder) {
    rc = make_permutation(t, d_mn, nb, s, scratch, &perm);
    if (rc != PTK_OK) return rc;
  }
  const uint32_t blocks = (uint32_t)((nb + 63) / 64);
  const auto* ranges = static_cast<const uint2*>(t->d_ranges);
  Timer timer(t, s);
  if (deep) {  // record stacks spilling to HBM, a few boxes per launch
    ptk::Record* spill = scratch.take<ptk::Record>((size_t)plan.piece * plan.cap);
    if (spill == nullptr) return fail(PTK_ERR_NOMEM, "scratch block too small");
    for (uint64_t lo = 0; lo < nb; lo += plan.piece) {
      const uint64_t n = std::min<uint64_t>(plan.piece, nb - lo);
      const uint32_t pb = (uint32_t)((n + 63) / 64);
      uint64_t* c = fill ? nullptr : d_counts + lo;
      const uint64_t* of = fill ? d_offsets + lo : nullptr;
      if (t->dim > 3) {
        ptk::DevTreeND dev = t->dev_nd;
        dev.deep_spill = spill;
        dev.deep_cap = plan.cap;
        if (fill) {
          rc = allow_lds(ptk::box_nd_kernel<16, -1, true>, nd_smem);
          if (rc != PTK_OK) return rc;
          hipLaunchKernelGGL((ptk::box_nd_kernel<16, -1, true>), dim3(pb), dim3(64), nd_smem, s, dev, ranges, d_root,
                             d_mn + lo * t->dim, d_mx + lo * t->dim, n, c, of, d_out, nullptr);
        } else {
          rc = allow_lds(ptk::box_nd_kernel<16, -1, false>, nd_smem);
          if (rc != PTK_OK) return rc;
          hipLaunchKernelGGL((ptk::box_nd_kernel<16, -1, false>), dim3(pb), dim3(64), nd_smem, s, dev, ranges, d_root,
                             d_mn + lo * t->dim, d_mx + lo * t->dim, n, c, of, d_out, nullptr);
        }
      } else {
        ptk::DevTree dev = t->dev;
        dev.deep_spill = spill;
        dev.deep_cap = plan.cap;
        if (fill)
          hipLaunchKernelGGL((ptk::box_kernel<16, -1, true>), dim3(pb), dim3(64), 16 * 64 * 8, s, dev, ranges, root,
                             d_mn + lo * t->dim, d_mx + lo * t->dim, t->dim, n, c, of, d_out, nullptr);
        else
          hipLaunchKernelGGL((ptk::box_kernel<16, -1, false>), dim3(pb), dim3(64), 16 * 64 * 8, s, dev, ranges, root,
                             d_mn + lo * t->dim, d_mx + lo * t->dim, t->dim, n, c, of, d_out, nullptr);
      }
      PTK_HIP(hipGetLastError());
    }
  } else if (!fill) {
    PTK_WITH_OVF(16, ([&]() -> int {
                   if (t->dim > 3) {
                     int lrc = allow_lds(ptk::box_nd_kernel<16, OVF, false>, nd_smem);
                     if (lrc != PTK_OK) return lrc;
                     hipLaunchKernelGGL((ptk::box_nd_kernel<16, OVF, false>), dim3(blocks), dim3(64), nd_smem, s, t->dev_nd,
                                        ranges, d_root, d_mn, d_mx, nb, d_counts, nullptr, nullptr, perm);
                     return PTK_OK;
                   }
                   if (topo)
                     hipLaunchKernelGGL((ptk::box_kernel<16, OVF, false, true>), dim3(blocks), dim3(64), 16 * 64 * 8, s, t->dev,
                                        ranges, root, d_mn, d_mx, t->dim, nb, d_counts, nullptr, nullptr, perm, s1_mask);
                   else
                     hipLaunchKernelGGL((ptk::box_kernel<16, OVF, false>), dim3(blocks), dim3(64), 16 * 64 * 8, s, t->dev,
                                        ranges, root, d_mn, d_mx, t->dim, nb, d_counts, nullptr, nullptr, perm, 0u);
                   return PTK_OK;
                 }()));
  } else {
    PTK_WITH_OVF(16, ([&]() -> int {
                   if (t->dim > 3) {
                     int lrc = allow_lds(ptk::box_nd_kernel<16, OVF, true>, nd_smem);
                     if (lrc != PTK_OK) return lrc;
                     hipLaunchKernelGGL((ptk::box_nd_kernel<16, OVF, true>), dim3(blocks), dim3(64), nd_smem, s, t->dev_nd,
                                        ranges, d_root, d_mn, d_mx, nb, nullptr, d_offsets, d_out, perm);
                     return PTK_OK;
                   }
                   if (topo)
                     hipLaunchKernelGGL((ptk::box_kernel<16, OVF, true, true>), dim3(blocks), dim3(64), 16 * 64 * 8, s, t->dev,
                                        ranges, root, d_mn, d_mx, t->dim, nb, nullptr, d_offsets, d_out, perm, s1_mask);
                   else
                     hipLaunchKernelGGL((ptk::box_kernel<16, OVF, true>), dim3(blocks), dim3(64), 16 * 64 * 8, s, t->dev,
                                        ranges, root, d_mn, d_mx, t->dim, nb, nullptr, d_offsets, d_out, perm, 0u);
                   return PTK_OK;
                 }()));
  }
  if (rc != PTK_OK) return rc;
  PTK_HIP(hipGetLastError());
  timer.stop(fill ? 3 : 0, fill ? 0 : nb);
  return PTK_OK;
}

int ptk_search_box_count_device(const ptk_tree* t, const float* d_mins, const float* d_maxs, uint64_t nb,
                                uint64_t* d_counts, void* stream) {
  if (nb > 0 && d_counts == nullptr) return fail(PTK_ERR_INVALID, "null counts buffer");
  return box_pass_device(t, d_mins, d_maxs, nb, false, d_counts, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

int ptk_search_box_fill_device(const ptk_tree* t, const float* d_mins, const float* d_maxs, uint64_t nb,
                               const uint64_t* d_offsets, int32_t* d_out, void* stream) {
  if (nb > 0 && (d_offsets == nullptr || d_out == nullptr)) return fail(PTK_ERR_INVALID, "null offsets / output buffer");
  return box_pass_device(t, d_mins, d_maxs, nb, true, nullptr, d_offsets, d_out, static_cast<hipStream_t>(stream));
}

int ptk_search_box(const ptk_tree* t, const float* mins, const float* maxs, uint64_t nb, uint64_t* offsets,
                   int32_t** out) {
  if (out == nullptr || offsets == nullptr) return fail(PTK_ERR_INVALID, "null output pointer");
  *out = nullptr;
  int rc = check_search(t, mins, nb);
  if (rc != PTK_OK) return rc;
  if (nb > 0 && maxs == nullptr) return fail(PTK_ERR_INVALID, "null box buffer");
  offsets[0] = 0;
  if (nb == 0) return PTK_OK;
  DeviceGuard guard(t->device);
  if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", t->device);
  float *d_mn = nullptr, *d_mx = nullptr;
  uint64_t *d_c = nullptr, *d_o = nullptr;
  int32_t* d_out = nullptr;
  void* tmp = nullptr;
  const size_t bbytes = (size_t)nb * t->dim * sizeof(float);
  hipError_t he = hipMalloc((void**)&d_mn, bbytes);
  if (he == hipSuccess) he = hipMalloc((void**)&d_mx, bbytes);
  if (he == hipSuccess) he = hipMalloc((void**)&d_c, (nb + 1) * 8);
  if (he == hipSuccess) he = hipMalloc((void**)&d_o, (nb + 1) * 8);
  if (he == hipSuccess) he = hipMemset(d_c, 0, (nb + 1) * 8);
  if (he == hipSuccess) he = hipMemcpy(d_mn, mins, bbytes, hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(d_mx, maxs, bbytes, hipMemcpyHostToDevice);
  uint64_t total = 0;
  if (he == hipSuccess) {
    rc = ptk_search_box_count_device(t, d_mn, d_mx, nb, d_c, nullptr);
    if (rc == PTK_OK) {
      size_t tmp_bytes = 0;
      he = rocprim::exclusive_scan(nullptr, tmp_bytes, d_c, d_o, (uint64_t)0, nb + 1, rocprim::plus<uint64_t>(),
                                   (hipStream_t) nullptr);
      if (he == hipSuccess) he = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16);
      if (he == hipSuccess)
        he = rocprim::exclusive_scan(tmp, tmp_bytes, d_c, d_o, (uint64_t)0, nb + 1, rocprim::plus<uint64_t>(),
                                     (hipStream_t) nullptr);
      if (he == hipSuccess) he = hipMemcpy(offsets, d_o, (nb + 1) * 8, hipMemcpyDeviceToHost);
      if (he == hipSuccess) {
        total = offsets[nb];
        he = hipMalloc((void**)&d_out, std::max<uint64_t>(total, 1) * 4);
      }
      if (he == hipSuccess) rc = ptk_search_box_fill_device(t, d_mn, d_mx, nb, d_o, d_out, nullptr);
      if (he == hipSuccess && rc == PTK_OK) {
        *out = static_cast<int32_t*>(std::malloc(std::max<uint64_t>(total, 1) * 4));
        if (*out == nullptr) {
          rc = fail(PTK_ERR_NOMEM, "out of memory");
        } else if (total > 0) {
          he = hipMemcpy(*out, d_out, total * 4, hipMemcpyDeviceToHost);
        }
      }
    }
  }
  if (tmp) (void)hipFree(tmp);
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

void ptk_free(void* p) { std::free(p); }

int ptk_host_alloc(uint64_t bytes, void** out) {
  if (out == nullptr) return fail(PTK_ERR_INVALID, "null out pointer");
  *out = nullptr;
  if (bytes == 0) return PTK_OK;
  if (hipHostMalloc(out, (size_t)bytes, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    *out = nullptr;
    return fail(PTK_ERR_NOMEM, "out of pinned host memory (%llu bytes)", (unsigned long long)bytes);
  }
  return PTK_OK;
}

void ptk_host_free(void* p) {
  if (p != nullptr && hipHostFree(p) != hipSuccess) (void)hipGetLastError();
}

int ptk_host_register(void* p, uint64_t bytes) {
  if (p == nullptr || bytes == 0) return fail(PTK_ERR_INVALID, "null or empty range");
  if (hipHostRegister(p, (size_t)bytes, hipHostRegisterPortable) != hipSuccess) {
    (void)hipGetLastError();
    return fail(PTK_ERR_NOMEM, "cannot page-lock %llu bytes at %p", (unsigned long long)bytes, p);
  }
  return PTK_OK;
}

int ptk_host_unregister(void* p) {
  if (p == nullptr) return fail(PTK_ERR_INVALID, "null pointer");
  if (hipHostUnregister(p) != hipSuccess) {
    (void)hipGetLastError();
    return fail(PTK_ERR_INVALID, "%p is not a registered range", p);
  }
  return PTK_OK;
}

int ptk_debug_knn1_counts(const ptk_tree* t, uint32_t counts[4]) {
  if (t == nullptr || counts == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  if (t->device < 0) return fail(PTK_ERR_DEVICE, "this handle has no device replica");
  DeviceGuard guard(t->device);
  // The scratch block of the LAST search on the handle's main workspace or on one of its per-stream ones (a k-NN call
  // on a stream of its own uses those): the first that still holds the counters of a two-phase search.
  Workspace* holder = nullptr;
  for (int i = 0; i <= ptk_tree::kExtraWs && holder == nullptr; ++i) {
    Workspace& w = i == 0 ? t->ws : t->extra_ws[i - 1];
    std::lock_guard<std::mutex> lock(w.mutex);
    if (w.last_meta != nullptr) holder = &w;
  }
  if (holder == nullptr) return fail(PTK_ERR_INVALID, "the last search of this handle was not a two-phase k = 1 search");
  std::lock_guard<std::mutex> lock(holder->mutex);
  if (holder->last_meta == nullptr) return fail(PTK_ERR_INVALID, "the last search of this handle was not a two-phase k = 1 search");
  uint32_t meta[ptk::kMetaWords];
  PTK_HIP(hipDeviceSynchronize());
  PTK_HIP(hipMemcpy(meta, holder->last_meta, sizeof(meta), hipMemcpyDeviceToHost));
  counts[0] = meta[0];
  counts[1] = meta[ptk::kMetaHeavy];
  counts[2] = meta[ptk::kMetaRedo];
  counts[3] = meta[1];
  return PTK_OK;
}

int ptk_debug_knn_coop_counts(const ptk_tree* t, uint32_t counts[7]) {
  if (t == nullptr || counts == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  if (t->device < 0) return fail(PTK_ERR_DEVICE, "this handle has no device replica");
  DeviceGuard guard(t->device);
  Workspace* holder = nullptr;
  for (int i = 0; i <= ptk_tree::kExtraWs && holder == nullptr; ++i) {
    Workspace& w = i == 0 ? t->ws : t->extra_ws[i - 1];
    std::lock_guard<std::mutex> lock(w.mutex);
    if (w.last_meta != nullptr) holder = &w;
  }
  if (holder == nullptr) return fail(PTK_ERR_INVALID, "the last search of this handle left no counters");
  std::lock_guard<std::mutex> lock(holder->mutex);
  if (holder->last_meta == nullptr) return fail(PTK_ERR_INVALID, "the last search of this handle left no counters");
  uint32_t meta[ptk::kMetaWords];
  PTK_HIP(hipDeviceSynchronize());
  PTK_HIP(hipMemcpy(meta, holder->last_meta, sizeof(meta), hipMemcpyDeviceToHost));
  counts[0] = meta[ptk::kMetaHeavy] + meta[ptk::kMetaHeavyRest];
  counts[1] = meta[ptk::kMetaRedo];
  counts[2] = meta[ptk::kKnnWhyPool];
  counts[3] = meta[ptk::kKnnWhyTie];
  counts[4] = meta[ptk::kKnnWhyBox];
  counts[5] = meta[ptk::kKnnWhyRange];
  counts[6] = meta[ptk::kKnnTieSweeps];
  return PTK_OK;
}

int ptk_debug_knn_cap(uint64_t nq, uint32_t k, float e, uint32_t* cap, uint64_t* list_entries) {
  if (cap == nullptr || list_entries == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  *cap = (k > 1 && k <= 64) ? knn_cap(e, nq, k) : 0u;
  *list_entries = *cap != 0u ? knn_max_handover(nq) : 0;
  return PTK_OK;
}

int ptk_debug_piles(const ptk_tree* t, uint64_t out[3]) {
  if (t == nullptr || out == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  if (t->device < 0) return fail(PTK_ERR_DEVICE, "this handle has no device replica");
  out[0] = t->n_piles;
  out[1] = t->pile_points;
  out[2] = knn1_depth(t);
  return PTK_OK;
}

int ptk_debug_create_phases(const ptk_tree* t, double ms[3]) {
  if (t == nullptr || ms == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  for (int i = 0; i < 3; ++i) ms[i] = t->create_ms[i];
  return PTK_OK;
}

#if defined(PTK_WAVE_TRACE)
// Experiment builds only (tools/wave_trace.py): where the general kernels leave {start, end, hardware id, cycles} of
// every wavefront (device memory, 32 bytes per block of the next launch; null = off).  Not declared in ptk.h.
int ptk_debug_wave_trace(void* d_trace) {
  unsigned long long* p = static_cast<unsigned long long*>(d_trace);
  PTK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(ptk::g_wave_trace), &p, sizeof(p)));
  return PTK_OK;
}
int ptk_debug_wave_trace_select(int kernel) {  // which of the k = 1 kernels records (ptk_kernels.hpp, PTK_TRACE_BEGIN_SEL)
  PTK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(ptk::g_wave_trace_sel), &kernel, sizeof(kernel)));
  return PTK_OK;
}
#endif

int ptk_debug_batch_order(const ptk_tree* t, int* how) {
  if (t == nullptr || how == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lock(t->ws.mutex);
  *how = t->ws.last_order;
  if (t->ws.last_verdict != nullptr && t->device >= 0) {  // sampled: what it was found to be is on the device
    DeviceGuard guard(t->device);
    uint32_t coherent = 0;
    PTK_HIP(hipDeviceSynchronize());
    PTK_HIP(hipMemcpy(&coherent, t->ws.last_verdict, 4, hipMemcpyDeviceToHost));
    if (coherent != 0u) *how = 2;
  }
  return PTK_OK;
}

int ptk_debug_batch_permutation(const ptk_tree* t, const float* d_q, uint64_t nq, uint32_t* d_perm) {
  int rc = check_search(t, d_q, nq);
  if (rc != PTK_OK) return rc;
  if (d_perm == nullptr || nq == 0) return fail(PTK_ERR_INVALID, "null permutation buffer or empty batch");
  DeviceGuard guard(t->device);
  if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", t->device);
  Scratch scratch(t, nullptr);
  rc = scratch.reserve(permutation_scratch_bytes(nq));
  if (rc != PTK_OK) return rc;
  uint32_t* perm = nullptr;
  rc = make_permutation(t, d_q, nq, nullptr, scratch, &perm);
  if (rc != PTK_OK) return rc;
  PTK_HIP(hipMemcpyAsync(d_perm, perm, nq * sizeof(uint32_t), hipMemcpyDeviceToDevice, nullptr));
  PTK_HIP(hipStreamSynchronize(nullptr));
  return PTK_OK;
}

int ptk_debug_key_bits(const ptk_tree* t, uint64_t nq, uint32_t bits[3]) {
  if (t == nullptr || bits == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  axis_bits(t, morton_bits(nq), bits);
  return PTK_OK;
}

int ptk_profile_enable(ptk_tree* t, int on) {
  if (t == nullptr) return fail(PTK_ERR_INVALID, "null tree");
  std::lock_guard<std::mutex> lock(t->profile.mutex);
  t->profile.enabled = on != 0;
  return PTK_OK;
}

int ptk_profile_get(const ptk_tree* t, ptk_profile* out, int reset) {
  return ptk_profile_get_sized(t, out, sizeof(ptk_profile), reset);
}

int ptk_profile_get_sized(const ptk_tree* t, void* out_bytes, uint64_t size, int reset) {
  if (t == nullptr || out_bytes == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lock(t->profile.mutex);
  for (PendingEvent& p : t->profile.pending) {
    float ms = 0;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      if (p.kind == 0) {
        t->profile.acc.search_ms += ms;
        t->profile.acc.launches += 1;
        t->profile.acc.queries += p.queries;
      } else if (p.kind == 3) {  // second traversal kernel of the same search
        t->profile.acc.search_ms += ms;
        t->profile.acc.search_tail_ms += ms;
      } else if (p.kind == 1) {
        t->profile.acc.reorder_ms += ms;
      } else {
        t->profile.acc.other_ms += ms;
      }
    }
    t->profile.idle.push_back(p.a);
    if (!p.keep_b) t->profile.idle.push_back(p.b);
  }
  t->profile.pending.clear();
  std::memcpy(out_bytes, &t->profile.acc, (size_t)std::min<uint64_t>(size, sizeof(ptk_profile)));
  if (reset) t->profile.acc = ptk_profile{};
  return PTK_OK;
}


// ---- kd-forest -----------------------------------------------------------------------------

}  // extern "C"

struct ptk_forest {
  uint32_t dim = 0;
  uint64_t n_points = 0;
  uint32_t n_trees = 0;
  int device = kDeviceNone;
  std::vector<float> rotations;       // n_trees x dim
  std::vector<void*> allocations;     // everything to hipFree
  ptk::ForestDev dev{};
  uint32_t* d_dropped = nullptr;
};

namespace {

template <class T>
int forest_upload(ptk_forest* f, const std::vector<T>& host, const T** dev) {
  void* p = nullptr;
  PTK_HIP(hipMalloc(&p, std::max<size_t>(host.size(), 1) * sizeof(T)));
  f->allocations.push_back(p);
  if (!host.empty()) PTK_HIP(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  *dev = static_cast<const T*>(p);
  return PTK_OK;
}

int forest_build(ptk_forest* f, const float* points, uint64_t max_leaf_size, uint64_t seed) {
  const uint64_t n = f->n_points;
  const uint32_t dim = f->dim;
  std::vector<ptk::ForestTreeDev> trees(f->n_trees);
  std::vector<float> rotated;
  for (uint32_t ti = 0; ti < f->n_trees; ++ti) {
    float* r = f->rotations.data() + (size_t)ti * dim;
    ptk::reflection_vector(seed, ti, dim, r);
    ptk::ForestTreeHost host;
    std::string err = ptk::build_forest_tree(points, n, dim, max_leaf_size, r, rotated, host, build_threads());
    if (!err.empty()) return fail(PTK_ERR_UNSUPPORTED, "tree %u: %s", ti, err.c_str());
    std::vector<float> rot(r, r + dim);
    int rc = forest_upload(f, host.nodes, &trees[ti].nodes);
    if (rc == PTK_OK) rc = forest_upload(f, host.indices, &trees[ti].indices);
    if (rc == PTK_OK) rc = forest_upload(f, rot, &trees[ti].rotation);
    if (rc != PTK_OK) return rc;
    trees[ti].root_ref = host.root_ref;
    trees[ti].cbits = host.cbits;
    trees[ti].cmask = (1u << host.cbits) - 1u;
    trees[ti].pad = 0;
  }
  std::vector<float> pts(points, points + n * dim);
  int rc = forest_upload(f, pts, &f->dev.points);
  if (rc == PTK_OK) rc = forest_upload(f, trees, &f->dev.trees);
  if (rc != PTK_OK) return rc;
  void* d = nullptr;
  PTK_HIP(hipMalloc(&d, 4));
  f->allocations.push_back(d);
  PTK_HIP(hipMemset(d, 0, 4));
  f->d_dropped = static_cast<uint32_t*>(d);
  f->dev.n_trees = f->n_trees;
  f->dev.dim = dim;
  return PTK_OK;
}

}  // namespace

extern "C" {

int ptk_forest_create(const float* points, uint64_t n_points, uint32_t dim, uint64_t max_leaf_size,
                      uint32_t forest_size, uint64_t seed, int32_t device, ptk_forest** out) {
  if (out == nullptr) return fail(PTK_ERR_INVALID, "null out pointer");
  *out = nullptr;
  if (points == nullptr) return fail(PTK_ERR_INVALID, "null points");
  if (dim == 0 || n_points == 0 || max_leaf_size == 0 || forest_size == 0)
    return fail(PTK_ERR_INVALID, "dim, n_points, max_leaf_size and forest_size must be positive");
  if (n_points >= (1ull << 31)) return fail(PTK_ERR_INVALID, "n_points must be < 2^31");
  const size_t lds = ((size_t)2 * dim + 2 * ptk::kForestQueue + 2 * ptk::kForestPath) * 4;
  if (lds > 64 * 1024) return fail(PTK_ERR_UNSUPPORTED, "dimension %u does not fit the forest kernel's LDS", dim);
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(PTK_ERR_DEVICE, "no HIP device is visible");
  int dev = device;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev >= count) return fail(PTK_ERR_INVALID, "device %d out of range (%d visible)", dev, count);
  ptk_forest* f = new (std::nothrow) ptk_forest;
  if (f == nullptr) return fail(PTK_ERR_NOMEM, "out of memory");
  f->dim = dim;
  f->n_points = n_points;
  f->n_trees = forest_size;
  f->device = dev;
  int rc = PTK_OK;
  try {
    f->rotations.resize((size_t)forest_size * dim);
    DeviceGuard guard(dev);
    rc = guard.ok ? forest_build(f, points, max_leaf_size, seed) : fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", dev);
  } catch (const std::bad_alloc&) {
    rc = fail(PTK_ERR_NOMEM, "out of memory");
  } catch (const std::length_error& err) {
    rc = fail(PTK_ERR_UNSUPPORTED, "%s", err.what());
  }
  if (rc != PTK_OK) {
    ptk_forest_destroy(f);
    return rc;
  }
  *out = f;
  return PTK_OK;
}

void ptk_forest_destroy(ptk_forest* f) {
  if (f == nullptr) return;
  if (f->device >= 0) {
    DeviceGuard guard(f->device);
    for (void* p : f->allocations) (void)hipFree(p);
  }
  delete f;
}

int ptk_forest_get_dropped(const ptk_forest* f, uint64_t* dropped) {
  if (f == nullptr || dropped == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  DeviceGuard guard(f->device);
  uint32_t v = 0;
  PTK_HIP(hipMemcpy(&v, f->d_dropped, 4, hipMemcpyDeviceToHost));
  *dropped = v;
  return PTK_OK;
}

int ptk_forest_get_rotations(const ptk_forest* f, float* out) {
  if (f == nullptr || out == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  std::memcpy(out, f->rotations.data(), f->rotations.size() * sizeof(float));
  return PTK_OK;
}

int ptk_forest_search_knn_device(const ptk_forest* f, const float* d_q, uint64_t nq, uint32_t k,
                                 uint64_t max_leaves_visited, ptk_neighbor* d_out, void* stream) {
  if (f == nullptr) return fail(PTK_ERR_INVALID, "null forest");
  if (k == 0 || k > 64) return fail(PTK_ERR_INVALID, "k must be in 1..64");
  if (nq == 0) return PTK_OK;
  if (d_q == nullptr || d_out == nullptr) return fail(PTK_ERR_INVALID, "null buffer");
  if (nq >= (1ull << 31)) return fail(PTK_ERR_UNSUPPORTED, "too many queries for one launch");
  DeviceGuard guard(f->device);
  if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", f->device);
  const size_t lds = ((size_t)2 * f->dim + 2 * ptk::kForestQueue + 2 * ptk::kForestPath) * 4;
  const uint32_t leaves = max_leaves_visited > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)max_leaves_visited;
  hipLaunchKernelGGL((ptk::forest_knn_kernel<64>), dim3((uint32_t)nq), dim3(64), lds, static_cast<hipStream_t>(stream),
                     f->dev, d_q, nq, k, leaves, reinterpret_cast<ptk::Neighbor*>(d_out), f->d_dropped);
  PTK_HIP(hipGetLastError());
  return PTK_OK;
}

int ptk_forest_search_knn(const ptk_forest* f, const float* q, uint64_t nq, uint32_t k, uint64_t max_leaves_visited,
                          ptk_neighbor* out) {
  if (f == nullptr) return fail(PTK_ERR_INVALID, "null forest");
  if (nq == 0) return PTK_OK;
  if (q == nullptr || out == nullptr) return fail(PTK_ERR_INVALID, "null buffer");
  DeviceGuard guard(f->device);
  if (!guard.ok) return fail(PTK_ERR_DEVICE, "hipSetDevice(%d) failed", f->device);
  float* d_q = nullptr;
  ptk_neighbor* d_out = nullptr;
  const size_t qb = (size_t)nq * f->dim * 4, ob = (size_t)nq * (k ? k : 1) * sizeof(ptk_neighbor);
  hipError_t he = hipMalloc((void**)&d_q, qb);
  if (he == hipSuccess) he = hipMalloc((void**)&d_out, ob);
  if (he == hipSuccess) he = hipMemcpy(d_q, q, qb, hipMemcpyHostToDevice);
  int rc = PTK_OK;
  if (he == hipSuccess) {
    rc = ptk_forest_search_knn_device(f, d_q, nq, k, max_leaves_visited, d_out, nullptr);
    if (rc == PTK_OK) he = hipMemcpy(out, d_out, ob, hipMemcpyDeviceToHost);
  }
  if (d_q) (void)hipFree(d_q);
  if (d_out) (void)hipFree(d_out);
  if (rc != PTK_OK) return rc;
  if (he != hipSuccess) return fail(PTK_ERR_DEVICE, "HIP error: %s", hipGetErrorString(he));
  return PTK_OK;
}

}  // extern "C"

#include "ptk_host_loop.hpp"
#include "ptk_multi.hpp"

// ---- double precision (ptk_tree64_* / ptk_search64_*) -------------------------------------
#include "ptk_backend_f64.hpp"
