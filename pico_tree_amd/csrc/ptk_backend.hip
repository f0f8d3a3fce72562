// ptk_backend.hip -- host side of libptk.so: the C ABI of include/ptk.h.
//
// Responsibilities: validate arguments, build (optionally) and re-encode the flat
// tree for the device, keep it resident in HBM, order query batches, launch the
// gfx950 kernels of ptk_kernels.hpp, and move results.  There is no CPU search
// path in this file: without a usable device every search entry point fails with
// PTK_ERR_DEVICE.

#include "ptk_families.hpp"
#include "ptk_kernels_lists.hpp"
#include "ptk_piles.hpp"
#include "ptk_build.hpp"
#include "ptk_sort.hpp"
#include "ptk_forest.hpp"

namespace {

void axis_bits(const ptk_tree* t, int bits, uint32_t b[3]);

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
    // spread over the axes like the bits of the order key.
    int cb = 0;
    while ((64ull << cb) <= t.n_points) ++cb;  // floor(log2(n / 32))
    cb = std::min(std::max(cb, 6), 22);
    {
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
  if (knob_int("pile_view", 1) != 0) {
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
      // (one code object per translation unit: each family loads its own)
      ptkf::warm_knn();
      ptkf::warm_radius();
      ptkf::warm_nd();
      ptkf::warm_topo();
      ptkf::warm_f64();
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
// 68 / 58 / 60 us for 900 k queries).
uint32_t sort_tile(uint64_t nq) {
  const uint64_t t = ((nq + 4095) / 4096 + 63) & ~(uint64_t)63;
  return (uint32_t)std::max<uint64_t>(t, (uint64_t)512);
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
  const int mode = knob_int("sort_block", -1);
  return mode < 0 ? nq >= (3ull << 18) : mode != 0 && nq >= ptk::kSortTile;
}
bool own_sort(uint64_t nq) {
  if (nq >= (1ull << 31)) return false;
  return knob_int("sort", 1) != 0;
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
  if (nq < 8192) return nullptr;
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
  if (heavy_first != 0u && t->cells.occ != nullptr) {
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
// with_heavy: room for what a capped list pass keeps of its hand-overs until the fill pass (ptk::RadiusHeavy: counters,
// four words per hand-over, the sorted entries), between the tables and the chunks.
bool prepare_capture(const ptk_tree* t, uint64_t nq, Workspace& ws, bool with_heavy = false) {
  const size_t budget = capture_budget_bytes(t);
  if (budget == 0 || nq == 0 || nq >= (1ull << 31)) return false;
  const size_t waves = (size_t)((nq + 63) / 64);
  const size_t chunk_bytes = (size_t)ptk::kLogChunk * sizeof(ptk::Neighbor);
  const size_t flags_at = (size_t)ptk::kCapSubPools * ptk::kCapCounterStride * 4;
  const size_t qids_at = flags_at + ((waves + 255) & ~(size_t)255);
  const size_t lens_at = qids_at + waves * 64 * 4;
  const size_t tables_at = lens_at + waves * 64 * 4;
  const size_t heavy_at = (tables_at + waves * ptk::kListMaxChunks * 4 + 255) & ~(size_t)255;
  const size_t mh = with_heavy ? (size_t)radius_max_handover(nq) : 0, ecap = with_heavy ? (size_t)radius_entry_cap(nq) : 0;
  const size_t heavy_bytes = with_heavy ? 256 + 4 * ((mh * 4 + 255) & ~(size_t)255) + ecap * 8 : 0;
  const size_t head = (heavy_at + heavy_bytes + 4095) & ~(size_t)4095;
  if (head + waves * chunk_bytes > budget) return false;
  const size_t dyn = std::min<size_t>((budget - head - waves * chunk_bytes) / chunk_bytes, waves * 64 * 2);
  const int forced = knob_int("radius_capture_chunks", -1);
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
  ws.cap_heavy = ptk::RadiusHeavy{};
  if (with_heavy) {
    const size_t row = (mh * 4 + 255) & ~(size_t)255;
    char* p = ws.cap_base + heavy_at;
    ws.cap_heavy.meta = reinterpret_cast<uint32_t*>(p);
    ws.cap_heavy.rows = reinterpret_cast<uint32_t*>(p + 256);
    ws.cap_heavy.own = reinterpret_cast<uint32_t*>(p + 256 + row);
    ws.cap_heavy.run_at = reinterpret_cast<uint32_t*>(p + 256 + 2 * row);
    ws.cap_heavy.run_n = reinterpret_cast<uint32_t*>(p + 256 + 3 * row);
    ws.cap_heavy.entries = reinterpret_cast<unsigned long long*>(p + 256 + 4 * row);
    ws.cap_heavy.max_heavy = (uint32_t)mh;
    ws.cap_heavy.entry_cap = (uint32_t)std::min<size_t>(ecap, 0xFFFFFFFFu);
  }
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
  p.per = nq < (2ull << 20) ? 512u : 1792u;
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
  const int cap = knob_int("p2_cap", nq >= (4ull << 20) ? 24 : 8);
  return cap < 0 ? 0u : (uint32_t)cap;
}

// direct_ids == nullptr: the list is `ho`'s.  Otherwise it is the ranked head of the class-sorted entries, searched
// straight from the continuation records of phase 1 (knn1_coop_kernel<.., DIRECT>), `lanes` lanes per query.
template <int G, class M = ptk::MetricL2>
int launch_knn1_coop_direct(const ptk_tree* t, const float4* qs, ptk::Neighbor* d_out, const ptk::Cont& cont,
                            const ptk::Handover& ho, uint32_t* redo_list, hipStream_t s, ptk::Task* spill,
                            const uint32_t* direct_ids) {
  constexpr size_t smem = (size_t)(64 / G) * (6 * kCoopPool + 2) * 4;
  // (the spill block is sized for coop_waves(t) x 64 / kCoopLanes groups: a wider group count would not fit)
  static_assert(G >= kCoopLanes, "the spill block is sized for groups of kCoopLanes lanes");
  const int resident = t->cus * (int)std::max<size_t>(1, std::min<size_t>(32, t->lds_per_cu / (smem + 512)));
  const int waves = std::min(resident, coop_waves(t) * (G / kCoopLanes));
  hipLaunchKernelGGL((ptk::knn1_coop_kernel<G, kCoopPool, true, true, M>), dim3(waves), dim3(64), smem, s, knn1_tree(t),
                     knn1_ranges(t), qs, d_out, cont, ho, redo_list, direct_ids, spill, kCoopSpill);
  PTK_HIP(hipGetLastError());
  return PTK_OK;
}

template <class M = ptk::MetricL2>
int launch_knn1_coop(const ptk_tree* t, const float4* qs, ptk::Neighbor* d_out, const ptk::Cont& cont,
                     const ptk::Handover& ho, uint32_t* redo_list, hipStream_t s, ptk::Task* spill,
                     const uint32_t* direct_ids = nullptr) {
  constexpr size_t smem = (size_t)(64 / kCoopLanes) * (6 * kCoopPool + 2) * 4;
  const int waves = coop_waves(t);
  const uint32_t spill_cap = spill ? kCoopSpill : 0u;
  if (direct_ids != nullptr) {
    // (32 lanes per query: 16 / 64 were measured, 1.91 / slower vs 1.72 ms of traversal kernels, profiles/r03_notes.txt item 3)
    return launch_knn1_coop_direct<32, M>(t, qs, d_out, cont, ho, redo_list, s, spill, direct_ids);
  }
  // What phase 2 hands over has been tightened by its first far children: a pool of 96 holds it (0 of 152 k queries of
  // BASELINE config 2 overflow; with the spill the step loop is 5 % slower): no spill here.
  (void)spill_cap;
  hipLaunchKernelGGL((ptk::knn1_coop_kernel<kCoopLanes, kCoopPool, false, false, M>), dim3(waves), dim3(64), smem, s, knn1_tree(t),
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
  return knob_int("coop_direct", deep || nq < (1ull << 20) ? 2 : 0);
}

// The k = 1 search under the default metric (ptk_kernels.hpp, "the two-phase k = 1 search"): phase 1 (which also
// packs the launch-order records), the class order of the continuations, phase 2, and for exact searches the
// cooperative search of what phase 2 handed over plus the replay of what that could not certify.
template <int OVF, class M = ptk::MetricL2>
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
  scratch.note_meta(cont.meta, 1);
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
  hipLaunchKernelGGL((ptk::knn1_phase1u_kernel<LEAFB, M>), dim3(blocks), dim3(64), 0, s, knn1_tree(t), d_q, t->dim, perm, nq,
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
    int rc = launch_knn1_coop<M>(t, qs, d_out, cont, ho, redo_list, side, spill_a, ids_out);
    if (rc != PTK_OK) return rc;
    PTK_HIP(hipEventRecord(join, side));
  } else if (direct == 1) {
    int rc = launch_knn1_coop<M>(t, qs, d_out, cont, ho, redo_list, s, spill_a, ids_out);
    if (rc != PTK_OK) return rc;
  }
  const dim3 p2_grid(blocks + 1 + extra_waves);
  // LDS ring of phase 2.  With the cap no stack grows deep: 12 slots = 6 KB per wave = 26 waves per CU beat
  // 16 (20 waves) and 8 (40 waves) on both clouds (profiles/r02_notes.txt items 10, 23); without it 16 slots
  // (r01l_notes item 8).
  if (cap) {
    hipLaunchKernelGGL((ptk::knn1_phase2_kernel<kP2Ring, OVF, LEAFB, M>), p2_grid, dim3(64), (size_t)kP2Ring * 64 * 8, s, knn1_tree(t), qs,
                       e_inv, d_out, cont, ids_out, cap, ho);
  } else {
    hipLaunchKernelGGL((ptk::knn1_phase2_kernel<16, OVF, LEAFB, M>), p2_grid, dim3(64), (size_t)16 * 64 * 8, s, knn1_tree(t), qs,
                       e_inv, d_out, cont, ids_out, 0u, ho);
  }
  PTK_HIP(hipGetLastError());
  if (cap) {  // the queries phase 2 gave up on, then whatever the cooperative search could not certify
    // (the two cooperative launches may run side by side: each has its own spill block, and they append to the one
    // redo list through one atomic counter)
    int rc = launch_knn1_coop<M>(t, qs, d_out, cont, ho, redo_list, s, spill_b);
    if (rc != PTK_OK) return rc;
    if (direct == 2) {  // join: the replay needs both lists complete
      PTK_HIP(hipStreamWaitEvent(s, join, 0));
      side_guard.side = nullptr;
    }
    hipLaunchKernelGGL((ptk::knn1_redo_kernel<16, OVF, LEAFB, M>), dim3(t->cus), dim3(64), (size_t)16 * 64 * 8, s, knn1_tree(t), qs,
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

template <class M>
int dispatch_knn1_of(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float e,
                     ptk::Neighbor* d_out, hipStream_t s, Scratch& scratch) {
  int rc = PTK_OK;
  switch (ovf_class_of(knn1_depth(t), 16)) {  // (the depth of what is traversed: the view without the piles if there is one)
    case 0: rc = launch_knn1_two_phase<64, M>(t, d_q, perm, nq, e, d_out, s, scratch); break;
    case 1: rc = launch_knn1_two_phase<256, M>(t, d_q, perm, nq, e, d_out, s, scratch); break;
    case 2: rc = launch_knn1_two_phase<2048, M>(t, d_q, perm, nq, e, d_out, s, scratch); break;
    default: rc = fail(PTK_ERR_UNSUPPORTED, "tree depth %u is too deep for the device stack", knn1_depth(t));
  }
  return rc;
}
// The two-phase k = 1 search: metric_l2_squared, and (r06) metric_l1 on trees without piles (see knn1_two_phase()).
int dispatch_knn1(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float e,
                  ptk::Neighbor* d_out, hipStream_t s, Scratch& scratch) {
  if (t->metric.load() == PTK_METRIC_L1) return dispatch_knn1_of<ptk::MetricL1>(t, d_q, perm, nq, e, d_out, s, scratch);
  return dispatch_knn1_of<ptk::MetricL2>(t, d_q, perm, nq, e, d_out, s, scratch);
}

}  // namespace

namespace ptkf {
int morton_bits(uint64_t nq) { return ::morton_bits(nq); }
size_t sort_tmp_bytes(uint64_t nq, int bits) { return ::sort_tmp_bytes(nq, bits); }
size_t permutation_scratch_bytes(uint64_t nq) { return ::permutation_scratch_bytes(nq); }
int sort_pairs_u32(void* tmp, size_t tmp_bytes, uint32_t* keys, uint32_t* keys_out, uint32_t* ids, uint32_t* ids_out,
                   uint64_t nq, int bits, hipStream_t s) {
  PTK_HIP(rocprim::radix_sort_pairs<MortonSortConfig>(tmp, tmp_bytes, keys, keys_out, ids, ids_out, nq, 0, bits, s));
  return PTK_OK;
}
}  // namespace ptkf

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
  // k = 1 takes the two-phase search (phase 1, class order, capped phase 2, cooperative search) under the default metric
  // and under metric_l1 on a tree without piles: both are sums of per-axis terms, so the box distance the reference
  // keeps is a lower bound of the point distances below it -- what the cooperative search's certificate needs.  (The
  // pile view resolves its stand-ins for the default metric only.)
  const bool two_phase = k == 1 && t->dim <= 3 && (l2 || (t->metric.load() == PTK_METRIC_L1 && t->n_piles == 0));
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
    return ptkf::knn_topo(t, d_q, perm, nq, k, e, reinterpret_cast<ptk::Neighbor*>(d_out), s, short_tree);
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
      if (t->dim > 3) {
        ptk::DevTreeND dev = t->dev_nd;
        dev.deep_spill = spill;
        dev.deep_cap = plan.cap;
        rc = ptkf::knn_nd_deep(t, dev, d_q + lo * t->dim, n, k, e, o + lo * k, s);
      } else {
        ptk::DevTree dev = t->dev;
        dev.deep_spill = spill;
        dev.deep_cap = plan.cap;
        rc = ptkf::knn_deep(t, dev, d_q + lo * t->dim, n, k, e, o + lo * k, s);
      }
      if (rc != PTK_OK) return rc;
      PTK_HIP(hipGetLastError());
    }
    timer.stop(0, nq);
    return PTK_OK;
  }
  const bool reorder = want_reorder(t, nq);
  Scratch scratch(t, s, /*per_stream=*/true);
  rc = scratch.reserve((reorder ? permutation_scratch_bytes(nq) : 0) +
                       (two_phase ? two_phase_scratch_bytes(t, nq) : 0) +
                       (k > 1 && k <= 64 && (l2 || t->metric.load() == PTK_METRIC_L1) && t->dim <= 3 && knn_cap(e, nq, k) != 0u
                            ? knn_coop_scratch_bytes(t, nq)
                            : 0));
  if (rc != PTK_OK) return rc;
  uint32_t* perm = nullptr;
  if (reorder) {  // Morton order along the first three axes, whatever the dimension
    // (the general kernels run every query to its end in its lane: the expensive queries to the front of the launch)
    // (the two-phase k = 1 search orders its own continuations: a batch that arrives coherent is not sorted again --
    // REORDER_AUTO only; the general kernels want the expensive queries in front whatever the order)
    const bool may_skip = two_phase && t->reorder.load() == PTK_REORDER_AUTO;
    rc = make_permutation(t, d_q, nq, s, scratch, &perm, t->dim <= 3 && !two_phase ? ptk::kCellsEmptyFirst : 0u, may_skip);
    if (rc != PTK_OK) return rc;
  }
  if (t->dim > 3) {
    return ptkf::knn_nd(t, d_q, perm, nq, k, e, reinterpret_cast<ptk::Neighbor*>(d_out), s, short_tree);
  }
  if (two_phase) {
    rc = dispatch_knn1(t, d_q, perm, nq, e, reinterpret_cast<ptk::Neighbor*>(d_out), s, scratch);
  } else if (k <= knn_reg_max(l2) && !short_tree) {
    rc = ptkf::knn_reg(t, d_q, perm, nq, k, e, reinterpret_cast<ptk::Neighbor*>(d_out), s, &scratch);
  } else {
    rc = ptkf::knn_rows(t, d_q, perm, nq, k, e, reinterpret_cast<ptk::Neighbor*>(d_out), s);
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
  const int forced = knob_int("host_piece", 0);
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
  const bool capped = !two_phase && k > 1 && k <= 64 && t->dim <= 3 &&
                      (t->metric.load() == PTK_METRIC_L2_SQUARED || t->metric.load() == PTK_METRIC_L1) &&
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
  const int n_search_streams = 2;
  // Arrays that are page-locked already (ptk_host_alloc: what the Python wrapper returns its rows in) are copied to
  // and from directly; pageable ones go through the handle's pinned rings.
  const bool in_pinned = knob_int("host_direct", 1) != 0 && is_pinned_host(q, (size_t)nq * row_in);
  const bool out_pinned = knob_int("host_direct", 1) != 0 && is_pinned_host(out, (size_t)nq * row_out);
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
    const int want = hc >= 16 ? 8 : (hc >= 4 ? (int)hc / 2 : 1);
    io.pool.reset(new CopyPool((unsigned)std::max(0, want - 1)));  // (the calling threads copy too)
  }
  float* d_q = reinterpret_cast<float*>(io.d_in);
  ptk_neighbor* d_out = reinterpret_cast<ptk_neighbor*>(io.d_out);
  const char* src = reinterpret_cast<const char*>(q);
  char* dst = reinterpret_cast<char*>(out);

  const bool trace = knob_int("host_trace", 0) != 0;
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
    // (rows to search again: those of wavefronts whose lists were lost, and handed-over rows whose entries were)
    const uint64_t n_over_max = nq + ws.cap_heavy.max_heavy;
    rc = scratch.reserve(n_over_max * 4 + 256);
    if (rc != PTK_OK) return rc;
    uint32_t* over_list = scratch.take<uint32_t>(n_over_max);
    uint32_t* n_over = scratch.take<uint32_t>(1);
    if (over_list == nullptr || n_over == nullptr) return fail(PTK_ERR_NOMEM, "scratch block too small");
    if (ws.cap_lists) {  // (3-D trees: the rows are made from the leaf lists of the count pass)
      rc = ptkf::radius_replay(t, d_q, e, ws.cap, d_offsets, reinterpret_cast<ptk::Neighbor*>(d_out), over_list, n_over, s,
                               ws.cap_heavy.max_heavy != 0u ? &ws.cap_heavy : nullptr);
      if (rc != PTK_OK) return rc;
    } else {
      rc = ptkf::radius_log_scatter(t, ws.cap, d_offsets, reinterpret_cast<ptk::Neighbor*>(d_out), over_list, n_over, s);
      if (rc != PTK_OK) return rc;
    }
    // Rows the capture could not hold (possibly none: the blocks then leave at once).
    if (nd) {
      rc = ptkf::radius_nd(t, d_q, nq, radius, e, true, nullptr, d_offsets, reinterpret_cast<ptk::Neighbor*>(d_out), s,
                           over_list, n_over);
    } else {
      rc = ptkf::radius_traverse(t, d_q, over_list, n_over_max, radius, e, true, nullptr, d_offsets,
                                 reinterpret_cast<ptk::Neighbor*>(d_out), s, n_over);
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
    rc = ptkf::radius_topo(t, d_q, perm, nq, radius, e, fill, d_counts, d_offsets, reinterpret_cast<ptk::Neighbor*>(d_out), s);
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
      uint64_t* c = fill ? nullptr : d_counts + lo;
      const uint64_t* of = fill ? d_offsets + lo : nullptr;
      if (nd) {
        ptk::DevTreeND dev = t->dev_nd;
        dev.deep_spill = spill;
        dev.deep_cap = plan.cap;
        rc = ptkf::radius_nd_deep(t, dev, d_q + lo * t->dim, n, radius, e, fill, c, of, o, s);
      } else {
        ptk::DevTree dev = t->dev;
        dev.deep_spill = spill;
        dev.deep_cap = plan.cap;
        rc = ptkf::radius_deep(t, dev, d_q + lo * t->dim, n, radius, e, fill, c, of, o, s);
      }
      if (rc != PTK_OK) return rc;
      PTK_HIP(hipGetLastError());
    }
    timer.stop(0, fill ? 0 : nq);
  } else {
    // (the long queries of a list pass are handed to wavefronts: ptk_kernels_coopr.hpp)
    const uint32_t far_cap = !fill && !nd && knob_int("radius_lists", 1) != 0 ? radius_cap(t, nq) : 0u;
    const bool capture = !fill && prepare_capture(t, nq, ws, far_cap != 0u);
    const bool lists = capture && !nd && knob_int("radius_lists", 1) != 0;
    if (!fill) ws.cap_valid = false;
    rc = scratch.reserve((reorder ? permutation_scratch_bytes(nq) : 0) + (lists && far_cap ? radius_coop_scratch_bytes(t, nq) : 0));
    if (rc != PTK_OK) return rc;
    uint32_t* perm = nullptr;
    if (reorder) {  // (a query costs what it finds: the densest cells to the front of the launch)
      rc = make_permutation(t, d_q, nq, s, scratch, &perm, nd ? 0u : ptk::kCellsDenseFirst);
      if (rc != PTK_OK) return rc;
    }
    if (capture) {
      if (nd) {
        rc = ptkf::radius_nd_capture(t, d_q, perm, nq, radius, e, d_counts, ws.cap, s);
      } else if (lists) {  // the count pass lists the leaves with hits for the fill pass
        rc = ptkf::radius_list(t, d_q, perm, nq, radius, e, d_counts, ws.cap, s, far_cap, &scratch, &ws.cap_heavy);
      } else {
        rc = ptkf::radius_capture(t, d_q, perm, nq, radius, e, d_counts, ws.cap, s);
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
      rc = ptkf::radius_nd(t, d_q, nq, radius, e, fill, d_counts, d_offsets, reinterpret_cast<ptk::Neighbor*>(d_out), s, perm);
    } else {
      rc = ptkf::radius_traverse(t, d_q, perm, nq, radius, e, fill, d_counts, d_offsets,
                                 reinterpret_cast<ptk::Neighbor*>(d_out), s);
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
  if (holder->last_meta == nullptr || holder->last_meta_kind != 1)
    return fail(PTK_ERR_INVALID, "the last search of this handle was not a two-phase k = 1 search");
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
  if (holder->last_meta == nullptr || holder->last_meta_kind != 2)
    return fail(PTK_ERR_INVALID, "the last search of this handle was not a capped k > 1 search");
  uint32_t meta[ptk::kMetaWords];
  PTK_HIP(hipDeviceSynchronize());
  PTK_HIP(hipMemcpy(meta, holder->last_meta, sizeof(meta), hipMemcpyDeviceToHost));
  // (a query that found its list full went on in its lane, Handover::full_keeps: the counters run past the lists)
  counts[0] = std::min(meta[ptk::kMetaHeavy], holder->last_meta_cap[0]) +
              std::min(meta[ptk::kMetaHeavyRest], holder->last_meta_cap[1]);
  counts[1] = meta[ptk::kMetaRedo];
  counts[2] = meta[ptk::kKnnWhyPool];
  counts[3] = meta[ptk::kKnnWhyTie];
  counts[4] = meta[ptk::kKnnWhyBox];
  counts[5] = meta[ptk::kKnnWhyRange];
  counts[6] = meta[ptk::kKnnTieSweeps];
  return PTK_OK;
}

int ptk_debug_radius_coop_counts(const ptk_tree* t, uint32_t counts[3]) {
  if (t == nullptr || counts == nullptr) return fail(PTK_ERR_INVALID, "null argument");
  if (t->device < 0) return fail(PTK_ERR_DEVICE, "this handle has no device replica");
  DeviceGuard guard(t->device);
  std::lock_guard<std::mutex> lock(t->ws.mutex);
  counts[0] = counts[1] = counts[2] = 0;
  if (!t->ws.cap_valid || !t->ws.cap_lists || t->ws.cap_heavy.max_heavy == 0u) return PTK_OK;
  uint32_t meta[ptk::kMetaWords];
  PTK_HIP(hipDeviceSynchronize());
  PTK_HIP(hipMemcpy(meta, t->ws.cap_heavy.meta, sizeof(meta), hipMemcpyDeviceToHost));
  counts[0] = std::min(meta[ptk::kMetaHeavy], t->ws.cap_heavy.max_heavy);
  counts[1] = meta[ptk::kMetaRedo];
  counts[2] = meta[28];  // ptk::kMetaRcEntries (ptk_kernels_coopr.hpp)
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

