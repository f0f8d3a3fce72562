// ptk_backend_core.hpp -- what the translation units of libptk.so share: the handle (`ptk_tree`), its scratch
// block and event timer, the error channel, and the small host helpers every launch wrapper uses.  The library is
// built from one translation unit per kernel family (one 3 554-line unit took 189 s to compile and
// every edit of any kernel header paid all of it):
//   ptk_backend.hip        the C ABI, tree creation, batch order, the two-phase k = 1 search, box search, forest, multi-GPU
//   ptk_family_knn.hip     general k-NN of 3-D float32 trees (register / row lists, the capped launch + cooperative search)
//   ptk_family_radius.hip  radius search of 3-D float32 trees (traversal, capture, leaf lists + replay)
//   ptk_family_nd.hip      any dimension > 3
//   ptk_family_topo.hip    topological metrics
//   ptk_family_f64.hip     double precision (ptk_backend_f64.hpp)
// Everything here is `inline` (or a type): every unit sees the same definitions, the linker keeps one.
#pragma once

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
// Geometry of the launches (measured optima, profiles/r02_notes.txt item 23, r03_notes.txt item 13):
constexpr int kP2Ring = 12;   // LDS ring of the capped phase 2 (records per lane; 8 / 10 / 16: 1.335 / 1.329 / 1.308 vs 1.224 ms)
#ifndef PTK_GEN_RING
#define PTK_GEN_RING 16
#endif
constexpr int kGenRing = PTK_GEN_RING;  // LDS ring of the general searches (k > 1, radius)
constexpr int kGenLeafB = 5;  // points per leaf round of the general searches (4 / 5 / 6: knn = 16 4.85 / 4.71 / 4.68 ms, radius capture 7.21 / 7.16 / 7.69)
#include "ptk_kernels_nd.hpp"

// Host-side builder: the product's own header-only flat-tree builder.
#include "pico_tree/internal/flat_tree.hpp"
#include "pico_tree/internal/stream.hpp"
#include "pico_tree/map.hpp"


static_assert(sizeof(ptk_neighbor) == 8 && sizeof(ptk::Neighbor) == 8, "neighbor layout");
static_assert(sizeof(ptk_node) == 16, "node layout");
static_assert(
    sizeof(pico_tree::internal::flat_node<int, float>) == sizeof(ptk_node), "flat node layout");


namespace ptkb {

inline thread_local std::string g_error = "";  // (one per thread for the whole library: C++17 inline variable)

inline int fail(int status, const char* fmt, ...) {
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
  const uint32_t* last_meta = nullptr;  // the counters block of the last search that keeps one (ptk_debug_knn1_counts,
                                        // ptk_debug_knn_coop_counts)
  int last_meta_kind = 0;               // which search wrote it: 1 = two-phase k = 1, 2 = capped k > 1
  uint32_t last_meta_cap[2] = {0, 0};   // capped k > 1: entries of its hand-over list(s) -- the counters run past them
  // Second stream of a small k = 1 batch: the cooperative search of the ranked classes runs beside phase 2
  // (launch_knn1_two_phase); forked and joined with events, so the caller's stream still orders everything.
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  // The coherence sample of a batch (ptk::coherence_sample_kernel): its state and verdict stay on the device.
  uint32_t* d_sample = nullptr;             // device (the tail of `base`): {windows counted << 16 | windows failed, verdict}, zero between batches
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
  ptk::RadiusHeavy cap_heavy{};  // ... and what the capped list pass handed to wavefronts (ptk_kernels_coopr.hpp); max_heavy == 0: nothing
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


}  // namespace ptkb
using namespace ptkb;

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


namespace ptkb {


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

inline int env_int(const char* name, int fallback);  // defined with the launch helpers below

// Host threads for the tree build (the result does not depend on it): PTK_BUILD_THREADS, else the
// hardware concurrency capped at 32.
inline unsigned build_threads() {
  const int v = env_int("PTK_BUILD_THREADS", 0);
  if (v > 0) return (unsigned)v;
  const unsigned hc = std::thread::hardware_concurrency();
  return hc == 0 ? 1u : (hc > 32u ? 32u : hc);
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
      const size_t want = (bytes + kAlign + (size_t(32) << 20)) & ~((size_t(32) << 20) - 1);
      ws_.d_sample = nullptr;
      if (hipMalloc((void**)&ws_.base, want) != hipSuccess) {
        (void)hipGetLastError();  // not sticky: the next launch must not report this again
        ws_.base = nullptr;
        return fail(PTK_ERR_NOMEM, "out of device memory (%zu bytes of search scratch)", want);
      }
      // The last kAlign bytes of the block are the two words of the coherence sample (sample_state()): made with the
      // block and zeroed on the stream of the call that made it, so that an entry point that "only enqueues" never
      // allocates or touches the null stream once the block has its size.
      ws_.capacity = want - kAlign;
      ws_.d_sample = reinterpret_cast<uint32_t*>(ws_.base + ws_.capacity);
      if (hipMemsetAsync(ws_.d_sample, 0, kAlign, s_) != hipSuccess) {
        (void)hipGetLastError();
        ws_.d_sample = nullptr;  // (no sample: every batch is sorted)
      }
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
  void note_meta(const uint32_t* meta, int kind, uint32_t cap0 = 0, uint32_t cap1 = 0) {
    ws_.last_meta = meta;
    ws_.last_meta_kind = kind;
    ws_.last_meta_cap[0] = cap0;
    ws_.last_meta_cap[1] = cap1;
  }
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
  // The two device words of the coherence sample of this block's batches (the tail of the block: reserve()).
  uint32_t* sample_state() { return ws_.d_sample; }
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
inline const ptk::DevTree& knn1_tree(const ptk_tree* t) { return t->n_piles ? t->dev1 : t->dev; }
inline const uint2* knn1_ranges(const ptk_tree* t) { return static_cast<const uint2*>(t->n_piles ? t->d_ranges1 : t->d_ranges); }
inline uint32_t knn1_depth(const ptk_tree* t) { return t->n_piles ? t->max_depth1 : t->max_depth; }

inline int ovf_class_of(uint32_t depth, int s_lds);
inline int ovf_class(const ptk_tree* t, int s_lds) { return ovf_class_of(t->max_depth, s_lds); }
inline int ovf_class_of(uint32_t depth, int s_lds) {
  const uint32_t need = 2 * depth + 2;
  if (need <= (uint32_t)s_lds + 64) return 0;
  if (need <= (uint32_t)s_lds + 256) return 1;
  if (need <= (uint32_t)s_lds + 2048) return 2;
  return kDeepClass;
}
inline bool deep_tree(const ptk_tree* t) { return ovf_class(t, 16) == kDeepClass; }

// A deep tree's launches: `cap` spill records per lane, `piece` queries per launch so that the
// block stays within PTK_DEEP_SPILL_MB (default 2048).
struct DeepPlan {
  uint32_t cap;
  uint64_t piece;
  size_t bytes() const { return (size_t)piece * cap * sizeof(ptk::Record); }
};
inline DeepPlan deep_plan(const ptk_tree* t, uint64_t n) {
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
inline int allow_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    PTK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  }
  return PTK_OK;
}

// An integer from the environment: the OPERATIONAL switches of the library (memory caps, threads, the device build;
// the table in DESIGN.md section 11 lists all of them).
inline int env_int(const char* name, int fallback) {
  const char* v = std::getenv(name);
  return v ? std::atoi(v) : fallback;
}

// A TEST HOOK: `name=value` in the comma-separated list PTK_TEST_KNOBS (e.g. PTK_TEST_KNOBS="knn_cap=4,knn_cap_min_nq=1").
// The hooks force a path a test wants to see taken -- a cap on every batch, the hand-over list of a small batch, one of
// the two sorts -- and are not tuning switches: each one's default is the measured optimum.
// Read on every call: a test may change the list between two searches.
inline int knob_int(const char* name, int fallback) {
  const char* v = std::getenv("PTK_TEST_KNOBS");
  if (v == nullptr) return fallback;
  const size_t n = std::strlen(name);
  while (*v != '\0') {
    while (*v == ',' || *v == ' ') ++v;
    if (std::strncmp(v, name, n) == 0 && v[n] == '=') return std::atoi(v + n + 1);
    while (*v != '\0' && *v != ',') ++v;
  }
  return fallback;
}


inline int check_search(const ptk_tree* t, const void* q, uint64_t nq) {
  if (t == nullptr) return fail(PTK_ERR_INVALID, "null tree");
  if (nq > 0 && q == nullptr) return fail(PTK_ERR_INVALID, "null query buffer");
  if (t->device == kDeviceNone) return fail(PTK_ERR_DEVICE, "this handle has no device replica");
  if (!t->gpu_layout) return fail(PTK_ERR_DEVICE, "this handle has no device replica");
  return PTK_OK;
}

inline float inv_ratio(float e) { return 1.0f / e; }


// Far children a query of the general k-NN kernel may enter before it is handed to the cooperative search
// (ptk_kernels_coopk.hpp; test hook knn_cap, 0 = every query runs to its end in its lane).  Exact searches with the default
// metric only: the argument that makes the merged result the reference's needs e = 1 and the error bounds of a sum of
// squares.
constexpr int kKnnCoopPool = 128;
constexpr uint32_t kKnnCoopSpill = 2048;  // tasks a wavefront of the cooperative search can park in HBM
inline uint64_t knn_max_handover(uint64_t nq);
// Wavefronts of the cooperative search: what is resident at once (32 per CU), and never more than the hand-over list has
// entries -- every wavefront owns a run of kKnnCoopSpill tasks of the spill block (48 KB), so a small batch must not
// reserve the block of a full one (403 MB for a 256-query call).
inline uint32_t knn_coop_blocks(const ptk_tree* t, uint64_t nq) {
  return (uint32_t)std::min<uint64_t>((uint64_t)t->cus * 32u, std::max<uint64_t>(64, knn_max_handover(nq)));
}
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
// Test hook knn_cap = n: that cap for every batch (0: no cap).
inline uint32_t knn_cap(float e, uint64_t nq, uint32_t k) {
  // (below a wavefront of queries the two extra launches cost what a tail may or may not: ms per call with / without the
  // cap, knn = 16, queries taken across config 2's scan -- some of them long --: 8 queries 0.19-0.21 / 0.16-0.34, 64
  // 0.23-0.24 / 0.28-0.31, 200 0.35 / 0.53-0.62 (tools/time_tiny_batches.py; r05 had measured the first 64 rows of the
  // batch, none of them long, and set 256).  Test hook knn_cap_min_nq.)
  if (k < 2 || e != 1.0f || nq < (uint64_t)std::max(1, knob_int("knn_cap_min_nq", 32))) return 0;  // (k = 1 has the two-phase search; where it does not apply -- metric_l1 on a tree with piles -- the general kernel runs uncapped)
  const int forced = knob_int("knn_cap", -1);
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
inline uint64_t knn_max_handover(uint64_t nq) { return std::max<uint64_t>(nq / 48, std::min<uint64_t>(nq, 24576)); }
// (two lists -- front and rest -- only from 2^22 queries on: the second one's floor of 24 576 entries)
inline size_t knn_coop_scratch_bytes(const ptk_tree* t, uint64_t nq) {
  return 3 * (nq * 4) + (knn_max_handover(nq) + (nq >= (1ull << 22) ? 24576 : 0)) * ptk::kMaxTasks * sizeof(ptk::Task) +
         ptk::kMetaWords * 4 + (size_t)knn_coop_blocks(t, nq) * kKnnCoopSpill * sizeof(ptk::Task) + 1024;
}

// ---- the capped list pass of the radius search (ptk_kernels_coopr.hpp) ----
// Far children a query of the radius list pass may enter before a wavefront takes it over; 0 = every query runs to its
// end in its lane.  A capped launch ends with the lanes that ran to their cap (~7 us per far child for a lonely lane),
// so the cap follows the batch as the k-NN one does (knn_cap): what it keeps roughly constant is the number of
// hand-overs.  Trees deeper than a key has bits for (kRcMaxDepth) and batches of a few wavefronts run uncapped.  Test
// hook radius_cap: that cap for every batch (0: none).
constexpr int kRadiusCoopPool = 64;
constexpr uint32_t kRadiusCoopSpill = 1024;  // tasks a wavefront of the cooperative count can park in HBM (32 KB each with their keys:
                                             // 134 MB for the 4 096 wavefronts of a full launch; what overflows is recounted by one lane)
inline uint32_t radius_cap(const ptk_tree* t, uint64_t nq) {
  // (51 = ptk::kRcMaxDepth, the branches a key has bits for: static_assert in ptk_family_radius.hip; a leaf of more than
  // 64 pieces of 32 points would run out of the key's piece bits)
  if (t->dim > 3 || t->max_depth > 51u || t->max_leaf_count > 2048u) return 0;
  const int forced = knob_int("radius_cap", -1);
  if (forced >= 0) return (uint32_t)forced;
  // Measured on BASELINE config 3's cloud (tools/time_radius_sizes.py, profiles/r06_notes.txt item 3; count + fill ms,
  // best cap against no cap): 20 k queries 0.37 (cap 8) / 2.27, 150 k 1.09 (64) / 2.41, 900 k 2.86 (256) / 3.10; at 2.4 M
  // and beyond the bulk of the launch hides the tail and every cap loses (4.23 (256) / 4.05; 7.2 M 11.4 / 10.2: no query
  // of that cloud enters 384 far children, 1.2 % enter more than 256).  What the best caps have in common is 8-17 thousand
  // hand-overs -- what the wavefronts of the cooperative count get through while the capped launch ends.
  // (no lower limit: a call of 8 queries that holds one long one takes 0.8 ms uncapped -- the count pass's tail --, and
  // the capped path costs a call without long queries ~30 us: tools/time_tiny_batches.py.  Test hook radius_cap_min_nq.)
  if (nq < (uint64_t)std::max(1, knob_int("radius_cap_min_nq", 1)) || nq >= 1500000) return 0;
  return (uint32_t)std::min(256.0, std::max(8.0, (double)nq / 2400.0));
}
inline uint64_t radius_max_handover(uint64_t nq) { return std::max<uint64_t>(nq / 48, std::min<uint64_t>(nq, 24576)); }
// Entries of all hand-overs together (a few queries may all be long: room for 64 of them at least).
inline uint64_t radius_entry_cap(uint64_t nq) { return std::max<uint64_t>(radius_max_handover(nq), 64) * 192; }
inline uint32_t radius_coop_blocks(const ptk_tree* t, uint64_t nq) {
  return (uint32_t)std::min<uint64_t>((uint64_t)t->cus * 16u, std::max<uint64_t>(64, radius_max_handover(nq)));
}
// Transient arrays of a capped list pass (the hand-over list with its tasks, the redo list, the spill runs).
inline size_t radius_coop_scratch_bytes(const ptk_tree* t, uint64_t nq) {
  const uint64_t mh = radius_max_handover(nq);
  return 3 * (mh * 4) + mh * ptk::kMaxTasks * sizeof(ptk::Task) + (size_t)radius_coop_blocks(t, nq) * kRadiusCoopSpill * 32 + 2048;
}

// The largest k whose list lives in registers (3-D kernels, every metric): 64 slots (a list of 40 in LDS took 74 ms on
// BASELINE config 3 where 32 in registers take 7: insert_sorted through LDS is a loop per lane).
inline uint32_t knn_reg_max(bool) { return 64u; }
// k <= 64: the k-list in registers (K = 4 / 8 / 16 / 32 / 64 slots compiled).

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
inline bool topological(const ptk_tree* t) {
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


}  // namespace ptkb
