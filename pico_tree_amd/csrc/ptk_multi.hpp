// ptk_multi.hpp -- one tree on several GPUs of a node, behind the C ABI (ptk_multi_* of ptk.h).
// Included by ptk_backend.hip (one translation unit).
//
// The path shards: queries are independent (the reference's batch harness is an
// OpenMP loop over rows, _pyco_tree/kd_tree.hpp:117-135), so the tree and its points are REPLICATED
// on every device (143 MB for BASELINE config 2) and a batch is cut into n contiguous row ranges of
// ceil(nq / n) rows, device r taking rows [r * per, (r + 1) * per): the gathered buffer is in the
// caller's row order without any reordering.  There is no data-path collective.
//
//   host buffers    every device uploads its own range, searches it and downloads its rows straight
//                   into the caller's array (one host thread per device): no inter-GPU traffic.
//   device buffers  (queries and results on devices[0]) the ranges go out and the (index, distance)
//                   rows come back over xGMI as grouped ncclSend / ncclRecv pairs -- each peer
//                   reaches devices[0] over its own link, no ring -- on one single-process
//                   communicator per device (ncclCommInitAll).  BASELINE configs[3].
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first device-buffer call that needs it):
// libptk.so itself does not link it, so single-GPU users never load a collective library, and a
// node without RCCL fails loudly in ptk_multi_search_knn_device only.

#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <thread>

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // torch (when it is in the process) has its own copy under the same SONAME: RTLD_NOLOAD first.
    r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (r.lib == nullptr) r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (r.lib == nullptr) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (r.lib == nullptr) {
      r.error = std::string("cannot load librccl.so.1: ") + dlerror();
      return;
    }
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(r.lib, name);
      if (p == nullptr) r.error = std::string("librccl has no ") + name;
      return p;
    };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.ok = r.error.empty();
  });
  return r;
}

#define PTK_NCCL(expr)                                                                             \
  do {                                                                                             \
    ncclResult_t r_ = (expr);                                                                      \
    if (r_ != ncclSuccess) return fail(PTK_ERR_DEVICE, "%s failed: %s", #expr, rccl().GetErrorString(r_)); \
  } while (0)

// An open RCCL group is closed on every way out of the scope: a send / recv that fails half-way must not leave
// the thread's group open (later collectives of the process would be queued into it and never start).
struct NcclGroup {
  Rccl& r;
  bool open = false;
  explicit NcclGroup(Rccl& rccl_) : r(rccl_) {}
  ncclResult_t start() {
    const ncclResult_t rc = r.GroupStart();
    open = rc == ncclSuccess;
    return rc;
  }
  ncclResult_t end() {
    open = false;
    return r.GroupEnd();
  }
  ~NcclGroup() {
    if (open) (void)r.GroupEnd();
  }
};

}  // namespace

struct ptk_multi {
  std::vector<int> devices;
  std::vector<ptk_tree*> trees;       // one replica per device
  std::vector<hipStream_t> streams;   // per device: the peers' work of a device-buffer call
  std::vector<hipEvent_t> done;       // per device: its part of the current call has been enqueued and finished
  std::vector<ncclComm_t> comms;      // made on the first device-buffer call with more than one device
  // staging of the device-buffer form on the peers (grow-only): their range of the queries and their rows
  std::vector<char*> d_q, d_o;
  std::vector<size_t> q_cap, o_cap;
  hipEvent_t ready = nullptr;         // on devices[0]: the caller's queries are complete
  std::mutex mutex;                   // device-buffer calls on one handle are serialised
  uint32_t dim = 0;
  // A device listed more than once (PTK_MULTI_ALLOW_REPLICAS=1, tests on a one-GPU box): every entry still gets its
  // own replica, stream, event and staging, the rows are cut the same way, and the device-buffer form moves the ranges
  // with copies on the entries' streams instead of RCCL (which wants one rank per device).
  bool replicas = false;
};

namespace {

// Rows of device r for a batch of nq rows on n devices: [lo, hi).
inline void shard_rows(uint64_t nq, uint32_t n, uint32_t r, uint64_t* lo, uint64_t* hi) {
  const uint64_t per = (nq + n - 1) / n;
  *lo = std::min<uint64_t>(nq, per * r);
  *hi = std::min<uint64_t>(nq, *lo + per);
}

int multi_comms(ptk_multi* m) {
  if (!m->comms.empty()) return PTK_OK;
  Rccl& r = rccl();
  if (!r.ok) return fail(PTK_ERR_DEVICE, "RCCL is not available: %s", r.error.c_str());
  std::vector<ncclComm_t> comms(m->devices.size());
  PTK_NCCL(r.CommInitAll(comms.data(), (int)m->devices.size(), m->devices.data()));
  m->comms = std::move(comms);
  return PTK_OK;
}

int multi_finish_create(ptk_multi* m, ptk_tree* host, const float* points, ptk_multi** out) {
  ptk_tree_desc d{};
  d.dim = host->dim;
  d.n_points = host->n_points;
  d.points = points;
  d.n_nodes = host->nodes.size();
  d.nodes = host->nodes.data();
  d.indices = host->indices.data();
  d.root_min = host->root_min.data();
  d.root_max = host->root_max.data();
  d.max_depth = host->max_depth;
  m->dim = host->dim;
  int rc = PTK_OK;
  for (int32_t dev : m->devices) g_warmup.start(dev);  // (every device loads the code object beside the first upload)
  for (size_t i = 0; i < m->devices.size() && rc == PTK_OK; ++i) {
    d.device = m->devices[i];
    ptk_tree* t = nullptr;
    rc = ptk_tree_create(&d, &t);
    if (rc != PTK_OK) break;
    m->trees.push_back(t);
    // the two outer bounds per branch (only the topological metrics read them; ptk_multi_set_metric)
    if (host->outer.size() == 2 * host->nodes.size())
      rc = ptk_tree_set_outer_bounds(t, host->outer.data(), host->nodes.size());
    if (rc != PTK_OK) break;
    DeviceGuard guard(m->devices[i]);
    hipStream_t s = nullptr;
    hipEvent_t e = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
      rc = fail(PTK_ERR_DEVICE, "cannot create a stream on device %d", m->devices[i]);
    m->streams.push_back(s);
    m->done.push_back(e);
    m->d_q.push_back(nullptr);
    m->d_o.push_back(nullptr);
    m->q_cap.push_back(0);
    m->o_cap.push_back(0);
  }
  if (rc == PTK_OK) {
    DeviceGuard guard(m->devices[0]);
    if (hipEventCreateWithFlags(&m->ready, hipEventDisableTiming) != hipSuccess)
      rc = fail(PTK_ERR_DEVICE, "cannot create an event on device %d", m->devices[0]);
  }
  ptk_tree_destroy(host);
  if (rc != PTK_OK) {
    const std::string keep = g_error;
    ptk_multi_destroy(m);
    g_error = keep;
    return rc;
  }
  *out = m;
  return PTK_OK;
}

int multi_check_devices(const int32_t* devices, uint32_t n_devices, bool* replicas) {
  *replicas = false;
  if (devices == nullptr || n_devices == 0) return fail(PTK_ERR_INVALID, "empty device list");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(PTK_ERR_DEVICE, "no HIP device is visible");
  const bool allow = env_int("PTK_MULTI_ALLOW_REPLICAS", 0) != 0;
  for (uint32_t i = 0; i < n_devices; ++i) {
    if (devices[i] < 0 || devices[i] >= count)
      return fail(PTK_ERR_INVALID, "device %d out of range (%d visible)", devices[i], count);
    for (uint32_t j = 0; j < i; ++j) {
      if (devices[j] != devices[i]) continue;
      if (!allow) return fail(PTK_ERR_INVALID, "device %d is listed twice", devices[i]);
      *replicas = true;
    }
  }
  return PTK_OK;
}

}  // namespace

extern "C" {

int ptk_multi_create_from_points(const float* points, uint64_t n_points, uint32_t dim, uint64_t max_leaf_size,
                                 const int32_t* devices, uint32_t n_devices, ptk_multi** out) {
  if (out == nullptr) return fail(PTK_ERR_INVALID, "null out pointer");
  *out = nullptr;
  bool replicas = false;
  int rc = multi_check_devices(devices, n_devices, &replicas);
  if (rc != PTK_OK) return rc;
  ptk_tree* host = nullptr;  // built once, on the host; every device gets the same nodes
  rc = ptk_tree_create_from_points(points, n_points, dim, max_leaf_size, PTK_DEVICE_NONE, &host);
  if (rc != PTK_OK) return rc;
  ptk_multi* m = new (std::nothrow) ptk_multi;
  if (m == nullptr) {
    ptk_tree_destroy(host);
    return fail(PTK_ERR_NOMEM, "out of memory");
  }
  m->devices.assign(devices, devices + n_devices);
  m->replicas = replicas;
  return multi_finish_create(m, host, points, out);
}

int ptk_multi_create(const ptk_tree_desc* d, const int32_t* devices, uint32_t n_devices, ptk_multi** out) {
  if (out == nullptr) return fail(PTK_ERR_INVALID, "null out pointer");
  *out = nullptr;
  bool replicas = false;
  int rc = multi_check_devices(devices, n_devices, &replicas);
  if (rc != PTK_OK) return rc;
  if (d == nullptr) return fail(PTK_ERR_INVALID, "null descriptor");
  ptk_tree_desc hd = *d;
  hd.device = PTK_DEVICE_NONE;
  ptk_tree* host = nullptr;  // validates the stream once
  rc = ptk_tree_create(&hd, &host);
  if (rc != PTK_OK) return rc;
  ptk_multi* m = new (std::nothrow) ptk_multi;
  if (m == nullptr) {
    ptk_tree_destroy(host);
    return fail(PTK_ERR_NOMEM, "out of memory");
  }
  m->devices.assign(devices, devices + n_devices);
  m->replicas = replicas;
  return multi_finish_create(m, host, d->points, out);
}

void ptk_multi_destroy(ptk_multi* m) {
  if (m == nullptr) return;
  for (size_t i = 0; i < m->devices.size(); ++i) {
    DeviceGuard guard(m->devices[i]);
    if (i < m->streams.size() && m->streams[i]) {
      (void)hipStreamSynchronize(m->streams[i]);
    }
  }
  if (!m->comms.empty() && rccl().ok)
    for (ncclComm_t c : m->comms) (void)rccl().CommDestroy(c);
  // The replicas go BEFORE the streams: a replica's per-stream scratch block remembers the stream of its last search
  // (the entry's stream, for every entry but the first) and waits for it when it is released.
  for (ptk_tree* t : m->trees) ptk_tree_destroy(t);
  m->trees.clear();
  for (size_t i = 0; i < m->devices.size(); ++i) {
    DeviceGuard guard(m->devices[i]);
    if (i < m->d_q.size() && m->d_q[i]) (void)hipFree(m->d_q[i]);
    if (i < m->d_o.size() && m->d_o[i]) (void)hipFree(m->d_o[i]);
    if (i < m->done.size() && m->done[i]) (void)hipEventDestroy(m->done[i]);
    if (i < m->streams.size() && m->streams[i]) (void)hipStreamDestroy(m->streams[i]);
  }
  if (m->ready) {
    DeviceGuard guard(m->devices[0]);
    (void)hipEventDestroy(m->ready);
  }
  delete m;
}

int ptk_multi_device_count(const ptk_multi* m) { return m == nullptr ? 0 : (int)m->devices.size(); }

int ptk_multi_get_tree(const ptk_multi* m, uint32_t i, const ptk_tree** tree) {
  if (m == nullptr || tree == nullptr || i >= m->trees.size()) return fail(PTK_ERR_INVALID, "bad argument");
  *tree = m->trees[i];
  return PTK_OK;
}

int ptk_multi_set_metric(ptk_multi* m, int metric) {
  if (m == nullptr) return fail(PTK_ERR_INVALID, "null handle");
  std::lock_guard<std::mutex> lock(m->mutex);
  for (ptk_tree* t : m->trees) {
    const int rc = ptk_tree_set_metric(t, metric);
    if (rc != PTK_OK) return rc;
  }
  return PTK_OK;
}

// Host buffers: one host thread per device, each with its contiguous range of rows.
int ptk_multi_search_knn(const ptk_multi* m, const float* q, uint64_t nq, uint32_t k, float e, ptk_neighbor* out) {
  if (m == nullptr) return fail(PTK_ERR_INVALID, "null handle");
  if (nq > 0 && (q == nullptr || out == nullptr)) return fail(PTK_ERR_INVALID, "null buffer");
  if (k == 0) return fail(PTK_ERR_INVALID, "k must be >= 1");
  if (!(e > 0.0f)) return fail(PTK_ERR_INVALID, "approximation ratio e must be > 0");
  const uint32_t n = (uint32_t)m->devices.size();
  std::vector<int> rcs(n, PTK_OK);
  std::vector<std::string> errors(n);
  auto work = [&](uint32_t r) {
    uint64_t lo, hi;
    shard_rows(nq, n, r, &lo, &hi);
    if (hi > lo) {
      rcs[r] = ptk_search_knn(m->trees[r], q + lo * m->dim, hi - lo, k, e, out + lo * k);
      if (rcs[r] != PTK_OK) errors[r] = g_error;  // thread-local: carry it to the caller's thread
    }
  };
  std::vector<std::thread> threads;
  for (uint32_t r = 1; r < n; ++r) threads.emplace_back(work, r);
  work(0);
  for (std::thread& t : threads) t.join();
  for (uint32_t r = 0; r < n; ++r)
    if (rcs[r] != PTK_OK) return fail(rcs[r], "device %d: %s", m->devices[r], errors[r].c_str());
  return PTK_OK;
}

int ptk_multi_search_radius(const ptk_multi* m, const float* q, uint64_t nq, float radius, float e, int sort,
                            uint64_t* offsets, ptk_neighbor** out) {
  if (m == nullptr) return fail(PTK_ERR_INVALID, "null handle");
  if (out == nullptr || offsets == nullptr) return fail(PTK_ERR_INVALID, "null output pointer");
  *out = nullptr;
  if (nq > 0 && q == nullptr) return fail(PTK_ERR_INVALID, "null query buffer");
  const uint32_t n = (uint32_t)m->devices.size();
  // Ragged rows: every device returns its own offsets (from 0) and rows; the ranges are contiguous
  // row ranges, so rank order is row order and the rows only need their offsets shifted.
  std::vector<int> rcs(n, PTK_OK);
  std::vector<std::string> errors(n);
  std::vector<ptk_neighbor*> rows(n, nullptr);
  offsets[0] = 0;
  // Every device fills offsets of its own (from 0); they are shifted into place afterwards.
  std::vector<std::vector<uint64_t>> local(n);
  auto work_local = [&](uint32_t r) {
    uint64_t lo, hi;
    shard_rows(nq, n, r, &lo, &hi);
    local[r].assign(hi - lo + 1, 0);
    if (hi > lo) {
      rcs[r] = ptk_search_radius(m->trees[r], q + lo * m->dim, hi - lo, radius, e, sort, local[r].data(), &rows[r]);
      if (rcs[r] != PTK_OK) errors[r] = g_error;
    }
  };
  std::vector<std::thread> threads;
  for (uint32_t r = 1; r < n; ++r) threads.emplace_back(work_local, r);
  work_local(0);
  for (std::thread& t : threads) t.join();
  int rc = PTK_OK;
  for (uint32_t r = 0; r < n && rc == PTK_OK; ++r)
    if (rcs[r] != PTK_OK) rc = fail(rcs[r], "device %d: %s", m->devices[r], errors[r].c_str());
  uint64_t total = 0;
  if (rc == PTK_OK) {
    for (uint32_t r = 0; r < n; ++r) {
      uint64_t lo, hi;
      shard_rows(nq, n, r, &lo, &hi);
      for (uint64_t i = 0; i < hi - lo; ++i) offsets[lo + i] = total + local[r][i];
      total += local[r][hi - lo];
    }
    offsets[nq] = total;
    ptk_neighbor* all = static_cast<ptk_neighbor*>(std::malloc(std::max<uint64_t>(total, 1) * sizeof(ptk_neighbor)));
    if (all == nullptr) {
      rc = fail(PTK_ERR_NOMEM, "out of memory");
    } else {
      uint64_t at = 0;
      for (uint32_t r = 0; r < n; ++r) {
        const uint64_t cnt = local[r].back();
        if (cnt > 0) std::memcpy(all + at, rows[r], cnt * sizeof(ptk_neighbor));
        at += cnt;
      }
      *out = all;
    }
  }
  for (ptk_neighbor* p : rows) std::free(p);
  return rc;
}

// Device buffers on devices[0]; `stream` is a stream of devices[0] (null: its default stream).
int ptk_multi_search_knn_device(ptk_multi* m, const float* d_q, uint64_t nq, uint32_t k, float e, ptk_neighbor* d_out,
                                void* stream) {
  if (m == nullptr) return fail(PTK_ERR_INVALID, "null handle");
  if (nq > 0 && (d_q == nullptr || d_out == nullptr)) return fail(PTK_ERR_INVALID, "null buffer");
  if (k == 0) return fail(PTK_ERR_INVALID, "k must be >= 1");
  if (!(e > 0.0f)) return fail(PTK_ERR_INVALID, "approximation ratio e must be > 0");
  if (nq == 0) return PTK_OK;
  std::lock_guard<std::mutex> lock(m->mutex);
  const uint32_t n = (uint32_t)m->devices.size();
  // PTK_MULTI_SELF_GATHER=1 (tests on a one-GPU box): devices[0] sends its own rows to itself too,
  // so that the RCCL path runs whatever the number of devices.
  const bool self = knob_int("multi_self_gather", 0) != 0;
  hipStream_t s0 = static_cast<hipStream_t>(stream);  // null = the default stream of devices[0], as everywhere in HIP
  int rc = PTK_OK;
  if (m->replicas) {
    // Entries of one device: the same row ranges, per-entry streams, events and staging; copies instead of RCCL.
    const size_t rq = (size_t)m->dim * sizeof(float), ro = (size_t)k * sizeof(ptk_neighbor);
    DeviceGuard guard(m->devices[0]);
    for (uint32_t r = 1; r < n; ++r) {
      uint64_t lo, hi;
      shard_rows(nq, n, r, &lo, &hi);
      rc = grow_device_block(&m->d_q[r], &m->q_cap[r], std::max<size_t>((hi - lo) * rq, 16));
      if (rc == PTK_OK) rc = grow_device_block(&m->d_o[r], &m->o_cap[r], std::max<size_t>((hi - lo) * ro, 16));
      if (rc != PTK_OK) return rc;
    }
    PTK_HIP(hipEventRecord(m->ready, s0));
    for (uint32_t r = 0; r < n; ++r) {
      uint64_t lo, hi;
      shard_rows(nq, n, r, &lo, &hi);
      if (hi == lo) continue;
      if (r == 0) {
        rc = ptk_search_knn_device(m->trees[0], d_q, hi - lo, k, e, d_out, s0);
      } else {
        DeviceGuard entry(m->devices[r]);
        PTK_HIP(hipStreamWaitEvent(m->streams[r], m->ready, 0));
        PTK_HIP(hipMemcpyAsync(m->d_q[r], d_q + lo * m->dim, (hi - lo) * rq, hipMemcpyDeviceToDevice, m->streams[r]));
        rc = ptk_search_knn_device(m->trees[r], reinterpret_cast<const float*>(m->d_q[r]), hi - lo, k, e,
                                   reinterpret_cast<ptk_neighbor*>(m->d_o[r]), m->streams[r]);
        if (rc == PTK_OK) {
          PTK_HIP(hipMemcpyAsync(d_out + lo * k, m->d_o[r], (hi - lo) * ro, hipMemcpyDeviceToDevice, m->streams[r]));
          PTK_HIP(hipEventRecord(m->done[r], m->streams[r]));
        }
      }
      if (rc != PTK_OK) {
        const std::string keep = g_error;
        for (uint32_t p = 1; p < n; ++p) (void)hipStreamSynchronize(m->streams[p]);
        g_error = keep;
        return rc;
      }
    }
    for (uint32_t r = 1; r < n; ++r) {
      uint64_t lo, hi;
      shard_rows(nq, n, r, &lo, &hi);
      if (hi > lo) PTK_HIP(hipStreamWaitEvent(s0, m->done[r], 0));
    }
    return PTK_OK;
  }
  if (n > 1 || self) {
    rc = multi_comms(m);
    if (rc != PTK_OK) return rc;
  }
  Rccl& nccl = rccl();
  const size_t row_q = (size_t)m->dim * sizeof(float), row_o = (size_t)k * sizeof(ptk_neighbor);
  // Staging on the peers (and on devices[0] for the self-gather form).
  for (uint32_t r = self ? 0 : 1; r < n; ++r) {
    uint64_t lo, hi;
    shard_rows(nq, n, r, &lo, &hi);
    DeviceGuard guard(m->devices[r]);
    rc = grow_device_block(&m->d_q[r], &m->q_cap[r], std::max<size_t>((hi - lo) * row_q, 16));
    if (rc == PTK_OK) rc = grow_device_block(&m->d_o[r], &m->o_cap[r], std::max<size_t>((hi - lo) * row_o, 16));
    if (rc != PTK_OK) return rc;
  }
  {
    DeviceGuard guard(m->devices[0]);
    PTK_HIP(hipEventRecord(m->ready, s0));
  }
  // 1. the ranges go out: grouped send (devices[0]) / recv (peer) pairs
  if (n > 1) {
    for (uint32_t r = 1; r < n; ++r) {
      DeviceGuard guard(m->devices[r]);
      PTK_HIP(hipStreamWaitEvent(m->streams[r], m->ready, 0));
    }
    NcclGroup group(nccl);
    PTK_NCCL(group.start());
    for (uint32_t r = 1; r < n; ++r) {
      uint64_t lo, hi;
      shard_rows(nq, n, r, &lo, &hi);
      if (hi == lo) continue;
      PTK_NCCL(nccl.Send(d_q + lo * m->dim, (hi - lo) * row_q, ncclInt8, (int)r, m->comms[0], s0));
      PTK_NCCL(nccl.Recv(m->d_q[r], (hi - lo) * row_q, ncclInt8, 0, m->comms[r], m->streams[r]));
    }
    PTK_NCCL(group.end());
  }
  // 2. every device searches its range
  for (uint32_t r = 0; r < n; ++r) {
    uint64_t lo, hi;
    shard_rows(nq, n, r, &lo, &hi);
    if (hi == lo) continue;
    const bool staged = r > 0 || self;
    const float* q_r = r == 0 ? d_q : reinterpret_cast<const float*>(m->d_q[r]);
    ptk_neighbor* o_r = staged ? reinterpret_cast<ptk_neighbor*>(m->d_o[r]) : d_out + lo * k;
    rc = ptk_search_knn_device(m->trees[r], q_r, hi - lo, k, e, o_r, r == 0 ? s0 : m->streams[r]);
    if (rc != PTK_OK) {  // the peers may hold received ranges and searches: let them finish before anything is reused
      const std::string keep = g_error;
      for (uint32_t p = 1; p < n; ++p) {
        DeviceGuard guard(m->devices[p]);
        (void)hipStreamSynchronize(m->streams[p]);
      }
      g_error = keep;
      return rc;
    }
  }
  // 3. the rows come back: each peer over its own link into its place of the caller's buffer
  if (n > 1 || self) {
    NcclGroup group(nccl);
    PTK_NCCL(group.start());
    for (uint32_t r = self ? 0 : 1; r < n; ++r) {
      uint64_t lo, hi;
      shard_rows(nq, n, r, &lo, &hi);
      if (hi == lo) continue;
      PTK_NCCL(nccl.Send(m->d_o[r], (hi - lo) * row_o, ncclInt8, 0, m->comms[r], r == 0 ? s0 : m->streams[r]));
      PTK_NCCL(nccl.Recv(d_out + lo * k, (hi - lo) * row_o, ncclInt8, (int)r, m->comms[0], s0));
    }
    PTK_NCCL(group.end());
  }
  // The peers' staging buffers are reused by the next call: it must not start before the caller's
  // stream has received everything, which the next call's `ready` event (recorded on s0) implies.
  return PTK_OK;
}

}  // extern "C"
