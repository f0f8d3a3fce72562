// tests/cpp/emulate_kernels.cpp -- TEST INFRASTRUCTURE.
//
// Compiles the product's device source (pico_tree_amd/csrc/ptk_kernels.hpp) and
// its host-side tree encoder (ptk_encode.hpp) with g++ against the HIP stand-in
// in tests/cpp/hip_stub, and runs every lane of every block sequentially.  This
// lets the CPU-only test tier check the exact kernel logic (record stack, undo
// records, tie handling, k-list insertion, radius passes) against the oracle
// before any GPU time is spent.  Float semantics are IEEE on both sides; what
// this cannot see is scheduling, so the `-m gpu` tests remain the parity proof.

#include <hip/hip_runtime.h>

thread_local dim3 threadIdx;
thread_local dim3 blockIdx;
thread_local dim3 gridDim;
thread_local dim3 blockDim;

#include <ucontext.h>

#include <algorithm>
#include <functional>
#include <sstream>
#include <string>
#include <vector>

#include "ptk.h"
#include "ptk_encode.hpp"
#include "ptk_kernels.hpp"
#include "ptk_kernels_lists.hpp"
#include "ptk_kernels_coopk.hpp"
#include "ptk_kernels_coopr.hpp"
#include "ptk_piles.hpp"
#include "ptk_sort.hpp"
#include "ptk_kernels_nd.hpp"
#include "ptk_kernels_topo.hpp"
#include "ptk_kernels_f64.hpp"
#include "ptk_kernels_coop64.hpp"
#include "pico_tree/internal/flat_tree.hpp"
#include "pico_tree/internal/stream.hpp"
#include "pico_tree/map.hpp"
#include "ptk_forest.hpp"

namespace ptk {
unsigned char ptk_smem[192 * 1024] __attribute__((aligned(16)));
}

// ---- wave-level rendezvous for kernels that use __ballot ---------------------------------
// The 64 lanes of one wavefront run as 64 cooperative fibers (ucontext) on the calling
// thread.  A ballot parks the lane; when every live lane has parked, the OR of the
// predicates is handed to all of them and they resume in lane order.  All lanes of a
// wave call __ballot the same number of times (the persistent kernels leave their loop
// on a wave-uniform condition), which the scheduler asserts.
namespace {
struct WaveFibers {
  static constexpr int kLanes = 64;
  static constexpr size_t kStack = 256 * 1024;
  ucontext_t scheduler;
  ucontext_t lane_ctx[kLanes];
  std::vector<unsigned char> stacks;
  bool done[kLanes];
  bool pred[kLanes];
  int value[kLanes];     // deposited by __shfl before parking
  int snapshot[kLanes];  // the values of the last rendezvous (stable until everyone parks again)
  unsigned long long result = 0;
  int current = -1;
  std::function<void()> body;
};
WaveFibers* g_fibers = nullptr;

void fiber_entry() {
  WaveFibers* w = g_fibers;
  w->body();
  w->done[w->current] = true;
  swapcontext(&w->lane_ctx[w->current], &w->scheduler);
}
}  // namespace

// ---- blocks of several wavefronts ------------------------------------------------------------------------------------
// The threads of one block run as fibers as well; a ballot or shuffle is a rendezvous of the caller's WAVEFRONT (64
// consecutive threads), __syncthreads one of the whole block.  A wavefront whose live lanes have all parked at a ballot
// goes on at once, whatever the other wavefronts are doing; the block goes on when every live thread waits at the barrier.
namespace {
struct BlockFibers {
  static constexpr size_t kStack = 256 * 1024;
  int n = 0;
  ucontext_t scheduler;
  std::vector<ucontext_t> ctx;
  std::vector<unsigned char> stacks;
  std::vector<char> done, runnable, at_barrier, pred;
  std::vector<int> value, snapshot;
  std::vector<unsigned long long> result;  // per wavefront
  int current = -1;
  std::function<void()> body;
};
BlockFibers* g_block = nullptr;

void block_fiber_entry() {
  BlockFibers* b = g_block;
  b->body();
  b->done[b->current] = 1;
  swapcontext(&b->ctx[b->current], &b->scheduler);
}
}  // namespace

void emu_barrier() {
  if (g_block == nullptr) {
    (void)emu_ballot(false);
    return;
  }
  BlockFibers* b = g_block;
  const int t = b->current;
  b->at_barrier[t] = 1;
  swapcontext(&b->ctx[t], &b->scheduler);
}

unsigned long long emu_ballot(bool pred) {
  if (g_block != nullptr) {
    BlockFibers* b = g_block;
    const int t = b->current;
    b->pred[t] = pred;
    b->at_barrier[t] = 0;
    swapcontext(&b->ctx[t], &b->scheduler);
    return b->result[t / 64];
  }
  WaveFibers* w = g_fibers;
  const int lane = w->current;
  w->pred[lane] = pred;
  swapcontext(&w->lane_ctx[lane], &w->scheduler);  // park until everyone has voted
  return w->result;
}

// A shuffle is a rendezvous too: every live lane deposits its value, parks, and reads the
// source lane's value from the snapshot taken when all had parked.
int emu_shfl(int v, int src_lane) {
  if (g_block != nullptr) {
    BlockFibers* b = g_block;
    const int t = b->current;
    b->value[t] = v;
    b->pred[t] = false;
    b->at_barrier[t] = 0;
    swapcontext(&b->ctx[t], &b->scheduler);
    return b->snapshot[(t / 64) * 64 + (src_lane & 63)];
  }
  WaveFibers* w = g_fibers;
  const int lane = w->current;
  w->value[lane] = v;
  w->pred[lane] = false;
  swapcontext(&w->lane_ctx[lane], &w->scheduler);
  return w->snapshot[src_lane & 63];
}

int emu_lane() { return g_block != nullptr ? g_block->current & 63 : g_fibers->current; }

namespace {

thread_local std::string g_err;

struct Emu {
  ptk::EncodedTree enc;
  ptk::EncodedTreeND enc_nd;  // dim > 3
  ptk::TreeStats st;
  ptk::DevTree dev{};
  ptk::DevTreeND dev_nd{};
  uint32_t dim;
  int metric = 0;
  std::vector<float2> outer;  // per branch {left_min, right_max}: topological metrics (emu_set_outer)
  // rows captured by emu_radius_capture (ptk::RadiusCapture)
  std::vector<ptk::Neighbor> cap_chunks;
  std::vector<uint32_t> cap_counters;
  std::vector<uint8_t> cap_flags;
  std::vector<uint32_t> cap_qids;
  std::vector<uint32_t> cap_lens, cap_tables;  // the capture as leaf lists (ptk_kernels_lists.hpp)
  ptk::RadiusHeavy hv{};                       // ... and what a capped list pass handed over (ptk_kernels_coopr.hpp)
  std::vector<uint32_t> hv_meta, hv_words;
  std::vector<uint64_t> hv_entries;
  ptk::PileView piles;     // emu_use_pile_view: the k = 1 view of a tree with piles (ptk_piles.hpp)
  ptk::EncodedTree enc1;
  uint64_t cap_nq = 0;
  ptk::RadiusCapture cap{};
};

// Kernels whose lanes are independent: every lane of every block, sequentially.
template <typename F>
void for_each_lane(uint64_t nq, F&& f, uint32_t block = ptk::kBlock) {
  const uint32_t blocks = (uint32_t)((nq + block - 1) / block);
  gridDim.x = blocks;
  blockDim.x = block;
  for (uint32_t b = 0; b < blocks; ++b) {
    blockIdx.x = b;
    for (uint32_t t = 0; t < block; ++t) {
      threadIdx.x = t;
      f();
    }
  }
}

// Runs `blocks` single-wave blocks, one after the other, 64 fibers each.
template <typename F>
void for_each_wave(uint32_t blocks, F&& f) {
  WaveFibers w;
  w.stacks.resize(WaveFibers::kLanes * WaveFibers::kStack);
  g_fibers = &w;
  gridDim.x = blocks;
  blockDim.x = 64;
  for (uint32_t b = 0; b < blocks; ++b) {
    blockIdx.x = b;
    w.body = [&] { f(); };
    for (int l = 0; l < WaveFibers::kLanes; ++l) {
      w.done[l] = false;
      w.pred[l] = false;
      w.value[l] = 0;
      getcontext(&w.lane_ctx[l]);
      w.lane_ctx[l].uc_stack.ss_sp = w.stacks.data() + (size_t)l * WaveFibers::kStack;
      w.lane_ctx[l].uc_stack.ss_size = WaveFibers::kStack;
      w.lane_ctx[l].uc_link = &w.scheduler;
      makecontext(&w.lane_ctx[l], fiber_entry, 0);
    }
    for (;;) {
      // Run every live lane up to its next ballot (or to completion).
      int live = 0;
      for (int l = 0; l < WaveFibers::kLanes; ++l) {
        if (w.done[l]) continue;
        w.current = l;
        threadIdx.x = (uint32_t)l;
        swapcontext(&w.scheduler, &w.lane_ctx[l]);
        if (!w.done[l]) ++live;
      }
      if (live == 0) break;
      unsigned long long bits = 0;
      for (int l = 0; l < WaveFibers::kLanes; ++l) {
        if (!w.done[l] && w.pred[l]) bits |= 1ull << l;
      }
      w.result = bits;
      for (int l = 0; l < WaveFibers::kLanes; ++l) w.snapshot[l] = w.value[l];
    }
  }
  g_fibers = nullptr;
}

// Runs `blocks` blocks of `threads` threads (a multiple of 64), one after the other, as fibers (see BlockFibers).
template <typename F>
void for_each_block(uint32_t blocks, uint32_t threads, F&& f) {
  BlockFibers b;
  b.n = (int)threads;
  b.ctx.resize(threads);
  b.stacks.resize((size_t)threads * BlockFibers::kStack);
  b.result.assign(threads / 64, 0ull);
  g_block = &b;
  gridDim.x = blocks;
  blockDim.x = threads;
  for (uint32_t blk = 0; blk < blocks; ++blk) {
    blockIdx.x = blk;
    b.body = [&] { f(); };
    b.done.assign(threads, 0);
    b.runnable.assign(threads, 1);
    b.at_barrier.assign(threads, 0);
    b.pred.assign(threads, 0);
    b.value.assign(threads, 0);
    b.snapshot.assign(threads, 0);
    for (uint32_t t = 0; t < threads; ++t) {
      getcontext(&b.ctx[t]);
      b.ctx[t].uc_stack.ss_sp = b.stacks.data() + (size_t)t * BlockFibers::kStack;
      b.ctx[t].uc_stack.ss_size = BlockFibers::kStack;
      b.ctx[t].uc_link = &b.scheduler;
      makecontext(&b.ctx[t], block_fiber_entry, 0);
    }
    for (;;) {
      bool ran = false;
      for (uint32_t t = 0; t < threads; ++t) {
        if (b.done[t] || !b.runnable[t]) continue;
        b.runnable[t] = 0;
        b.current = (int)t;
        threadIdx.x = t;
        swapcontext(&b.scheduler, &b.ctx[t]);
        ran = true;
      }
      uint32_t live = 0, waiting = 0;
      for (uint32_t t = 0; t < threads; ++t) {
        live += b.done[t] ? 0u : 1u;
        waiting += !b.done[t] && b.at_barrier[t] ? 1u : 0u;
      }
      if (live == 0) break;
      bool released = false;
      for (uint32_t w = 0; w < threads / 64; ++w) {  // wavefronts whose live lanes all wait at a ballot / shuffle
        uint32_t wl = 0, wb = 0;
        unsigned long long bits = 0;
        for (uint32_t l = 0; l < 64; ++l) {
          const uint32_t t = w * 64 + l;
          if (b.done[t]) continue;
          ++wl;
          if (!b.at_barrier[t]) {
            ++wb;
            if (b.pred[t]) bits |= 1ull << l;
          }
        }
        if (wl == 0 || wb == 0) continue;
        if (wb != wl) {
          std::fprintf(stderr, "emulator: a wavefront is split between a ballot and a barrier\n");
          std::abort();
        }
        b.result[w] = bits;
        for (uint32_t l = 0; l < 64; ++l) {
          const uint32_t t = w * 64 + l;
          b.snapshot[t] = b.value[t];
          if (!b.done[t]) b.runnable[t] = 1;
        }
        released = true;
      }
      if (!released && waiting == live) {  // the barrier
        for (uint32_t t = 0; t < threads; ++t) {
          if (!b.done[t]) {
            b.at_barrier[t] = 0;
            b.runnable[t] = 1;
          }
        }
        released = true;
      }
      if (!released && !ran) {
        std::fprintf(stderr, "emulator: block of %u threads cannot go on\n", threads);
        std::abort();
      }
    }
  }
  g_block = nullptr;
}

// The launch-order query records {x, y, z, bits(row)} phase 1 has to produce (host restatement).
std::vector<float4> pack(const float* q, uint32_t dim, const uint32_t* perm, uint64_t nq) {
  std::vector<float4> qs(nq ? nq : 1);
  for (uint64_t i = 0; i < nq; ++i) {
    const uint32_t qi = perm ? perm[i] : (uint32_t)i;
    const float* p = q + (uint64_t)qi * dim;
    qs[i] = make_float4(p[0], dim > 1 ? p[1] : 0.0f, dim > 2 ? p[2] : 0.0f, __uint_as_float(qi));
  }
  return qs;
}

}  // namespace

// The launches the backend makes for a metric other than L2 squared: register k-list for k <= 32
// (k = 1 included -- the two-phase search is L2 only), LDS k-list above, any-dimension kernels.
template <class M>
int emu_knn_metric(Emu* t, const float* q, uint64_t nq, uint32_t k, float e_inv, const uint32_t* perm,
                   int small_stack, ptk::Neighbor* o) {
  if (t->dim > 3) {  // as launch_knn_nd: registers for k <= 32, else the list in LDS
    if (k <= 4)
      for_each_lane(nq, [&] { ptk::knn_nd_reg_kernel<4, 4, 2048, M>(t->dev_nd, q, perm, nq, k, e_inv, o); }, 64);
    else if (k <= 32)
      for_each_lane(nq, [&] { ptk::knn_nd_reg_kernel<32, 16, 2048, M>(t->dev_nd, q, perm, nq, k, e_inv, o); }, 64);
    else if (small_stack)
      for_each_lane(nq, [&] { ptk::knn_nd_kernel<4, 2048, true, M>(t->dev_nd, q, perm, nq, k, e_inv, o); }, 64);
    else
      for_each_lane(nq, [&] { ptk::knn_nd_kernel<16, 2048, true, M>(t->dev_nd, q, perm, nq, k, e_inv, o); }, 64);
    return 0;
  }
  if (k <= 32) {
    if (small_stack) {
      if (k <= 4) for_each_lane(nq, [&] { ptk::knn_reg_kernel<4, 4, 2048, 64, 1, M>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
      else for_each_lane(nq, [&] { ptk::knn_reg_kernel<32, 4, 2048, 64, 1, M>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
    } else {
      if (k <= 8) for_each_lane(nq, [&] { ptk::knn_reg_kernel<8, 16, 2048, 64, 4, M>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
      else for_each_lane(nq, [&] { ptk::knn_reg_kernel<32, 16, 2048, 64, 4, M>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
    }
  } else {
    for_each_lane(nq, [&] { ptk::knn_kernel<16, 2048, 64, 4, true, M>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
  }
  return 0;
}

template <class M>
void emu_radius_metric(Emu* t, const float* q, uint64_t nq, float radius, float e_inv, const uint32_t* perm,
                       uint64_t* counts, const uint64_t* offsets, ptk::Neighbor* o) {
  if (t->dim > 3) {
    if (o == nullptr)
      for_each_lane(nq, [&] { ptk::radius_nd_kernel<8, 2048, false, M>(t->dev_nd, q, nq, radius, e_inv, counts, nullptr, nullptr); }, 64);
    else
      for_each_lane(nq, [&] { ptk::radius_nd_kernel<16, 2048, true, M>(t->dev_nd, q, nq, radius, e_inv, nullptr, offsets, o); }, 64);
    return;
  }
  if (o == nullptr)
    for_each_lane(nq, [&] { ptk::radius_kernel<8, 2048, 64, 4, false, M>(t->dev, q, t->dim, perm, nq, radius, e_inv, counts, nullptr, nullptr); }, 64);
  else
    for_each_lane(nq, [&] { ptk::radius_kernel<16, 2048, 64, 4, true, M>(t->dev, q, t->dim, perm, nq, radius, e_inv, nullptr, offsets, o); }, 64);
}

template <class T>
int emu_knn_topo(Emu* t, const float* q, uint64_t nq, uint32_t k, float e_inv, const uint32_t* perm, int small_stack,
                 ptk::Neighbor* o) {
  if (k <= 32) {
    if (small_stack) for_each_lane(nq, [&] { ptk::knn_topo_reg_kernel<32, 4, 2048, T>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
    else if (k <= 8) for_each_lane(nq, [&] { ptk::knn_topo_reg_kernel<8, 16, 2048, T>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
    else for_each_lane(nq, [&] { ptk::knn_topo_reg_kernel<32, 16, 2048, T>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
  } else {
    for_each_lane(nq, [&] { ptk::knn_topo_kernel<16, 2048, T>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
  }
  return 0;
}
template <class T>
void emu_radius_topo(Emu* t, const float* q, uint64_t nq, float radius, float e_inv, const uint32_t* perm,
                     uint64_t* counts, const uint64_t* offsets, ptk::Neighbor* o) {
  if (o == nullptr)
    for_each_lane(nq, [&] { ptk::radius_topo_kernel<4, 2048, false, T>(t->dev, q, t->dim, perm, nq, radius, e_inv, counts, nullptr, nullptr); }, 64);
  else
    for_each_lane(nq, [&] { ptk::radius_topo_kernel<16, 2048, true, T>(t->dev, q, t->dim, perm, nq, radius, e_inv, nullptr, offsets, o); }, 64);
}

// The general k-NN search with its long queries finished cooperatively (ptk_kernels_coopk.hpp): the capped launch
// (lanes one after the other), knn_coop_kernel (64 fibers per wavefront: ballots, shuffles, the shared pool), the
// reference search of what could not be certified.  counts = {queries handed over, queries redone}.
// pool_small != 0: a pool of 64 subtrees (overflows on long searches: the redo path).
template <int K, class M>
int emu_knn_capped_km(Emu* t, const float* q, uint64_t nq, uint32_t k, const uint32_t* perm, uint32_t cap, int pool_small,
                     uint32_t max_heavy, ptk::Neighbor* o, uint32_t* counts) {
  std::vector<uint32_t> meta(ptk::kMetaWords, 0), heavy(nq + 1), ntasks(nq + 1), redo(nq + 1);
  std::vector<ptk::Task> tasks((size_t)std::max<uint32_t>(max_heavy, 1) * ptk::kMaxTasks);
  ptk::Handover ho{};
  ho.counter = ptk::kMetaHeavy;
  ho.meta = meta.data();
  ho.heavy_list = heavy.data();
  ho.ntasks = ntasks.data();
  ho.tasks = tasks.data();
  ho.max_heavy = max_heavy;
  ho.full_keeps = 1u;  // (as launch_knn_reg: a query that finds the list full finishes in its lane)
  for_each_lane(nq, [&] { ptk::knn_reg_kernel<K, 16, 2048, 64, 4, M, true>(t->dev, q, t->dim, perm, nq, k, 1.0f, o, cap, ho); }, 64);
  // (pool_small: a pool of 64 subtrees and 40 spill slots per wavefront -- long searches park subtrees in HBM and a
  // few overflow even that: the redo path)
  const uint32_t spill_cap = pool_small ? 40u : 4096u;
  std::vector<ptk::Task> spill((size_t)3 * spill_cap);
  const auto* ranges = reinterpret_cast<const uint2*>(t->enc.ranges.data());
  for_each_wave(3, [&] {
    if (pool_small) ptk::knn_coop_kernel<K, 64, M>(t->dev, ranges, q, t->dim, k, o, ho, redo.data(), ptk::kMetaRedo, spill.data(), spill_cap);
    else ptk::knn_coop_kernel<K, 128, M>(t->dev, ranges, q, t->dim, k, o, ho, redo.data(), ptk::kMetaRedo, spill.data(), spill_cap);
  });
  for_each_lane(128, [&] { ptk::knn_redo_kernel<K, 16, 2048, 4, M>(t->dev, q, t->dim, k, 1.0f, o, meta.data(), ptk::kMetaRedo, redo.data()); }, 64);
  counts[0] = meta[ptk::kMetaHeavy];
  counts[1] = meta[ptk::kMetaRedo];
  counts[2] = meta[ptk::kKnnTieSweeps];
  return 0;
}

// (the capped k > 1 search is shipped for metric_l2_squared and metric_l1: launch_knn_reg of ptk_family_knn.hip)
template <int K>
int emu_knn_capped_k(Emu* t, const float* q, uint64_t nq, uint32_t k, const uint32_t* perm, uint32_t cap, int pool_small,
                     uint32_t max_heavy, ptk::Neighbor* o, uint32_t* counts) {
  if (t->metric == 1) return emu_knn_capped_km<K, ptk::MetricL1>(t, q, nq, k, perm, cap, pool_small, max_heavy, o, counts);
  return emu_knn_capped_km<K, ptk::MetricL2>(t, q, nq, k, perm, cap, pool_small, max_heavy, o, counts);
}

extern "C" {

const char* emu_last_error() { return g_err.c_str(); }

void* emu_create(const float* points, uint64_t n, uint32_t dim, const ptk_node* nodes, uint64_t n_nodes,
                 const int32_t* indices) {
  auto* e = new Emu;
  bool unsupported = false;
  if (dim > 3) {
    g_err = ptk::encode_tree_nd(dim, n, points, nodes, n_nodes, indices, e->st, e->enc_nd, unsupported);
    if (!g_err.empty()) {
      delete e;
      return nullptr;
    }
    e->dim = dim;
    e->dev_nd.nodes = reinterpret_cast<const uint4*>(e->enc_nd.nodes.data());
    e->dev_nd.axes = e->enc_nd.axes.data();
    e->dev_nd.pts = e->enc_nd.points.data();
    e->dev_nd.index = e->enc_nd.index.data();
    e->dev_nd.root_ref = e->enc_nd.root_ref;
    e->dev_nd.cbits = e->enc_nd.cbits;
    e->dev_nd.cmask = (1u << e->enc_nd.cbits) - 1u;
    e->dev_nd.dim = dim;
    return e;
  }
  g_err = ptk::encode_tree(dim, n, points, nodes, n_nodes, indices, e->st, e->enc, unsupported);
  if (!g_err.empty()) {
    delete e;
    return nullptr;
  }
  e->dim = dim;
  e->dev.nodes = reinterpret_cast<const uint4*>(e->enc.nodes.data());
  e->dev.pts = reinterpret_cast<const float4*>(e->enc.points.data());
  e->dev.root_ref = e->enc.root_ref;
  e->dev.cbits = e->enc.cbits;
  e->dev.cmask = (1u << e->enc.cbits) - 1u;
  e->dev.n_points = (uint32_t)n;
  return e;
}

void emu_destroy(void* h) { delete static_cast<Emu*>(h); }

// Switches the handle to the k = 1 view of its tree (ptk_piles.hpp): branch records and subtree ranges of the stream
// with every pile collapsed, the point records it already has.  Returns the number of piles (0: the tree has none and
// nothing changes) or a negative status.  Only emu_knn1_two_phase + emu_resolve_piles are meaningful afterwards.
int64_t emu_use_pile_view(void* h, const float* points, uint64_t n, const ptk_node* nodes, uint64_t n_nodes,
                          const int32_t* indices) {
  auto* e = static_cast<Emu*>(h);
  if (e->dim > 3) return -3;
  ptk::build_pile_view(e->dim, n, points, nodes, n_nodes, indices, e->piles);
  if (e->piles.empty()) return 0;
  ptk::TreeStats st1;
  bool unsupported = false;
  g_err = ptk::encode_tree(e->dim, n, nullptr, e->piles.nodes.data(), e->piles.nodes.size(), indices, st1, e->enc1, unsupported,
                           /*with_points=*/false, e->piles.single.data(), e->enc.cbits);
  if (!g_err.empty()) return -1;
  e->dev.nodes = reinterpret_cast<const uint4*>(e->enc1.nodes.data());
  e->dev.root_ref = e->enc1.root_ref;
  e->enc.ranges = e->enc1.ranges;
  e->st.max_depth = st1.max_depth;
  return (int64_t)e->piles.piles.size();
}

// The pass over the rows of a k = 1 search on the view (resolve_piles_kernel).
void emu_resolve_piles(void* h, const float* q, uint64_t nq, ptk_neighbor* out) {
  auto* e = static_cast<Emu*>(h);
  if (e->piles.empty()) return;
  ptk::DevPiles piles;
  piles.of_point = e->piles.pile_of_point.data();
  piles.recs = reinterpret_cast<const ptk::DevPileRecord*>(e->piles.piles.data());
  piles.n_points = (uint32_t)e->piles.pile_of_point.size();
  for_each_lane(nq, [&] { ptk::resolve_piles_kernel(q, e->dim, nq, piles, reinterpret_cast<ptk::Neighbor*>(out)); });
}

// 0 L2 squared (default), 1 L1, 2 LPInf, 3 LNInf, 4 SO2, 5 SE2 squared: the metric of the searches that
// follow (ptk_tree_set_metric).  The topological ones need emu_set_outer first.
void emu_set_metric(void* h, int metric) { static_cast<Emu*>(h)->metric = metric; }

// The outer bounds of the host tree (2 floats per node, ptk_tree_get_outer_bounds) in branch order.
int emu_set_outer(void* h, const ptk_node* nodes, uint64_t n_nodes, uint64_t n_points, const float* outer) {
  auto* e = static_cast<Emu*>(h);
  ptk::TreeStats st;
  std::vector<uint32_t> branch_id;
  g_err = ptk::analyse_stream(e->dim, n_points, nodes, n_nodes, st, &branch_id);
  if (!g_err.empty()) return -1;
  e->outer.assign(std::max<size_t>(n_nodes - st.n_leaves, 1), float2{0.0f, 0.0f});
  for (uint64_t i = 0; i < n_nodes; ++i)
    if (nodes[i].right != PTK_LEAF) e->outer[branch_id[i]] = float2{outer[2 * i], outer[2 * i + 1]};
  e->dev.outer = e->outer.data();
  return 0;
}


uint32_t emu_max_depth(void* h) { return static_cast<Emu*>(h)->st.max_depth; }

// small_stack != 0 runs the smallest LDS ring (4 slots), so nearly every record
// takes the spill / refill path; otherwise the shipped geometries are used.
int emu_knn(void* h, const float* q, uint64_t nq, uint32_t k, float e, const uint32_t* perm, int small_stack,
            int list_in_lds, ptk_neighbor* out) {
  auto* t = static_cast<Emu*>(h);
  auto* o = reinterpret_cast<ptk::Neighbor*>(out);
  const float e_inv = 1.0f / e;
  const uint32_t need = 2 * t->st.max_depth + 2;
  if (need > 4 + 2048) return -2;
  if (t->metric == 1) return emu_knn_metric<ptk::MetricL1>(t, q, nq, k, e_inv, perm, small_stack, o);
  if (t->metric == 2) return emu_knn_metric<ptk::MetricLInf>(t, q, nq, k, e_inv, perm, small_stack, o);
  if (t->metric == 3) return emu_knn_metric<ptk::MetricLNInf>(t, q, nq, k, e_inv, perm, small_stack, o);
  if (t->metric == 4) return emu_knn_topo<ptk::TopoSO2>(t, q, nq, k, e_inv, perm, small_stack, o);
  if (t->metric == 5) return emu_knn_topo<ptk::TopoSE2>(t, q, nq, k, e_inv, perm, small_stack, o);
  if (t->dim > 3) {  // any-dimension kernels
    if (list_in_lds == 2) {  // k-list in registers (k <= 32)
      if (k > 32) return -2;
      if (k <= 8) for_each_lane(nq, [&] { ptk::knn_nd_reg_kernel<8, 4, 2048>(t->dev_nd, q, perm, nq, k, e_inv, o); }, 64);
      else if (k <= 16) for_each_lane(nq, [&] { ptk::knn_nd_reg_kernel<16, 16, 2048>(t->dev_nd, q, perm, nq, k, e_inv, o); }, 64);
      else for_each_lane(nq, [&] { ptk::knn_nd_reg_kernel<32, 16, 2048>(t->dev_nd, q, perm, nq, k, e_inv, o); }, 64);
      return 0;
    }
    const size_t base = (size_t)(small_stack ? 4 : 16) * 64 * 8 + (size_t)t->dim * 64 * 8;
    if (base + (list_in_lds ? (size_t)k * 64 * 8 : 0) > sizeof(ptk::ptk_smem)) return -2;
    if (small_stack && list_in_lds)
      for_each_lane(nq, [&] { ptk::knn_nd_kernel<4, 2048, true>(t->dev_nd, q, perm, nq, k, e_inv, o); }, 64);
    else if (small_stack)
      for_each_lane(nq, [&] { ptk::knn_nd_kernel<4, 2048, false>(t->dev_nd, q, perm, nq, k, e_inv, o); }, 64);
    else if (list_in_lds)
      for_each_lane(nq, [&] { ptk::knn_nd_kernel<16, 2048, true>(t->dev_nd, q, perm, nq, k, e_inv, o); }, 64);
    else
      for_each_lane(nq, [&] { ptk::knn_nd_kernel<16, 2048, false>(t->dev_nd, q, perm, nq, k, e_inv, o); }, 64);
    return 0;
  }
  if (k == 1 || list_in_lds == 2) {  // k-list in registers (k <= 32; what the backend launches for k = 1 too)
    if (k > 32) return -2;
    if (small_stack) {
      if (k <= 4) for_each_lane(nq, [&] { ptk::knn_reg_kernel<4, 4, 2048, 64, 1>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
      else if (k <= 16) for_each_lane(nq, [&] { ptk::knn_reg_kernel<16, 4, 2048, 64, 1>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
      else for_each_lane(nq, [&] { ptk::knn_reg_kernel<32, 4, 2048, 64, 1>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
    } else {
      if (k <= 8) for_each_lane(nq, [&] { ptk::knn_reg_kernel<8, 16, 2048, 64, 4>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
      else if (k <= 16) for_each_lane(nq, [&] { ptk::knn_reg_kernel<16, 16, 2048, 64, 4>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
      else for_each_lane(nq, [&] { ptk::knn_reg_kernel<32, 16, 2048, 64, 4>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
    }
  } else if (list_in_lds) {
    if ((size_t)(16 + k) * 64 * 8 > sizeof(ptk::ptk_smem)) return -2;
    if (small_stack)
      for_each_lane(nq, [&] { ptk::knn_kernel<4, 2048, 64, 4, true>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
    else
      for_each_lane(nq, [&] { ptk::knn_kernel<16, 2048, 64, 4, true>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 64);
  } else {
    for_each_lane(nq, [&] { ptk::knn_kernel<16, 2048, 256, 8, false>(t->dev, q, t->dim, perm, nq, k, e_inv, o); }, 256);
  }
  return 0;
}

int emu_knn_capped(void* h, const float* q, uint64_t nq, uint32_t k, const uint32_t* perm, uint32_t cap, int pool_small,
                   uint32_t max_heavy, ptk_neighbor* out, uint32_t* counts) {
  auto* t = static_cast<Emu*>(h);
  auto* o = reinterpret_cast<ptk::Neighbor*>(out);
  if (t->dim > 3 || t->metric > 1 || k < 1 || k > 64) return -2;  // (metric_l2_squared or metric_l1)
  if (2 * t->st.max_depth + 2 > 16 + 2048) return -2;
  if (k <= 4) return emu_knn_capped_k<4>(t, q, nq, k, perm, cap, pool_small, max_heavy, o, counts);
  if (k <= 8) return emu_knn_capped_k<8>(t, q, nq, k, perm, cap, pool_small, max_heavy, o, counts);
  if (k <= 16) return emu_knn_capped_k<16>(t, q, nq, k, perm, cap, pool_small, max_heavy, o, counts);
  if (k <= 32) return emu_knn_capped_k<32>(t, q, nq, k, perm, cap, pool_small, max_heavy, o, counts);
  return emu_knn_capped_k<64>(t, q, nq, k, perm, cap, pool_small, max_heavy, o, counts);
}

int emu_radius_count(void* h, const float* q, uint64_t nq, float radius, float e, const uint32_t* perm,
                     uint64_t* counts) {
  auto* t = static_cast<Emu*>(h);
  const float e_inv = 1.0f / e;
  if (t->metric != 0) {
    if (t->metric == 4) emu_radius_topo<ptk::TopoSO2>(t, q, nq, radius, e_inv, perm, counts, nullptr, nullptr);
    else if (t->metric == 5) emu_radius_topo<ptk::TopoSE2>(t, q, nq, radius, e_inv, perm, counts, nullptr, nullptr);
    else if (t->metric == 1) emu_radius_metric<ptk::MetricL1>(t, q, nq, radius, e_inv, perm, counts, nullptr, nullptr);
    else if (t->metric == 3) emu_radius_metric<ptk::MetricLNInf>(t, q, nq, radius, e_inv, perm, counts, nullptr, nullptr);
    else emu_radius_metric<ptk::MetricLInf>(t, q, nq, radius, e_inv, perm, counts, nullptr, nullptr);
    return 0;
  }
  if (t->dim > 3) {
    for_each_lane(nq, [&] {
      ptk::radius_nd_kernel<8, 2048, false>(t->dev_nd, q, nq, radius, e_inv, counts, nullptr, nullptr);
    }, 64);
    return 0;
  }
  for_each_lane(nq, [&] {
    ptk::radius_kernel<8, 2048, 64, 4, false>(t->dev, q, t->dim, perm, nq, radius, e_inv, counts, nullptr, nullptr);
  }, 64);
  return 0;
}

int emu_radius_fill(void* h, const float* q, uint64_t nq, float radius, float e, const uint32_t* perm,
                    const uint64_t* offsets, ptk_neighbor* out, int sort) {
  auto* t = static_cast<Emu*>(h);
  auto* o = reinterpret_cast<ptk::Neighbor*>(out);
  const float e_inv = 1.0f / e;
  if (t->metric != 0) {
    if (t->metric == 4) emu_radius_topo<ptk::TopoSO2>(t, q, nq, radius, e_inv, perm, nullptr, offsets, o);
    else if (t->metric == 5) emu_radius_topo<ptk::TopoSE2>(t, q, nq, radius, e_inv, perm, nullptr, offsets, o);
    else if (t->metric == 1) emu_radius_metric<ptk::MetricL1>(t, q, nq, radius, e_inv, perm, nullptr, offsets, o);
    else if (t->metric == 3) emu_radius_metric<ptk::MetricLNInf>(t, q, nq, radius, e_inv, perm, nullptr, offsets, o);
    else emu_radius_metric<ptk::MetricLInf>(t, q, nq, radius, e_inv, perm, nullptr, offsets, o);
    if (sort) for_each_lane(nq, [&] { ptk::sort_rows_kernel(nq, offsets, o); });
    return 0;
  }
  if (t->dim > 3) {
    for_each_lane(nq, [&] {
      ptk::radius_nd_kernel<16, 2048, true>(t->dev_nd, q, nq, radius, e_inv, nullptr, offsets, o);
    }, 64);
    if (sort) for_each_lane(nq, [&] { ptk::sort_rows_kernel(nq, offsets, o); });
    return 0;
  }
  for_each_lane(nq, [&] {
    ptk::radius_kernel<16, 2048, 64, 4, true>(t->dev, q, t->dim, perm, nq, radius, e_inv, nullptr, offsets, o);
  }, 64);
  if (sort) for_each_lane(nq, [&] { ptk::sort_rows_kernel(nq, offsets, o); });
  return 0;
}

// Slots of a chunk of the capture log (header and masks included).
int emu_log_chunk() { return (int)ptk::kLogChunk; }

// The capturing count pass (L2 family): sub_cap dynamic chunks per sub-pool -- 0 leaves only the static first
// chunk of every wavefront, so wavefronts with more hits than it holds take the re-traversal path.
int emu_radius_capture(void* h, const float* q, uint64_t nq, float radius, float e, const uint32_t* perm,
                       uint64_t* counts, uint32_t sub_cap) {
  auto* t = static_cast<Emu*>(h);
  if (t->metric != 0) return -3;
  const size_t waves = (size_t)((nq + 63) / 64);
  t->cap_chunks.assign((waves + (size_t)sub_cap * ptk::kCapSubPools) * ptk::kLogChunk, ptk::Neighbor{-1, -1.0f});
  t->cap_counters.assign(ptk::kCapSubPools * ptk::kCapCounterStride, 0u);
  t->cap_flags.assign(waves, 2);
  t->cap_qids.assign(waves * 64, 0u);
  t->cap.chunks = t->cap_chunks.data();
  t->cap.counters = t->cap_counters.data();
  t->cap.captured = t->cap_flags.data();
  t->cap.qids = t->cap_qids.data();
  t->cap.n_static = (uint32_t)waves;
  t->cap.sub_cap = sub_cap;
  t->cap_nq = nq;
  const float e_inv = 1.0f / e;
  if (t->dim > 3) {
    for_each_lane(waves * 64, [&] {
      ptk::radius_nd_capture_kernel<8, 2048>(t->dev_nd, q, perm, nq, radius, e_inv, counts, t->cap);
    }, 64);
    return 0;
  }
  for_each_lane(waves * 64, [&] {
    ptk::radius_capture_kernel<8, 2048, 64, 4>(t->dev, q, t->dim, perm, nq, radius, e_inv, counts, t->cap);
  }, 64);
  return 0;
}

// The fill pass of a captured batch: copy, then the ordinary fill kernel over the listed rows.
// Returns the number of rows that had to be searched again (or a negative status).
int64_t emu_radius_fill_captured(void* h, const float* q, uint64_t nq, float radius, float e,
                                 const uint64_t* offsets, ptk_neighbor* out, int sort) {
  auto* t = static_cast<Emu*>(h);
  auto* o = reinterpret_cast<ptk::Neighbor*>(out);
  if (t->cap_nq != nq || t->cap_flags.empty()) return -1;
  std::vector<uint32_t> over(nq + 1, 0u);
  uint32_t n_over = 0;
  const float e_inv = 1.0f / e;
  for_each_lane((uint64_t)t->cap.n_static * 64,
                [&] { ptk::radius_log_scatter_kernel<4>(t->cap, offsets, o, over.data(), &n_over); }, 256);
  if (t->dim > 3) {
    for_each_lane(nq, [&] {
      ptk::radius_nd_kernel<16, 2048, true>(t->dev_nd, q, nq, radius, e_inv, nullptr, offsets, o, over.data(), &n_over);
    }, 64);
  } else {
    for_each_lane(nq, [&] {
      ptk::radius_kernel<16, 2048, 64, 4, true>(t->dev, q, t->dim, over.data(), nq, radius, e_inv, nullptr, offsets, o,
                                                &n_over);
    }, 64);
  }
  if (sort) for_each_lane(nq, [&] { ptk::sort_rows_kernel(nq, offsets, o); });
  return (int64_t)n_over;
}

// The radius search with the rows made from leaf lists (ptk_kernels_lists.hpp).  Count pass (fill = 0): lane by lane
// (the lanes of a wavefront share the chunk table).  Fill pass: 64 fibers per wavefront (ballots, the ring), then the
// ordinary fill kernel for the queries of wavefronts whose lists were lost; returns their number (or a negative status).
int64_t emu_radius_lists(void* h, const float* q, uint64_t nq, float radius, float e, const uint32_t* perm,
                         uint32_t sub_cap, int fill, uint64_t* counts, const uint64_t* offsets, ptk_neighbor* out) {
  auto* t = static_cast<Emu*>(h);
  if (t->dim > 3) return -3;
  const size_t waves = (size_t)((nq + 63) / 64);
  const float e_inv = 1.0f / e;
  auto* o = reinterpret_cast<ptk::Neighbor*>(out);
  if (!fill) {
    t->cap_chunks.assign((waves + (size_t)sub_cap * ptk::kCapSubPools) * ptk::kLogChunk, ptk::Neighbor{-1, -1.0f});
    t->cap_counters.assign(ptk::kCapSubPools * ptk::kCapCounterStride, 0u);
    t->cap_flags.assign(waves, 2);
    t->cap_qids.assign(waves * 64, 0u);
    t->cap_lens.assign(waves * 64, 0u);
    t->cap_tables.assign(waves * ptk::kListMaxChunks, 0xDEADBEEFu);
    t->cap.chunks = t->cap_chunks.data();
    t->cap.counters = t->cap_counters.data();
    t->cap.captured = t->cap_flags.data();
    t->cap.qids = t->cap_qids.data();
    t->cap.lens = t->cap_lens.data();
    t->cap.tables = t->cap_tables.data();
    t->cap.n_static = (uint32_t)waves;
    t->cap.sub_cap = sub_cap;
    t->cap_nq = nq;
    // (the instantiations the backend picks between: leaves of more than 32 points or not, e = 1 or not)
    const bool big = t->dev.cmask >= ptk::kListMaskBits, exact = e_inv == 1.0f;
    for_each_lane(waves * 64, [&] {
      if (t->metric == 1) ptk::radius_list_kernel<8, 2048, 4, ptk::MetricL1>(t->dev, q, t->dim, perm, nq, radius, e_inv, counts, t->cap);
      else if (big && exact) ptk::radius_list_kernel<8, 2048, 4, ptk::MetricL2, true, true>(t->dev, q, t->dim, perm, nq, radius, e_inv, counts, t->cap);
      else if (big) ptk::radius_list_kernel<8, 2048, 4, ptk::MetricL2, true, false>(t->dev, q, t->dim, perm, nq, radius, e_inv, counts, t->cap);
      else if (exact) ptk::radius_list_kernel<8, 2048, 4, ptk::MetricL2, false, true>(t->dev, q, t->dim, perm, nq, radius, e_inv, counts, t->cap);
      else ptk::radius_list_kernel<8, 2048, 4, ptk::MetricL2, false, false>(t->dev, q, t->dim, perm, nq, radius, e_inv, counts, t->cap);
    }, 64);
    return 0;
  }
  if (t->cap_nq != nq || t->cap_lens.empty()) return -1;
  std::vector<uint32_t> over(nq + 1, 0u);
  uint32_t n_over = 0;
  for_each_wave((uint32_t)waves, [&] {
    if (t->metric == 1) ptk::radius_replay_kernel<3, 32, ptk::MetricL1>(t->dev, q, t->dim, e_inv, t->cap, offsets, o, over.data(), &n_over);
    else ptk::radius_replay_kernel<5, 16>(t->dev, q, t->dim, e_inv, t->cap, offsets, o, over.data(), &n_over);
  });
  for_each_lane(nq, [&] {
    if (t->metric == 1) ptk::radius_kernel<16, 2048, 64, 4, true, ptk::MetricL1>(t->dev, q, t->dim, over.data(), nq, radius, e_inv, nullptr, offsets, o, &n_over);
    else ptk::radius_kernel<16, 2048, 64, 4, true>(t->dev, q, t->dim, over.data(), nq, radius, e_inv, nullptr, offsets, o, &n_over);
  }, 64);
  return (int64_t)n_over;
}

// The radius search with the long queries of the list pass finished by a wavefront each (ptk_kernels_coopr.hpp): the
// capped list pass, the cooperative count of what it handed over, the recount of what that could not finish; then the
// replay of the lists, the cooperative replay of the handed-over tails and the ordinary fill kernel for what was
// lost.  pool_small != 0: a pool of 64 subtrees and 8 spill slots (overflows on long searches: the redo path).
// stats (fill pass) = {queries handed over, rows recounted from the root, rows filled by radius_kernel<FILL>}.
int64_t emu_radius_lists_capped(void* h, const float* q, uint64_t nq, float radius, float e, const uint32_t* perm,
                                uint32_t sub_cap, uint32_t far_cap, int pool_small, uint32_t max_heavy, uint32_t entry_cap,
                                int fill, uint64_t* counts, const uint64_t* offsets, ptk_neighbor* out, uint32_t* stats) {
  auto* t = static_cast<Emu*>(h);
  if (t->dim > 3) return -3;
  const size_t waves = (size_t)((nq + 63) / 64);
  const float e_inv = 1.0f / e;
  auto* o = reinterpret_cast<ptk::Neighbor*>(out);
  if (!fill) {
    t->cap_chunks.assign((waves + (size_t)sub_cap * ptk::kCapSubPools) * ptk::kLogChunk, ptk::Neighbor{-1, -1.0f});
    t->cap_counters.assign(ptk::kCapSubPools * ptk::kCapCounterStride, 0u);
    t->cap_flags.assign(waves, 2);
    t->cap_qids.assign(waves * 64, 0u);
    t->cap_lens.assign(waves * 64, 0u);
    t->cap_tables.assign(waves * ptk::kListMaxChunks, 0xDEADBEEFu);
    t->cap.chunks = t->cap_chunks.data();
    t->cap.counters = t->cap_counters.data();
    t->cap.captured = t->cap_flags.data();
    t->cap.qids = t->cap_qids.data();
    t->cap.lens = t->cap_lens.data();
    t->cap.tables = t->cap_tables.data();
    t->cap.n_static = (uint32_t)waves;
    t->cap.sub_cap = sub_cap;
    t->cap_nq = nq;
    t->hv_meta.assign(ptk::kMetaWords, 0u);
    t->hv_words.assign((size_t)4 * std::max<uint32_t>(max_heavy, 1), 0xDEADBEEFu);
    t->hv_entries.assign(std::max<uint32_t>(entry_cap, 1), ~0ull);
    t->hv.meta = t->hv_meta.data();
    t->hv.rows = t->hv_words.data();
    t->hv.own = t->hv_words.data() + max_heavy;
    t->hv.run_at = t->hv_words.data() + 2 * (size_t)max_heavy;
    t->hv.run_n = t->hv_words.data() + 3 * (size_t)max_heavy;
    t->hv.entries = reinterpret_cast<unsigned long long*>(t->hv_entries.data());
    t->hv.max_heavy = max_heavy;
    t->hv.entry_cap = entry_cap;
    std::vector<uint32_t> heavy(max_heavy + 1), ntasks(max_heavy + 1), redo(max_heavy + 1);
    std::vector<ptk::Task> tasks((size_t)std::max<uint32_t>(max_heavy, 1) * ptk::kMaxTasks);
    ptk::Handover ho{};
    ho.counter = ptk::kMetaHeavy;
    ho.meta = t->hv_meta.data();
    ho.heavy_list = heavy.data();
    ho.ntasks = ntasks.data();
    ho.tasks = tasks.data();
    ho.max_heavy = max_heavy;
    ho.full_keeps = 1u;
    const bool big = t->dev.cmask >= ptk::kListMaskBits, exact = e_inv == 1.0f;
    for_each_lane(waves * 64, [&] {
      if (t->metric == 1) ptk::radius_list_kernel<8, 2048, 4, ptk::MetricL1, true, false, true>(t->dev, q, t->dim, perm, nq, radius, e_inv, counts, t->cap, far_cap, ho);
      else if (big && exact) ptk::radius_list_kernel<8, 2048, 4, ptk::MetricL2, true, true, true>(t->dev, q, t->dim, perm, nq, radius, e_inv, counts, t->cap, far_cap, ho);
      else if (big) ptk::radius_list_kernel<8, 2048, 4, ptk::MetricL2, true, false, true>(t->dev, q, t->dim, perm, nq, radius, e_inv, counts, t->cap, far_cap, ho);
      else if (exact) ptk::radius_list_kernel<8, 2048, 4, ptk::MetricL2, false, true, true>(t->dev, q, t->dim, perm, nq, radius, e_inv, counts, t->cap, far_cap, ho);
      else ptk::radius_list_kernel<8, 2048, 4, ptk::MetricL2, false, false, true>(t->dev, q, t->dim, perm, nq, radius, e_inv, counts, t->cap, far_cap, ho);
    }, 64);
    const uint32_t spill_cap = pool_small ? 8u : 2048u;
    std::vector<char> spill((size_t)3 * spill_cap * 32);
    auto* sp = reinterpret_cast<ptk::Task*>(spill.data());
    for_each_wave(3, [&] {
      if (t->metric == 1) ptk::radius_coop_count_kernel<128, ptk::MetricL1, false>(t->dev, q, t->dim, radius, e_inv, counts, ho, t->hv, redo.data(), sp, spill_cap);
      else if (pool_small && exact) ptk::radius_coop_count_kernel<64, ptk::MetricL2, true>(t->dev, q, t->dim, radius, e_inv, counts, ho, t->hv, redo.data(), sp, spill_cap);
      else if (pool_small) ptk::radius_coop_count_kernel<64, ptk::MetricL2, false>(t->dev, q, t->dim, radius, e_inv, counts, ho, t->hv, redo.data(), sp, spill_cap);
      else if (exact) ptk::radius_coop_count_kernel<128, ptk::MetricL2, true>(t->dev, q, t->dim, radius, e_inv, counts, ho, t->hv, redo.data(), sp, spill_cap);
      else ptk::radius_coop_count_kernel<128, ptk::MetricL2, false>(t->dev, q, t->dim, radius, e_inv, counts, ho, t->hv, redo.data(), sp, spill_cap);
    });
    for_each_lane(max_heavy, [&] {
      if (t->metric == 1) ptk::radius_kernel<8, 2048, 64, 4, false, ptk::MetricL1>(t->dev, q, t->dim, redo.data(), max_heavy, radius, e_inv, counts, nullptr, nullptr, t->hv_meta.data() + ptk::kMetaRedo);
      else ptk::radius_kernel<8, 2048, 64, 4, false>(t->dev, q, t->dim, redo.data(), max_heavy, radius, e_inv, counts, nullptr, nullptr, t->hv_meta.data() + ptk::kMetaRedo);
    }, 64);
    return 0;
  }
  if (t->cap_nq != nq || t->cap_lens.empty() || t->hv_meta.empty()) return -1;
  std::vector<uint32_t> over(nq + t->hv.max_heavy + 1, 0u);
  uint32_t n_over = 0;
  for_each_wave((uint32_t)waves, [&] {
    if (t->metric == 1) ptk::radius_replay_kernel<3, 32, ptk::MetricL1>(t->dev, q, t->dim, e_inv, t->cap, offsets, o, over.data(), &n_over);
    else ptk::radius_replay_kernel<5, 16>(t->dev, q, t->dim, e_inv, t->cap, offsets, o, over.data(), &n_over);
  });
  for_each_wave(3, [&] {
    if (t->metric == 1) ptk::radius_coop_replay_kernel<ptk::MetricL1>(t->dev, q, t->dim, e_inv, t->hv, offsets, o, over.data(), &n_over);
    else ptk::radius_coop_replay_kernel<ptk::MetricL2>(t->dev, q, t->dim, e_inv, t->hv, offsets, o, over.data(), &n_over);
  });
  for_each_lane(nq + t->hv.max_heavy, [&] {
    if (t->metric == 1) ptk::radius_kernel<16, 2048, 64, 4, true, ptk::MetricL1>(t->dev, q, t->dim, over.data(), nq + t->hv.max_heavy, radius, e_inv, nullptr, offsets, o, &n_over);
    else ptk::radius_kernel<16, 2048, 64, 4, true>(t->dev, q, t->dim, over.data(), nq + t->hv.max_heavy, radius, e_inv, nullptr, offsets, o, &n_over);
  }, 64);
  if (stats != nullptr) {
    stats[0] = std::min(t->hv_meta[ptk::kMetaHeavy], t->hv.max_heavy);
    stats[1] = t->hv_meta[ptk::kMetaRedo];
    stats[2] = n_over;
  }
  return (int64_t)n_over;
}

// The two-phase k = 1 search: phase 1 (ballots: lanes run as fibers), sort by continuation key (stable
// sort standing in for the device radix pass), phase 2.  variant:
//   3  the form of an approximate search: no cap, full key order, one narrow tier
//   4  as 3 with one-point leaf batches, 4-slot rings (spill paths) and three narrow tiers (1, 4 and 16
//      lanes per wave)
//   5  as 3, phase 2 capped at 2 far children per query, the rest through the cooperative search
//      (16 lanes per query) and the redo pass; room for the stacks of a part of the queries only
//   6  cap 1, 64 lanes per query, no room for stacks: every search starts again from the root
//   7  cap 1, 8 lanes per query            8  cap 3, 32 lanes per query
// emu_last_coop(): {queries phase 2 gave up on, queries the cooperative search could not certify}.
static uint32_t g_last_heavy = 0, g_last_redo = 0;
uint32_t g_last_spilled = 0;
uint32_t emu_last_spilled() { return g_last_spilled; }  // spill slots the cooperative launches of variant 9 wrote
void emu_last_coop(uint32_t* heavy, uint32_t* redo) {
  *heavy = g_last_heavy;
  *redo = g_last_redo;
}
}  // extern "C"

template <class M>
int emu_knn1_two_phase_m(void* h, const float* q, uint64_t nq, float e, const uint32_t* perm, int variant,
                       ptk_neighbor* out) {
  auto* t = static_cast<Emu*>(h);
  auto* o = reinterpret_cast<ptk::Neighbor*>(out);
  const float e_inv = 1.0f / e;
  if (nq == 0) return 0;
  if (2 * t->st.max_depth + 2 > 4 + 2048) return -2;
  std::vector<float4> qs = pack(q, t->dim, perm, nq);
  std::vector<ptk::Record> crec(nq * ptk::kContSlots + 8);
  std::vector<uint4> cbest(nq);
  std::vector<ptk::ContKey> ckey(nq, 0xEEEE);
  std::vector<uint32_t> cids(nq, 0xEEEEEEEEu), meta(ptk::kMetaWords, 0);
  ptk::Cont cont{cbest.data(), crec.data(), ckey.data(), cids.data(), meta.data(), nq};
  if (variant < 3 || variant > 9) return -3;
  // the class order of the capped variants: tile counts from phase 1, scanned in segments of 4 tiles (so that
  // small batches exercise several segments), chunks of 192 slots
  const uint32_t ntiles = (uint32_t)((nq + 63) / 64), cstride = (ntiles + 3u) & ~3u;
  const uint32_t cseg = 4u * std::max<uint32_t>(1u, (ntiles + 4u * ptk::kClassMaxSegs - 1u) / (4u * ptk::kClassMaxSegs));
  const uint32_t csegs = (ntiles + cseg - 1u) / cseg;
  std::vector<uint32_t> tile_counts((size_t)ptk::kClassBuckets * cstride, 0xEEEEEEEEu);
  {
    std::vector<float4> packed(nq);  // written by the kernel itself
    for_each_wave((uint32_t)((nq + 63) / 64), [&] {
      if (variant >= 5)
        ptk::knn1_phase1u_kernel<4, M>(t->dev, q, t->dim, perm, nq, e_inv, o, cont, packed.data(), tile_counts.data(), cstride);
      else if (variant != 4)
        ptk::knn1_phase1u_kernel<4, M>(t->dev, q, t->dim, perm, nq, e_inv, o, cont, packed.data());
      else
        ptk::knn1_phase1u_kernel<1, M>(t->dev, q, t->dim, perm, nq, e_inv, o, cont, packed.data());
    });
    for (uint64_t i = 0; i < nq; ++i) {
      if (std::memcmp(&packed[i], &qs[i], sizeof(float4)) != 0) return -4;
    }
  }
  // stable sort by key (what the device's radix pass does)
  std::vector<uint32_t> sorted(nq);
  std::vector<ptk::ContKey> sorted_key(nq);
  {
    std::vector<uint32_t> order(nq);
    for (uint64_t i = 0; i < nq; ++i) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ckey[a] < ckey[b]; });
    for (uint64_t i = 0; i < nq; ++i) {
      sorted[i] = cids[order[i]];
      sorted_key[i] = ckey[order[i]];
    }
  }
  const uint32_t top_extra = variant == 4 ? (uint32_t)nq + 2u : (uint32_t)(nq / 64) + 2u;
  ptk::TierSpec tiers{};
  if (variant == 4) {
    tiers.permille[0] = 200; tiers.lanes[0] = 1;
    tiers.permille[1] = 500; tiers.lanes[1] = 4;
    tiers.permille[2] = 800; tiers.lanes[2] = 16;
  } else if (variant == 3) {  // what the backend uses without the cap
    tiers.permille[0] = 60; tiers.lanes[0] = 4;
  }
  if (variant >= 5) {  // the shipped class order: counting sort over the three class bits + tier table from the counters
    const uint32_t per = 192, chunks = (uint32_t)((nq + per - 1) / per);
    {
      uint64_t run = 0;  // the counts of phase 1 cover every slot once
      for (uint32_t b = 0; b < ptk::kClassBuckets; ++b)
        for (uint32_t tl = 0; tl < ntiles; ++tl) run += tile_counts[(size_t)b * cstride + tl];
      if (run != nq) return -6;
    }
    std::vector<uint32_t> seg_totals((size_t)ptk::kClassBuckets * ptk::kClassMaxSegs, 0xEEEEEEEEu);
    for_each_wave(ptk::kClassBuckets * csegs, [&] {
      ptk::class_scan_kernel(tile_counts.data(), ntiles, cstride, cseg, seg_totals.data());
    });
    std::vector<uint32_t> by_count(nq, 0xEEEEEEEEu);
    for_each_wave(chunks, [&] {
      ptk::class_order_kernel(ckey.data(), (uint32_t)nq, per, tile_counts.data(), cstride, cseg, csegs, seg_totals.data(),
                              by_count.data(), cont, tiers, top_extra, variant == 9 ? 1u : 0u);
    });
    for (uint64_t i = 0; i < nq; ++i) {
      // the same permutation as the stable sort on the class bits (the light classes keep their order)
      if ((sorted_key[i] >> 13) != (ckey[by_count[i]] >> 13)) return -7;
    }
    std::vector<uint32_t> by_class(nq);
    {
      std::vector<uint32_t> order(nq);
      for (uint64_t i = 0; i < nq; ++i) order[i] = (uint32_t)i;
      std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return (ckey[a] >> 13) < (ckey[b] >> 13); });
      for (uint64_t i = 0; i < nq; ++i) by_class[i] = cids[order[i]];
    }
    if (by_class != by_count) return -8;
    sorted = by_count;
  } else {
    gridDim.x = 1;
    blockIdx.x = 0;
    threadIdx.x = 0;
    ptk::knn1_phase_meta_kernel(sorted_key.data(), (uint32_t)nq, cont, tiers, top_extra);
  }
  const uint32_t blocks = (uint32_t)((nq + 63) / 64) + 1 + top_extra;
  const uint32_t cap = (variant == 5 || variant == 9) ? 2u : (variant == 6 || variant == 7) ? 1u : variant == 8 ? 3u : 0u;
  std::vector<uint32_t> heavy_list(nq, 0xEEEEEEEEu), redo_list(nq, 0xEEEEEEEEu), ntasks(nq, 0xEEEEEEEEu);
  // Room for the stacks of two thirds of the queries handed over at most: the rest starts from the root.
  const uint32_t max_heavy = variant == 6 ? 0u : (uint32_t)(nq / 6 + 1);
  std::vector<ptk::Task> tasks((size_t)max_heavy * ptk::kMaxTasks + 1);
  ptk::Handover ho{ptk::kMetaHeavy, meta.data(), heavy_list.data(), ntasks.data(), tasks.data(), max_heavy, 0u};
  const auto* ranges = reinterpret_cast<const uint2*>(t->enc.ranges.data());
  uint32_t direct_listed = 0;
  // What the groups park in HBM: 3 waves x 4 groups (x 2 for the narrow variants) of `spill_cap` tasks.  The direct
  // launch of variant 9 runs with a pool of 12 tasks, so the spill really is used.
  const uint32_t spill_cap = 64;
  std::vector<ptk::Task> spill((size_t)3 * 8 * spill_cap + 1, ptk::Task{0xEEEEEEEEu, 0, 0, 0, 0, 0});
  if (variant == 9) {  // the ranked classes straight from phase 1 to the cooperative search, phase 2 without them
    direct_listed = meta[ptk::kMetaRanked];
    for_each_wave(3, [&] {
      ptk::knn1_coop_kernel<16, 12, true, true, M>(t->dev, ranges, qs.data(), o, cont, ho, redo_list.data(), sorted.data(),
                                          spill.data(), spill_cap);
    });
  }
  gridDim.x = blocks;
  blockDim.x = 64;
  for (uint32_t b = 0; b < blocks; ++b) {
    blockIdx.x = b;
    for (uint32_t l = 0; l < 64; ++l) {
      threadIdx.x = l;
      if (variant == 4)
        ptk::knn1_phase2_kernel<4, 2048, 1, M>(t->dev, qs.data(), e_inv, o, cont, sorted.data());
      else if (variant == 3)
        ptk::knn1_phase2_kernel<16, 2048, 4, M>(t->dev, qs.data(), e_inv, o, cont, sorted.data());
      else
        ptk::knn1_phase2_kernel<12, 2048, 4, M>(t->dev, qs.data(), e_inv, o, cont, sorted.data(), cap, ho);
    }
  }
  g_last_heavy = meta[ptk::kMetaHeavy] + direct_listed;
  g_last_redo = 0;
  if (cap) {
    for_each_wave(3, [&] {
      if (variant == 5) ptk::knn1_coop_kernel<16, 96, false, false, M>(t->dev, ranges, qs.data(), o, cont, ho, redo_list.data());
      else if (variant == 9)
        ptk::knn1_coop_kernel<16, 64, false, false, M>(t->dev, ranges, qs.data(), o, cont, ho, redo_list.data(), nullptr, spill.data(),
                                             spill_cap);
      else if (variant == 6) ptk::knn1_coop_kernel<64, 192, false, false, M>(t->dev, ranges, qs.data(), o, cont, ho, redo_list.data());
      else if (variant == 7) ptk::knn1_coop_kernel<8, 64, false, false, M>(t->dev, ranges, qs.data(), o, cont, ho, redo_list.data());
      else ptk::knn1_coop_kernel<32, 128, false, false, M>(t->dev, ranges, qs.data(), o, cont, ho, redo_list.data());
    });
  }
  g_last_spilled = 0;
  for (const ptk::Task& k : spill) g_last_spilled += k.ref != 0xEEEEEEEEu ? 1u : 0u;
  if (cap) {
    g_last_redo = meta[ptk::kMetaRedo];
    gridDim.x = 2;
    for (uint32_t b = 0; b < 2; ++b) {
      blockIdx.x = b;
      for (uint32_t l = 0; l < 64; ++l) {
        threadIdx.x = l;
        ptk::knn1_redo_kernel<16, 2048, 4, M>(t->dev, qs.data(), e_inv, o, cont, redo_list.data());
      }
    }
  }
  return (int)meta[0];
}

extern "C" {

int emu_knn1_two_phase(void* h, const float* q, uint64_t nq, float e, const uint32_t* perm, int variant,
                       ptk_neighbor* out) {
  // (the two-phase search is shipped for metric_l2_squared and metric_l1: ptk_search_knn_device)
  if (static_cast<Emu*>(h)->metric == 1) return emu_knn1_two_phase_m<ptk::MetricL1>(h, q, nq, e, perm, variant, out);
  return emu_knn1_two_phase_m<ptk::MetricL2>(h, q, nq, e, perm, variant, out);
}

// Box search: count pass then fill pass; offsets must hold nb + 1 entries, out is resized by the caller
// through a second call (pass out == nullptr to get the counts / offsets only).
int emu_box(void* h, const float* mins, const float* maxs, uint64_t nb, const float* root_min, const float* root_max,
            uint64_t* offsets, int32_t* out) {
  auto* t = static_cast<Emu*>(h);
  if (t->dim > 3) {  // any-dimension kernel
    std::vector<float> root(2 * (size_t)t->dim);
    for (uint32_t d = 0; d < t->dim; ++d) {
      root[d] = root_min[d];
      root[t->dim + d] = root_max[d];
    }
    const auto* nd_ranges = reinterpret_cast<const uint2*>(t->enc_nd.ranges.data());
    if ((size_t)16 * 64 * 8 + (size_t)4 * t->dim * 64 * 4 > sizeof(ptk::ptk_smem)) return -2;
    if (out == nullptr) {
      std::vector<uint64_t> counts(nb + 1, 0);
      for_each_lane(nb, [&] {
        ptk::box_nd_kernel<4, 2048, false>(t->dev_nd, nd_ranges, root.data(), mins, maxs, nb, counts.data(), nullptr,
                                           nullptr);
      }, 64);
      offsets[0] = 0;
      for (uint64_t i = 0; i < nb; ++i) offsets[i + 1] = offsets[i] + counts[i];
      return 0;
    }
    for_each_lane(nb, [&] {
      ptk::box_nd_kernel<16, 2048, true>(t->dev_nd, nd_ranges, root.data(), mins, maxs, nb, nullptr, offsets, out);
    }, 64);
    return 0;
  }
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  for (uint32_t d = 0; d < t->dim; ++d) {
    mn[d] = root_min[d];
    mx[d] = root_max[d];
  }
  const ptk::BoxState root{mn[0], mn[1], mn[2], mx[0], mx[1], mx[2]};
  const auto* ranges = reinterpret_cast<const uint2*>(t->enc.ranges.data());
  if (out == nullptr) {
    std::vector<uint64_t> counts(nb + 1, 0);
    for_each_lane(nb, [&] {
      ptk::box_kernel<4, 2048, false>(t->dev, ranges, root, mins, maxs, t->dim, nb, counts.data(), nullptr, nullptr);
    }, 64);
    offsets[0] = 0;
    for (uint64_t i = 0; i < nb; ++i) offsets[i + 1] = offsets[i] + counts[i];
    return 0;
  }
  for_each_lane(nb, [&] {
    ptk::box_kernel<16, 2048, true>(t->dev, ranges, root, mins, maxs, t->dim, nb, nullptr, offsets, out);
  }, 64);
  return 0;
}

// Phase 1 only: the class (0..7) and the home-leaf best distance of every query (analysis tools).
int emu_phase1(void* h, const float* q, uint64_t nq, uint8_t* cls_out, float* best_out) {
  auto* t = static_cast<Emu*>(h);
  if (nq == 0) return 0;
  std::vector<float4> qs(nq);
  std::vector<ptk::Record> crec(nq * ptk::kContSlots + 8);
  std::vector<uint4> cbest(nq);
  std::vector<ptk::ContKey> ckey(nq, 0xEEEE);
  std::vector<uint32_t> cids(nq), meta(ptk::kMetaWords, 0);
  std::vector<ptk::Neighbor> o(nq);
  ptk::Cont cont{cbest.data(), crec.data(), ckey.data(), cids.data(), meta.data(), nq};
  for_each_wave((uint32_t)((nq + 63) / 64), [&] {
    ptk::knn1_phase1u_kernel<4>(t->dev, q, t->dim, nullptr, nq, 1.0f, o.data(), cont, qs.data());
  });
  for (uint64_t i = 0; i < nq; ++i) {
    const bool final_ = (ckey[i] >> 13) == 7;
    cls_out[i] = final_ ? 0 : (uint8_t)cbest[i].z;
    best_out[i] = final_ ? o[i].distance : __uint_as_float(cbest[i].y);
  }
  return 0;
}

// The forest: the product's host build (ptk_forest_host.hpp) + the kernel, one wave per query.
// rotations_out receives the reflection vectors in use (n_trees x dim).
int emu_forest_knn(const float* points, uint64_t n, uint32_t dim, uint64_t max_leaf, uint32_t n_trees, uint64_t seed,
                   const float* q, uint64_t nq, uint32_t k, uint32_t max_leaves, float* rotations_out,
                   ptk_neighbor* out, uint32_t* dropped) {
  std::vector<ptk::ForestTreeHost> host(n_trees);
  std::vector<ptk::ForestTreeDev> trees(n_trees);
  std::vector<float> rotated;
  for (uint32_t i = 0; i < n_trees; ++i) {
    float* r = rotations_out + (size_t)i * dim;
    ptk::reflection_vector(seed, i, dim, r);
    g_err = ptk::build_forest_tree(points, n, dim, max_leaf, r, rotated, host[i]);
    if (!g_err.empty()) return -1;
    trees[i].nodes = host[i].nodes.data();
    trees[i].indices = host[i].indices.data();
    trees[i].rotation = r;
    trees[i].root_ref = host[i].root_ref;
    trees[i].cbits = host[i].cbits;
    trees[i].cmask = (1u << host[i].cbits) - 1u;
  }
  ptk::ForestDev f{trees.data(), points, n_trees, dim};
  for_each_wave((uint32_t)nq, [&] {
    ptk::forest_knn_kernel<64>(f, q, nq, k, max_leaves, reinterpret_cast<ptk::Neighbor*>(out), dropped);
  });
  return 0;
}

// The block -> tile mapping of the traversal kernels (ptk::xcd_runs) for every block of a grid of nb blocks.
void emu_xcd_runs(uint32_t nb, uint32_t* tiles) {
  for (uint32_t b = 0; b < nb; ++b) tiles[b] = ptk::xcd_runs(b, nb);
}

// Morton keys + identity ids exactly as the device computes them.
void emu_morton(const float* q, uint32_t dim, uint64_t nq, const float* lo, const float* inv, const uint32_t* bits,
                uint32_t* keys, uint32_t* ids) {
  float3 l = make_float3(lo[0], lo[1], lo[2]);
  float3 i = make_float3(inv[0], inv[1], inv[2]);
  for_each_lane(nq, [&] { ptk::morton_kernel(q, dim, nq, l, i, make_uint3(bits[0], bits[1], bits[2]), keys, ids); });
}

// The batch order as make_permutation() of the backend produces it with the library's own radix sort
// (ptk_sort.hpp): key + histogram kernel, then scan and stable scatter per 8-bit pass.  keys = the Morton keys.
void emu_radix_sort(const float* q, uint32_t dim, uint64_t nq, const float* lo, const float* inv, const uint32_t* bits,
                    uint32_t key_bits, uint32_t tile, uint32_t* keys, uint32_t* perm) {
  const float3 l = make_float3(lo[0], lo[1], lo[2]);
  const float3 i = make_float3(inv[0], inv[1], inv[2]);
  const uint3 b3 = make_uint3(bits[0], bits[1], bits[2]);
  const uint32_t tiles = (uint32_t)((nq + tile - 1) / tile), stride = (tiles + 3u) & ~3u;
  std::vector<uint32_t> hist((size_t)ptk::kRadixBins * stride, 0xEEEEEEEEu), totals(ptk::kRadixBins, 0xEEEEEEEEu);
  std::vector<uint32_t> k0(nq, 0xEEEEEEEEu), out_perm(nq, 0xEEEEEEEEu);
  std::vector<uint2> pa(nq, uint2{0xEEEEEEEEu, 0xEEEEEEEEu}), pb(nq, uint2{0xEEEEEEEEu, 0xEEEEEEEEu});
  const uint32_t passes = (key_bits + 7) / 8;
  const uint2* in = nullptr;
  for (uint32_t p = 0; p < passes; ++p) {
    const uint32_t shift = 8 * p;
    const bool first = p == 0, last = p + 1 == passes;
    uint2* out = in == pa.data() ? pb.data() : pa.data();
    for_each_wave(tiles, [&] {
      if (first) ptk::radix_hist_kernel<true>(q, dim, (uint32_t)nq, l, i, b3, k0.data(), in, shift, tile, stride, hist.data());
      else ptk::radix_hist_kernel<false>(q, dim, (uint32_t)nq, l, i, b3, k0.data(), in, shift, tile, stride, hist.data());
    });
    for_each_wave(ptk::kRadixBins, [&] { ptk::radix_scan_kernel(hist.data(), tiles, stride, totals.data()); });
    for_each_wave(tiles, [&] {
      if (first && last)
        ptk::radix_scatter_kernel<true, true>(k0.data(), in, out, out_perm.data(), (uint32_t)nq, shift, tile, stride, hist.data(), totals.data());
      else if (first)
        ptk::radix_scatter_kernel<true, false>(k0.data(), in, out, out_perm.data(), (uint32_t)nq, shift, tile, stride, hist.data(), totals.data());
      else if (last)
        ptk::radix_scatter_kernel<false, true>(k0.data(), in, out, out_perm.data(), (uint32_t)nq, shift, tile, stride, hist.data(), totals.data());
      else
        ptk::radix_scatter_kernel<false, false>(k0.data(), in, out, out_perm.data(), (uint32_t)nq, shift, tile, stride, hist.data(), totals.data());
    });
    in = out;
  }
  std::memcpy(keys, k0.data(), nq * 4);
  std::memcpy(perm, out_perm.data(), nq * 4);
}

// The same passes with blocks of four wavefronts on tiles of 4 096 items (radix_block_hist_kernel /
// radix_block_scatter_kernel): 256 fibers per block, ballots per wavefront, barriers per block.
void emu_radix_sort_blocks(const float* q, uint32_t dim, uint64_t nq, const float* lo, const float* inv,
                           const uint32_t* bits, uint32_t key_bits, uint32_t* keys, uint32_t* perm) {
  const float3 l = make_float3(lo[0], lo[1], lo[2]);
  const float3 i = make_float3(inv[0], inv[1], inv[2]);
  const uint3 b3 = make_uint3(bits[0], bits[1], bits[2]);
  const uint32_t tiles = (uint32_t)((nq + ptk::kSortTile - 1) / ptk::kSortTile), stride = (tiles + 3u) & ~3u;
  std::vector<uint32_t> hist((size_t)ptk::kRadixBins * stride, 0xEEEEEEEEu), totals(ptk::kRadixBins, 0xEEEEEEEEu);
  std::vector<uint32_t> k0(nq, 0xEEEEEEEEu), out_perm(nq, 0xEEEEEEEEu);
  std::vector<uint2> pa(nq, uint2{0xEEEEEEEEu, 0xEEEEEEEEu}), pb(nq, uint2{0xEEEEEEEEu, 0xEEEEEEEEu});
  const uint32_t passes = (key_bits + 7) / 8;
  const uint2* in = nullptr;
  for (uint32_t p = 0; p < passes; ++p) {
    const uint32_t shift = 8 * p;
    const bool first = p == 0, last = p + 1 == passes;
    uint2* out = in == pa.data() ? pb.data() : pa.data();
    for_each_block(tiles, ptk::kSortBlock, [&] {
      if (first) ptk::radix_block_hist_kernel<true>(q, dim, (uint32_t)nq, l, i, b3, k0.data(), in, shift, stride, hist.data());
      else ptk::radix_block_hist_kernel<false>(q, dim, (uint32_t)nq, l, i, b3, k0.data(), in, shift, stride, hist.data());
    });
    for_each_wave(ptk::kRadixBins, [&] { ptk::radix_scan_kernel(hist.data(), tiles, stride, totals.data()); });
    for_each_block(tiles, ptk::kSortBlock, [&] {
      if (first && last)
        ptk::radix_block_scatter_kernel<true, true>(k0.data(), in, out, out_perm.data(), (uint32_t)nq, shift, stride, hist.data(), totals.data());
      else if (first)
        ptk::radix_block_scatter_kernel<true, false>(k0.data(), in, out, out_perm.data(), (uint32_t)nq, shift, stride, hist.data(), totals.data());
      else if (last)
        ptk::radix_block_scatter_kernel<false, true>(k0.data(), in, out, out_perm.data(), (uint32_t)nq, shift, stride, hist.data(), totals.data());
      else
        ptk::radix_block_scatter_kernel<false, false>(k0.data(), in, out, out_perm.data(), (uint32_t)nq, shift, stride, hist.data(), totals.data());
    });
    in = out;
  }
  std::memcpy(keys, k0.data(), nq * 4);
  std::memcpy(perm, out_perm.data(), nq * 4);
}

}  // extern "C"

// ---- double precision (ptk_kernels_f64.hpp) ----------------------------------------------------
// The tree is built here with the product's host builder instantiated over double (what
// ptk_tree64_create_from_points does) and encoded with ptk::encode_tree64.
namespace {
struct Emu64 {
  using flat_t = pico_tree::internal::flat_tree<int, double, pico_tree::dynamic_extent>;
  flat_t flat{1};
  ptk::EncodedTree64 enc;
  ptk::TreeStats st;
  ptk::DevTree64 dev;
  std::vector<double> root;
  std::vector<double> outer;      // per branch {left_min, right_max}, the order of enc.nodes
  std::vector<ptk::Rec64> stack;  // one block's worth: blocks run one after the other
  uint32_t slots = 0;
  int metric = 0;  // PTK_METRIC_* (4 / 5: the topological metrics)
};

template <class F>
void for_each_lane64(Emu64* t, uint64_t n, F&& f) {
  // Every block reuses stack columns [0, slots * 64): blockIdx.x stays 0 while the lanes run.
  gridDim.x = 1;
  blockDim.x = 64;
  blockIdx.x = 0;
  for (uint64_t q0 = 0; q0 < n; q0 += 64) {
    const uint64_t m = std::min<uint64_t>(64, n - q0);
    for (uint32_t l = 0; l < 64; ++l) {
      threadIdx.x = l;
      f(q0, m);
    }
  }
}
}  // namespace

extern "C" {

void* emu64_create(const double* points, uint64_t n, uint32_t dim, uint64_t max_leaf) {
  using namespace pico_tree;
  auto* e = new Emu64;
  using space_t = space_map<point_map<double const, dynamic_extent>>;
  space_t space(points, n, dim);
  internal::space_view<space_t> view(space);
  e->flat = internal::build_flat_tree<int>(view, max_leaf_size_t(max_leaf), bounds_from_space, sliding_midpoint_max_side,
                                           true, 1u);
  bool unsupported = false;
  g_err = ptk::encode_tree64(dim, n, points, e->flat.nodes.data(), e->flat.nodes.size(), e->flat.indices.data(), e->st,
                             e->enc, unsupported);
  if (g_err.empty())
    g_err = ptk::encode_outer64(dim, n, e->flat.nodes.data(), e->flat.nodes.size(), e->flat.outer_bounds.data(), e->outer);
  if (!g_err.empty()) {
    delete e;
    return nullptr;
  }
  e->dev.outer = reinterpret_cast<const double2*>(e->outer.data());
  e->dev.nodes = reinterpret_cast<const ptk::Node64*>(e->enc.nodes.data());
  e->dev.pts = e->enc.points.data();
  e->dev.index = e->flat.indices.data();
  e->dev.ranges = reinterpret_cast<const uint2*>(e->enc.ranges.data());
  e->dev.root_ref = e->enc.root_ref;
  e->dev.cbits = e->enc.cbits;
  e->dev.cmask = (1u << e->enc.cbits) - 1u;
  e->dev.dim = dim;
  e->dev.stride = e->enc.stride;
  e->dev.n_points = (uint32_t)n;
  e->root.assign(e->flat.root_box.min(), e->flat.root_box.min() + dim);
  e->root.insert(e->root.end(), e->flat.root_box.max(), e->flat.root_box.max() + dim);
  e->slots = 2 * e->st.max_depth + 4;
  e->stack.resize((size_t)e->slots * 64);
  return e;
}
void emu64_destroy(void* h) { delete static_cast<Emu64*>(h); }
void emu64_set_metric(void* h, int metric) { static_cast<Emu64*>(h)->metric = metric; }

// The kd_tree::save stream of the tree (pico_tree/internal/stream.hpp); returns its size.
uint64_t emu64_save(void* h, void* buf, uint64_t cap) {
  std::ostringstream os(std::ios::out | std::ios::binary);
  auto* e = static_cast<Emu64*>(h);
  e->flat.keep_outer_bounds = e->metric == 4 || e->metric == 5;  // the four-bound stream of a topological tree
  pico_tree::internal::write_flat_tree(e->flat, os);
  const std::string bytes = os.str();
  if (buf != nullptr && cap >= bytes.size()) std::memcpy(buf, bytes.data(), bytes.size());
  return bytes.size();
}

#define EMU64_M(CALL)                          \
  if (t->metric == 1) {                        \
    using M = ptk::Metric64L1;                 \
    CALL;                                      \
  } else if (t->metric == 2) {                 \
    using M = ptk::Metric64LInf;               \
    CALL;                                      \
  } else if (t->metric == 3) {                 \
    using M = ptk::Metric64LNInf;              \
    CALL;                                      \
  } else if (t->metric == 4) {                 \
    using M = ptk::Topo64SO2;                  \
    CALL;                                      \
  } else if (t->metric == 5) {                 \
    using M = ptk::Topo64SE2;                  \
    CALL;                                      \
  } else {                                     \
    using M = ptk::Metric64L2;                 \
    CALL;                                      \
  }
// D3: the register form the backend launches for dim <= 3.
#define EMU64_WITH_METRIC(CALL)                \
  do {                                         \
    if (t->dev.dim <= 3) {                     \
      constexpr bool D3 = true;                \
      EMU64_M(CALL)                            \
    } else {                                   \
      constexpr bool D3 = false;               \
      EMU64_M(CALL)                            \
    }                                          \
  } while (0)

int emu64_knn(void* h, const double* q, uint64_t nq, uint32_t k, double e, int list_in_registers, ptk::Neighbor64* out) {
  auto* t = static_cast<Emu64*>(h);
  if (ptk::lds64_bytes(2, t->dev.dim) > sizeof(ptk::ptk_smem)) return -2;
  // A launch-order permutation (here: the batch backwards); results land in the query's own row.
  std::vector<uint32_t> order(nq);
  for (uint64_t i = 0; i < nq; ++i) order[i] = (uint32_t)(nq - 1 - i);
  const uint32_t* perm = order.data();
  if (list_in_registers && k > 1 && k <= 4) {
    EMU64_WITH_METRIC(for_each_lane64(t, nq, [&](uint64_t q0, uint64_t m) {
      ptk::knn64_reg_kernel<M, 4, D3>(t->dev, q, perm, q0, m, k, 1.0 / e, out, t->stack.data(), t->slots);
    }));
  } else if (list_in_registers && k > 1 && k <= 16) {
    EMU64_WITH_METRIC(for_each_lane64(t, nq, [&](uint64_t q0, uint64_t m) {
      ptk::knn64_reg_kernel<M, 16, D3>(t->dev, q, perm, q0, m, k, 1.0 / e, out, t->stack.data(), t->slots);
    }));
  } else if (list_in_registers && k > 1 && k <= 32) {
    EMU64_WITH_METRIC(for_each_lane64(t, nq, [&](uint64_t q0, uint64_t m) {
      ptk::knn64_reg_kernel<M, 32, D3>(t->dev, q, perm, q0, m, k, 1.0 / e, out, t->stack.data(), t->slots);
    }));
  } else {
    EMU64_WITH_METRIC(for_each_lane64(t, nq, [&](uint64_t q0, uint64_t m) {
      ptk::knn64_kernel<M, D3>(t->dev, q, perm, q0, m, k, 1.0 / e, out, t->stack.data(), t->slots);
    }));
  }
  return 0;
}

}  // extern "C"

// The double k-NN search with its long queries finished cooperatively (ptk_kernels_coop64.hpp): the capped launch
// (lanes one after the other), knn64_coop_kernel (64 fibers per wavefront), the reference search of what could not be
// certified.  counts = {queries handed over, queries redone, second sweeps}.  pool_small != 0: 40 spill slots per
// wavefront (long searches overflow them: the redo path).
namespace {
template <int K, class M>
int emu64_knn_capped_km(Emu64* t, const double* q, uint64_t nq, uint32_t k, const uint32_t* perm, uint32_t cap,
                        int pool_small, uint32_t max_heavy, ptk::Neighbor64* out, uint32_t* counts) {
  std::vector<uint32_t> meta(ptk::kMetaWords, 0), heavy(max_heavy + 1), ntasks(max_heavy + 1), redo(nq + 1);
  std::vector<ptk::Task64> tasks((size_t)std::max<uint32_t>(max_heavy, 1) * ptk::kMaxTasks);
  ptk::Handover64 ho{};
  ho.counter = ptk::kMetaHeavy;
  ho.meta = meta.data();
  ho.heavy_list = heavy.data();
  ho.ntasks = ntasks.data();
  ho.tasks = tasks.data();
  ho.max_heavy = max_heavy;
  for_each_lane64(t, nq, [&](uint64_t q0, uint64_t m) {
    ptk::knn64_capped_kernel<M, K>(t->dev, q, perm, q0, m, k, out, t->stack.data(), t->slots, cap, &ho);
  });
  const uint32_t spill_cap = pool_small ? 40u : 4096u;
  std::vector<ptk::Task64> spill((size_t)3 * spill_cap);
  for_each_wave(3, [&] {
    ptk::knn64_coop_kernel<K, 64, M>(t->dev, q, k, out, ho, redo.data(), ptk::kMetaRedo, spill.data(), spill_cap);
  });
  gridDim.x = 1;
  blockDim.x = 64;
  blockIdx.x = 0;
  for (uint32_t l = 0; l < 64; ++l) {
    threadIdx.x = l;
    ptk::knn64_redo_kernel<M, K>(t->dev, q, k, out, meta.data(), ptk::kMetaRedo, redo.data(), t->stack.data(), t->slots);
  }
  counts[0] = meta[ptk::kMetaHeavy];
  counts[1] = meta[ptk::kMetaRedo];
  counts[2] = meta[ptk::kKnnTieSweeps];
  return 0;
}
template <int K>
int emu64_knn_capped_k(Emu64* t, const double* q, uint64_t nq, uint32_t k, const uint32_t* perm, uint32_t cap, int pool_small,
                       uint32_t max_heavy, ptk::Neighbor64* out, uint32_t* counts) {
  if (t->metric == 1) return emu64_knn_capped_km<K, ptk::Metric64L1>(t, q, nq, k, perm, cap, pool_small, max_heavy, out, counts);
  return emu64_knn_capped_km<K, ptk::Metric64L2>(t, q, nq, k, perm, cap, pool_small, max_heavy, out, counts);
}
}  // namespace

extern "C" {

// dim <= 3, metric_l2_squared / metric_l1, k <= 32 (what launch_knn64 caps).
int emu64_knn_capped(void* h, const double* q, uint64_t nq, uint32_t k, const uint32_t* perm, uint32_t cap, int pool_small,
                     uint32_t max_heavy, ptk::Neighbor64* out, uint32_t* counts) {
  auto* t = static_cast<Emu64*>(h);
  if (t->dev.dim > 3 || t->metric > 1 || k == 0 || k > 32) return -2;
  if (k == 1) return emu64_knn_capped_k<1>(t, q, nq, k, perm, cap, pool_small, max_heavy, out, counts);
  if (k <= 4) return emu64_knn_capped_k<4>(t, q, nq, k, perm, cap, pool_small, max_heavy, out, counts);
  if (k <= 8) return emu64_knn_capped_k<8>(t, q, nq, k, perm, cap, pool_small, max_heavy, out, counts);
  if (k <= 16) return emu64_knn_capped_k<16>(t, q, nq, k, perm, cap, pool_small, max_heavy, out, counts);
  return emu64_knn_capped_k<32>(t, q, nq, k, perm, cap, pool_small, max_heavy, out, counts);
}

}  // extern "C"

// The double radius search with its long queries finished cooperatively (ptk_kernels_coop64.hpp): both passes.
// counts = {queries handed over, rows recounted / refilled from the root, sorted entries kept}.
namespace {
template <class M>
int emu64_radius_capped_m(Emu64* t, const double* q, uint64_t nq, double radius, double e, uint32_t cap, int small,
                          uint32_t max_heavy, uint64_t* offsets, std::vector<ptk::Neighbor64>& rows, uint32_t* counts) {
  std::vector<uint32_t> meta(ptk::kMetaWords, 0), heavy(max_heavy + 1), ntasks(max_heavy + 1), redo(max_heavy + 1), over(max_heavy + 1);
  std::vector<uint32_t> h_rows(max_heavy + 1), h_own(max_heavy + 1), h_at(max_heavy + 1), h_n(max_heavy + 1);
  std::vector<ptk::Task64> tasks((size_t)std::max<uint32_t>(max_heavy, 1) * ptk::kMaxTasks);
  const uint32_t entry_cap = small ? 600u : (uint32_t)std::max<uint64_t>(1, (uint64_t)max_heavy * 192);
  std::vector<unsigned long long> entries(entry_cap);
  std::vector<uint8_t> flag(nq + 1, 0);
  ptk::Handover64 ho{};
  ho.counter = ptk::kMetaHeavy;
  ho.meta = meta.data();
  ho.heavy_list = heavy.data();
  ho.ntasks = ntasks.data();
  ho.tasks = tasks.data();
  ho.max_heavy = max_heavy;
  ptk::RadiusHeavy64 hv{};
  hv.meta = meta.data();
  hv.rows = h_rows.data();
  hv.own = h_own.data();
  hv.run_at = h_at.data();
  hv.run_n = h_n.data();
  hv.entries = entries.data();
  hv.max_heavy = max_heavy;
  hv.entry_cap = entry_cap;
  std::vector<uint32_t> order(nq);
  for (uint64_t i = 0; i < nq; ++i) order[i] = (uint32_t)(nq - 1 - i);  // (a launch-order permutation: the batch backwards)
  const uint32_t* perm = order.data();
  const uint32_t spill_cap = small ? 24u : 2048u;
  std::vector<ptk::Task64> spill((size_t)3 * spill_cap);
  std::vector<uint64_t> cnt(nq + 1, 0);
  const double e_inv = 1.0 / e;
  auto one_block = [&](auto&& f) {
    gridDim.x = 1;
    blockDim.x = 64;
    blockIdx.x = 0;
    for (uint32_t l = 0; l < 64; ++l) {
      threadIdx.x = l;
      f();
    }
  };
  for_each_lane64(t, nq, [&](uint64_t q0, uint64_t m) {
    ptk::radius64_capped_kernel<M, false>(t->dev, q, perm, q0, m, radius, e_inv, cnt.data(), nullptr, nullptr, t->stack.data(),
                                          t->slots, cap, &ho, flag.data());
  });
  for_each_wave(3, [&] {
    ptk::radius64_coop_count_kernel<64, M>(t->dev, q, radius, e_inv, cnt.data(), ho, hv, redo.data(), spill.data(), spill_cap);
  });
  one_block([&] {
    ptk::radius64_redo_kernel<M, false>(t->dev, q, radius, e_inv, cnt.data(), nullptr, nullptr, meta.data(), ptk::kMetaRedo,
                                        redo.data(), t->stack.data(), t->slots);
  });
  offsets[0] = 0;
  for (uint64_t i = 0; i < nq; ++i) offsets[i + 1] = offsets[i] + cnt[i];
  rows.assign(offsets[nq] + 1, ptk::Neighbor64{});
  for_each_lane64(t, nq, [&](uint64_t q0, uint64_t m) {
    ptk::radius64_capped_kernel<M, true>(t->dev, q, perm, q0, m, radius, e_inv, nullptr, offsets, rows.data(), t->stack.data(),
                                         t->slots, cap, &ho, flag.data());
  });
  for_each_wave(3, [&] { ptk::radius64_coop_replay_kernel<M>(t->dev, q, e_inv, hv, offsets, rows.data(), over.data()); });
  one_block([&] {
    ptk::radius64_redo_kernel<M, true>(t->dev, q, radius, e_inv, nullptr, offsets, rows.data(), meta.data(), ptk::kMetaRc64Over,
                                       over.data(), t->stack.data(), t->slots);
  });
  rows.resize(offsets[nq]);
  counts[0] = std::min(meta[ptk::kMetaHeavy], max_heavy);
  counts[1] = meta[ptk::kMetaRedo];
  counts[2] = meta[ptk::kMetaRc64Entries];
  return 0;
}
std::vector<ptk::Neighbor64> g_rows64;
}  // namespace

extern "C" {

// dim <= 3, the four non-topological metrics.  Two calls: out == nullptr runs both passes and fills offsets; the second
// copies the rows of that run.
int emu64_radius_capped(void* h, const double* q, uint64_t nq, double radius, double e, uint32_t cap, int small,
                        uint32_t max_heavy, uint64_t* offsets, ptk::Neighbor64* out, uint32_t* counts) {
  auto* t = static_cast<Emu64*>(h);
  if (t->dev.dim > 3 || t->metric > 3) return -2;
  if (out != nullptr) {
    std::memcpy(out, g_rows64.data(), g_rows64.size() * sizeof(ptk::Neighbor64));
    return 0;
  }
  if (t->metric == 1) return emu64_radius_capped_m<ptk::Metric64L1>(t, q, nq, radius, e, cap, small, max_heavy, offsets, g_rows64, counts);
  if (t->metric == 2) return emu64_radius_capped_m<ptk::Metric64LInf>(t, q, nq, radius, e, cap, small, max_heavy, offsets, g_rows64, counts);
  if (t->metric == 3) return emu64_radius_capped_m<ptk::Metric64LNInf>(t, q, nq, radius, e, cap, small, max_heavy, offsets, g_rows64, counts);
  return emu64_radius_capped_m<ptk::Metric64L2>(t, q, nq, radius, e, cap, small, max_heavy, offsets, g_rows64, counts);
}

// out == nullptr: count pass (offsets filled); otherwise the fill pass (+ optional row sort).
int emu64_radius(void* h, const double* q, uint64_t nq, double radius, double e, int sort, uint64_t* offsets,
                 ptk::Neighbor64* out) {
  auto* t = static_cast<Emu64*>(h);
  if (ptk::lds64_bytes(2, t->dev.dim) > sizeof(ptk::ptk_smem)) return -2;
  if (out == nullptr) {
    std::vector<uint64_t> counts(nq + 1, 0);
    EMU64_WITH_METRIC(for_each_lane64(t, nq, [&](uint64_t q0, uint64_t m) {
      ptk::radius64_kernel<M, false, D3>(t->dev, q, nullptr, q0, m, radius, 1.0 / e, counts.data(), nullptr, nullptr, t->stack.data(),
                                     t->slots);
    }));
    offsets[0] = 0;
    for (uint64_t i = 0; i < nq; ++i) offsets[i + 1] = offsets[i] + counts[i];
    return 0;
  }
  EMU64_WITH_METRIC(for_each_lane64(t, nq, [&](uint64_t q0, uint64_t m) {
    ptk::radius64_kernel<M, true, D3>(t->dev, q, nullptr, q0, m, radius, 1.0 / e, nullptr, offsets, out, t->stack.data(), t->slots);
  }));
  if (sort) for_each_lane(nq, [&] { ptk::sort_rows64_kernel(offsets, nq, out); });
  return 0;
}

int emu64_box(void* h, const double* mins, const double* maxs, uint64_t nb, uint64_t* offsets, int32_t* out) {
  auto* t = static_cast<Emu64*>(h);
  if (ptk::lds64_bytes(4, t->dev.dim) > sizeof(ptk::ptk_smem)) return -2;
  const bool topo = t->metric == 4 || t->metric == 5;
  const uint32_t s1_mask = !topo ? 0u : (t->metric == 4 ? 1u : 4u);
  if (out == nullptr) {
    std::vector<uint64_t> counts(nb + 1, 0);
    for_each_lane64(t, nb, [&](uint64_t b0, uint64_t m) {
      if (topo)
        ptk::box64_kernel<false, true>(t->dev, t->root.data(), mins, maxs, b0, m, counts.data(), nullptr, nullptr,
                                       t->stack.data(), t->slots, s1_mask);
      else
        ptk::box64_kernel<false>(t->dev, t->root.data(), mins, maxs, b0, m, counts.data(), nullptr, nullptr,
                                 t->stack.data(), t->slots);
    });
    offsets[0] = 0;
    for (uint64_t i = 0; i < nb; ++i) offsets[i + 1] = offsets[i] + counts[i];
    return 0;
  }
  for_each_lane64(t, nb, [&](uint64_t b0, uint64_t m) {
    if (topo)
      ptk::box64_kernel<true, true>(t->dev, t->root.data(), mins, maxs, b0, m, nullptr, offsets, out, t->stack.data(),
                                    t->slots, s1_mask);
    else
      ptk::box64_kernel<true>(t->dev, t->root.data(), mins, maxs, b0, m, nullptr, offsets, out, t->stack.data(), t->slots);
  });
  return 0;
}

}  // extern "C"
