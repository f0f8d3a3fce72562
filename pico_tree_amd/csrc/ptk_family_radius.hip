// ptk_family_radius.hip -- radius search of 3-D float32 trees: the count / fill traversal, the count pass that captures a log of hits, the
// leaf lists + replay of ptk_kernels_lists.hpp, trees deeper than the private stack classes.
// One of the translation units of libptk.so (ptk_backend_core.hpp).

#include "ptk_families.hpp"
#include "ptk_kernels_lists.hpp"
#include "ptk_kernels_coopr.hpp"

namespace {

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
                       uint64_t* d_counts, const ptk::RadiusCapture& cap, hipStream_t s, uint32_t far_cap,
                       ptkb::Scratch* scratch, const ptk::RadiusHeavy* heavy) {
  static_assert(ptk::kRcMaxDepth == 51u, "radius_cap() of ptk_backend_core.hpp names this depth");
  const size_t smem = (size_t)S * 64 * 8 + ptk::kListLds;  // + the group buffers and the chunk table of the wavefront
  Timer timer(t, s);
  PTK_HIP(hipMemsetAsync(cap.counters, 0, ptk::kCapSubPools * ptk::kCapCounterStride * 4, s));
  // (leaves of more than kListMaskBits points -- a count of 32 needs six bits -- / an approximate search: see RadiusListPolicy)
  const bool big = t->dev.cmask >= ptk::kListMaskBits, exact = e == 1.0f;
  const bool capped = far_cap != 0u && scratch != nullptr && heavy != nullptr && heavy->max_heavy != 0u;
  ptk::Handover ho{};
  uint32_t* redo_list = nullptr;
  ptk::Task* spill = nullptr;
  uint32_t coop_blocks = 0;
  if (capped) {  // the hand-over list of this launch (transient) and the counters the fill pass reads again
    coop_blocks = radius_coop_blocks(t, nq);
    ho.counter = ptk::kMetaHeavy;
    ho.meta = heavy->meta;
    ho.max_heavy = heavy->max_heavy;
    ho.full_keeps = 1u;  // a query that finds the list full goes on in its lane
    ho.heavy_list = scratch->take<uint32_t>(heavy->max_heavy);
    ho.ntasks = scratch->take<uint32_t>(heavy->max_heavy);
    ho.tasks = scratch->take<ptk::Task>((size_t)heavy->max_heavy * ptk::kMaxTasks);
    redo_list = scratch->take<uint32_t>(heavy->max_heavy);
    spill = reinterpret_cast<ptk::Task*>(scratch->take<char>((size_t)coop_blocks * kRadiusCoopSpill * 32));
    if (!ho.heavy_list || !ho.ntasks || !ho.tasks || !redo_list || !spill) return fail(PTK_ERR_NOMEM, "scratch block too small");
    PTK_HIP(hipMemsetAsync(heavy->meta, 0, ptk::kMetaWords * 4, s));
  }
#define PTK_LAUNCH_LIST(BIG, EXACT)                                                                                   \
  do {                                                                                                                \
    if (capped)                                                                                                       \
      hipLaunchKernelGGL((ptk::radius_list_kernel<S, OVF, LEAFB, M, BIG, EXACT, true>), dim3(cap.n_static), dim3(64),  \
                         smem, s, t->dev, d_q, t->dim, perm, nq, radius, inv_ratio(e), d_counts, cap, far_cap, ho);    \
    else                                                                                                              \
      hipLaunchKernelGGL((ptk::radius_list_kernel<S, OVF, LEAFB, M, BIG, EXACT, false>), dim3(cap.n_static), dim3(64), \
                         smem, s, t->dev, d_q, t->dim, perm, nq, radius, inv_ratio(e), d_counts, cap, 0u,              \
                         ptk::Handover{});                                                                            \
  } while (0)
  if (big && exact) PTK_LAUNCH_LIST(true, true);
  else if (big) PTK_LAUNCH_LIST(true, false);
  else if (exact) PTK_LAUNCH_LIST(false, true);
  else PTK_LAUNCH_LIST(false, false);
#undef PTK_LAUNCH_LIST
  PTK_HIP(hipGetLastError());
  if (capped) {
    // The queries handed over, a wavefront each (sorted leaf entries for the fill pass, the count completed); then the
    // rows that could not be finished that way, counted again from the root by one lane each.
    const size_t coop_smem = (size_t)ptk::radius_coop_lds_words(kRadiusCoopPool) * 4;
    if (exact) {
      int rc = allow_lds(ptk::radius_coop_count_kernel<kRadiusCoopPool, M, true>, coop_smem);
      if (rc != PTK_OK) return rc;
      hipLaunchKernelGGL((ptk::radius_coop_count_kernel<kRadiusCoopPool, M, true>), dim3(coop_blocks), dim3(64), coop_smem, s,
                         t->dev, d_q, t->dim, radius, inv_ratio(e), d_counts, ho, *heavy, redo_list, spill, kRadiusCoopSpill);
    } else {
      int rc = allow_lds(ptk::radius_coop_count_kernel<kRadiusCoopPool, M, false>, coop_smem);
      if (rc != PTK_OK) return rc;
      hipLaunchKernelGGL((ptk::radius_coop_count_kernel<kRadiusCoopPool, M, false>), dim3(coop_blocks), dim3(64), coop_smem, s,
                         t->dev, d_q, t->dim, radius, inv_ratio(e), d_counts, ho, *heavy, redo_list, spill, kRadiusCoopSpill);
    }
    const uint32_t redo_blocks = (heavy->max_heavy + 63u) / 64u;
    hipLaunchKernelGGL((ptk::radius_kernel<S, OVF, 64, LEAFB, false, M>), dim3(redo_blocks), dim3(64), (size_t)S * 64 * 8, s,
                       t->dev, d_q, t->dim, redo_list, (uint64_t)heavy->max_heavy, radius, inv_ratio(e), d_counts, nullptr,
                       nullptr, heavy->meta + ptk::kMetaRedo);
    PTK_HIP(hipGetLastError());
  }
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
                         hipStream_t s, const ptk::RadiusHeavy* heavy) {
  Timer timer(t, s);
  PTK_HIP(hipMemsetAsync(n_over, 0, 4, s));
  hipLaunchKernelGGL((ptk::radius_replay_kernel<kReplayHits, kReplayRing, M>), dim3(cap.n_static), dim3(64),
                     ptk::replay_lds(kReplayRing), s, t->dev, d_q, t->dim, inv_ratio(e), cap, d_offsets, d_out, over_list,
                     n_over);
  if (heavy != nullptr)  // the rows' tails the capped list pass handed to wavefronts (ptk_kernels_coopr.hpp)
    hipLaunchKernelGGL((ptk::radius_coop_replay_kernel<M>), dim3(std::min<uint32_t>(heavy->max_heavy, (uint32_t)t->cus * 16u)),
                       dim3(64), 0, s, t->dev, d_q, t->dim, inv_ratio(e), *heavy, d_offsets, d_out, over_list, n_over);
  PTK_HIP(hipGetLastError());
  timer.stop(0, 0);
  return PTK_OK;
}

static __global__ void warm_radius_kernel() {}

}  // namespace

namespace ptkf {

int radius_traverse(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float radius, float e, bool fill,
                    uint64_t* d_counts, const uint64_t* d_offsets, ptk::Neighbor* d_out, hipStream_t s,
                    const uint32_t* n_dev) {
  int rc = PTK_OK;
  PTK_WITH_METRIC(PTK_WITH_OVF(16, (launch_radius<16, OVF, 64, kGenLeafB, M>(t, d_q, perm, nq, radius, e, fill, d_counts, d_offsets, d_out, s, n_dev))));
  return rc;
}

int radius_capture(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float radius, float e,
                   uint64_t* d_counts, const ptk::RadiusCapture& cap, hipStream_t s) {
  int rc = PTK_OK;
  PTK_WITH_METRIC(PTK_WITH_OVF(kGenRing, (launch_radius_capture<kGenRing, OVF, 64, kGenLeafB, M>(t, d_q, perm, nq, radius, e, d_counts, cap, s))));
  return rc;
}

int radius_list(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float radius, float e,
                uint64_t* d_counts, const ptk::RadiusCapture& cap, hipStream_t s, uint32_t far_cap, ptkb::Scratch* scratch,
                const ptk::RadiusHeavy* heavy) {
  int rc = PTK_OK;
  PTK_WITH_METRIC(PTK_WITH_OVF(kGenRing, (launch_radius_list<kGenRing, OVF, kGenLeafB, M>(t, d_q, perm, nq, radius, e, d_counts, cap, s, far_cap, scratch, heavy))));
  return rc;
}

int radius_replay(const ptk_tree* t, const float* d_q, float e, const ptk::RadiusCapture& cap, const uint64_t* d_offsets,
                  ptk::Neighbor* d_out, uint32_t* over_list, uint32_t* n_over, hipStream_t s, const ptk::RadiusHeavy* heavy) {
  int rc = PTK_OK;
  PTK_WITH_METRIC((rc = launch_radius_replay<M>(t, d_q, e, cap, d_offsets, d_out, over_list, n_over, s, heavy)));
  return rc;
}

// The rows of a captured log of hits (r03's form: dim > 3, PTK_RADIUS_LISTS=0) scattered to their places.
int radius_log_scatter(const ptk_tree* t, const ptk::RadiusCapture& cap, const uint64_t* d_offsets, ptk::Neighbor* d_out,
                       uint32_t* over_list, uint32_t* n_over, hipStream_t s) {
  Timer timer(t, s);
  PTK_HIP(hipMemsetAsync(n_over, 0, 4, s));
  constexpr int W = 1;  // (one wavefront per block: the LDS of a CU divides evenly)
  const size_t hold = (size_t)W * ptk::kLogScatterLds;  // a staged and a sorted chunk per wavefront
  int rc = allow_lds(ptk::radius_log_scatter_kernel<W>, hold);
  if (rc != PTK_OK) return rc;
  hipLaunchKernelGGL((ptk::radius_log_scatter_kernel<W>), dim3((cap.n_static + W - 1) / W), dim3(64 * W), hold, s, cap,
                     d_offsets, d_out, over_list, n_over);
  PTK_HIP(hipGetLastError());
  timer.stop(0, 0);
  return PTK_OK;
}

// One piece of a batch on a tree deeper than the private spill classes (count or fill; no capture).
int radius_deep(const ptk_tree* t, const ptk::DevTree& dev, const float* d_q, uint64_t n, float radius, float e, bool fill,
                uint64_t* d_counts, const uint64_t* d_offsets, ptk::Neighbor* d_out, hipStream_t s) {
  const uint32_t blocks = (uint32_t)((n + 63) / 64);
  PTK_WITH_METRIC({
    if (fill)
      hipLaunchKernelGGL((ptk::radius_kernel<16, -1, 64, 4, true, M>), dim3(blocks), dim3(64), (size_t)16 * 64 * 8, s, dev,
                         d_q, t->dim, nullptr, n, radius, inv_ratio(e), d_counts, d_offsets, d_out, nullptr);
    else
      hipLaunchKernelGGL((ptk::radius_kernel<16, -1, 64, 4, false, M>), dim3(blocks), dim3(64), (size_t)16 * 64 * 8, s, dev,
                         d_q, t->dim, nullptr, n, radius, inv_ratio(e), d_counts, d_offsets, d_out, nullptr);
  });
  PTK_HIP(hipGetLastError());
  return PTK_OK;
}

// (loads this unit's code object on the calling thread's device: ProcessWarmup of ptk_backend.hip)
void warm_radius() {
  hipLaunchKernelGGL(warm_radius_kernel, dim3(1), dim3(1), 0, nullptr);
}

}  // namespace ptkf
