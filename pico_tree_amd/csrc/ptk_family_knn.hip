// ptk_family_knn.hip -- general k-NN of 3-D float32 trees: the k-list in registers (k <= 64) or in the output row, the capped launch with its
// cooperative search (ptk_kernels_coopk.hpp), trees deeper than the private stack classes.
// One of the translation units of libptk.so (ptk_backend_core.hpp).

#include "ptk_families.hpp"
#include "ptk_kernels_coopk.hpp"

namespace {

template <int S, int OVF, int BLOCK, int LEAFB, class M = ptk::MetricL2>
int launch_knn(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, uint32_t k, float e,
               ptk::Neighbor* d_out, hipStream_t s) {
  const uint32_t blocks = (uint32_t)((nq + BLOCK - 1) / BLOCK);
  const size_t stack_bytes = (size_t)S * BLOCK * 8;
  const size_t list_bytes = (size_t)k * BLOCK * 8;
  // The k-list goes to LDS while a wavefront's block stays under 48 KiB (k <= 80); beyond that the output row itself is
  // the list.  (Kernel ms on 900 k queries of config 3, list in LDS / in the row: knn = 65 47 / 71, knn = 100 138 / 138,
  // knn = 200 881 / 409 -- a list that leaves a CU two wavefronts loses to one in HBM.  Both forms are insert_sorted as a loop per lane: k beyond 64 wants a design of its own.)
  const bool list_lds = stack_bytes + list_bytes <= (size_t)48 * 1024;
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

template <int S, int OVF, int BLOCK, int LEAFB, class M = ptk::MetricL2>
int launch_knn_reg(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, uint32_t k, float e,
                   ptk::Neighbor* d_out, hipStream_t s, Scratch* scratch = nullptr) {
  const uint32_t blocks = (uint32_t)((nq + BLOCK - 1) / BLOCK);
  const size_t smem = (size_t)S * BLOCK * 8;
  Timer timer(t, s);
  // (the capped launch + cooperative search: the default metric and metric_l1, whose box distances are lower bounds)
  if constexpr ((std::is_same<M, ptk::MetricL2>::value || std::is_same<M, ptk::MetricL1>::value) && BLOCK == 64) {
    const uint32_t cap = scratch != nullptr ? knn_cap(e, nq, k) : 0u;
    if (cap != 0u) {
      // The capped launch, the cooperative search of what it handed over, the reference search of what that could
      // not certify: the counts stay on the device.  A batch of four million queries or more goes through as TWO capped
      // launches side by side -- the front of the launch order (the expensive rows: `perm` puts them first) on a second
      // stream, the rest on the caller's -- so that the cooperative search of what the front handed over runs BESIDE the
      // rest instead of behind it (PTK_KNN_OVERLAP_PCT: the front's share of the rows, 0 = one launch).
      const uint32_t coop_blocks = knn_coop_blocks(t, nq);
      uint64_t n_front = 0;
      hipStream_t side = nullptr;
      hipEvent_t fork = nullptr, join = nullptr;
      // (kernel ms, two launches / one: 7.2 M queries knn = 4 / 8 / 16 / 32 2.15 / 2.67 / 3.78 / 7.07 against 2.20 / 2.71 /
      // 3.86 / 7.11, 4.8 M 1.58 / 2.01 / 2.75 / 5.00 against 1.59 / 1.95 / 2.86 / 5.23; at 2.4 M and below, and for
      // knn = 2, the second launch costs more than the overlap returns: 1.82 against 1.67 at knn = 16)
      if (perm != nullptr && nq >= (1ull << 22) && k > 2) {
        const uint64_t pct = (uint64_t)std::min(90, std::max(0, knob_int("knn_overlap_pct", 20)));
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
      scratch->note_meta(meta, 2, n_front ? cap_front : cap_rest, n_front ? cap_rest : 0u);
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
    hipLaunchKernelGGL((ptk::knn_coop_kernel<KK, kKnnCoopPool, M>), dim3(cb), dim3(64), coop_smem, st, t->dev, ranges,      \
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


static __global__ void warm_knn_kernel() {}

}  // namespace

namespace ptkf {

int knn_reg(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, uint32_t k, float e,
            ptk::Neighbor* d_out, hipStream_t s, ptkb::Scratch* scratch) {
  int rc = PTK_OK;
  PTK_WITH_METRIC(PTK_WITH_OVF(kGenRing, (launch_knn_reg<kGenRing, OVF, 64, kGenLeafB, M>(t, d_q, perm, nq, k, e, d_out, s, scratch))));
  return rc;
}

int knn_rows(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, uint32_t k, float e,
             ptk::Neighbor* d_out, hipStream_t s) {
  int rc = PTK_OK;
  PTK_WITH_METRIC(PTK_WITH_OVF(16, (launch_knn<16, OVF, 64, 4, M>(t, d_q, perm, nq, k, e, d_out, s))));
  return rc;
}

// One piece of a batch on a tree deeper than the private spill classes: the record stacks spill to `dev.deep_spill`.
int knn_deep(const ptk_tree* t, const ptk::DevTree& dev, const float* d_q, uint64_t n, uint32_t k, float e,
             ptk::Neighbor* d_out, hipStream_t s) {
  const uint32_t blocks = (uint32_t)((n + 63) / 64);
  PTK_WITH_METRIC({
    hipLaunchKernelGGL((ptk::knn_kernel<16, -1, 64, 4, false, M>), dim3(blocks), dim3(64), (size_t)16 * 64 * 8, s, dev,
                       d_q, t->dim, nullptr, n, k, inv_ratio(e), d_out);
  });
  PTK_HIP(hipGetLastError());
  return PTK_OK;
}

// (loads this unit's code object on the calling thread's device: ProcessWarmup of ptk_backend.hip)
void warm_knn() {
  hipLaunchKernelGGL(warm_knn_kernel, dim3(1), dim3(1), 0, nullptr);
}

}  // namespace ptkf
