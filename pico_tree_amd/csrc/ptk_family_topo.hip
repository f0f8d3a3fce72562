// ptk_family_topo.hip -- the searches under the topological metrics (metric_so2, metric_se2_squared; ptk_kernels_topo.hpp).
// One of the translation units of libptk.so (ptk_backend_core.hpp).

#include "ptk_families.hpp"
#include "ptk_kernels_topo.hpp"

namespace {

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


static __global__ void warm_topo_kernel() {}

}  // namespace

namespace ptkf {

int knn_topo(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, uint32_t k, float e,
             ptk::Neighbor* d_out, hipStream_t s, bool no_register_list) {
  int rc = PTK_OK;
  PTK_WITH_OVF(16, (launch_knn_topo<OVF>(t, d_q, perm, nq, k, e, d_out, s, no_register_list)));
  return rc;
}

int radius_topo(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float radius, float e, bool fill,
                uint64_t* d_counts, const uint64_t* d_offsets, ptk::Neighbor* d_out, hipStream_t s) {
  int rc = PTK_OK;
  PTK_WITH_OVF(16, (launch_radius_topo<OVF>(t, d_q, perm, nq, radius, e, fill, d_counts, d_offsets, d_out, s)));
  return rc;
}

// (loads this unit's code object on the calling thread's device: ProcessWarmup of ptk_backend.hip)
void warm_topo() {
  hipLaunchKernelGGL(warm_topo_kernel, dim3(1), dim3(1), 0, nullptr);
}

}  // namespace ptkf
