// ptk_family_nd.hip -- the searches of trees of any dimension > 3 (ptk_kernels_nd.hpp): k-NN, radius (traversal and captured log), trees
// deeper than the private stack classes.
// One of the translation units of libptk.so (ptk_backend_core.hpp).

#include "ptk_families.hpp"
#include "ptk_kernels_nd.hpp"

namespace {

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


static __global__ void warm_nd_kernel() {}

}  // namespace

namespace ptkf {

int knn_nd(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, uint32_t k, float e,
           ptk::Neighbor* d_out, hipStream_t s, bool no_register_list) {
  int rc = PTK_OK;
  PTK_WITH_METRIC(PTK_WITH_OVF(16, (launch_knn_nd<OVF, M>(t, d_q, perm, nq, k, e, d_out, s, no_register_list))));
  return rc;
}

int knn_nd_deep(const ptk_tree* t, const ptk::DevTreeND& dev, const float* d_q, uint64_t n, uint32_t k, float e,
                ptk::Neighbor* d_out, hipStream_t s) {
  const uint32_t blocks = (uint32_t)((n + 63) / 64);
  const size_t smem = (size_t)16 * 64 * 8 + (size_t)t->dim * 64 * 8;
  if (smem > t->lds_per_block)
    return fail(PTK_ERR_UNSUPPORTED, "dimension %u does not fit the LDS staging of the device search", t->dim);
  int rc = PTK_OK;
  PTK_WITH_METRIC({
    rc = allow_lds(ptk::knn_nd_kernel<16, -1, false, M>, smem);
    if (rc == PTK_OK)
      hipLaunchKernelGGL((ptk::knn_nd_kernel<16, -1, false, M>), dim3(blocks), dim3(64), smem, s, dev, d_q, nullptr, n, k,
                         inv_ratio(e), d_out);
  });
  if (rc != PTK_OK) return rc;
  PTK_HIP(hipGetLastError());
  return PTK_OK;
}

int radius_nd(const ptk_tree* t, const float* d_q, uint64_t nq, float radius, float e, bool fill, uint64_t* d_counts,
              const uint64_t* d_offsets, ptk::Neighbor* d_out, hipStream_t s, const uint32_t* perm, const uint32_t* n_dev) {
  int rc = PTK_OK;
  PTK_WITH_METRIC(PTK_WITH_OVF(16, (launch_radius_nd<OVF, M>(t, d_q, nq, radius, e, fill, d_counts, d_offsets, d_out, s, perm, n_dev))));
  return rc;
}

int radius_nd_capture(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float radius, float e,
                      uint64_t* d_counts, const ptk::RadiusCapture& cap, hipStream_t s) {
  int rc = PTK_OK;
  PTK_WITH_METRIC(PTK_WITH_OVF(16, (launch_radius_nd_capture<OVF, M>(t, d_q, perm, nq, radius, e, d_counts, cap, s))));
  return rc;
}

int radius_nd_deep(const ptk_tree* t, const ptk::DevTreeND& dev, const float* d_q, uint64_t n, float radius, float e,
                   bool fill, uint64_t* d_counts, const uint64_t* d_offsets, ptk::Neighbor* d_out, hipStream_t s) {
  const uint32_t blocks = (uint32_t)((n + 63) / 64);
  const size_t smem = (size_t)16 * 64 * 8 + (size_t)t->dim * 64 * 8;
  if (smem > t->lds_per_block)
    return fail(PTK_ERR_UNSUPPORTED, "dimension %u does not fit the LDS staging of the device search", t->dim);
  int rc = PTK_OK;
  PTK_WITH_METRIC({
    if (fill) {
      rc = allow_lds(ptk::radius_nd_kernel<16, -1, true, M>, smem);
      if (rc == PTK_OK)
        hipLaunchKernelGGL((ptk::radius_nd_kernel<16, -1, true, M>), dim3(blocks), dim3(64), smem, s, dev, d_q, n, radius,
                           inv_ratio(e), d_counts, d_offsets, d_out, nullptr, nullptr);
    } else {
      rc = allow_lds(ptk::radius_nd_kernel<16, -1, false, M>, smem);
      if (rc == PTK_OK)
        hipLaunchKernelGGL((ptk::radius_nd_kernel<16, -1, false, M>), dim3(blocks), dim3(64), smem, s, dev, d_q, n, radius,
                           inv_ratio(e), d_counts, d_offsets, d_out, nullptr, nullptr);
    }
  });
  if (rc != PTK_OK) return rc;
  PTK_HIP(hipGetLastError());
  return PTK_OK;
}

// (loads this unit's code object on the calling thread's device: ProcessWarmup of ptk_backend.hip)
void warm_nd() {
  hipLaunchKernelGGL(warm_nd_kernel, dim3(1), dim3(1), 0, nullptr);
}

}  // namespace ptkf
