// ptk_families.hpp -- the entry points of the kernel-family translation units (see ptk_backend_core.hpp).  Each one
// binds the handle's metric and stack class to the template arguments of its launch wrappers; the callers in
// ptk_backend.hip pass plain arguments.
#pragma once

#include "ptk_backend_core.hpp"

namespace ptkf {
// ptk_backend.hip: the batch order (the library's radix sort of 32-bit keys; rocprim is compiled into that unit only)
int morton_bits(uint64_t nq);
size_t sort_tmp_bytes(uint64_t nq, int bits);
size_t permutation_scratch_bytes(uint64_t nq);
int sort_pairs_u32(void* tmp, size_t tmp_bytes, uint32_t* keys, uint32_t* keys_out, uint32_t* ids, uint32_t* ids_out,
                   uint64_t nq, int bits, hipStream_t s);
// ptk_family_knn.hip: 3-D float32 trees, k > 1 (and k = 1 of the other metrics)
int knn_reg(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, uint32_t k, float e,
            ptk::Neighbor* d_out, hipStream_t s, ptkb::Scratch* scratch);
int knn_rows(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, uint32_t k, float e,
             ptk::Neighbor* d_out, hipStream_t s);
int knn_deep(const ptk_tree* t, const ptk::DevTree& dev, const float* d_q, uint64_t n, uint32_t k, float e,
             ptk::Neighbor* d_out, hipStream_t s);
void warm_knn();
// ptk_family_radius.hip: 3-D float32 trees
int radius_traverse(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float radius, float e, bool fill,
                    uint64_t* d_counts, const uint64_t* d_offsets, ptk::Neighbor* d_out, hipStream_t s,
                    const uint32_t* n_dev = nullptr);
int radius_capture(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float radius, float e,
                   uint64_t* d_counts, const ptk::RadiusCapture& cap, hipStream_t s);
// (far_cap != 0: the capped list pass + the cooperative count of what it hands over; `heavy` is where the batch keeps
// that for the fill pass, `scratch` has radius_coop_scratch_bytes() left)
int radius_list(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float radius, float e,
                uint64_t* d_counts, const ptk::RadiusCapture& cap, hipStream_t s, uint32_t far_cap = 0,
                ptkb::Scratch* scratch = nullptr, const ptk::RadiusHeavy* heavy = nullptr);
int radius_replay(const ptk_tree* t, const float* d_q, float e, const ptk::RadiusCapture& cap, const uint64_t* d_offsets,
                  ptk::Neighbor* d_out, uint32_t* over_list, uint32_t* n_over, hipStream_t s,
                  const ptk::RadiusHeavy* heavy = nullptr);
int radius_log_scatter(const ptk_tree* t, const ptk::RadiusCapture& cap, const uint64_t* d_offsets, ptk::Neighbor* d_out,
                       uint32_t* over_list, uint32_t* n_over, hipStream_t s);
int radius_deep(const ptk_tree* t, const ptk::DevTree& dev, const float* d_q, uint64_t n, float radius, float e, bool fill,
                uint64_t* d_counts, const uint64_t* d_offsets, ptk::Neighbor* d_out, hipStream_t s);
void warm_radius();
// ptk_family_nd.hip: dim > 3
int knn_nd(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, uint32_t k, float e,
           ptk::Neighbor* d_out, hipStream_t s, bool no_register_list);
int knn_nd_deep(const ptk_tree* t, const ptk::DevTreeND& dev, const float* d_q, uint64_t n, uint32_t k, float e,
                ptk::Neighbor* d_out, hipStream_t s);
int radius_nd(const ptk_tree* t, const float* d_q, uint64_t nq, float radius, float e, bool fill, uint64_t* d_counts,
              const uint64_t* d_offsets, ptk::Neighbor* d_out, hipStream_t s, const uint32_t* perm = nullptr,
              const uint32_t* n_dev = nullptr);
int radius_nd_capture(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float radius, float e,
                      uint64_t* d_counts, const ptk::RadiusCapture& cap, hipStream_t s);
int radius_nd_deep(const ptk_tree* t, const ptk::DevTreeND& dev, const float* d_q, uint64_t n, float radius, float e,
                   bool fill, uint64_t* d_counts, const uint64_t* d_offsets, ptk::Neighbor* d_out, hipStream_t s);
void warm_nd();
// ptk_family_topo.hip: metric_so2 / metric_se2_squared
int knn_topo(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, uint32_t k, float e,
             ptk::Neighbor* d_out, hipStream_t s, bool no_register_list);
int radius_topo(const ptk_tree* t, const float* d_q, const uint32_t* perm, uint64_t nq, float radius, float e, bool fill,
                uint64_t* d_counts, const uint64_t* d_offsets, ptk::Neighbor* d_out, hipStream_t s);
void warm_topo();
// ptk_family_f64.hip
void warm_f64();
}  // namespace ptkf
