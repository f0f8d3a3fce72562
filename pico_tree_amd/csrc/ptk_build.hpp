// ptk_build.hpp -- the top levels of the tree build on the device (tree build acceleration).
//
// The reference builds its tree top-down (internal/kd_tree_builder.hpp:229-276, :352-396: sliding midpoint, box of
// a child = box of the parent cut at the plane, std::partition of the index range).  Everything a level needs is
// known once its parent level is done, and the only part that touches every point is the partition.  The first
// levels of a large cloud are a handful of partitions of millions of indices each -- the part the host's threads
// scale worst on (one std::partition per node, serial at the root) -- so those are made here:
//
//   flags    pass[i] = point[idx[i]][axis] < plane for every position of every node of the level (one launch)
//   scan     exclusive sum of the flags over the whole permutation (rocprim)
//   cut      per node: m = passing points, and how many of them already lie in the first m positions
//   lists    std::partition on a range (libstdc++, bidirectional scheme) swaps the k-th failing position from the
//            left of the cut with the k-th passing position from the right end: both lists, by their scan ranks
//   swap     the pairs
//
// which leaves the permutation std::partition leaves, position for position (the same argument as
// pico_tree::internal::parallel_partition, include/pico_tree/internal/flat_tree.hpp).  Planes, boxes and the split
// axis are computed on the host exactly as the host builder computes them (split_top_levels()); nodes of at most
// `threshold` points go back to the host, where a pool of workers builds each subtree the ordinary way
// (build_flat_tree_below()).  A node whose partition leaves one side empty needs the sliding step, std::nth_element,
// whose permutation only the host library defines: its range makes a round trip to the host (on the scan-like cloud of
// BASELINE config 2 an outlier makes the plane of a 963 k-point node slide four times in a row).
// tests/test_gpu_parity.py compares the stream of a tree built this way with the host build byte for byte.

#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <utility>
#include <vector>

#include "pico_tree/internal/flat_tree.hpp"

// A kernel that is not a template is defined in a header several translation units include (ptk_backend_core.hpp):
// internal linkage, so every unit that launches it has its own and the units that do not emit nothing.
#ifndef PTK_GLOBAL
#define PTK_GLOBAL static __global__
#endif


namespace ptk {

struct BuildSeg {
  uint32_t begin, end, axis;
  float plane;
  uint32_t m;          // out: points below the plane
  uint32_t pass_left;  // out: of those, already among the first m positions
  uint32_t pad0, pad1;
};

constexpr uint32_t kBuildMaxDim = 8;
constexpr uint32_t kBuildBoxBlocks = 1024;

PTK_GLOBAL __launch_bounds__(256) void build_iota_kernel(int32_t* __restrict__ idx, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) idx[i] = (int32_t)i;
}

// Per block: min and max of every coordinate over a slice of the points (exact: any grouping gives the same box).
PTK_GLOBAL __launch_bounds__(256) void build_box_kernel(
    const float* __restrict__ pts, uint32_t n, uint32_t dim, float* __restrict__ partial) {
  __shared__ float lo[256], hi[256];
  for (uint32_t a = 0; a < dim; ++a) {
    float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
      const float v = pts[i * dim + a];
      mn = v < mn ? v : mn;
      mx = v > mx ? v : mx;
    }
    lo[threadIdx.x] = mn;
    hi[threadIdx.x] = mx;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) {
        lo[threadIdx.x] = lo[threadIdx.x + s] < lo[threadIdx.x] ? lo[threadIdx.x + s] : lo[threadIdx.x];
        hi[threadIdx.x] = hi[threadIdx.x + s] > hi[threadIdx.x] ? hi[threadIdx.x + s] : hi[threadIdx.x];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      partial[(blockIdx.x * dim + a) * 2 + 0] = lo[0];
      partial[(blockIdx.x * dim + a) * 2 + 1] = hi[0];
    }
    __syncthreads();
  }
}

// The node of the level position i lies in (the nodes are in position order); nsegs if none.
__device__ __forceinline__ uint32_t build_find(const BuildSeg* __restrict__ segs, uint32_t nsegs, uint32_t i) {
  uint32_t lo = 0, hi = nsegs;  // first node with begin > i
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (segs[mid].begin <= i) lo = mid + 1;
    else hi = mid;
  }
  if (lo == 0) return nsegs;
  return i < segs[lo - 1].end ? lo - 1 : nsegs;
}

PTK_GLOBAL __launch_bounds__(256) void build_flag_kernel(
    const float* __restrict__ pts, uint32_t dim, const int32_t* __restrict__ idx, const BuildSeg* __restrict__ segs,
    uint32_t nsegs, uint32_t n, uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i > n) return;
  uint32_t f = 0;
  if (i < n) {
    const uint32_t s = build_find(segs, nsegs, i);
    if (s < nsegs) f = pts[(uint64_t)(uint32_t)idx[i] * dim + segs[s].axis] < segs[s].plane ? 1u : 0u;
  }
  flags[i] = f;  // (flags[n] = 0: the scan then also yields the total)
}

PTK_GLOBAL void build_cut_kernel(const uint32_t* __restrict__ sums, BuildSeg* __restrict__ segs, uint32_t nsegs) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsegs) return;
  const uint32_t b = segs[s].begin;
  const uint32_t m = sums[segs[s].end] - sums[b];
  segs[s].m = m;
  segs[s].pass_left = sums[b + m] - sums[b];
}

PTK_GLOBAL __launch_bounds__(256) void build_list_kernel(
    const uint32_t* __restrict__ flags, const uint32_t* __restrict__ sums, const BuildSeg* __restrict__ segs,
    uint32_t nsegs, uint32_t n, uint32_t* __restrict__ from_left, uint32_t* __restrict__ from_right) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = build_find(segs, nsegs, i);
  if (s >= nsegs) return;
  const uint32_t b = segs[s].begin, m = segs[s].m;
  const uint32_t before = sums[i] - sums[b];  // passing positions of this node before i
  if (i < b + m) {
    if (!flags[i]) from_left[b + (i - b) - before] = i;  // its rank among the failing positions of the left part
  } else if (flags[i]) {
    from_right[b + before - segs[s].pass_left] = i;  // its rank among the passing positions of the right part
  }
}

PTK_GLOBAL __launch_bounds__(256) void build_swap_kernel(
    int32_t* __restrict__ idx, const BuildSeg* __restrict__ segs, uint32_t nsegs, uint32_t n,
    const uint32_t* __restrict__ from_left, const uint32_t* __restrict__ from_right) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = build_find(segs, nsegs, i);
  if (s >= nsegs) return;
  const uint32_t b = segs[s].begin, swaps = segs[s].m - segs[s].pass_left, k = i - b;
  if (k >= swaps) return;
  const uint32_t l = from_left[b + k], r = from_right[b + swaps - 1u - k];
  const int32_t t = idx[l];
  idx[l] = idx[r];
  idx[r] = t;
}

// ---- the sliding step (std::nth_element) ---------------------------------------------------------------------------
// When a partition leaves one side empty the reference slides the plane to the nearest point with std::nth_element
// (internal/kd_tree_builder.hpp:255-275), and the permutation that call leaves is part of the tree.  libstdc++'s
// nth_element is an introselect (bits/stl_algo.h:1964-1986): rounds of "median of three to the front, Hoare partition
// of the rest around it, keep the side that holds nth" until at most three elements are left.  The first rounds --
// the ones that touch a million indices -- are made here; the partition of a round is, like std::partition above,
// a pairing by rank:
//
//   __unguarded_partition(first + 1, last, pivot = first) moves a left pointer up to the next element that is NOT
//   below the pivot and a right pointer down to the next that is NOT above it, swaps the two and goes on, until the
//   pointers meet; until then both pointers only see elements nobody has moved yet.  So the k-th "left stop"
//   (positions with !(v < pivot), ascending) is swapped with the k-th "right stop" (positions with !(pivot < v),
//   descending) for every k below K = the first k whose left stop is not left of its right stop; the function returns
//   where the left pointer stands at the end (slide_pair_kernel).
//
// flags (both kinds, packed in one 64-bit word), one scan, the two rank lists, the pairs; the host reads the cut and
// decides which side is kept, exactly as __introselect does.  Once the range is small the host finishes the SAME
// call -- std::__introselect on what is left, with what is left of its depth limit -- so the outcome is libstdc++'s
// to the last swap (tests/test_device_build.py: byte-identical trees on clouds whose planes slide).
PTK_GLOBAL void slide_pivot_kernel(const float* __restrict__ pts, uint32_t dim, uint32_t axis, int32_t* __restrict__ idx,
                                   uint32_t first, uint32_t last, float* __restrict__ pivot) {
  // __move_median_to_first(result = first, a = first + 1, b = mid, c = last - 1)
  const uint32_t a = first + 1u, b = first + (last - first) / 2u, c = last - 1u;
  auto val = [&](uint32_t i) { return pts[(uint64_t)(uint32_t)idx[i] * dim + axis]; };
  const float va = val(a), vb = val(b), vc = val(c);
  uint32_t pick;
  if (va < vb) {
    if (vb < vc) pick = b;
    else if (va < vc) pick = c;
    else pick = a;
  } else if (va < vc) {
    pick = a;
  } else if (vb < vc) {
    pick = c;
  } else {
    pick = b;
  }
  const int32_t t = idx[first];
  idx[first] = idx[pick];
  idx[pick] = t;
  *pivot = val(first);
}

// flags[i - lo] = {left stop, right stop << 32} for the positions [lo, hi); flags[hi - lo] = 0 (the scan's total).
PTK_GLOBAL __launch_bounds__(256) void slide_flag_kernel(const float* __restrict__ pts, uint32_t dim, uint32_t axis,
                                                         const int32_t* __restrict__ idx, uint32_t lo, uint32_t hi,
                                                         const float* __restrict__ pivot,
                                                         unsigned long long* __restrict__ flags) {
  const uint32_t i = lo + blockIdx.x * 256u + threadIdx.x;
  if (i > hi) return;
  unsigned long long f = 0ull;
  if (i < hi) {
    const float v = pts[(uint64_t)(uint32_t)idx[i] * dim + axis], p = *pivot;
    f = (!(v < p) ? 1ull : 0ull) | (!(p < v) ? 1ull << 32 : 0ull);
  }
  flags[i - lo] = f;
}

// left_list[k] = position of the k-th left stop (ascending), right_list[k] = position of the k-th right stop (descending).
PTK_GLOBAL __launch_bounds__(256) void slide_list_kernel(const unsigned long long* __restrict__ flags,
                                                         const unsigned long long* __restrict__ sums, uint32_t lo,
                                                         uint32_t hi, uint32_t* __restrict__ left_list,
                                                         uint32_t* __restrict__ right_list) {
  const uint32_t i = lo + blockIdx.x * 256u + threadIdx.x;
  if (i >= hi) return;
  const unsigned long long f = flags[i - lo], before = sums[i - lo], total = sums[hi - lo];
  if (f & 0xFFFFFFFFull) left_list[(uint32_t)before] = i;
  if (f >> 32) right_list[(uint32_t)(total >> 32) - (uint32_t)(before >> 32) - 1u] = i;
}

// The pairs below K are swapped (K = the first k whose left stop is missing or not left of its right stop).  What
// the function returns is where the left pointer stands then: it has moved on from left stop K - 1 to the next element
// that is not below the pivot -- left stop K, unless it first meets the element it swapped into right stop K - 1:
// *cut = min(left stop K, right stop K - 1)   (K = 0: left stop 0, which a median-of-three pivot guarantees).
PTK_GLOBAL __launch_bounds__(256) void slide_pair_kernel(int32_t* __restrict__ idx, const unsigned long long* __restrict__ sums,
                                                         uint32_t n_range, const uint32_t* __restrict__ left_list,
                                                         const uint32_t* __restrict__ right_list, uint32_t* __restrict__ cut) {
  const unsigned long long total = sums[n_range];
  const uint32_t n_left = (uint32_t)total, n_right = (uint32_t)(total >> 32);
  const uint32_t pairs = n_left < n_right ? n_left : n_right;
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k > pairs) return;
  const bool active = k < pairs && left_list[k] < right_list[k];
  if (active) {
    const uint32_t l = left_list[k], r = right_list[k];
    const int32_t t = idx[l];
    idx[l] = idx[r];
    idx[r] = t;
  } else if (k == 0u || left_list[k - 1u] < right_list[k - 1u]) {  // k = K
    const uint32_t l_k = k < n_left ? left_list[k] : 0xFFFFFFFFu;
    *cut = k == 0u ? l_k : (l_k < right_list[k - 1u] ? l_k : right_list[k - 1u]);
  }
}

// Device buffers of one build (freed on every path).
struct BuildBuffers {
  float* pts = nullptr;
  int32_t* idx = nullptr;
  uint32_t *flags = nullptr, *sums = nullptr, *from_left = nullptr, *from_right = nullptr;
  BuildSeg* segs = nullptr;
  float* partial = nullptr;
  void* scan_tmp = nullptr;
  // the sliding step (allocated on the first slide)
  unsigned long long *flags64 = nullptr, *sums64 = nullptr;
  void* scan_tmp64 = nullptr;
  size_t scan_bytes64 = 0;
  float* pivot = nullptr;  // {pivot value, cut (as uint32)}
  ~BuildBuffers() {
    for (void* p : {(void*)pts, (void*)idx, (void*)flags, (void*)sums, (void*)from_left, (void*)from_right, (void*)segs,
                    (void*)partial, scan_tmp, (void*)flags64, (void*)sums64, scan_tmp64, (void*)pivot})
      if (p != nullptr) (void)hipFree(p);
  }
};

// Builds the sliding-midpoint tree of `points` with the partitions of the large nodes made on the current device.
// Returns false (and leaves `tree` alone) when the build has to be made on the host instead: a HIP failure, a
// plane that slides in the top levels, an unsupported shape.  `top_ms` (optional): time spent in the device levels.
template <typename SpaceView_, typename Tree_>
bool device_top_build(const float* points, uint64_t n, uint32_t dim, size_t max_leaf_size, unsigned threads,
                      SpaceView_ const& view, Tree_& tree, double* phase_ms = nullptr, const char** why = nullptr) {
  using namespace pico_tree;
  using box_type = typename Tree_::box_type;
  const char* unused = nullptr;
  const char*& reason = why != nullptr ? *why : unused;
  reason = "shape not supported";
  if (dim == 0 || dim > kBuildMaxDim || n < 2 || n >= (1ull << 31)) return false;
  const auto t_start = std::chrono::steady_clock::now();
  const bool verbose = std::getenv("PTK_CREATE_TIMING") != nullptr && std::atoi(std::getenv("PTK_CREATE_TIMING")) > 1;
  auto t_last = t_start;
  auto lap = [&](const char* what) {
    if (!verbose) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[ptk build]   %-34s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  double slide_ms = 0.0;
  auto ok = [&reason](hipError_t e) {
    if (e == hipSuccess) return true;
    reason = hipGetErrorString(e);
    (void)hipGetLastError();
    return false;
  };
  const uint32_t n32 = (uint32_t)n;
  // Nodes above this many points are split here: about eight subtrees per worker, none smaller than 16 k points.
  const size_t threshold = std::max<size_t>(std::max<size_t>(16384, max_leaf_size), n / ((size_t)std::max(1u, threads) * 8));
  const size_t max_segs = 2 * (n / threshold) + 64;
  BuildBuffers b;
  size_t scan_bytes = 0;
  if (!ok(rocprim::exclusive_scan(nullptr, scan_bytes, b.flags, b.sums, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(),
                                  (hipStream_t) nullptr)))
    return false;
  if (!ok(hipMalloc((void**)&b.pts, n * dim * sizeof(float))) || !ok(hipMalloc((void**)&b.idx, n * 4)) ||
      !ok(hipMalloc((void**)&b.flags, (n + 1) * 4)) || !ok(hipMalloc((void**)&b.sums, (n + 1) * 4)) ||
      !ok(hipMalloc((void**)&b.from_left, n * 4)) || !ok(hipMalloc((void**)&b.from_right, n * 4)) ||
      !ok(hipMalloc((void**)&b.segs, max_segs * sizeof(BuildSeg))) ||
      !ok(hipMalloc((void**)&b.partial, (size_t)kBuildBoxBlocks * dim * 2 * sizeof(float))) ||
      !ok(hipMalloc(&b.scan_tmp, scan_bytes + 256)))
    return false;
  lap("device buffers");
  if (!ok(hipMemcpy(b.pts, points, n * dim * sizeof(float), hipMemcpyHostToDevice))) return false;
  lap("points to the device");
  const uint32_t blocks = (n32 + 256u) / 256u;  // covers position n as well
  hipLaunchKernelGGL(build_iota_kernel, dim3(blocks), dim3(256), 0, nullptr, b.idx, n32);
  hipLaunchKernelGGL(build_box_kernel, dim3(kBuildBoxBlocks), dim3(256), 0, nullptr, b.pts, n32, dim, b.partial);
  if (!ok(hipGetLastError())) return false;  // (a launch that failed: the caller builds on the host instead)
  std::vector<float> partial((size_t)kBuildBoxBlocks * dim * 2);
  if (!ok(hipMemcpy(partial.data(), b.partial, partial.size() * sizeof(float), hipMemcpyDeviceToHost))) return false;
  box_type root(dim);
  root.invert();
  for (uint32_t blk = 0; blk < kBuildBoxBlocks; ++blk)
    for (uint32_t a = 0; a < dim; ++a) {
      const float lo = partial[((size_t)blk * dim + a) * 2], hi = partial[((size_t)blk * dim + a) * 2 + 1];
      if (lo < root.min(a)) root.min(a) = lo;
      if (hi > root.max(a)) root.max(a) = hi;
    }

  std::vector<BuildSeg> host_segs;
  auto on_device = [&](std::vector<internal::top_segment<float>>& segments) -> bool {
    if (segments.size() > max_segs) {
      reason = "more nodes in a level than planned for";
      return false;
    }
    const uint32_t nsegs = (uint32_t)segments.size();
    host_segs.resize(nsegs);
    for (uint32_t s = 0; s < nsegs; ++s)
      host_segs[s] = BuildSeg{(uint32_t)segments[s].begin, (uint32_t)segments[s].end, segments[s].axis, segments[s].plane, 0, 0, 0, 0};
    if (!ok(hipMemcpy(b.segs, host_segs.data(), nsegs * sizeof(BuildSeg), hipMemcpyHostToDevice))) return false;
    hipLaunchKernelGGL(build_flag_kernel, dim3(blocks), dim3(256), 0, nullptr, b.pts, dim, b.idx, b.segs, nsegs, n32, b.flags);
    size_t bytes = scan_bytes + 256;
    if (!ok(rocprim::exclusive_scan(b.scan_tmp, bytes, b.flags, b.sums, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(),
                                    (hipStream_t) nullptr)))
      return false;
    hipLaunchKernelGGL(build_cut_kernel, dim3((nsegs + 63) / 64), dim3(64), 0, nullptr, b.sums, b.segs, nsegs);
    hipLaunchKernelGGL(build_list_kernel, dim3(blocks), dim3(256), 0, nullptr, b.flags, b.sums, b.segs, nsegs, n32, b.from_left,
                       b.from_right);
    hipLaunchKernelGGL(build_swap_kernel, dim3(blocks), dim3(256), 0, nullptr, b.idx, b.segs, nsegs, n32, b.from_left,
                       b.from_right);
    if (!ok(hipGetLastError())) return false;
    if (!ok(hipMemcpy(host_segs.data(), b.segs, nsegs * sizeof(BuildSeg), hipMemcpyDeviceToHost))) return false;
    for (uint32_t s = 0; s < nsegs; ++s) segments[s].cut = segments[s].begin + host_segs[s].m;
    return true;
  };
  // A plane that slides: the node's range comes to the host, std::nth_element runs on (coordinate, index) pairs -- the
  // comparisons, and with them the permutation, are those of the builder's nth_element on the indices -- and goes back.
  std::vector<int> slid;
  std::vector<std::pair<float, int>> keyed;
  unsigned n_slides = 0;
  auto slide = [&](internal::top_segment<float>& seg, size_t nth) -> bool {
    if (++n_slides > 64) {  // (a pile of coincident outliers peels one point per level: the host builder's business)
      reason = "too many sliding planes in the top levels";
      return false;
    }
    const auto t_in = std::chrono::steady_clock::now();
    struct on_exit {
      double& sum;
      std::chrono::steady_clock::time_point t0;
      ~on_exit() { sum += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
    } timer{slide_ms, t_in};
    const uint32_t axis = seg.axis;
    // The rounds of libstdc++'s introselect while the range is large (see slide_pivot_kernel): on the device.
    constexpr size_t kHostBelow = 32768;
    // (The device rounds replay libstdc++'s std::nth_element and hand what is left to its std::__introselect -- internals
    // whose shape is pinned to the releases this was checked against.  Any other standard library, or a libstdc++ from
    // after them, takes the whole range to the host's std::nth_element below: the same permutation, no device rounds.)
#if defined(__GLIBCXX__) && defined(_GLIBCXX_RELEASE) && _GLIBCXX_RELEASE >= 9 && _GLIBCXX_RELEASE <= 15
    constexpr bool kKnownIntroselect = true;
#else
    constexpr bool kKnownIntroselect = false;
#endif
    if (kKnownIntroselect && seg.end - seg.begin > kHostBelow && knob_int("device_slide_off", 0) == 0) {
      size_t first = seg.begin, last = seg.end;
      const size_t nth_pos = seg.begin + nth;
      long depth_limit = 2 * (long)std::__lg((long)(last - first));
      if (b.flags64 == nullptr) {
        if (!ok(rocprim::exclusive_scan(nullptr, b.scan_bytes64, b.flags64, b.sums64, 0ull, (size_t)n + 1,
                                        rocprim::plus<unsigned long long>(), (hipStream_t) nullptr)) ||
            !ok(hipMalloc((void**)&b.flags64, (n + 1) * 8)) || !ok(hipMalloc((void**)&b.sums64, (n + 1) * 8)) ||
            !ok(hipMalloc(&b.scan_tmp64, b.scan_bytes64 + 256)) || !ok(hipMalloc((void**)&b.pivot, 16)))
          return false;
      }
      uint32_t* d_cut = reinterpret_cast<uint32_t*>(b.pivot) + 1;
      while (last - first > kHostBelow && depth_limit > 0) {
        --depth_limit;
        const uint32_t lo = (uint32_t)first + 1u, hi = (uint32_t)last, n_range = hi - lo;
        const uint32_t none = 0xFFFFFFFFu;
        if (!ok(hipMemcpyAsync(d_cut, &none, 4, hipMemcpyHostToDevice, nullptr))) return false;
        hipLaunchKernelGGL(slide_pivot_kernel, dim3(1), dim3(1), 0, nullptr, b.pts, dim, axis, b.idx, (uint32_t)first,
                           (uint32_t)last, b.pivot);
        hipLaunchKernelGGL(slide_flag_kernel, dim3(n_range / 256u + 1u), dim3(256), 0, nullptr, b.pts, dim, axis, b.idx, lo,
                           hi, b.pivot, b.flags64);
        size_t bytes = b.scan_bytes64 + 256;
        if (!ok(rocprim::exclusive_scan(b.scan_tmp64, bytes, b.flags64, b.sums64, 0ull, (size_t)n_range + 1,
                                        rocprim::plus<unsigned long long>(), (hipStream_t) nullptr)))
          return false;
        hipLaunchKernelGGL(slide_list_kernel, dim3(n_range / 256u + 1u), dim3(256), 0, nullptr, b.flags64, b.sums64, lo, hi,
                           b.from_left, b.from_right);
        hipLaunchKernelGGL(slide_pair_kernel, dim3((n_range + 1u) / 256u + 1u), dim3(256), 0, nullptr, b.idx, b.sums64,
                           n_range, b.from_left, b.from_right, d_cut);
        if (!ok(hipGetLastError())) return false;
        uint32_t cut = none;
        if (!ok(hipMemcpy(&cut, d_cut, 4, hipMemcpyDeviceToHost))) return false;
        if (cut == none || cut < lo || cut > hi) {
          reason = "the partition of a sliding step found no cut";
          return false;
        }
        if (cut <= nth_pos) first = cut;
        else last = cut;
      }
      // What is left of the same call, on the host: the range that still holds nth, the depth limit that is left.
      const size_t count = last - first;
      slid.resize(count);
      if (!ok(hipMemcpy(slid.data(), b.idx + first, count * 4, hipMemcpyDeviceToHost))) return false;
      auto below = [&](int x, int y) { return points[(size_t)x * dim + axis] < points[(size_t)y * dim + axis]; };
#if defined(__GLIBCXX__) && defined(_GLIBCXX_RELEASE) && _GLIBCXX_RELEASE >= 9 && _GLIBCXX_RELEASE <= 15
      std::__introselect(slid.begin(), slid.begin() + (nth_pos - first), slid.end(), depth_limit,
                         __gnu_cxx::__ops::__iter_comp_iter(below));
#else
      (void)below;
      reason = "the device rounds of a sliding step need libstdc++'s introselect";
      return false;
#endif
      seg.plane = points[(size_t)slid[nth_pos - first] * dim + axis];
      return ok(hipMemcpy(b.idx + first, slid.data(), count * 4, hipMemcpyHostToDevice));
    }
    const size_t count = seg.end - seg.begin;
    slid.resize(count);
    keyed.resize(count);
    if (!ok(hipMemcpy(slid.data(), b.idx + seg.begin, count * 4, hipMemcpyDeviceToHost))) return false;
    auto fill = [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) keyed[i] = std::make_pair(points[(size_t)slid[i] * dim + axis], slid[i]);
    };
    const unsigned workers = (unsigned)std::min<size_t>(std::max(1u, threads), count / 65536 + 1);
    if (workers > 1) {
      std::vector<std::thread> pool;
      const size_t per = (count + workers - 1) / workers;
      for (unsigned w = 1; w < workers; ++w) pool.emplace_back(fill, std::min(count, per * w), std::min(count, per * (w + 1)));
      fill(0, std::min(count, per));
      for (auto& t : pool) t.join();
    } else {
      fill(0, count);
    }
    std::nth_element(keyed.begin(), keyed.begin() + nth, keyed.end(),
                     [](const std::pair<float, int>& x, const std::pair<float, int>& y) { return x.first < y.first; });
    for (size_t i = 0; i < count; ++i) slid[i] = keyed[i].second;
    seg.plane = keyed[nth].first;
    return ok(hipMemcpy(b.idx + seg.begin, slid.data(), count * 4, hipMemcpyHostToDevice));
  };
  std::vector<internal::top_branch<float>> top;
  std::vector<std::pair<size_t, size_t>> frontier;
  reason = "a level of the top failed";
  if (!internal::split_top_levels(root, (size_t)n, threshold, on_device, slide, top, frontier)) return false;
  lap("levels (slides included)");
  if (verbose) std::fprintf(stderr, "[ptk build]   %-34s %8.2f ms (%zu top branches, %zu subtrees)\n", "of which slides", slide_ms, top.size(), frontier.size());
  std::vector<int> indices(n);
  if (!ok(hipMemcpy(indices.data(), b.idx, n * 4, hipMemcpyDeviceToHost))) return false;
  lap("permutation to the host");
  const auto t_top = std::chrono::steady_clock::now();
  double below_ms[2] = {0, 0};
  tree = internal::build_flat_tree_below<int>(view, max_leaf_size_t(max_leaf_size), sliding_midpoint_max_side, root,
                                              std::move(indices), top, frontier, true, threads, below_ms);
  if (verbose) std::fprintf(stderr, "[ptk build]   %-34s %8.2f ms\n[ptk build]   %-34s %8.2f ms\n", "subtrees (workers)", below_ms[0], "splice", below_ms[1]);
  if (phase_ms != nullptr) {
    phase_ms[0] = std::chrono::duration<double, std::milli>(t_top - t_start).count();
    phase_ms[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_top).count();
  }
  return true;
}

}  // namespace ptk
