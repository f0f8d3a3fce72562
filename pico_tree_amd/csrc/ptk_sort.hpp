// ptk_sort.hpp -- the batch order of the searches: Morton keys + a stable LSD radix sort of (key, row) pairs,
// written for batches of 10^5 .. 10^7 queries where what counts is the NUMBER of dependent launches.
//
// rocprim's onesweep sort of a 900 k-query shard (one eighth of BASELINE config 2) is a key kernel, a histogram
// kernel, a scan, and per 8-bit pass two buffer fills and a pass whose decoupled look-back takes 26-28 us whatever
// the size: 10 launches and 96 us for two passes (profiles/r03a_shard_timeline.txt).  Here a pass is
//
//   histogram   one wavefront per tile of `tile` consecutive items: digit counts in LDS -> hist[digit][tile]
//               (the first pass computes the Morton keys in the same kernel: radix_hist_kernel<true>)
//   scan        one wavefront per digit: exclusive prefix over the tiles of its row, row total -> totals[digit]
//   scatter     one wavefront per tile: position = (digits below) + (same digit in earlier tiles) + (same digit
//               earlier in the tile); the rank inside a round of 64 items comes from eight ballots
//
// with nothing to clear, no look-back and no spinning: every dependency is a kernel boundary.  Stable, so the result
// is the permutation a stable sort of the keys gives (tests/test_kernel_emulation.py compares it with numpy).
// The caller's row order is untouched -- results are written by original row; the order only decides which
// queries share a wavefront, i.e. cache lines and the wave-uniform prefix of phase 1.

#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptk_kernels.hpp"

namespace ptk {

constexpr uint32_t kRadixBins = 256;  // 8 bits per pass

// Lanes of the wavefront holding the same 8-bit digit as this one (among `valid` lanes): eight ballots.
__device__ __forceinline__ uint64_t same_digit_lanes(uint32_t digit, bool valid) {
  uint64_t peers = __ballot(valid);
#pragma unroll
  for (uint32_t b = 0; b < 8; ++b) {
    const bool bit = ((digit >> b) & 1u) != 0u;
    const uint64_t m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

// Inclusive prefix sum over the 64 lanes of the wavefront.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)v, d);
    if ((int)lane >= d) v += up;
  }
  return v;
}

constexpr uint32_t kRadixGroup = 4;  // rounds of 64 items whose loads are issued together (one memory latency)

// Digit histogram of tile blockIdx.x (items [tile * blockIdx.x, +tile)) -> hist[digit * stride + blockIdx.x].
// FROM_QUERIES: the items are query points; their Morton keys are computed, stored to `keys` and counted.
// Otherwise the items are the (key, value) pairs a scatter pass wrote.
template <bool FROM_QUERIES>
__global__ __launch_bounds__(64) void radix_hist_kernel(
    const float* __restrict__ queries, uint32_t dim, uint32_t nq, float3 lo, float3 inv, uint3 bits,
    uint32_t* __restrict__ keys, const uint2* __restrict__ pairs, uint32_t shift, uint32_t tile, uint32_t stride,
    uint32_t* __restrict__ hist, CellTable cells = CellTable{}) {
  typedef PTK_LDS uint32_t LdsU32;
  LdsU32* cnt = (LdsU32*)ptk_smem;  // [256]
  const uint32_t lane = threadIdx.x;
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) cnt[lane + 64u * j] = 0u;
  const uint32_t base = blockIdx.x * tile;
  for (uint32_t j0 = 0; j0 < tile; j0 += 64u * kRadixGroup) {
    uint32_t key[kRadixGroup];
    bool valid[kRadixGroup];
    if (FROM_QUERIES) {
      float x[kRadixGroup], y[kRadixGroup], z[kRadixGroup];
#pragma unroll
      for (uint32_t u = 0; u < kRadixGroup; ++u) {
        const uint32_t i = base + j0 + 64u * u + lane;
        valid[u] = j0 + 64u * u < tile && i < nq;
        load_query(queries, dim, valid[u] ? i : nq - 1u, x[u], y[u], z[u]);
      }
#pragma unroll
      for (uint32_t u = 0; u < kRadixGroup; ++u) {
        key[u] = order_key(x[u], y[u], z[u], lo, inv, bits, cells);
        if (valid[u]) keys[base + j0 + 64u * u + lane] = key[u];
      }
    } else {
#pragma unroll
      for (uint32_t u = 0; u < kRadixGroup; ++u) {
        const uint32_t i = base + j0 + 64u * u + lane;
        valid[u] = j0 + 64u * u < tile && i < nq;
        key[u] = valid[u] ? pairs[i].x : 0u;
      }
    }
    // One LDS update per distinct digit of a round (by its highest lane): no atomics, and in the lane-by-lane
    // emulator of the test tier every reader of a counter has run before its writer.
#pragma unroll
    for (uint32_t u = 0; u < kRadixGroup; ++u) {
      const uint32_t digit = (key[u] >> shift) & 255u;
      const uint64_t peers = same_digit_lanes(digit, valid[u]);
      if (valid[u] && (peers >> lane) == 1ull) cnt[digit] += (uint32_t)__popcll(peers);
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) hist[(lane + 64u * j) * stride + blockIdx.x] = cnt[lane + 64u * j];
}

// Row blockIdx.x of hist (one digit: `tiles` counters, row stride a multiple of 4): exclusive prefix in place, row
// total -> totals[blockIdx.x].  1024 counters per step: a lane takes 16 consecutive ones (four 16-byte loads), one
// wavefront scan joins the 64 runs.
__global__ __launch_bounds__(64) void radix_scan_kernel(uint32_t* __restrict__ hist, uint32_t tiles, uint32_t stride,
                                                        uint32_t* __restrict__ totals) {
  const uint32_t lane = threadIdx.x;
  uint4* row = reinterpret_cast<uint4*>(hist + (uint64_t)blockIdx.x * stride);
  const uint32_t n4 = stride / 4u;
  uint32_t carry = 0;
  for (uint32_t c0 = 0; c0 < n4; c0 += 256u) {
    uint4 v[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const uint32_t i4 = c0 + lane * 4u + u;
      v[u] = i4 < n4 ? row[i4] : make_uint4(0u, 0u, 0u, 0u);
      // (what lies between `tiles` and the row stride was never written)
      const uint32_t e = i4 * 4u;
      v[u].x = e + 0u < tiles ? v[u].x : 0u;
      v[u].y = e + 1u < tiles ? v[u].y : 0u;
      v[u].z = e + 2u < tiles ? v[u].z : 0u;
      v[u].w = e + 3u < tiles ? v[u].w : 0u;
    }
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) sum += v[u].x + v[u].y + v[u].z + v[u].w;
    const uint32_t incl = wave_inclusive_sum(sum, lane);
    uint32_t run = carry + incl - sum;
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      uint4 o;
      o.x = run;
      o.y = o.x + v[u].x;
      o.z = o.y + v[u].y;
      o.w = o.z + v[u].z;
      run = o.w + v[u].w;
      const uint32_t i4 = c0 + lane * 4u + u;
      if (i4 < n4) row[i4] = o;
    }
    carry += (uint32_t)__shfl((int)incl, 63);
  }
  if (lane == 0u) totals[blockIdx.x] = carry;
}

// The stable scatter of tile blockIdx.x.  FIRST: the items are keys[] and their values the item numbers; otherwise
// (key, value) pairs.  LAST: only the values are written (the permutation); otherwise pairs.
template <bool FIRST, bool LAST>
__global__ __launch_bounds__(64) void radix_scatter_kernel(
    const uint32_t* __restrict__ keys_in, const uint2* __restrict__ pairs_in, uint2* __restrict__ pairs_out,
    uint32_t* __restrict__ vals_out, uint32_t n, uint32_t shift, uint32_t tile, uint32_t stride,
    const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals) {
  typedef PTK_LDS uint32_t LdsU32;
  LdsU32* pos = (LdsU32*)ptk_smem;  // [256] next free output position of every digit, for this tile
  const uint32_t lane = threadIdx.x;
  const uint32_t base = blockIdx.x * tile;
  auto load_group = [&](uint32_t j0, uint32_t (&key)[kRadixGroup], uint32_t (&val)[kRadixGroup], bool (&valid)[kRadixGroup]) {
#pragma unroll
    for (uint32_t u = 0; u < kRadixGroup; ++u) {
      const uint32_t i = base + j0 + 64u * u + lane;
      valid[u] = j0 + 64u * u < tile && i < n;
      if (FIRST) {
        key[u] = valid[u] ? keys_in[i] : 0u;
        val[u] = i;
      } else {
        const uint2 p = valid[u] ? pairs_in[i] : make_uint2(0u, 0u);
        key[u] = p.x;
        val[u] = p.y;
      }
    }
  };
  uint32_t key[kRadixGroup], val[kRadixGroup];
  bool valid[kRadixGroup];
  load_group(0u, key, val, valid);  // in flight while the digit positions are worked out
  {
    // digits 4 lane .. 4 lane + 3: where the digit starts in the output (exclusive prefix of the totals) + what
    // earlier tiles hold of it
    const uint4 t4 = reinterpret_cast<const uint4*>(totals)[lane];
    const uint32_t t[4] = {t4.x, t4.y, t4.z, t4.w};
    uint32_t h[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) h[j] = hist[(uint64_t)(4u * lane + j) * stride + blockIdx.x];
    const uint32_t sum = t[0] + t[1] + t[2] + t[3];
    uint32_t run = wave_inclusive_sum(sum, lane) - sum;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      pos[4u * lane + j] = run + h[j];
      run += t[j];
    }
  }
  __syncthreads();
  const uint64_t below = (1ull << lane) - 1ull;
  for (uint32_t j0 = 0; j0 < tile; j0 += 64u * kRadixGroup) {
    if (j0 != 0u) load_group(j0, key, val, valid);
#pragma unroll
    for (uint32_t u = 0; u < kRadixGroup; ++u) {
      const uint32_t digit = (key[u] >> shift) & 255u;
      const uint64_t peers = same_digit_lanes(digit, valid[u]);
      const uint32_t p = pos[valid[u] ? digit : 0u];
      if (valid[u]) {
        const uint32_t dst = p + (uint32_t)__popcll(peers & below);
        if (LAST) vals_out[dst] = val[u];
        else pairs_out[dst] = make_uint2(key[u], val[u]);
        if ((peers >> lane) == 1ull) pos[digit] = p + (uint32_t)__popcll(peers);  // the highest lane of the group
      }
    }
  }
}

}  // namespace ptk
