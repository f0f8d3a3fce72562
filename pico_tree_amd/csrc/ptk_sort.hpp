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
// with nothing to clear, no look-back and no spinning: every dependency is a kernel boundary.  (r05 built the other form
// -- digit totals of all passes counted beside the keys, then ONE kernel per pass that takes a ticket, publishes its
// tile's digit counts and looks back over its predecessors' -- 5 launches instead of 10, every pass's input read once:
// correct and SLOWER, 7.2 M rows 0.206 -> 0.266 ms, 2 M 0.098 -> 0.126, 900 k +5 us: a look-back step is an L2 round
// trip across XCDs that every tile pays in line.  profiles/r05_notes.txt item 5; the code is in the history.)  Stable, so the result
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
    uint32_t* __restrict__ hist, CellTable cells = CellTable{}, const uint32_t* __restrict__ as_given = nullptr) {
  if (as_given != nullptr && *as_given != 0u) return;  // (uniform) a coherent batch is not sorted: coherence_sample_kernel
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
PTK_GLOBAL __launch_bounds__(64) void radix_scan_kernel(uint32_t* __restrict__ hist, uint32_t tiles, uint32_t stride,
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
    const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals, const uint32_t* __restrict__ as_given = nullptr) {
  if (as_given != nullptr && *as_given != 0u) return;  // (uniform) a coherent batch is not sorted
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

// ---- the same passes with BLOCKS of eight wavefronts and tiles of 4 096 items (r04) --------------------------------
//
// What the one-wavefront scatter above pays for is its stores: every item goes to its place on its own, 8 bytes into a
// 32-byte sector (7.2 M queries: 0.32 ms for three passes against 0.26 for rocprim's onesweep, whose blocks put a tile
// in digit order in LDS first).  Here a tile is 4 096 items: the eight wavefronts of a block rank an eighth each (same
// ballots, their digit counts side by side in LDS), the parts are joined by a prefix over the wavefronts, the
// items are put in digit order in LDS and leave with consecutive lanes on consecutive items -- runs of 16 items of a
// digit on average, whole sectors.  Histogram (LDS atomics: the order of counting does not matter), scan (the kernel
// above) and scatter stay three launches per pass with nothing to clear and no look-back.  Stable like the passes above:
// the permutation is the one a stable sort of the keys gives.
// (threads x items per thread, reorder ms of 7.2 M rows: 256 x 16 0.205, 512 x 8 0.192, 512 x 12 0.211, 256 x 8 0.205)
constexpr uint32_t kSortBlock = 512, kSortItems = 8, kSortTile = kSortBlock * kSortItems;
constexpr uint32_t sort_scatter_lds(uint32_t block, uint32_t items) {
  return ((block / 64u) * kRadixBins + 2u * kRadixBins + 32u) * 4u + block * items * 8u;
}
constexpr uint32_t kSortScatterLds = sort_scatter_lds(kSortBlock, kSortItems);

// Exclusive prefix of `v` over the 256 values held by the first 256 threads of the block (tmp: 4 words of LDS; two
// barriers; every thread of the block calls it, threads beyond 255 with v = 0).
__device__ __forceinline__ uint32_t block_exclusive_sum(uint32_t v, uint32_t t, PTK_LDS uint32_t* tmp) {
  const uint32_t lane = t & 63u, w = t >> 6;
  const uint32_t incl = wave_inclusive_sum(v, lane);
  if (lane == 63u && w < 4u) tmp[w] = incl;
  __syncthreads();
  uint32_t before = 0;
#pragma unroll
  for (uint32_t j = 0; j < 3; ++j) before += j < w ? tmp[j] : 0u;
  __syncthreads();
  return before + incl - v;
}

// Digit histogram of tile blockIdx.x -> hist[digit * stride + blockIdx.x]; FROM_QUERIES as radix_hist_kernel.
template <bool FROM_QUERIES, uint32_t BLOCK = kSortBlock, uint32_t ITEMS = kSortItems>
__global__ __launch_bounds__(BLOCK) void radix_block_hist_kernel(
    const float* __restrict__ queries, uint32_t dim, uint32_t nq, float3 lo, float3 inv, uint3 bits,
    uint32_t* __restrict__ keys, const uint2* __restrict__ pairs, uint32_t shift, uint32_t stride,
    uint32_t* __restrict__ hist, CellTable cells = CellTable{}, const uint32_t* __restrict__ as_given = nullptr) {
  if (as_given != nullptr && *as_given != 0u) return;  // (uniform) a coherent batch is not sorted: coherence_sample_kernel
  typedef PTK_LDS uint32_t LdsU32;
  LdsU32* cnt = (LdsU32*)ptk_smem;  // [256]
  const uint32_t t = threadIdx.x;
  if (t < kRadixBins) cnt[t] = 0u;
  __syncthreads();
  const uint32_t base = blockIdx.x * (BLOCK * ITEMS);
  uint32_t key[ITEMS];
  bool valid[ITEMS];
  if (FROM_QUERIES) {
    float x[ITEMS], y[ITEMS], z[ITEMS];  // (all the rows of a thread in flight together)
#pragma unroll
    for (uint32_t j = 0; j < ITEMS; ++j) {
      const uint32_t i = base + j * BLOCK + t;
      valid[j] = i < nq;
      load_query(queries, dim, valid[j] ? i : nq - 1u, x[j], y[j], z[j]);
    }
#pragma unroll
    for (uint32_t j = 0; j < ITEMS; ++j) {
      key[j] = order_key(x[j], y[j], z[j], lo, inv, bits, cells);
      if (valid[j]) keys[base + j * BLOCK + t] = key[j];
    }
  } else {
#pragma unroll
    for (uint32_t j = 0; j < ITEMS; ++j) {
      const uint32_t i = base + j * BLOCK + t;
      valid[j] = i < nq;
      key[j] = valid[j] ? pairs[i].x : 0u;
    }
  }
#pragma unroll
  for (uint32_t j = 0; j < ITEMS; ++j) {
    if (valid[j]) lds_add_u32(&cnt[(key[j] >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (t < kRadixBins) hist[t * stride + blockIdx.x] = cnt[t];
}

// The stable scatter of tile blockIdx.x; FIRST / LAST as radix_scatter_kernel.  hist and totals as the scan left them.
template <bool FIRST, bool LAST, uint32_t BLOCK = kSortBlock, uint32_t ITEMS = kSortItems>
__global__ __launch_bounds__(BLOCK) void radix_block_scatter_kernel(
    const uint32_t* __restrict__ keys_in, const uint2* __restrict__ pairs_in, uint2* __restrict__ pairs_out,
    uint32_t* __restrict__ vals_out, uint32_t n, uint32_t shift, uint32_t stride, const uint32_t* __restrict__ hist,
    const uint32_t* __restrict__ totals, const uint32_t* __restrict__ as_given = nullptr) {
  if (as_given != nullptr && *as_given != 0u) return;  // (uniform) a coherent batch is not sorted
  constexpr uint32_t WAVES = BLOCK / 64u, TILE = BLOCK * ITEMS;
  static_assert(BLOCK >= kRadixBins && BLOCK % 64u == 0u, "a thread per digit");
  typedef PTK_LDS uint32_t LdsU32;
  LdsU32* wave_cnt = (LdsU32*)ptk_smem;            // [WAVES][256] digit counts of each wavefront's part, then their prefix
  LdsU32* lbase = wave_cnt + WAVES * kRadixBins;    // [256] where a digit begins in the tile
  LdsU32* gbase = lbase + kRadixBins;               // [256] where this tile's items of a digit begin in the output
  LdsU32* tmp = gbase + kRadixBins;                 // [32]
  LdsWord* buf = (LdsWord*)(tmp + 32);              // [TILE] the tile in digit order
  const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
  const uint32_t base = blockIdx.x * TILE;
  for (uint32_t j = t; j < WAVES * kRadixBins; j += BLOCK) wave_cnt[j] = 0u;
  // This wavefront's part of the tile, in index order: round r = items [w * ITEMS * 64 + r * 64, + 64).
  uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
  bool valid[ITEMS];
#pragma unroll
  for (uint32_t r = 0; r < ITEMS; ++r) {
    const uint32_t i = base + w * (ITEMS * 64u) + r * 64u + lane;
    valid[r] = i < n;
    if (FIRST) {
      key[r] = valid[r] ? keys_in[i] : 0u;
      val[r] = i;
    } else {
      const uint2 p = valid[r] ? pairs_in[i] : make_uint2(0u, 0u);
      key[r] = p.x;
      val[r] = p.y;
    }
  }
  __syncthreads();
  // Rank among the items of the same digit in this part (the LDS operations of a wavefront are executed in order:
  // a round reads what the round before wrote).
  const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
  for (uint32_t r = 0; r < ITEMS; ++r) {
    const uint32_t digit = (key[r] >> shift) & 255u;
    const uint64_t peers = same_digit_lanes(digit, valid[r]);
    const uint32_t before = wave_cnt[w * kRadixBins + (valid[r] ? digit : 0u)];
    rank[r] = before + (uint32_t)__popcll(peers & below);
    if (valid[r] && (peers >> lane) == 1ull) wave_cnt[w * kRadixBins + digit] = before + (uint32_t)__popcll(peers);
  }
  __syncthreads();
  {
    // digit t: the parts' counts become their prefix; the tile's count gives the digit's place in the tile; the totals
    // and the scanned histogram its place in the output
    uint32_t sum = 0;
    if (t < kRadixBins) {
#pragma unroll
      for (uint32_t j = 0; j < WAVES; ++j) {
        const uint32_t c = wave_cnt[j * kRadixBins + t];
        wave_cnt[j * kRadixBins + t] = sum;
        sum += c;
      }
    }
    const uint32_t in_tile = block_exclusive_sum(t < kRadixBins ? sum : 0u, t, tmp);
    const uint32_t in_all = block_exclusive_sum(t < kRadixBins ? totals[t] : 0u, t, tmp + 8);
    if (t < kRadixBins) {
      lbase[t] = in_tile;
      gbase[t] = in_all + hist[t * stride + blockIdx.x];
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t r = 0; r < ITEMS; ++r) {
    if (valid[r]) {
      const uint32_t digit = (key[r] >> shift) & 255u;
      buf[lbase[digit] + wave_cnt[w * kRadixBins + digit] + rank[r]] = (unsigned long long)key[r] | ((unsigned long long)val[r] << 32);
    }
  }
  __syncthreads();
  const uint32_t held = n - base < TILE ? n - base : TILE;
#pragma unroll
  for (uint32_t j = 0; j < ITEMS; ++j) {
    const uint32_t at = j * BLOCK + t;
    if (at < held) {
      const unsigned long long kv = buf[at];
      const uint32_t digit = ((uint32_t)kv >> shift) & 255u;
      const uint32_t dst = gbase[digit] + (at - lbase[digit]);
      if (LAST) vals_out[dst] = (uint32_t)(kv >> 32);
      else pairs_out[dst] = make_uint2((uint32_t)kv, (uint32_t)(kv >> 32));
    }
  }
}

}  // namespace ptk
