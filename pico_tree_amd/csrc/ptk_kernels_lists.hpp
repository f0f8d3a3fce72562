// ptk_kernels_lists.hpp -- the radius search of the 3-D kernels with the rows made from LEAF LISTS.
//
// The reference's radius visitor (search_visitor.hpp:127-156) never changes its bound: which leaves a query visits,
// and in which order, is decided by the traversal alone (kd_tree_search.hpp:52-105), and what a visited leaf adds to
// the row is decided by the leaf alone (its points in index order, :54-59, those with `radius > distance`, :141).
// So the two passes of the C ABI divide the work where it stops being divergent:
//
//   count    (radius_list_kernel)   one query per lane, the reference traversal with the count visitor.  Besides the
//            count, every leaf that holds a hit is appended to the lane's list, in visit order, with the mask of
//            the points that are hits: 8 bytes per leaf instead of 8 bytes per hit.  The lists of a wavefront share
//            chunks of 16 entries x 64 lanes, laid out [group of 4][lane][4]: a lane collects four entries in registers and
//            writes them as one 32-byte sector (a store inside the traversal loop delays the next node load: stores
//            and loads share a counter on gfx9), and the fill pass reads a group of all 64 lanes as 2 KB of contiguous
//            memory.  A wavefront owns one static chunk and takes further ones from the pools of the capture block
//            as its longest list grows.
//   fill     (radius_replay_kernel) one query per lane again, but every lane now walks its LIST (the groups of all 64
//            lanes are coalesced loads, fetched one group ahead): only the points that ARE hits are fetched, HITS of
//            them at a time across the entries of a group, their distances computed with
//            the arithmetic of the leaf scan of traverse<> (same operations in the same order: bit-identical), and
//            all lanes are busy for as long as the longest list of the wavefront lasts -- neighbours in the Morton
//            order have lists of similar length.  The hits go to the row in list order = the reference's visit
//            order.  A lane's hits wait in a ring of 32 entries in LDS until they complete a 64-byte line of the
//            row; the complete lines of all lanes leave together, four lanes to a line, so that HBM sees whole
//            lines: a row is written once, by whole lines but for its first and last one, and nothing else is
//            written at all -- no log of hits.
//
// Against the capture log this replaces on 3-D trees (radius_capture_kernel + radius_log_scatter_kernel: 8.5 GB of
// log written, 8.4 GB read back, 7.7 GB of rows written for 6.06 GB of rows on BASELINE config 3) the lists are
// 8 bytes per leaf with a hit (1.3 GB), written once and read once.
//
// A wavefront whose lists could not grow (pools exhausted, or a list beyond kListMaxChunks chunks) is marked; its
// queries are listed by the fill pass and searched by radius_kernel<FILL>, exactly as a failed capture was.
#pragma once

#include "ptk_kernels.hpp"

namespace ptk {

constexpr uint32_t kListSlots = kLogChunk / 64u;  // list entries per lane and chunk (16; an entry is a slot of the log)
constexpr uint32_t kListGroup = 4;                // entries a lane writes (and the fill pass reads) together: 32 bytes
constexpr uint32_t kListGroups = kListSlots / kListGroup;
constexpr uint32_t kListMaxChunks = 64;           // chunks per wavefront: 1024 listed leaves per query
constexpr uint32_t kListMaskBits = 32;            // points per entry (a larger leaf is listed in pieces)

#if defined(__HIP_DEVICE_COMPILE__)
typedef volatile PTK_LDS uint32_t ListTable;
#else
typedef uint32_t ListTable;
#endif

struct alignas(16) ListPair {
  unsigned long long a, b;
};
// Where group g of a lane begins in the chunk array (in entries).
__device__ __forceinline__ uint64_t list_group_at(uint32_t chunk, uint32_t g, uint32_t lane) {
  return (uint64_t)chunk * kLogChunk + (uint64_t)g * (64u * kListGroup) + lane * kListGroup;
}

// An entry: {(first point << cbits) | points, mask of the hits among them}.
__device__ __forceinline__ unsigned long long pack_list_entry(uint32_t ref, uint32_t mask) {
  return (unsigned long long)ref | ((unsigned long long)mask << 32);
}

// BIG: the tree has leaves of more than kListMaskBits points (a leaf is then listed in pieces; the test for the end of
// a piece is four instructions of every point visit).  EXACT: e = 1 -- the approximate visitor's multiplication by 1 / e
// (:265) is the identity then and is left out.  The launch picks the instantiation (uniform for a tree and a call):
// 8 % of the kernel's vector instructions on BASELINE config 3 are these two.
template <bool BIG, bool EXACT>
struct RadiusListPolicy {  // search_visitor.hpp:127-156 / :252-288, counting
  static constexpr bool kLeafHooks = true;
  float radius;  // already scaled by 1/e for the approximate search (:265)
  float e_inv;
  uint64_t count;
  uint32_t n;            // entries this lane has listed
  uint32_t first, pos, mask, cbits;  // the piece of a leaf being measured: its first point, points seen, hits among them
  unsigned long long* slots;  // the chunks
  unsigned long long b0, b1, b2, b3;  // the group of entries this lane is collecting
  uint32_t* counters;         // RadiusCapture::counters
  ListTable* table;           // LDS: [0] = chunks the wavefront holds (kLogEnd: its lists are lost), [1 + c] = chunk c
  uint32_t sub, sub_cap, n_static, lane;

  __device__ __forceinline__ float max() const { return radius; }
  // One lane: the wavefront's next chunk (number `have`).
  __device__ __forceinline__ void grow(uint32_t have) {
    uint32_t id = kLogEnd;
    if (have < kListMaxChunks) {
      const uint32_t nx = atomicAdd(&counters[sub * kCapCounterStride], 1u);
      if (nx < sub_cap) id = n_static + sub * sub_cap + nx;
    }
    if (id == kLogEnd) {
      table[0] = kLogEnd;
    } else {
      table[1u + have] = id;
      table[0] = have + 1u;
    }
  }
  // The group of four entries that ends at entry n (or the last, incomplete one) leaves the LDS buffer.
  __device__ __forceinline__ void write_group(uint32_t group) {
    const uint32_t c = group / kListGroups;
#if defined(__HIP_DEVICE_COMPILE__)
    // The lanes that write together: the first one that needs a chunk the wavefront does not hold yet takes it
    // (LDS operations of a wavefront are executed in order: the next turn reads what this one wrote).
    for (;;) {
      const uint32_t have = table[0];
      const bool need = have != kLogEnd && c >= have;
      const uint64_t m = __ballot(need);
      if (m == 0ull) break;
      if (lane == (uint32_t)__builtin_ctzll(m)) grow(have);
    }
#else
    while (table[0] != kLogEnd && c >= table[0]) grow(table[0]);  // (the test tier runs one lane at a time)
#endif
    if (table[0] == kLogEnd) return;
    ListPair* at = reinterpret_cast<ListPair*>(slots + list_group_at(table[1u + c], group % kListGroups, lane));
    ListPair lo, hi;
    lo.a = b0;
    lo.b = b1;
    hi.a = b2;
    hi.b = b3;
    at[0] = lo;
    at[1] = hi;
  }
  __device__ __forceinline__ void append(unsigned long long entry) {
    const uint32_t k = n % kListGroup;
    b0 = k == 0u ? entry : b0;
    b1 = k == 1u ? entry : b1;
    b2 = k == 2u ? entry : b2;
    b3 = k == 3u ? entry : b3;
    ++n;
    if (k == kListGroup - 1u) write_group(n / kListGroup - 1u);
  }
  __device__ __forceinline__ void finish() {
    if (n % kListGroup != 0u) write_group(n / kListGroup);  // (the slots behind entry n - 1 are never read)
  }
  __device__ __forceinline__ void close_piece() {
    if (mask != 0u) {
      count += (uint32_t)__popcll((unsigned long long)mask);
      append(pack_list_entry((first << cbits) | pos, mask));
    }
    first += pos;
    pos = 0u;
    mask = 0u;
  }
  __device__ __forceinline__ void leaf_begin(uint32_t ref, const DevTree& t) {
    first = (ref & 0x7FFFFFFFu) >> t.cbits;
    pos = 0u;
    mask = 0u;
  }
  __device__ __forceinline__ void leaf_end() { close_piece(); }
  __device__ __forceinline__ void visit(int32_t, float d) {
    if constexpr (!EXACT) d = f_mul(d, e_inv);
    mask |= (radius > d ? 1u : 0u) << pos;  // strict; the hits are counted when the piece is closed
    ++pos;
    if constexpr (BIG) {
      if (pos == kListMaskBits) close_piece();  // (leaves of up to 32 points, the usual case: one piece)
    }
  }
};

// The count pass: see the head of this file.  One wavefront per block; LDS = the record stack and the chunk table.
constexpr uint32_t kListLds = (1u + kListMaxChunks) * 4u;  // behind the record stack
// CAPPED (ptk_kernels_coopr.hpp): a query that has entered more than `far_cap` far children stops -- its list and its
// count so far stay, its pending subtrees go to `ho` and a wavefront finishes it (radius_coop_count_kernel).
template <int S, int OVF, int LEAFB, class M = MetricL2, bool BIG = true, bool EXACT = false, bool CAPPED = false>
__global__ __launch_bounds__(64) void radius_list_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim, const uint32_t* __restrict__ perm, uint64_t nq,
    float radius, float e_inv, uint64_t* __restrict__ counts, RadiusCapture cap, uint32_t far_cap = 0u,
    Handover ho = Handover{}) {
  PTK_TRACE_BEGIN();
  const uint32_t tile = xcd_runs(blockIdx.x, gridDim.x, kXcdRunGeneral);
  const uint64_t i = (uint64_t)tile * 64u + threadIdx.x;
  ListTable* table = (ListTable*)(ptk_smem + (size_t)S * 64 * 8);
  if (threadIdx.x == 0) {
    table[0] = 1u;
    table[1] = tile;  // the static chunk
  }
  RadiusListPolicy<BIG, EXACT> pol;
  pol.n = 0u;
  if (i < nq) {
    const uint64_t qi = perm ? perm[i] : i;
    cap.qids[i] = (uint32_t)qi;
    float qx, qy, qz;
    load_query(queries, dim, qi, qx, qy, qz);
    pad_query<M>(dim, qy, qz);
    Record spill[OVF > 0 ? OVF : 1];
    Stack<S, OVF, 64> st;
    st.init((LdsWord*)ptk_smem, threadIdx.x, spill);
    pol.radius = f_mul(radius, e_inv);
    pol.e_inv = e_inv;
    pol.count = 0;
    pol.cbits = t.cbits;
    pol.slots = reinterpret_cast<unsigned long long*>(cap.chunks);
    pol.b0 = pol.b1 = pol.b2 = pol.b3 = 0ull;
    pol.counters = cap.counters;
    pol.table = table;
    pol.sub = (blockIdx.x * 0x9E3779B1u) >> 24;  // kCapSubPools = 256: the top byte of a hash of the block
    pol.sub_cap = cap.sub_cap;
    pol.n_static = cap.n_static;
    pol.lane = threadIdx.x;
    if constexpr (CAPPED) {
      ho.slot = (uint32_t)qi;
      traverse<LEAFB, false, M, true, true>(t, qx, qy, qz, pol, st, far_cap, &ho);
    } else {
      traverse<LEAFB, false, M>(t, qx, qy, qz, pol, st);
    }
    pol.finish();
    counts[qi] = pol.count;
  } else {
    cap.qids[i] = kLogEnd;
  }
  cap.lens[i] = pol.n;
#if defined(__HIP_DEVICE_COMPILE__)
  __syncthreads();  // (one wavefront: every lane is out of the traversal before the table is read)
  const uint32_t have = table[0];
  if (have != kLogEnd && threadIdx.x < have) cap.tables[(uint64_t)tile * kListMaxChunks + threadIdx.x] = table[1u + threadIdx.x];
  if (threadIdx.x == 0) cap.captured[tile] = have != kLogEnd ? 1 : 0;
  PTK_TRACE_END();
#else
  // (one lane at a time: the last lane's view of the table is the wavefront's)
  const uint32_t have = table[0];
  for (uint32_t c = 0; have != kLogEnd && c < have; ++c) cap.tables[(uint64_t)tile * kListMaxChunks + c] = table[1u + c];
  cap.captured[tile] = have != kLogEnd ? 1 : 0;
#endif
}

// The ring of the fill pass: RING entries per lane, [slot][lane] with a row of 65 entries (the four lanes that write
// a line of one row read its eight slots of ONE lane: a row of 64 would put them all in one bank).
constexpr uint32_t kRingRow = 65;
constexpr uint32_t replay_lds(uint32_t ring) { return ring * kRingRow * 8u + 64u * 8u + 64u * 4u; }  // ring, {row}, {lane | count << 8} of the lines that leave

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)v, d);
    v = o > v ? o : v;
  }
  return v;
}

// The fill pass: see the head of this file.  grid = cap.n_static blocks of one wavefront; HITS = hits fetched
// together, RING = entries a lane can hold back (a power of two >= HITS + 7 + 4).  The queries of wavefronts without
// lists go to over_list.
template <int HITS, int RING, class M = MetricL2>
__global__ __launch_bounds__(64) void radius_replay_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim, float e_inv, RadiusCapture cap,
    const uint64_t* __restrict__ offsets, Neighbor* __restrict__ out, uint32_t* __restrict__ over_list,
    uint32_t* __restrict__ n_over) {
  static_assert((RING & (RING - 1)) == 0 && RING >= 16, "the ring is addressed by the low bits of the row position");
  // A lane holds at most kHoldMax entries before a round of HITS more: what is beyond leaves first.
  constexpr uint32_t kHoldMax = RING - HITS;
  static_assert(kHoldMax >= 8u + 3u, "a lane that has just let its complete lines go (<= 7 left) must be below the mark");
  const uint32_t lane = threadIdx.x;
  const uint32_t tile = xcd_runs(blockIdx.x, gridDim.x, kXcdRunGeneral);
  const uint32_t qi = cap.qids[(uint64_t)tile * 64u + lane];
  const bool valid = qi != kLogEnd;
  if (!cap.captured[tile]) {  // (uniform)
    if (valid) over_list[atomicAdd(n_over, 1u)] = qi;
    return;
  }
  float qx, qy, qz;
  load_query(queries, dim, valid ? qi : 0u, qx, qy, qz);
  pad_query<M>(dim, qy, qz);
  const uint32_t n = valid ? cap.lens[(uint64_t)tile * 64u + lane] : 0u;
  const uint32_t groups = uniform_value((wave_max_u32(n) + kListGroup - 1u) / kListGroup);  // (a scalar: the loops below are the wavefront's)
  const ListPair* __restrict__ pairs = reinterpret_cast<const ListPair*>(cap.chunks);
  const uint32_t my_chunk = cap.tables[(uint64_t)tile * kListMaxChunks + lane];  // lane c: chunk c of the wavefront (beyond its last: never read)
  const float4* __restrict__ pts = t.pts;

  // `row` = where this lane's next entry to LEAVE goes (in entries from `out`), `held` = entries in the ring.  Entry
  // number x of the row sits in ring slot x % RING whether it is written or read: a line of the row (8 entries,
  // aligned) is eight consecutive slots.
  uint64_t row = valid ? offsets[qi] : 0ull;
  uint32_t held = 0u;
  LdsWord* ring = (LdsWord*)ptk_smem;
  LdsWord* line_row = ring + RING * kRingRow;
  PTK_LDS uint32_t* line_src = (PTK_LDS uint32_t*)(line_row + 64);
  unsigned long long* __restrict__ dst = reinterpret_cast<unsigned long long*>(out);

  // Lines that are complete (tail = false) or, after the last leaf, whatever is held (tail = true) leave the ring.
  auto flush = [&](bool tail) {
    for (;;) {
      const uint32_t inl = (uint32_t)row & 7u;
      const bool ready = tail ? held != 0u : inl + held >= 8u;
      const uint64_t m = __ballot(ready);
      if (m == 0ull) break;
      const uint32_t n_out = tail ? (held < 8u - inl ? held : 8u - inl) : 8u - inl;
      if (ready) {
        const uint32_t rank = lanes_below(m, lane);
        line_row[rank] = row;
        line_src[rank] = lane | (n_out << 8);
      }
      __syncthreads();  // (one wavefront: no instruction; the test tier's lanes meet here)
      const uint32_t lines = (uint32_t)__popcll(m);
      for (uint32_t l0 = 0; l0 < lines; l0 += 16u) {  // four lanes to a line, two entries (16 bytes) per lane
        const uint32_t line = l0 + (lane >> 2);
        if (line < lines) {
          const uint64_t r = line_row[line];
          const uint32_t sc = line_src[line];
          const uint32_t src = sc & 63u, cnt = sc >> 8, first = (uint32_t)r & 7u, e = (lane & 3u) * 2u;
          const bool v0 = e >= first && e < first + cnt, v1 = e + 1u >= first && e + 1u < first + cnt;
          const uint64_t x = (r & ~7ull) + e;
          const uint32_t at = ((uint32_t)x & (uint32_t)(RING - 1)) * kRingRow + src;
          if (v0 && v1) {
            ListPair two;
            two.a = ring[at];
            two.b = ring[at + kRingRow];
            *reinterpret_cast<ListPair*>(dst + x) = two;
          } else if (v0) {
            dst[x] = ring[at];
          } else if (v1) {
            dst[x + 1u] = ring[at + kRingRow];
          }
        }
      }
      __syncthreads();  // (the lines have been read before their slots are written again)
      if (ready) {
        row += n_out;
        held -= n_out;
      }
    }
  };

  auto load_group = [&](uint32_t g, ListPair& lo, ListPair& hi) {
    const uint32_t chunk = (uint32_t)__builtin_amdgcn_readlane((int)my_chunk, (int)(g / kListGroups));
    const ListPair* at = pairs + list_group_at(chunk, g % kListGroups, lane) / 2u;
    lo = at[0];
    hi = at[1];
  };
  ListPair lo = {}, hi = {};
  if (groups != 0u) load_group(0u, lo, hi);
  for (uint32_t g = 0; g < groups; ++g) {  // (uniform)
    // This group's entries; the next group's are on their way while they are worked through.
    const uint32_t f0 = ((uint32_t)lo.a & 0x7FFFFFFFu) >> t.cbits, f1 = ((uint32_t)lo.b & 0x7FFFFFFFu) >> t.cbits,
                   f2 = ((uint32_t)hi.a & 0x7FFFFFFFu) >> t.cbits, f3 = ((uint32_t)hi.b & 0x7FFFFFFFu) >> t.cbits;
    const uint32_t m0 = (uint32_t)(lo.a >> 32), m1 = (uint32_t)(lo.b >> 32), m2 = (uint32_t)(hi.a >> 32),
                   m3 = (uint32_t)(hi.b >> 32);
    if (g + 1u < groups) load_group(g + 1u, lo, hi);
    const uint32_t nv = n > g * kListGroup ? (n - g * kListGroup < kListGroup ? n - g * kListGroup : kListGroup) : 0u;
    // The hits of the group as one stream: entry k's from the lowest bit up, then entry k + 1's (every entry has one).
    uint32_t k = 0u, first = f0, mask = nv != 0u ? m0 : 0u;
    while (__ballot(mask != 0u || k + 1u < nv) != 0ull) {  // (uniform)
      bool hit[HITS];
      float4 p[HITS];
#pragma unroll
      for (int u = 0; u < HITS; ++u) {
        if (mask == 0u && k + 1u < nv) {
          ++k;
          first = k == 1u ? f1 : (k == 2u ? f2 : f3);
          mask = k == 1u ? m1 : (k == 2u ? m2 : m3);
        }
        hit[u] = mask != 0u;
        const uint32_t b = hit[u] ? (uint32_t)__builtin_ctz(mask) : 0u;
        mask &= mask - 1u;  // (0 stays 0)
        if (hit[u]) p[u] = pts[first + b];
      }
#pragma unroll
      for (int u = 0; u < HITS; ++u) {
        if (hit[u]) {
          PTK_KEEP4(p[u]);
          float dx = f_sub(qx, p[u].x), dy = f_sub(qy, p[u].y), dz = f_sub(qz, p[u].z);
          PTK_SCALAR(dx);
          PTK_SCALAR(dy);
          PTK_SCALAR(dz);
          Neighbor nb;
          nb.index = __float_as_int(p[u].w);
          nb.distance = f_mul(point_distance3<M>(dx, dy, dz), e_inv);
          ring[(((uint32_t)row + held) & (uint32_t)(RING - 1)) * kRingRow + lane] = pack_neighbor(nb);
          ++held;
        }
      }
      // (only when a lane could not take another round: the lines of all lanes then leave together)
      if (__ballot(held > kHoldMax) != 0ull) flush(false);
    }
  }
  flush(true);
}

}  // namespace ptk
