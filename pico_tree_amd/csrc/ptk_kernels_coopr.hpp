// ptk_kernels_coopr.hpp -- the radius search of the 3-D kernels: the LONG queries of a batch finished by a whole
// wavefront each (count pass and fill pass).
//
// What this is for (VERDICT r05 "missing" item 1, profiles/r05_notes.txt item 13).  The list
// pass and the replay of ptk_kernels_lists.hpp run every query to its end in one lane, and a lane whose wavefront has
// emptied takes ~4 us per leaf: on BASELINE config 3's cloud ANY batch took 2.2 ms (20 k queries 2.23 ms, 600 k 2.78),
// the duration of its longest query -- 1.7 ms of list pass and 0.6 ms of replay for one lane.  A shard of an 8-GPU
// run and a piece of a host-buffer call paid the full batch's longest query.
//
// So the list pass is CAPPED like the general k-NN kernel (traverse<.., CAPPED>): a query that has entered more than
// `cap` far children stops; the leaves it has listed so far stay in its lane's list (the main replay writes them),
// its count so far stays in counts[row], and its stack -- every pending far child that can still matter with the
// state it would be entered with (Task) -- goes to the hand-over list.  radius_coop_count_kernel then gives each
// such query a wavefront.
//
// Why a wavefront may search in ANY order here, without a certificate (unlike the k-NN searches of
// ptk_kernels_coopk.hpp): the reference's radius visitor never changes its bound (search_visitor.hpp:127-156,
// `max()` is the radius).  Whether a far child is entered is `radius >= node_box_distance`
// (kd_tree_search.hpp:99), and a node's box distance is a function of its root path alone (`nbd - old_offset +
// new_offset` with the offsets the path set, :91-97) -- Task carries {nbd, off[3]} down every path exactly as
// traverse<> does.  So the SET of leaves the reference visits below a pending subtree is decided node by node,
// whatever the order the nodes are looked at in, and so is what a leaf adds to the row (its points in index order,
// those with `radius > distance`, :141).  Only the ORDER of the leaves in the row is the traversal's: depth-first,
// the nearer child first.  Every subtree in the shared pool therefore carries a KEY -- the handed-over task it
// belongs to (they are handed over next-to-visit first), then one bit per branch below it, 0 for the child the
// reference enters first, 1 for the other -- and the leaves with hits are listed as {key, leaf entry} in whatever
// order the lanes meet them.  Leaves are never ancestors of one another, so the left-aligned keys are distinct and
// their numeric order IS the reference's visit order: a bitonic sort of the wavefront's entries (at most 512, in
// LDS) restores it.  The sorted entries are a contiguous run in HBM; the fill pass (radius_coop_replay_kernel) gives
// the row's tail to a wavefront again: 64 entries at a time, a prefix sum over their hit counts, every lane writes
// the hits of its entry -- same arithmetic as the leaf scan (bit-identical distances).
//
// What cannot be finished here (more than 512 leaves with hits, pool and spill full, the entry block exhausted, a
// hand-over without tasks) is searched again from the root by one lane: the count pass recounts such a row with
// radius_kernel<COUNT> over the redo list, the fill pass lists it for radius_kernel<FILL> (over_list) -- the same
// values at the same places the other writers of that row put them.
#pragma once

#include "ptk_kernels_lists.hpp"

namespace ptk {

constexpr uint32_t kRcMaxEntries = 512;      // leaves with hits of one query the sort holds (LDS: 8 KB)
constexpr uint32_t kRcLost = 0xFFFFFFFFu;    // RadiusHeavy::run_n: this query's entries are not here (searched again)
constexpr uint32_t kRcKeyTop = 57;           // the first path bit of a key (bits 63:58 = the task)
constexpr uint32_t kRcKeyLow = 6;            // bits 5:0 = the piece of a large leaf
constexpr uint32_t kMetaRcEntries = 28;      // word of the counters block: entries handed out of RadiusHeavy::entries
// The deepest tree the capped list pass is used on: a key has room for kRcKeyTop - kRcKeyLow + 1 branches below a task.
constexpr uint32_t kRcMaxDepth = kRcKeyTop - kRcKeyLow;

// (RadiusHeavy -- what a batch keeps of its handed-over queries from the count pass to the fill pass -- is declared in
// ptk_kernels.hpp next to RadiusCapture: the handle holds one)

// LDS of a wavefront of radius_coop_count_kernel, in 32-bit words: the pool [8][POOL], then keys and values of the
// entries [kRcMaxEntries] 64-bit words each.
constexpr uint32_t radius_coop_lds_words(uint32_t pool) { return 8u * pool + 4u * kRcMaxEntries; }

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
  return v;
}
// Inclusive prefix sum over the lanes.
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)v, d);
    if (lane >= (uint32_t)d) v += o;
  }
  return v;
}

// grid: any number of one-wavefront blocks; block b takes entries b, b + grid, ... of the hand-over list.
// counts[row] holds what the capped lane had counted; the total replaces it.  Rows that could not be finished go to
// redo_list (counted at meta[kMetaRedo]).
template <int POOL, class M = MetricL2, bool EXACT = true>
__global__ __launch_bounds__(64) void radius_coop_count_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim, float radius, float e_inv,
    uint64_t* __restrict__ counts, Handover ho, RadiusHeavy hv, uint32_t* __restrict__ redo_list,
    Task* __restrict__ spill, uint32_t spill_cap) {
  static_assert(POOL >= (int)kMaxTasks, "the pool must hold what a query starts with");
  typedef PTK_LDS uint32_t LdsU32;
  const uint4* __restrict__ nodes = t.nodes;
  const float4* __restrict__ pts = t.pts;
  // (a wavefront's run of the spill block: spill_cap tasks of 24 bytes, then their keys, 8 bytes each)
  char* const spill_base = reinterpret_cast<char*>(spill) + (uint64_t)blockIdx.x * spill_cap * 32u;
  Task* const spill_w = reinterpret_cast<Task*>(spill_base);
  unsigned long long* const spill_k = reinterpret_cast<unsigned long long*>(spill_base + (uint64_t)spill_cap * sizeof(Task));
  const uint32_t lane = threadIdx.x;
  const uint64_t below = (1ull << lane) - 1ull;
  LdsU32* pool = (LdsU32*)ptk_smem;       // [field][slot]: ref, nbd, off0, off1, off2, key low, key high, next key bit
  LdsU32* ekey = pool + 8 * POOL;         // [kRcMaxEntries][2]
  LdsU32* eval = ekey + 2 * kRcMaxEntries;  // [kRcMaxEntries][2]
  const uint32_t n_heavy = ho.meta[ho.counter] > ho.max_heavy ? ho.max_heavy : ho.meta[ho.counter];
  const float bound = f_mul(radius, e_inv);  // (already scaled by 1 / e for the approximate search, :265)

  for (uint32_t entry = blockIdx.x; entry < n_heavy; entry += gridDim.x) {  // (uniform)
    const uint32_t qi = ho.heavy_list[entry];
    const uint32_t nt = ho.ntasks[entry];
    const Task* src = ho.tasks + (uint64_t)entry * kMaxTasks;
    float qx, qy, qz;
    load_query(queries, dim, qi, qx, qy, qz);
    pad_query<M>(dim, qy, qz);
    const uint32_t own = (uint32_t)counts[qi];
    bool lost = nt == kTasksFromRoot || nt > kMaxTasks;  // (uniform; non-monotone box distances do not matter here)
    const uint32_t ntasks = lost ? 0u : (nt == kTasksRedo ? 0u : nt);
    if (nt == kTasksRedo) lost = true;  // (the capped traversal wrote no tasks for it)

    // The handed-over stack, next-to-visit on top; task i gets the key prefix i.
    uint32_t count = ntasks;  // subtrees in the pool (uniform)
    for (uint32_t i = lane; i < ntasks; i += 64u) {
      const Task tk = src[i];
      const uint32_t sl = ntasks - 1u - i;
      pool[0 * POOL + sl] = tk.ref;
      pool[1 * POOL + sl] = __float_as_uint(tk.nbd);
      pool[2 * POOL + sl] = __float_as_uint(tk.off0);
      pool[3 * POOL + sl] = __float_as_uint(tk.off1);
      pool[4 * POOL + sl] = __float_as_uint(tk.off2);
      pool[5 * POOL + sl] = 0u;
      pool[6 * POOL + sl] = i << (kRcKeyTop + 1u - 32u);
      pool[7 * POOL + sl] = kRcKeyTop | 0x80000000u;  // sign bit: the handed-over form (the parent branch is read first)
    }
    bool busy = false, fresh = false;
    uint32_t ref = 0, spill_n = 0, kbit = 0, n_ent = 0, hits = 0;
    unsigned long long key = 0ull;
    float nbd = 0.0f, off0 = 0.0f, off1 = 0.0f, off2 = 0.0f;
    // The piece of a leaf this lane is measuring: first point, points seen, hits among them, its number in the leaf.
    uint32_t l_first = 0, l_pos = 0, l_mask = 0, l_piece = 0;

    for (;;) {
      if (count == 0u && spill_n != 0u) {  // (uniform) a drained pool takes back what was parked in HBM
        const uint32_t m = spill_n < (uint32_t)(POOL / 2) ? spill_n : (uint32_t)(POOL / 2);
        for (uint32_t i = lane; i < m; i += 64u) {
          const Task tk = spill_w[spill_n - m + i];
          pool[0 * POOL + i] = tk.ref;
          pool[1 * POOL + i] = __float_as_uint(tk.nbd);
          pool[2 * POOL + i] = __float_as_uint(tk.off0);
          pool[3 * POOL + i] = __float_as_uint(tk.off1);
          pool[4 * POOL + i] = __float_as_uint(tk.off2);
          const unsigned long long k2 = spill_k[spill_n - m + i];  // (its next key bit travels in the word a hand-over uses for gmax)
          pool[5 * POOL + i] = (uint32_t)k2;
          pool[6 * POOL + i] = (uint32_t)(k2 >> 32);
          pool[7 * POOL + i] = __float_as_uint(tk.gmax);
        }
        count = m;
        spill_n -= m;
      }
      // (the ballot is also where the lanes meet after the pool was written)
      const bool want = !busy;
      const uint64_t wmask = __ballot(want);
      if (want) {
        const uint32_t rank = (uint32_t)__popcll(wmask & below);
        if (rank < count) {
          const uint32_t sl = count - 1u - rank;
          ref = pool[0 * POOL + sl];
          nbd = __uint_as_float(pool[1 * POOL + sl]);
          off0 = __uint_as_float(pool[2 * POOL + sl]);
          off1 = __uint_as_float(pool[3 * POOL + sl]);
          off2 = __uint_as_float(pool[4 * POOL + sl]);
          key = (unsigned long long)pool[5 * POOL + sl] | ((unsigned long long)pool[6 * POOL + sl] << 32);
          const uint32_t kb = pool[7 * POOL + sl];
          kbit = kb & 0x7FFFFFFFu;
          fresh = (kb >> 31) != 0u;
          busy = true;
          if (!fresh && (ref & kLeafBit) != 0u) {  // a leaf begins
            l_first = (ref & 0x7FFFFFFFu) >> t.cbits;
            l_pos = l_mask = l_piece = 0u;
          }
        }
      }
      {
        const uint32_t nw = (uint32_t)__popcll(wmask);
        count -= nw < count ? nw : count;
      }

      // One node per lane.
      bool push = false, emit = false;
      uint32_t p_ref = 0, p_kbit = 0;
      unsigned long long p_key = 0ull, e_key = 0ull, e_val = 0ull;
      float p_nbd = 0.0f, p_off0 = 0.0f, p_off1 = 0.0f, p_off2 = 0.0f;
      if (busy) {
        const bool is_leaf = !fresh && (ref & kLeafBit) != 0u;
        const uint32_t lv = ref & 0x7FFFFFFFu;
        const uint32_t begin = lv >> t.cbits;
        const uint32_t cnt = lv & t.cmask;
        const uint4* from = is_leaf ? reinterpret_cast<const uint4*>(pts + begin)
                                    : nodes + (fresh ? (ref & kRecIdxMask) : (ref & kBranchIdxMask));
        const uint4 w0 = *from;
        if (!is_leaf) {
          // A branch, or (fresh) the parent branch of a pending record whose far child is entered as traverse()
          // enters it: the same arithmetic with the side given instead of chosen.
          const uint32_t axis = fresh ? (ref >> 28) & 3u : (ref >> 29) & 3u;
          const float left_max = __uint_as_float(w0.x);
          const float right_min = __uint_as_float(w0.y);
          const float v = sel3(axis, qx, qy, qz);
          const bool near_left = f_sub(f_sub(f_add(left_max, right_min), v), v) > 0.0f;
          const bool go_left = fresh ? (ref & kRecSide) != 0u : near_left;
          const float dv = f_sub(go_left ? right_min : left_max, v);
          const float new_off = M::one(dv);
          const uint32_t far_ref = go_left ? w0.w : w0.z;
          if (fresh) {
            off0 = axis == 0 ? new_off : off0;
            off1 = axis == 1 ? new_off : off1;
            off2 = axis == 2 ? new_off : off2;
            ref = far_ref;
            fresh = false;
          } else {
            const float far_nbd = f_add(f_sub(nbd, sel3(axis, off0, off1, off2)), new_off);
            if (bound >= far_nbd) {  // the test of kd_tree_search.hpp:99 with the radius visitor's constant max()
              push = true;
              p_ref = far_ref;
              p_nbd = far_nbd;
              p_off0 = axis == 0 ? new_off : off0;
              p_off1 = axis == 1 ? new_off : off1;
              p_off2 = axis == 2 ? new_off : off2;
              p_key = key | (1ull << kbit);  // visited second
              p_kbit = kbit - 1u;
            }
            ref = go_left ? w0.z : w0.w;  // visited first: its bit stays 0
            kbit -= 1u;
          }
          if ((ref & kLeafBit) != 0u) {  // the lane's next node is a leaf
            l_first = (ref & 0x7FFFFFFFu) >> t.cbits;
            l_pos = l_mask = l_piece = 0u;
          }
        } else {
          float4 p[4];
          p[0] = make_float4(__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w));
#pragma unroll
          for (int u = 1; u < 4; ++u) p[u] = pts[begin + u];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if ((uint32_t)u < cnt) {
              PTK_KEEP4(p[u]);
              float dx = f_sub(qx, p[u].x);
              float dy = f_sub(qy, p[u].y);
              float dz = f_sub(qz, p[u].z);
              PTK_SCALAR(dx);
              PTK_SCALAR(dy);
              PTK_SCALAR(dz);
              float d = point_distance3<M>(dx, dy, dz);
              if constexpr (!EXACT) d = f_mul(d, e_inv);
              l_mask |= (bound > d ? 1u : 0u) << l_pos;  // strict (:141)
              ++l_pos;
            }
          }
          // (a piece closes after kListMaskBits = 32 points -- eight steps of four -- or at the leaf's end, as
          // RadiusListPolicy closes it)
          const bool leaf_done = cnt <= 4u;
          if (l_pos == kListMaskBits || leaf_done) {
            if (l_mask != 0u) {
              emit = true;
              e_key = key | (unsigned long long)l_piece;
              e_val = pack_list_entry((l_first << t.cbits) | l_pos, l_mask);
              hits += (uint32_t)__popcll((unsigned long long)l_mask);
            }
            l_first += l_pos;
            l_pos = l_mask = 0u;
            ++l_piece;
          }
          if (!leaf_done) {
            ref = kLeafBit | ((begin + 4u) << t.cbits) | (cnt - 4u);
          } else {
            busy = false;
          }
        }
      }

      // The leaves with hits of this step go behind the entries (all lanes take part in the ballot).
      {
        const uint64_t em = __ballot(emit);
        if (emit) {
          const uint32_t at = n_ent + (uint32_t)__popcll(em & below);
          if (at < kRcMaxEntries) {
            ekey[2u * at] = (uint32_t)e_key;
            ekey[2u * at + 1u] = (uint32_t)(e_key >> 32);
            eval[2u * at] = (uint32_t)e_val;
            eval[2u * at + 1u] = (uint32_t)(e_val >> 32);
          }
        }
        n_ent += (uint32_t)__popcll(em);
      }

      // Far children kept in this step go onto the pool.
      const uint64_t pmask = __ballot(push);
      if (push) {
        const uint32_t sl = count + (uint32_t)__popcll(pmask & below);
        if (sl < (uint32_t)POOL) {
          pool[0 * POOL + sl] = p_ref;
          pool[1 * POOL + sl] = __float_as_uint(p_nbd);
          pool[2 * POOL + sl] = __float_as_uint(p_off0);
          pool[3 * POOL + sl] = __float_as_uint(p_off1);
          pool[4 * POOL + sl] = __float_as_uint(p_off2);
          pool[5 * POOL + sl] = (uint32_t)p_key;
          pool[6 * POOL + sl] = (uint32_t)(p_key >> 32);
          pool[7 * POOL + sl] = p_kbit;
        } else if (spill_n + (sl - (uint32_t)POOL) < spill_cap) {  // no room in LDS: parked in HBM
          Task tk;
          tk.ref = p_ref;
          tk.nbd = p_nbd;
          tk.off0 = p_off0;
          tk.off1 = p_off1;
          tk.off2 = p_off2;
          tk.gmax = __uint_as_float(p_kbit);
          spill_w[spill_n + (sl - (uint32_t)POOL)] = tk;
          spill_k[spill_n + (sl - (uint32_t)POOL)] = p_key;
        }
      }
      count += (uint32_t)__popcll(pmask);
      if (count > (uint32_t)POOL) {
        spill_n += count - (uint32_t)POOL;
        count = (uint32_t)POOL;
        if (spill_n > spill_cap) {  // a subtree was lost: this row is counted again from the root
          lost = true;
          count = 0;
          spill_n = 0;
          busy = false;
        }
      }
      if (__ballot(busy) == 0ull && count == 0u && spill_n == 0u) break;
    }
    if (n_ent > kRcMaxEntries) lost = true;

    // The entries in the reference's visit order: a bitonic sort by key (n_ent padded to a power of two with keys
    // above every real one).
    uint32_t run = 0;
    if (!lost && n_ent != 0u) {
      uint32_t np = 64u;
      while (np < n_ent) np <<= 1;
      for (uint32_t i = n_ent + lane; i < np; i += 64u) {
        ekey[2u * i] = 0xFFFFFFFFu;
        ekey[2u * i + 1u] = 0xFFFFFFFFu;
      }
      __syncthreads();
      for (uint32_t k = 2u; k <= np; k <<= 1) {    // (uniform)
        for (uint32_t j = k >> 1; j != 0u; j >>= 1) {  // (uniform)
          for (uint32_t i = lane; i < np; i += 64u) {
            const uint32_t o = i ^ j;
            if (o > i) {
              const unsigned long long a = (unsigned long long)ekey[2u * i] | ((unsigned long long)ekey[2u * i + 1u] << 32);
              const unsigned long long b = (unsigned long long)ekey[2u * o] | ((unsigned long long)ekey[2u * o + 1u] << 32);
              const bool up = (i & k) == 0u;
              if (up ? a > b : a < b) {
                ekey[2u * i] = (uint32_t)b;
                ekey[2u * i + 1u] = (uint32_t)(b >> 32);
                ekey[2u * o] = (uint32_t)a;
                ekey[2u * o + 1u] = (uint32_t)(a >> 32);
                const uint32_t v0 = eval[2u * i], v1 = eval[2u * i + 1u];
                eval[2u * i] = eval[2u * o];
                eval[2u * i + 1u] = eval[2u * o + 1u];
                eval[2u * o] = v0;
                eval[2u * o + 1u] = v1;
              }
            }
          }
          __syncthreads();
        }
      }
      // A run of the entry block for them.
      if (lane == 0) run = atomicAdd(&hv.meta[kMetaRcEntries], n_ent);
      run = (uint32_t)__shfl((int)run, 0);
      if ((uint64_t)run + n_ent > (uint64_t)hv.entry_cap) {
        lost = true;
      } else {
        for (uint32_t i = lane; i < n_ent; i += 64u)
          hv.entries[run + i] = (unsigned long long)eval[2u * i] | ((unsigned long long)eval[2u * i + 1u] << 32);
      }
      __syncthreads();  // (the entries have been read before the next query writes its own)
    }
    const uint32_t found = wave_sum_u32(hits);
    if (lane == 0) {
      hv.rows[entry] = qi;
      hv.own[entry] = own;
      hv.run_at[entry] = run;
      hv.run_n[entry] = lost ? kRcLost : n_ent;
      if (lost) {
        redo_list[atomicAdd(&hv.meta[kMetaRedo], 1u)] = qi;
      } else {
        counts[qi] = (uint64_t)own + found;
      }
    }
  }
}

// The fill pass of the handed-over queries: block b takes hand-overs b, b + grid, ...; the row's first `own` hits are
// the main replay's (they are in the lane's list), the rest is written here from the sorted entries.  Rows whose
// entries were lost are listed for radius_kernel<FILL> (over_list), whatever the main replay made of their lists.
template <class M = MetricL2>
__global__ __launch_bounds__(64) void radius_coop_replay_kernel(
    DevTree t, const float* __restrict__ queries, uint32_t dim, float e_inv, RadiusHeavy hv,
    const uint64_t* __restrict__ offsets, Neighbor* __restrict__ out, uint32_t* __restrict__ over_list,
    uint32_t* __restrict__ n_over) {
  const uint32_t lane = threadIdx.x;
  const float4* __restrict__ pts = t.pts;
  const uint32_t n_heavy = hv.meta[kMetaHeavy] > hv.max_heavy ? hv.max_heavy : hv.meta[kMetaHeavy];
  for (uint32_t h = blockIdx.x; h < n_heavy; h += gridDim.x) {  // (uniform)
    const uint32_t qi = hv.rows[h];
    const uint32_t n = hv.run_n[h];
    if (n == kRcLost) {
      if (lane == 0) over_list[atomicAdd(n_over, 1u)] = qi;
      continue;
    }
    float qx, qy, qz;
    load_query(queries, dim, qi, qx, qy, qz);
    pad_query<M>(dim, qy, qz);
    const unsigned long long* __restrict__ ent = hv.entries + hv.run_at[h];
    uint64_t at = offsets[qi] + hv.own[h];
    for (uint32_t i0 = 0; i0 < n; i0 += 64u) {  // (uniform)
      const bool have = i0 + lane < n;
      const unsigned long long e = have ? ent[i0 + lane] : 0ull;
      uint32_t mask = (uint32_t)(e >> 32);
      const uint32_t first = ((uint32_t)e & 0x7FFFFFFFu) >> t.cbits;
      const uint32_t c = (uint32_t)__popcll((unsigned long long)mask);
      const uint32_t incl = wave_scan_u32(c, lane);
      uint64_t w = at + (incl - c);
      while (mask != 0u) {
        const uint32_t b = (uint32_t)__builtin_ctz(mask);
        mask &= mask - 1u;
        float4 p = pts[first + b];
        PTK_KEEP4(p);
        float dx = f_sub(qx, p.x), dy = f_sub(qy, p.y), dz = f_sub(qz, p.z);
        PTK_SCALAR(dx);
        PTK_SCALAR(dy);
        PTK_SCALAR(dz);
        Neighbor nb;
        nb.index = __float_as_int(p.w);
        nb.distance = f_mul(point_distance3<M>(dx, dy, dz), e_inv);
        out[w++] = nb;
      }
      at += (uint32_t)__shfl((int)incl, 63);
    }
  }
}

}  // namespace ptk
