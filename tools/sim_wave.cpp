// sim_wave.cpp -- CPU-only design study (not product, not a test): how the lanes of a wavefront spend the
// instructions of the general traversal kernel (traverse<> of ptk_kernels.hpp: k > 1, radius) on BASELINE config 3.
//
//   g++ -O2 -std=c++17 -fopenmp -ffp-contract=off -Iinclude tools/sim_wave.cpp -o /tmp/sim/sim_wave
//   /tmp/sim/sim_wave /tmp/sim/pts_L.f32 /tmp/sim/q_L.f32 K [sample_waves] [radius]
//
// Every query is searched as the kernel searches it (same turns: descent loop, leaf rounds of 5 points, one batch of
// 8 stack records per turn) and leaves a trace of its turns; 64 consecutive queries of the Morton-ordered batch are
// then stepped together the way a wavefront executes them, and the executions of every code region are counted with
// the lanes that were active in them.
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "pico_tree/internal/flat_tree.hpp"
#include "pico_tree/map.hpp"

using namespace pico_tree;
using node_t = internal::flat_node<int, float>;

struct Tree {
  std::vector<node_t> nodes;
  std::vector<int> indices;
  const float* pts;
};

constexpr int kLeafB = 5, kUnwind = 8;

struct Turn {
  uint8_t ndesc;     // descent steps of this turn
  uint8_t count;     // points of the leaf (0: none)
  uint16_t accept;   // bit u: point u was accepted (list insertion / radius hit)
  uint8_t popped;    // stack records looked at in the batch
  uint8_t enter;     // a far child is entered at the end of the turn
};

struct Rec {
  uint32_t meta;  // bit 31 undo, bit 30 side / nbd; else branch | axis << 28
  float val;
};

struct Query {
  std::vector<Turn> turns;
  uint32_t leaves = 0, accepted = 0, first_desc = 0;
};

// k-NN list as the reference keeps it (sorted, stable).
// SIM_DEFER=Q (environment): accepted candidates wait in a queue of Q per lane and the bound the traversal prunes with
// is the k-th distance as of the last insertion (stale, never too small); when the queue is full its oldest entry is
// inserted (the wavefront-level policy -- one pass whenever ANY lane is full -- drains a lane sooner than this: the
// model is the stalest the bound can get).
static int g_defer = getenv("SIM_DEFER") ? atoi(getenv("SIM_DEFER")) : 0;
struct KList {
  int k;
  std::vector<float> d;
  std::vector<float> pend;
  explicit KList(int kk) : k(kk), d(kk, 3.402823466e+38f) {}
  float max() const { return d[k - 1]; }
  void insert(float x) {
    if (!(d[k - 1] > x)) return;
    int j = k - 1;
    while (j > 0 && x < d[j - 1]) {
      d[j] = d[j - 1];
      --j;
    }
    d[j] = x;
  }
  bool visit(float x) {
    if (!(d[k - 1] > x)) return false;
    if (g_defer > 0) {
      pend.push_back(x);
      if ((int)pend.size() >= g_defer) {
        insert(pend.front());
        pend.erase(pend.begin());
      }
      return true;
    }
    insert(x);
    return true;
  }
};

static void trace_query(const Tree& t, const float* q, int k, float radius, Query& out) {
  KList list(k > 0 ? k : 1);
  auto bound = [&]() { return k > 0 ? list.max() : radius; };
  std::vector<Rec> st;
  uint32_t ni = 0;
  bool at_leaf_dummy = false;
  float nbd = 0, off[3] = {0, 0, 0};
  bool first = true;
  for (;;) {
    Turn turn{};
    if (!at_leaf_dummy) {
      uint32_t nd = 0;
      while (!t.nodes[ni].is_leaf()) {
        const node_t& n = t.nodes[ni];
        const uint32_t ax = n.split_dim;
        const float v = q[ax];
        const bool go_left = ((n.left_max + n.right_min) - v) - v > 0;
        const float plane = go_left ? n.right_min : n.left_max;
        const float dv = plane - v;
        const float new_off = dv * dv;
        const float far_nbd = (nbd - off[ax]) + new_off;
        if (bound() >= far_nbd) st.push_back({ni | (ax << 28) | (go_left ? 0x40000000u : 0u), far_nbd});
        ni = go_left ? ni + 1 : (uint32_t)n.right;
        ++nd;
      }
      turn.ndesc = (uint8_t)std::min<uint32_t>(nd, 255);
      if (first) out.first_desc = nd;
      first = false;
      const node_t& n = t.nodes[ni];
      turn.count = (uint8_t)(n.end - n.begin);
      for (int i = n.begin; i < n.end; ++i) {
        const float* p = t.pts + 3 * (size_t)t.indices[i];
        const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
        const float d = (dx * dx + dy * dy) + dz * dz;
        bool acc;
        if (k > 0) acc = list.visit(d); else acc = radius > d;
        if (acc) {
          turn.accept |= (uint16_t)(1u << (i - n.begin));
          ++out.accepted;
        }
      }
      ++out.leaves;
    }
    // one batch of records
    bool enter = false;
    uint32_t em = 0;
    float ev = 0;
    int used = 0;
    while (used < kUnwind && !st.empty() && !enter) {
      const Rec r = st.back();
      st.pop_back();
      ++used;
      if (r.meta & 0x80000000u) {
        if (r.meta & 0x40000000u) nbd = r.val; else off[(r.meta >> 28) & 3u] = r.val;
      } else if (bound() >= r.val) {
        enter = true;
        em = r.meta;
        ev = r.val;
      }
    }
    turn.popped = (uint8_t)used;
    turn.enter = enter;
    out.turns.push_back(turn);
    if (!enter) {
      if (st.empty()) return;
      at_leaf_dummy = true;
      continue;
    }
    at_leaf_dummy = false;
    const uint32_t idx = em & 0x0FFFFFFFu, ax = (em >> 28) & 3u;
    const bool far_right = (em & 0x40000000u) != 0;
    const node_t& n = t.nodes[idx];
    const float plane = far_right ? n.right_min : n.left_max;
    const float dv = plane - q[ax];
    const float new_off = dv * dv;
    st.push_back({0x80000000u | (ax << 28), off[ax]});
    st.push_back({0xC0000000u, nbd});
    off[ax] = new_off;
    nbd = ev;
    ni = far_right ? (uint32_t)n.right : idx + 1;
  }
}

struct Region {
  double execs = 0, lanes = 0;
  void add(int active) {
    if (active > 0) {
      execs += 1;
      lanes += active;
    }
  }
};

struct WaveStats {
  Region desc, round, visit, chain, unwind, enter;
  double turns = 0, lane_turns = 0;
  double chain_defer = 0;  // chain executions if a turn's accepted points were inserted max-per-lane at a time
  Region chain_q;          // SIM_DEFER: passes of the wavefront-level queue policy (one pass whenever any lane is full)
  void operator+=(const WaveStats& o) {
    auto a = [](Region& x, const Region& y) { x.execs += y.execs; x.lanes += y.lanes; };
    a(desc, o.desc); a(round, o.round); a(visit, o.visit); a(chain, o.chain); a(unwind, o.unwind); a(enter, o.enter);
    turns += o.turns; lane_turns += o.lane_turns; chain_defer += o.chain_defer; a(chain_q, o.chain_q);
  }
};

// One wavefront: lanes[l] = trace of its query, starting at turn start[l].
static WaveStats run_wave(const std::vector<const Query*>& lanes, const std::vector<uint32_t>& start) {
  WaveStats s;
  const int n = (int)lanes.size();
  std::vector<int> pending(n, 0);
  auto pass = [&]() {
    int a = 0;
    for (int l = 0; l < n; ++l)
      if (pending[l] > 0) --pending[l], ++a;
    s.chain_q.add(a);
  };
  for (uint32_t t = 0;; ++t) {
    int alive = 0;
    uint32_t max_desc = 0, max_cnt = 0;
    for (int l = 0; l < n; ++l) {
      const uint32_t tt = start[l] + t;
      if (tt >= lanes[l]->turns.size()) continue;
      ++alive;
      const Turn& u = lanes[l]->turns[tt];
      max_desc = std::max<uint32_t>(max_desc, u.ndesc);
      max_cnt = std::max<uint32_t>(max_cnt, u.count);
    }
    if (!alive) {
      for (;;) {  // what is left when the last lane has finished
        int mx = 0;
        for (int l = 0; l < n; ++l) mx = std::max(mx, pending[l]);
        if (mx == 0) break;
        pass();
      }
      break;
    }
    s.turns += 1;
    s.lane_turns += alive;
    for (uint32_t i = 0; i < max_desc; ++i) {
      int a = 0;
      for (int l = 0; l < n; ++l) {
        const uint32_t tt = start[l] + t;
        if (tt < lanes[l]->turns.size() && lanes[l]->turns[tt].ndesc > i) ++a;
      }
      s.desc.add(a);
    }
    int max_acc = 0;
    for (int l = 0; l < n; ++l) {
      const uint32_t tt = start[l] + t;
      if (tt < lanes[l]->turns.size()) max_acc = std::max(max_acc, __builtin_popcount(lanes[l]->turns[tt].accept));
    }
    s.chain_defer += max_acc;
    for (uint32_t r = 0; r * kLeafB < max_cnt; ++r) {
      int a = 0;
      for (int l = 0; l < n; ++l) {
        const uint32_t tt = start[l] + t;
        if (tt < lanes[l]->turns.size() && lanes[l]->turns[tt].count > r * kLeafB) ++a;
      }
      s.round.add(a);
      for (uint32_t u = 0; u < kLeafB; ++u) {
        int av = 0, ac = 0;
        const uint32_t p = r * kLeafB + u;
        for (int l = 0; l < n; ++l) {
          const uint32_t tt = start[l] + t;
          if (tt >= lanes[l]->turns.size()) continue;
          const Turn& x = lanes[l]->turns[tt];
          if (x.count > p) ++av;
          if (x.accept & (1u << p)) ++ac, ++pending[l];
        }
        s.visit.add(av);
        s.chain.add(ac);
        if (g_defer > 0) {
          for (;;) {
            bool full = false;
            for (int l = 0; l < n; ++l) full = full || pending[l] >= g_defer;
            if (!full) break;
            pass();
          }
        }
      }
    }
    int ent = 0;
    for (int l = 0; l < n; ++l) {
      const uint32_t tt = start[l] + t;
      if (tt < lanes[l]->turns.size() && lanes[l]->turns[tt].enter) ++ent;
    }
    s.unwind.add(alive);
    s.enter.add(ent);
  }
  return s;
}

static void report(const char* name, const WaveStats& s, double waves, int K) {
  // static instruction estimates per execution (vector instructions): see the header of the report
  const double c_desc = 24, c_round = 14, c_visit = 11, c_chain = 4.0 * K + 6, c_unwind = 8 * 9 + 12, c_enter = 34;
  const double inst = s.desc.execs * c_desc + s.round.execs * c_round + s.visit.execs * c_visit + s.chain.execs * c_chain +
                      s.unwind.execs * c_unwind + s.enter.execs * c_enter;
  const double lanes = s.desc.lanes * c_desc + s.round.lanes * c_round + s.visit.lanes * c_visit + s.chain.lanes * c_chain +
                       s.unwind.lanes * c_unwind + s.enter.lanes * c_enter;
  printf("== %s: %.0f waves, turns per wave %.1f (lane turns / (64 x turns) = %.3f)\n", name, waves, s.turns / waves,
         s.lane_turns / (64.0 * s.turns));
  auto row = [&](const char* n, const Region& r, double c) {
    printf("   %-8s execs/wave %8.1f  active lanes %5.1f  est. instr/wave %8.0f (%4.1f %%)\n", n, r.execs / waves,
           r.execs ? r.lanes / r.execs : 0.0, r.execs * c / waves, 100.0 * r.execs * c / inst);
  };
  row("descent", s.desc, c_desc);
  row("round", s.round, c_round);
  row("visit", s.visit, c_visit);
  row("chain", s.chain, c_chain);
  row("unwind", s.unwind, c_unwind);
  row("enter", s.enter, c_enter);
  printf("   est. vector instructions per wave %.0f, active-lane fraction %.3f; chain execs if deferred per turn: %.1f/wave\n",
         inst / waves, lanes / (64.0 * inst), s.chain_defer / waves);
  if (g_defer > 0)
    printf("   queue of %d per lane: %.1f passes/wave at %.1f lanes; est. instr/wave with the chain run per pass %.0f\n", g_defer,
           s.chain_q.execs / waves, s.chain_q.execs ? s.chain_q.lanes / s.chain_q.execs : 0.0,
           (inst - s.chain.execs * c_chain + s.chain_q.execs * (c_chain + 8) + s.visit.execs * 4 + s.round.execs * 4) / waves);
}

int main(int argc, char** argv) {
  if (argc < 4) return 1;
  const int K = atoi(argv[3]);  // 0 = radius
  const size_t sample_waves = argc > 4 ? atoll(argv[4]) : 4000;
  const float radius = argc > 5 ? (float)atof(argv[5]) : 1.0f;
  auto load = [](const char* path, std::vector<float>& v) {
    FILE* f = fopen(path, "rb");
    if (!f) exit(3);
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(sz / 4);
    if (fread(v.data(), 4, v.size(), f) != v.size()) exit(2);
    fclose(f);
  };
  std::vector<float> pts, qs;
  load(argv[1], pts);
  load(argv[2], qs);
  const size_t n = pts.size() / 3;
  const size_t nq = qs.size() / 3;
  Tree tree;
  float lo[3], hi[3];
  {
    using space_t = space_map<point_map<float const, dynamic_extent>>;
    space_t space(pts.data(), n, 3);
    internal::space_view<space_t> view(space);
    auto flat = internal::build_flat_tree<int>(view, max_leaf_size_t(10), bounds_from_space, sliding_midpoint_max_side,
                                               false, 8);
    tree.nodes.assign(flat.nodes.begin(), flat.nodes.end());
    tree.indices = std::move(flat.indices);
    tree.pts = pts.data();
    for (int a = 0; a < 3; ++a) lo[a] = 3e38f, hi[a] = -3e38f;
    for (size_t i = 0; i < n; ++i)
      for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], pts[3 * i + a]), hi[a] = std::max(hi[a], pts[3 * i + a]);
  }
  // Morton order of the batch: bits per axis in proportion to the extent (a stand-in for axis_bits()).
  std::vector<uint32_t> order(nq);
  {
    int b[3] = {8, 8, 8};
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    if (ez < 0.3f * std::min(ex, ey)) b[0] = 11, b[1] = 10, b[2] = 3;
    std::vector<uint32_t> key(nq);
#pragma omp parallel for
    for (size_t i = 0; i < nq; ++i) {
      uint32_t c[3];
      for (int a = 0; a < 3; ++a) {
        float f = (qs[3 * i + a] - lo[a]) / (hi[a] - lo[a]);
        f = std::min(std::max(f, 0.0f), 0.999999f);
        c[a] = (uint32_t)(f * (float)(1u << b[a]));
      }
      uint32_t kk = 0;
      int left[3] = {b[0], b[1], b[2]};
      for (int lvl = 0; lvl < 15; ++lvl)
        for (int a = 0; a < 3; ++a)
          if (left[a] >= 15 - lvl && left[a] > 0) {
            // axis a joins when its own bits begin
          }
      // simple interleave from the top: take the highest remaining bit of the axis with most bits left
      for (int given = 0; given < b[0] + b[1] + b[2]; ++given) {
        int best = 0;
        for (int a = 1; a < 3; ++a)
          if (left[a] > left[best]) best = a;
        --left[best];
        kk = (kk << 1) | ((c[best] >> left[best]) & 1u);
      }
      key[i] = kk;
    }
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b2) { return key[a] < key[b2]; });
  }
  const size_t waves_total = nq / 64;
  const size_t nw = std::min(sample_waves, waves_total);
  const size_t wstep = waves_total / nw;
  std::vector<Query> qt(nw * 64);
#pragma omp parallel for schedule(dynamic, 64)
  for (size_t i = 0; i < nw * 64; ++i) {
    const size_t w = i / 64, l = i % 64;
    const uint32_t qi = order[(w * wstep) * 64 + l];
    trace_query(tree, qs.data() + 3 * (size_t)qi, K, radius, qt[i]);
  }
  double leaves = 0, acc = 0, turns = 0, fd = 0;
  for (auto& q : qt) leaves += q.leaves, acc += q.accepted, turns += q.turns.size(), fd += q.first_desc;
  printf("K = %d%s: %zu queries sampled; per query: leaves %.2f, accepted points %.2f, turns %.2f, first descent %.1f\n", K,
         K ? "" : " (radius)", qt.size(), leaves / qt.size(), acc / qt.size(), turns / qt.size(), fd / qt.size());


  {
    std::vector<uint32_t> acc_n;
    for (auto& q : qt) acc_n.push_back(q.accepted);
    std::sort(acc_n.begin(), acc_n.end());
    auto pc = [&](double p) { return acc_n[(size_t)(p * (acc_n.size() - 1))]; };
    printf("accepted per query: p50 %u p90 %u p99 %u p99.9 %u p99.99 %u max %u\n", pc(.5), pc(.9), pc(.99), pc(.999), pc(.9999), pc(1.0));
  }
  // (1) the kernel as it is: 64 consecutive queries per wave
  WaveStats cur;
  for (size_t w = 0; w < nw; ++w) {
    std::vector<const Query*> lanes(64);
    std::vector<uint32_t> start(64, 0);
    for (int l = 0; l < 64; ++l) lanes[l] = &qt[w * 64 + l];
    cur += run_wave(lanes, start);
  }
  report("as launched (Morton order)", cur, (double)nw, K ? K : 0);

  // (2) upper bound of any regrouping: queries sorted by their number of turns before they are cut into waves
  {
    std::vector<uint32_t> idx(qt.size());
    std::iota(idx.begin(), idx.end(), 0u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b2) { return qt[a].turns.size() > qt[b2].turns.size(); });
    WaveStats s;
    for (size_t w = 0; w < nw; ++w) {
      std::vector<const Query*> lanes(64);
      std::vector<uint32_t> start(64, 0);
      for (int l = 0; l < 64; ++l) lanes[l] = &qt[idx[w * 64 + l]];
      s += run_wave(lanes, start);
    }
    report("sorted by total turns (oracle knowledge)", s, (double)nw, K ? K : 0);
  }

  // (3) two phases: the first P turns in Morton order, the rest regrouped by remaining turns (oracle knowledge) or by
  //     the number of stack records that pass the bound at the split (what a kernel can know)
  for (uint32_t P : {1u, 2u, 3u, 4u, 6u}) {
    WaveStats a, b;
    for (size_t w = 0; w < nw; ++w) {
      std::vector<Query> head(64);
      std::vector<const Query*> lanes(64);
      std::vector<uint32_t> start(64, 0);
      for (int l = 0; l < 64; ++l) {
        const Query& q = qt[w * 64 + l];
        head[l].turns.assign(q.turns.begin(), q.turns.begin() + std::min<size_t>(P, q.turns.size()));
        lanes[l] = &head[l];
      }
      a += run_wave(lanes, start);
    }
    std::vector<uint32_t> idx;
    for (uint32_t i = 0; i < qt.size(); ++i)
      if (qt[i].turns.size() > P) idx.push_back(i);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return qt[x].turns.size() > qt[y].turns.size(); });
    const size_t w2 = (idx.size() + 63) / 64;
    for (size_t w = 0; w < w2; ++w) {
      std::vector<const Query*> lanes;
      std::vector<uint32_t> start;
      for (size_t l = w * 64; l < std::min(idx.size(), w * 64 + 64); ++l) lanes.push_back(&qt[idx[l]]), start.push_back(P);
      b += run_wave(lanes, start);
    }
    char name[128];
    snprintf(name, sizeof name, "split after %u turns: phase 1", P);
    report(name, a, (double)nw, K ? K : 0);
    snprintf(name, sizeof name, "split after %u turns: phase 2, %zu continuations, sorted by remaining turns", P, idx.size());
    report(name, b, (double)nw, K ? K : 0);
  }

  // (4) node-step traversal: every lane takes ONE branch per iteration; lanes that have reached a leaf wait until
  //     at least T lanes wait (or nobody descends), then all of them scan one round; a finished leaf pops the next
  //     far child (full-state entries: no undo records, no batch of eight).
  {
    struct Ev { uint8_t leaf, count; uint16_t accept; uint8_t rejected; };
    std::vector<std::vector<Ev>> ev(qt.size());
    for (size_t i = 0; i < qt.size(); ++i) {
      const Query& q = qt[i];
      uint32_t rejected = 0;
      for (size_t t = 0; t < q.turns.size(); ++t) {
        const Turn& u = q.turns[t];
        for (uint32_t d = 0; d < u.ndesc; ++d) ev[i].push_back({0, 0, 0, 0});
        if (u.count || u.ndesc || t == 0) ev[i].push_back({1, u.count, u.accept, 0});
        // pending records looked at and rejected before the next entry: popped minus undo records is unknown here;
        // a turn without an entry stands for up to 8 rejected records
        if (!u.enter) ev[i].back().rejected = (uint8_t)std::min<uint32_t>(255, ev[i].back().rejected + u.popped);
      }
    }
    for (int T : {64, 48, 32, 16, 1}) {
      Region rb, rl, rv, rc, rp;
      double iters = 0;
      for (size_t w = 0; w < nw; ++w) {
        uint32_t pos[64] = {0}, round[64] = {0};
        for (;;) {
          int nb = 0, nl = 0, alive = 0;
          for (int l = 0; l < 64; ++l) {
            const auto& e = ev[w * 64 + l];
            if (pos[l] >= e.size()) continue;
            ++alive;
            if (e[pos[l]].leaf) ++nl; else ++nb;
          }
          if (!alive) break;
          iters += 1;
          const bool fire = nl >= T || nb == 0;
          if (nb) {
            rb.add(nb);
            for (int l = 0; l < 64; ++l) {
              const auto& e = ev[w * 64 + l];
              if (pos[l] < e.size() && !e[pos[l]].leaf) ++pos[l];
              else if (pos[l] < e.size() && !fire) { /* waits */ }
            }
          }
          if (fire && nl) {
            // lanes that were at a leaf BEFORE this iteration's branch step
            int act = 0, fin = 0;
            int av[kLeafB] = {0}, ac[kLeafB] = {0};
            for (int l = 0; l < 64; ++l) {
              const auto& e = ev[w * 64 + l];
              if (pos[l] >= e.size() || !e[pos[l]].leaf) continue;
              // (a lane that arrived at its leaf in this very iteration also joins: it is at a leaf now)
              const Ev& x = e[pos[l]];
              ++act;
              for (int u = 0; u < kLeafB; ++u) {
                const uint32_t p = round[l] * kLeafB + u;
                if (x.count > p) ++av[u];
                if (x.accept & (1u << p)) ++ac[u];
              }
              ++round[l];
              if (round[l] * kLeafB >= x.count) {
                ++fin;
                round[l] = 0;
                ++pos[l];
              }
            }
            rl.add(act);
            for (int u = 0; u < kLeafB; ++u) rv.add(av[u]), rc.add(ac[u]);
            rp.add(fin);
          }
        }
      }
      const double cB = 36, cL = 11, cV = 10, cC_now = 4.8 * K + 1, cC_log = K + 4, cP = 16, cCap = 108;
      auto tot = [&](double cc, double& lanes_out) {
        double inst = rb.execs * cB + rl.execs * cL + rv.execs * cV + rp.execs * cP;
        double lanes = rb.lanes * cB + rl.lanes * cL + rv.lanes * cV + rp.lanes * cP;
        if (K) inst += rc.execs * cc, lanes += rc.lanes * cc; else inst += rl.execs * cCap, lanes += rl.lanes * cCap;
        lanes_out = lanes / (64.0 * inst);
        return inst / nw;
      };
      double u1, u2;
      const double a = tot(cC_now, u1), b = tot(cC_log, u2);
      printf("== node-step, T = %2d: iterations/wave %.1f; branch execs %.1f (%.1f lanes), leaf rounds %.1f (%.1f lanes), chains %.1f (%.1f lanes), pops %.1f\n"
             "   est. instr/wave: chain as now %.0f (active %.3f), chain with deferred indices %.0f (active %.3f)\n",
             T, iters / nw, rb.execs / nw, rb.lanes / rb.execs, rl.execs / nw, rl.lanes / rl.execs, rc.execs / nw,
             rc.execs ? rc.lanes / rc.execs : 0.0, rp.execs / nw, a, u1, b, u2);
    }
  }
  return 0;
}
