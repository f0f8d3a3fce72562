// sim_split.cpp -- CPU-only design study (not product, not a test): how the k = 1 traversal of
// BASELINE config 2 splits when a query's far-side work is cut after R leaf visits and the rest
// is searched with a FIXED bound (the best at the cut), as independent tasks.
//
//   g++ -O2 -std=c++17 -fopenmp -ffp-contract=off -Iinclude tools/sim_split.cpp -o /tmp/sim/sim_split
//   /tmp/sim/sim_split /tmp/sim/pts_L.f32 /tmp/sim/q_L.f32 [R] [R3] [sample]
//
// Prints: the distribution of leaf visits per query (exact search), how many queries are cut at
// R, how many leaves the fixed-bound search visits behind the cut against the exact search, and
// how many rounds of "run R3 leaf visits, then hand every pending far child to a lane of its
// own" it takes until every task is done.
#include <omp.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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

struct Task {
  uint32_t node;
  float nbd;
  float off[3];
};

struct Sim {
  const Tree& t;
  const float* q;
  float off[3] = {0, 0, 0};
  float best = 3.402823466e+38f;
  int best_i = 0;
  // exact run
  uint32_t leaves = 0;
  uint32_t cut = 0;  // R
  bool suspended = false;
  float cut_best = 0;
  std::vector<Task> tasks;
  Sim(const Tree& tr, const float* qq) : t(tr), q(qq) {}

  void scan(const node_t& n, bool update) {
    for (int i = n.begin; i < n.end; ++i) {
      const float* p = t.pts + 3 * (size_t)t.indices[i];
      const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
      const float d = (dx * dx + dy * dy) + dz * dz;
      if (update && best > d) {
        best = d;
        best_i = t.indices[i];
      }
    }
  }
  // The reference's recursion with a cut after `cut` leaf visits.
  void exact(uint32_t ni, float nbd) {
    const node_t& n = t.nodes[ni];
    if (n.is_leaf()) {
      scan(n, true);
      ++leaves;
      if (cut && leaves == cut) {
        suspended = true;
        cut_best = best;
      }
      return;
    }
    const uint32_t ax = n.split_dim;
    const float v = q[ax];
    uint32_t near, far;
    float new_off;
    if ((n.left_max + n.right_min - v - v) > 0) {
      near = ni + 1;
      far = n.right;
      new_off = n.right_min - v;
    } else {
      near = n.right;
      far = ni + 1;
      new_off = n.left_max - v;
    }
    new_off = new_off * new_off;
    exact(near, nbd);
    const float old = off[ax];
    const float far_nbd = nbd - old + new_off;
    if (suspended) {
      if (cut_best >= far_nbd) {
        Task k;
        k.node = far;
        k.nbd = far_nbd;
        k.off[0] = off[0], k.off[1] = off[1], k.off[2] = off[2];
        k.off[ax] = new_off;
        tasks.push_back(k);
      }
      return;
    }
    if (best >= far_nbd) {
      off[ax] = new_off;
      exact(far, far_nbd);
      off[ax] = old;
    }
  }
};

// Fixed-bound search of one task, at most `cap` leaf visits, the rest handed out as new tasks.
struct Fixed {
  const Tree& t;
  const float* q;
  float B;
  float off[3];
  uint32_t leaves = 0, cap = 0;
  bool suspended = false;
  float dmin = 3.402823466e+38f;
  std::vector<Task>* out;
  Fixed(const Tree& tr, const float* qq, float b) : t(tr), q(qq), B(b) {}
  void run(uint32_t ni, float nbd) {
    const node_t& n = t.nodes[ni];
    if (n.is_leaf()) {
      for (int i = n.begin; i < n.end; ++i) {
        const float* p = t.pts + 3 * (size_t)t.indices[i];
        const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
        const float d = (dx * dx + dy * dy) + dz * dz;
        if (d < dmin) dmin = d;
      }
      ++leaves;
      if (cap && leaves == cap) suspended = true;
      return;
    }
    const uint32_t ax = n.split_dim;
    const float v = q[ax];
    uint32_t near, far;
    float new_off;
    if ((n.left_max + n.right_min - v - v) > 0) {
      near = ni + 1;
      far = n.right;
      new_off = n.right_min - v;
    } else {
      near = n.right;
      far = ni + 1;
      new_off = n.left_max - v;
    }
    new_off = new_off * new_off;
    run(near, nbd);
    const float old = off[ax];
    const float far_nbd = nbd - old + new_off;
    if (!(B >= far_nbd)) return;
    if (suspended) {
      Task k;
      k.node = far;
      k.nbd = far_nbd;
      k.off[0] = off[0], k.off[1] = off[1], k.off[2] = off[2];
      k.off[ax] = new_off;
      out->push_back(k);
      return;
    }
    off[ax] = new_off;
    run(far, far_nbd);
    off[ax] = old;
  }
};

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  const uint32_t R = argc > 3 ? atoi(argv[3]) : 16;
  const uint32_t R3 = argc > 4 ? atoi(argv[4]) : 16;
  const size_t sample = argc > 5 ? atoll(argv[5]) : 0;
  auto load = [](const char* path, std::vector<float>& v) {
    FILE* f = fopen(path, "rb");
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
  size_t nq = qs.size() / 3;
  Tree tree;
  {
    using space_t = space_map<point_map<float const, dynamic_extent>>;
    space_t space(pts.data(), n, 3);
    internal::space_view<space_t> view(space);
    auto flat = internal::build_flat_tree<int>(view, max_leaf_size_t(10), bounds_from_space,
                                               sliding_midpoint_max_side, false, 8);
    tree.nodes.assign(flat.nodes.begin(), flat.nodes.end());
    tree.indices = std::move(flat.indices);
    tree.pts = pts.data();
  }
  const size_t step = sample && sample < nq ? nq / sample : 1;
  std::vector<size_t> sel;
  for (size_t i = 0; i < nq; i += step) sel.push_back(i);
  const size_t ns = sel.size();
  fprintf(stderr, "tree %zu nodes, %zu queries sampled (every %zu), R = %u, R3 = %u\n", tree.nodes.size(), ns, step, R, R3);

  std::vector<uint32_t> n_exact(ns), n_relaxed(ns, 0), n_iter(ns, 0), n_tasks(ns, 0), n_mism(ns, 0);
#pragma omp parallel for schedule(dynamic, 256)
  for (size_t s = 0; s < ns; ++s) {
    const float* q = qs.data() + 3 * sel[s];
    Sim full(tree, q);
    full.exact(0, 0.0f);
    n_exact[s] = full.leaves;
    if (full.leaves <= R) continue;
    Sim cut(tree, q);
    cut.cut = R;
    cut.exact(0, 0.0f);
    std::vector<Task> cur = cut.tasks, next;
    float dmin = 3.402823466e+38f;
    uint32_t iters = 0, total = 0, ntasks = 0;
    while (!cur.empty()) {
      ++iters;
      ntasks += cur.size();
      next.clear();
      for (const Task& k : cur) {
        Fixed f(tree, q, cut.cut_best);
        f.off[0] = k.off[0], f.off[1] = k.off[1], f.off[2] = k.off[2];
        f.cap = R3;
        f.out = &next;
        f.run(k.node, k.nbd);
        total += f.leaves;
        dmin = std::min(dmin, f.dmin);
      }
      cur.swap(next);
    }
    n_relaxed[s] = total;
    n_iter[s] = iters;
    n_tasks[s] = ntasks;
    const float want = full.best;
    const float got = std::min(cut.cut_best, dmin);
    n_mism[s] = want != got;
  }
  // Report.
  auto pct = [&](std::vector<uint32_t> v, double p) {
    std::sort(v.begin(), v.end());
    return v[(size_t)(p * (v.size() - 1))];
  };
  uint64_t sum_exact = 0;
  for (auto x : n_exact) sum_exact += x;
  printf("leaf visits per query (exact): mean %.2f  p50 %u p90 %u p99 %u p99.9 %u max %u\n", (double)sum_exact / ns,
         pct(n_exact, .5), pct(n_exact, .9), pct(n_exact, .99), pct(n_exact, .999), pct(n_exact, 1.0));
  for (uint32_t thr : {2u, 4u, 8u, 16u, 32u, 64u, 128u, 256u, 512u, 1024u}) {
    uint64_t c = 0, w = 0;
    for (auto x : n_exact)
      if (x > thr) ++c, w += x - thr;
    printf("  > %4u leaves: %8lu queries (%.4f %%), leaf visits beyond: %lu (%.2f %% of all)\n", thr, c, 100.0 * c / ns, w,
           100.0 * w / sum_exact);
  }
  uint64_t cutq = 0, ex_after = 0, rel = 0, tasks = 0, mism = 0;
  std::vector<uint32_t> iters;
  for (size_t s = 0; s < ns; ++s)
    if (n_exact[s] > R) {
      ++cutq;
      ex_after += n_exact[s] - R;
      rel += n_relaxed[s];
      tasks += n_tasks[s];
      mism += n_mism[s];
      iters.push_back(n_iter[s]);
    }
  {
    std::vector<double> ratio;
    std::vector<uint32_t> relv;
    for (size_t s = 0; s < ns; ++s)
      if (n_exact[s] > R) { ratio.push_back((double)n_relaxed[s] / (n_exact[s] - R)); relv.push_back(n_relaxed[s]); }
    std::sort(ratio.begin(), ratio.end());
    if (!ratio.empty())
      printf("  fixed/exact ratio per query: p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f; fixed visits p50 %u p90 %u p99 %u max %u\n",
             ratio[ratio.size() / 10], ratio[ratio.size() / 2], ratio[ratio.size() * 9 / 10], ratio[ratio.size() * 99 / 100],
             ratio.back(), pct(relv, .5), pct(relv, .9), pct(relv, .99), pct(relv, 1.0));
  }
  if (cutq) {
    printf("cut at R = %u: %lu queries; exact leaf visits behind the cut %lu, fixed-bound visits %lu (x %.3f); tasks %lu (%.1f per query)\n",
           R, cutq, ex_after, rel, (double)rel / ex_after, tasks, (double)tasks / cutq);
    printf("  rounds of R3 = %u until done: p50 %u p90 %u p99 %u max %u;  min-distance mismatches vs exact: %lu\n", R3,
           pct(iters, .5), pct(iters, .9), pct(iters, .99), pct(iters, 1.0), mism);
  }
  return 0;
}
