// sim_union.cpp -- CPU-only design study (not product, not a test; profiles/r04_notes.txt item 5): a wavefront that walks
// the tree ONCE for its 64 queries of a radius search (every node any lane's reference traversal visits, in one fixed
// order), each lane keeping its own reference state and taking part where its own traversal would.
//
//   g++ -O2 -std=c++17 -fopenmp -ffp-contract=off -Iinclude tools/sim_union.cpp -o /tmp/sim/sim_union
//   /tmp/sim/sim_union pts.f32 queries.f32 [sampled groups] [radius^2] [lanes per group] [shuffle 0/1]
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
struct Tree { std::vector<node_t> nodes; std::vector<int> indices; const float* pts; };

struct Stats {
  double u_branch = 0, u_leaf = 0, u_pts = 0, l_branch = 0, l_leaf = 0, l_pts = 0, nontrivial = 0, swapped = 0, disagree = 0, hits = 0;
  double maxframe = 0, maxdepth = 0, waves = 0, lane_leaf_hits = 0, lane_leaf_visits_with_hit = 0;
  double maxframe_max = 0, leaves_max = 0;
  double frame_hist[64] = {0};
  void operator+=(const Stats& o) {
    u_branch += o.u_branch; u_leaf += o.u_leaf; u_pts += o.u_pts; l_branch += o.l_branch; l_leaf += o.l_leaf; l_pts += o.l_pts;
    nontrivial += o.nontrivial; swapped += o.swapped; disagree += o.disagree; hits += o.hits; maxframe += o.maxframe; maxdepth += o.maxdepth; waves += o.waves;
    lane_leaf_visits_with_hit += o.lane_leaf_visits_with_hit;
    maxframe_max = std::max(maxframe_max, o.maxframe_max); leaves_max = std::max(leaves_max, o.leaves_max);
    for (int i = 0; i < 64; ++i) frame_hist[i] += o.frame_hist[i];
  }
};

struct Lane { float q[3]; float off[3]; };

struct Walker {
  const Tree& t; int G; float radius; Lane* lanes; Stats s; int curframe = 0, maxframe = 0, maxdepth = 0;
  void run(uint32_t ni, const std::vector<float>& nbd, uint64_t active, int depth) {
    maxdepth = std::max(maxdepth, depth);
    const node_t& n = t.nodes[ni];
    const int na = __builtin_popcountll(active);
    if (n.is_leaf()) {
      s.u_leaf += 1; s.u_pts += n.end - n.begin; s.l_leaf += na; s.l_pts += na * (n.end - n.begin);
      for (int l = 0; l < G; ++l) if (active >> l & 1) {
        int h = 0;
        for (int i = n.begin; i < n.end; ++i) {
          const float* p = t.pts + 3 * (size_t)t.indices[i];
          const float dx = lanes[l].q[0]-p[0], dy = lanes[l].q[1]-p[1], dz = lanes[l].q[2]-p[2];
          const float d = (dx*dx + dy*dy) + dz*dz;
          if (radius > d) ++h;
        }
        s.hits += h; if (h) s.lane_leaf_visits_with_hit += 1;
      }
      return;
    }
    s.u_branch += 1; s.l_branch += na;
    const uint32_t ax = n.split_dim;
    uint64_t go_left_m = 0, far_ok = 0;
    std::vector<float> far_nbd(G), new_off(G);
    for (int l = 0; l < G; ++l) if (active >> l & 1) {
      const float v = lanes[l].q[ax];
      const bool gl = ((n.left_max + n.right_min) - v) - v > 0;
      const float plane = gl ? n.right_min : n.left_max;
      const float dv = plane - v;
      new_off[l] = dv * dv;
      far_nbd[l] = (nbd[l] - lanes[l].off[ax]) + new_off[l];
      if (gl) go_left_m |= 1ull << l;
      if (radius >= far_nbd[l]) far_ok |= 1ull << l;
    }
    const uint64_t left_act = active & (go_left_m | far_ok), right_act = active & (~go_left_m | far_ok);
    if ((active & go_left_m) && (active & ~go_left_m)) s.disagree += 1;
    const bool both = left_act && right_act;
    if (both) { s.nontrivial += 1; ++curframe; maxframe = std::max(maxframe, curframe); }
    s.swapped += __builtin_popcountll(active & far_ok & ~go_left_m);
    // left child
    if (left_act) {
      std::vector<float> cn(nbd); std::vector<float> old(G);
      for (int l = 0; l < G; ++l) if ((left_act >> l & 1) && !(go_left_m >> l & 1)) { old[l] = lanes[l].off[ax]; lanes[l].off[ax] = new_off[l]; cn[l] = far_nbd[l]; }
      run(ni + 1, cn, left_act, depth + 1);
      for (int l = 0; l < G; ++l) if ((left_act >> l & 1) && !(go_left_m >> l & 1)) lanes[l].off[ax] = old[l];
    }
    if (right_act) {
      std::vector<float> cn(nbd); std::vector<float> old(G);
      for (int l = 0; l < G; ++l) if ((right_act >> l & 1) && (go_left_m >> l & 1)) { old[l] = lanes[l].off[ax]; lanes[l].off[ax] = new_off[l]; cn[l] = far_nbd[l]; }
      run(n.right, cn, right_act, depth + 1);
      for (int l = 0; l < G; ++l) if ((right_act >> l & 1) && (go_left_m >> l & 1)) lanes[l].off[ax] = old[l];
    }
    if (both) --curframe;
  }
};

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  const size_t sample_waves = argc > 3 ? atoll(argv[3]) : 2000;
  const float radius = argc > 4 ? (float)atof(argv[4]) : 1.0f;
  const int G = argc > 5 ? atoi(argv[5]) : 64;
  const int shuffle = argc > 6 ? atoi(argv[6]) : 0;
  auto load = [](const char* path, std::vector<float>& v) {
    FILE* f = fopen(path, "rb"); if (!f) exit(3);
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    v.resize(sz / 4); if (fread(v.data(), 4, v.size(), f) != v.size()) exit(2); fclose(f);
  };
  std::vector<float> pts, qs; load(argv[1], pts); load(argv[2], qs);
  const size_t n = pts.size() / 3, nq = qs.size() / 3;
  Tree tree; float lo[3], hi[3];
  {
    using space_t = space_map<point_map<float const, dynamic_extent>>;
    space_t space(pts.data(), n, 3);
    internal::space_view<space_t> view(space);
    auto flat = internal::build_flat_tree<int>(view, max_leaf_size_t(10), bounds_from_space, sliding_midpoint_max_side, false, 8);
    tree.nodes.assign(flat.nodes.begin(), flat.nodes.end());
    tree.indices = std::move(flat.indices);
    tree.pts = pts.data();
    for (int a = 0; a < 3; ++a) lo[a] = 3e38f, hi[a] = -3e38f;
    for (size_t i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], pts[3*i+a]), hi[a] = std::max(hi[a], pts[3*i+a]);
  }
  std::vector<uint32_t> order(nq);
  {
    int b[3] = {8, 8, 8};
    const float ex = hi[0]-lo[0], ey = hi[1]-lo[1], ez = hi[2]-lo[2];
    if (ez < 0.3f * std::min(ex, ey)) b[0] = 11, b[1] = 10, b[2] = 3;
    const int bscale = argc > 7 ? atoi(argv[7]) : 0;  // extra bits per axis
    for (int a = 0; a < 3; ++a) b[a] += bscale;
    std::vector<uint64_t> key(nq);
#pragma omp parallel for
    for (size_t i = 0; i < nq; ++i) {
      uint32_t c[3];
      for (int a = 0; a < 3; ++a) {
        float f = (qs[3*i+a] - lo[a]) / (hi[a] - lo[a]);
        f = std::min(std::max(f, 0.0f), 0.999999f);
        c[a] = (uint32_t)(f * (float)(1u << b[a]));
      }
      uint64_t kk = 0; int left[3] = {b[0], b[1], b[2]};
      for (int given = 0; given < b[0]+b[1]+b[2]; ++given) {
        int best = 0; for (int a = 1; a < 3; ++a) if (left[a] > left[best]) best = a;
        --left[best]; kk = (kk << 1) | ((c[best] >> left[best]) & 1u);
      }
      key[i] = kk;
    }
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b2) { return key[a] < key[b2]; });
    if (shuffle) { srand(1); for (size_t i = nq - 1; i > 0; --i) std::swap(order[i], order[rand() % (i + 1)]); }
  }
  const size_t waves_total = nq / G;
  const size_t nw = std::min(sample_waves, waves_total);
  const size_t wstep = waves_total / nw;
  Stats tot;
  std::vector<double> instr_u(nw), instr_l(nw);
#pragma omp parallel
  {
    Stats loc;
#pragma omp for schedule(dynamic, 16)
    for (size_t w = 0; w < nw; ++w) {
      std::vector<Lane> lanes(G);
      for (int l = 0; l < G; ++l) {
        const uint32_t qi = order[(w * wstep) * G + l];
        for (int a = 0; a < 3; ++a) lanes[l].q[a] = qs[3*(size_t)qi+a], lanes[l].off[a] = 0;
      }
      Walker wk{tree, G, radius, lanes.data()};
      std::vector<float> nbd(G, 0.0f);
      wk.run(0, nbd, G == 64 ? ~0ull : ((1ull << G) - 1), 0);
      wk.s.maxframe = wk.maxframe; wk.s.maxdepth = wk.maxdepth; wk.s.waves = 1; wk.s.maxframe_max = wk.maxframe; wk.s.leaves_max = wk.s.u_leaf;
      wk.s.frame_hist[std::min(63, wk.maxframe)] += 1;
      loc += wk.s;
    }
#pragma omp critical
    tot += loc;
  }
  const double W = tot.waves;
  printf("G = %d, radius %.3f, %zu groups sampled%s\n", G, radius, nw, shuffle ? " (SHUFFLED batch)" : "");
  printf("per group: union branches %.1f leaves %.1f points %.1f | per-lane sums: branches %.1f leaves %.1f points %.1f (x%.1f / x%.1f / x%.1f)\n",
         tot.u_branch / W, tot.u_leaf / W, tot.u_pts / W, tot.l_branch / W, tot.l_leaf / W, tot.l_pts / W,
         tot.l_branch / tot.u_branch, tot.l_leaf / tot.u_leaf, tot.l_pts / tot.u_pts);
  printf("per lane: branches %.1f leaves %.1f points %.1f hits %.1f; leaf visits with a hit %.1f\n", tot.l_branch / W / G, tot.l_leaf / W / G, tot.l_pts / W / G, tot.hits / W / G, tot.lane_leaf_visits_with_hit / W / G);
  printf("nontrivial branches (both children walked) %.1f, near sides disagree at %.1f, swapped (lane, node) pairs %.1f per group\n", tot.nontrivial / W, tot.disagree / W, tot.swapped / W);
  printf("frames nested: mean of max %.1f, max %.0f; tree depth mean of max %.1f; leaves per group max %.0f\n", tot.maxframe / W, tot.maxframe_max, tot.maxdepth / W, tot.leaves_max);
  printf("frame hist:"); for (int i = 0; i < 64; ++i) if (tot.frame_hist[i]) printf(" %d:%.0f", i, tot.frame_hist[i]); printf("\n");
  const double est = tot.u_branch / W * 30 + tot.u_leaf / W * 20 + tot.u_pts / W * 11;
  printf("est. vector instructions per group: %.0f (30 / branch, 20 / leaf, 11 / point)\n", est);
  return 0;
}
