// sim_coop.cpp -- CPU-only design study (not product, not a test; profiles/r05_notes.txt): ONE walk per wavefront that
// keeps every lane's own visit ORDER.  At a branch the active lanes split into those whose nearer child is the left one
// (AL) and those whose nearer child is the right one (AR).  The wavefront visits
//     left  with AL                              (their near child)
//     right with AR + the lanes of AL that still want their far child (tested now, after their near side)
//     left  with the lanes of AR that still want their far child     (tested now)
// or the mirror image, whichever has no third visit (or the smaller one): every lane sees the nodes it would see alone,
// in the order it would see them, with its bound tested when the reference tests it -- exact for every visitor.
// The price is that a subtree may be walked twice, by disjoint sets of lanes.  This counts what that costs on BASELINE
// config 3 (radius search r^2 = 1, and k-NN with a k-list per lane).
//
//   g++ -O2 -std=c++17 -fopenmp -ffp-contract=off -Iinclude tools/sim_coop.cpp -o /tmp/sim/sim_coop
//   /tmp/sim/sim_coop pts.f32 queries.f32 K(0 = radius) [sampled groups] [radius^2] [lanes per group]
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
  double branch = 0, branch_lanes = 0, leaf = 0, leaf_lanes = 0, pts = 0, pts_lanes = 0, third = 0, frames = 0, maxframe = 0,
         waves = 0, hits = 0, chain = 0, chain_lanes = 0, leaf_with_hit = 0, leaf_with_hit_lanes = 0, maxframe_max = 0;
  double hist[40] = {0};
  void operator+=(const Stats& o) {
    branch += o.branch; branch_lanes += o.branch_lanes; leaf += o.leaf; leaf_lanes += o.leaf_lanes; pts += o.pts; pts_lanes += o.pts_lanes;
    third += o.third; frames += o.frames; maxframe += o.maxframe; waves += o.waves; hits += o.hits; chain += o.chain; chain_lanes += o.chain_lanes;
    leaf_with_hit += o.leaf_with_hit; leaf_with_hit_lanes += o.leaf_with_hit_lanes; maxframe_max = std::max(maxframe_max, o.maxframe_max);
    for (int i = 0; i < 40; ++i) hist[i] += o.hist[i];
  }
};

struct KList {
  int k; std::vector<float> d;
  explicit KList(int kk) : k(kk), d(kk > 0 ? kk : 1, 3.402823466e+38f) {}
  float max() const { return d[k - 1]; }
  bool visit(float x) {
    if (!(d[k - 1] > x)) return false;
    int j = k - 1;
    while (j > 0 && x < d[j - 1]) { d[j] = d[j - 1]; --j; }
    d[j] = x; return true;
  }
};

struct Lane { float q[3]; float off[3]; float nbd; KList* list; };

struct Walker {
  const Tree& t; int G; int K; float radius; Lane* lanes; Stats s; int cur = 0, maxframe = 0;
  float bound(int l) const { return K > 0 ? lanes[l].list->max() : radius; }
  void run(uint32_t ni, uint64_t active) {
    const node_t& n = t.nodes[ni];
    const int na = __builtin_popcountll(active);
    if (n.is_leaf()) {
      s.leaf += 1; s.leaf_lanes += na;
      uint64_t anyhit_lanes = 0;
      for (int i = n.begin; i < n.end; ++i) {
        const float* p = t.pts + 3 * (size_t)t.indices[i];
        int acc = 0;
        for (int l = 0; l < G; ++l) if (active >> l & 1) {
          const float dx = lanes[l].q[0]-p[0], dy = lanes[l].q[1]-p[1], dz = lanes[l].q[2]-p[2];
          const float d = (dx*dx + dy*dy) + dz*dz;
          bool a;
          if (K > 0) a = lanes[l].list->visit(d); else a = radius > d;
          if (a) { ++acc; anyhit_lanes |= 1ull << l; }
        }
        s.pts += 1; s.pts_lanes += na; s.hits += acc;
        if (acc) { s.chain += 1; s.chain_lanes += acc; }
      }
      if (anyhit_lanes) { s.leaf_with_hit += 1; s.leaf_with_hit_lanes += __builtin_popcountll(anyhit_lanes); }
      return;
    }
    s.branch += 1; s.branch_lanes += na;
    const uint32_t ax = n.split_dim;
    uint64_t gl = 0;
    std::vector<float> far_nbd(G), new_off(G);
    for (int l = 0; l < G; ++l) if (active >> l & 1) {
      const float v = lanes[l].q[ax];
      const bool g = ((n.left_max + n.right_min) - v) - v > 0;
      const float plane = g ? n.right_min : n.left_max;
      const float dv = plane - v;
      new_off[l] = dv * dv;
      far_nbd[l] = (lanes[l].nbd - lanes[l].off[ax]) + new_off[l];
      if (g) gl |= 1ull << l;
    }
    const uint64_t AL = active & gl, AR = active & ~gl;
    auto far_ok = [&](uint64_t m) { uint64_t r = 0; for (int l = 0; l < G; ++l) if ((m >> l & 1) && bound(l) >= far_nbd[l]) r |= 1ull << l; return r; };
    // visit a child with `near` lanes (state unchanged) and `far` lanes (state replaced for the visit)
    auto visit = [&](bool left, uint64_t near, uint64_t far) {
      if (!(near | far)) return;
      std::vector<float> old_off(G), old_nbd(G);
      for (int l = 0; l < G; ++l) if (far >> l & 1) { old_off[l] = lanes[l].off[ax]; old_nbd[l] = lanes[l].nbd; lanes[l].off[ax] = new_off[l]; lanes[l].nbd = far_nbd[l]; }
      run(left ? ni + 1 : (uint32_t)n.right, near | far);
      for (int l = 0; l < G; ++l) if (far >> l & 1) { lanes[l].off[ax] = old_off[l]; lanes[l].nbd = old_nbd[l]; }
    };
    // Which side first?  With a constant bound (radius) both far sets are known now; with a k-list only an estimate is.
    const uint64_t fL = far_ok(AL), fR = far_ok(AR);  // (estimate for the choice; re-tested at the right time below)
    bool left_first;
    if (!AR) left_first = true; else if (!AL) left_first = false;
    else if (!fR) left_first = true;            // left(AL), right(AR + far of AL): no third visit
    else if (!fL) left_first = false;           // right(AR), left(AL + far of AR): no third visit
    else left_first = __builtin_popcountll(fR) <= __builtin_popcountll(fL);  // the smaller third visit
    const bool framed = (AL && AR) || fL || fR;
    if (framed) { ++cur; maxframe = std::max(maxframe, cur); s.frames += 1; }
    if (left_first) {
      visit(true, AL, 0);
      visit(false, AR, far_ok(AL));
      const uint64_t third = far_ok(AR);
      if (third) { if (AL) s.third += 1; visit(true, 0, third); }
    } else {
      visit(false, AR, 0);
      visit(true, AL, far_ok(AR));
      const uint64_t third = far_ok(AL);
      if (third) { if (AR) s.third += 1; visit(false, 0, third); }
    }
    if (framed) --cur;
  }
};

int main(int argc, char** argv) {
  if (argc < 4) return 1;
  const int K = atoi(argv[3]);
  const size_t sample_waves = argc > 4 ? atoll(argv[4]) : 2000;
  const float radius = argc > 5 ? (float)atof(argv[5]) : 1.0f;
  const int G = argc > 6 ? atoi(argv[6]) : 64;
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
  }
  const size_t waves_total = nq / G;
  const size_t nw = std::min(sample_waves, waves_total);
  const size_t wstep = waves_total / nw;
  Stats tot;
#pragma omp parallel
  {
    Stats loc;
#pragma omp for schedule(dynamic, 16)
    for (size_t w = 0; w < nw; ++w) {
      std::vector<Lane> lanes(G);
      std::vector<KList> lists(G, KList(K));
      for (int l = 0; l < G; ++l) {
        const uint32_t qi = order[(w * wstep) * G + l];
        for (int a = 0; a < 3; ++a) lanes[l].q[a] = qs[3*(size_t)qi+a], lanes[l].off[a] = 0;
        lanes[l].nbd = 0; lanes[l].list = &lists[l];
      }
      Walker wk{tree, G, K, radius, lanes.data()};
      wk.run(0, G == 64 ? ~0ull : ((1ull << G) - 1));
      wk.s.maxframe = wk.maxframe; wk.s.maxframe_max = wk.maxframe; wk.s.waves = 1; wk.s.hist[std::min(39, wk.maxframe)] += 1;
      loc += wk.s;
    }
#pragma omp critical
    tot += loc;
  }
  const double W = tot.waves;
  printf("G = %d, %s, %zu groups sampled\n", G, K ? "k-NN" : "radius", nw);
  printf("per group: branch visits %.1f (%.1f lanes), leaf visits %.1f (%.1f lanes), point visits %.1f (%.1f lanes), third visits %.1f, frames %.1f\n",
         tot.branch / W, tot.branch_lanes / tot.branch, tot.leaf / W, tot.leaf_lanes / tot.leaf, tot.pts / W, tot.pts_lanes / tot.pts, tot.third / W, tot.frames / W);
  printf("           point visits with an accepting lane %.1f (%.1f lanes), leaf visits with a hit %.1f (%.1f lanes), accepted per lane %.1f\n",
         tot.chain / W, tot.chain ? tot.chain_lanes / tot.chain : 0.0, tot.leaf_with_hit / W, tot.leaf_with_hit ? tot.leaf_with_hit_lanes / tot.leaf_with_hit : 0.0, tot.hits / W / G);
  printf("frames nested: mean of max %.1f, max %.0f; hist:", tot.maxframe / W, tot.maxframe_max);
  for (int i = 0; i < 40; ++i) if (tot.hist[i]) printf(" %d:%.0f", i, tot.hist[i]);
  printf("\n");
  const double chain_c = K ? 4.8 * K + 1 : 0;
  const double est = tot.branch / W * 30 + tot.leaf / W * 14 + tot.pts / W * 11 + (K ? tot.chain / W * chain_c : tot.leaf_with_hit / W * 30);
  printf("est. vector instructions per group: %.0f (30 / branch, 14 / leaf, 11 / point, %s)\n", est, K ? "chain per accepting point visit" : "30 per leaf with a hit for the list");
  return 0;
}
