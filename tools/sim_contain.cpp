// sim_contain.cpp -- CPU-only design study (not product, not a test; profiles/r04_notes.txt item 5): how much of a radius
// search would go away if subtrees wholly inside the ball were reported without testing their points (the running
// node box of a query against the ball, a margin of 1e-4 on the squared radius).
//
//   g++ -O2 -std=c++17 -fopenmp -ffp-contract=off -Iinclude tools/sim_contain.cpp -o /tmp/sim/sim_contain
//   /tmp/sim/sim_contain pts.f32 queries.f32 [radius^2] [largest subtree height taken whole; -1: none]
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
struct Tree { std::vector<node_t> nodes; std::vector<int> indices; const float* pts; std::vector<uint8_t> height; std::vector<uint32_t> npts, nleaves; };
static int HCAP = 8;
struct Q { const Tree& t; const float* q; float r; float off[3] = {0,0,0}; float lo[3], hi[3];
  uint64_t nodes = 0, leaves = 0, pts = 0, hits = 0, contained = 0, contained_pts = 0, contained_leaves = 0, hitleaves = 0;
  bool inside(uint32_t ni) const { float s = 0; for (int a = 0; a < 3; ++a) { float d1 = q[a] - lo[a], d2 = hi[a] - q[a]; float d = std::max(d1*d1, d2*d2); s += d; } return s < r * 0.9999f; }
  void run(uint32_t ni, float nbd) {
    const node_t& n = t.nodes[ni];
    if (t.height[ni] <= HCAP && inside(ni)) { ++contained; contained_pts += t.npts[ni]; contained_leaves += t.nleaves[ni]; hits += t.npts[ni]; return; }
    if (n.is_leaf()) { int h = 0; for (int i = n.begin; i < n.end; ++i) { const float* p = t.pts + 3*(size_t)t.indices[i]; float dx=q[0]-p[0], dy=q[1]-p[1], dz=q[2]-p[2]; float d=(dx*dx+dy*dy)+dz*dz; if (r > d) ++h; } ++leaves; pts += n.end - n.begin; hits += h; if (h) ++hitleaves; return; }
    ++nodes;
    const uint32_t ax = n.split_dim; const float v = q[ax]; uint32_t near, far; float new_off; bool gl = ((n.left_max + n.right_min) - v) - v > 0;
    if (gl) { near = ni+1; far = n.right; new_off = n.right_min - v; } else { near = n.right; far = ni+1; new_off = n.left_max - v; }
    new_off *= new_off;
    const float slo = lo[ax], shi = hi[ax];
    if (gl) hi[ax] = n.left_max; else lo[ax] = n.right_min;
    run(near, nbd);
    lo[ax] = slo; hi[ax] = shi;
    const float old = off[ax]; const float far_nbd = (nbd - old) + new_off;
    if (r >= far_nbd) { off[ax] = new_off; if (gl) lo[ax] = n.right_min; else hi[ax] = n.left_max; run(far, far_nbd); off[ax] = old; lo[ax] = slo; hi[ax] = shi; }
  } };
int main(int argc, char** argv) {
  const float radius = argc > 3 ? atof(argv[3]) : 1.0f; if (argc > 4) HCAP = atoi(argv[4]);
  auto load = [](const char* path, std::vector<float>& v) { FILE* f = fopen(path, "rb"); if (!f) exit(3); fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET); v.resize(sz/4); if (fread(v.data(), 4, v.size(), f) != v.size()) exit(2); fclose(f); };
  std::vector<float> pts, qs; load(argv[1], pts); load(argv[2], qs);
  const size_t n = pts.size()/3, nq = qs.size()/3;
  Tree tree; float blo[3], bhi[3];
  { using space_t = space_map<point_map<float const, dynamic_extent>>; space_t space(pts.data(), n, 3); internal::space_view<space_t> view(space);
    auto flat = internal::build_flat_tree<int>(view, max_leaf_size_t(10), bounds_from_space, sliding_midpoint_max_side, false, 8);
    tree.nodes.assign(flat.nodes.begin(), flat.nodes.end()); tree.indices = std::move(flat.indices); tree.pts = pts.data();
    for (int a = 0; a < 3; ++a) blo[a] = 3e38f, bhi[a] = -3e38f;
    for (size_t i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) blo[a] = std::min(blo[a], pts[3*i+a]), bhi[a] = std::max(bhi[a], pts[3*i+a]);
    const size_t nn = tree.nodes.size(); tree.height.assign(nn, 0); tree.npts.assign(nn, 0); tree.nleaves.assign(nn, 0);
    for (size_t i = nn; i-- > 0;) { const node_t& nd = tree.nodes[i]; if (nd.is_leaf()) { tree.height[i] = 0; tree.npts[i] = nd.end - nd.begin; tree.nleaves[i] = 1; } else { tree.height[i] = (uint8_t)std::min(255, 1 + std::max<int>(tree.height[i+1], tree.height[nd.right])); tree.npts[i] = tree.npts[i+1] + tree.npts[nd.right]; tree.nleaves[i] = tree.nleaves[i+1] + tree.nleaves[nd.right]; } }
  }
  uint64_t nodes = 0, leaves = 0, p = 0, hits = 0, cont = 0, cpts = 0, cleaves = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+:nodes,leaves,p,hits,cont,cpts,cleaves)
  for (size_t i = 0; i < nq; ++i) { Q q{tree, qs.data() + 3*i, radius}; for (int a = 0; a < 3; ++a) q.lo[a] = blo[a], q.hi[a] = bhi[a]; q.run(0, 0.0f); nodes += q.nodes; leaves += q.leaves; p += q.pts; hits += q.hits; cont += q.contained; cpts += q.contained_pts; cleaves += q.contained_leaves; }
  printf("height cap %d: per query: branch nodes %.1f, leaves tested %.1f, points tested %.1f, hits %.1f; contained subtrees %.2f holding %.1f points in %.1f leaves\n", HCAP, (double)nodes/nq, (double)leaves/nq, (double)p/nq, (double)hits/nq, (double)cont/nq, (double)cpts/nq, (double)cleaves/nq);
}
