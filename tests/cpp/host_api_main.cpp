// tests/cpp/host_api_main.cpp -- TEST PROGRAM for the header-only C++ API in include/pico_tree.
//
//   host_api_main host  <dir>   per-query members on the host (no libptk needed at run time)
//   host_api_main batch <dir>   batched members through the C ABI (needs a GPU)
//
// <dir> holds points.bin / queries.bin (float32 row-major, 3-D) written by the pytest
// driver (tests/test_cpp_api.py); results are written back as .bin files which the driver
// compares bit-for-bit with the oracle.  The calls mirror how the reference's own examples
// use the API (examples/kd_tree/kd_tree_search.cpp, kd_tree_creation.cpp,
// kd_tree_custom_search_visitor.cpp, kd_tree_dynamic_arrays.cpp, kd_tree_save_and_load.cpp).

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include <pico_tree/array_traits.hpp>
#include <pico_tree/kd_tree.hpp>
#include <pico_tree/map_traits.hpp>
#include <pico_tree/vector_traits.hpp>

using point3 = std::array<float, 3>;
using space3 = std::vector<point3>;
using neighbor = pico_tree::neighbor<int, float>;

static std::vector<float> read_floats(std::string const& path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::streamsize bytes = f.tellg();
  f.seekg(0);
  std::vector<float> v(static_cast<size_t>(bytes) / sizeof(float));
  f.read(reinterpret_cast<char*>(v.data()), bytes);
  return v;
}

template <typename T>
static void write_raw(std::string const& path, T const* data, size_t count) {
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<char const*>(data), static_cast<std::streamsize>(count * sizeof(T)));
}

static space3 to_space(std::vector<float> const& v) {
  space3 s(v.size() / 3);
  for (size_t i = 0; i < s.size(); ++i) s[i] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
  return s;
}

// A user-defined visitor (concept: operator()(index, distance) + max()).
struct counting_nn {
  neighbor& nn;
  int count = 0;
  explicit counting_nn(neighbor& n) : nn(n) { nn.distance = std::numeric_limits<float>::max(); }
  void operator()(int idx, float d) {
    if (max() > d) nn = {idx, d};
    ++count;
  }
  float const& max() const { return nn.distance; }
};

// parallel_partition must leave exactly what std::partition leaves (flat_tree.hpp).
static int check_parallel_partition() {
  std::uint64_t state = 12345;
  auto next = [&state] {
    state = state * 6364136223846793005ull + 1442695040888963407ull;
    return static_cast<std::uint32_t>(state >> 33);
  };
  for (int round = 0; round < 24; ++round) {
    size_t const n = 65536 + next() % 300000;
    std::uint32_t const cut = round == 0 ? 0u : round == 1 ? 1000u : next() % 1001u;  // per mille that pass
    std::vector<int> a(n);
    for (auto& v : a) v = static_cast<int>(next() % 1000000u);
    if (round == 2) std::sort(a.begin(), a.end());
    auto pred = [cut](int v) { return static_cast<std::uint32_t>(v) % 1000u < cut; };
    std::vector<int> want = a;
    auto wm = std::partition(want.begin(), want.end(), pred);
    for (unsigned threads : {2u, 5u, 32u}) {
      std::vector<int> got = a;
      int* gm = pico_tree::internal::parallel_partition(got.data(), got.data() + n, pred, threads);
      if (gm - got.data() != wm - want.begin() || got != want) {
        std::fprintf(stderr, "parallel_partition differs: round %d n %zu cut %u threads %u\n", round, n, cut, threads);
        return 50;
      }
    }
  }
  return 0;
}

static int run_host(std::string const& dir) {
  if (int rc = check_parallel_partition()) return rc;
  std::vector<float> pv = read_floats(dir + "/points.bin");
  std::vector<float> qv = read_floats(dir + "/queries.bin");
  space3 pts = to_space(pv);
  space3 qs = to_space(qv);
  size_t const nq = qs.size();
  size_t const k = 7;
  float const radius = 0.0009f;

  // (1) borrow the space, CTAD, default rule = sliding midpoint.
  pico_tree::kd_tree tree(std::ref(pts), pico_tree::max_leaf_size_t(10));
  static_assert(std::is_same_v<decltype(tree), pico_tree::kd_tree<std::reference_wrapper<space3>>>);

  std::vector<neighbor> nn(nq), knn(nq * k), aknn(nq * k);
  std::vector<int> visits(nq);
  std::vector<std::uint64_t> roff(nq + 1, 0), boff(nq + 1, 0);
  std::vector<neighbor> rflat, rsorted;
  std::vector<int> bflat;
  std::vector<neighbor> tmp;
  std::vector<int> idxs;
  for (size_t i = 0; i < nq; ++i) {
    tree.search_nn(qs[i], nn[i]);
    tree.search_knn(qs[i], knn.begin() + i * k, knn.begin() + (i + 1) * k);  // iterator form
    tree.search_knn(qs[i], k, 1.44f, tmp);                                    // approximate, vector form
    std::copy(tmp.begin(), tmp.end(), aknn.begin() + i * k);
    neighbor cn{-1, 0.0f};
    counting_nn v(cn);
    tree.search_nearest(qs[i], v);
    visits[i] = (cn.index == nn[i].index && cn.distance == nn[i].distance) ? v.count : -1;
    tree.search_radius(qs[i], radius, tmp);
    rflat.insert(rflat.end(), tmp.begin(), tmp.end());
    roff[i + 1] = rflat.size();
    tree.search_radius(qs[i], radius, tmp, true);
    rsorted.insert(rsorted.end(), tmp.begin(), tmp.end());
    point3 lo = {qs[i][0] - 0.02f, qs[i][1] - 0.02f, qs[i][2] - 0.02f};
    point3 hi = {qs[i][0] + 0.02f, qs[i][1] + 0.02f, qs[i][2] + 0.02f};
    tree.search_box(lo, hi, idxs);
    bflat.insert(bflat.end(), idxs.begin(), idxs.end());
    boff[i + 1] = bflat.size();
  }
  write_raw(dir + "/nn.bin", nn.data(), nn.size());
  write_raw(dir + "/knn.bin", knn.data(), knn.size());
  write_raw(dir + "/aknn.bin", aknn.data(), aknn.size());
  write_raw(dir + "/visits.bin", visits.data(), visits.size());
  write_raw(dir + "/radius_off.bin", roff.data(), roff.size());
  write_raw(dir + "/radius_flat.bin", rflat.data(), rflat.size());
  write_raw(dir + "/radius_sorted.bin", rsorted.data(), rsorted.size());
  write_raw(dir + "/box_off.bin", boff.data(), boff.size());
  write_raw(dir + "/box_flat.bin", bflat.data(), bflat.size());

  // (2) save in the reference's byte format, reload, same answers.
  {
    std::stringstream ss(std::ios::in | std::ios::out | std::ios::binary);
    decltype(tree)::save(tree, ss);
    std::string bytes = ss.str();
    write_raw(dir + "/save.bin", bytes.data(), bytes.size());
    auto loaded = decltype(tree)::load(std::ref(pts), ss);
    for (size_t i = 0; i < nq; i += 17) {
      neighbor a;
      loaded.search_nn(qs[i], a);
      if (a.index != nn[i].index || a.distance != nn[i].distance) return 10;
    }
    if (loaded.leaf_ranges().size() != tree.leaf_ranges().size()) return 11;
  }

  // (3) run-time dimension through space_map / point_map, owning move, raw float[3] query.
  {
    using dyn_point = pico_tree::point_map<float const, pico_tree::dynamic_extent>;
    pico_tree::space_map<dyn_point> dyn(pv.data(), pts.size(), 3);
    pico_tree::kd_tree<pico_tree::space_map<dyn_point>> dtree(dyn, pico_tree::max_leaf_size_t(10));
    for (size_t i = 0; i < nq; i += 13) {
      float raw[3] = {qs[i][0], qs[i][1], qs[i][2]};
      neighbor a{-1, 0.0f};
      dtree.search_nn(raw, a);
      if (a.index != nn[i].index || a.distance != nn[i].distance) return 20;
    }
  }

  // (4) the other build parameters and metrics: properties against brute force.
  {
    auto brute = [&](point3 const& q, auto metric) {
      float best = std::numeric_limits<float>::max();
      for (auto const& p : pts) best = std::min(best, metric(q.begin(), q.end(), p.begin()));
      return best;
    };
    pico_tree::kd_tree<space3> median(pts, pico_tree::max_leaf_size_t(4), pico_tree::bounds_from_space,
                                      pico_tree::median_max_side);
    pico_tree::kd_tree<space3> midpoint(pts, pico_tree::max_leaf_depth_t(9),
                                        pico_tree::bounds_t(point3{-1, -1, -1}, point3{2, 2, 2}),
                                        pico_tree::midpoint_max_side);
    auto l1 = pico_tree::make_kd_tree<pico_tree::metric_l1>(std::ref(pts), pico_tree::max_leaf_size_t(6));
    auto linf = pico_tree::make_kd_tree<pico_tree::metric_lpinf>(std::ref(pts), pico_tree::max_leaf_size_t(6));
    for (size_t i = 0; i < nq; i += 29) {
      neighbor a, b, c, d;
      median.search_nn(qs[i], a);
      midpoint.search_nn(qs[i], b);
      l1.search_nn(qs[i], c);
      linf.search_nn(qs[i], d);
      if (a.distance != nn[i].distance || b.distance != nn[i].distance) return 30;
      if (c.distance != brute(qs[i], pico_tree::metric_l1())) return 31;
      // L-infinity: the incremental box distance is a SUM of per-axis terms (as in the
      // reference, kd_tree_search.hpp:91-94), which over-estimates a max-norm, so the result
      // is only guaranteed to be no closer than the brute-force optimum.
      if (d.distance < brute(qs[i], pico_tree::metric_lpinf())) return 32;
    }
    // double precision instantiation
    std::vector<std::array<double, 3>> dp(pts.size());
    for (size_t i = 0; i < pts.size(); ++i) dp[i] = {pts[i][0], pts[i][1], pts[i][2]};
    pico_tree::kd_tree dtree(std::ref(dp), pico_tree::max_leaf_size_t(10));
    pico_tree::neighbor<int, double> dn;
    dtree.search_nn(dp[5], dn);
    if (dn.distance != 0.0) return 33;
  }
  {  // topological spaces (reference metric.hpp:186-257, kd_tree_search.hpp:115-229): host members
    using space1 = std::vector<std::array<float, 1>>;
    space1 p1(pts.size()), q1(qs.size());
    for (size_t i = 0; i < pts.size(); ++i) p1[i] = {pts[i][0]};
    for (size_t i = 0; i < qs.size(); ++i) q1[i] = {qs[i][0]};
    pico_tree::kd_tree<space1, pico_tree::metric_so2> so2(std::cref(p1).get(), pico_tree::max_leaf_size_t(10));
    pico_tree::kd_tree<std::reference_wrapper<space3>, pico_tree::metric_se2_squared> se2(
        std::ref(pts), pico_tree::max_leaf_size_t(10));
    size_t const nq = qs.size(), k = 7;
    std::vector<neighbor> a(nq * k), b(nq * k), c(nq * k);
    std::vector<std::uint64_t> roff(nq + 1, 0), boff(nq + 1, 0);
    std::vector<neighbor> rflat, row;
    std::vector<int> bflat, brow;
    for (size_t i = 0; i < nq; ++i) {
      so2.search_knn(q1[i], a.begin() + i * k, a.begin() + (i + 1) * k);
      se2.search_knn(qs[i], b.begin() + i * k, b.begin() + (i + 1) * k);
      se2.search_knn(qs[i], 1.44f, c.begin() + i * k, c.begin() + (i + 1) * k);
      se2.search_radius(qs[i], 0.0009f, row);
      rflat.insert(rflat.end(), row.begin(), row.end());
      roff[i + 1] = rflat.size();
      point3 lo = qs[i], hi = qs[i];
      for (int d = 0; d < 3; ++d) {
        lo[d] -= 0.02f;
        hi[d] += 0.02f;
      }
      // (the angle's interval through the seam 0 ~ 1: its min comes out above its max, box.hpp:300-376)
      if (lo[2] < 0.0f) lo[2] += 1.0f;
      if (hi[2] > 1.0f) hi[2] -= 1.0f;
      se2.search_box(lo, hi, brow);
      bflat.insert(bflat.end(), brow.begin(), brow.end());
      boff[i + 1] = bflat.size();
    }
    write_raw(dir + "/t_so2_knn.bin", a.data(), a.size());
    write_raw(dir + "/t_se2_knn.bin", b.data(), b.size());
    write_raw(dir + "/t_se2_aknn.bin", c.data(), c.size());
    write_raw(dir + "/t_se2_radius_off.bin", roff.data(), roff.size());
    write_raw(dir + "/t_se2_radius_flat.bin", rflat.data(), rflat.size());
    write_raw(dir + "/t_se2_box_off.bin", boff.data(), boff.size());
    write_raw(dir + "/t_se2_box_flat.bin", bflat.data(), bflat.size());
    {  // four bounds per branch in the tree file (kd_tree_branch_double), and back
      std::stringstream ss(std::ios::in | std::ios::out | std::ios::binary);
      decltype(se2)::save(se2, ss);
      std::string const bytes = ss.str();
      write_raw(dir + "/t_se2_save.bin", bytes.data(), bytes.size());
      auto loaded = decltype(se2)::load(std::ref(pts), ss);
      neighbor x, y;
      for (size_t i = 0; i < nq; i += 37) {
        se2.search_nn(qs[i], x);
        loaded.search_nn(qs[i], y);
        if (x.index != y.index || x.distance != y.distance) return 34;
      }
    }
  }
  std::printf("host ok\n");
  return 0;
}

#ifndef PTK_TEST_HOST_ONLY
static int run_batch(std::string const& dir) {
  std::vector<float> pv = read_floats(dir + "/points.bin");
  std::vector<float> qv = read_floats(dir + "/queries.bin");
  space3 pts = to_space(pv);
  space3 qs = to_space(qv);
  size_t const nq = qs.size();
  size_t const k = 7;
  float const radius = 0.0009f;
  pico_tree::kd_tree<space3> tree(std::move(pts), pico_tree::max_leaf_size_t(10));
  tree.prepare_device();

  std::vector<neighbor> nn(nq), knn(nq * k), aknn(nq * k);
  tree.search_nn(qs, nn.data());                  // vector<array> query space (zero copy)
  pico_tree::space_map<pico_tree::point_map<float const, 3>> qmap(qv.data(), nq);
  tree.search_knn(qmap, k, knn.data());           // raw-pointer query space
  tree.search_knn(std::cref(qs), k, 1.44f, aknn.data());
  std::vector<std::vector<neighbor>> rows;
  tree.search_radius(qs, radius, rows);
  std::vector<std::uint64_t> roff(nq + 1, 0);
  std::vector<neighbor> rflat;
  for (size_t i = 0; i < nq; ++i) {
    rflat.insert(rflat.end(), rows[i].begin(), rows[i].end());
    roff[i + 1] = rflat.size();
  }
  std::vector<std::uint64_t> soff;
  std::vector<neighbor> sflat;
  tree.search_radius(qs, radius, soff, sflat, true);
  write_raw(dir + "/b_nn.bin", nn.data(), nn.size());
  write_raw(dir + "/b_knn.bin", knn.data(), knn.size());
  write_raw(dir + "/b_aknn.bin", aknn.data(), aknn.size());
  write_raw(dir + "/b_radius_off.bin", roff.data(), roff.size());
  write_raw(dir + "/b_radius_flat.bin", rflat.data(), rflat.size());
  write_raw(dir + "/b_radius_sorted.bin", sflat.data(), sflat.size());
  {  // batched box search: a small box around every query
    space3 lo = qs, hi = qs;
    for (size_t i = 0; i < nq; ++i)
      for (int d = 0; d < 3; ++d) {
        lo[i][d] -= 0.02f;
        hi[i][d] += 0.02f;
      }
    std::vector<std::uint64_t> boff;
    std::vector<int> bflat;
    tree.search_box(lo, hi, boff, bflat);
    write_raw(dir + "/b_box_off.bin", boff.data(), boff.size());
    write_raw(dir + "/b_box_flat.bin", bflat.data(), bflat.size());
  }
  {  // the same batched members under metric_l1 / metric_lpinf (ptk_tree_set_metric)
    auto l1 = pico_tree::make_kd_tree<pico_tree::metric_l1>(std::cref(qs), pico_tree::max_leaf_size_t(10));
    auto linf = pico_tree::make_kd_tree<pico_tree::metric_lpinf>(std::cref(qs), pico_tree::max_leaf_size_t(10));
    std::vector<neighbor> a(nq * k), b(nq * k);
    l1.search_knn(qmap, k, a.data());
    linf.search_knn(qmap, k, b.data());
    write_raw(dir + "/b_l1_knn.bin", a.data(), a.size());
    write_raw(dir + "/b_linf_knn.bin", b.data(), b.size());
    std::vector<std::uint64_t> off;
    std::vector<neighbor> flat;
    l1.search_radius(qs, 0.03f, off, flat, false);
    write_raw(dir + "/b_l1_radius_off.bin", off.data(), off.size());
    write_raw(dir + "/b_l1_radius_flat.bin", flat.data(), flat.size());
  }
  {  // metric_lninf and the topological metric_se2_squared (x, y, angle), batched on the device too
    auto lninf = pico_tree::make_kd_tree<pico_tree::metric_lninf>(std::cref(qs), pico_tree::max_leaf_size_t(10));
    auto se2 = pico_tree::make_kd_tree<pico_tree::metric_se2_squared>(std::cref(qs), pico_tree::max_leaf_size_t(10));
    std::vector<neighbor> a(nq * k), b(nq * k);
    lninf.search_knn(qmap, k, a.data());
    se2.search_knn(qmap, k, b.data());
    write_raw(dir + "/b_lninf_knn.bin", a.data(), a.size());
    write_raw(dir + "/b_se2_knn.bin", b.data(), b.size());
    std::vector<std::uint64_t> off;
    std::vector<neighbor> flat;
    se2.search_radius(qs, 0.0009f, off, flat, false);
    write_raw(dir + "/b_se2_radius_off.bin", off.data(), off.size());
    write_raw(dir + "/b_se2_radius_flat.bin", flat.data(), flat.size());
  }
  {  // double precision: kd_tree over double points takes the ptk_tree64_* / ptk_search64_* entry points
    using neighbor64 = pico_tree::neighbor<int, double>;
    std::vector<std::array<double, 3>> dp(tree.space().size()), dq(nq);
    for (size_t i = 0; i < dp.size(); ++i)
      for (int d = 0; d < 3; ++d) dp[i][d] = static_cast<double>(tree.space()[i][d]) * 1.0000001;
    for (size_t i = 0; i < nq; ++i)
      for (int d = 0; d < 3; ++d) dq[i][d] = static_cast<double>(qs[i][d]) * 1.0000001;
    pico_tree::kd_tree dtree(std::cref(dp), pico_tree::max_leaf_size_t(10));
    std::vector<neighbor64> a(nq), b(nq * k);
    std::memset(static_cast<void*>(b.data()), 0, b.size() * sizeof(neighbor64));
    dtree.search_nn(dq, a.data());
    dtree.search_knn(dq, k, b.data());
    for (size_t i = 0; i < nq; ++i) {  // the batched answer is the per-query member's answer
      neighbor64 one;
      dtree.search_nn(dq[i], one);
      if (one.index != a[i].index || one.distance != a[i].distance) return 41;
      if (b[i * k].index != one.index || b[i * k].distance != one.distance) return 42;
    }
    std::vector<std::uint64_t> off, boff;
    std::vector<neighbor64> flat;
    std::vector<int> bflat;
    dtree.search_radius(dq, 0.0009, off, flat, false);
    auto lo = dq, hi = dq;
    for (size_t i = 0; i < nq; ++i)
      for (int d = 0; d < 3; ++d) {
        lo[i][d] -= 0.02;
        hi[i][d] += 0.02;
      }
    dtree.search_box(lo, hi, boff, bflat);
    std::vector<int> idx(b.size()), ridx(flat.size());
    std::vector<double> dist(b.size()), rdist(flat.size());
    for (size_t i = 0; i < b.size(); ++i) idx[i] = b[i].index, dist[i] = b[i].distance;
    for (size_t i = 0; i < flat.size(); ++i) ridx[i] = flat[i].index, rdist[i] = flat[i].distance;
    write_raw(dir + "/d_knn_idx.bin", idx.data(), idx.size());
    write_raw(dir + "/d_knn_dist.bin", dist.data(), dist.size());
    write_raw(dir + "/d_radius_off.bin", off.data(), off.size());
    write_raw(dir + "/d_radius_idx.bin", ridx.data(), ridx.size());
    write_raw(dir + "/d_radius_dist.bin", rdist.data(), rdist.size());
    write_raw(dir + "/d_box_off.bin", boff.data(), boff.size());
    write_raw(dir + "/d_box_flat.bin", bflat.data(), bflat.size());
  }
  {  // the topological metric_se2_squared over DOUBLE points, batched on the device (four bounds per branch cross the
     // boundary in the reference's own stream: ptk_tree64_create_from_topological_stream)
    using neighbor64 = pico_tree::neighbor<int, double>;
    std::vector<std::array<double, 3>> sq(nq);
    for (size_t i = 0; i < nq; ++i)
      for (int d = 0; d < 3; ++d) sq[i][d] = static_cast<double>(qs[i][d]);
    auto se2d = pico_tree::make_kd_tree<pico_tree::metric_se2_squared>(std::cref(sq), pico_tree::max_leaf_size_t(10));
    std::vector<neighbor64> b(nq * k);
    std::memset(static_cast<void*>(b.data()), 0, b.size() * sizeof(neighbor64));
    se2d.search_knn(sq, k, b.data());
    for (size_t i = 0; i < nq; i += 7) {  // the batched answer is the per-query member's answer
      std::vector<neighbor64> one;
      se2d.search_knn(sq[i], k, one);
      for (size_t j = 0; j < k; ++j)
        if (one[j].index != b[i * k + j].index || one[j].distance != b[i * k + j].distance) return 45;
    }
    std::vector<std::uint64_t> off, boff;
    std::vector<neighbor64> flat;
    std::vector<int> bflat;
    se2d.search_radius(sq, 0.0009, off, flat, false);
    auto lo = sq, hi = sq;
    for (size_t i = 0; i < nq; ++i)
      for (int d = 0; d < 3; ++d) {
        lo[i][d] -= 0.02;
        hi[i][d] += 0.02;
      }
    se2d.search_box(lo, hi, boff, bflat);
    std::vector<int> idx(b.size()), ridx(flat.size());
    std::vector<double> dist(b.size()), rdist(flat.size());
    for (size_t i = 0; i < b.size(); ++i) idx[i] = b[i].index, dist[i] = b[i].distance;
    for (size_t i = 0; i < flat.size(); ++i) ridx[i] = flat[i].index, rdist[i] = flat[i].distance;
    write_raw(dir + "/d_se2_knn_idx.bin", idx.data(), idx.size());
    write_raw(dir + "/d_se2_knn_dist.bin", dist.data(), dist.size());
    write_raw(dir + "/d_se2_radius_off.bin", off.data(), off.size());
    write_raw(dir + "/d_se2_radius_idx.bin", ridx.data(), ridx.size());
    write_raw(dir + "/d_se2_radius_dist.bin", rdist.data(), rdist.size());
    write_raw(dir + "/d_se2_box_off.bin", boff.data(), boff.size());
    write_raw(dir + "/d_se2_box_flat.bin", bflat.data(), bflat.size());
  }
  {  // a call the device search refuses (a topological tree some 3000 levels deep: coincident angles, leaf size 1) is
     // THROWN back at the caller -- the batched path has no CPU fallback -- unless the caller has asked, once, for such
     // calls to be served as a loop of the per-query members (the reference's own loop): pico_tree::allow_host_loop
    std::vector<std::array<float, 1>> ring(4000);
    for (size_t i = 0; i < ring.size(); ++i) ring[i][0] = i < 3000 ? 0.25f : static_cast<float>(i % 997) / 997.0f;
    auto so2 = pico_tree::make_kd_tree<pico_tree::metric_so2>(std::cref(ring), pico_tree::max_leaf_size_t(1));
    std::vector<std::array<float, 1>> rq(300);
    for (size_t i = 0; i < rq.size(); ++i) rq[i][0] = static_cast<float>(i) / 300.0f;
    std::vector<neighbor> got(rq.size() * 3);
    try {
      so2.search_knn(rq, 3, got.data());
      return 46;  // (refused calls must not be served silently)
    } catch (std::runtime_error const&) {
    }
    pico_tree::allow_host_loop(true);
    so2.search_knn(rq, 3, got.data());
    std::vector<std::uint64_t> off;
    std::vector<neighbor> flat;
    so2.search_radius(rq, 0.001f, off, flat, false);
    for (size_t i = 0; i < rq.size(); ++i) {
      std::vector<neighbor> one;
      so2.search_knn(rq[i], 3, one);
      for (int j = 0; j < 3; ++j)
        if (one[j].index != got[i * 3 + j].index || one[j].distance != got[i * 3 + j].distance) return 43;
      so2.search_radius(rq[i], 0.001f, one);
      if (one.size() != off[i + 1] - off[i]) return 44;
      for (size_t j = 0; j < one.size(); ++j)
        if (one[j].index != flat[off[i] + j].index || one[j].distance != flat[off[i] + j].distance) return 45;
    }
    pico_tree::allow_host_loop(false);
  }
  // wrong dimension must throw, not crash
  try {
    std::vector<std::array<float, 2>> bad(4);
    tree.search_nn(bad, nn.data());
    return 40;
  } catch (std::invalid_argument const&) {
  }
  std::printf("batch ok\n");
  return 0;
}

#else
static int run_batch(std::string const&) {
  std::fprintf(stderr, "built without the batched members\n");
  return 4;
}
#endif

int main(int argc, char** argv) {
  if (argc != 3) {
    std::fprintf(stderr, "usage: %s host|batch <dir>\n", argv[0]);
    return 2;
  }
  try {
    return std::string(argv[1]) == "host" ? run_host(argv[2]) : run_batch(argv[2]);
  } catch (std::exception const& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 3;
  }
}
