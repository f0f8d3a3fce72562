// oracle/ptk_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the pico_tree hot path named by BASELINE.json.north_star:
// build a kd-tree with the sliding-midpoint rule and a max-leaf-size stop, then
// answer nn / knn / radius (exact and approximate) and box queries over it with
// metric_l2_squared on float32 points and int indices.  Every function cites the
// reference lines (relative to /root/reference) whose behaviour it restates.
//
// PARITY PIN: this file is checked bit-for-bit (indices AND distance bits, and
// the kd_tree::save byte stream) against the actual reference compiled from its
// own headers (oracle/ref_driver.cpp -> oracle/_ref/libptk_ref.so) by
// tests/test_oracle_vs_reference.py in the authoring container, and against the
// committed golden vectors in tests/golden/ (generated from the compiled
// reference by tests/golden/make_golden.py) everywhere else.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// the library built from this file.  The product (include/, pico_tree_amd/)
// never includes, links or calls it.
//
// Canonical flags: -O3 -ffp-contract=off (an FMA-contracted build changes
// distance bits).

#include <omp.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

// The scalar type of the points (the reference's kd_tree is generic over it: point_traits<>::
// scalar_type).  liboracle.so is the float build; liboracle64.so is this file compiled with
// -DPTKOR_DOUBLE (no forest section, no 16-byte flat export).
#ifdef PTKOR_DOUBLE
typedef double scalar_t;
#else
typedef float scalar_t;
#endif

namespace {

struct neighbor_t {  // core.hpp:24-46
  int index;
  scalar_t distance;
};
static_assert(sizeof(neighbor_t) == (sizeof(scalar_t) == 4 ? 8 : 16), "layout");

// kd_tree_node.hpp:7-27,31-50: two child pointers + union{leaf, branch}.
struct node_t {
  node_t* left = nullptr;
  node_t* right = nullptr;
  union {
    struct {
      int begin_idx;
      int end_idx;
    } leaf;
    struct {
      int split_dim;
      scalar_t left_max;
      scalar_t right_min;
      // The outer bounds of kd_tree_node_topological (kd_tree_node.hpp:56-67,106-113); only the
      // kd_forest's priority search reads them.
      scalar_t left_min;
      scalar_t right_max;
    } branch;
  } data;
  std::uint32_t id = 0;  // position in the DFS pre-order stream (forest queue tie-break)
  bool is_leaf() const { return left == nullptr && right == nullptr; }
};

// box.hpp:161-193 with run-time size: min[dim] then max[dim].
struct box_t {
  std::vector<scalar_t> c;
  size_t dim;
  explicit box_t(size_t d) : c(2 * d), dim(d) {}
  scalar_t& mn(size_t i) { return c[i]; }
  scalar_t& mx(size_t i) { return c[dim + i]; }
  scalar_t mn(size_t i) const { return c[i]; }
  scalar_t mx(size_t i) const { return c[dim + i]; }

  void fill_inverse_max() {  // box.hpp:54-59
    for (size_t i = 0; i < dim; ++i) {
      mn(i) = std::numeric_limits<scalar_t>::max();
      mx(i) = std::numeric_limits<scalar_t>::lowest();
    }
  }
  void max_side(size_t& idx, scalar_t& val) const {  // box.hpp:71-82
    val = std::numeric_limits<scalar_t>::lowest();
    for (size_t i = 0; i < dim; ++i) {
      scalar_t const delta = mx(i) - mn(i);
      if (delta > val) {
        idx = i;
        val = delta;
      }
    }
  }
  void fit(scalar_t const* x) {  // box.hpp:86-95
    for (size_t i = 0; i < dim; ++i) {
      if (x[i] < mn(i)) mn(i) = x[i];
      if (x[i] > mx(i)) mx(i) = x[i];
    }
  }
  void fit(box_t const& o) {  // box.hpp:99-110
    for (size_t i = 0; i < dim; ++i) {
      if (o.mn(i) < mn(i)) mn(i) = o.mn(i);
      if (o.mx(i) > mx(i)) mx(i) = o.mx(i);
    }
  }
  bool contains(scalar_t const* x) const {  // box.hpp:31-40
    for (size_t i = 0; i < dim; ++i) {
      if (mn(i) > x[i] || mx(i) < x[i]) return false;
    }
    return true;
  }
  bool contains(box_t const& o) const {  // box.hpp:44-47
    return contains(o.c.data()) && contains(o.c.data() + dim);
  }
};

struct visit_counters {
  std::uint32_t n_branch = 0;
  std::uint32_t n_leaf = 0;
  std::uint32_t n_pts = 0;
  std::uint32_t n_first = 0;  // branches passed before the first leaf (depth of the home leaf)
  // Far children of the first descent whose box distance does not exceed the best distance
  // found in the home leaf: the candidates a two-phase search hands from phase 1 to phase 2.
  std::uint32_t n_cand = 0;
  std::vector<scalar_t> first_far;  // scratch: far box distances along the first descent
};

struct tree_t {
  size_t dim = 0;
  size_t n = 0;
  size_t max_leaf = 0;
  int metric = 0;                  // 0 metric_l2_squared, 1 metric_l1, 2 metric_lpinf (searches only)
  std::vector<scalar_t> pts;          // n x dim row-major, original order
  std::vector<int> indices;        // kd_tree_data.hpp:67
  box_t root_box{1};               // kd_tree_data.hpp:69
  std::vector<std::unique_ptr<node_t[]>> chunks;  // stands in for memory.hpp's pool
  size_t chunk_used = 256;
  node_t* root = nullptr;

  scalar_t const* point(int idx) const {  // space_wrapper.hpp:30-32
    return pts.data() + static_cast<size_t>(idx) * dim;
  }

  node_t* allocate() {
    if (chunk_used == 256) {
      chunks.emplace_back(new node_t[256]);
      chunk_used = 0;
    }
    return &chunks.back()[chunk_used++];
  }

  // splitter_sliding_midpoint_max_side::operator(), kd_tree_builder.hpp:229-276.
  void split_sliding_midpoint(int* begin, int* end, box_t const& box,
                              int*& split, size_t& split_dim,
                              scalar_t& split_val) const {
    scalar_t max_delta;
    box.max_side(split_dim, max_delta);
    split_val = max_delta / 2.0f + box.mn(split_dim);  // :240 (divide, then add)

    size_t const sd = split_dim;
    scalar_t const sv = split_val;
    split = std::partition(begin, end, [this, sd, sv](int const i) -> bool {
      return point(i)[sd] < sv;  // :243-247
    });

    auto const by_coord = [this, sd](int const a, int const b) -> bool {
      return point(a)[sd] < point(b)[sd];
    };
    if (split == end) {  // :255-264: everything went left, slide one point right
      --split;
      std::nth_element(begin, split, end, by_coord);
      split_val = point(*split)[sd];
    } else if (split == begin) {  // :265-275: everything went right
      ++split;
      std::nth_element(begin, split, end, by_coord);
      split_val = point(*split)[sd];
    }
  }

  // build_kd_tree_impl::create_node, kd_tree_builder.hpp:352-396, with the
  // max_leaf_size_t stop (:414-415).
  node_t* create_node(int* begin, int* end, box_t& box) {
    node_t* node = allocate();
    if (static_cast<size_t>(end - begin) <= max_leaf) {
      node->left = nullptr;  // kd_tree_node.hpp:14-19
      node->right = nullptr;
      node->data.leaf.begin_idx = static_cast<int>(begin - indices.data());
      node->data.leaf.end_idx = static_cast<int>(end - indices.data());
      box.fill_inverse_max();  // :399-407
      for (int* it = begin; it < end; ++it) box.fit(point(*it));
    } else {
      int* split;
      size_t split_dim;
      scalar_t split_val;
      split_sliding_midpoint(begin, end, box, split, split_dim, split_val);

      box_t right = box;  // :379-383
      box.mx(split_dim) = split_val;
      right.mn(split_dim) = split_val;

      node->left = create_node(begin, split, box);    // :385
      node->right = create_node(split, end, right);   // :386

      node->data.branch.split_dim = static_cast<int>(split_dim);  // node.hpp:86-92
      node->data.branch.left_max = box.mx(split_dim);
      node->data.branch.right_min = right.mn(split_dim);
      node->data.branch.left_min = box.mn(split_dim);    // node.hpp:109-112
      node->data.branch.right_max = right.mx(split_dim);

      box.fit(right);  // :392
    }
    return node;
  }

  // build_kd_tree::operator(), kd_tree_builder.hpp:471-494 with bounds_from_space.
  void build() {
    indices.resize(n);
    for (size_t i = 0; i < n; ++i) indices[i] = static_cast<int>(i);  // :485-486
    root_box = box_t(dim);
    root_box.fill_inverse_max();  // space_wrapper.hpp:34-40
    for (size_t i = 0; i < n; ++i) root_box.fit(point(static_cast<int>(i)));
    box_t work = root_box;  // :345-348 copies the root box
    root = create_node(indices.data(), indices.data() + n, work);
  }
};

// metric_l2_squared, range form: metric.hpp:107-117 -> internal::sum :36-51 ->
// squared_r1_distance distance.hpp:38-41.  d starts at 0 and accumulates left to
// right; every operation rounds to float on its own (no contraction).
inline scalar_t l2sq(scalar_t const* a, scalar_t const* b, size_t dim) {
  scalar_t d = 0.0f;
  for (size_t i = 0; i < dim; ++i) {
    scalar_t const t = a[i] - b[i];
    d += t * t;
  }
  return d;
}

// metric_l1 (metric.hpp:78-99: sum of r1_distance = |x - y|, distance.hpp:23-27) and
// metric_lpinf (metric.hpp:126-152: d = std::max(d, |x - y|), d from 0).
inline scalar_t l1(scalar_t const* a, scalar_t const* b, size_t dim) {
  scalar_t d = 0.0f;
  for (size_t i = 0; i < dim; ++i) d += std::abs(a[i] - b[i]);
  return d;
}
inline scalar_t lpinf(scalar_t const* a, scalar_t const* b, size_t dim) {
  scalar_t d = 0.0f;
  for (size_t i = 0; i < dim; ++i) d = std::max(d, std::abs(a[i] - b[i]));
  return d;
}
// metric_lninf (metric.hpp:157-186): d = std::min(d, |x - y|), d from the largest scalar.
inline scalar_t lninf(scalar_t const* a, scalar_t const* b, size_t dim) {
  scalar_t d = std::numeric_limits<scalar_t>::max();
  for (size_t i = 0; i < dim; ++i) d = std::min(d, std::abs(a[i] - b[i]));
  return d;
}
// metric: 0 L2 squared, 1 L1, 2 LPInf, 5 LNInf (3 and 4 are the topological metrics of the compiled
// reference only, oracle/ref_driver.cpp).
inline scalar_t point_distance(int metric, scalar_t const* a, scalar_t const* b, size_t dim) {
  return metric == 1 ? l1(a, b, dim) : metric == 2 ? lpinf(a, b, dim) : metric == 5 ? lninf(a, b, dim) : l2sq(a, b, dim);
}
// The one-dimensional form the searches apply to a split offset (metric.hpp:95-98, :120-123, :147-150).
inline scalar_t scalar_distance(int metric, scalar_t x) { return metric == 0 ? x * x : std::abs(x); }

// ---- visitors: search_visitor.hpp ------------------------------------------

struct visit_nn {  // :42-65 (exact) and :165-193 (approximate: scale by 1/e)
  neighbor_t* nn;
  bool approx;
  scalar_t e_inv;
  visit_nn(neighbor_t* out, bool a, scalar_t e) : nn(out), approx(a), e_inv(1.0f / e) {
    nn->distance = std::numeric_limits<scalar_t>::max();
  }
  scalar_t max() const { return nn->distance; }
  void operator()(int idx, scalar_t dst) {
    if (approx) dst = dst * e_inv;
    if (max() > dst) {
      nn->index = idx;
      nn->distance = dst;
    }
  }
};

struct visit_knn {  // :83-123 (exact) and :198-247 (approximate)
  neighbor_t* begin;
  neighbor_t* end;
  neighbor_t* active_end;
  bool approx;
  scalar_t e_inv;
  visit_knn(neighbor_t* b, neighbor_t* e_, bool a, scalar_t e)
      : begin(b), end(e_), active_end(b), approx(a), e_inv(1.0f / e) {
    (end - 1)->distance = std::numeric_limits<scalar_t>::max();  // :102
  }
  scalar_t max() const { return (end - 1)->distance; }
  void operator()(int idx, scalar_t dst) {
    if (approx) dst = dst * e_inv;
    if (max() > dst) {
      if (active_end < end) ++active_end;  // :108-110
      // insert_sorted, :24-38: shift while item < *prev (strict => stable).
      neighbor_t* it = active_end - 1;
      for (; it > begin && dst < (it - 1)->distance; --it) *it = *(it - 1);
      it->index = idx;
      it->distance = dst;
    }
  }
};

struct visit_radius {  // :127-156 (exact) and :252-288 (approximate)
  std::vector<neighbor_t>* out;
  scalar_t radius;
  bool approx;
  scalar_t e_inv;
  visit_radius(scalar_t r, std::vector<neighbor_t>* o, bool a, scalar_t e)
      : out(o), radius(r), approx(a), e_inv(1.0f / e) {
    if (approx) radius = r * e_inv;  // :265
    out->clear();                    // :136
  }
  scalar_t max() const { return radius; }
  void operator()(int idx, scalar_t dst) {
    if (approx) dst = dst * e_inv;
    if (max() > dst) out->push_back(neighbor_t{idx, dst});  // strict, :141
  }
};

// search_nearest_euclidean::search_nearest, kd_tree_search.hpp:52-105.
template <typename Visitor>
struct nearest_search {
  tree_t const& tree;
  scalar_t const* q;
  Visitor& visitor;
  std::vector<scalar_t> offset;  // node_box_offset_, :111, zeroed per query :47
  visit_counters* counters;

  nearest_search(tree_t const& t, scalar_t const* query, Visitor& v,
                 visit_counters* c)
      : tree(t), q(query), visitor(v), offset(t.dim, 0.0f), counters(c) {}

  void run() { descend(tree.root, 0.0f); }

  void descend(node_t const* node, scalar_t node_box_distance) {
    if (node->is_leaf()) {  // :54-59
      if (counters) {
        if (counters->n_leaf == 0) counters->n_first = counters->n_branch;
        ++counters->n_leaf;
      }
      bool const home_leaf = counters && counters->n_leaf == 1;
      for (int i = node->data.leaf.begin_idx; i < node->data.leaf.end_idx; ++i) {
        int const idx = tree.indices[static_cast<size_t>(i)];
        if (counters) ++counters->n_pts;
        visitor(idx, point_distance(tree.metric, q, tree.point(idx), tree.dim));
      }
      if (home_leaf) {
        for (scalar_t f : counters->first_far) counters->n_cand += visitor.max() >= f ? 1u : 0u;
      }
      return;
    }
    if (counters) ++counters->n_branch;
    size_t const sd = static_cast<size_t>(node->data.branch.split_dim);  // :63
    scalar_t const v = q[sd];
    scalar_t const left_max = node->data.branch.left_max;
    scalar_t const right_min = node->data.branch.right_min;
    scalar_t new_offset;
    node_t const* first;
    node_t const* second;
    if ((left_max + right_min - v - v) > 0) {  // :76 (left-assoc: ((a+b)-v)-v)
      first = node->left;
      second = node->right;
      new_offset = scalar_distance(tree.metric, right_min - v);  // :80, metric.hpp:120-123
    } else {
      first = node->right;
      second = node->left;
      new_offset = scalar_distance(tree.metric, left_max - v);  // :84
    }
    if (counters && counters->n_leaf == 0) {
      counters->first_far.push_back(node_box_distance - offset[sd] + new_offset);
    }
    descend(first, node_box_distance);  // :88

    scalar_t const old_offset = offset[sd];  // :93
    node_box_distance = node_box_distance - old_offset + new_offset;  // :94
    if (visitor.max() >= node_box_distance) {  // :99
      offset[sd] = new_offset;
      descend(second, node_box_distance);
      offset[sd] = old_offset;
    }
  }
};

// search_box, kd_tree_search.hpp:238-381: report every index inside [min,max].
struct box_search {
  tree_t const& tree;
  box_t const& query;
  std::vector<int>& out;

  void report_all(node_t const* node) {  // :349-361
    if (node->is_leaf()) {
      for (int i = node->data.leaf.begin_idx; i < node->data.leaf.end_idx; ++i)
        out.push_back(tree.indices[static_cast<size_t>(i)]);
    } else {
      report_all(node->left);
      report_all(node->right);
    }
  }

  void descend(node_t const* node, box_t& box) {  // :273-345
    if (node->is_leaf()) {
      for (int i = node->data.leaf.begin_idx; i < node->data.leaf.end_idx; ++i) {
        int const idx = tree.indices[static_cast<size_t>(i)];
        if (query.contains(tree.point(idx))) out.push_back(idx);
      }
      return;
    }
    size_t const sd = static_cast<size_t>(node->data.branch.split_dim);
    scalar_t old_value = box.mx(sd);
    box.mx(sd) = node->data.branch.left_max;
    if (query.contains(box)) {
      report_all(node->left);
    } else if (query.mn(sd) <= node->data.branch.left_max) {
      descend(node->left, box);
    }
    box.mx(sd) = old_value;

    old_value = box.mn(sd);
    box.mn(sd) = node->data.branch.right_min;
    if (query.contains(box)) {
      report_all(node->right);
    } else if (query.mx(sd) >= node->data.branch.right_min) {
      descend(node->right, box);
    }
    box.mn(sd) = old_value;
  }
};

// kd_tree_data::write, kd_tree_data.hpp:109-135 + save :50-54, through
// stream_wrapper.hpp's raw POD writes (size_t counts, native endianness).
struct byte_sink {
  std::string s;
  template <typename T>
  void put(T const& v) {
    s.append(reinterpret_cast<char const*>(&v), sizeof(T));
  }
  template <typename T>
  void put_n(T const* p, size_t count) {
    s.append(reinterpret_cast<char const*>(p), sizeof(T) * count);
  }
};

void write_node(node_t const* node, byte_sink& out) {
  if (node->is_leaf()) {
    out.put(true);
    out.put(node->data.leaf);
  } else {
    out.put(false);
    // kd_tree_branch_single (kd_tree_node.hpp:43-50) is {int split_dim; float left_max; float
    // right_min}: 12 bytes; the forest's outer bounds are not part of the euclidean stream.
    // The reference writes the struct whole (kd_tree_data.hpp:116): with double scalars that
    // includes 4 bytes of padding after split_dim (indeterminate there, zero here).
    struct {
      int split_dim;
      scalar_t left_max;
      scalar_t right_min;
    } rec;
    std::memset(&rec, 0, sizeof(rec));
    rec.split_dim = node->data.branch.split_dim;
    rec.left_max = node->data.branch.left_max;
    rec.right_min = node->data.branch.right_min;
    out.put(rec);
    write_node(node->left, out);
    write_node(node->right, out);
  }
}

#ifndef PTKOR_DOUBLE
// Flat DFS pre-order export: mirrors the save stream, one 16-byte record per
// node.  Branch: {left_max, right_min, right_child_index, split_dim};
// leaf: {begin, end, 0xFFFFFFFF, 0}.  Used by tests to check the product's own
// flattening.
struct flat_node {
  std::uint32_t w[4];
};
static_assert(sizeof(flat_node) == 16, "layout");

std::uint32_t flatten(node_t const* node, std::vector<flat_node>& out,
                      std::uint32_t depth, std::uint32_t& max_depth) {
  std::uint32_t const self = static_cast<std::uint32_t>(out.size());
  out.push_back(flat_node{});
  if (depth > max_depth) max_depth = depth;
  if (node->is_leaf()) {
    std::int32_t b = node->data.leaf.begin_idx, e = node->data.leaf.end_idx;
    std::memcpy(&out[self].w[0], &b, 4);
    std::memcpy(&out[self].w[1], &e, 4);
    out[self].w[2] = 0xFFFFFFFFu;
    out[self].w[3] = 0;
  } else {
    flatten(node->left, out, depth + 1, max_depth);
    std::uint32_t const r = flatten(node->right, out, depth + 1, max_depth);
    std::memcpy(&out[self].w[0], &node->data.branch.left_max, 4);
    std::memcpy(&out[self].w[1], &node->data.branch.right_min, 4);
    out[self].w[2] = r;
    out[self].w[3] = static_cast<std::uint32_t>(node->data.branch.split_dim);
  }
  return self;
}

#endif  // !PTKOR_DOUBLE

constexpr int kChunk = 128;  // _pyco_tree/kd_tree.hpp:94

struct ragged_nb {
  std::vector<std::vector<neighbor_t>> rows;
};
struct ragged_idx {
  std::vector<std::vector<int>> rows;
};

}  // namespace

extern "C" {

void* ptkor_create(scalar_t const* pts, size_t n, size_t dim, size_t max_leaf) {
  if (n == 0 || dim == 0 || max_leaf == 0) return nullptr;  // builder.hpp:93,479
  auto* t = new tree_t;
  t->dim = dim;
  t->n = n;
  t->max_leaf = max_leaf;
  t->pts.assign(pts, pts + n * dim);
  t->build();
  return t;
}

void ptkor_destroy(void* t) { delete static_cast<tree_t*>(t); }

// The kd_tree::save byte stream (kd_tree.hpp:367-370).
size_t ptkor_save(void* handle, unsigned char* buf, size_t cap) {
  auto* t = static_cast<tree_t*>(handle);
  byte_sink out;
  out.put(static_cast<size_t>(t->dim));                       // data.hpp:52
  out.put(static_cast<size_t>(t->indices.size()));            // stream_wrapper.hpp:77-81
  out.put_n(t->indices.data(), t->indices.size());
  out.put_n(t->root_box.c.data(), t->dim);                    // min
  out.put_n(t->root_box.c.data() + t->dim, t->dim);           // max
  write_node(t->root, out);
  if (buf != nullptr && cap >= out.s.size())
    std::memcpy(buf, out.s.data(), out.s.size());
  return out.s.size();
}

#ifndef PTKOR_DOUBLE
// Flat export. Call with nodes == nullptr to get the node count.
size_t ptkor_flatten(void* handle, void* nodes, size_t cap, int* indices,
                     scalar_t* root_min, scalar_t* root_max,
                     std::uint32_t* max_depth_out) {
  auto* t = static_cast<tree_t*>(handle);
  std::vector<flat_node> out;
  std::uint32_t max_depth = 0;
  flatten(t->root, out, 0, max_depth);
  if (nodes != nullptr && cap >= out.size()) {
    std::memcpy(nodes, out.data(), out.size() * sizeof(flat_node));
    if (indices) std::memcpy(indices, t->indices.data(), t->n * sizeof(int));
    if (root_min) std::memcpy(root_min, t->root_box.c.data(), t->dim * 4);
    if (root_max) std::memcpy(root_max, t->root_box.c.data() + t->dim, t->dim * 4);
    if (max_depth_out) *max_depth_out = max_depth;
  }
  return out.size();
}

#endif  // !PTKOR_DOUBLE

void ptkor_set_threads(int threads) {
  if (threads > 0) omp_set_num_threads(threads);
}

int ptkor_max_threads() { return omp_get_max_threads(); }

// kd_tree::search_nn over a batch (kd_tree.hpp:126-129,155-159); approx != 0
// selects the approximate visitor with ratio e.  counters: optional nq x 5
// uint32 {n_branch, n_leaf, n_pts, n_first, n_cand}.
void ptkor_search_nn(void* handle, scalar_t const* q, size_t nq, int approx,
                     scalar_t e, void* out, std::uint32_t* counters) {
  auto* t = static_cast<tree_t*>(handle);
  auto* o = static_cast<neighbor_t*>(out);
  std::ptrdiff_t const count = static_cast<std::ptrdiff_t>(nq);
#pragma omp parallel for schedule(dynamic, kChunk)
  for (std::ptrdiff_t i = 0; i < count; ++i) {
    visit_counters c;
    visit_nn v(o + i, approx != 0, e);
    nearest_search<visit_nn> s(*t, q + static_cast<size_t>(i) * t->dim, v,
                               counters ? &c : nullptr);
    s.run();
    if (counters) {
      counters[5 * i + 0] = c.n_branch;
      counters[5 * i + 1] = c.n_leaf;
      counters[5 * i + 2] = c.n_pts;
      counters[5 * i + 3] = c.n_first;
      counters[5 * i + 4] = c.n_cand;
    }
  }
}

// kd_tree::search_knn(x, begin, end) over a batch (kd_tree.hpp:169-181,205-218):
// row i of out holds k slots.  As in the reference, slots beyond min(k, n) are
// only meaningful when n >= k (the vector overload clamps, kd_tree.hpp:193);
// callers pass k <= n.
void ptkor_search_knn(void* handle, scalar_t const* q, size_t nq, size_t k,
                      int approx, scalar_t e, void* out,
                      std::uint32_t* counters) {
  auto* t = static_cast<tree_t*>(handle);
  auto* o = static_cast<neighbor_t*>(out);
  std::ptrdiff_t const count = static_cast<std::ptrdiff_t>(nq);
#pragma omp parallel for schedule(dynamic, kChunk)
  for (std::ptrdiff_t i = 0; i < count; ++i) {
    size_t const ui = static_cast<size_t>(i);
    visit_counters c;
    visit_knn v(o + ui * k, o + ui * k + k, approx != 0, e);
    nearest_search<visit_knn> s(*t, q + ui * t->dim, v, counters ? &c : nullptr);
    s.run();
    if (counters) {
      counters[5 * ui + 0] = c.n_branch;
      counters[5 * ui + 1] = c.n_leaf;
      counters[5 * ui + 2] = c.n_pts;
      counters[5 * ui + 3] = c.n_first;
      counters[5 * ui + 4] = c.n_cand;
    }
  }
}

// kd_tree::search_radius over a batch (kd_tree.hpp:257-290). sort != 0 applies
// std::sort by distance (search_visitor.hpp:148), which is unstable: rows with
// equal distances are only comparable as multisets.
void* ptkor_search_radius(void* handle, scalar_t const* q, size_t nq, scalar_t radius,
                          int sort, int approx, scalar_t e, std::uint64_t* offsets,
                          std::uint32_t* counters) {
  auto* t = static_cast<tree_t*>(handle);
  auto* r = new ragged_nb;
  r->rows.resize(nq);
  std::ptrdiff_t const count = static_cast<std::ptrdiff_t>(nq);
#pragma omp parallel for schedule(dynamic, kChunk)
  for (std::ptrdiff_t i = 0; i < count; ++i) {
    size_t const ui = static_cast<size_t>(i);
    visit_counters c;
    visit_radius v(radius, &r->rows[ui], approx != 0, e);
    nearest_search<visit_radius> s(*t, q + ui * t->dim, v,
                                   counters ? &c : nullptr);
    s.run();
    if (sort) {
      std::sort(r->rows[ui].begin(), r->rows[ui].end(),
                [](neighbor_t const& a, neighbor_t const& b) {
                  return a.distance < b.distance;  // core.hpp:49-54
                });
    }
    if (counters) {
      counters[5 * ui + 0] = c.n_branch;
      counters[5 * ui + 1] = c.n_leaf;
      counters[5 * ui + 2] = c.n_pts;
      counters[5 * ui + 3] = c.n_first;
      counters[5 * ui + 4] = c.n_cand;
    }
  }
  std::uint64_t acc = 0;
  for (size_t i = 0; i < nq; ++i) {
    offsets[i] = acc;
    acc += r->rows[i].size();
  }
  offsets[nq] = acc;
  return r;
}

void ptkor_radius_copy(void* h, void* out) {
  auto* r = static_cast<ragged_nb*>(h);
  auto* o = static_cast<neighbor_t*>(out);
  for (auto const& row : r->rows) {
    if (!row.empty()) std::memcpy(o, row.data(), row.size() * sizeof(neighbor_t));
    o += row.size();
  }
}

void ptkor_radius_free(void* h) { delete static_cast<ragged_nb*>(h); }

// kd_tree::search_box over a batch (kd_tree.hpp:296-318).
void* ptkor_search_box(void* handle, scalar_t const* mins, scalar_t const* maxs,
                       size_t nb, std::uint64_t* offsets) {
  auto* t = static_cast<tree_t*>(handle);
  auto* r = new ragged_idx;
  r->rows.resize(nb);
  std::ptrdiff_t const count = static_cast<std::ptrdiff_t>(nb);
#pragma omp parallel for schedule(dynamic, kChunk)
  for (std::ptrdiff_t i = 0; i < count; ++i) {
    size_t const ui = static_cast<size_t>(i);
    box_t query(t->dim);
    for (size_t d = 0; d < t->dim; ++d) {
      query.mn(d) = mins[ui * t->dim + d];
      query.mx(d) = maxs[ui * t->dim + d];
    }
    box_t work = t->root_box;
    r->rows[ui].clear();
    box_search s{*t, query, r->rows[ui]};
    s.descend(t->root, work);
  }
  std::uint64_t acc = 0;
  for (size_t i = 0; i < nb; ++i) {
    offsets[i] = acc;
    acc += r->rows[i].size();
  }
  offsets[nb] = acc;
  return r;
}

void ptkor_box_copy(void* h, int* out) {
  auto* r = static_cast<ragged_idx*>(h);
  for (auto const& row : r->rows) {
    if (!row.empty()) std::memcpy(out, row.data(), row.size() * sizeof(int));
    out += row.size();
  }
}

void ptkor_box_free(void* h) { delete static_cast<ragged_idx*>(h); }

// Known-answer hook mirroring test/pico_tree/kd_tree_builder_test.cpp:134-197.
void ptkor_sliding_midpoint_2d(scalar_t const* pts, size_t n, int* indices,
                               scalar_t const* box_min, scalar_t const* box_max,
                               size_t* split_offset, size_t* split_dim,
                               scalar_t* split_val) {
  tree_t t;
  t.dim = 2;
  t.n = n;
  t.pts.assign(pts, pts + 2 * n);
  box_t box(2);
  for (size_t i = 0; i < 2; ++i) {
    box.mn(i) = box_min[i];
    box.mx(i) = box_max[i];
  }
  int* split = nullptr;
  size_t sd = 0;
  scalar_t sv = 0;
  t.split_sliding_midpoint(indices, indices + n, box, split, sd, sv);
  *split_offset = static_cast<size_t>(split - indices);
  *split_dim = sd;
  *split_val = sv;
}

// metric_l2_squared known answers (test/pico_tree/metric_test.cpp:37-45).
scalar_t ptkor_l2sq(scalar_t const* a, scalar_t const* b, size_t dim) {
  return l2sq(a, b, dim);
}
scalar_t ptkor_l2sq_scalar(scalar_t x) { return x * x; }
// metric_l1 / metric_lpinf known answers (metric_test.cpp:27-35, :47-55) and the search metric.
scalar_t ptkor_distance(int metric, scalar_t const* a, scalar_t const* b, size_t dim) {
  return point_distance(metric, a, b, dim);
}
scalar_t ptkor_distance_scalar(int metric, scalar_t x) { return scalar_distance(metric, x); }
int ptkor_set_metric(void* tree, int metric) {
  if (tree == nullptr || !(metric == 0 || metric == 1 || metric == 2 || metric == 5)) return -1;
  static_cast<tree_t*>(tree)->metric = metric;
  return 0;
}


#ifdef PTKOR_DOUBLE
}  // extern "C"
#else
// =====================================================================================
// kd_forest (BASELINE config 5).  Restates
//   /root/reference/examples/pico_understory/pico_understory/kd_forest.hpp:70-115
//   .../internal/kd_tree_priority_search.hpp:48-130
//   .../internal/rkd_tree_hh_data.hpp:56-90 (Householder reflection)
// with the two differences the product documents (pico_tree_amd/csrc/ptk_forest.hpp): the shared
// k-list is de-duplicated by index and point distances are measured in the original space; the
// queue orders equal distances by (branch before leaf, DFS stream position) and holds at most
// kForestQueue nodes.
// Reflection vectors are INPUTS (the product reports the ones it drew).
}  // extern "C"

namespace {

constexpr size_t kForestQueue = 1024;  // == ptk::kForestQueue

struct forest_t {
  size_t dim = 0;
  size_t n = 0;
  std::vector<float> pts;                      // original space
  std::vector<std::vector<float>> rotation;    // per tree
  std::vector<std::unique_ptr<tree_t>> trees;  // over reflected copies
};

void number_nodes(node_t* node, std::uint32_t& next) {
  node->id = next++;
  if (!node->is_leaf()) {
    number_nodes(node->left, next);
    number_nodes(node->right, next);
  }
}

void reflect(std::vector<float> const& r, float const* x, float* y, size_t dim) {  // hh_data.hpp:80-90
  float dot = 0.0f;
  for (size_t i = 0; i < dim; ++i) dot += r[i] * x[i];
  dot *= 2.0f;
  for (size_t i = 0; i < dim; ++i) y[i] = x[i] - (dot * r[i]);
}

// Point distance of the forest search, in the original space.  For dimensions that are a multiple
// of 128 the product's wavefront reads a row with 64 lanes (lane l: elements 2l, 2l + 1 of every
// 128-float segment) and adds the 64 partial sums in a fixed tree: pairs of lanes differing in
// bit 5, then bit 4, ... bit 0.  Other dimensions: left to right like l2sq.
inline float forest_l2sq(float const* q, float const* p, size_t dim) {
  if (dim % 128 != 0) return l2sq(q, p, dim);
  float s[64];
  for (size_t l = 0; l < 64; ++l) {
    float acc = 0.0f;
    for (size_t m = 0; m < dim / 128; ++m) {
      float const d0 = q[128 * m + 2 * l] - p[128 * m + 2 * l];
      float const d1 = q[128 * m + 2 * l + 1] - p[128 * m + 2 * l + 1];
      acc = acc + d0 * d0;
      acc = acc + d1 * d1;
    }
    s[l] = acc;
  }
  for (size_t bit = 32; bit >= 1; bit >>= 1)
    for (size_t l = 0; l < 64; ++l)
      if ((l & bit) == 0) s[l] = s[l] + s[l | bit];
  return s[0];
}

struct forest_query {
  forest_t const& f;
  float const* q;   // original space
  size_t k;
  std::vector<neighbor_t> list;  // ascending, distinct indices
  struct entry {
    float d;
    node_t const* node;
  };
  std::vector<entry> queue;

  float max() const { return list.size() == k ? list.back().distance : std::numeric_limits<float>::max(); }

  void visit(int idx, float d) {
    if (!(max() > d)) return;  // search_visitor.hpp:107
    for (neighbor_t const& nb : list)
      if (nb.index == idx) return;  // de-duplication
    size_t pos = list.size();
    while (pos > 0 && d < list[pos - 1].distance) --pos;  // insert_sorted, :24-38 (stable)
    list.insert(list.begin() + static_cast<std::ptrdiff_t>(pos), neighbor_t{idx, d});
    if (list.size() > k) list.pop_back();
  }

  // priority_search::search_nearest, :66-130.
  void descend(tree_t const& t, float const* qr, node_t const* node, float nbd) {
    if (node->is_leaf()) {
      for (int i = node->data.leaf.begin_idx; i < node->data.leaf.end_idx; ++i) {
        int const idx = t.indices[static_cast<size_t>(i)];
        visit(idx, forest_l2sq(q, f.pts.data() + static_cast<size_t>(idx) * f.dim, f.dim));
      }
      return;
    }
    auto const& b = node->data.branch;
    float const v = qr[b.split_dim];
    float old_offset, new_offset;
    node_t const *first, *second;
    if ((b.left_max + b.right_min - v - v) > 0) {  // :97
      first = node->left;
      second = node->right;
      old_offset = v > b.left_min ? 0.0f : (b.left_min - v) * (b.left_min - v);
      new_offset = (b.right_min - v) * (b.right_min - v);
    } else {
      first = node->right;
      second = node->left;
      old_offset = v < b.right_max ? 0.0f : (b.right_max - v) * (b.right_max - v);
      new_offset = (b.left_max - v) * (b.left_max - v);
    }
    descend(t, qr, first, nbd);
    nbd = nbd - old_offset + new_offset;  // :123
    if (max() > nbd && queue.size() < kForestQueue) queue.push_back({nbd, second});  // :126
  }

  void search_tree(tree_t const& t, float const* qr, size_t max_leaves) {  // :48-63
    queue.clear();
    queue.push_back({0.0f, t.root});
    size_t leaves = 0;
    while (!queue.empty()) {
      size_t best = 0;
      for (size_t i = 1; i < queue.size(); ++i) {
        // Equal distances: branches before leaves, then depth-first (stream) order.
        auto const rank = [](node_t const* n) { return std::make_pair(n->is_leaf(), n->id); };
        if (queue[i].d < queue[best].d || (queue[i].d == queue[best].d && rank(queue[i].node) < rank(queue[best].node)))
          best = i;
      }
      entry const top = queue[best];
      if (leaves >= max_leaves || max() < top.d) break;
      queue[best] = queue.back();
      queue.pop_back();
      descend(t, qr, top.node, top.d);
      ++leaves;
    }
  }
};

}  // namespace

extern "C" {

void* ptkor_forest_create(float const* pts, size_t n, size_t dim, size_t max_leaf, size_t n_trees,
                          float const* rotations) {
  if (n == 0 || dim == 0 || max_leaf == 0 || n_trees == 0) return nullptr;
  auto* f = new forest_t;
  f->dim = dim;
  f->n = n;
  f->pts.assign(pts, pts + n * dim);
  for (size_t ti = 0; ti < n_trees; ++ti) {
    f->rotation.emplace_back(rotations + ti * dim, rotations + (ti + 1) * dim);
    auto t = std::make_unique<tree_t>();
    t->dim = dim;
    t->n = n;
    t->max_leaf = max_leaf;
    t->pts.resize(n * dim);
    for (size_t i = 0; i < n; ++i) reflect(f->rotation.back(), pts + i * dim, t->pts.data() + i * dim, dim);
    t->build();
    std::uint32_t next = 0;
    number_nodes(t->root, next);
    f->trees.push_back(std::move(t));
  }
  return f;
}

void ptkor_forest_destroy(void* f) { delete static_cast<forest_t*>(f); }

void ptkor_forest_search_knn(void* handle, float const* q, size_t nq, size_t k, size_t max_leaves, void* out) {
  auto* f = static_cast<forest_t*>(handle);
  auto* rows = static_cast<neighbor_t*>(out);
  long long const count = static_cast<long long>(nq);
#pragma omp parallel for schedule(dynamic, 16)
  for (long long i = 0; i < count; ++i) {
    forest_query s{*f, q + static_cast<size_t>(i) * f->dim, k, {}, {}};
    std::vector<float> qr(f->dim);
    for (size_t ti = 0; ti < f->trees.size(); ++ti) {
      reflect(f->rotation[ti], s.q, qr.data(), f->dim);
      s.search_tree(*f->trees[ti], qr.data(), max_leaves);
    }
    for (size_t j = 0; j < k; ++j)
      rows[static_cast<size_t>(i) * k + j] =
          j < s.list.size() ? s.list[j] : neighbor_t{-1, std::numeric_limits<float>::max()};
  }
}

}  // extern "C"

#endif  // PTKOR_DOUBLE
