/* ptk.h -- C ABI of the MI355X batched k-NN backend (libptk.so).
 *
 * This is the drop-in boundary for the hot path named by BASELINE.json: batched
 * search_nn / search_knn / search_radius (and search_box) on a pico_tree
 * kd_tree<Space, metric_l2_squared, int> over float32 points.  Plain C types
 * only: no exceptions, no STL, no torch types cross this line.  Every entry
 * point returns PTK_OK (0) or a negative ptk_status; the message for the last
 * failure on the calling thread is available from ptk_last_error().
 *
 * What each entry point replaces in the reference (paths relative to
 * /root/reference):
 *
 *   ptk_tree_create_from_points   kd_tree ctor with max_leaf_size_t, bounds from
 *                                 the space, sliding midpoint rule:
 *                                 src/pico_tree/pico_tree/kd_tree.hpp:76-88,
 *                                 internal/kd_tree_builder.hpp:456-512; as bound
 *                                 by the Python module in
 *                                 src/pyco_tree/pico_tree/_pyco_tree/kd_tree.hpp:112-115
 *   ptk_tree_create               the same tree handed over already built and
 *                                 flattened (what include/pico_tree/kd_tree.hpp
 *                                 does for arbitrary Space types); the node
 *                                 stream is the DFS pre-order of
 *                                 internal/kd_tree_data.hpp:109-135
 *   ptk_search_knn[_device]       the OpenMP batch loop over kd_tree::search_knn:
 *                                 _pyco_tree/kd_tree.hpp:117-135 (exact) and
 *                                 :144-168 (approximate), i.e. kd_tree.hpp:169-181,
 *                                 :205-218 per row; k == 1 is search_nn
 *                                 (kd_tree.hpp:126-129)
 *   ptk_search_radius_*           the batch loop over kd_tree::search_radius:
 *                                 _pyco_tree/kd_tree.hpp:179-200, :211-233, i.e.
 *                                 kd_tree.hpp:257-290 per row
 *   ptk_search_box_*              the batch loop over kd_tree::search_box:
 *                                 _pyco_tree/kd_tree.hpp:245-268, kd_tree.hpp:296-318
 *   ptk_tree_destroy              ~kd_tree
 *   ptk_tree64_* / ptk_search64_* the same entry points for kd_trees over double
 *                                 points (_pyco_tree/kd_tree.hpp:383-445 dispatches
 *                                 on the array dtype)
 *
 * Results contract: neighbour indices are bit-identical to the reference CPU
 * kd_tree on the same inputs, squared distances bit-identical to the reference
 * compiled without FMA contraction (-ffp-contract=off).
 *
 * Threading: a ptk_tree is immutable after creation; search calls on one handle
 * may be issued concurrently from several host threads (each call uses its own
 * scratch; calls that share a HIP stream serialise on it).
 */
#ifndef PTK_H_
#define PTK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTK_VERSION 101 /* 0.1.1: ptk_warmup, ptk_profile_get_sized */

typedef enum ptk_status {
  PTK_OK = 0,
  PTK_ERR_INVALID = -1,     /* bad argument (null pointer, k == 0, dim == 0 ...) */
  PTK_ERR_UNSUPPORTED = -2, /* valid request this build cannot run on the GPU   */
  PTK_ERR_DEVICE = -3,      /* HIP runtime error, or no usable gfx950 device    */
  PTK_ERR_NOMEM = -4        /* host or device allocation failed                 */
} ptk_status;

/* Layout-identical to pico_tree::neighbor<int, float> (core.hpp:24-46) and to
 * the Python binding's [('index','<i4'),('distance','<f4')] record. */
typedef struct ptk_neighbor {
  int32_t index;
  float distance;
} ptk_neighbor;

/* One node of the flat tree, DFS pre-order, 16 bytes.  The left child of a
 * branch is the next node (self + 1).
 *   branch: a = bits of float left_max, b = bits of float right_min,
 *           right = index of the right child, split_dim = split axis
 *   leaf:   a = begin, b = end (positions in the indices array, int32),
 *           right = PTK_LEAF, split_dim = 0
 * Field meanings follow internal/kd_tree_node.hpp:31-50. */
#define PTK_LEAF 0xFFFFFFFFu
typedef struct ptk_node {
  uint32_t a;
  uint32_t b;
  uint32_t right;
  uint32_t split_dim;
} ptk_node;

typedef struct ptk_tree_desc {
  uint32_t dim;          /* spatial dimension (>= 1)                            */
  uint64_t n_points;     /* number of points (>= 1, < 2^31)                     */
  const float* points;   /* n_points x dim, row-major, ORIGINAL order (host)    */
  uint64_t n_nodes;      /* number of nodes in the DFS stream                   */
  const ptk_node* nodes; /* host                                                */
  const int32_t* indices;/* n_points entries: leaf-ordered permutation (host)   */
  const float* root_min; /* dim floats: start bounds of the build, or NULL =     */
  const float* root_max; /*   bounding box of the points (both or neither)      */
  uint32_t max_depth;    /* informational; recomputed from the node stream      */
  int32_t device;        /* HIP device ordinal, PTK_DEVICE_CURRENT or _NONE     */
} ptk_tree_desc;

#define PTK_DEVICE_CURRENT (-1) /* whatever hipGetDevice() returns                 */
#define PTK_DEVICE_NONE (-2)    /* host-only handle: structure queries work, every
                                   search returns PTK_ERR_DEVICE                    */

typedef struct ptk_tree ptk_tree; /* opaque */

typedef struct ptk_tree_info {
  uint32_t dim;
  uint64_t n_points;
  uint64_t n_nodes;      /* nodes in the DFS stream (branches + leaves)         */
  uint64_t n_leaves;
  uint32_t max_depth;
  uint32_t max_leaf_count;
  uint64_t device_bytes; /* HBM held by the handle                              */
  int32_t device;
} ptk_tree_info;

/* Query-order policy for the nearest-neighbour kernels.  Coherent (spatially
 * sorted) batches traverse the tree several times faster; with PTK_REORDER_ON
 * the backend sorts the batch along a Morton curve on the device and scatters
 * results back, so the caller-visible row order never changes.  PTK_REORDER_AUTO
 * (the default) sorts batches of 8 192 queries or more; a k = 1 batch is sampled on
 * the device first and, if it already is in a coherent order (a scan in scan order),
 * searched in the caller's order -- the verdict never travels to the host. */
enum { PTK_REORDER_AUTO = 0, PTK_REORDER_ON = 1, PTK_REORDER_OFF = 2 };

/* ---- library ---------------------------------------------------------- */
int ptk_version(void);
const char* ptk_last_error(void); /* thread-local, never NULL */
int ptk_device_count(void);       /* number of visible HIP devices, or < 0; no side effects */
/* Starts loading the library's device code for `device` (< 0: the calling thread's current device)
 * on a thread of the library and returns at once: the first ptk_tree_create* for that device then
 * does not wait for it (~0.2 s).  Optional -- the first creation starts it itself; one load per
 * device and process; PTK_EAGER_WARMUP=0 in the environment turns it off. */
int ptk_warmup(int32_t device);

/* ---- tree lifetime ---------------------------------------------------- */

/* Builds the kd-tree on the host (sliding midpoint, max_leaf_size stop, bounds
 * from the points) and uploads it.  points may be freed after the call. */
int ptk_tree_create_from_points(const float* points, uint64_t n_points,
                                uint32_t dim, uint64_t max_leaf_size,
                                int32_t device, ptk_tree** out);

/* Uploads an already built flat tree.  All host arrays may be freed after. */
int ptk_tree_create(const ptk_tree_desc* desc, ptk_tree** out);

void ptk_tree_destroy(ptk_tree* tree);

int ptk_tree_get_info(const ptk_tree* tree, ptk_tree_info* info);

/* Copies the host-side flat tree out (for bindings that save it or inspect it).
 * Pass NULL to skip an array.  nodes needs n_nodes entries, indices n_points. */
int ptk_tree_get_flat(const ptk_tree* tree, ptk_node* nodes, int32_t* indices,
                      float* root_min, float* root_max);

int ptk_tree_set_reorder(ptk_tree* tree, int mode);

/* Metric of the searches on this handle (the tree itself does not depend on it).  The
 * reference's euclidean search is generic over the metric (kd_tree.hpp:23, metric.hpp:72-150):
 *   PTK_METRIC_L2_SQUARED  metric_l2_squared (default): squared distances, radius squared
 *   PTK_METRIC_L1          metric_l1:    sum of |differences|
 *   PTK_METRIC_LPINF       metric_lpinf: max of |differences|
 *   PTK_METRIC_LNINF       metric_lninf: min of |differences| (metric.hpp:157-186)
 *   PTK_METRIC_SO2         metric_so2 (metric.hpp:186-220): points on the circle [0, 1] / 0 ~ 1; dim 1 only
 *   PTK_METRIC_SE2_SQUARED metric_se2_squared (metric.hpp:228-257): x, y on the plane, angle on the circle;
 *                          dim 3 only.  The two topological metrics are searched as the reference's
 *                          search_nearest_topological does (kd_tree_search.hpp:115-229) and need the four
 *                          bounds per branch: trees made by ptk_tree_create_from_points have them; for
 *                          ptk_tree_create hand them over with ptk_tree_set_outer_bounds first.  float32,
 *                          knn and radius searches (the box search of these trees stays on the host members).
 * Set once, before the first search. */
enum { PTK_METRIC_L2_SQUARED = 0, PTK_METRIC_L1 = 1, PTK_METRIC_LPINF = 2, PTK_METRIC_LNINF = 3, PTK_METRIC_SO2 = 4,
       PTK_METRIC_SE2_SQUARED = 5 };
int ptk_tree_set_metric(ptk_tree* tree, int metric);
/* The two bounds of every branch that ptk_node does not hold -- outer[2 i] = min of the left
 * child's box, outer[2 i + 1] = max of the right child's box along the split axis of node i (the
 * other half of the reference's kd_tree_node_topological, kd_tree_node.hpp:56-67); entries of leaf
 * nodes are ignored.  Needed (before ptk_tree_set_metric) by the topological metrics on a tree made
 * by ptk_tree_create. */
int ptk_tree_set_outer_bounds(ptk_tree* tree, const float* outer, uint64_t n_nodes);
/* The same array back (2 floats per node; PTK_ERR_INVALID if the tree has none). */
int ptk_tree_get_outer_bounds(const ptk_tree* tree, float* outer);

/* The tree in the reference's own binary format (kd_tree::save / kd_tree::load,
 * kd_tree.hpp:336-370, internal/kd_tree_data.hpp:43-58,90-135): sdim, indices,
 * root box, nodes depth-first.  Files written by the reference load here and
 * vice versa (the Python module's PKD header, _pyco_tree/kd_tree.hpp:547-614,
 * is added by the binding).  ptk_tree_serialize: pass buf == NULL to get the
 * size; PTK_ERR_INVALID if cap is too small.  Like the reference, loading does
 * not check that the stream belongs to the points (kd_tree.hpp:346-351). */
int ptk_tree_serialize(const ptk_tree* tree, void* buf, uint64_t cap, uint64_t* size);
int ptk_tree_create_from_stream(const float* points, uint64_t n_points, uint32_t dim,
                                const void* stream, uint64_t stream_bytes,
                                int32_t device, ptk_tree** out);
/* The same for a tree over a topological space (kd_tree<space, metric_so2 | metric_se2_squared>::save /
 * ::load): its branches carry four bounds on disk (kd_tree_branch_double, internal/kd_tree_node.hpp:52-67).
 * Loading keeps the two outer ones with the handle (as ptk_tree_set_outer_bounds would), so that
 * ptk_tree_set_metric(PTK_METRIC_SO2 | PTK_METRIC_SE2_SQUARED) can follow at once; writing needs them. */
int ptk_tree_serialize_topological(const ptk_tree* tree, void* buf, uint64_t cap, uint64_t* size);
int ptk_tree_create_from_topological_stream(const float* points, uint64_t n_points, uint32_t dim,
                                            const void* stream, uint64_t stream_bytes,
                                            int32_t device, ptk_tree** out);

/* ---- k nearest neighbours --------------------------------------------- */

/* Host buffers.  queries: nq x dim row-major.  out: nq x k row-major; row i is
 * the ascending k-list of query i.  k is clamped by the CALLER to <= n_points
 * (kd_tree.hpp:193); k > n_points is PTK_ERR_INVALID.  e is the approximation
 * ratio in metric units (kd_tree.hpp:131-153); e == 1 is the exact search. */
int ptk_search_knn(const ptk_tree* tree, const float* queries, uint64_t nq,
                   uint32_t k, float e, ptk_neighbor* out);

/* Device buffers on the tree's device; asynchronous on `stream` (a hipStream_t,
 * NULL = the default stream).  No host synchronisation is performed: the call only
 * enqueues -- the decisions that depend on the batch (is it already in a coherent order?
 * which queries need the cooperative search?) are taken on the device. */
int ptk_search_knn_device(const ptk_tree* tree, const float* d_queries,
                          uint64_t nq, uint32_t k, float e,
                          ptk_neighbor* d_out, void* stream);

/* ---- radius search (ragged output) ------------------------------------ */

/* Pass 1: counts[i] = number of points with distance < radius (strict,
 * search_visitor.hpp:141).  radius is in metric units (squared for L2^2). */
int ptk_search_radius_count(const ptk_tree* tree, const float* queries,
                            uint64_t nq, float radius, float e,
                            uint64_t* counts);
/* Pass 2: offsets has nq + 1 entries (exclusive scan of counts); out receives
 * offsets[nq] records, row i in reference traversal order, or ascending by
 * distance when sort != 0 (ties in unspecified order, like std::sort). */
int ptk_search_radius_fill(const ptk_tree* tree, const float* queries,
                           uint64_t nq, float radius, float e,
                           const uint64_t* offsets, ptk_neighbor* out,
                           int sort);

/* Device forms.  The count pass keeps the rows it finds in a block of device
 * memory owned by the handle (PTK_RADIUS_CAPTURE_MB, default 16384,
 * 0 = off).  A fill call whose (d_queries, nq, radius, e, stream) repeat the
 * LAST count call on the handle -- the normal sequence -- copies them out
 * instead of searching again; any other fill call searches again.  The
 * contents of d_queries must not change between the two calls (they must not
 * in the two-pass form either: d_offsets would no longer fit). */
int ptk_search_radius_count_device(const ptk_tree* tree, const float* d_queries,
                                   uint64_t nq, float radius, float e,
                                   uint64_t* d_counts, void* stream);
int ptk_search_radius_fill_device(const ptk_tree* tree, const float* d_queries,
                                  uint64_t nq, float radius, float e,
                                  const uint64_t* d_offsets,
                                  ptk_neighbor* d_out, int sort, void* stream);

/* Convenience: both passes + the scan on the device.  offsets (nq + 1, host)
 * is filled; *out is malloc'ed by the library (free with ptk_free). */
int ptk_search_radius(const ptk_tree* tree, const float* queries, uint64_t nq,
                      float radius, float e, int sort, uint64_t* offsets,
                      ptk_neighbor** out);

/* ---- box search (ragged output) --------------------------------------- */
/* mins / maxs: nb x dim.  Row i lists the indices inside [min_i, max_i]
 * (closed), in reference traversal order. */
int ptk_search_box(const ptk_tree* tree, const float* mins, const float* maxs,
                   uint64_t nb, uint64_t* offsets, int32_t** out);
/* Device forms, asynchronous on `stream`, as ptk_search_radius_*_device: the
 * count pass, a scan of the counts by the caller (d_offsets: nb + 1 entries),
 * the fill pass.  d_mins / d_maxs must not change between the two calls. */
int ptk_search_box_count_device(const ptk_tree* tree, const float* d_mins,
                                const float* d_maxs, uint64_t nb,
                                uint64_t* d_counts, void* stream);
int ptk_search_box_fill_device(const ptk_tree* tree, const float* d_mins,
                               const float* d_maxs, uint64_t nb,
                               const uint64_t* d_offsets, int32_t* d_out,
                               void* stream);

void ptk_free(void* p);

/* ---- the host loop for calls the device search refuses ------------------- */
/* A device search may answer PTK_ERR_UNSUPPORTED for a valid tree (a topological tree deeper than the
 * device stack, a dimension beyond the LDS staging of the kernels).  The reference answers every call with a loop
 * of per-query searches over the rows (_pyco_tree/kd_tree.hpp:117-135, :179-200, :245-268); these entry points ARE
 * that loop, on the handle's flat tree and the caller's points (n_points x dim, as given at creation: a handle keeps
 * them on the device only), threads taking 128 rows at a time.  Same rows, same order, same bits as the device
 * search would give.  They are never entered by ptk_search_* themselves -- a missing device or a HIP error stays an
 * error; a wrapper calls them by name after PTK_ERR_UNSUPPORTED, with a warning (pico_tree_amd.KdTree does; the C++
 * batched members loop their own per-query members).  float32 handles. */
int ptk_host_search_knn(const ptk_tree* tree, const float* points, const float* queries, uint64_t nq, uint32_t k,
                        float e, ptk_neighbor* out);
int ptk_host_search_radius(const ptk_tree* tree, const float* points, const float* queries, uint64_t nq, float radius,
                           float e, int sort, uint64_t* offsets, ptk_neighbor** out); /* *out: ptk_free */
int ptk_host_search_box(const ptk_tree* tree, const float* points, const float* mins, const float* maxs, uint64_t nb,
                        uint64_t* offsets, int32_t** out);                            /* *out: ptk_free */

/* Page-locked host memory for query and result arrays of the host-buffer entry points.  ptk_search_knn copies to
 * and from such arrays directly (no staging through the handle's pinned rings, no first touch of fresh pages per
 * call): a binding that returns a new result array per call -- as _pyco_tree does, def_kd_tree.cpp:73-82 -- keeps a
 * pool of these blocks and hands them out again (pico_tree_amd.KdTree does).  Usable from every device of the node. */
int ptk_host_alloc(uint64_t bytes, void** out);
void ptk_host_free(void* p);
/* Page-locks an array the CALLER owns, in place (hipHostRegister), and releases it again: a query or result array that
 * lives as long as the application -- a ring of scan buffers -- is then moved by ptk_search_* without the staging copy
 * a pageable array needs, exactly like memory from ptk_host_alloc.  The caller must unregister the range before it
 * frees or unmaps it; the library never registers a caller's array on its own (a registration cached by address would
 * outlive the array: an allocator hands the same address out again, and the device would read the old pages). */
int ptk_host_register(void* p, uint64_t bytes);
int ptk_host_unregister(void* p);

/* ---- double precision ---------------------------------------------------- */
/* The reference's kd_tree is generic over the scalar type and its Python module
 * builds KdTree objects over float64 arrays too (dispatch on the array dtype:
 * src/pyco_tree/pico_tree/_pyco_tree/kd_tree.hpp:383-445; neighbor dtype
 * [('index','<i4'),('distance','<f8')], def_core.hpp:17-18).  These entry points
 * are that instantiation: kd_tree<space of double points, metric, int>.  Same
 * meaning, same error behaviour and the same results contract as their float32
 * counterparts above (bit-identical to the reference built over doubles with
 * -ffp-contract=off); the tree file format is the reference's for double
 * scalars (branch records are written as whole structs: 24 bytes). */

/* Layout-identical to pico_tree::neighbor<int, double>: 16 bytes, the distance
 * at offset 8. */
typedef struct ptk_neighbor64 {
  int32_t index;
  int32_t pad_; /* zero */
  double distance;
} ptk_neighbor64;

typedef struct ptk_tree64 ptk_tree64; /* opaque */

int ptk_tree64_create_from_points(const double* points, uint64_t n_points,
                                  uint32_t dim, uint64_t max_leaf_size,
                                  int32_t device, ptk_tree64** out);
int ptk_tree64_create_from_stream(const double* points, uint64_t n_points,
                                  uint32_t dim, const void* stream,
                                  uint64_t stream_bytes, int32_t device,
                                  ptk_tree64** out);
void ptk_tree64_destroy(ptk_tree64* tree);
int ptk_tree64_get_info(const ptk_tree64* tree, ptk_tree_info* info);
/* Any PTK_METRIC_* value.  The two topological metrics -- kd_tree<space of double points, metric_so2 |
 * metric_se2_squared> (metric.hpp:186-257), searched as search_nearest_topological does
 * (internal/kd_tree_search.hpp:115-229) -- need dim 1 / dim 3 and the four bounds per branch: a tree made by
 * ptk_tree64_create_from_points or ptk_tree64_create_from_topological_stream has them, one read from a plain
 * stream does not (PTK_ERR_INVALID). */
int ptk_tree64_set_metric(ptk_tree64* tree, int metric);
int ptk_tree64_serialize(const ptk_tree64* tree, void* buf, uint64_t cap,
                         uint64_t* size);
/* As ptk_tree_serialize_topological / ptk_tree_create_from_topological_stream: the stream kd_tree<topological
 * space of double points, ...>::save writes (kd_tree_branch_double records, kd_tree_node.hpp:52-67: 40 bytes). */
int ptk_tree64_serialize_topological(const ptk_tree64* tree, void* buf,
                                     uint64_t cap, uint64_t* size);
int ptk_tree64_create_from_topological_stream(const double* points,
                                              uint64_t n_points, uint32_t dim,
                                              const void* stream,
                                              uint64_t stream_bytes,
                                              int32_t device, ptk_tree64** out);

/* As ptk_search_knn / ptk_search_knn_device. */
int ptk_search64_knn(const ptk_tree64* tree, const double* queries, uint64_t nq,
                     uint32_t k, double e, ptk_neighbor64* out);
int ptk_search64_knn_device(const ptk_tree64* tree, const double* d_queries,
                            uint64_t nq, uint32_t k, double e,
                            ptk_neighbor64* d_out, void* stream);
/* As ptk_search_radius: *out is malloc'ed by the library (ptk_free).  With
 * sort != 0 rows ascend by distance, equal distances by index. */
int ptk_search64_radius(const ptk_tree64* tree, const double* queries,
                        uint64_t nq, double radius, double e, int sort,
                        uint64_t* offsets, ptk_neighbor64** out);
/* As ptk_search_box. */
int ptk_search64_box(const ptk_tree64* tree, const double* mins,
                     const double* maxs, uint64_t nb, uint64_t* offsets,
                     int32_t** out);

/* ---- randomised kd-forest (approximate k-NN in high dimensions) -------- */
/* Replaces the per-query loop over pico_tree::kd_forest::search_nearest /
 * search_nn (examples/pico_understory/pico_understory/kd_forest.hpp:70-85, as
 * driven by examples/kd_forest/kd_forest.cpp:60-100): forest_size trees over
 * Householder-reflected copies of the points (internal/rkd_tree_hh_data.hpp),
 * best-bin-first search bounded by max_leaves_visited per tree
 * (internal/kd_tree_priority_search.hpp:48-63), one k-list shared by the trees.
 * Differences from the reference, both deliberate (see ptk_forest.hpp): the
 * k-list is de-duplicated by index and distances are measured in the original
 * space; reflection vectors derive from `seed` instead of std::random_device.
 * Results are approximate by design; quality is recall against the exact
 * kd_tree, not bit-identity with the reference. */
typedef struct ptk_forest ptk_forest; /* opaque */

int ptk_forest_create(const float* points, uint64_t n_points, uint32_t dim,
                      uint64_t max_leaf_size, uint32_t forest_size,
                      uint64_t seed, int32_t device, ptk_forest** out);
void ptk_forest_destroy(ptk_forest* forest);

/* Queue entries dropped so far because a query's per-tree queue (1024 nodes) was full: a
 * dropped node is simply never searched (the result stays a valid approximate answer). */
int ptk_forest_get_dropped(const ptk_forest* forest, uint64_t* dropped);

/* The unit reflection vectors in use: forest_size x dim floats. */
int ptk_forest_get_rotations(const ptk_forest* forest, float* out);

/* Host buffers.  out: nq x k, row i ascending; rows with fewer than k distinct
 * points found are padded with {index -1, distance FLT_MAX}.  k <= 64. */
int ptk_forest_search_knn(const ptk_forest* forest, const float* queries,
                          uint64_t nq, uint32_t k, uint64_t max_leaves_visited,
                          ptk_neighbor* out);
/* Device buffers, asynchronous on `stream`. */
int ptk_forest_search_knn_device(const ptk_forest* forest, const float* d_queries,
                                 uint64_t nq, uint32_t k,
                                 uint64_t max_leaves_visited,
                                 ptk_neighbor* d_out, void* stream);

/* ---- several GPUs of one node ------------------------------------------- */
/* One tree replicated on `n_devices` devices of this process (single process, no launcher):
 * SURVEY.md 8(b) "multi-GPU variant takes a device list", 8(e).  What it replaces in the reference
 * is the same OpenMP loop over query rows as ptk_search_knn (_pyco_tree/kd_tree.hpp:117-135) --
 * queries are independent, so a batch is cut into n contiguous row ranges of ceil(nq / n) rows,
 * range r on devices[r]; results land in the caller's row order.
 *   ptk_multi_search_knn / _radius   host buffers: every device moves its own range; no collective.
 *   ptk_multi_search_knn_device      queries and results on devices[0] (BASELINE configs[3]): the
 *                                    ranges and the (index, distance) rows travel over xGMI as
 *                                    grouped ncclSend / ncclRecv pairs; RCCL is loaded on first use
 *                                    (PTK_ERR_DEVICE if it cannot be).  `stream`: a stream of
 *                                    devices[0], or NULL.  Asynchronous like ptk_search_knn_device. */
typedef struct ptk_multi ptk_multi;
int ptk_multi_create_from_points(const float* points, uint64_t n_points, uint32_t dim, uint64_t max_leaf_size,
                                 const int32_t* devices, uint32_t n_devices, ptk_multi** out);
int ptk_multi_create(const ptk_tree_desc* desc, const int32_t* devices, uint32_t n_devices, ptk_multi** out);
void ptk_multi_destroy(ptk_multi* multi);
int ptk_multi_device_count(const ptk_multi* multi);
/* The replica on devices[i] (owned by `multi`): for ptk_tree_set_metric, ptk_profile_*, ... */
int ptk_multi_get_tree(const ptk_multi* multi, uint32_t i, const ptk_tree** tree);
/* ptk_tree_set_metric on every replica.  Handles made by ptk_multi_create_from_points carry the outer
   bounds the topological metrics need; a descriptor (ptk_multi_create) has none, so metric_so2 /
   metric_se2_squared answer PTK_ERR_INVALID there, as ptk_tree_set_metric does. */
int ptk_multi_set_metric(ptk_multi* multi, int metric);
int ptk_multi_search_knn(const ptk_multi* multi, const float* queries, uint64_t nq, uint32_t k, float e,
                         ptk_neighbor* out);
int ptk_multi_search_radius(const ptk_multi* multi, const float* queries, uint64_t nq, float radius, float e, int sort,
                            uint64_t* offsets, ptk_neighbor** out);
int ptk_multi_search_knn_device(ptk_multi* multi, const float* d_queries, uint64_t nq, uint32_t k, float e,
                                ptk_neighbor* d_out, void* stream);

/* ---- measurement ------------------------------------------------------ */
/* When enabled, every internal kernel launch of this handle is bracketed by HIP
 * events recorded on its stream (no host synchronisation at launch time, so it
 * may stay on inside a timed region).  ptk_profile_get waits for the recorded
 * events and returns the accumulated per-kernel elapsed times. */
typedef struct ptk_profile {
  double search_ms;   /* the traversal kernel(s)                               */
  double reorder_ms;  /* Morton keys + sort + permutation                      */
  double other_ms;    /* scans, fills, copies issued by the backend            */
  uint64_t launches;  /* traversal kernel launches accumulated                 */
  uint64_t queries;   /* queries those launches processed                      */
  double search_tail_ms; /* the part of search_ms behind the first traversal kernel of a call: k = 1 -- phase 2,
                            the cooperative search and the replay (search_ms - search_tail_ms = phase 1)   */
} ptk_profile;
int ptk_profile_enable(ptk_tree* tree, int on);
int ptk_profile_get(const ptk_tree* tree, ptk_profile* out, int reset);
/* The same for a caller built against another version of this header: writes the first `size`
 * bytes of the record only (fields are only ever appended to ptk_profile). */
int ptk_profile_get_sized(const ptk_tree* tree, void* out, uint64_t size, int reset);
/* Counters of the last two-phase k = 1 search on the handle's default scratch block (synchronises
 * the device): counts[0] = queries that needed phase 2, [1] = queries phase 2 handed to the
 * cooperative search, [2] = queries that search could not certify (redone by the reference
 * traversal from the root), [3] = queries of the classes dealt across wavefronts. */
int ptk_debug_knn1_counts(const ptk_tree* tree, uint32_t counts[4]);
/* After a k-NN search with 1 < k <= 56 on a 3-D tree (default metric, exact): {queries the general kernel handed to the
 * cooperative search because they had entered more than the cap of far children (ptk_debug_knn_cap), queries that search could not certify
 * and the reference search redid, and why: a pool of subtrees and its spill that overflowed, more equal distances
 * than the second sweep can rank, a box distance above the k-th distance on the way to a neighbour, a k-th distance
 * outside [1e-30, 1e30]; last: queries whose equal distances a second sweep put in the reference's order}.
 * Synchronises the device. */
int ptk_debug_knn_coop_counts(const ptk_tree* tree, uint32_t counts[7]);
/* After a radius count pass on a 3-D tree whose rows are kept as leaf lists: {queries the list pass handed to a
 * wavefront because they had entered more than its cap of far children (ptk_kernels_coopr.hpp), rows that search could
 * not finish and one lane counted again from the root, sorted leaf entries the fill pass will read}.  Zeros when the
 * pass ran uncapped.  Synchronises the device. */
int ptk_debug_radius_coop_counts(const ptk_tree* tree, uint32_t counts[3]);
/* ptk_debug_knn_coop_counts for the double-precision tree: the counters of the last k-NN call that ran capped (exact,
 * dim <= 3, metric_l2_squared / metric_l1, k <= 32, 32 queries or more: ptk_kernels_coop64.hpp; and of the last radius call, which is capped at any size); zeros if none has
 * since the stack block was last re-allocated.  Synchronises the device. */
int ptk_tree64_debug_knn_coop_counts(const ptk_tree64* tree, uint32_t counts[7]);
/* The far children a query of such a search may enter before a wavefront takes it over, for a batch of nq queries
 * (it follows the batch: a capped launch ends with the lanes that ran to their cap; 0 = this search runs uncapped --
 * e != 1, fewer than 32 queries, k outside 2 .. 56), and the entries of the hand-over list of that batch (a query that
 * finds it full goes on in its lane).  No device needed; honours the test hooks knn_cap / knn_cap_min_nq of PTK_TEST_KNOBS. */
int ptk_debug_knn_cap(uint64_t nq, uint32_t k, float e, uint32_t* cap, uint64_t* list_entries);
/* Piles -- subtrees all of whose points are one and the same point (the reference's builder peels one of them off per
 * level, kd_tree_builder.hpp:255-275) -- of this handle's device replica (dim <= 3): out[0] = piles, [1] = points they
 * hold, [2] = depth of the view without them that the k = 1 searches of the default metric traverse (0 piles: the tree
 * itself is searched and [2] is its depth). */
int ptk_debug_piles(const ptk_tree* tree, uint64_t out[3]);
/* The order a batch of nq query rows (device buffer) is searched in: d_perm[i] (device, nq words) = the row that is
 * searched i-th -- a stable sort of the rows by their Morton keys (ptk_debug_key_bits says how the key bits are spread
 * over the axes).  Synchronises the device. */
int ptk_debug_batch_permutation(const ptk_tree* tree, const float* d_queries, uint64_t nq, uint32_t* d_perm);
/* Where the creation of this handle went, in ms: [0] host build of the tree (ptk_tree_create_from_points only),
 * [1] re-encoding for the device and stream checks, [2] upload and the point gather on the device. */
int ptk_debug_create_phases(const ptk_tree* tree, double ms[3]);
/* How a batch of nq queries would be ordered on the device: bits[a] = bits of the Morton key spent on axis a (the
 * first three axes; in proportion to how often a root-to-leaf path of this tree splits on each).  Works on handles
 * without a device replica too. */
int ptk_debug_key_bits(const ptk_tree* tree, uint64_t nq, uint32_t bits[3]);
/* What the last search on the handle's default scratch block did with the order of its batch: 0 = taken as it came
 * (small batch, PTK_REORDER_OFF), 1 = sorted on the device, 2 = sampled, found coherent and searched in the caller's
 * order (k = 1 under PTK_REORDER_AUTO; the reference's loop walks the rows as given, _pyco_tree/kd_tree.hpp:128-134). */
int ptk_debug_batch_order(const ptk_tree* tree, int* how);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* PTK_H_ */
