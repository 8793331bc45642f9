/* sdfhip.h — C ABI of the MI355X-native SDF engine (libsdfhip.so).
 *
 * Drop-in boundary for ONE hot path of UPC-ViRVIG/SdfLib: construction of OctreeSdf / ExactOctreeSdf from a
 * triangle mesh and their getDistance() queries.  Plain pointers and sizes only; no C++/torch types.
 * Every function returns 0 on success or a negative SDFHIP_E_* code; sdfhip_last_error() gives the text of the
 * last failure on this thread.  Nothing here ever falls back to a CPU implementation: without a usable HIP
 * device the context cannot be created and every call fails loudly.
 *
 * Reference interfaces replaced (file:line in /root/reference):
 *   sdfhip_mesh_create        <- sdflib::Mesh(vec3*, n, u32*, n)                 src/utils/Mesh.cpp:34-42
 *   sdfhip_mesh_create_ex     <- sdflib::Mesh(path) + computeBoundingBox()       src/utils/Mesh.cpp:44-62, 90-106
 *                                + TriangleUtils::calculateMeshTriangleData       src/utils/TriangleUtils.cpp:7-428
 *                                + ICG(mesh) (tmd::TriangleMeshDistance BVH)      include/SdfLib/TrianglesInfluence.h:886-924
 *   sdfhip_mesh_nearest       <- ICG::getNearestTriangle                          include/SdfLib/TrianglesInfluence.h:898-905
 *   sdfhip_octree_build*      <- OctreeSdf::OctreeSdf / buildOctree / initOctree  include/SdfLib/OctreeSdf.h:156-172,
 *                                                                                 src/sdf/OctreeSdf.cpp:17-86, src/sdf/OctreeSdfDepthFirst.h:32-558
 *                                (algorithm CONTINUITY: initOctreeWithContinuityNoDelay  src/sdf/OctreeSdfBreadthFirstNoDelay.h:84-1224)
 *   sdfhip_octree_build_shard / _emit_shard  <- the OpenMP loop over start cells + merge/rebase   src/sdf/OctreeSdfDepthFirst.h:433-503
 *   sdfhip_octree_from_data   <- SdfFunction::loadFromFile (OCTREE payload)       src/sdf/SdfFunction.cpp:43-79, include/SdfLib/OctreeSdf.h:222-238
 *   sdfhip_octree_query       <- OctreeSdf::getDistance (both overloads)          src/sdf/OctreeSdf.cpp:93-152
 *   sdfhip_octree_query_grid  <- the per-pixel/lattice loops of the tools         src/tools/SdfError/main.cpp:60-66
 *   sdfhip_octree_download    <- OctreeSdf::getOctreeData / getters               include/SdfLib/OctreeSdf.h:177-219
 *   sdfhip_exact_build        <- ExactOctreeSdf::ExactOctreeSdf / initOctree      src/sdf/ExactOctreeSdf.cpp:7-31,
 *                                                                                 include/SdfLib/ExactOctreeSdfDepthFirst.h:28-681
 *   sdfhip_exact_build_shard / _emit_shard / _from_parts  <- its OpenMP loop over start cells + merge   include/SdfLib/ExactOctreeSdfDepthFirst.h:534-622
 *   sdfhip_exact_from_data    <- SdfFunction::loadFromFile (EXACT_OCTREE payload) include/SdfLib/ExactOctreeSdf.h:138-165
 *   sdfhip_exact_download / _triangle_data  <- getOctreeData / getTrianglesData   include/SdfLib/ExactOctreeSdf.h:99-134
 *   sdfhip_exact_query        <- ExactOctreeSdf::getDistance (both overloads)     src/sdf/ExactOctreeSdf.cpp:38-320
 *   The Unity-style handle API (createOctreeSdf, getDistance, ...) of src/tools/SdfLibUnity/SdfExportFunc.h:16-58
 *   is provided on top of this header by include/SdfLib/SdfExportFunc.h.
 */
#ifndef SDFHIP_H
#define SDFHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDFHIP_OK 0
#define SDFHIP_E_INVALID (-1)     /* bad argument */
#define SDFHIP_E_HIP (-2)         /* HIP runtime error (text in sdfhip_last_error) */
#define SDFHIP_E_NO_DEVICE (-3)   /* no usable gfx950 device: there is no CPU fallback */
#define SDFHIP_E_TOO_LARGE (-4)   /* structure exceeds the 30-bit node index of the reference layout */
#define SDFHIP_E_UNSUPPORTED (-5)
#define SDFHIP_E_HOST (-6)        /* host-side failure (out of memory, thread creation ...): reported, never thrown */

/* where a caller-supplied buffer lives.
 * SDFHIP_HOST:   the call returns when the host buffers hold the results (blocking, like the reference's CPU calls).
 * SDFHIP_DEVICE: the work is ENQUEUED on the context's stream and the call returns at once; inputs must be ready in that
 *                stream's order and outputs are valid in that stream's order (sdfhip_ctx_synchronize, or share the stream:
 *                SDFHIP_STREAM_BORROWED).  Point queries of any size are accepted (processed in chunks). */
#define SDFHIP_HOST 0
#define SDFHIP_DEVICE 1

/* OctreeSdf::TerminationRule (include/SdfLib/OctreeSdf.h:100-106) */
#define SDFHIP_RULE_NONE 0
#define SDFHIP_RULE_TRAPEZOIDAL 1
#define SDFHIP_RULE_SIMPSONS 2
#define SDFHIP_RULE_BY_DISTANCE 3

/* OctreeSdf::InitAlgorithm (include/SdfLib/OctreeSdf.h:23-28) */
#define SDFHIP_ALG_UNIFORM 0        /* not provided (test-only in the reference) */
#define SDFHIP_ALG_NO_CONTINUITY 1
#define SDFHIP_ALG_CONTINUITY 2     /* SdfExporter / Unity default; not sharded (replicas), FIT_EXACT only */

/* node-array layout: which reference branch's array is reproduced */
#define SDFHIP_LAYOUT_GLOBAL_DFS 0  /* numThreads < 2  (src/sdf/OctreeSdfDepthFirst.h:394-416) */
#define SDFHIP_LAYOUT_SUBTREES 1    /* numThreads >= 2 (src/sdf/OctreeSdfDepthFirst.h:417-503): [grid][cell 0 body][cell 1 body]... */

/* query arithmetic */
#define SDFHIP_EVAL_EXACT 0   /* the reference's literal term order, no FMA: bit-identical to the CPU path */
#define SDFHIP_EVAL_FAST 1    /* separable Horner with FMA: <= 1e-5 abs from EXACT, higher throughput */

/* tricubic fit arithmetic during construction */
#define SDFHIP_FIT_EXACT 0    /* scalar, reference summation order (defines the topology) */
#define SDFHIP_FIT_MFMA 1     /* 64x64 fit on v_mfma_f32_32x32x2_f32; borderline decisions re-checked with EXACT */

typedef struct sdfhip_ctx sdfhip_ctx;
typedef struct sdfhip_mesh sdfhip_mesh;
typedef struct sdfhip_octree sdfhip_octree;
typedef struct sdfhip_exact sdfhip_exact;

const char* sdfhip_last_error(void);
/* Which order of the reference's interpolateValue this library computes in (everything else is identical): 0 = the literal order of a
 * SDFLIB_USE_ENOKI=OFF build (include/SdfLib/InterpolationMethods.h:432-439; libsdfhip.so), 1 = the four-wide dot products of a
 * SDFLIB_USE_ENOKI=ON build, the reference's CMake default (:383-430; libsdfhip_enoki.so).  A compile-time option of the reference
 * (CMakeLists.txt:24, 96-100), a link-time choice here: both libraries export the same ABI. */
int sdfhip_interpolation_flavour(void);
const char* sdfhip_version(void);
/* sizeof(sdfhip_octree_info), sizeof(sdfhip_octree_params), sizeof(sdfhip_exact_info): lets a binding verify its struct mirrors */
void sdfhip_abi_sizes(uint64_t out[3]);

/* One context per (process, device).  Thread safety: queries on finished trees may come from any number of host threads at once;
 * builds (mesh preparation, BVH, octrees) issued on one context from several threads are serialised by the context;
 * sdfhip_last_error() is thread-local.
 * stream_mode SDFHIP_STREAM_PRIVATE: the context creates its own non-blocking stream (`stream` ignored).
 * stream_mode SDFHIP_STREAM_BORROWED: run on the caller's hipStream_t `stream` (e.g. torch's current stream);
 *                                     NULL then means the device's default (null) stream. */
#define SDFHIP_STREAM_PRIVATE 0
#define SDFHIP_STREAM_BORROWED 1
int sdfhip_ctx_create(int device_id, void* stream, int stream_mode, sdfhip_ctx** out);
int sdfhip_ctx_destroy(sdfhip_ctx* ctx);
int sdfhip_ctx_synchronize(sdfhip_ctx* ctx);
/* Device memory the context keeps for reuse (transient blocks of past builds in per-stream caches, the nearest search's candidate
 * lists, host-pointer staging buffers).  Kept automatically below a high-water mark (SDFHIP_CACHE_KEEP_MB, default 1/32 of the device memory: ONE mark per device, shared by all of its contexts and streams; applied when a
 * build returns); sdfhip_ctx_trim waits for the context's stream and frees what exceeds keep_bytes (0: everything). */
int sdfhip_ctx_trim(sdfhip_ctx* ctx, uint64_t keep_bytes);
int sdfhip_ctx_cached_bytes(sdfhip_ctx* ctx, uint64_t* out_bytes);
void* sdfhip_ctx_stream(sdfhip_ctx* ctx);

/* Multi-GPU CONTINUITY build: the breadth-first builder couples neighbouring start cells in its serial
 * second iteration (src/sdf/OctreeSdfBreadthFirstNoDelay.h:440-482), so its trees are not built per cell.  What shards is the
 * part that costs the time: the nearest-triangle traversals of a level's sample points.  With an exchange installed every
 * rank runs the whole build, but of each deduplicated sample batch it traverses only its share (128-sample blocks dealt
 * round-robin), writes the triangle ids into a zero-filled buffer obtained from `acquire`, and `all_reduce_sum` makes the
 * buffer complete on every rank (ids of the other ranks' blocks + zeros).  All ranks then hold bit-identical trees.
 * The library synchronises its stream before calling all_reduce_sum; the callee must return only when the result is visible
 * to later work on ANY stream of the device (e.g. RCCL all-reduce followed by a device synchronise).
 * x == NULL (or world < 1) removes the exchange; world == 1 is allowed (the all-reduce is then an identity: used to test a
 * host's callbacks on one device).  Builds with an exchange installed are collective calls: every rank must
 * run the same builds in the same order.  NO_CONTINUITY / Exact builds shard by start cell instead and ignore it. */
typedef struct sdfhip_exchange {
    void* user;
    uint32_t* (*acquire)(void* user, uint64_t count);        /* device buffer of `count` u32, zero-filled; valid until the next acquire */
    int (*all_reduce_sum)(void* user, uint64_t count);      /* in place on the acquired buffer; 0 = ok */
    int32_t rank, world;
} sdfhip_exchange;
int sdfhip_ctx_set_exchange(sdfhip_ctx* ctx, const sdfhip_exchange* x);

/* ---- mesh: vertices (3 floats each), triangle indices (3 u32 each); host pointers, copied ------------- */
int sdfhip_mesh_create(sdfhip_ctx* ctx, const float* xyz, uint32_t num_vertices, const uint32_t* indices,
                       uint32_t num_triangles, sdfhip_mesh** out);
/* Same, with the mesh bounding box [min xyz, max xyz] the reference's file loader computes (src/utils/Mesh.cpp:90-106).
 * A non-NULL box enables the non-manifold seam welding of calculateMeshTriangleData (src/utils/TriangleUtils.cpp:292-420):
 * coincident vertices of single-owner edges are merged (threshold 1e-5 / largest extent) and their edge / vertex
 * pseudonormals summed.  The reference's raw-pointer Mesh constructor leaves the box at (+inf,-inf), i.e. no welding —
 * that is sdfhip_mesh_create. */
int sdfhip_mesh_create_ex(sdfhip_ctx* ctx, const float* xyz, uint32_t num_vertices, const uint32_t* indices,
                          uint32_t num_triangles, const float* bbox6, sdfhip_mesh** out);
/* Same with options.  SDFHIP_MESH_PLAN_BVH_EARLY: the sphere BVH is planned (host threads) WHILE the device prepares the TriangleData, for
 * callers that know an OctreeSdf will be built from the mesh (the C++ OctreeSdf constructor, sdflib_amd.OctreeSdf): sdfhip_mesh_build_bvh
 * — or the first build — then only installs it.  An ExactOctreeSdf never needs the BVH.  Has an effect only under SDFHIP_BVH_BUILD=host: by default
 * the tree is built on the device (bvh.hip::buildTreeOnDevice) and nothing is planned on the host. */
#define SDFHIP_MESH_PLAN_BVH_EARLY 1u
int sdfhip_mesh_create_opt(sdfhip_ctx* ctx, const float* xyz, uint32_t num_vertices, const uint32_t* indices,
                           uint32_t num_triangles, const float* bbox6, uint32_t flags, sdfhip_mesh** out);
int sdfhip_mesh_destroy(sdfhip_mesh* mesh);
/* edges owned by one triangle before welding / half-edges re-paired by the welding (either pointer may be NULL) */
int sdfhip_mesh_edge_stats(sdfhip_mesh* mesh, uint32_t* unmatched_edges, uint32_t* welded_half_edges);
/* 37 floats (148 B) per triangle, field order of TriangleUtils::TriangleData (TriangleUtils.h:56-71) */
int sdfhip_mesh_triangle_data(sdfhip_mesh* mesh, float* out_host);
/* build (host planner, fp64) + upload the bounding-sphere BVH; implicit on first use. seconds may be NULL */
int sdfhip_mesh_build_bvh(sdfhip_mesh* mesh, double* seconds);
/* The planned tree of a mesh with T triangles: 8 doubles (spheres of the two children) and 2 int32 (child references: >= 0 inner node,
 * < 0 ~triangle) per inner node, T - 1 of them in the reference's pre-order (one dummy node when T == 1).  Multi-GPU hosts plan the
 * tree ONCE (the planner is host code that wants all the cores: eight ranks planning the same tree at the same time take 5x longer
 * than one), broadcast the two arrays and import them on the other ranks.  export builds the tree if needed. */
int sdfhip_mesh_bvh_export(sdfhip_mesh* mesh, double* out_spheres, int32_t* out_children, int where);
int sdfhip_mesh_bvh_import(sdfhip_mesh* mesh, const double* spheres, const int32_t* children, int where);
/* nearest triangle id per point (fp64 BVH traversal on the device) */
int sdfhip_mesh_nearest(sdfhip_mesh* mesh, const float* xyz, uint64_t n, uint32_t* out_ids, int where);
/* development probe: per query [triangle id, inner nodes entered, deferred children popped, triangles evaluated] */
int sdfhip_mesh_nearest_stats(sdfhip_mesh* mesh, const float* xyz, uint64_t n, uint32_t* out4);
/* the same counters for a second run in which every query starts from the bound of its own answer: the fewest visits ANY visiting order
 * needs with this tree and these bounds (tools/gpu_near_hist.py) */
int sdfhip_mesh_nearest_stats_preseeded(sdfhip_mesh* mesh, const float* xyz, uint64_t n, uint32_t* out4);
/* Hermite sample [d, gx, gy, gz, 0,0,0,0] at each point for a given triangle id
 * (TriCubicInterpolation::calculatePointValues, InterpolationMethods.h:273-290) */
int sdfhip_mesh_point_values(sdfhip_mesh* mesh, const float* xyz, const uint32_t* tri_ids, uint64_t n, float* out8, int where);

/* ---- OctreeSdf ------------------------------------------------------------------------------------------ */
typedef struct sdfhip_octree_info {
    float box_min[3], box_max[3];   /* cube-ified box (OctreeSdf.cpp:43-46) */
    int32_t start_grid_size;
    uint32_t max_depth;
    float value_range;              /* mValueRange */
    float min_border_value;         /* mMinBorderValue */
    uint64_t num_words;             /* size of the full node array (u32 words) */
    uint64_t num_leaves;
    uint64_t num_nodes;             /* leaves + inner nodes from the start depth down */
    uint64_t num_samples;           /* nearest-triangle samples the reference would issue (8 per root corner set, 19 per evaluated node) */
    /* sharded builds: this shard's part of the array */
    uint32_t cell_begin, cell_end;  /* start-grid cells [begin, end) owned (z-major cell index) */
    uint64_t body_words;            /* words in this shard's bodies */
    uint64_t body_offset;           /* absolute word offset of this shard's bodies in the full array */
    double seconds_samples, seconds_decide, seconds_total;
    uint64_t leaves_per_depth[16];  /* leaves at each depth (this shard) */
    uint64_t fit_rechecks;          /* FIT_MFMA: nodes whose decision was re-evaluated with the reference-ordered fit */
    uint64_t num_traversals;        /* BVH traversals actually run (samples sharing a lattice point AND position bits share one) */
    uint64_t post_pass_scheduled;   /* CONTINUITY: leaves Iter 2 scheduled for re-subdivision (OctreeSdfBreadthFirstNoDelay.h:506-512) */
    float start_grid_cell_size;     /* mStartGridCellSize: a BUILT tree takes it from the input box's largest extent (OctreeSdf.cpp:43-52),
                                       a LOADED one from the stored box (OctreeSdf.h:233); far from the origin the two differ in the last bit */
    float reserved0;
    uint64_t num_nearest_fallbacks; /* of num_traversals: queries the two-phase nearest search (fp32 candidates + exact tie replay) handed to the
                                       order-exact fp64 traversal because it could not decide them with certainty */
    /* the candidate search of those traversals (SURVEY.md 8(d) "B": the build's dominant kernel), counted and timed by the build itself */
    uint64_t near_expansions;       /* 4-wide BVH nodes expanded (128-byte records fetched) */
    uint64_t near_triangle_tests;   /* fp32 point/triangle evaluations (48-byte records fetched) */
    double seconds_near_candidates; /* device time of the candidate kernel's launches (HIP events on the build's stream) */
    double seconds_near_search;     /* ... of the whole search: candidates + long queries + exact resolution + fallback */
} sdfhip_octree_info;

typedef struct sdfhip_octree_params {
    float box_min[3], box_max[3];
    uint32_t depth, start_depth;
    int32_t rule;                   /* SDFHIP_RULE_* */
    float rule_params[2];           /* [0] = expected error (threshold), [1] = decay for BY_DISTANCE */
    int32_t algorithm;              /* SDFHIP_ALG_* */
    int32_t layout;                 /* SDFHIP_LAYOUT_* */
    int32_t fit_mode;               /* SDFHIP_FIT_* */
    uint32_t cell_begin, cell_end;  /* shard: build only the subtrees of these start-grid cells; 0,0 = all */
} sdfhip_octree_params;

/* Whole build on one device. */
int sdfhip_octree_build(sdfhip_ctx* ctx, sdfhip_mesh* mesh, const sdfhip_octree_params* params, sdfhip_octree** out);

/* Sharded build (one process per GPU):
 *   1. every rank:  sdfhip_octree_build_shard(...)           -> info.body_words for its cells
 *   2. host:        exchange body_words (all-gather), prefix-sum -> body_offset of every rank
 *   3. every rank:  sdfhip_octree_emit_shard(tree, body_offset, dst_grid, dst_body)
 *                   writes its start-grid words and its bodies with ABSOLUTE indices
 *   4. host:        RCCL all-gather of the bodies / grid slices, then sdfhip_octree_from_data on the result. */
int sdfhip_octree_build_shard(sdfhip_ctx* ctx, sdfhip_mesh* mesh, const sdfhip_octree_params* params, sdfhip_octree** out);
int sdfhip_octree_emit_shard(sdfhip_octree* tree, uint64_t body_offset, uint32_t* dst_grid_words /* cell_end-cell_begin */,
                             uint32_t* dst_body_words /* body_words */, int where);

/* Wrap an existing node array (all-gathered shards, or a file) for queries; data is copied to the device. */
int sdfhip_octree_from_data(sdfhip_ctx* ctx, const uint32_t* words, uint64_t num_words, int where, const float box_min[3],
                            const float box_max[3], int32_t start_grid_size, uint32_t max_depth, float value_range,
                            float min_border_value, sdfhip_octree** out);
/* A tree reassembled from the shards of a BUILD answers like the built tree only with the build's cell size (info.start_grid_cell_size
 * of any shard); sdfhip_octree_from_data alone gives it the loaded tree's. */
int sdfhip_octree_set_start_grid_cell_size(sdfhip_octree* tree, float cell_size);
int sdfhip_octree_destroy(sdfhip_octree* tree);
int sdfhip_octree_get_info(sdfhip_octree* tree, sdfhip_octree_info* out);
/* copy the node array (getOctreeData(): u32 words, leaf bit31, 64 float coefficients per leaf) */
int sdfhip_octree_download(sdfhip_octree* tree, uint32_t* out_words, int where);
/* device address of the node array (rebuilt first if the tree was compacted).  Valid until the tree is destroyed or
 * sdfhip_octree_compact is called on it; the automatic compaction of large trees leaves an array alone once its address was handed out. */
const uint32_t* sdfhip_octree_device_words(sdfhip_octree* tree);
/* Device footprint.  Queries run on a packed layout of the tree (node words breadth-first + 256-byte-aligned coefficient blocks) that is
 * made from the node array on the first query and holds everything the array holds; sdfhip_octree_compact makes it and RELEASES the
 * array (half the footprint); download / device_words rebuild it from the layout, bit for bit, when asked.  Arrays of
 * SDFHIP_COMPACT_ABOVE_MB (default 1024) and more are compacted automatically by their first query.  Not applied to an array with
 * words that belong to no node or coefficient block (it keeps its array). */
int sdfhip_octree_compact(sdfhip_octree* tree);
int sdfhip_octree_device_bytes(sdfhip_octree* tree, uint64_t* out_bytes);

/* batched getDistance: xyz = n points (3 floats each); out_grad may be NULL (normalised gradient otherwise) */
int sdfhip_octree_query(sdfhip_octree* tree, const float* xyz, uint64_t n, float* out_dist, float* out_grad, int where, int eval_mode);
/* lattice of nx*ny*nz points origin + (i,j,k)*step, x fastest; outputs nx*ny*nz floats (+3x for the gradient) */
int sdfhip_octree_query_grid(sdfhip_octree* tree, const float origin[3], const float step[3], uint32_t nx, uint32_t ny, uint32_t nz,
                             float* out_dist, float* out_grad, int where, int eval_mode);

/* ---- ExactOctreeSdf -------------------------------------------------------------------------------------- */
typedef struct sdfhip_exact_info {
    float box_min[3], box_max[3];
    int32_t start_grid_size;
    uint32_t start_depth, max_depth, bit_encoding_start_depth, bits_per_index;
    uint32_t min_triangles_in_leafs, max_triangles_in_leafs, max_triangles_encoded_in_leafs;
    uint64_t num_nodes, num_set_words, num_mask_bytes, num_triangles;
    uint64_t cull_tests;            /* IsNearMinimize evaluations during the build */
    double seconds_total;
    float start_grid_cell_size;     /* as in sdfhip_octree_info; sdfhip_exact_from_parts uses it when > 0, sdfhip_exact_from_data never */
    float reserved0;
} sdfhip_exact_info;

int sdfhip_exact_build(sdfhip_ctx* ctx, sdfhip_mesh* mesh, const float box_min[3], const float box_max[3], uint32_t max_depth,
                       uint32_t start_depth, uint32_t min_triangles_per_node, sdfhip_exact** out);
/* Wrap existing arrays (e.g. read from a .bin file) for queries; everything is copied to the device.
 * nodes: 2 u32 per node; triangle_data: 37 floats per triangle. */
int sdfhip_exact_from_data(sdfhip_ctx* ctx, const sdfhip_exact_info* info, const uint32_t* nodes, const uint32_t* sets, const uint8_t* masks,
                           const float* triangle_data, sdfhip_exact** out);
/* Multi-GPU construction (one process per GPU), same decomposition as the reference's OpenMP loop over start cells
 * (include/SdfLib/ExactOctreeSdfDepthFirst.h:534-622), with the three arrays laid out as its single-thread build does:
 *   1. build_shard: levels above start_depth are computed by every rank; from start_depth on only the start cells whose position
 *      in the reference's emission order (children 7..0 at every level) lies in [rank_begin, rank_end).  get_info then reports the
 *      shard's LOCAL sizes: num_nodes = body nodes (start-grid slots not counted), num_set_words, num_mask_bytes.
 *   2. host: exclusive prefix sums over ranks -> node_offset (= 8^start_depth + lower ranks' body nodes), set_offset, mask_offset.
 *   3. emit_shard: writes the shard's start-grid slots (2 u32 per cell, in the order of shard_cells), body nodes, sets, masks with
 *      ABSOLUTE indices.  4. host: all-gather, scatter the grid slots by cell id, concatenate the rest, sdfhip_exact_from_parts. */
int sdfhip_exact_build_shard(sdfhip_ctx* ctx, sdfhip_mesh* mesh, const float box_min[3], const float box_max[3], uint32_t max_depth,
                             uint32_t start_depth, uint32_t min_triangles_per_node, uint32_t rank_begin, uint32_t rank_end, sdfhip_exact** out);
/* z-major ids (x fastest) of the shard's start cells, ascending; rank_end - rank_begin entries */
int sdfhip_exact_shard_cells(sdfhip_exact* shard, uint32_t* out_cells);
int sdfhip_exact_emit_shard(sdfhip_exact* shard, uint64_t node_offset, uint64_t set_offset, uint64_t mask_offset, uint32_t* dst_grid_nodes,
                            uint8_t* dst_grid_has, uint32_t* dst_body_nodes, uint8_t* dst_body_has, uint32_t* dst_sets, uint8_t* dst_masks, int where);
/* Wrap assembled arrays for queries; TriangleData stays in `mesh` (which must outlive the tree).  has may be NULL (all 1). */
int sdfhip_exact_from_parts(sdfhip_ctx* ctx, sdfhip_mesh* mesh, const sdfhip_exact_info* info, const uint32_t* nodes, const uint8_t* node_has_tri_idx,
                            const uint32_t* sets, const uint8_t* masks, int where, sdfhip_exact** out);
/* A copy of a BUILT tree made through sdfhip_exact_from_data answers like the original only with the build's cell size
 * (info.start_grid_cell_size; see sdfhip_octree_set_start_grid_cell_size); before the first batched query. */
int sdfhip_exact_set_start_grid_cell_size(sdfhip_exact* tree, float cell_size);
int sdfhip_exact_destroy(sdfhip_exact* tree);
int sdfhip_exact_get_info(sdfhip_exact* tree, sdfhip_exact_info* out);
/* nodes: 2 u32 per node {childrenIndex, trianglesArrayIndex}; node_has_tri_idx: 1 where the reference writes
 * trianglesArrayIndex (it leaves the others uninitialised; here they are 0) */
int sdfhip_exact_download(sdfhip_exact* tree, uint32_t* nodes, uint8_t* node_has_tri_idx, uint32_t* sets, uint8_t* masks);
/* the TriangleData array the tree queries against (ExactOctreeSdf::getTrianglesData, ExactOctreeSdf.h:132); 37 floats each */
int sdfhip_exact_triangle_data(sdfhip_exact* tree, float* out_host);
/* ExactOctreeSdf::getDistance x2 (src/sdf/ExactOctreeSdf.cpp:38-320), batched.  The first batch of 16 384 points or more makes the tree's query
 * tables (once, 1.4 ms at C3): per node a query can end in, its set / mask offsets and the DECODED list of the triangles that survive the
 * two mask levels (4 bytes per surviving entry — 0.23 GB at C3, beside 63 MB of nodes / sets / masks).  Trees whose lists would exceed
 * SDFHIP_EXACT_LISTS_MB (default 4096) (or half of the device's free memory, or whose allocation fails) decode per batch instead.  Same results either way. */
int sdfhip_exact_query(sdfhip_exact* tree, const float* xyz, uint64_t n, float* out_dist, float* out_grad /* nullable */,
                       uint32_t* out_triangle /* nullable */, int where);

/* ---- several GPUs in ONE process (C / C++ callers; the multi-PROCESS flavour is sdflib_amd/distributed.py) -------------------------
 * Reference decomposition: the OpenMP loops over start cells + merge of OctreeSdf (src/sdf/OctreeSdfDepthFirst.h:433-503) and of
 * ExactOctreeSdf (include/SdfLib/ExactOctreeSdfDepthFirst.h:534-622); what SdfExporter gets with --num_threads (src/tools/SdfExporter/main.cpp:143-171).
 * sdfhip_multi_create makes one context per device id and, when the ids are distinct, an RCCL communicator over them (librccl.so is bound
 * at run time).  A build shards the start cells over the devices (one host thread each), emits every shard at its absolute offsets and
 * reassembles the array(s) on EVERY device with one in-place all-gather-v (ncclBroadcast per shard inside one group, over xGMI).  The same
 * device id listed several times (one-GPU test boxes) selects device-to-device copies instead of RCCL.  Outputs: out_trees[rank] (and
 * out_meshes[rank]) live on device_ids[rank]; all trees are identical and identical to the single-device build.
 * NO_CONTINUITY and ExactOctreeSdf builds are sharded by start cell; a CONTINUITY tree (not separable by start cell) is built by every device
 * with the BVH traversals of each sample batch shared out and completed by one sum all-reduce per batch (ncclAllReduce; staged copies on
 * the copy transport), as sdflib_amd/distributed.py does across processes: identical trees everywhere, nothing broadcast afterwards. */
typedef struct sdfhip_multi sdfhip_multi;
typedef struct sdfhip_multi_stats {
    int32_t ranks, uses_rccl;
    uint64_t bytes_exchanged;                                   /* bytes every device received in the last build's all-gather */
    double seconds_bvh, seconds_shards, seconds_exchange;       /* of the last build: BVH plan + install, slowest shard, exchange + wrap */
} sdfhip_multi_stats;
int sdfhip_multi_create(const int* device_ids, int n, sdfhip_multi** out);
int sdfhip_multi_destroy(sdfhip_multi* multi);
int sdfhip_multi_size(sdfhip_multi* multi);
sdfhip_ctx* sdfhip_multi_ctx(sdfhip_multi* multi, int rank);
const char* sdfhip_multi_transport(sdfhip_multi* multi);       /* "rccl" or "copy" */
int sdfhip_multi_get_stats(sdfhip_multi* multi, sdfhip_multi_stats* out);
/* bbox6: the loader's bounding box (enables seam welding, see sdfhip_mesh_create_ex) or NULL.  out_meshes may be NULL (the meshes are then freed). */
int sdfhip_multi_octree_build(sdfhip_multi* multi, const float* xyz, uint32_t num_vertices, const uint32_t* indices, uint32_t num_triangles, const float* bbox6,
                              const sdfhip_octree_params* params, sdfhip_mesh** out_meshes, sdfhip_octree** out_trees);
/* out_meshes is required: an ExactOctreeSdf reads its mesh's TriangleData (free trees first, then meshes). */
int sdfhip_multi_exact_build(sdfhip_multi* multi, const float* xyz, uint32_t num_vertices, const uint32_t* indices, uint32_t num_triangles, const float* bbox6,
                             const float box_min[3], const float box_max[3], uint32_t max_depth, uint32_t start_depth, uint32_t min_triangles_per_node,
                             sdfhip_mesh** out_meshes, sdfhip_exact** out_trees);

/* ---- building blocks exposed for parity tests (device execution, host pointers) --------------------------- */
int sdfhip_tricubic_fit(sdfhip_ctx* ctx, const float* values_8x8, const float* node_sizes, uint64_t n, float* out64, int fit_mode);
int sdfhip_is_near_minimize(sdfhip_ctx* ctx, const float* half, const float* radius8, const float* tri9, const float* thr,
                            uint64_t n, uint8_t* out);
/* (test / calibration hooks: include/sdfhip_test.h) */

#ifdef __cplusplus
}
#endif
#endif /* SDFHIP_H */
