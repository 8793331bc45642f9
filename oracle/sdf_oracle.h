/* ORACLE — TEST INFRASTRUCTURE ONLY.
 * C ABI over the CPU restatement of SdfLib's OctreeSdf / ExactOctreeSdf hot path (oracle/orc_*.h).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (sdflib_amd/, include/sdfhip.h) never does.  PARITY UNPINNED at the ulp level: see orc_math.h. */
#ifndef SDF_ORACLE_H
#define SDF_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_mesh orc_mesh;
typedef struct orc_octree orc_octree;
typedef struct orc_exact orc_exact;

orc_mesh* orc_mesh_create(const float* xyz, uint32_t num_vertices, const uint32_t* indices, uint32_t num_triangles);
/* bbox6 (nullable) = the mesh bounding box the reference's file loader computes; enables the non-manifold seam welding */
orc_mesh* orc_mesh_create_ex(const float* xyz, uint32_t num_vertices, const uint32_t* indices, uint32_t num_triangles, const float* bbox6);
void orc_mesh_destroy(orc_mesh*);
void orc_mesh_triangle_data(orc_mesh*, float* out /* 37 floats per triangle */);
double orc_mesh_build_bvh(orc_mesh*);  /* returns seconds */
uint64_t orc_bvh_num_nodes(orc_mesh*);
void orc_bvh_export(orc_mesh*, double* spheres /* 8 per node: left(cx,cy,cz,r) right(cx,cy,cz,r) */, int32_t* left_right /* 2 per node */);
void orc_bvh_nearest(orc_mesh*, const float* pts, uint64_t n, uint32_t* out_ids, double* out_dist /* nullable */);

float orc_sqdist_point_triangle(orc_mesh*, uint32_t tri, const float p[3]);
float orc_sqdist_point_triangle_raw(const float p[3], const float a[3], const float b[3], const float c[3]);
float orc_signed_dist_point_triangle(orc_mesh*, uint32_t tri, const float p[3]);
uint32_t orc_triangle_distance_test(unsigned seed, uint32_t n, float* max_raw_vs_data, float* max_signed_vs_data, float* sum_raw, float* sum_data);
float orc_signed_dist_point_triangle_grad(orc_mesh*, uint32_t tri, const float p[3], float out_grad[3]);
float orc_signed_dist_point_triangle_grad_local(orc_mesh*, uint32_t tri, const float p[3], float out_grad[3]);
void orc_point_values(orc_mesh*, const float* pts, const uint32_t* tris, uint64_t n, float* out8);

void orc_fit_matrix(int32_t* out_64x64);
void orc_tricubic_fit(const float* in_8x8, float node_size, float* out64);
float orc_tricubic_value(const float* c64, const float frac[3]);            /* the library's flavour (see orc_interpolation_flavour) */
float orc_tricubic_value_literal(const float* c64, const float frac[3]);    /* InterpolationMethods.h:432-439 */
float orc_tricubic_value_enoki(const float* c64, const float frac[3]);      /* InterpolationMethods.h:383-430 */
int orc_interpolation_flavour(void);                                        /* 0: SDFLIB_USE_ENOKI=OFF order (libsdf_oracle.so), 1: =ON order (libsdf_oracle_enoki.so) */
void orc_tricubic_gradient(const float* c64, const float frac[3], float out[3]);
void orc_tricubic_vertex_values(const float* c64, const float frac[3], float node_size, float out8[8]);
float orc_rule_value(int rule, const float* c64, const float* mid_19x8, float param1);
void orc_stencil(int32_t* child_src_8x8, float* mid_rel_19x3, float* mid_weight_19);

int orc_is_near_minimize(float half, const float radius[8], const float tri[9], float thr, uint32_t* iters);

/* OctreeSdf (NO_CONTINUITY).  box6 = min xyz, max xyz.  layout: 0 = numThreads<2 array, 1 = numThreads>=2 array. */
orc_octree* orc_octree_build(orc_mesh*, const float box6[6], uint32_t depth, uint32_t start_depth, int rule,
                             float p0, float p1, int vertex_cache, int layout);
/* OctreeSdf CONTINUITY builder (canonical mode, single layout) */
/* Iter 1 of every level runs under OpenMP like the reference's (OctreeSdfBreadthFirstNoDelay.h:246-370); canonical mode is thread-count invariant */
orc_octree* orc_octree_build_continuity(orc_mesh*, const float box6[6], uint32_t depth, uint32_t start_depth, int rule, float p0, float p1);
void orc_stencil_tables(float* mid_rel_57, int32_t* child_src_64, float* mid_weight_19);
void orc_neighbour_masks(uint32_t* out24);
void orc_octree_destroy(orc_octree*);
uint64_t orc_octree_size(orc_octree*);
void orc_octree_data(orc_octree*, uint32_t* out);
void orc_octree_info(orc_octree*, float box6[6], int32_t* start_grid_size, float* cell_size, float* value_range,
                     float* min_border, uint64_t* num_bvh_queries);
void orc_octree_query(orc_octree*, const float* pts, uint64_t n, float* out_dist, float* out_grad /* nullable */, int num_threads);
/* Query an externally supplied node array with the oracle's getDistance (used to cross-check device-built trees). */
void orc_octree_query_raw(const uint32_t* data, uint64_t size, const float box6[6], int32_t start_grid_size, float min_border,
                          const float* pts, uint64_t n, float* out_dist, float* out_grad, int num_threads);

orc_exact* orc_exact_build(orc_mesh*, const float box6[6], uint32_t depth, uint32_t start_depth, uint32_t min_triangles, int vertex_cache);
/* the same arrays from `threads` OpenMP threads over the start cells (0 = all cores); canonical mode (vertex_cache = 0) only, otherwise sequential */
orc_exact* orc_exact_build_mt(orc_mesh*, const float box6[6], uint32_t depth, uint32_t start_depth, uint32_t min_triangles, int vertex_cache, int threads);
void orc_exact_destroy(orc_exact*);
void orc_exact_sizes(orc_exact*, uint64_t* num_nodes, uint64_t* num_set_words, uint64_t* num_mask_bytes, uint32_t* bits_per_index,
                     uint32_t* max_tri_in_leafs, uint32_t* max_tri_encoded, uint64_t* cull_tests);
void orc_exact_data(orc_exact*, uint32_t* nodes /* 2 per node */, uint8_t* node_has_tri_idx, uint32_t* sets, uint8_t* masks);
void orc_exact_query(orc_exact*, const float* pts, uint64_t n, float* out_dist, float* out_grad /* nullable */, uint32_t* out_tri /* nullable */, int num_threads);

#ifdef __cplusplus
}
#endif
#endif
