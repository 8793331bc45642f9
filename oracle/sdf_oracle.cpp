// ORACLE — TEST INFRASTRUCTURE ONLY.  C ABI over oracle/orc_*.h (see sdf_oracle.h).
#include "sdf_oracle.h"
#include "orc_exact.h"
#include "orc_continuity.h"
#include <chrono>
#include <memory>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

struct orc_mesh {
    std::vector<V3> vertices;
    std::vector<uint32_t> indices;
    std::vector<TriangleData> td;
    SphereBvh bvh;
    bool hasBvh = false;
    MeshView view() const { return MeshView{vertices.data(), (uint32_t)vertices.size(), indices.data(), (uint32_t)(indices.size() / 3)}; }
    void ensureBvh() { if (!hasBvh) { bvh.build(vertices.data(), (uint32_t)vertices.size(), indices.data(), (uint32_t)(indices.size() / 3)); hasBvh = true; } }
};
struct orc_octree { OctreeSdfData d; };
struct orc_exact { ExactOctreeData d; uint64_t cullTests = 0; };

static inline V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
static inline Box ldbox(const float* b) { return Box{ld3(b), ld3(b + 3)}; }

extern "C" {

orc_mesh* orc_mesh_create_ex(const float* xyz, uint32_t nv, const uint32_t* idx, uint32_t nt, const float* bbox6) {
    orc_mesh* m = new orc_mesh();
    m->vertices.resize(nv);
    std::memcpy(m->vertices.data(), xyz, sizeof(float) * 3 * (size_t)nv);
    m->indices.assign(idx, idx + 3 * (size_t)nt);
    MeshBox mb; if (bbox6) { mb.min = ld3(bbox6); mb.max = ld3(bbox6 + 3); }
    m->td = meshTriangleData(m->vertices.data(), nv, m->indices.data(), nt, bbox6 ? &mb : nullptr);
    return m;
}
orc_mesh* orc_mesh_create(const float* xyz, uint32_t nv, const uint32_t* idx, uint32_t nt) { return orc_mesh_create_ex(xyz, nv, idx, nt, nullptr); }
void orc_mesh_destroy(orc_mesh* m) { delete m; }
void orc_mesh_triangle_data(orc_mesh* m, float* out) { std::memcpy(out, m->td.data(), m->td.size() * sizeof(TriangleData)); }
double orc_mesh_build_bvh(orc_mesh* m) {
    auto t0 = std::chrono::steady_clock::now();
    m->hasBvh = false; m->ensureBvh();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
uint64_t orc_bvh_num_nodes(orc_mesh* m) { m->ensureBvh(); return m->bvh.nodes.size(); }
void orc_bvh_export(orc_mesh* m, double* sph, int32_t* lr) {
    m->ensureBvh();
    for (size_t i = 0; i < m->bvh.nodes.size(); i++) {
        const BvhNode& n = m->bvh.nodes[i];
        double* s = sph + 8 * i;
        s[0] = n.bvLeft.center.x; s[1] = n.bvLeft.center.y; s[2] = n.bvLeft.center.z; s[3] = n.bvLeft.radius;
        s[4] = n.bvRight.center.x; s[5] = n.bvRight.center.y; s[6] = n.bvRight.center.z; s[7] = n.bvRight.radius;
        lr[2 * i] = n.left; lr[2 * i + 1] = n.right;
    }
}
void orc_bvh_nearest(orc_mesh* m, const float* pts, uint64_t n, uint32_t* ids, double* dist) {
    m->ensureBvh();
    #pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        double d;
        ids[i] = m->bvh.nearestTriangle(ld3(pts + 3 * i), &d);
        if (dist) dist[i] = d;
    }
}

float orc_sqdist_point_triangle(orc_mesh* m, uint32_t t, const float p[3]) { return sqDistPointTriangle(ld3(p), m->td[t]); }
float orc_sqdist_point_triangle_raw(const float p[3], const float a[3], const float b[3], const float c[3]) {
    return sqDistPointTriangleRaw(ld3(p), ld3(a), ld3(b), ld3(c));
}
float orc_signed_dist_point_triangle(orc_mesh* m, uint32_t t, const float p[3]) { return signedDistPointTriangle(ld3(p), m->td[t]); }

// The reference's TriangleDistanceTest as it stands (src/tools/TriangleDistanceTest/main.cpp:12-64): srand(seed), n points of
// 2*rand()/RAND_MAX-1 per coordinate around the fixed triangle; returns the number of points violating either assert and the
// largest deviations seen (the two `total` sums it prints are returned too: they differ between its two distance routines
// only by rounding).
uint32_t orc_triangle_distance_test(unsigned seed, uint32_t n, float* maxRawVsData, float* maxSignedVsData, float* sumRaw, float* sumData) {
    std::srand(seed);
    const V3 v1 = v3(-0.5f, -0.5f, 0.0f), v2 = v3(0.5f, -0.5f, 0.0f), w3 = v3(0.0f, 0.5f, 0.0f);
    const TriangleData td = makeTriangleData(v1, v2, w3);
    auto rnd = []() { return 2.0f * (static_cast<float>(std::rand()) / static_cast<float>(RAND_MAX)) - 1.0f; };
    uint32_t bad = 0; float m1 = 0.f, m2 = 0.f, t1 = 0.f, t2 = 0.f;
    for (uint32_t i = 0; i < n; i++) {
        const float x = rnd(), y = rnd(), z = rnd();
        const V3 s = v3(x, y, z);
        const float a = sqDistPointTriangleRaw(s, v1, v2, w3), b = sqDistPointTriangle(s, td), sg = signedDistPointTriangle(s, td);
        t1 += a; t2 += b;
        const float e1 = std::fabs(a - b), e2 = std::fabs(sg * sg - b);
        m1 = gmax(m1, e1); m2 = gmax(m2, e2);
        if (!(e1 < 0.001f) || !(e2 < 0.001f)) bad++;
    }
    *maxRawVsData = m1; *maxSignedVsData = m2; *sumRaw = t1; *sumData = t2;
    return bad;
}
float orc_signed_dist_point_triangle_grad(orc_mesh* m, uint32_t t, const float p[3], float g[3]) {
    V3 n;
    const float d = signedDistPointTriangleGrad(ld3(p), m->td[t], m->vertices[m->indices[3 * t]], m->vertices[m->indices[3 * t + 1]], m->vertices[m->indices[3 * t + 2]], n);
    g[0] = n.x; g[1] = n.y; g[2] = n.z; return d;
}
float orc_signed_dist_point_triangle_grad_local(orc_mesh* m, uint32_t t, const float p[3], float g[3]) {
    V3 n; const float d = signedDistPointTriangleGradLocal(ld3(p), m->td[t], n);
    g[0] = n.x; g[1] = n.y; g[2] = n.z; return d;
}
void orc_point_values(orc_mesh* m, const float* pts, const uint32_t* tris, uint64_t n, float* out8) {
    const MeshView mv = m->view();
    for (uint64_t i = 0; i < n; i++) pointValues(ld3(pts + 3 * i), tris[i], mv, m->td, out8 + 8 * i);
}

void orc_fit_matrix(int32_t* out) { const FitMatrix& f = fitMatrix(); for (int r = 0; r < 64; r++) for (int c = 0; c < 64; c++) out[64 * r + c] = f.m[r][c]; }
void orc_tricubic_fit(const float* in, float ns, float* out) { tricubicFit(reinterpret_cast<const float(*)[8]>(in), ns, out); }
float orc_tricubic_value(const float* c, const float f[3]) { return tricubicValue(c, ld3(f)); }
float orc_tricubic_value_literal(const float* c, const float f[3]) { return tricubicValueLiteral(c, ld3(f)); }
float orc_tricubic_value_enoki(const float* c, const float f[3]) { return tricubicValueEnoki(c, ld3(f)); }
int orc_interpolation_flavour(void) {
#ifdef ORC_ENOKI_ORDER
    return 1;
#else
    return 0;
#endif
}
void orc_tricubic_gradient(const float* c, const float f[3], float o[3]) { V3 g = tricubicGradient(c, ld3(f)); o[0] = g.x; o[1] = g.y; o[2] = g.z; }
void orc_tricubic_vertex_values(const float* c, const float f[3], float ns, float o[8]) { tricubicVertexValues(c, ld3(f), ns, o); }
float orc_rule_value(int rule, const float* c, const float* mid, float p1) { return ruleValue(rule, c, reinterpret_cast<const float(*)[8]>(mid), p1); }
void orc_stencil(int32_t* cs, float* rel, float* w) {
    const Stencil& s = stencil();
    for (int c = 0; c < 8; c++) for (int j = 0; j < 8; j++) cs[8 * c + j] = s.childSrc[c][j];
    for (int m = 0; m < 19; m++) { rel[3 * m] = s.midRel[m].x; rel[3 * m + 1] = s.midRel[m].y; rel[3 * m + 2] = s.midRel[m].z; w[m] = s.midWeight[m]; }
}

int orc_is_near_minimize(float half, const float radius[8], const float tri[9], float thr, uint32_t* iters) {
    V3 t[3] = {ld3(tri), ld3(tri + 3), ld3(tri + 6)};
    return isNearMinimize(half, radius, t, thr, iters) ? 1 : 0;
}

orc_octree* orc_octree_build(orc_mesh* m, const float box6[6], uint32_t depth, uint32_t startDepth, int rule, float p0, float p1, int cache, int layout) {
    m->ensureBvh();
    orc_octree* o = new orc_octree();
    OctreeBuilder b(m->view(), m->td, m->bvh, o->d);
    b.run(ldbox(box6), depth, startDepth, rule, p0, p1, cache != 0, layout);
    return o;
}
orc_octree* orc_octree_build_continuity(orc_mesh* m, const float box6[6], uint32_t depth, uint32_t startDepth, int rule, float p0, float p1) {
    m->ensureBvh();
    orc_octree* o = new orc_octree();
    ContinuityBuilder b(m->view(), m->td, m->bvh, o->d);
    b.run(ldbox(box6), depth, startDepth, rule, p0, p1);
    computeMinBorder(o->d);
    return o;
}
// the rule-generated stencil tables, for tools/check_ref_expressions.py: 19 x 3 relative positions, 8 x 8 child sources, 19 weights
void orc_stencil_tables(float* midRel57, int32_t* childSrc64, float* midWeight19) {
    const Stencil& st = stencil();
    for (int m = 0; m < 19; m++) { midRel57[3 * m] = st.midRel[m].x; midRel57[3 * m + 1] = st.midRel[m].y; midRel57[3 * m + 2] = st.midRel[m].z; midWeight19[m] = st.midWeight[m]; }
    for (int c = 0; c < 8; c++) for (int j = 0; j < 8; j++) childSrc64[8 * c + j] = st.childSrc[c][j];
}
void orc_neighbour_masks(uint32_t* out24) { const NeighbourMasks& n = neighbourMasks(); for (int i = 0; i < 24; i++) out24[i] = n.m[i]; }
void orc_octree_destroy(orc_octree* o) { delete o; }
uint64_t orc_octree_size(orc_octree* o) { return o->d.data.size(); }
void orc_octree_data(orc_octree* o, uint32_t* out) { std::memcpy(out, o->d.data.data(), o->d.data.size() * 4); }
void orc_octree_info(orc_octree* o, float box6[6], int32_t* g, float* cell, float* vr, float* mb, uint64_t* nq) {
    box6[0] = o->d.box.min.x; box6[1] = o->d.box.min.y; box6[2] = o->d.box.min.z;
    box6[3] = o->d.box.max.x; box6[4] = o->d.box.max.y; box6[5] = o->d.box.max.z;
    *g = o->d.startGridSize; *cell = o->d.startGridCellSize; *vr = o->d.valueRange; *mb = o->d.minBorderValue; *nq = o->d.numBvhQueries;
}
static void queryLoop(const OctreeSdfData& d, const float* pts, uint64_t n, float* dist, float* grad, int nt) {
#ifdef _OPENMP
    if (nt <= 0) nt = omp_get_max_threads();
#endif
    (void)nt;
    #pragma omp parallel for schedule(static) num_threads(nt)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        if (grad) { V3 g{0.f, 0.f, 0.f}; dist[i] = octreeDistance(d, ld3(pts + 3 * i), &g); grad[3 * i] = g.x; grad[3 * i + 1] = g.y; grad[3 * i + 2] = g.z; }
        else dist[i] = octreeDistance(d, ld3(pts + 3 * i));
    }
}
void orc_octree_query(orc_octree* o, const float* pts, uint64_t n, float* dist, float* grad, int nt) { queryLoop(o->d, pts, n, dist, grad, nt); }
void orc_octree_query_raw(const uint32_t* data, uint64_t size, const float box6[6], int32_t g, float minBorder,
                          const float* pts, uint64_t n, float* dist, float* grad, int nt) {
    OctreeSdfData d;
    d.box = ldbox(box6); d.startGridSize = g; d.startGridXY = g * g;
    d.startGridCellSize = d.box.size().x / (float)g;
    d.minBorderValue = minBorder;
    d.data.assign(data, data + size);
    queryLoop(d, pts, n, dist, grad, nt);
}

orc_exact* orc_exact_build(orc_mesh* m, const float box6[6], uint32_t depth, uint32_t startDepth, uint32_t minTri, int cache) {
    return orc_exact_build_mt(m, box6, depth, startDepth, minTri, cache, 1);
}
// threads > 1 (0 = all cores): canonical mode only; start cells built concurrently, arrays identical to the sequential build
orc_exact* orc_exact_build_mt(orc_mesh* m, const float box6[6], uint32_t depth, uint32_t startDepth, uint32_t minTri, int cache, int threads) {
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
    orc_exact* e = new orc_exact();
    ExactBuilder b(m->view(), e->d, &m->td);
    b.run(ldbox(box6), depth, startDepth, minTri, cache != 0, threads);
    e->cullTests = b.cullTests;
    return e;
}
void orc_exact_destroy(orc_exact* e) { delete e; }
void orc_exact_sizes(orc_exact* e, uint64_t* nn, uint64_t* ns, uint64_t* nm, uint32_t* bits, uint32_t* maxLeaf, uint32_t* maxEnc, uint64_t* cull) {
    *nn = e->d.nodes.size() / 2; *ns = e->d.sets.size(); *nm = e->d.masks.size(); *bits = e->d.bitsPerIndex;
    *maxLeaf = e->d.maxTrianglesInLeafs; *maxEnc = e->d.maxTrianglesEncodedInLeafs; *cull = e->cullTests;
}
void orc_exact_data(orc_exact* e, uint32_t* nodes, uint8_t* has, uint32_t* sets, uint8_t* masks) {
    std::memcpy(nodes, e->d.nodes.data(), e->d.nodes.size() * 4);
    for (size_t i = 0; i < e->d.nodeHasTriIdx.size(); i++) has[i] = e->d.nodeHasTriIdx[i] ? 1 : 0;      // internally 1 = set offset, 2 = mask offset
    std::memcpy(sets, e->d.sets.data(), e->d.sets.size() * 4);
    if (!e->d.masks.empty()) std::memcpy(masks, e->d.masks.data(), e->d.masks.size());
}
void orc_exact_query(orc_exact* e, const float* pts, uint64_t n, float* dist, float* grad, uint32_t* tri, int nt) {
#ifdef _OPENMP
    if (nt <= 0) nt = omp_get_max_threads();
#endif
    (void)nt;
    #pragma omp parallel for schedule(static) num_threads(nt)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        V3 g{0.f, 0.f, 0.f}; uint32_t t = 0;
        dist[i] = exactDistance(e->d, ld3(pts + 3 * i), grad ? &g : nullptr, &t);
        if (grad) { grad[3 * i] = g.x; grad[3 * i + 1] = g.y; grad[3 * i + 2] = g.z; }
        if (tri) tri[i] = t;
    }
}

}  // extern "C"
