// Bounding-sphere BVH: host planner (fp64, identical tree to the reference's) + device nearest-triangle kernels.
// PRODUCT code — independent of oracle/.
//
// Host planner reproduces tmd::TriangleMeshDistance::_build_tree (reference libs/InteractiveComputerGraphics/
// InteractiveComputerGraphics/TriangleMeshDistance.h:421-490): median split of the triangle range after a
// std::sort by the FIRST vertex's coordinate on the widest AABB axis; node centre = mean of the range's vertices
// accumulated in range order; radius = max distance to them; leaves hold one triangle.  The reference's array
// is filled in DFS pre-order, so a subtree over n triangles occupies exactly 2n-1 consecutive slots: node ids are
// known up front and the two halves of a range can be planned by different host threads without changing a bit
// of the result.  The sort works on {key, triangle} pairs instead of the reference's 80-byte structs: std::sort's
// permutation depends only on comparison outcomes, which are the same.
#include "sdfhip_internal.h"
#include "dev_bvh.h"
#include <algorithm>
#include <limits>
#include <thread>
#include <cmath>
#include <cstring>

namespace sdfhip {

struct HostBvhBuilder {
    const float* verts; const uint32_t* idx;
    double* sph;       // 8 doubles per inner node: spheres of the left and of the right child
    int* kids;         // 2 ints per inner node: child references (>= 0 inner node, < 0 ~triangle)
    std::vector<int> order;
    int maxParallelDepth = 0;

    struct D { double x, y, z; };
    D vtx(int t, int k) const { const uint32_t v = idx[3 * (size_t)t + k]; return D{(double)verts[3 * v], (double)verts[3 * v + 1], (double)verts[3 * v + 2]}; }
    static double comp(const D& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

    // Plans the subtree over order[begin,end): writes its bounding sphere into out[0..3] and returns the reference to it.
    // `innerId` = pre-order index this subtree's root gets if it is an inner node (n > 1).
    int build(int innerId, double* out, int begin, int end, int depth) {
        const int n = end - begin;
        if (n == 1) {
            const int t = order[begin];
            const D a = vtx(t, 0), b = vtx(t, 1), c = vtx(t, 2);
            const D s = D{(a.x + b.x) + c.x, (a.y + b.y) + c.y, (a.z + b.z) + c.z};
            const D ce = D{s.x / 3.0, s.y / 3.0, s.z / 3.0};
            auto dist = [&](const D& p) { const double dx = p.x - ce.x, dy = p.y - ce.y, dz = p.z - ce.z; return std::sqrt(dx * dx + dy * dy + dz * dz); };
            out[0] = ce.x; out[1] = ce.y; out[2] = ce.z;
            out[3] = std::max(std::max(dist(a), dist(b)), dist(c));
            return ~t;
        }
        const double lo = std::numeric_limits<double>::lowest(), hi = std::numeric_limits<double>::max();
        D top{lo, lo, lo}, bot{hi, hi, hi}, ce{0, 0, 0};
        for (int i = begin; i < end; i++)
            for (int k = 0; k < 3; k++) {
                const D p = vtx(order[i], k);
                ce.x += p.x; ce.y += p.y; ce.z += p.z;
                top.x = std::max(top.x, p.x); bot.x = std::min(bot.x, p.x);
                top.y = std::max(top.y, p.y); bot.y = std::min(bot.y, p.y);
                top.z = std::max(top.z, p.z); bot.z = std::min(bot.z, p.z);
            }
        const double cnt = (double)(3 * n);
        ce.x /= cnt; ce.y /= cnt; ce.z /= cnt;
        const double diag[3] = {top.x - bot.x, top.y - bot.y, top.z - bot.z};
        const int dim = (int)(std::max_element(diag, diag + 3) - diag);
        double r2 = 0.0;
        for (int i = begin; i < end; i++)
            for (int k = 0; k < 3; k++) {
                const D p = vtx(order[i], k);
                const double dx = ce.x - p.x, dy = ce.y - p.y, dz = ce.z - p.z;
                r2 = std::max(r2, dx * dx + dy * dy + dz * dz);
            }
        out[0] = ce.x; out[1] = ce.y; out[2] = ce.z; out[3] = std::sqrt(r2);

        {   // median split: sort the range by the first vertex's coordinate along `dim`
            struct KeyTri { double key; int tri; };
            std::vector<KeyTri> tmp((size_t)n);
            for (int i = 0; i < n; i++) { const int t = order[begin + i]; tmp[i] = KeyTri{comp(vtx(t, 0), dim), t}; }
            std::sort(tmp.begin(), tmp.end(), [](const KeyTri& a, const KeyTri& b) { return a.key < b.key; });
            for (int i = 0; i < n; i++) order[begin + i] = tmp[i].tri;
        }
        const int mid = (int)(0.5 * (begin + end));
        // pre-order numbering of inner nodes: the left subtree holds (mid - begin) - 1 of them
        const int leftId = innerId + 1, rightId = innerId + (mid - begin);
        double* nd = sph + 8 * (size_t)innerId;
        int refs[2];
        if (depth < maxParallelDepth && n > 8192) {
            std::thread th([&]() { refs[0] = build(leftId, nd, begin, mid, depth + 1); });
            refs[1] = build(rightId, nd + 4, mid, end, depth + 1);
            th.join();
        } else {
            refs[0] = build(leftId, nd, begin, mid, depth + 1);
            refs[1] = build(rightId, nd + 4, mid, end, depth + 1);
        }
        kids[2 * (size_t)innerId] = refs[0]; kids[2 * (size_t)innerId + 1] = refs[1];
        return innerId;
    }
};

__global__ void k_tri_verts(const float* __restrict__ verts, const uint32_t* __restrict__ idx, uint32_t numTriangles, float* __restrict__ triV) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = gid / 12u, k = gid - 12u * t;
    if (t >= numTriangles) return;
    triV[gid] = (k < 9u) ? verts[3 * (size_t)idx[3 * (size_t)t + k / 3u] + (k % 3u)] : 0.f;
}

__global__ void __launch_bounds__(128) k_nearest(BvhDev bvh, const float* __restrict__ pts, uint64_t n, uint32_t* __restrict__ out) {
    __shared__ uint32_t s_stack[BVH_STACK * 128];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = bvhNearest<128>(bvh, F3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, s_stack + threadIdx.x);
}

// dev probe: traversal statistics per query
__global__ void __launch_bounds__(128) k_nearest_stats(BvhDev bvh, const float* __restrict__ pts, uint64_t n, uint32_t* __restrict__ out4) {
    __shared__ uint32_t s_stack[BVH_STACK * 128];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c[3] = {0, 0, 0};
    out4[4 * i] = bvhNearest<128, true>(bvh, F3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, s_stack + threadIdx.x, c);
    out4[4 * i + 1] = c[0]; out4[4 * i + 2] = c[1]; out4[4 * i + 3] = c[2];
}

__global__ void k_point_values(const float* __restrict__ verts, const uint32_t* __restrict__ idx, const float* __restrict__ td,
                               const float* __restrict__ pts, const uint32_t* __restrict__ tris, uint64_t n, float* __restrict__ out8) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = tris[i];
    const uint32_t a = idx[3 * t], b = idx[3 * t + 1], c = idx[3 * t + 2];
    F3 g;
    const float d = signedDistPointTriangleGrad(F3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, td + (size_t)TD_FLOATS * t,
                                                F3{verts[3 * a], verts[3 * a + 1], verts[3 * a + 2]}, F3{verts[3 * b], verts[3 * b + 1], verts[3 * b + 2]},
                                                F3{verts[3 * c], verts[3 * c + 1], verts[3 * c + 2]}, g);
    float* o = out8 + 8 * i;
    o[0] = d; o[1] = g.x; o[2] = g.y; o[3] = g.z; o[4] = 0.f; o[5] = 0.f; o[6] = 0.f; o[7] = 0.f;
}

}  // namespace sdfhip

using namespace sdfhip;

int sdfhip_mesh_ensure_bvh(sdfhip_mesh* mesh) {
    if (mesh->hasBvh) return SDFHIP_OK;
    return sdfhip_mesh_build_bvh(mesh, nullptr);
}

extern "C" {

int sdfhip_mesh_build_bvh(sdfhip_mesh* mesh, double* seconds) {
    SDF_REQUIRE(mesh != nullptr, "mesh is NULL");
    { int depth = 1; while ((1ull << (depth - 1)) < mesh->numTriangles) depth++; SDF_REQUIRE(depth + 1 <= BVH_STACK, "mesh too large for the traversal stack"); }
    const double t0 = nowSeconds();
    const uint32_t T = mesh->numTriangles;
    const uint64_t nn = T - 1;                                   // inner nodes
    std::vector<double> sph(8 * (size_t)(nn ? nn : 1), 0.0);
    std::vector<int> kids(2 * (size_t)(nn ? nn : 1), 0);
    HostBvhBuilder b;
    b.verts = mesh->hVerts.data(); b.idx = mesh->hIdx.data(); b.sph = sph.data(); b.kids = kids.data();
    b.order.resize(T);
    for (uint32_t i = 0; i < T; i++) b.order[i] = (int)i;
    unsigned hc = std::thread::hardware_concurrency();
    int pd = 0; while ((1u << pd) < (hc ? hc : 1u) && pd < 6) pd++;
    b.maxParallelDepth = pd;
    double rootSphere[4];
    b.build(0, rootSphere, 0, (int)T, 0);
    SDF_HIP_CHECK(hipSetDevice(mesh->ctx->device));
    hipStream_t st = mesh->ctx->stream;
    SDF_TRY(mesh->dBvhSph.reserve(sph.size())); SDF_TRY(mesh->dBvhKids.reserve(kids.size())); SDF_TRY(mesh->dTriVerts.reserve(12ull * T));
    SDF_HIP_CHECK(hipMemcpyAsync(mesh->dBvhSph.p, sph.data(), sph.size() * sizeof(double), hipMemcpyHostToDevice, st));
    SDF_HIP_CHECK(hipMemcpyAsync(mesh->dBvhKids.p, kids.data(), kids.size() * sizeof(int), hipMemcpyHostToDevice, st));
    {
        std::vector<float> s32(sph.size());
        for (size_t i = 0; i < sph.size(); i++) s32[i] = (float)sph[i];
        float scale = 0.f;
        for (float c : mesh->hVerts) scale = std::max(scale, std::fabs(c));
        mesh->bvhCoordScale = scale;
        SDF_TRY(mesh->dBvhSph32.reserve(s32.size()));
        SDF_HIP_CHECK(hipMemcpyAsync(mesh->dBvhSph32.p, s32.data(), s32.size() * sizeof(float), hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipStreamSynchronize(st));
    }
    k_tri_verts<<<gridFor(12ull * T, 256), 256, 0, st>>>(mesh->dVerts.p, mesh->dIdx.p, T, mesh->dTriVerts.p);
    SDF_HIP_CHECK(hipGetLastError());
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    mesh->numBvhNodes = nn;
    mesh->hasBvh = true;
    if (seconds) *seconds = nowSeconds() - t0;
    return SDFHIP_OK;
}

int sdfhip_mesh_nearest(sdfhip_mesh* mesh, const float* xyz, uint64_t n, uint32_t* out_ids, int where) {
    SDF_REQUIRE(mesh && xyz && out_ids, "NULL argument");
    SDF_TRY(sdfhip_mesh_ensure_bvh(mesh));
    if (n == 0) return SDFHIP_OK;
    hipStream_t st = mesh->ctx->stream;
    SDF_HIP_CHECK(hipSetDevice(mesh->ctx->device));
    DevBuf<float> dp; DevBuf<uint32_t> dout;
    const float* p = xyz; uint32_t* o = out_ids;
    if (where == SDFHIP_HOST) {
        SDF_TRY(dp.reserve(3 * n)); SDF_TRY(dout.reserve(n));
        SDF_HIP_CHECK(hipMemcpyAsync(dp.p, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
        p = dp.p; o = dout.p;
    }
    k_nearest<<<gridFor(n, 128), 128, 0, st>>>(meshBvh(mesh), p, n, o);
    SDF_HIP_CHECK(hipGetLastError());
    if (where == SDFHIP_HOST) SDF_HIP_CHECK(hipMemcpyAsync(out_ids, dout.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    return SDFHIP_OK;
}

int sdfhip_mesh_nearest_stats(sdfhip_mesh* mesh, const float* xyz, uint64_t n, uint32_t* out4) {
    SDF_REQUIRE(mesh && xyz && out4, "NULL argument");
    SDF_TRY(sdfhip_mesh_ensure_bvh(mesh));
    hipStream_t st = mesh->ctx->stream;
    DevBuf<float> dp; DevBuf<uint32_t> dout;
    SDF_TRY(dp.reserve(3 * n)); SDF_TRY(dout.reserve(4 * n));
    SDF_HIP_CHECK(hipMemcpyAsync(dp.p, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
    k_nearest_stats<<<gridFor(n, 128), 128, 0, st>>>(meshBvh(mesh), dp.p, n, dout.p);
    SDF_HIP_CHECK(hipGetLastError());
    SDF_HIP_CHECK(hipMemcpyAsync(out4, dout.p, sizeof(uint32_t) * 4 * n, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    return SDFHIP_OK;
}

int sdfhip_mesh_point_values(sdfhip_mesh* mesh, const float* xyz, const uint32_t* tri_ids, uint64_t n, float* out8, int where) {
    SDF_REQUIRE(mesh && xyz && tri_ids && out8, "NULL argument");
    if (n == 0) return SDFHIP_OK;
    hipStream_t st = mesh->ctx->stream;
    SDF_HIP_CHECK(hipSetDevice(mesh->ctx->device));
    DevBuf<float> dp, dout; DevBuf<uint32_t> dt;
    const float* p = xyz; const uint32_t* t = tri_ids; float* o = out8;
    if (where == SDFHIP_HOST) {
        SDF_TRY(dp.reserve(3 * n)); SDF_TRY(dt.reserve(n)); SDF_TRY(dout.reserve(8 * n));
        SDF_HIP_CHECK(hipMemcpyAsync(dp.p, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipMemcpyAsync(dt.p, tri_ids, sizeof(uint32_t) * n, hipMemcpyHostToDevice, st));
        p = dp.p; t = dt.p; o = dout.p;
    }
    k_point_values<<<gridFor(n, 128), 128, 0, st>>>(mesh->dVerts.p, mesh->dIdx.p, mesh->dTri.p, p, t, n, o);
    SDF_HIP_CHECK(hipGetLastError());
    if (where == SDFHIP_HOST) SDF_HIP_CHECK(hipMemcpyAsync(out8, dout.p, sizeof(float) * 8 * n, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    return SDFHIP_OK;
}

}  // extern "C"
