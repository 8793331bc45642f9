// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// Restates the nearest-triangle query the reference's OctreeSdf build uses (VHQueries -> ICG ->
// tmd::TriangleMeshDistance), all in fp64:
//   _build_tree   libs/InteractiveComputerGraphics/InteractiveComputerGraphics/TriangleMeshDistance.h:421-490
//   _query        ...TriangleMeshDistance.h:492-540   (nearer child first, strict '<' update and prune)
//   point_triangle_sq_unsigned (Eberly's point/triangle regions)   ...TriangleMeshDistance.h:542-797
//   ICG::getNearestTriangle   include/SdfLib/TrianglesInfluence.h:898-905 (only triangle_id is consumed)
// The tree is a bounding-sphere BVH with one triangle per leaf; the median split sorts the range by the
// FIRST vertex's coordinate along the widest AABB axis with std::sort.  Ties in that key are ubiquitous
// (triangles sharing their first vertex), so the resulting permutation depends on libstdc++'s introsort;
// the same std::sort call with the same comparison outcomes is used here to obtain the same permutation.
#pragma once
#include "orc_math.h"
#include <vector>
#include <algorithm>
#include <limits>
#include <cmath>

namespace orc {

struct D3 { double x, y, z; };
static inline D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline double dnorm(D3 a) { return std::sqrt(ddot(a, a)); }
static inline double dcomp(const D3& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

struct Sphere { D3 center; double radius; };
struct BvhNode { Sphere bvLeft, bvRight; int left = -1, right = -1; };   // left == -1: leaf, right = triangle id

enum Entity { E_V0, E_V1, E_V2, E_E01, E_E12, E_E02, E_F };

// Squared distance point/triangle in fp64 (Eberly), returning also the nearest entity.
static inline double pointTriangleSq(const D3& point, const D3& v0, const D3& v1, const D3& v2, Entity& ent) {
    const D3 diff = v0 - point, e0 = v1 - v0, e1 = v2 - v0;
    const double a00 = ddot(e0, e0), a01 = ddot(e0, e1), a11 = ddot(e1, e1);
    const double b0 = ddot(diff, e0), b1 = ddot(diff, e1), c = ddot(diff, diff);
    const double det = std::abs(a00 * a11 - a01 * a01);
    double s = a01 * b1 - a11 * b0;
    double t = a01 * b0 - a00 * b1;
    double d2 = -1.0;

    auto atV0 = [&]() { ent = E_V0; d2 = c; };
    auto atV1 = [&]() { ent = E_V1; d2 = a00 + (2) * b0 + c; };
    auto atV2 = [&]() { ent = E_V2; d2 = a11 + (2) * b1 + c; };
    auto onE01 = [&]() { ent = E_E01; s = -b0 / a00; d2 = b0 * s + c; };
    auto onE02 = [&]() { ent = E_E02; t = -b1 / a11; d2 = b1 * t + c; };
    auto quad = [&]() { d2 = s * (a00 * s + a01 * t + (2) * b0) + t * (a01 * s + a11 * t + (2) * b1) + c; };
    auto alongE02 = [&]() { if (b1 >= 0) atV0(); else if (-b1 >= a11) atV2(); else onE02(); };
    auto alongE01 = [&]() { if (b0 >= 0) atV0(); else if (-b0 >= a00) atV1(); else onE01(); };

    if (s + t <= det) {
        if (s < 0) {
            if (t < 0) {                      // region 4
                if (b0 < 0) { if (-b0 >= a00) atV1(); else onE01(); }
                else alongE02();
            } else alongE02();                // region 3
        } else if (t < 0) alongE01();         // region 5
        else {                                // region 0 (interior)
            ent = E_F;
            const double invDet = (1) / det;
            s *= invDet; t *= invDet;
            quad();
        }
    } else {
        if (s < 0) {                          // region 2
            const double tmp0 = a01 + b0, tmp1 = a11 + b1;
            if (tmp1 > tmp0) {
                const double numer = tmp1 - tmp0, denom = a00 - (2) * a01 + a11;
                if (numer >= denom) atV1();
                else { ent = E_E12; s = numer / denom; t = 1 - s; quad(); }
            } else {
                if (tmp1 <= 0) atV2(); else if (b1 >= 0) atV0(); else onE02();
            }
        } else if (t < 0) {                   // region 6
            const double tmp0 = a01 + b1, tmp1 = a00 + b0;
            if (tmp1 > tmp0) {
                const double numer = tmp1 - tmp0, denom = a00 - (2) * a01 + a11;
                if (numer >= denom) atV2();
                else { ent = E_E12; t = numer / denom; s = 1 - t; quad(); }
            } else {
                if (tmp1 <= 0) atV1(); else if (b0 >= 0) atV0(); else onE01();
            }
        } else {                              // region 1
            const double numer = a11 + b1 - a01 - b0;
            if (numer <= 0) atV2();
            else {
                const double denom = a00 - (2) * a01 + a11;
                if (numer >= denom) atV1();
                else { ent = E_E12; s = numer / denom; t = 1 - s; quad(); }
            }
        }
    }
    if (d2 < 0) d2 = 0;
    return d2;
}

struct SphereBvh {
    std::vector<D3> verts;                 // fp64 copies of the mesh vertices
    std::vector<int> tri;                  // 3 ints per triangle
    std::vector<BvhNode> nodes;
    Sphere rootSphere;

    void build(const V3* vertices, uint32_t numVertices, const uint32_t* indices, uint32_t numTriangles) {
        verts.resize(numVertices);
        for (uint32_t i = 0; i < numVertices; i++) verts[i] = D3{(double)vertices[i].x, (double)vertices[i].y, (double)vertices[i].z};
        tri.resize(3 * (size_t)numTriangles);
        for (size_t i = 0; i < 3 * (size_t)numTriangles; i++) tri[i] = (int)indices[i];
        std::vector<int> order(numTriangles);
        for (uint32_t i = 0; i < numTriangles; i++) order[i] = (int)i;
        nodes.clear();
        nodes.reserve(2 * (size_t)numTriangles);
        nodes.push_back(BvhNode());
        buildRange(0, rootSphere, order, 0, (int)numTriangles);
    }

    const D3& vtx(int t, int k) const { return verts[tri[3 * (size_t)t + k]]; }

    void buildRange(int nodeId, Sphere& bs, std::vector<int>& order, int begin, int end) {
        const int n = end - begin;
        if (n == 1) {
            const int t = order[begin];
            nodes[nodeId].left = -1;
            nodes[nodeId].right = t;
            const D3 a = vtx(t, 0), b = vtx(t, 1), c = vtx(t, 2);
            const D3 s = (a + b) + c;
            const D3 center = D3{s.x / 3.0, s.y / 3.0, s.z / 3.0};
            bs.center = center;
            bs.radius = std::max(std::max(dnorm(a - center), dnorm(b - center)), dnorm(c - center));
            return;
        }
        const double lo = std::numeric_limits<double>::lowest(), hi = std::numeric_limits<double>::max();
        D3 top{lo, lo, lo}, bottom{hi, hi, hi}, center{0, 0, 0};
        for (int i = begin; i < end; i++)
            for (int k = 0; k < 3; k++) {
                const D3& p = vtx(order[i], k);
                center.x += p.x; center.y += p.y; center.z += p.z;
                top.x = std::max(top.x, p.x); bottom.x = std::min(bottom.x, p.x);
                top.y = std::max(top.y, p.y); bottom.y = std::min(bottom.y, p.y);
                top.z = std::max(top.z, p.z); bottom.z = std::min(bottom.z, p.z);
            }
        const double cnt = (double)(3 * n);
        center.x /= cnt; center.y /= cnt; center.z /= cnt;
        const double diag[3] = {top.x - bottom.x, top.y - bottom.y, top.z - bottom.z};
        const int dim = (int)(std::max_element(diag, diag + 3) - diag);
        double r2 = 0.0;
        for (int i = begin; i < end; i++)
            for (int k = 0; k < 3; k++) { const D3 d = center - vtx(order[i], k); r2 = std::max(r2, ddot(d, d)); }
        bs.center = center;
        bs.radius = std::sqrt(r2);

        std::sort(order.begin() + begin, order.begin() + end,
                  [this, dim](int a, int b) { return dcomp(vtx(a, 0), dim) < dcomp(vtx(b, 0), dim); });

        const int mid = (int)(0.5 * (begin + end));
        const int l = (int)nodes.size();
        nodes[nodeId].left = l;
        nodes.push_back(BvhNode());
        { Sphere sp; buildRange(l, sp, order, begin, mid); nodes[nodeId].bvLeft = sp; }
        const int r = (int)nodes.size();
        nodes[nodeId].right = r;
        nodes.push_back(BvhNode());
        { Sphere sp; buildRange(r, sp, order, mid, end); nodes[nodeId].bvRight = sp; }
    }

    struct Hit { double distance = std::numeric_limits<double>::max(); int tri = -1; Entity ent = E_F; };

    void query(Hit& h, int nodeId, const D3& p) const {
        const BvhNode& nd = nodes[nodeId];
        if (nd.left == -1) {
            Entity e;
            const double d2 = pointTriangleSq(p, vtx(nd.right, 0), vtx(nd.right, 1), vtx(nd.right, 2), e);
            if (d2 < h.distance * h.distance) { h.distance = std::sqrt(d2); h.tri = nd.right; h.ent = e; }
            return;
        }
        const double dl = dnorm(p - nd.bvLeft.center) - nd.bvLeft.radius;
        const double dr = dnorm(p - nd.bvRight.center) - nd.bvRight.radius;
        if (dl < dr) {
            if (dl < h.distance) query(h, nd.left, p);
            if (dr < h.distance) query(h, nd.right, p);
        } else {
            if (dr < h.distance) query(h, nd.right, p);
            if (dl < h.distance) query(h, nd.left, p);
        }
    }

    // ICG::getNearestTriangle: float point widened to double, triangle id of the unsigned nearest query.
    uint32_t nearestTriangle(V3 p, double* outDist = nullptr) const {
        Hit h;
        query(h, 0, D3{(double)p.x, (double)p.y, (double)p.z});
        if (outDist) *outDist = h.distance;
        return (uint32_t)h.tri;
    }
};

}  // namespace orc
