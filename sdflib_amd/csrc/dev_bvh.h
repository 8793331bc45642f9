// fp64 bounding-sphere BVH traversal on the device.  PRODUCT code — independent of oracle/.
//
// Behaviour reproduced: tmd::TriangleMeshDistance::_query + point_triangle_sq_unsigned as used by
// ICG::getNearestTriangle (reference libs/InteractiveComputerGraphics/InteractiveComputerGraphics/
// TriangleMeshDistance.h:492-540, 542-797; include/SdfLib/TrianglesInfluence.h:898-905).  The nearest-triangle ID
// depends on the traversal order (nearer child first, strict '<' on both the prune and the update), so the
// recursion is unrolled onto an explicit stack that defers the second child's prune test to pop time — the test
// then sees the distance found in the first child's subtree, exactly like the recursive original.
// Compile with -ffp-contract=off (fp64 products and sums must round separately).
//
// Node layout in HBM: 10 doubles (80 B, 16-B aligned): [lcx lcy lcz lr | rcx rcy rcz rr | {left,right} as 2 x i32 | pad]
#pragma once
#include "dev_math.h"

namespace sdfhip {

constexpr int BVH_NODE_DOUBLES = 10;

struct D3 { double x, y, z; };
SDF_DEV D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
SDF_DEV double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Eberly's point/triangle squared distance in fp64; only the value is needed (the id decides everything else).
SDF_DEV double pointTriangleSq(D3 point, D3 v0, D3 v1, D3 v2) {
    const D3 diff = v0 - point, e0 = v1 - v0, e1 = v2 - v0;
    const double a00 = ddot(e0, e0), a01 = ddot(e0, e1), a11 = ddot(e1, e1);
    const double b0 = ddot(diff, e0), b1 = ddot(diff, e1), c = ddot(diff, diff);
    const double det = fabs(a00 * a11 - a01 * a01);
    double s = a01 * b1 - a11 * b0;
    double t = a01 * b0 - a00 * b1;
    double d2;
    // results of the seven possible nearest features
    const double dV0 = c;
    const double dV1 = a00 + 2.0 * b0 + c;
    const double dV2 = a11 + 2.0 * b1 + c;
    enum { V0, V1, V2, E01, E02, QUAD } kind;
    if (s + t <= det) {
        if (s < 0) {
            if (t < 0) {                                   // region 4
                if (b0 < 0) kind = (-b0 >= a00) ? V1 : E01;
                else kind = (b1 >= 0) ? V0 : ((-b1 >= a11) ? V2 : E02);
            } else kind = (b1 >= 0) ? V0 : ((-b1 >= a11) ? V2 : E02);   // region 3
        } else if (t < 0) kind = (b0 >= 0) ? V0 : ((-b0 >= a00) ? V1 : E01);   // region 5
        else {                                             // region 0
            const double invDet = 1.0 / det;
            s *= invDet; t *= invDet;
            kind = QUAD;
        }
    } else {
        if (s < 0) {                                       // region 2
            const double tmp0 = a01 + b0, tmp1 = a11 + b1;
            if (tmp1 > tmp0) {
                const double numer = tmp1 - tmp0, denom = a00 - 2.0 * a01 + a11;
                if (numer >= denom) kind = V1;
                else { s = numer / denom; t = 1.0 - s; kind = QUAD; }
            } else kind = (tmp1 <= 0) ? V2 : ((b1 >= 0) ? V0 : E02);
        } else if (t < 0) {                                // region 6
            const double tmp0 = a01 + b1, tmp1 = a00 + b0;
            if (tmp1 > tmp0) {
                const double numer = tmp1 - tmp0, denom = a00 - 2.0 * a01 + a11;
                if (numer >= denom) kind = V2;
                else { t = numer / denom; s = 1.0 - t; kind = QUAD; }
            } else kind = (tmp1 <= 0) ? V1 : ((b0 >= 0) ? V0 : E01);
        } else {                                           // region 1
            const double numer = a11 + b1 - a01 - b0;
            if (numer <= 0) kind = V2;
            else {
                const double denom = a00 - 2.0 * a01 + a11;
                if (numer >= denom) kind = V1;
                else { s = numer / denom; t = 1.0 - s; kind = QUAD; }
            }
        }
    }
    switch (kind) {
        case V0: d2 = dV0; break;
        case V1: d2 = dV1; break;
        case V2: d2 = dV2; break;
        case E01: { const double ss = -b0 / a00; d2 = b0 * ss + c; break; }
        case E02: { const double tt = -b1 / a11; d2 = b1 * tt + c; break; }
        default: d2 = s * (a00 * s + a01 * t + 2.0 * b0) + t * (a01 * s + a11 * t + 2.0 * b1) + c; break;
    }
    if (d2 < 0) d2 = 0;
    return d2;
}

SDF_DEV D3 loadVertexD(const float* __restrict__ verts, uint32_t v) {
    return D3{(double)verts[3 * v], (double)verts[3 * v + 1], (double)verts[3 * v + 2]};
}

// Traversal stack: one 4-byte entry per deferred child = (parent node index << 1) | (deferred child is the RIGHT one).
// It lives in LDS (column `threadIdx` of a [BVH_STACK][blockDim] array, so lanes never conflict) instead of scratch:
// the first profile showed the per-lane scratch stack turning 94 MB of algorithmic output into 35 GB of writes per
// launch.  The deferred child's distance is RECOMPUTED at pop time from the parent's sphere (same inputs, same
// rounding), so nothing but the index needs to be kept.
constexpr int BVH_STACK = 32;          // >= tree depth; the median-split tree over T triangles has depth ceil(log2 T) + 1

SDF_DEV double sphereDist(const double2* nd, int which, D3 p) {
    const double2 a = nd[2 * which], b = nd[2 * which + 1];
    const D3 d = p - D3{a.x, a.y, b.x};
    return sqrt(ddot(d, d)) - b.y;
}

constexpr double BVH_NO_BOUND = 1.7976931348623157e308;     // std::numeric_limits<double>::max(): the reference's start value

// Nearest triangle id for a float point (widened to double), root = node 0.  STRIDE = LDS stride between entries.
//
// initBest: an UPPER bound on the distance from the point to the mesh that is strictly larger than the true nearest
// distance (BVH_NO_BOUND reproduces the reference literally).  Starting from a finite bound returns the SAME triangle as
// the reference's traversal: the visiting order is unchanged; a subtree skipped only because of the bound has a sphere
// lower bound >= initBest > nearest distance, so it holds no triangle that the reference could end up with; the first
// triangle closer than the bound is adopted by both traversals (the reference holds either +inf or a farther triangle at
// that moment) and from then on both carry the identical `best`, hence take identical decisions.  If the bound turns
// out to be wrong (no triangle adopted) the caller falls back to the unbounded traversal.
template <int STRIDE>
SDF_DEV uint32_t bvhNearest(const double* __restrict__ nodes, const float* __restrict__ verts, const uint32_t* __restrict__ idx, F3 pf,
                            uint32_t* __restrict__ stk, double initBest = BVH_NO_BOUND) {
    const D3 p = D3{(double)pf.x, (double)pf.y, (double)pf.z};
    double best = initBest;
    int bestTri = -1;
    int sp = 0;
    int cur = 0;
    for (;;) {
        const double2* nd = reinterpret_cast<const double2*>(nodes + (size_t)BVH_NODE_DOUBLES * cur);
        const double2 q4 = nd[4];
        const int left = __double2loint(q4.x), right = __double2hiint(q4.x);
        bool descend = false;
        if (left == -1) {
            const uint32_t t = (uint32_t)right;
            const double d2 = pointTriangleSq(p, loadVertexD(verts, idx[3 * t]), loadVertexD(verts, idx[3 * t + 1]), loadVertexD(verts, idx[3 * t + 2]));
            if (d2 < best * best) { best = sqrt(d2); bestTri = right; }
        } else {
            const double distL = sphereDist(nd, 0, p);
            const double distR = sphereDist(nd, 1, p);
            const bool leftFirst = distL < distR;
            const double dFirst = leftFirst ? distL : distR;
            stk[sp * STRIDE] = ((uint32_t)cur << 1) | (leftFirst ? 1u : 0u);     // deferred = right if the left goes first
            sp++;
            if (dFirst < best) { cur = leftFirst ? left : right; descend = true; }
        }
        if (descend) continue;
        bool found = false;
        while (sp > 0) {
            sp--;
            const uint32_t e = stk[sp * STRIDE];
            const double2* pn = reinterpret_cast<const double2*>(nodes + (size_t)BVH_NODE_DOUBLES * (e >> 1));
            const int which = (int)(e & 1u);
            if (sphereDist(pn, which, p) < best) {
                const double2 c4 = pn[4];
                cur = which ? __double2hiint(c4.x) : __double2loint(c4.x);
                found = true; break;
            }
        }
        if (!found) break;
    }
    return (uint32_t)bestTri;
}

}  // namespace sdfhip
