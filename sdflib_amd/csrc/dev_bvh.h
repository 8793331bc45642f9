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
// Layout in HBM (what a visit touches, sized to cache lines):
//   sph  : one 64-B line per INNER node = the bounding spheres of its two children [lcx lcy lcz lr | rcx rcy rcz rr] (fp64)
//   kids : int2 per inner node = {left, right} child references; >= 0: inner node index, < 0: ~triangle id (the child is a
//          leaf, so visiting it means evaluating that triangle — leaf nodes are not materialised)
//   sph32: 32 B per inner node = the same two spheres rounded to fp32 [cx cy cz r | cx cy cz r]: the traversal first brackets
//          each sphere distance in fp32 from these (half the bytes, a quarter of the load instructions per lane) and only
//          fetches the fp64 line when the bracket cannot decide a comparison.
//   triV : 48 B per triangle = its three fp32 vertices gathered once [v0.xyz v1.x | v1.yz v2.xy | v2.z flag - -], so a leaf
//          visit reads one contiguous record instead of three indices plus three scattered vertices; flag != 0 marks a (nearly)
//          degenerate triangle, whose fp32 distance is not trusted by the candidate search (dev_bvh_fast.h).
//   wide : 128 B per inner node at an EVEN depth = that node and its children collapsed into one 4-wide node for the order-free
//          candidate search (dev_bvh_fast.h): [origin.xyz scale | 4 x 16 B (centre as 3 x u16 on the node's own grid, radius as a
//          half, slab direction as 3 x snorm16, slab half-width as a half; both halves rounded up after measuring them against the
//          DECODED centre / direction) | 4 child references (>= 0: inner node at the next even depth, < 0: ~triangle) | pad];
//          an empty slot has radius -inf, a child without a slab has width +inf.
//   triRank: u32 per triangle = its position in the tree's leaf order (the planner's final `order` array); the node over the
//          leaf range [b, e) splits at (b + e) / 2 and inner nodes are numbered in pre-order, so ranks make the tree navigable
//          by arithmetic alone (dev_bvh_fast.h).
// Inner nodes are numbered in the reference's pre-order (left subtree first); the traversal order is the reference's.
#pragma once
#include "dev_math.h"
#include "sdfhip_internal.h"

namespace sdfhip {

constexpr double BVH_NO_BOUND = 1.7976931348623157e308;     // std::numeric_limits<double>::max(): the reference's start value
constexpr double BVH_HUGE = 1e300;

struct BvhDev { const double2* sph; const int2* kids; const float4* triV; uint32_t numTriangles; const float4* sph32; float coordScale; const uint32_t* triRank; const float4* wide; };
static inline BvhDev meshBvh(const sdfhip_mesh* m) {
    return BvhDev{reinterpret_cast<const double2*>(m->dBvhSph.p), reinterpret_cast<const int2*>(m->dBvhKids.p), reinterpret_cast<const float4*>(m->dTriVerts.p), m->numTriangles,
                  reinterpret_cast<const float4*>(m->dBvhSph32.p), m->bvhCoordScale, m->dTriRank.p, reinterpret_cast<const float4*>(m->dBvhWide.p)};
}

struct D3 { double x, y, z; };
SDF_DEV D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
SDF_DEV double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Eberly's point/triangle squared distance in fp64; only the value is needed (the id decides everything else).
// Same operations per region as the reference's point_triangle_sq_unsigned; the region logic only SELECTS the operands
// of the single division each region performs (1/det, numer/denom, -b0/a00, -b1/a11), so that a wave whose lanes fall into
// different regions executes one division and one straight-line tail instead of six divergent ones.
SDF_DEV double pointTriangleSq(D3 point, D3 v0, D3 v1, D3 v2) {
    const D3 diff = v0 - point, e0 = v1 - v0, e1 = v2 - v0;
    const double a00 = ddot(e0, e0), a01 = ddot(e0, e1), a11 = ddot(e1, e1);
    const double b0 = ddot(diff, e0), b1 = ddot(diff, e1), c = ddot(diff, diff);
    const double det = fabs(a00 * a11 - a01 * a01);
    double s = a01 * b1 - a11 * b0;
    double t = a01 * b0 - a00 * b1;
    enum { V0, V1, V2, E01, E02, R0, E12S, E12T };      // E12S: s = numer/denom, t = 1-s ; E12T: t = numer/denom, s = 1-t
    // region logic as pure selects (every operand is cheap and side-effect free): no divergent branches in a wave
    const double e12den = a00 - 2.0 * a01 + a11;
    const int alongE02 = (b1 >= 0) ? V0 : ((-b1 >= a11) ? V2 : E02);
    const int alongE01 = (b0 >= 0) ? V0 : ((-b0 >= a00) ? V1 : E01);
    const bool sNeg = s < 0, tNeg = t < 0;
    // s + t <= det: regions 4, 3, 5, 0
    const int kIn = sNeg ? (tNeg ? ((b0 < 0) ? ((-b0 >= a00) ? V1 : E01) : alongE02) : alongE02) : (tNeg ? alongE01 : R0);
    // otherwise: regions 2, 6, 1
    const double r2a = a01 + b0, r2b = a11 + b1, n2 = r2b - r2a;
    const int k2 = (r2b > r2a) ? ((n2 >= e12den) ? V1 : E12S) : ((r2b <= 0) ? V2 : ((b1 >= 0) ? V0 : E02));
    const double r6a = a01 + b1, r6b = a00 + b0, n6 = r6b - r6a;
    const int k6 = (r6b > r6a) ? ((n6 >= e12den) ? V2 : E12T) : ((r6b <= 0) ? V1 : ((b0 >= 0) ? V0 : E01));
    const double n1 = a11 + b1 - a01 - b0;
    const int k1 = (n1 <= 0) ? V2 : ((n1 >= e12den) ? V1 : E12S);
    const bool inside = s + t <= det;
    const int kind = inside ? kIn : (sNeg ? k2 : (tNeg ? k6 : k1));
    double numer = sNeg ? n2 : (tNeg ? n6 : n1);
    double denom;
    numer = (kind == R0) ? 1.0 : ((kind == E01) ? -b0 : ((kind == E02) ? -b1 : ((kind >= E12S) ? numer : 0.0)));
    denom = (kind == R0) ? det : ((kind == E01) ? a00 : ((kind == E02) ? a11 : ((kind >= E12S) ? e12den : 1.0)));
    const double q = numer / denom;
    const double sq = (kind == R0) ? s * q : ((kind == E12S) ? q : 1.0 - q);
    const double tq = (kind == R0) ? t * q : ((kind == E12T) ? q : 1.0 - q);
    const double dQuad = sq * (a00 * sq + a01 * tq + 2.0 * b0) + tq * (a01 * sq + a11 * tq + 2.0 * b1) + c;
    double d2 = dQuad;
    d2 = (kind == V0) ? c : d2;
    d2 = (kind == V1) ? a00 + 2.0 * b0 + c : d2;
    d2 = (kind == V2) ? a11 + 2.0 * b1 + c : d2;
    d2 = (kind == E01) ? b0 * q + c : d2;
    d2 = (kind == E02) ? b1 * q + c : d2;
    if (d2 < 0) d2 = 0;
    return d2;
}

SDF_DEV double triangleSq(const BvhDev& b, uint32_t t, D3 p) {
    const float4 q0 = b.triV[3 * (size_t)t], q1 = b.triV[3 * (size_t)t + 1], q2 = b.triV[3 * (size_t)t + 2];
    return pointTriangleSq(p, D3{(double)q0.x, (double)q0.y, (double)q0.z}, D3{(double)q0.w, (double)q1.x, (double)q1.y}, D3{(double)q1.z, (double)q1.w, (double)q2.x});
}

// Traversal stack: one 4-byte entry per deferred child = (parent node index << 1) | (deferred child is the RIGHT one).
// It lives in LDS (column `threadIdx` of a [BVH_STACK][blockDim] array, so lanes never conflict) instead of scratch:
// the first profile showed the per-lane scratch stack turning 94 MB of algorithmic output into 35 GB of writes per
// launch.  The deferred child's distance is RECOMPUTED at pop time from the parent's sphere (same inputs, same
// rounding), so nothing but the index needs to be kept.
constexpr int BVH_STACK = 32;          // >= tree depth; the median-split tree over T triangles has depth ceil(log2 T) + 1

// Sphere lower bound exactly as the reference computes it: |p - center| - radius, fp64.
struct SphereTerms { double dd, r; };
SDF_DEV SphereTerms sphereTerms(const double2* nd, int which, D3 p) {
    const double2 a = nd[2 * which], b = nd[2 * which + 1];
    const D3 d = p - D3{a.x, a.y, b.x};
    return SphereTerms{ddot(d, d), b.y};
}
SDF_DEV double sphereDistExact(SphereTerms t) { return sqrt(t.dd) - t.r; }
SDF_DEV double sphereDist(const double2* nd, int which, D3 p) { return sphereDistExact(sphereTerms(nd, which, p)); }

// The traversal only needs the OUTCOME of `dist < best` and `distL < distR`.  An fp32 hardware square root of dd (relative
// error <= 2^-23 from the conversion and the 1-ulp v_sqrt_f32) brackets the exact fp64 value within `slack`; whenever the
// bracket decides the comparison the IEEE fp64 square root (~25 instructions) is skipped, otherwise it is evaluated — so
// every decision equals the reference's.  dd below 1e-30 (fp32 denormal range after the root) always takes the exact path.
struct SphereApprox { double x, slack; };
SDF_DEV SphereApprox sphereApprox(SphereTerms t) {
    const double a = (double)__builtin_amdgcn_sqrtf((float)t.dd);
    return SphereApprox{a - t.r, (t.dd > 1e-30 && t.dd < 1e30) ? (4e-7 * a + 1e-12 * t.r) : BVH_HUGE};
}
// Bracket of the same distance from the fp32 copy of the sphere: |c32 - c| <= 2^-24 |c| per component (<= 1.1e-7 * coordScale
// in length), |r32 - r| <= 6e-8 r, and the fp32 evaluation of sqrt(|p - c32|^2) is within 4e-7 relative; `slack` covers
// all of it with a margin of 2x.
SDF_DEV SphereApprox sphereApprox32(float4 sp, F3 p, float coordScale) {
    const float dx = p.x - sp.x, dy = p.y - sp.y, dz = p.z - sp.z;
    const float a = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
    return SphereApprox{(double)(a - sp.w), (double)(1e-6f * (a + sp.w) + 5e-7f * coordScale)};
}
// dist < best ?
SDF_DEV bool sphereCloser(SphereTerms t, double best) {
    const SphereApprox q = sphereApprox(t);
    if (q.x + q.slack < best) return true;
    if (q.x - q.slack >= best) return false;
    return sphereDistExact(t) < best;
}


// ---- traversal ------------------------------------------------------------------------------------------------------
// The reference's recursion (nearer child first; the farther child's prune test is made AFTER the nearer subtree has been
// searched; strict '<' both for pruning and for adopting a triangle) unrolled into two alternating steps on an explicit
// stack of deferred children:  pop one deferred child and test its sphere  /  enter the validated inner node (order its
// children, defer the farther one, test the nearer one) or evaluate the validated triangle.
struct TraversalLane {
    D3 p; F3 p32; double best; int bestTri; int sp; int ref; bool haveRef;
    SDF_DEV void start(const BvhDev& b, F3 pf, double best0 = BVH_NO_BOUND) {
        p = D3{(double)pf.x, (double)pf.y, (double)pf.z}; p32 = pf;
        best = best0; bestTri = (b.numTriangles == 1u) ? 0 : -1; sp = 0; ref = 0;
        haveRef = b.numTriangles > 1u;           // the root is entered without a test (one-triangle mesh: nothing to traverse)
    }
    // step 1; returns true when the query is finished (nothing validated, nothing deferred)
    template <int STRIDE>
    SDF_DEV bool popStep(const BvhDev& b, const uint32_t* __restrict__ stk) {
        if (sp == 0) return true;
        sp--;
        const uint32_t e = stk[sp * STRIDE];
        const int which = (int)(e & 1u);
        const int2 k = b.kids[e >> 1];
        const SphereApprox q = sphereApprox32(b.sph32[2 * (size_t)(e >> 1) + which], p32, b.coordScale);
        bool visit;
        if (q.x + q.slack < best) visit = true;
        else if (q.x - q.slack >= best) visit = false;
        else visit = sphereCloser(sphereTerms(b.sph + 4 * (size_t)(e >> 1), which, p), best);
        if (visit) { ref = which ? k.y : k.x; haveRef = true; }
        return false;
    }
    // step 2b: evaluate the validated triangle
    SDF_DEV void triStep(const BvhDev& b) {
        const double d2 = triangleSq(b, (uint32_t)~ref, p);
        if (d2 < best * best) { best = sqrt(d2); bestTri = ~ref; }
        haveRef = false;
    }
    // step 2a: enter the validated inner node
    template <int STRIDE>
    SDF_DEV void enterStep(const BvhDev& b, uint32_t* __restrict__ stk) {
        const int2 k = b.kids[ref];
        const SphereApprox qL = sphereApprox32(b.sph32[2 * (size_t)ref], p32, b.coordScale), qR = sphereApprox32(b.sph32[2 * (size_t)ref + 1], p32, b.coordScale);
        bool leftFirst, visit;
        const bool orderKnown = (qL.x + (qL.slack + qR.slack) < qR.x) || (qL.x - (qL.slack + qR.slack) > qR.x);
        leftFirst = qL.x < qR.x;
        const SphereApprox qF = leftFirst ? qL : qR;
        const bool visitKnown = (qF.x + qF.slack < best) || (qF.x - qF.slack >= best);
        visit = qF.x + qF.slack < best;
        if (!(orderKnown && visitKnown)) {           // rare: decide both from the fp64 spheres, as the reference does
            const double2* nd = b.sph + 4 * (size_t)ref;
            const double dL = sphereDistExact(sphereTerms(nd, 0, p)), dR = sphereDistExact(sphereTerms(nd, 1, p));
            leftFirst = dL < dR;
            visit = (leftFirst ? dL : dR) < best;
        }
        stk[sp * STRIDE] = ((uint32_t)ref << 1) | (leftFirst ? 1u : 0u);     // deferred = right if the left goes first
        sp++;
        if (visit) ref = leftFirst ? k.x : k.y;
        else haveRef = false;
    }
};

// Nearest triangle id for a float point (widened to double).  STRIDE = LDS stride between the lane's stack entries.
// counts[0..2] (optional, dev probe): inner nodes entered, iterations of the WAVE's loop while this lane was in it, triangles evaluated.
// best0: a bound STRICTLY above the nearest distance (default: none, as the reference starts).  The result is the same: until the
// unbounded run adopts its first triangle below best0 its `best` is >= best0, so it enters every node and adopts that very triangle
// too, and from there the two runs are in the same state; what the bounded run skipped adopts nothing below best0 (dev_bvh_fast.h).
template <int STRIDE, bool STATS = false>
SDF_DEV uint32_t bvhNearest(const BvhDev& b, F3 pf, uint32_t* __restrict__ stk, uint32_t* counts = nullptr, double best0 = BVH_NO_BOUND) {
    TraversalLane L;
    L.start(b, pf, best0);
    // wave-synchronous form: every iteration runs the pop step, then the enter / triangle step, for the lanes that need them
    bool alive = true;
    while (__ballot(alive) != 0ull) {
        if (STATS) counts[1]++;
        if (alive && !L.haveRef) {
            if (L.template popStep<STRIDE>(b, stk)) alive = false;
        }
        if (alive && L.haveRef && L.ref >= 0) { L.template enterStep<STRIDE>(b, stk); if (STATS) counts[0]++; }
        if (alive && L.haveRef && L.ref < 0) { L.triStep(b); if (STATS) counts[2]++; }
    }
    return (uint32_t)L.bestTri;
}

}  // namespace sdfhip
