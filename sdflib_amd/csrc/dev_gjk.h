// Frank-Wolfe "is this triangle near the node" test on the device.  PRODUCT code — independent of oracle/.
//
// Behaviour reproduced op for op (fp32, no FMA): GJK::IsNearMinimize(halfNodeSize, vertRadius[8], triangle[3],
// distThreshold) and its support functions (reference src/utils/GJK.cpp:830-866, 715-738, 644-652): at most 15
// Frank-Wolfe iterations on the Minkowski difference (hull of the 8 corner spheres) (-) triangle, early accept when
// |x|^2 < thr^2, early reject through the separating-plane bound.  Compile with -ffp-contract=off.
#pragma once
#include "dev_math.h"

namespace sdfhip {

SDF_DEV F3 cornerRel(int i) { return F3{(i & 1) ? 1.f : -1.f, (i & 2) ? 1.f : -1.f, (i & 4) ? 1.f : -1.f}; }

SDF_DEV F3 furthestOnHull(float half, const float* __restrict__ radius, F3 dir) {
    float best = dot(F3{-half, -half, -half}, dir) + radius[0];
    int bi = 0;
#pragma unroll
    for (int i = 1; i < 8; i++) {
        const float v = dot(cornerRel(i) * half, dir) + radius[i];
        if (v > best) { best = v; bi = i; }
    }
    return cornerRel(bi) * half + radius[bi] * dir;
}

SDF_DEV F3 furthestOnTriangle(F3 t0, F3 t1, F3 t2, F3 dir) {
    const float d1 = dot(t0, dir), d2 = dot(t1, dir), d3 = dot(t2, dir);
    if (d1 > d2) return (d1 > d3) ? t0 : t2;
    return (d2 > d3) ? t1 : t2;
}

SDF_DEV bool isNearMinimize(float half, const float* __restrict__ radius, F3 t0, F3 t1, F3 t2, float thr) {
    const float sqThr = thr * thr;
    F3 cur = -t0;
    unsigned iter = 0;
    bool isNear = false;
    float distToP, distToO;
    do {
        const F3 g = normalize(-cur);
        const F3 p = furthestOnHull(half, radius, g) - furthestOnTriangle(t0, t1, t2, -g);
        distToP = dot(g, p - cur);
        distToO = dot(g, -cur);
        const F3 dir = p - cur;
        const float d = dot(dir, -cur);
        if ((double)d < 1.0e-5) return distToO <= distToP + thr;
        cur = cur + dir * gmin(d / dot(dir, dir), 1.0f);
        isNear = dot(cur, cur) < sqThr;
    } while (!isNear && distToO <= distToP + thr && ++iter < 15u);
    return isNear || iter >= 15u;
}

// The same test as a resumable state machine (one Frank-Wolfe iteration per step) for kernels that keep a wave busy by giving a
// lane the next triangle as soon as its current one is decided.  Identical arithmetic and termination logic.
struct NearMinimizeState {
    F3 t0, t1, t2, cur; unsigned iter;
    SDF_DEV void start(F3 a, F3 b, F3 c) { t0 = a; t1 = b; t2 = c; cur = -a; iter = 0; }
    // returns true when decided; `result` is then the value isNearMinimize would have returned
    SDF_DEV bool step(float half, const float* __restrict__ radius, float thr, bool& result) {
        const F3 g = normalize(-cur);
        const F3 p = furthestOnHull(half, radius, g) - furthestOnTriangle(t0, t1, t2, -g);
        const float distToP = dot(g, p - cur);
        const float distToO = dot(g, -cur);
        const F3 dir = p - cur;
        const float d = dot(dir, -cur);
        if ((double)d < 1.0e-5) { result = distToO <= distToP + thr; return true; }
        cur = cur + dir * gmin(d / dot(dir, dir), 1.0f);
        const bool isNear = dot(cur, cur) < thr * thr;
        if (isNear) { result = true; return true; }
        if (!(distToO <= distToP + thr)) { result = iter >= 15u; return true; }     // (++iter is not evaluated in this case)
        if (++iter < 15u) return false;
        result = true;                                                               // isNear || iter >= 15
        return true;
    }
};

}  // namespace sdfhip
