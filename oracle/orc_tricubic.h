// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// Restates TriCubicInterpolation and the subdivision rules (reference file:line):
//   calculateCoefficients    include/SdfLib/InterpolationMethods.h:292-378
//   interpolateValue         include/SdfLib/InterpolationMethods.h:432-439   (scalar, ENOKI off flavour; -DORC_ENOKI_ORDER: :383-430)
//   interpolateGradient      include/SdfLib/InterpolationMethods.h:442-455
//   interpolateVertexValues  include/SdfLib/InterpolationMethods.h:457-497
//   trapezoid / Simpson / by-distance rules   include/SdfLib/OctreeSdfUtils.h:60-85, 213-238, 87-138
//
// The reference writes all of these as generated straight-line code.  The generating RULES are restated:
//  * fit: coeff[i+4j+16k] = sum over (vertex v ascending, value slot q ascending) of M * in[v][q], zero
//    entries skipped, every product rounded, summed left to right.  M is the inverse of the Hermite
//    constraint system = H (x) H (x) H with the 1-D cubic Hermite matrix H below (derived, not copied; the
//    reference's own derivation tool is src/tools/CalculateInterpolationParameters/main.cpp:22-143).
//  * eval: 0.0f + sum over n = i+4j+16k ascending of ((c[n] * x..i times) * y..j times) * z..k times.
//  * derivatives: terms with non-zero integer factor f (i, j, k, i*j, i*k, j*k, i*j*k) in ascending n,
//    each term ((float(f) * c[n]) * x.. * y.. * z..) with the differentiated powers reduced by one.
// tools/check_ref_expressions.py parses the reference's literal expressions (when /root/reference is
// present) and verifies these rules term by term.
#pragma once
#include "orc_math.h"

namespace orc {

// 1-D cubic Hermite: power i from (f(0), f'(0), f(1), f'(1)).
static const int HERMITE_1D[4][4] = {
    { 1,  0,  0,  0},
    { 0,  1,  0,  0},
    {-3, -2,  3, -1},
    { 2,  1, -2,  1},
};

// Value slot q of a Hermite vector [f, fx, fy, fz, fxy, fxz, fyz, fxyz] -> derivative flags (ex, ey, ez).
static const int SLOT_DERIV[8][3] = {{0,0,0},{1,0,0},{0,1,0},{0,0,1},{1,1,0},{1,0,1},{0,1,1},{1,1,1}};

struct FitMatrix {
    int m[64][64];    // [coefficient n][8*vertex + slot]
    FitMatrix() {
        for (int k = 0; k < 4; k++) for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) {
            const int n = i + 4 * j + 16 * k;
            for (int v = 0; v < 8; v++) for (int q = 0; q < 8; q++) {
                const int bx = v & 1, by = (v >> 1) & 1, bz = (v >> 2) & 1;
                m[n][8 * v + q] = HERMITE_1D[i][2 * bx + SLOT_DERIV[q][0]] *
                                  HERMITE_1D[j][2 * by + SLOT_DERIV[q][1]] *
                                  HERMITE_1D[k][2 * bz + SLOT_DERIV[q][2]];
            }
        }
    }
};
static inline const FitMatrix& fitMatrix() { static const FitMatrix fm; return fm; }

// in: 8 vertices x 8 slots (world-space derivatives); nodeSize = 2*halfSize.
static inline void tricubicFit(const float in[8][8], float nodeSize, float out[64]) {
    float s[64];
    const float sq = nodeSize * nodeSize;
    for (int v = 0; v < 8; v++) {
        s[8 * v + 0] = in[v][0];
        s[8 * v + 1] = in[v][1] * nodeSize;
        s[8 * v + 2] = in[v][2] * nodeSize;
        s[8 * v + 3] = in[v][3] * nodeSize;
        s[8 * v + 4] = in[v][4] * sq;
        s[8 * v + 5] = in[v][5] * sq;
        s[8 * v + 6] = in[v][6] * sq;
        s[8 * v + 7] = in[v][7] * (sq * nodeSize);
    }
    const FitMatrix& fm = fitMatrix();
    for (int n = 0; n < 64; n++) {
        float acc = 0.0f; bool first = true;
        for (int col = 0; col < 64; col++) {
            const int c = fm.m[n][col];
            if (c == 0) continue;
            const float term = (float)c * s[col];
            if (first) { acc = term; first = false; } else acc = acc + term;
        }
        out[n] = acc;
    }
}

static inline float powTerm(float t, float x, int i, float y, int j, float z, int k) {
    for (int a = 0; a < i; a++) t = t * x;
    for (int a = 0; a < j; a++) t = t * y;
    for (int a = 0; a < k; a++) t = t * z;
    return t;
}

static inline float tricubicValueLiteral(const float c[64], V3 f) {
    float acc = 0.0f;
    for (int n = 0; n < 64; n++) {
        const int i = n & 3, j = (n >> 2) & 3, k = n >> 4;
        acc = acc + powTerm(c[n], f.x, i, f.y, j, f.z, k);
    }
    return acc;
}
// interpolateValue of the SDFLIB_USE_ENOKI=ON flavour (include/SdfLib/InterpolationMethods.h:383-430; the reference's CMake default,
// CMakeLists.txt:24): four 4-wide power vectors x1 = (1, x, x x, (x x) x), x2 = y x1, x3 = y x2, x4 = y x3, every z-slab adds
// dot(x1, c[16k..]) + dot(x2, c[16k+4..]) + dot(x3, c[16k+8..]) + dot(x4, c[16k+12..]) (left to right) to the sum and then multiplies
// the four vectors by z.  enoki::dot of two Array<float, 4> = (a0 b0 + a1 b1) + (a2 b2 + a3 b3), every product rounded: that is
// DPPS's documented order (Enoki's SSE4.2 path, _mm_dp_ps) and equally the order of Enoki's generic hsum(a * b) (low half + high half).
// Enoki itself is NOT under /root/reference (fetched by libs/CMakeLists.txt:99-131): this order is restated from those semantics and
// is "parity unpinned" like everything else that needs the absent third-party headers.
static inline float tricubicValueEnoki(const float c[64], V3 f) {
    float x[4][4];
    x[0][0] = 1.0f; x[0][1] = f.x; x[0][2] = f.x * f.x; x[0][3] = f.x * f.x * f.x;
    for (int j = 1; j < 4; j++) for (int i = 0; i < 4; i++) x[j][i] = f.y * x[j - 1][i];
    float sum = 0.0f;
    for (int k = 0; k < 4; k++) {
        if (k > 0) for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) x[j][i] = f.z * x[j][i];
        float d[4];
        for (int j = 0; j < 4; j++) {
            const float* v = c + 16 * k + 4 * j;
            d[j] = (x[j][0] * v[0] + x[j][1] * v[1]) + (x[j][2] * v[2] + x[j][3] * v[3]);
        }
        const float slab = d[0] + d[1] + d[2] + d[3];
        sum = (k == 0) ? slab : sum + slab;
    }
    return sum;
}
// the flavour every direct caller of interpolateValue sees (rules, getDistance, minimum border value, the CONTINUITY builder's own
// error estimate); interpolateVertexValues spells its value out literally in both flavours (InterpolationMethods.h:459-464)
static inline float tricubicValue(const float c[64], V3 f) {
#ifdef ORC_ENOKI_ORDER
    return tricubicValueEnoki(c, f);
#else
    return tricubicValueLiteral(c, f);
#endif
}

// Generic derivative sum: ex/ey/ez in {0,1} select which variables are differentiated once.
static inline float tricubicDeriv(const float c[64], V3 f, int ex, int ey, int ez) {
    float acc = 0.0f; bool first = true;
    for (int n = 0; n < 64; n++) {
        const int i = n & 3, j = (n >> 2) & 3, k = n >> 4;
        const int fac = (ex ? i : 1) * (ey ? j : 1) * (ez ? k : 1);
        if (fac == 0) continue;
        const float term = powTerm((float)fac * c[n], f.x, i - ex, f.y, j - ey, f.z, k - ez);
        if (first) { acc = term; first = false; } else acc = acc + term;
    }
    return acc;
}

static inline V3 tricubicGradient(const float c[64], V3 f) {
    return V3{tricubicDeriv(c, f, 1, 0, 0), tricubicDeriv(c, f, 0, 1, 0), tricubicDeriv(c, f, 0, 0, 1)};
}

static inline void tricubicVertexValues(const float c[64], V3 f, float nodeSize, float out[8]) {
    out[0] = tricubicValueLiteral(c, f);
    out[1] = tricubicDeriv(c, f, 1, 0, 0) / nodeSize;
    out[2] = tricubicDeriv(c, f, 0, 1, 0) / nodeSize;
    out[3] = tricubicDeriv(c, f, 0, 0, 1) / nodeSize;
    const float sq = nodeSize * nodeSize;
    out[4] = tricubicDeriv(c, f, 1, 1, 0) / sq;
    out[5] = tricubicDeriv(c, f, 1, 0, 1) / sq;
    out[6] = tricubicDeriv(c, f, 0, 1, 1) / sq;
    out[7] = tricubicDeriv(c, f, 1, 1, 1) / (sq * nodeSize);
}

// ---- the 27-point stencil of a node -------------------------------------------------------------------
// Grid point g = gx + 3*gy + 9*gz, g? in {0,1,2} <-> relative position g? - 1 in {-1,0,1}.
// Corners (all g? in {0,2}) are the 8 node vertices, vertex id = bx + 2*by + 4*bz
// (include/SdfLib/TrianglesInfluence.h:25-36); the 19 others, in ascending g, are the reference's
// nodeSamplePoints (src/sdf/OctreeSdfDepthFirst.h:139-162).
struct Stencil {
    int midGrid[19];        // mid-point m -> grid index g
    int gridToMid[27];      // g -> mid-point index or -1
    int gridToCorner[27];   // g -> vertex id or -1
    V3 midRel[19];          // relative position in {-1,0,1}^3
    float midWeight[19];    // 2, 4 or 8 (number of half-axes), used by the trapezoid rule as w/64
    // child c, vertex j  -> source: >= 0 mid-point index, < 0 : -(parent vertex id) - 1
    int childSrc[8][8];
    Stencil() {
        int m = 0;
        for (int g = 0; g < 27; g++) {
            const int gx = g % 3, gy = (g / 3) % 3, gz = g / 9;
            const bool corner = gx != 1 && gy != 1 && gz != 1;
            gridToMid[g] = -1; gridToCorner[g] = -1;
            if (corner) gridToCorner[g] = (gx >> 1) + 2 * (gy >> 1) + 4 * (gz >> 1);
            else {
                midGrid[m] = g; gridToMid[g] = m;
                midRel[m] = V3{(float)(gx - 1), (float)(gy - 1), (float)(gz - 1)};
                const int halves = (gx == 1) + (gy == 1) + (gz == 1);
                midWeight[m] = (float)(1 << halves);
                m++;
            }
        }
        for (int c = 0; c < 8; c++) for (int j = 0; j < 8; j++) {
            const int gx = (c & 1) + (j & 1), gy = ((c >> 1) & 1) + ((j >> 1) & 1), gz = ((c >> 2) & 1) + ((j >> 2) & 1);
            const int g = gx + 3 * gy + 9 * gz;
            childSrc[c][j] = gridToMid[g] >= 0 ? gridToMid[g] : -(gridToCorner[g]) - 1;
        }
    }
};
static inline const Stencil& stencil() { static const Stencil st; return st; }

enum TerminationRule { RULE_NONE = 0, RULE_TRAPEZOIDAL = 1, RULE_SIMPSONS = 2, RULE_BY_DISTANCE = 3 };

// mid[m][0] is the exact distance at mid-point m.
static inline float ruleTrapezoid(const float c[64], const float mid[19][8]) {
    const Stencil& st = stencil();
    float acc = 0.0f;
    for (int m = 0; m < 19; m++) {
        const V3 f = V3{0.5f * st.midRel[m].x + 0.5f, 0.5f * st.midRel[m].y + 0.5f, 0.5f * st.midRel[m].z + 0.5f};
        const float e = mid[m][0] - tricubicValue(c, f);
        const float term = (st.midWeight[m] / 64.0f) * (e * e);
        acc = (m == 0) ? term : acc + term;
    }
    return acc;
}

static inline float ruleSimpson(const float c[64], const float mid[19][8]) {
    const Stencil& st = stencil();
    float acc = 0.0f;
    for (int m = 0; m < 19; m++) {
        const V3 f = V3{0.5f * st.midRel[m].x + 0.5f, 0.5f * st.midRel[m].y + 0.5f, 0.5f * st.midRel[m].z + 0.5f};
        const float e = mid[m][0] - tricubicValue(c, f);
        const float w = st.midWeight[m] * st.midWeight[m];      // 4, 16, 64
        const float term = (w / 216.0f) * (e * e);
        acc = (m == 0) ? term : acc + term;
    }
    return acc;
}

static inline float ruleByDistance(const float c[64], const float mid[19][8], float decay) {
    const Stencil& st = stencil();
    float acc = 0.0f;
    for (int m = 0; m < 19; m++) {
        const V3 f = V3{0.5f * st.midRel[m].x + 0.5f, 0.5f * st.midRel[m].y + 0.5f, 0.5f * st.midRel[m].z + 0.5f};
        const float v = tricubicValue(c, f);
        const float e = gmax(std::fabs(mid[m][0] - v) - decay * std::fabs(v), 0.0f);
        acc += (st.midWeight[m] / 64.0f) * (e * e);
    }
    return acc;
}

static inline float ruleValue(int rule, const float c[64], const float mid[19][8], float param1) {
    switch (rule) {
        case RULE_TRAPEZOIDAL: return ruleTrapezoid(c, mid);
        case RULE_SIMPSONS: return ruleSimpson(c, mid);
        case RULE_BY_DISTANCE: return ruleByDistance(c, mid, param1);
        default: return INFINITY;
    }
}

}  // namespace orc
