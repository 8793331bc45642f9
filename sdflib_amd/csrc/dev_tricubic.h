// Tricubic fit / evaluation on the device.  PRODUCT code — independent of oracle/.
//
// Reference behaviour reproduced (include/SdfLib/InterpolationMethods.h):
//   calculateCoefficients :292-378   64 coefficients = constant 64x64 integer matrix x 64 scaled Hermite values
//   interpolateValue      :432-439   sum of c[i+4j+16k] x^i y^j z^k, literal left-to-right products and sums (-DSDFHIP_ENOKI_ORDER: :383-430)
//   interpolateGradient   :442-455   three derivative sums (normalised by the caller)
//   interpolateVertexValues :457-497 value + 7 derivatives divided by nodeSize powers
// and the subdivision rules of include/SdfLib/OctreeSdfUtils.h:60-85, 87-138, 213-238.
//
// The fit matrix is the inverse of the Hermite constraint system, i.e. H (x) H (x) H for the 1-D cubic Hermite
// matrix H; it is expanded at COMPILE TIME into straight-line code so that zero entries vanish and the summation
// order (vertex-major, slot-minor, left to right, each product rounded) equals the reference's generated code.
// Compile with -ffp-contract=off.
#pragma once
#include "dev_math.h"
#include <utility>

namespace sdfhip {

constexpr int kH[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {-3, -2, 3, -1}, {2, 1, -2, 1}};
constexpr int kSlotD[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};

// Matrix entry for coefficient n (= i + 4j + 16k) and column col (= 8*vertex + slot).
constexpr int fitCoef(int n, int col) {
    const int i = n & 3, j = (n >> 2) & 3, k = n >> 4;
    const int v = col >> 3, q = col & 7;
    return kH[i][2 * (v & 1) + kSlotD[q][0]] * kH[j][2 * ((v >> 1) & 1) + kSlotD[q][1]] * kH[k][2 * ((v >> 2) & 1) + kSlotD[q][2]];
}
SDF_HD int fitCoefRuntime(int n, int col) {      // same entry, evaluated at run time (fills the MFMA B operand)
    const int h[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {-3, -2, 3, -1}, {2, 1, -2, 1}};
    const int sd[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
    const int i = n & 3, j = (n >> 2) & 3, k = n >> 4, v = col >> 3, q = col & 7;
    return h[i][2 * (v & 1) + sd[q][0]] * h[j][2 * ((v >> 1) & 1) + sd[q][1]] * h[k][2 * ((v >> 2) & 1) + sd[q][2]];
}
constexpr int fitFirstCol(int n) {
    for (int c = 0; c < 64; c++) if (fitCoef(n, c) != 0) return c;
    return 64;
}

template <int N, int COL>
SDF_DEV void fitTerm(const float (&s)[64], float& acc) {
    constexpr int c = fitCoef(N, COL);
    if constexpr (c != 0) {
        if constexpr (COL == fitFirstCol(N)) acc = (float)c * s[COL];
        else acc = acc + (float)c * s[COL];
    }
}
template <int N, int... COLS>
SDF_DEV float fitRow(const float (&s)[64], std::integer_sequence<int, COLS...>) {
    float acc = 0.0f;
    (fitTerm<N, COLS>(s, acc), ...);
    return acc;
}
template <int... NS>
SDF_DEV void fitAllRows(const float (&s)[64], float (&out)[64], std::integer_sequence<int, NS...>) {
    ((out[NS] = fitRow<NS>(s, std::make_integer_sequence<int, 64>{})), ...);
}

// s: 8 vertices x 8 slots, world-space derivatives; scaled in place by nodeSize powers, then fitted.
SDF_DEV void tricubicFit(float (&s)[64], float nodeSize, float (&out)[64]) {
    const float sq = nodeSize * nodeSize;
    const float cu = sq * nodeSize;
#pragma unroll
    for (int v = 0; v < 8; v++) {
        s[8 * v + 1] *= nodeSize; s[8 * v + 2] *= nodeSize; s[8 * v + 3] *= nodeSize;
        s[8 * v + 4] *= sq; s[8 * v + 5] *= sq; s[8 * v + 6] *= sq;
        s[8 * v + 7] *= cu;
    }
    fitAllRows(s, out, std::make_integer_sequence<int, 64>{});
}

// ---- literal-order evaluation ("EXACT") -------------------------------------------------------------------
template <typename CF>   // CF: callable n -> coefficient
SDF_HD float tricubicValueLiteral(CF c, F3 f) {
    float acc = 0.0f;
#pragma unroll
    for (int n = 0; n < 64; n++) {
        const int i = n & 3, j = (n >> 2) & 3, k = n >> 4;
        float t = c(n);
#pragma unroll
        for (int a = 0; a < i; a++) t = t * f.x;
#pragma unroll
        for (int a = 0; a < j; a++) t = t * f.y;
#pragma unroll
        for (int a = 0; a < k; a++) t = t * f.z;
        acc = acc + t;
    }
    return acc;
}

// interpolateValue of the reference's SDFLIB_USE_ENOKI=ON flavour (InterpolationMethods.h:383-430, the CMake default): power vectors
// x1 = (1, x, x x, (x x) x), x2 = y x1, x3 = y x2, x4 = y x3; every z-slab adds dot(x1, c[16k..]) + dot(x2, ..) + dot(x3, ..) + dot(x4, ..)
// and the vectors are then multiplied by z; enoki::dot of two 4-vectors = (a0 b0 + a1 b1) + (a2 b2 + a3 b3) (DPPS's order, and that of
// Enoki's generic hsum(a * b)).  Enoki's headers are not in the image: restated from those semantics, unpinned.
template <typename CF>
SDF_HD float tricubicValueEnoki(CF c, F3 f) {
    float x[4][4];
    x[0][0] = 1.0f; x[0][1] = f.x; x[0][2] = f.x * f.x; x[0][3] = f.x * f.x * f.x;
#pragma unroll
    for (int j = 1; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) x[j][i] = f.y * x[j - 1][i];
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (k > 0) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) x[j][i] = f.z * x[j][i];
        }
        float d[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int b = 16 * k + 4 * j;
            d[j] = (x[j][0] * c(b) + x[j][1] * c(b + 1)) + (x[j][2] * c(b + 2) + x[j][3] * c(b + 3));
        }
        const float slab = d[0] + d[1] + d[2] + d[3];
        sum = (k == 0) ? slab : sum + slab;
    }
    return sum;
}
// What the reference's interpolateValue computes in the flavour this library is built for: libsdfhip.so = SDFLIB_USE_ENOKI OFF (the
// literal order), libsdfhip_enoki.so (-DSDFHIP_ENOKI_ORDER) = ON.  Callers: the subdivision rules, getDistance, the minimum border value,
// the CONTINUITY builder's own estimate.  interpolateVertexValues spells its value out literally in BOTH flavours (:459-464) and calls
// tricubicValueLiteral.
template <typename CF>
SDF_HD float tricubicValueExact(CF c, F3 f) {
#ifdef SDFHIP_ENOKI_ORDER
    return tricubicValueEnoki(c, f);
#else
    return tricubicValueLiteral(c, f);
#endif
}

template <int EX, int EY, int EZ, typename CF>
SDF_HD float tricubicDerivExact(CF c, F3 f) {
    float acc = 0.0f;
    bool first = true;
#pragma unroll
    for (int n = 0; n < 64; n++) {
        const int i = n & 3, j = (n >> 2) & 3, k = n >> 4;
        const int fac = (EX ? i : 1) * (EY ? j : 1) * (EZ ? k : 1);
        if (fac == 0) continue;
        float t = (float)fac * c(n);
#pragma unroll
        for (int a = 0; a < i - EX; a++) t = t * f.x;
#pragma unroll
        for (int a = 0; a < j - EY; a++) t = t * f.y;
#pragma unroll
        for (int a = 0; a < k - EZ; a++) t = t * f.z;
        if (first) { acc = t; first = false; } else acc = acc + t;
    }
    return acc;
}

// ---- separable Horner with FMA ("FAST"): same polynomial, different rounding (<= 1e-5 abs in practice) -------
template <typename CF>
SDF_HD float tricubicValueFast(CF c, F3 f) {
    float zacc = 0.f;
#pragma unroll
    for (int k = 3; k >= 0; k--) {
        float yacc = 0.f;
#pragma unroll
        for (int j = 3; j >= 0; j--) {
            const int b = 4 * j + 16 * k;
            const float xacc = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(c(b + 3), f.x, c(b + 2)), f.x, c(b + 1)), f.x, c(b));
            yacc = __builtin_fmaf(yacc, f.y, xacc);
        }
        zacc = __builtin_fmaf(zacc, f.z, yacc);
    }
    return zacc;
}
// value and gradient together: for each (j,k) row evaluate p(x) and p'(x), then two nested Horner passes.
template <typename CF>
SDF_HD float tricubicValueGradFast(CF c, F3 f, F3& g) {
    float v = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int k = 3; k >= 0; k--) {
        float yv = 0.f, ygx = 0.f, ygy = 0.f;
#pragma unroll
        for (int j = 3; j >= 0; j--) {
            const int b = 4 * j + 16 * k;
            const float c0 = c(b), c1 = c(b + 1), c2 = c(b + 2), c3 = c(b + 3);
            const float px = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(c3, f.x, c2), f.x, c1), f.x, c0);
            const float dx = __builtin_fmaf(__builtin_fmaf(3.f * c3, f.x, 2.f * c2), f.x, c1);
            ygy = __builtin_fmaf(ygy, f.y, yv);      // d/dy of the running Horner (uses the previous yv)
            yv = __builtin_fmaf(yv, f.y, px);
            ygx = __builtin_fmaf(ygx, f.y, dx);
        }
        gz = __builtin_fmaf(gz, f.z, v);
        v = __builtin_fmaf(v, f.z, yv);
        gx = __builtin_fmaf(gx, f.z, ygx);
        gy = __builtin_fmaf(gy, f.z, ygy);
    }
    g = F3{gx, gy, gz};
    return v;
}

// ---- the 27-point stencil ------------------------------------------------------------------------------------
// Grid point g = gx + 3 gy + 9 gz (g? in {0,1,2}); the 19 non-corner points in ascending g are the reference's
// nodeSamplePoints (src/sdf/OctreeSdfDepthFirst.h:139-162); corners are vertex bx + 2 by + 4 bz.
constexpr int midGrid(int m) {
    int cnt = 0;
    for (int g = 0; g < 27; g++) {
        const int gx = g % 3, gy = (g / 3) % 3, gz = g / 9;
        if (gx != 1 && gy != 1 && gz != 1) continue;
        if (cnt == m) return g;
        cnt++;
    }
    return -1;
}
constexpr int gridToMid(int g) {
    int cnt = 0;
    for (int h = 0; h < 27; h++) {
        const int gx = h % 3, gy = (h / 3) % 3, gz = h / 9;
        if (gx != 1 && gy != 1 && gz != 1) { if (h == g) return -1; continue; }
        if (h == g) return cnt;
        cnt++;
    }
    return -1;
}
// source of vertex j of child c: >= 0 -> mid-point index ; < 0 -> -(parent vertex) - 1
constexpr int childSrc(int c, int j) {
    const int gx = (c & 1) + (j & 1), gy = ((c >> 1) & 1) + ((j >> 1) & 1), gz = ((c >> 2) & 1) + ((j >> 2) & 1);
    const int g = gx + 3 * gy + 9 * gz;
    const int m = gridToMid(g);
    return m >= 0 ? m : -((gx >> 1) + 2 * (gy >> 1) + 4 * (gz >> 1)) - 1;
}
struct StencilTables {
    float relx[19], rely[19], relz[19], weight[19];
    int src[8][8];
};
constexpr StencilTables makeStencil() {
    StencilTables t{};
    for (int m = 0; m < 19; m++) {
        const int g = midGrid(m);
        const int gx = g % 3, gy = (g / 3) % 3, gz = g / 9;
        t.relx[m] = (float)(gx - 1); t.rely[m] = (float)(gy - 1); t.relz[m] = (float)(gz - 1);
        t.weight[m] = (float)(1 << ((gx == 1) + (gy == 1) + (gz == 1)));
    }
    for (int c = 0; c < 8; c++) for (int j = 0; j < 8; j++) t.src[c][j] = childSrc(c, j);
    return t;
}
static constexpr StencilTables kStencil = makeStencil();          // host / compile-time view
static __constant__ const StencilTables kStencilDev = makeStencil();   // device view (run-time indexing)

// mid-point m -> grid index (the 19 non-corner points of the 3x3x3 stencil in ascending order)
static __constant__ const unsigned char kMidGridTab[19] = {1, 3, 4, 5, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17, 19, 21, 22, 23, 25};
static_assert(midGrid(0) == 1 && midGrid(4) == 7 && midGrid(9) == 13 && midGrid(14) == 19 && midGrid(18) == 25, "stencil order");

SDF_DEV F3 midRel(int m) {
    const int g = kMidGridTab[m];
    return F3{(float)(g % 3 - 1), (float)((g / 3) % 3 - 1), (float)(g / 9 - 1)};
}
SDF_DEV float midWeight(int m) {
    const int g = kMidGridTab[m];
    return (float)(1 << (((g % 3) == 1) + (((g / 3) % 3) == 1) + ((g / 9) == 1)));
}

// Error estimate of the active rule from the 64 coefficients and the 19 exact mid-point distances
// (OctreeSdfUtils.h:60-85 trapezoid, :213-238 Simpson, :87-138 by-distance).
template <typename CF, typename MF>   // CF: n -> coefficient, MF: m -> exact distance at mid-point m
SDF_DEV float ruleValue(int rule, CF c, MF mid, float param1) {
    if (rule == 0) return INFINITY;
    float acc = 0.0f;
#pragma unroll 1
    for (int m = 0; m < 19; m++) {
        const F3 r = midRel(m);
        const F3 f = F3{0.5f * r.x + 0.5f, 0.5f * r.y + 0.5f, 0.5f * r.z + 0.5f};
        const float w = midWeight(m);
        const float v = tricubicValueExact(c, f);
        float term;
        if (rule == 1) { const float e = mid(m) - v; term = (w / 64.0f) * (e * e); }
        else if (rule == 2) { const float e = mid(m) - v; term = ((w * w) / 216.0f) * (e * e); }
        else { const float e = gmax(fabsf(mid(m) - v) - param1 * fabsf(v), 0.0f); term = (w / 64.0f) * (e * e); }
        acc = (m == 0) ? term : acc + term;
    }
    return acc;
}

}  // namespace sdfhip
