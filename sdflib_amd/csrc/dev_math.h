// Device-side geometry primitives of the SDF engine (gfx950).  PRODUCT code — independent of oracle/.
//
// Bit-parity contract: every fp32 expression below is evaluated in the operation order the reference obtains
// through glm 0.9.8 (dot = (x*x' + y*y') + z*z', normalize = v * (1/sqrt(dot)), mat3*vec3 column sums,
// cofactor inverse), with IEEE sqrt/div and NO fused multiply-add: this translation unit must be compiled
// with -ffp-contract=off (hipcc defaults to 'fast'), see sdflib_amd/csrc/Makefile.  Fast paths that want FMA
// call __builtin_fmaf explicitly.
//
// Reference: include/SdfLib/utils/TriangleUtils.h:20-376 (TriangleData and the point/triangle functions).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SDF_DEV __device__ __forceinline__
#define SDF_HD __host__ __device__ __forceinline__

// XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs of the MI355X (block b -> XCD b % 8), each with a private
// 4 MB L2.  When consecutive LOGICAL blocks work on neighbouring data (sorted queries, lattice rows, Morton-ordered points),
// giving XCD x the x-th contiguous eighth of the logical blocks keeps each L2's working set to one region of the data.
// Launch with xcdGrid(blocks) blocks (a multiple of 8) and guard the tail in the kernel.
__device__ __forceinline__ unsigned xcdLogicalBlock() { const unsigned per = gridDim.x >> 3; return (blockIdx.x & 7u) * per + (blockIdx.x >> 3); }
static inline unsigned xcdGrid(unsigned blocks) { return (blocks + 7u) / 8u * 8u; }

namespace sdfhip {

struct F3 { float x, y, z; };
struct F2 { float x, y; };

SDF_HD F3 f3(float a, float b, float c) { return F3{a, b, c}; }
SDF_HD F3 operator+(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
SDF_HD F3 operator-(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
SDF_HD F3 operator*(F3 a, F3 b) { return F3{a.x * b.x, a.y * b.y, a.z * b.z}; }
SDF_HD F3 operator*(F3 a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
SDF_HD F3 operator*(float s, F3 a) { return F3{s * a.x, s * a.y, s * a.z}; }
SDF_HD F3 operator/(F3 a, float s) { return F3{a.x / s, a.y / s, a.z / s}; }
SDF_HD F3 operator+(F3 a, float s) { return F3{a.x + s, a.y + s, a.z + s}; }
SDF_HD F3 operator-(F3 a, float s) { return F3{a.x - s, a.y - s, a.z - s}; }
SDF_HD F3 operator-(F3 a) { return F3{-a.x, -a.y, -a.z}; }
SDF_HD float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
SDF_HD float dot(F2 a, F2 b) { return a.x * b.x + a.y * b.y; }
SDF_HD F3 cross(F3 x, F3 y) { return F3{x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
SDF_HD F3 normalize(F3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
SDF_HD F2 normalize(F2 v) { const float s = 1.0f / sqrtf(dot(v, v)); return F2{v.x * s, v.y * s}; }
SDF_HD float length(F3 v) { return sqrtf(dot(v, v)); }
SDF_HD float gmin(float a, float b) { return (b < a) ? b : a; }
SDF_HD float gmax(float a, float b) { return (a < b) ? b : a; }
SDF_HD float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }

// acosf as glibc computes it (sysdeps/ieee754/flt-32/e_acosf.c, the fdlibm float routine: a rational approximation on |x| < 0.5, the
// half-angle identities with a split square root beyond), operation for operation in fp32 with IEEE sqrt and division — so that the
// corner angles of the vertex pseudonormals (TriangleUtils.cpp:85-86 calls glm::acos = the platform's acosf) are the ones the reference
// gets on such a platform WITHOUT a round trip to the host.  libm's acosf is not correctly rounded, so this is an identity of
// algorithms, not of specifications: sdfhip_test_acosf_mismatches (tests/test_abi.py) compares the host compilation of this very
// function with the running libm on EVERY float of [-1, 1] (2 130 706 434 values: none differs on glibc 2.35), the GPU tests compare
// the device compilation with libm; on a platform whose libm is another algorithm the first mesh's self-check (ctx_mesh.hip, hostAcosNeeded) sends the arc cosines to the host's libm instead.
SDF_HD float acosfGlibc(float x) {
    const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f;
    const float qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const int hx = (int)__builtin_bit_cast(unsigned, x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {                      // |x| < 0.5
        if (ix <= 0x23000000) return pio2_hi + pio2_lo;
        const float z = x * x;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx < 0) {                               // x < -0.5
        const float z = (one + x) * 0.5f;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float s = sqrtf(z);
        const float r = p / q;
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float z = (one - x) * 0.5f;           // x > 0.5
    const float s = sqrtf(z);
    const float df = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    const float w = r * s + c;
    return 2.0f * (df + w);
}
SDF_HD float gsign(float x) { return (float)(0.0f < x) - (float)(x < 0.0f); }
SDF_HD float gfract(float x) { return x - floorf(x); }

// TriangleData: 37 floats, field order of the reference struct (TriangleUtils.h:56-71).
//  [0..2] origin, [3..11] transform (column-major: m[col][row] at 3+3*col+row), [12,13] b, [14,15] c,
//  [16] v2, [17,18] v3, [19..27] edgesNormal[3], [28..36] verticesNormal[3]
constexpr int TD_FLOATS = 37;

struct TriFrame {            // the part every distance evaluation needs (19 floats)
    F3 origin;
    float m[9];              // m[3*col+row]
    F2 b, c;
    float v2;
    F2 v3;
};

SDF_HD F3 mulM(const float* m, F3 v) {
    return F3{m[0] * v.x + m[3] * v.y + m[6] * v.z,
              m[1] * v.x + m[4] * v.y + m[7] * v.z,
              m[2] * v.x + m[5] * v.y + m[8] * v.z};
}
SDF_HD F3 mulMT(const float* m, F3 v) {
    return F3{m[0] * v.x + m[1] * v.y + m[2] * v.z,
              m[3] * v.x + m[4] * v.y + m[5] * v.z,
              m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
SDF_HD F3 triNormal(const float* m) { return F3{m[2], m[5], m[8]}; }

SDF_HD void loadFrame(const float* __restrict__ td, TriFrame& f) {
    f.origin = F3{td[0], td[1], td[2]};
#pragma unroll
    for (int i = 0; i < 9; i++) f.m[i] = td[3 + i];
    f.b = F2{td[12], td[13]}; f.c = F2{td[14], td[15]};
    f.v2 = td[16]; f.v3 = F2{td[17], td[18]};
}

// Packed copy of the 19 frame floats (+1 pad) = 80 B = 5 aligned dwordx4 per triangle: what the brute-force distance
// loops gather, instead of 19 scalar loads out of the 148-B TriangleData records.
constexpr int FRAME_FLOATS = 20;
SDF_DEV void loadFramePacked(const float* __restrict__ frames, uint32_t t, TriFrame& f) {
    const float4* p = reinterpret_cast<const float4*>(frames) + 5 * (size_t)t;
    const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    f.origin = F3{a.x, a.y, a.z};
    f.m[0] = a.w; f.m[1] = b.x; f.m[2] = b.y; f.m[3] = b.z; f.m[4] = b.w; f.m[5] = c.x; f.m[6] = c.y; f.m[7] = c.z; f.m[8] = c.w;
    f.b = F2{d.x, d.y}; f.c = F2{d.z, d.w};
    f.v2 = e.x; f.v3 = F2{e.y, e.z};
}

// Build the frame of a triangle (TriangleData ctor, TriangleUtils.h:23-42); writes td[0..18] and the default
// pseudonormals (0,0,1) into td[19..36].
SDF_HD void makeTriangleData(F3 p1, F3 p2, F3 p3, float* td) {
    const F3 sx = normalize(p2 - p1);
    const F3 sz = normalize(cross(p2 - p1, p3 - p1));
    const F3 sy = cross(sz, sx);
    const float m00 = sx.x, m01 = sx.y, m02 = sx.z, m10 = sy.x, m11 = sy.y, m12 = sy.z, m20 = sz.x, m21 = sz.y, m22 = sz.z;
    const float ood = 1.0f / (+m00 * (m11 * m22 - m21 * m12) - m10 * (m01 * m22 - m21 * m02) + m20 * (m01 * m12 - m11 * m02));
    float m[9];
    m[0] = +(m11 * m22 - m21 * m12) * ood;   // [0][0]
    m[3] = -(m10 * m22 - m20 * m12) * ood;   // [1][0]
    m[6] = +(m10 * m21 - m20 * m11) * ood;   // [2][0]
    m[1] = -(m01 * m22 - m21 * m02) * ood;   // [0][1]
    m[4] = +(m00 * m22 - m20 * m02) * ood;   // [1][1]
    m[7] = -(m00 * m21 - m20 * m01) * ood;   // [2][1]
    m[2] = +(m01 * m12 - m11 * m02) * ood;   // [0][2]
    m[5] = -(m00 * m12 - m10 * m02) * ood;   // [1][2]
    m[8] = +(m00 * m11 - m10 * m01) * ood;   // [2][2]
    td[0] = p1.x; td[1] = p1.y; td[2] = p1.z;
#pragma unroll
    for (int i = 0; i < 9; i++) td[3 + i] = m[i];
    F3 e = mulM(m, p3 - p2);
    const F2 b = normalize(F2{e.x, e.y});
    e = mulM(m, p1 - p3);
    const F2 c = normalize(F2{e.x, e.y});
    td[12] = b.x; td[13] = b.y; td[14] = c.x; td[15] = c.y;
    td[16] = mulM(m, p2 - p1).x;
    e = mulM(m, p3 - p1);
    td[17] = e.x; td[18] = e.y;
#pragma unroll
    for (int k = 0; k < 6; k++) { td[19 + 3 * k] = 0.f; td[20 + 3 * k] = 0.f; td[21 + 3 * k] = 1.f; }
}

enum Region { R_V1 = 0, R_V2, R_V3, R_E1, R_E2, R_E3, R_F };

struct Proj { F3 p; float de1, de2, de3; int r; };

SDF_HD Proj classify(F3 point, const TriFrame& d) {
    Proj o;
    o.p = mulM(d.m, point - d.origin);
    const F3 p = o.p;
    o.de1 = -p.y;
    o.de2 = (p.x - d.v2) * d.b.y - p.y * d.b.x;
    o.de3 = p.x * d.c.y - p.y * d.c.x;
    if (o.de1 >= 0) {
        if (p.x <= 0) o.r = R_V1;
        else if (p.x >= d.v2) o.r = R_V2;
        else o.r = R_E1;
    } else if (o.de2 >= 0) {
        if ((p.x - d.v2) * d.b.x + p.y * d.b.y <= 0) o.r = R_V2;
        else if ((p.x - d.v3.x) * d.b.x + (p.y - d.v3.y) * d.b.y >= 0) o.r = R_V3;
        else o.r = R_E2;
    } else if (o.de3 >= 0) {
        if (p.x * d.c.x + p.y * d.c.y >= 0) o.r = R_V1;
        else if ((p.x - d.v3.x) * d.c.x + (p.y - d.v3.y) * d.c.y <= 0) o.r = R_V3;
        else o.r = R_E3;
    } else o.r = R_F;
    return o;
}

SDF_HD float sqDistFromProj(const Proj& o, const TriFrame& d) {
    const F3 p = o.p;
    switch (o.r) {
        case R_V1: return dot(p, p);
        case R_V2: { const F3 q = p - F3{d.v2, 0.f, 0.f}; return dot(q, q); }
        case R_V3: { const F3 q = p - F3{d.v3.x, d.v3.y, 0.f}; return dot(q, q); }
        case R_E1: return o.de1 * o.de1 + p.z * p.z;
        case R_E2: return o.de2 * o.de2 + p.z * p.z;
        case R_E3: return o.de3 * o.de3 + p.z * p.z;
        default: return p.z * p.z;
    }
}

// getSqDistPointAndTriangle(point, data)  (TriangleUtils.h:76-135)
SDF_HD float sqDistPointTriangle(F3 point, const TriFrame& d) { return sqDistFromProj(classify(point, d), d); }

// The same value without branches: every region's expression is evaluated (the very operations of classify + sqDistFromProj) and the
// region tests pick one.  Brute-force loops run it with 64 lanes on 64 different triangles, where the branchy form executes most
// regions' code one after the other anyway; comparisons with NaN fall through exactly as the if / else chain does.
SDF_HD float sqDistPointTriangleSelect(F3 point, const TriFrame& d) {
    const F3 p = mulM(d.m, point - d.origin);
    const float px2 = p.x - d.v2, qx3 = p.x - d.v3.x, qy3 = p.y - d.v3.y;
    const float de1 = -p.y;
    const float de2 = px2 * d.b.y - p.y * d.b.x;
    const float de3 = p.x * d.c.y - p.y * d.c.x;
    const float xx = p.x * p.x, yy = p.y * p.y, zz = p.z * p.z;
    const float dV1 = xx + yy + zz;
    const float dV2 = px2 * px2 + yy + zz;
    const float dV3 = qx3 * qx3 + qy3 * qy3 + zz;
    const float dE1 = yy + zz;                       // de1 * de1 = (-p.y) * (-p.y)
    const float dE2 = de2 * de2 + zz, dE3 = de3 * de3 + zz;
    const float tb2 = px2 * d.b.x + p.y * d.b.y, tb3 = qx3 * d.b.x + qy3 * d.b.y;
    const float tc1 = p.x * d.c.x + p.y * d.c.y, tc3 = qx3 * d.c.x + qy3 * d.c.y;
    const float dA = (p.x <= 0) ? dV1 : ((p.x >= d.v2) ? dV2 : dE1);
    const float dB = (tb2 <= 0) ? dV2 : ((tb3 >= 0) ? dV3 : dE2);
    const float dC = (tc1 >= 0) ? dV1 : ((tc3 <= 0) ? dV3 : dE3);
    return (de1 >= 0) ? dA : ((de2 >= 0) ? dB : ((de3 >= 0) ? dC : zz));
}

// Pseudonormal that signs region r; q = the vector it is dotted with.
SDF_HD float regionSign(const Proj& o, const TriFrame& d, const float* __restrict__ td) {
    const F3 p = o.p;
    switch (o.r) {
        case R_V1: return gsign(dot(F3{td[28], td[29], td[30]}, p));
        case R_V2: return gsign(dot(F3{td[31], td[32], td[33]}, p - F3{d.v2, 0.f, 0.f}));
        case R_V3: return gsign(dot(F3{td[34], td[35], td[36]}, p - F3{d.v3.x, d.v3.y, 0.f}));
        case R_E1: return gsign(dot(F3{td[19], td[20], td[21]}, p));
        case R_E2: return gsign(dot(F3{td[22], td[23], td[24]}, p - F3{d.v2, 0.f, 0.f}));
        case R_E3: return gsign(dot(F3{td[25], td[26], td[27]}, p));
        default: return 1.0f;
    }
}

// getSignedDistPointAndTriangle(point, data)  (TriangleUtils.h:137-196)
SDF_HD float signedDistPointTriangle(F3 point, const float* __restrict__ td) {
    TriFrame d; loadFrame(td, d);
    const Proj o = classify(point, d);
    if (o.r == R_F) return o.p.z;
    return regionSign(o, d, td) * sqrtf(sqDistFromProj(o, d));
}

// getSignedDistPointAndTriangle(point, data, v1, v2, v3, outNormal)  (TriangleUtils.h:198-290): the variant
// TriCubicInterpolation::calculatePointValues uses; a NaN direction falls back to the triangle normal.
SDF_HD float signedDistPointTriangleGrad(F3 point, const float* __restrict__ td, F3 w1, F3 w2, F3 w3, F3& outN) {
    TriFrame d; loadFrame(td, d);
    const Proj o = classify(point, d);
    const F3 p = o.p;
    if (o.r == R_F) { outN = triNormal(d.m); return p.z; }
    const float s = regionSign(o, d, td);
    F3 dirv;
    switch (o.r) {
        case R_V1: dirv = point - w1; break;
        case R_V2: dirv = point - w2; break;
        case R_V3: dirv = point - w3; break;
        case R_E1: dirv = mulMT(d.m, F3{0.f, p.y, p.z}); break;
        case R_E2: { const float t = (p.x - d.v2) * d.b.x + p.y * d.b.y; dirv = mulMT(d.m, F3{(p.x - d.v2) - t * d.b.x, p.y - t * d.b.y, p.z}); break; }
        default:   { const float t = p.x * d.c.x + p.y * d.c.y; dirv = mulMT(d.m, F3{p.x - t * d.c.x, p.y - t * d.c.y, p.z}); break; }
    }
    F3 n = normalize(dirv);
    const float chk = n.x + n.y + n.z;
    if (chk != chk) n = triNormal(d.m);
    outN = s * n;
    return s * sqrtf(sqDistFromProj(o, d));
}

// getSignedDistPointAndTriangle(point, data, outNormal)  (TriangleUtils.h:292-376): ExactOctreeSdf's gradient query.
SDF_HD float signedDistPointTriangleGradLocal(F3 point, const float* __restrict__ td, F3& outN) {
    TriFrame d; loadFrame(td, d);
    const Proj o = classify(point, d);
    const F3 p = o.p;
    if (o.r == R_F) { outN = triNormal(d.m); return p.z; }
    const float s = regionSign(o, d, td);
    F3 dirv;
    switch (o.r) {
        case R_V1: dirv = point - d.origin; break;
        case R_V2: dirv = point - d.origin - mulMT(d.m, F3{d.v2, 0.f, 0.f}); break;
        case R_V3: dirv = point - d.origin - mulMT(d.m, F3{d.v3.x, d.v3.y, 0.f}); break;
        case R_E1: dirv = mulMT(d.m, F3{0.f, p.y, p.z}); break;
        case R_E2: { const float t = (p.x - d.v2) * d.b.x + p.y * d.b.y; dirv = mulMT(d.m, F3{(p.x - d.v2) - t * d.b.x, p.y - t * d.b.y, p.z}); break; }
        default:   { const float t = p.x * d.c.x + p.y * d.c.y; dirv = mulMT(d.m, F3{p.x - t * d.c.x, p.y - t * d.c.y, p.z}); break; }
    }
    outN = s * normalize(dirv);
    return s * sqrtf(sqDistFromProj(o, d));
}

}  // namespace sdfhip
