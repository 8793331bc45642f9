// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
//
// CPU restatement of the arithmetic SdfLib's hot path performs through glm.
// glm is a third-party dependency of the reference that is NOT vendored under /root/reference
// (g-truc/glm @ 89e52e327d7a3ae61eb402850ba36ac4dd111987 "0.9.8", reference libs/CMakeLists.txt:6-9),
// so the operation ORDER of each helper below restates glm 0.9.8's published implementation:
//   dot(vec3)      : tmp = a*b ; tmp.x + tmp.y + tmp.z           (glm/detail/func_geometric.inl)
//   cross          : (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)
//   normalize      : v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x)
//   mat3 * vec3    : m[0][r]*v.x + m[1][r]*v.y + m[2][r]*v.z     (glm/detail/type_mat3x3.inl)
//   inverse(mat3)  : cofactors * OneOverDeterminant              (glm/detail/func_matrix.inl)
//   sign           : (0 < x) - (x < 0) ; min(a,b) = (b<a)?b:a ; max(a,b) = (a<b)?b:a ; fract = x - floor(x)
// PARITY UNPINNED at the ulp level FOR THIS FILE ONLY: the reference holds no test that pins results at the glm boundary, and glm's source is
// not here to compare with.  Everything that CALLS these helpers (orc_triangle.h, orc_exact.h, orc_octree.h ...) is pinned to the reference's
// text by tools/refpin (tests/test_ref_text_pin.py): same operations, same operands, same order - with these operators as vocabulary.
// Must be compiled with -ffp-contract=off (no FMA), like the reference's default x86-64 build.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstddef>

namespace orc {

struct V2 { float x, y; };
struct V3 {
    float x, y, z;
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
struct M3 { V3 c[3]; };   // column-major like glm: c[col][row]

static inline V3 v3(float a, float b, float c) { return V3{a, b, c}; }
static inline V3 v3(float a) { return V3{a, a, a}; }
static inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
static inline V3 operator+(V3 a, float s) { return V3{a.x + s, a.y + s, a.z + s}; }
static inline V3 operator-(V3 a, float s) { return V3{a.x - s, a.y - s, a.z - s}; }
static inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
static inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
static inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
static inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
static inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
static inline V3& operator-=(V3& a, V3 b) { a = a - b; return a; }

static inline float dot(V3 a, V3 b) { V3 t = a * b; return t.x + t.y + t.z; }
static inline float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
static inline V3 cross(V3 x, V3 y) {
    return V3{x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
static inline float length(V3 v) { return std::sqrt(dot(v, v)); }
static inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
static inline V2 normalize(V2 v) { float s = 1.0f / std::sqrt(dot(v, v)); return V2{v.x * s, v.y * s}; }

static inline float gmin(float a, float b) { return (b < a) ? b : a; }
static inline float gmax(float a, float b) { return (a < b) ? b : a; }
static inline uint32_t gmin(uint32_t a, uint32_t b) { return (b < a) ? b : a; }
static inline uint32_t gmax(uint32_t a, uint32_t b) { return (a < b) ? b : a; }
struct IV3 { int x, y, z; };
static inline IV3 iv3(V3 v) { return IV3{(int)v.x, (int)v.y, (int)v.z}; }   // glm::ivec3(vec3): per-component conversion
static inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
static inline float gsign(float x) { return float(0.0f < x) - float(x < 0.0f); }
static inline float gfract(float x) { return x - std::floor(x); }
static inline V3 gfract(V3 v) { return V3{gfract(v.x), gfract(v.y), gfract(v.z)}; }
static inline V3 gabs(V3 v) { return V3{std::fabs(v.x), std::fabs(v.y), std::fabs(v.z)}; }
static inline V3 gmax(V3 a, V3 b) { return V3{gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)}; }

static inline V3 mul(const M3& m, V3 v) {
    return V3{m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z,
              m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z,
              m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z};
}
// transpose(m) * v  == rows of m dotted (in glm's mat*vec order) with v
static inline V3 mulT(const M3& m, V3 v) {
    return V3{m.c[0].x * v.x + m.c[0].y * v.y + m.c[0].z * v.z,
              m.c[1].x * v.x + m.c[1].y * v.y + m.c[1].z * v.z,
              m.c[2].x * v.x + m.c[2].y * v.y + m.c[2].z * v.z};
}
static inline M3 inverse(const M3& m) {
    const float m00 = m.c[0].x, m01 = m.c[0].y, m02 = m.c[0].z;
    const float m10 = m.c[1].x, m11 = m.c[1].y, m12 = m.c[1].z;
    const float m20 = m.c[2].x, m21 = m.c[2].y, m22 = m.c[2].z;
    const float ood = 1.0f / (+m00 * (m11 * m22 - m21 * m12) - m10 * (m01 * m22 - m21 * m02) + m20 * (m01 * m12 - m11 * m02));
    M3 r;
    r.c[0].x = +(m11 * m22 - m21 * m12) * ood;
    r.c[1].x = -(m10 * m22 - m20 * m12) * ood;
    r.c[2].x = +(m10 * m21 - m20 * m11) * ood;
    r.c[0].y = -(m01 * m22 - m21 * m02) * ood;
    r.c[1].y = +(m00 * m22 - m20 * m02) * ood;
    r.c[2].y = -(m00 * m21 - m20 * m01) * ood;
    r.c[0].z = +(m01 * m12 - m11 * m02) * ood;
    r.c[1].z = -(m00 * m12 - m10 * m02) * ood;
    r.c[2].z = +(m00 * m11 - m10 * m01) * ood;
    return r;
}

}  // namespace orc
