// Minimal glm types for the SdfLib-compatible C++ API when the real glm is not installed.
// Layout-compatible with glm::vec3 / glm::ivec3 (three packed scalars); only what the API surface needs.
#ifndef SDFLIB_GLM_COMPAT_H
#define SDFLIB_GLM_COMPAT_H
#if defined(__has_include)
#  if __has_include(<glm/glm.hpp>)
#    include <glm/glm.hpp>
#    include <glm/gtc/matrix_transform.hpp>
#    define SDFLIB_HAVE_GLM 1
#  endif
#endif
#ifndef SDFLIB_HAVE_GLM
#include <cmath>
namespace glm {
template <typename T> struct tvec3 {
    T x, y, z;
    tvec3() : x(0), y(0), z(0) {}
    explicit tvec3(T a) : x(a), y(a), z(a) {}
    tvec3(T a, T b, T c) : x(a), y(b), z(c) {}
    T& operator[](int i) { return (&x)[i]; }
    const T& operator[](int i) const { return (&x)[i]; }
};
typedef tvec3<float> vec3;
typedef tvec3<int> ivec3;
inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline float max(float a, float b) { return (a < b) ? b : a; }
inline float min(float a, float b) { return (b < a) ? b : a; }
// vec4 / mat4 (column major) with glm 0.9.8's operation order, enough for the tools' model normalisation
// (scale(mat4(1), s) * translate(mat4(1), -centre), src/tools/SdfExporter/main.cpp:83-90, applied by Mesh::applyTransform)
struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(vec3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
};
inline vec4 operator+(vec4 a, vec4 b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator*(vec4 a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
struct mat4 {
    vec4 c[4];
    mat4() {}
    explicit mat4(float d) { c[0] = vec4(d, 0, 0, 0); c[1] = vec4(0, d, 0, 0); c[2] = vec4(0, 0, d, 0); c[3] = vec4(0, 0, 0, d); }
    vec4& operator[](int i) { return c[i]; }
    const vec4& operator[](int i) const { return c[i]; }
};
inline vec4 operator*(const mat4& m, vec4 v) { return (m[0] * v.x + m[1] * v.y) + (m[2] * v.z + m[3] * v.w); }
inline mat4 operator*(const mat4& a, const mat4& b) {
    mat4 r;
    for (int k = 0; k < 4; k++) r[k] = a[0] * b[k].x + a[1] * b[k].y + a[2] * b[k].z + a[3] * b[k].w;
    return r;
}
inline mat4 translate(const mat4& m, vec3 v) { mat4 r = m; r[3] = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3]; return r; }
inline mat4 scale(const mat4& m, vec3 v) { mat4 r; r[0] = m[0] * v.x; r[1] = m[1] * v.y; r[2] = m[2] * v.z; r[3] = m[3]; return r; }
}  // namespace glm
#endif
#endif
