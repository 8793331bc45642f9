// Minimal glm types for the SdfLib-compatible C++ API when the real glm is not installed.
// Layout-compatible with glm::vec3 / glm::ivec3 (three packed scalars); only what the API surface needs.
#ifndef SDFLIB_GLM_COMPAT_H
#define SDFLIB_GLM_COMPAT_H
#if defined(__has_include)
#  if __has_include(<glm/glm.hpp>)
#    include <glm/glm.hpp>
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
}  // namespace glm
#endif
#endif
