// vec.hpp -- the handful of fp32 vector operations the ray/scene path needs,
// with the reference's exact operation order (bit-exactness of hit records
// depends on it).  Shared by host C++ and gfx950 device code; build with
// -ffp-contract=off so no multiply-add is fused on either side.
//
// Reference semantics restated (L: = reference cpu/src/linalg.h):
//   dot       L:501 via sum/fold L:367,L:401  -> ((0 + x) + y) + z
//   cross     L:500
//   normalize L:503-504  -> a / sqrt(dot(a,a)), three IEEE divides
//   mul(mat3,vec3) L:536 -> c0*v.x + c1*v.y + c2*v.z
//   determinant(mat3) L:567 (columns a,b,c)
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD inline
#endif

namespace pt {

struct vec3 {
    float x, y, z;
};
struct vec2 {
    float x, y;
};

PT_HD vec3 mk3(float x, float y, float z) { return vec3{x, y, z}; }
PT_HD vec3 operator+(vec3 a, vec3 b) { return vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_HD vec3 operator-(vec3 a, vec3 b) { return vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_HD vec3 operator-(vec3 a) { return vec3{-a.x, -a.y, -a.z}; }
PT_HD vec3 operator*(vec3 a, vec3 b) { return vec3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_HD vec3 operator*(vec3 a, float s) { return vec3{a.x * s, a.y * s, a.z * s}; }
PT_HD vec3 operator*(float s, vec3 a) { return vec3{s * a.x, s * a.y, s * a.z}; }
PT_HD vec3 operator/(vec3 a, float s) { return vec3{a.x / s, a.y / s, a.z / s}; }

PT_HD float dot(vec3 a, vec3 b) { return ((0.0f + a.x * b.x) + a.y * b.y) + a.z * b.z; }
PT_HD float dot(vec2 a, vec2 b) { return (0.0f + a.x * b.x) + a.y * b.y; }
PT_HD vec3 cross(vec3 a, vec3 b) {
    return vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
PT_HD float length2(vec3 a) { return dot(a, a); }
PT_HD vec3 normalize(vec3 a) { return a / sqrtf(dot(a, a)); }
PT_HD vec2 normalize(vec2 a) {
    float l = sqrtf(dot(a, a));
    return vec2{a.x / l, a.y / l};
}
// determinant of the 3x3 matrix whose columns are a, b, c
PT_HD float det3(vec3 a, vec3 b, vec3 c) {
    return a.x * (b.y * c.z - c.y * b.z) + a.y * (b.z * c.x - c.z * b.x) + a.z * (b.x * c.y - c.x * b.y);
}
// mul(mat(c0,c1,c2), v)
PT_HD vec3 mulcols(vec3 c0, vec3 c1, vec3 c2, vec3 v) { return (c0 * v.x + c1 * v.y) + c2 * v.z; }
PT_HD vec2 mulcols(vec2 c0, vec2 c1, vec2 c2, vec3 v) {
    return vec2{(c0.x * v.x + c1.x * v.y) + c2.x * v.z, (c0.y * v.x + c1.y * v.y) + c2.y * v.z};
}

}  // namespace pt
