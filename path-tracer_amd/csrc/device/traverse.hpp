// traverse.hpp -- device functions of the threaded-BVH walk: slab test,
// Cramer ray-triangle test.  One ray per lane; the stepping policy of a
// wavefront lives in kernels_trace.hip.
//
// Bit-exactness contract (tests compare hit ids / t / barycentrics bitwise
// with the CPU oracle): fp32 only, the reference's operation order, IEEE
// division results (never a plain reciprocal multiply), no FMA contraction of
// the reference's expressions (-ffp-contract=off), compare-and-swap chains
// instead of fmin/fmax (NaN behaviour differs).
// Reference: cs488.h:571-608 (AABB::intersect), :777-813 (raytraceTriangle),
// :1661-1705 (StacklessBVH::intersect / isHit).
#pragma once
#include "../pt/vec.hpp"
#include "device_scene.hpp"

namespace ptd {

using pt::vec3;

__device__ __forceinline__ int f2i(float f) { return __float_as_int(f); }
__device__ __forceinline__ float i2f(int i) { return __int_as_float(i); }

// ---- correctly rounded a / d from a correctly rounded reciprocal ----------
// y = RN(1/d) is computed ONCE per ray and axis with a true IEEE division.
// q0 = RN(a*y) is within 2 ulp of a/d; one residual step makes it faithful
// (q1), a second one rounds it correctly (Markstein, IBM J. R&D 34(1), 1990:
// with y = RN(1/d), q faithful and r = a - q*d exact, RN(q + r*y) = RN(a/d)).
// Preconditions, guaranteed by the callers (fast_div_ok): d finite, nonzero,
// 2^-40 <= |d| <= 2^40, and a == 0 or 2^-63 <= |a| <= 2^41, so that no
// intermediate overflows, underflows or becomes subnormal.  The sequence is
// 1 mul + 4 fma; hipcc expands an IEEE `/` into ~11 instructions incl. a
// quarter-rate v_rcp_f32.  tests/test_gpu_trace.py::test_fast_division_is_ieee
// checks it against `/` on 2^33 operand pairs incl. the structured corner cases.
__device__ __forceinline__ float div_by_recip(float a, float d, float y) {
    const float q0 = a * y;
    const float r0 = __builtin_fmaf(-q0, d, a);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-q1, d, a);
    return __builtin_fmaf(r1, y, q1);
}
// magnitude is 0 or inside [2^lo, 2^hi]
__device__ __forceinline__ bool mag_zero_or_within(float v, int lo, int hi) {
    const unsigned e = ((unsigned)f2i(v) >> 23) & 0xffu;
    return (((unsigned)f2i(v) << 1) == 0u) || (e >= (unsigned)(127 + lo) && e <= (unsigned)(127 + hi));
}
__device__ __forceinline__ bool mag_within(float v, int lo, int hi) {
    const unsigned e = ((unsigned)f2i(v) >> 23) & 0xffu;
    return e >= (unsigned)(127 + lo) && e <= (unsigned)(127 + hi);
}
// may this ray use div_by_recip in the slab test?  (scene planes are checked on the host: DevScene::planesTame)
__device__ __forceinline__ bool fast_div_ok(const vec3 o, const vec3 d) {
    return mag_zero_or_within(o.x, -40, 40) && mag_zero_or_within(o.y, -40, 40) && mag_zero_or_within(o.z, -40, 40) &&
           mag_within(d.x, -40, 40) && mag_within(d.y, -40, 40) && mag_within(d.z, -40, 40);
}

// AABB::intersect, cs488.h:571-608 (only the boolean is needed by the threaded walk), literal form.
__device__ __forceinline__ bool slab_hit_ieee(const float4 lo, const float4 hi, const vec3 o, const vec3 d) {
    float tx1 = (lo.x - o.x) / d.x;
    float ty1 = (lo.y - o.y) / d.y;
    float tz1 = (lo.z - o.z) / d.z;
    float tx2 = (hi.x - o.x) / d.x;
    float ty2 = (hi.y - o.y) / d.y;
    float tz2 = (hi.z - o.z) / d.z;
    if (tx1 > tx2) { const float tmp = tx1; tx1 = tx2; tx2 = tmp; }
    if (ty1 > ty2) { const float tmp = ty1; ty1 = ty2; ty2 = tmp; }
    if (tz1 > tz2) { const float tmp = tz1; tz1 = tz2; tz2 = tmp; }
    float t1 = tx1; if (t1 < ty1) t1 = ty1; if (t1 < tz1) t1 = tz1;
    float t2 = tx2; if (t2 > ty2) t2 = ty2; if (t2 > tz2) t2 = tz2;
    if (t1 > t2) return false;
    if ((t1 < 0.0f) && (t2 < 0.0f)) return false;
    return true;
}
// The same predicate for a ray that satisfies fast_div_ok in a scene with tame planes.  All six quotients are
// then finite and bit-identical to the IEEE ones (div_by_recip), which allows two exact shortcuts:
//  * lo <= hi and RN subtraction/division are monotone, so after the reference's swap the smaller quotient of
//    an axis is the lo-plane's when d > 0 and the hi-plane's when d < 0 (equal quotients: the swap is moot):
//    the planes are selected by the ray's direction signs instead of comparing and swapping;
//  * without NaNs the if-chains are max3 / min3 (a +-0 tie cannot change any of the comparisons that follow).
struct SlabRay {
    vec3 o, d, y;
    bool nx, ny, nz;  // d.x < 0, ...
};
__device__ __forceinline__ bool slab_hit_fast(const float4 lo, const float4 hi, const SlabRay& r) {
    const float nearx = div_by_recip((r.nx ? hi.x : lo.x) - r.o.x, r.d.x, r.y.x);
    const float neary = div_by_recip((r.ny ? hi.y : lo.y) - r.o.y, r.d.y, r.y.y);
    const float nearz = div_by_recip((r.nz ? hi.z : lo.z) - r.o.z, r.d.z, r.y.z);
    const float farx = div_by_recip((r.nx ? lo.x : hi.x) - r.o.x, r.d.x, r.y.x);
    const float fary = div_by_recip((r.ny ? lo.y : hi.y) - r.o.y, r.d.y, r.y.y);
    const float farz = div_by_recip((r.nz ? lo.z : hi.z) - r.o.z, r.d.z, r.y.z);
    const float t1 = __builtin_fmaxf(__builtin_fmaxf(nearx, neary), nearz);
    const float t2 = __builtin_fminf(__builtin_fminf(farx, fary), farz);
    return !(t1 > t2) && !((t1 < 0.0f) && (t2 < 0.0f));
}

// raytraceTriangle, cs488.h:782-812, on the precomputed edges a = p0-p1, b = p0-p2.
// Returns true when the hit is ACCEPTED (inside + t window); t/beta/gamma are
// then the candidate's values.
__device__ __forceinline__ bool tri_test(const vec3 p0, const vec3 a, const vec3 b, const vec3 o, const vec3 c,
                                         const float tMin, const float tMax, float& t, float& beta, float& gamma) {
    const vec3 d = p0 - o;
    const float D = pt::det3(a, b, c);
    const float A = pt::det3(d, b, c);
    const float B = pt::det3(a, d, c);
    const float C = pt::det3(a, b, d);
    beta = A / D;
    gamma = B / D;
    const float alpha = 1.0f - beta - gamma;
    if (alpha > 1.0f || alpha < 0.0f || beta < 0.0f || gamma < 0.0f) return false;
    t = C / D;
    return t > tMin && t < tMax && (double)t > 0.00001;
}

}  // namespace ptd
