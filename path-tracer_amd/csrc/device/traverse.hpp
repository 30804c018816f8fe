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
// 2^-40 <= |d| <= 2^40, and a == 0 or 2^-84 <= |a| <= 2^41.  Then y, q0, q1
// and q are normal (|q| >= 2^-124), nothing overflows (|q| <= 2^81), and the
// residuals -- multiples of ulp(q)*ulp(d) >= 2^-131 that fit 24 bits -- are
// exact (the kernels run with fp32 denormals enabled, denorm mode 3).
// a = plane - o reaches this with o and the plane each 0 or of magnitude in
// [2^-61, 2^40]: a nonzero difference of two such floats is a multiple of
// ulp(2^-61) = 2^-84.  The sequence is
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
constexpr int kTameLo = -61, kTameHi = 40;  // origins and box planes: 0 or magnitude in [2^kTameLo, 2^kTameHi]
constexpr int kDirLo = -40, kDirHi = 40;    // direction components: magnitude in [2^kDirLo, 2^kDirHi]
constexpr int kNumLo = -84, kNumHi = 41;    // resulting numerators: 0 or magnitude in [2^kNumLo, 2^kNumHi]
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
    return mag_zero_or_within(o.x, kTameLo, kTameHi) && mag_zero_or_within(o.y, kTameLo, kTameHi) &&
           mag_zero_or_within(o.z, kTameLo, kTameHi) && mag_within(d.x, kDirLo, kDirHi) &&
           mag_within(d.y, kDirLo, kDirHi) && mag_within(d.z, kDirLo, kDirHi);
}

// the same predicate with a lower bound of its own for the direction components (2^dirLo <= |d| <= 2^kDirHi, dirLo >=
// kDirLo), evaluated WITHOUT short-circuit branches: six range checks joined by bitwise ands are a dozen compares and scalar
// ands, where the && chain is five nested branches on the execution mask (k_trace_wide's refill runs it for every ray)
__device__ __forceinline__ bool fast_div_ok_flat(const float ox, const float oy, const float oz, const float dx, const float dy,
                                                 const float dz, const int dirLo) {
    const unsigned ex = ((unsigned)f2i(ox) >> 23) & 0xffu, ey = ((unsigned)f2i(oy) >> 23) & 0xffu, ez = ((unsigned)f2i(oz) >> 23) & 0xffu;
    const unsigned fx = ((unsigned)f2i(dx) >> 23) & 0xffu, fy = ((unsigned)f2i(dy) >> 23) & 0xffu, fz = ((unsigned)f2i(dz) >> 23) & 0xffu;
    const unsigned span = (unsigned)(kTameHi - kTameLo), lo = (unsigned)(127 + kTameLo);
    const int okx = (int)(((unsigned)f2i(ox) << 1) == 0u) | (int)(ex - lo <= span);
    const int oky = (int)(((unsigned)f2i(oy) << 1) == 0u) | (int)(ey - lo <= span);
    const int okz = (int)(((unsigned)f2i(oz) << 1) == 0u) | (int)(ez - lo <= span);
    const unsigned dspan = (unsigned)(kDirHi - dirLo), dlo = (unsigned)(127 + dirLo);
    const int okd = (int)(fx - dlo <= dspan) & (int)(fy - dlo <= dspan) & (int)(fz - dlo <= dspan);
    return (okx & oky & okz & okd) != 0;
}

// native vector types: loadable through address-space-qualified pointers (LDS / global), and the packed fp32 math
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

// AABB::intersect, cs488.h:571-608 (only the boolean is needed by the threaded walk), literal form.
// Node record (device_scene.hpp): n0 = {min.x, max.x, min.y, max.y}, n1 = {min.z, max.z, idMiss, link}.
__device__ __forceinline__ bool slab_hit_ieee(const float4v n0, const float4v n1, const vec3 o, const vec3 d) {
    float tx1 = (n0.x - o.x) / d.x;
    float ty1 = (n0.z - o.y) / d.y;
    float tz1 = (n1.x - o.z) / d.z;
    float tx2 = (n0.y - o.x) / d.x;
    float ty2 = (n0.w - o.y) / d.y;
    float tz2 = (n1.y - o.z) / d.z;
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
//  * without NaNs the reference's swap of an axis' two quotients is min / max of the pair, and its if-chains are
//    max3 / min3 (a +-0 tie cannot change any of the comparisons that follow);
//  * the (min, max) planes of an axis sit in adjacent registers, so the subtraction and the five steps of the
//    division run as packed fp32 instructions (v_pk_add / v_pk_mul / v_pk_fma_f32: two lanes of IEEE fp32 per
//    issue slot, each rounding exactly like the scalar instruction) -- 18 packed instructions for six quotients.
__device__ __forceinline__ float2v div_by_recip2(const float2v a, const float d, const float y) {
    const float2v dd = {d, d}, yy = {y, y};
    const float2v q0 = a * yy;
    const float2v r0 = __builtin_elementwise_fma(-q0, dd, a);
    const float2v q1 = __builtin_elementwise_fma(r0, yy, q0);
    const float2v r1 = __builtin_elementwise_fma(-q1, dd, a);
    return __builtin_elementwise_fma(r1, yy, q1);
}
// o, d: the ray; y = RN(1/d) per component.  (Scalars by value: a struct of them, passed by reference, was left
// on the stack by the compiler.)
__device__ __forceinline__ bool slab_hit_fast(const float4v n0, const float4v n1, const float ox, const float oy,
                                              const float oz, const float dx, const float dy, const float dz,
                                              const float yx, const float yy, const float yz) {
    const float2v qx = div_by_recip2(float2v{n0.x, n0.y} - float2v{ox, ox}, dx, yx);
    const float2v qy = div_by_recip2(float2v{n0.z, n0.w} - float2v{oy, oy}, dy, yy);
    const float2v qz = div_by_recip2(float2v{n1.x, n1.y} - float2v{oz, oz}, dz, yz);
    const float t1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(qx.x, qx.y), __builtin_fminf(qy.x, qy.y)),
                                     __builtin_fminf(qz.x, qz.y));
    const float t2 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(qx.x, qx.y), __builtin_fmaxf(qy.x, qy.y)),
                                     __builtin_fmaxf(qz.x, qz.y));
    // !(t1 > t2) && !(t1 < 0 && t2 < 0) for finite t1, t2: when t1 <= t2 the second clause is just !(t2 < 0), so
    // the whole predicate is max(t1, 0) <= t2 (a -0 compares like +0 on either side) -- one compare instead of three
    return !(__builtin_fmaxf(t1, 0.0f) > t2);
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
    // `t > 0.00001` in the reference compares (double)t with the double 1e-5 (cs488.h:812).  (float)1e-5 =
    // 9.99999974738e-06 lies BELOW 1e-5 and the next float, 1.00000006569e-05, above it, so for every float t
    // (double)t > 1e-5  <=>  t > (float)1e-5: the same predicate without the two fp64 instructions.
    static_assert((double)0.00001f < 0.00001, "the float nearest to 1e-5 must lie below it");
    return t > tMin && t < tMax && t > 0.00001f;
}

}  // namespace ptd
