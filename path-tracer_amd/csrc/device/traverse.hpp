// traverse.hpp -- device functions of the threaded-BVH walk: slab test,
// Cramer ray-triangle test, closest-hit and any-hit loops.  One ray per lane.
//
// Bit-exactness contract (tests compare hit ids / t / barycentrics bitwise
// with the CPU oracle): fp32 only, the reference's operation order, IEEE
// divide (never a reciprocal multiply), no FMA contraction (-ffp-contract=off),
// compare-and-swap chains instead of fmin/fmax (NaN behaviour differs).
// Reference: cs488.h:571-608 (AABB::intersect), :777-813 (raytraceTriangle),
// :1661-1705 (StacklessBVH::intersect / isHit).
#pragma once
#include "../pt/vec.hpp"
#include "device_scene.hpp"

namespace ptd {

using pt::vec3;

__device__ __forceinline__ int f2i(float f) { return __float_as_int(f); }
__device__ __forceinline__ float i2f(int i) { return __int_as_float(i); }

// AABB::intersect, cs488.h:571-608 (only the boolean is needed by the threaded walk)
__device__ __forceinline__ bool slab_hit(const float4 lo, const float4 hi, const vec3 o, const vec3 d) {
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

struct Hit {
    int tri;  // original triangle id, -1 = miss
    float t, beta, gamma;
};

// StacklessBVH::intersect (closest hit, strict '<' in visit order) when
// ANYHIT is false; StacklessBVH::isHit (first accepted triangle) when true.
template <bool ANYHIT, bool COUNT>
__device__ __forceinline__ Hit walk(const float4* __restrict__ nodes, const float4* __restrict__ leafTris, const vec3 o,
                                    const vec3 d, const float tMin, const float tMax, unsigned& nodeVisits,
                                    unsigned& triTests) {
    Hit best;
    best.tri = -1;
    best.t = 3.402823466e+38f;
    best.beta = 0.0f;
    best.gamma = 0.0f;
    int node = 0;
    do {
        const float4 lo = nodes[2 * node];
        const float4 hi = nodes[2 * node + 1];
        if (COUNT) nodeVisits++;
        const int miss = f2i(lo.w);
        if (slab_hit(lo, hi, o, d)) {
            const int link = f2i(hi.w);
            if (link < 0) {
                int k = ~link;
                int last;
                do {
                    const float4 q0 = leafTris[3 * k];
                    const float4 q1 = leafTris[3 * k + 1];
                    const float4 q2 = leafTris[3 * k + 2];
                    last = f2i(q1.w);
                    if (COUNT) triTests++;
                    float t, be, ga;
                    if (tri_test(vec3{q0.x, q0.y, q0.z}, vec3{q1.x, q1.y, q1.z}, vec3{q2.x, q2.y, q2.z}, o, d, tMin,
                                 tMax, t, be, ga)) {
                        if (ANYHIT) {
                            best.tri = f2i(q0.w);
                            best.t = t;
                            best.beta = be;
                            best.gamma = ga;
                            return best;
                        }
                        if (t < best.t) {
                            best.tri = f2i(q0.w);
                            best.t = t;
                            best.beta = be;
                            best.gamma = ga;
                        }
                    }
                    ++k;
                } while (!last);
                node = miss;  // a leaf's idHit is its idMiss
            } else {
                node = link;
            }
        } else {
            node = miss;
        }
    } while (node != 0);
    return best;
}

}  // namespace ptd
