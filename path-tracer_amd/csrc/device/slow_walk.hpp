// slow_walk.hpp -- one ray, one WAVEFRONT: the exact walk for the handful of rays per launch that the order-free
// traversal hands back (a direction component that is exactly zero, a verdict that failed: kernels_trace_wide.hip).
// The threaded walk (cs488.h:1661-1705) visits a node exactly when its parent's box was hit, so the wavefront keeps a
// stack of nodes still to be visited, pops up to 64 of them at a time, tests their boxes with the reference's IEEE
// predicate (NaN quotients included: nothing here relies on nesting), pushes both children of every interior node that
// was hit and tests the triangles of every leaf that was hit.  Visit order is not the walk's, so ties in t go to the
// smaller leafTris index, which IS the walk's order.  Only for trees that build_wide has verified (s.nodeRight exists
// only then).  The stack cannot overflow: above capacity - kSlowReserve entries a step pops ONE node, which is a plain
// depth-first walk and adds at most one entry per level (binary depth <= 96 < kSlowReserve - 128: DevScene::slowOk,
// checked when the scene is made); below it a step pops at most 64 and pushes at most 128.
// Users: k_trace_slow (kernels_trace.hip: a launch of its own, 32 KB of stack per wavefront) and the end of k_trace_wide
// (the wavefronts that finish first trace the handed-back rays on the 4 KB their lanes' stacks occupy).
#pragma once
#include "traverse.hpp"

namespace ptd {

constexpr unsigned kSlowMaxRays = 4096;  // lists up to this long are traced a wavefront per ray, longer ones by list-mode k_trace, a lane per ray
constexpr int kSlowReserve = 256;  // stack entries kept free for one-node-at-a-time steps

// order within one wavefront: its LDS operations complete in program order, the compiler must keep them there
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// stack: entry i lives at stack[(i >> 6) * ROW + (i & 63)] (ROW = 64: contiguous; ROW = block size: the rows of
// k_trace_wide's per-lane stacks, `stack` pointing at the wavefront's first lane); capacity >= kSlowReserve + 256.
// Every lane of the wavefront calls this with the same ray; the result is valid in lane 0.
template <bool ANYHIT, bool COUNT, int ROW>
__device__ __forceinline__ void slow_walk_wave(const DevScene& s, const vec3 o, const vec3 d, const float tMin, const float tMax,
                                               int* stack, const int capacity, const int lane, int& bestTri, int& bestSeq,
                                               float& bestT, float& bestB, float& bestG, unsigned& nv, unsigned& nt) {
    const unsigned long long below = (1ull << lane) - 1ull;
    const __attribute__((address_space(1))) float4v* nodesGlobal = (const __attribute__((address_space(1))) float4v*)s.nodes;
    const __attribute__((address_space(1))) float4v* trisGlobal = (const __attribute__((address_space(1))) float4v*)s.leafTris;
#define PT_SLOW_AT(i) stack[((i) >> 6) * ROW + ((i) & 63)]
    bestTri = -1; bestSeq = 0x7fffffff;
    bestT = 3.402823466e+38f; bestB = 0.0f; bestG = 0.0f;
    int top = s.nRoots;  // <= 64 (DevScene::slowOk): the root of every object's tree; the order is free
    if (lane < top) PT_SLOW_AT(lane) = s.roots[lane];
    wave_lds_sync();
    while (top > 0) {
        const int take = top > capacity - kSlowReserve ? 1 : (top < 64 ? top : 64);
        int node = -1;
        if (lane < take) node = PT_SLOW_AT(top - 1 - lane);
        top -= take;
        wave_lds_sync();  // every pop is done before a push reuses its slot
        int left = 0, right = 0;
        if (node >= 0) {
            const float4v n0 = nodesGlobal[2 * (size_t)node], n1 = nodesGlobal[2 * (size_t)node + 1];
            if (COUNT) nv++;
            if (slab_hit_ieee(n0, n1, o, d)) {
                const int link = f2i(n1.w);
                if (link > 0) {
                    left = link;
                    right = s.nodeRight[node];
                } else {
                    for (int cur = ~link;; ++cur) {  // the leaf's triangles (cs488.h:1671-1675)
                        const float4v q0 = trisGlobal[3 * (size_t)cur], q1 = trisGlobal[3 * (size_t)cur + 1],
                                      q2 = trisGlobal[3 * (size_t)cur + 2];
                        if (COUNT) nt++;
                        float t, be, ga;
                        if (tri_test(vec3{q0.x, q0.y, q0.z}, vec3{q1.x, q1.y, q1.z}, vec3{q2.x, q2.y, q2.z}, o, d, tMin, tMax,
                                     t, be, ga) &&
                            (t < bestT || (t == bestT && cur < bestSeq))) {
                            bestTri = f2i(q0.w); bestSeq = cur; bestT = t; bestB = be; bestG = ga;
                        }
                        if (f2i(q1.w)) break;
                    }
                }
            }
        }
        if (ANYHIT && __ballot(bestTri >= 0) != 0ull) break;  // isHit: any accepted triangle decides
        const unsigned long long mPush = __ballot(left > 0);
        if (left > 0) {
            const int at = top + 2 * __popcll(mPush & below);
            PT_SLOW_AT(at) = right;
            PT_SLOW_AT(at + 1) = left;
        }
        top += 2 * __popcll(mPush);
        wave_lds_sync();
    }
#undef PT_SLOW_AT
    // the wavefront's answer: smallest t, ties to the smaller leafTris index
    for (int off = 32; off > 0; off >>= 1) {
        const float t2 = __shfl_down(bestT, off), b2 = __shfl_down(bestB, off), g2 = __shfl_down(bestG, off);
        const int tri2 = __shfl_down(bestTri, off), seq2 = __shfl_down(bestSeq, off);
        if (tri2 >= 0 && (bestTri < 0 || t2 < bestT || (t2 == bestT && seq2 < bestSeq))) {
            bestTri = tri2; bestSeq = seq2; bestT = t2; bestB = b2; bestG = g2;
        }
    }
}

}  // namespace ptd
