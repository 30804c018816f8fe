// kernels_trace_bittrail.hip -- StacklessBVH2::intersect (cs488.h:1732-1803), the traversal the reference's Scene
// instantiates (cs488.h:2008), literally: near-first descent that tests BOTH children of a node, culls a child whose
// box entry distance is not below the best hit so far, records "the other child was hit too" in a 32-bit trail and
// comes back up through idParent / idSibling.  One ray per lane; nothing is reordered or batched, so the box tests
// and triangle tests a ray performs -- and their counts -- are the reference's own (tests compare them with the
// oracle's counters).  This is the second traversal of SURVEY 8(a) row 9 / 8(f).4: a parity kernel, selected with
// mipt_context_set_traversal(ctx, MIPT_TRAVERSAL_BITTRAIL); the fast path is k_trace_wide, whose answers equal this
// kernel's bit for bit on every ray tested.  Only for trees the host has verified (a proper binary tree, depth <=
// 31: beyond that the reference's own trail overflows); a step budget ends any walk that a damaged link table
// could otherwise keep going.
#include "device_scene.hpp"
#include "kernels.hpp"
#include "traverse.hpp"

namespace ptd {

using pt::vec3;

// AABB::intersect (cs488.h:571-608) with its output: minHit.t = t1, written only when the box is hit
__device__ __forceinline__ bool box_hit_t(const float4v n0, const float4v n1, const vec3 o, const vec3 d, float& tOut) {
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
    tOut = t1;
    return true;
}

__device__ __forceinline__ bool live_slot(const QueueView& q, unsigned i) {
    if (!q.counts) return true;
    const unsigned sh = i / q.regionSize;
    return (i - sh * q.regionSize) < q.counts[(size_t)sh * kCntStride];
}

template <bool ANYHIT, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace_bittrail(DevScene s, const float4* __restrict__ rays, const int* __restrict__ index,
                                                           const int* __restrict__ target, QueueView q, float4* __restrict__ hits,
                                                           unsigned char* __restrict__ occ, unsigned long long* __restrict__ counters) {
    const float4v* nodes = (const float4v*)s.nodes;
    const float4v* tris = (const float4v*)s.leafTris;
    unsigned nv = 0, nt = 0;
    const long long budget = 4ll * s.nNodes + 64;  // every node is entered at most once going down and once coming up
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < q.space; i += gridDim.x * kBlock) {
        if (!live_slot(q, i)) continue;
        const unsigned slot = i;
        const unsigned raySlot = index ? (unsigned)index[slot] : slot;
        const size_t rs = ANYHIT ? (size_t)slot : (size_t)raySlot;  // shadow rays live in queue order
        const float4 r0 = rays[2 * rs], r1 = rays[2 * rs + 1];
        const vec3 o{r0.x, r0.y, r0.z}, d{r1.x, r1.y, r1.z};
        const float tMin = r0.w, tMax = r1.w;
        // Scene::intersect / isHit (C:2056-2085): every object's BVH in turn, each traversal starting afresh (its own minHit,
        // its own trail); between objects the strict '<' of C:2078, and isHit stops at the first object that reports a hit
        bool sceneHit = false;
        int sceneTri = -1;
        float sceneT = 3.402823466e+38f, sceneB = 0.0f, sceneG = 0.0f;  // minHit.t = FLT_MAX, C:2073
        for (int ob = 0; ob < s.nRoots; ob++) {
        const int root = s.roots[ob];
        int node = root, trail = 0;
        bool hit = false;
        int bestTri = -1;
        float minT = 3.402823466e+38f, bestB = 0.0f, bestG = 0.0f;  // minHit.t = FLT_MAX, C:1738
        float tTmp = 0.0f;
        long long steps = 0;
        if (COUNT) nv++;
        bool walking = box_hit_t(nodes[2 * (size_t)root], nodes[2 * (size_t)root + 1], o, d, tTmp);  // C:1743
        while (walking) {
            // descend until you can't, C:1749-1771
            int link = f2i(nodes[2 * (size_t)node + 1].w);
            while (link > 0 && ++steps < budget) {
                const int L = link, R = s.btLinks[node].x;
                float tL = 0.0f, tR = 0.0f;
                bool hit1 = box_hit_t(nodes[2 * (size_t)L], nodes[2 * (size_t)L + 1], o, d, tL);
                bool hit2 = box_hit_t(nodes[2 * (size_t)R], nodes[2 * (size_t)R + 1], o, d, tR);
                if (COUNT) nv += 2;
                hit1 = hit1 && (tL < minT);
                hit2 = hit2 && (tR < minT);
                if (hit1 || hit2) trail = (int)((unsigned)trail << 1);
                if (hit1 && hit2) {
                    trail ^= 1;
                    node = (tL < tR) ? L : R;
                } else if (hit1) {
                    node = L;
                } else if (hit2) {
                    node = R;
                } else {
                    break;
                }
                link = f2i(nodes[2 * (size_t)node + 1].w);
            }
            if (link < 0) {  // a leaf, C:1773-1779
                for (int cur = ~link;; ++cur) {
                    const float4v q0 = tris[3 * (size_t)cur], q1 = tris[3 * (size_t)cur + 1], q2 = tris[3 * (size_t)cur + 2];
                    if (COUNT) nt++;
                    float t, be, ga;
                    if (tri_test(vec3{q0.x, q0.y, q0.z}, vec3{q1.x, q1.y, q1.z}, vec3{q2.x, q2.y, q2.z}, o, d, tMin, tMax, t, be, ga)) {
                        hit = true;
                        if (t < minT) { minT = t; bestTri = f2i(q0.w); bestB = be; bestG = ga; }
                    }
                    if (f2i(q1.w)) break;
                }
            }
            // go up until there is a sibling to go to, C:1780-1799
            while (true) {
                if (trail == 0 || ++steps >= budget) { walking = false; break; }
                while ((trail & 1) == 0) {
                    trail >>= 1;  // arithmetic, as on the int of the reference
                    node = s.btLinks[node].y;
                }
                node = s.btLinks[node].z;
                trail ^= 1;
                if (COUNT) nv++;
                box_hit_t(nodes[2 * (size_t)node], nodes[2 * (size_t)node + 1], o, d, tTmp);  // "newhit", always true
                if (tTmp < minT) break;
            }
        }
        if (hit && minT < sceneT) { sceneHit = true; sceneTri = bestTri; sceneT = minT; sceneB = bestB; sceneG = bestG; }  // C:2077-2081
        if (ANYHIT && hit) { sceneHit = true; break; }  // C:2062-2065
        }
        if (ANYHIT) {
            if (sceneHit) occ[target ? (unsigned)target[slot] : slot] = 1;  // Scene::isHit, C:2056-2068: intersect() != false
        } else {
            hits[raySlot] = make_float4(i2f(sceneHit ? sceneTri : -1), sceneHit ? sceneT : 3.402823466e+38f, sceneB, sceneG);
        }
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {
            nv += __shfl_down(nv, off);
            nt += __shfl_down(nt, off);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&counters[0], (unsigned long long)nv);
            atomicAdd(&counters[1], (unsigned long long)nt);
        }
    }
}

static inline unsigned bt_grid(size_t n) {
    size_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > 256 * 32) g = 256 * 32;
    return (unsigned)g;
}
void launch_trace_bittrail_closest(hipStream_t st, const DevScene& s, const float4* rays, const int* index, QueueView q,
                                   float4* hits, unsigned long long* counters) {
    if (counters)
        hipLaunchKernelGGL((k_trace_bittrail<false, true>), dim3(bt_grid(q.space)), dim3(kBlock), 0, st, s, rays, index, nullptr, q, hits, nullptr, counters);
    else
        hipLaunchKernelGGL((k_trace_bittrail<false, false>), dim3(bt_grid(q.space)), dim3(kBlock), 0, st, s, rays, index, nullptr, q, hits, nullptr, counters);
}
void launch_trace_bittrail_occluded(hipStream_t st, const DevScene& s, const float4* rays, const int* target, QueueView q,
                                    unsigned char* occ, unsigned long long* counters) {
    if (counters)
        hipLaunchKernelGGL((k_trace_bittrail<true, true>), dim3(bt_grid(q.space)), dim3(kBlock), 0, st, s, rays, nullptr, target, q, nullptr, occ, counters);
    else
        hipLaunchKernelGGL((k_trace_bittrail<true, false>), dim3(bt_grid(q.space)), dim3(kBlock), 0, st, s, rays, nullptr, target, q, nullptr, occ, counters);
}

}  // namespace ptd
