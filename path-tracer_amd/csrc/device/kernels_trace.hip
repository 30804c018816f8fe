// kernels_trace.hip -- the traversal kernels (closest-hit "extend", any-hit
// "occluded") and the small kernels around the parity entry points: eye-ray
// generation, HitInfo resolution, environment lookup.
//
// Launch shape: 256-thread blocks (4 wavefronts of 64 lanes), one ray per
// lane at a time.  Queue lengths live in device memory (sharded counters), so
// a whole frame is enqueued without host syncs.
#include "kernels.hpp"
#include "shading.hpp"

namespace ptd {

// ---------------------------------------------------------------------------
// k_trace<ANYHIT, COUNT>: the threaded walk of StacklessBVH::intersect /
// ::isHit (cs488.h:1661-1705) for a queue of rays.
//
// Every ray visits exactly the nodes and triangles the reference visits, in
// the reference's order (the links are fixed, nothing is culled by distance),
// so results are bit-identical.  What is scheduled freely is WHEN a lane takes
// its next step, to keep the 64 lanes of a wavefront busy (the first version,
// one ray per lane until the slowest lane finished, ran with 16 of 64 lanes
// active per VALU instruction -- profiles/r01):
//   * persistent wavefronts with lane refill: a lane whose ray is finished
//     takes the next ray of the queue (chunks of 64 indices per atomic);
//   * a lane that reaches a leaf parks there; the wavefront votes per step
//     between "box step" (lanes at a node) and "triangle step" (parked lanes),
//     so both code paths run with many lanes instead of interleaving;
//   * the six slab quotients use a per-ray reciprocal with a residual
//     correction that reproduces IEEE division exactly (traverse.hpp).
// Queue: dense (q.counts == nullptr, q.space rays) or sharded (kernels.hpp);
// the shard counters are prefix-summed into LDS so that rays are addressed by
// a dense index.  index (nullable): queue slot -> ray slot; the result goes to
// hits[ray slot] (closest hit) or occ[target[queue slot]] (any hit).
constexpr int kRefillLanes = 16;  // refill when this many lanes are empty
constexpr int kLeafLanes = 12;    // triangle step when this many lanes are parked at a leaf
constexpr int kBoxBurst = 4;      // box steps between two votes

template <bool ANYHIT, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace(DevScene s, const float4* __restrict__ rays,
                                                  const int* __restrict__ index, const int* __restrict__ target,
                                                  QueueView q, unsigned* __restrict__ cursor,
                                                  float4* __restrict__ hits, unsigned char* __restrict__ occ,
                                                  unsigned long long* __restrict__ counters) {
    __shared__ unsigned prefix[kShards + 1];
    const int lane = threadIdx.x & 63;
    if (q.counts) {
        if (threadIdx.x < 64) {
            unsigned v = q.counts[(size_t)lane * kCntStride];
            for (int off = 1; off < 64; off <<= 1) {  // inclusive scan over the 64 shard counts
                const unsigned up = __shfl_up(v, off);
                if (lane >= off) v += up;
            }
            prefix[lane + 1] = v;
            if (lane == 0) prefix[0] = 0;
        }
        __syncthreads();
    }
    const unsigned total = q.counts ? prefix[kShards] : q.space;

    // lane state
    enum { EMPTY = 0, SEARCH = 1, LEAF = 2 };
    int state = EMPTY;
    int node = 0, k = 0, cont = 0;
    unsigned raySlot = 0, queueSlot = 0;
    SlabRay r{{0, 0, 0}, {0, 0, 1}, {0, 0, 1}, false, false, false};
    float tMin = 0.0f, tMax = 0.0f;
    bool fast = false;
    int bestTri = -1;
    float bestT = 3.402823466e+38f, bestB = 0.0f, bestG = 0.0f;
    unsigned nv = 0, nt = 0;
    // wavefront-uniform chunk of ray indices
    unsigned chunkBase = 0, chunkLeft = 0;
    bool exhausted = (total == 0);

    auto finish = [&]() {  // the ray of this lane is done: publish its result, free the lane
        if (ANYHIT) {
            if (bestTri >= 0) occ[target ? (unsigned)target[queueSlot] : queueSlot] = 1;
        } else {
            hits[raySlot] = make_float4(i2f(bestTri), bestT, bestB, bestG);
        }
        state = EMPTY;
    };
    // one node for a searching lane (cs488.h:1669-1680)
    auto box_step = [&]() {
        const float4 lo = s.nodes[2 * (size_t)node];
        const float4 hi = s.nodes[2 * (size_t)node + 1];
        if (COUNT) nv++;
        const int miss = f2i(lo.w);
        const bool hit = fast ? slab_hit_fast(lo, hi, r) : slab_hit_ieee(lo, hi, r.o, r.d);
        if (hit) {
            const int link = f2i(hi.w);
            if (link < 0) { state = LEAF; k = ~link; cont = miss; }
            else node = link;
        } else {
            node = miss;
        }
        if (state == SEARCH && node == 0) finish();
    };
    // one triangle for a parked lane (cs488.h:1671-1675 / 1693-1697)
    auto tri_step = [&]() {
        const float4 q0 = s.leafTris[3 * (size_t)k];
        const float4 q1 = s.leafTris[3 * (size_t)k + 1];
        const float4 q2 = s.leafTris[3 * (size_t)k + 2];
        if (COUNT) nt++;
        float t, be, ga;
        const bool acc = tri_test(vec3{q0.x, q0.y, q0.z}, vec3{q1.x, q1.y, q1.z}, vec3{q2.x, q2.y, q2.z}, r.o, r.d, tMin,
                                  tMax, t, be, ga);
        if (acc && (ANYHIT || t < bestT)) {
            bestTri = f2i(q0.w); bestT = t; bestB = be; bestG = ga;
        }
        if (ANYHIT && acc) {
            finish();
        } else if (f2i(q1.w)) {  // last triangle of the leaf: the walk resumes at the leaf's idHit (= idMiss)
            if (cont == 0) finish();
            else { node = cont; state = SEARCH; }
        } else {
            ++k;
        }
    };

    while (true) {
        const unsigned long long mEmpty = __ballot(state == EMPTY);
        const int nEmpty = __popcll(mEmpty);
        if (nEmpty == 64 && exhausted) break;
        if (!exhausted && nEmpty >= kRefillLanes) {
            // ---- refill: empty lanes take the next rays of the queue ----
            if (chunkLeft == 0) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(cursor, 64u);
                base = __shfl(base, 0);
                if (base >= total) {
                    exhausted = true;
                } else {
                    chunkBase = base;
                    chunkLeft = (total - base) < 64u ? (total - base) : 64u;
                }
            }
            if (chunkLeft != 0) {
                const unsigned rank = (unsigned)__popcll(mEmpty & ((1ull << lane) - 1ull));
                const unsigned take = chunkLeft < (unsigned)nEmpty ? chunkLeft : (unsigned)nEmpty;
                if (state == EMPTY && rank < take) {
                    const unsigned j = chunkBase + rank;  // dense ray index
                    unsigned slot = j;
                    if (q.counts) {  // dense index -> shard by binary search over the prefix sums
                        unsigned lo = 0, hi = kShards;
                        while (hi - lo > 1) {
                            const unsigned mid = (lo + hi) >> 1;
                            if (prefix[mid] <= j) lo = mid; else hi = mid;
                        }
                        slot = lo * q.regionSize + (j - prefix[lo]);
                    }
                    queueSlot = slot;
                    raySlot = index ? (unsigned)index[slot] : slot;
                    const size_t rs = ANYHIT ? (size_t)slot : (size_t)raySlot;  // shadow rays live in queue order
                    const float4 r0 = rays[2 * rs], r1 = rays[2 * rs + 1];
                    r.o = vec3{r0.x, r0.y, r0.z}; tMin = r0.w;
                    r.d = vec3{r1.x, r1.y, r1.z}; tMax = r1.w;
                    fast = s.planesTame && fast_div_ok(r.o, r.d);
                    r.y = vec3{1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
                    r.nx = r.d.x < 0.0f; r.ny = r.d.y < 0.0f; r.nz = r.d.z < 0.0f;
                    bestTri = -1; bestT = 3.402823466e+38f; bestB = 0.0f; bestG = 0.0f;
                    node = 0;
                    state = SEARCH;
                }
                chunkBase += take;
                chunkLeft -= take;
            }
            if (nEmpty == 64) continue;  // nothing else to do yet
        }
        // ---- box burst: searching lanes advance up to kBoxBurst nodes without re-voting ----
#pragma unroll
        for (int b = 0; b < kBoxBurst; b++)
            if (state == SEARCH) box_step();
        // ---- triangle steps: when enough lanes are parked at a leaf, or nobody is searching ----
        unsigned long long mLeaf = __ballot(state == LEAF);
        if (mLeaf != 0ull && (__popcll(mLeaf) >= kLeafLanes || __ballot(state == SEARCH) == 0ull)) {
            do {
                if (state == LEAF) tri_step();
                mLeaf = __ballot(state == LEAF);
            } while (__popcll(mLeaf) >= kLeafLanes);
        }
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {  // wave-level reduction, one atomic pair per wavefront
            nv += __shfl_down(nv, off);
            nt += __shfl_down(nt, off);
        }
        if (lane == 0) {
            atomicAdd(&counters[0], (unsigned long long)nv);
            atomicAdd(&counters[1], (unsigned long long)nt);
        }
    }
}

// Scene::eyeRay for pixels [first, first+count), row-major (j,i) order
__global__ __launch_bounds__(kBlock) void k_eye_rays(CameraDev cam, size_t first, unsigned count,
                                                     float4* __restrict__ rays) {
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock) {
        const size_t pix = first + i;
        vec3 o, d;
        eye_ray(cam, (int)(pix % (size_t)cam.width), (int)(pix / (size_t)cam.width), o, d);
        rays[2 * (size_t)i] = make_float4(o.x, o.y, o.z, 0.0f);
        rays[2 * (size_t)i + 1] = make_float4(d.x, d.y, d.z, 3.402823466e+38f);
    }
}

// HitInfo {t, P3, N3, T2, material} from hit records (10 floats per ray)
__global__ __launch_bounds__(kBlock) void k_resolve_hits(DevScene s, const float4* __restrict__ rays,
                                                         const float4* __restrict__ hits, unsigned n,
                                                         float* __restrict__ out) {
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 h = hits[i];
        const float4 r1 = rays[2 * (size_t)i + 1];
        float* q = out + (size_t)10 * i;
        const int tri = f2i(h.x);
        if (tri >= 0) {
            const HitInfo hi = reconstruct_hit(s, tri, h.y, h.z, h.w, vec3{r1.x, r1.y, r1.z});
            q[0] = hi.t; q[1] = hi.P.x; q[2] = hi.P.y; q[3] = hi.P.z; q[4] = hi.N.x; q[5] = hi.N.y; q[6] = hi.N.z;
            q[7] = hi.T.x; q[8] = hi.T.y; q[9] = (float)hi.material;
        } else {
            q[0] = 3.402823466e+38f;
            for (int k = 1; k < 9; k++) q[k] = 0.0f;
            q[9] = -1.0f;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_background(DevScene s, const float* __restrict__ dir, unsigned n,
                                                       float* __restrict__ rgb) {
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const vec3 c = get_background(s, vec3{dir[3 * (size_t)i], dir[3 * (size_t)i + 1], dir[3 * (size_t)i + 2]});
        rgb[3 * (size_t)i] = c.x;
        rgb[3 * (size_t)i + 1] = c.y;
        rgb[3 * (size_t)i + 2] = c.z;
    }
}

// self-test of div_by_recip against IEEE division on pseudo-random and structured operand pairs that satisfy
// its preconditions (mipt_selftest_fast_division)
__global__ __launch_bounds__(kBlock) void k_divcheck(unsigned seed, unsigned perThread, unsigned long long* mismatches) {
    unsigned st = pcg_hash(seed ^ pcg_hash(blockIdx.x * kBlock + threadIdx.x));
    unsigned bad = 0;
    for (unsigned it = 0; it < perThread; it++) {
        st = pcg_hash(st);
        const unsigned r0 = st;
        st = pcg_hash(st);
        const unsigned r1 = st;
        st = pcg_hash(st);
        const unsigned r2 = st;
        // mantissas: random, or one of the corner patterns (all ones, all zeros, one bit, all-but-one)
        auto mant = [](unsigned r, unsigned sel) -> unsigned {
            switch (sel & 7u) {
            case 0: return 0x7fffffu;
            case 1: return 0u;
            case 2: return 1u << (r % 23u);
            case 3: return 0x7fffffu ^ (1u << (r % 23u));
            default: return r & 0x7fffffu;
            }
        };
        const unsigned md = mant(r0, r2), ma = mant(r1, r2 >> 3);
        const int ed = 127 - 40 + (int)((r2 >> 6) % 81u);           // |d| in [2^-40, 2^40]
        const int ea = 127 - 63 + (int)((r2 >> 14) % 105u);         // |a| in [2^-63, 2^41]
        const float d = i2f((int)(((r2 >> 30) & 1u) << 31 | (unsigned)ed << 23 | md));
        float a = i2f((int)(((r2 >> 31) & 1u) << 31 | (unsigned)ea << 23 | ma));
        if ((r2 >> 22 & 63u) == 0u) a = 0.0f;
        const float y = 1.0f / d;
        const float q = div_by_recip(a, d, y);
        const float ref = a / d;
        if (f2i(q) != f2i(ref)) bad++;
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}
void launch_divcheck(hipStream_t st, unsigned seed, unsigned blocks, unsigned perThread, unsigned long long* mismatches) {
    hipLaunchKernelGGL(k_divcheck, dim3(blocks), dim3(kBlock), 0, st, seed, perThread, mismatches);
}

// ---- launchers ----
static inline unsigned grid_for(size_t n) {
    size_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > kMaxGrid) g = kMaxGrid;
    return (unsigned)g;
}

// persistent grid: enough wavefronts to fill the chip (256 CUs x 28 resident), never more than the rays need
static inline unsigned trace_grid(size_t nMax) {
    size_t g = (nMax + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (unsigned)g;
}
void launch_trace_closest(hipStream_t st, const DevScene& s, const float4* rays, const int* index, QueueView q,
                          unsigned* cursor, float4* hits, unsigned long long* counters) {
    const unsigned g = trace_grid(q.space);
    if (counters)
        hipLaunchKernelGGL((k_trace<false, true>), dim3(g), dim3(kBlock), 0, st, s, rays, index, nullptr, q, cursor, hits,
                           nullptr, counters);
    else
        hipLaunchKernelGGL((k_trace<false, false>), dim3(g), dim3(kBlock), 0, st, s, rays, index, nullptr, q, cursor, hits,
                           nullptr, counters);
}
void launch_trace_occluded(hipStream_t st, const DevScene& s, const float4* rays, const int* target, QueueView q,
                           unsigned* cursor, unsigned char* occ, unsigned long long* counters) {
    const unsigned g = trace_grid(q.space);
    if (counters)
        hipLaunchKernelGGL((k_trace<true, true>), dim3(g), dim3(kBlock), 0, st, s, rays, nullptr, target, q, cursor,
                           nullptr, occ, counters);
    else
        hipLaunchKernelGGL((k_trace<true, false>), dim3(g), dim3(kBlock), 0, st, s, rays, nullptr, target, q, cursor,
                           nullptr, occ, counters);
}
void launch_eye_rays(hipStream_t st, const CameraDev& cam, size_t first, size_t count, float4* rays) {
    hipLaunchKernelGGL(k_eye_rays, dim3(grid_for(count)), dim3(kBlock), 0, st, cam, first, (unsigned)count, rays);
}
void launch_resolve_hits(hipStream_t st, const DevScene& s, const float4* rays, const float4* hits, size_t n,
                         float* out) {
    hipLaunchKernelGGL(k_resolve_hits, dim3(grid_for(n)), dim3(kBlock), 0, st, s, rays, hits, (unsigned)n, out);
}
void launch_background(hipStream_t st, const DevScene& s, const float* dir, size_t n, float* rgb) {
    hipLaunchKernelGGL(k_background, dim3(grid_for(n)), dim3(kBlock), 0, st, s, dir, (unsigned)n, rgb);
}

}  // namespace ptd
