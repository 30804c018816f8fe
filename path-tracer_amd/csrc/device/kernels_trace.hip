// kernels_trace.hip -- the traversal kernels (closest-hit "extend", any-hit
// "occluded") and the small kernels around the parity entry points: eye-ray
// generation, HitInfo resolution, environment lookup.
//
// Launch shape: k_trace runs 1024-thread blocks (16 wavefronts of 64 lanes), two
// per CU, one ray per lane at a time; the small kernels use 256-thread blocks.  Queue lengths live in device memory (sharded counters), so
// a whole frame is enqueued without host syncs.
#include "kernels.hpp"
#include "shading.hpp"
#include "slow_walk.hpp"

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
//   * a lane that reaches a leaf parks the leaf (a 4-deep FIFO) and walks on;
//     the wavefront runs "triangle steps" only when many lanes hold parked
//     leaves, so both code paths run with many lanes instead of interleaving;
//   * the top of the tree (DevScene::nTopNodes nodes, numbered first) is read
//     from LDS: a wavefront's node loads are 64 different cache lines, which
//     the CU's L1 takes one per cycle -- that, not HBM or the VALU, bounded the
//     kernel (profiles/r01/pmc_v5.json: TA busy 0.75) -- and about half of all
//     node visits fall into the top 2032 nodes;
//   * the six slab quotients use a per-ray reciprocal with a residual
//     correction that reproduces IEEE division exactly (traverse.hpp).
// Queue: dense (q.counts == nullptr, q.space rays) or sharded (kernels.hpp);
// the shard counters are prefix-summed into LDS so that rays are addressed by
// a dense index.  index (nullable): queue slot -> ray slot; the result goes to
// hits[ray slot] (closest hit) or occ[target[queue slot]] (any hit).
// Scheduling constants (tools/tune_trace.sh builds variants with -DPT_TUNE_...; the defaults are the measured best)
#ifndef PT_TUNE_REFILL
#define PT_TUNE_REFILL 16
#endif
#ifndef PT_TUNE_LEAF
#define PT_TUNE_LEAF 40
#endif
#ifndef PT_TUNE_LEAF_KEEP
#define PT_TUNE_LEAF_KEEP 16
#endif
#ifndef PT_TUNE_STUCK
#define PT_TUNE_STUCK 8
#endif
#ifndef PT_TUNE_BURST
#define PT_TUNE_BURST 4
#endif
#ifndef PT_TUNE_PEND_ANYHIT
#define PT_TUNE_PEND_ANYHIT 1
#endif
#ifndef PT_TUNE_LEAF_ANYHIT
#define PT_TUNE_LEAF_ANYHIT PT_TUNE_LEAF
#endif
#ifndef PT_TUNE_LEAF_KEEP_ANYHIT
#define PT_TUNE_LEAF_KEEP_ANYHIT PT_TUNE_LEAF_KEEP
#endif
constexpr int kRefillLanes = PT_TUNE_REFILL;  // refill when this many lanes are empty
constexpr int kStuckLanes = PT_TUNE_STUCK;    // ... or when this many lanes cannot walk on before their triangles are tested
constexpr int kBoxBurst = PT_TUNE_BURST;      // box steps between two votes
constexpr int kPendClosest = 4;               // leaves a lane may park before it must wait (closest hit)
constexpr int kPendAnyHit = PT_TUNE_PEND_ANYHIT;  // any-hit rays end at their first accepted triangle: parking more wastes box steps

// 8 wavefronts per SIMD (two 1024-thread blocks per CU): the compiler keeps the kernel within 64 VGPRs (the
// closest-hit variant spills 7 rarely used values); measured 2-3 % faster than 6 per SIMD with 768-thread blocks
#ifndef PT_TUNE_WAVES_PER_EU
#define PT_TUNE_WAVES_PER_EU 8
#endif
#define PT_TRACE_ATTR __attribute__((amdgpu_waves_per_eu(PT_TUNE_WAVES_PER_EU, PT_TUNE_WAVES_PER_EU)))
constexpr int kSlowGrid = 256;           // wavefronts of k_trace_slow (an idle launch costs a few microseconds per 256)
constexpr int kSlowStack = 8192;         // node stack of one k_trace_slow wavefront (32 KB of LDS)

template <bool ANYHIT, bool COUNT>
__global__ __launch_bounds__(kTraceBlock) PT_TRACE_ATTR void k_trace(DevScene s, const float4* __restrict__ rays,
                                                  const int* __restrict__ index, const int* __restrict__ target,
                                                  QueueView q, unsigned* __restrict__ cursor,
                                                  float4* __restrict__ hits, unsigned char* __restrict__ occ,
                                                  unsigned long long* __restrict__ counters,
                                                  int* __restrict__ remap, const unsigned* __restrict__ remapCount) {
    // remap != nullptr: the queue is the list remap[0 .. *remapCount) of queue slots (the rays k_trace_wide handed
    // back, kernels_trace_wide.hip); most launches find it empty, or find every entry taken (-1: k_trace_wide's own
    // wavefronts empty short lists, and with MIPT_DRAIN=0 k_trace_slow below does) -- what is left is traced here, a lane per ray
    if (remap && (size_t)blockIdx.x * kTraceBlock >= (size_t)*remapCount) return;  // a block per kTraceBlock listed rays is plenty
    __shared__ float4 top[2 * kTopNodes];
    __shared__ unsigned prefix[kShards + 1];
    const int lane = threadIdx.x & 63;
    const int nTop = s.nTopNodes;
    for (int i = threadIdx.x; i < 2 * nTop; i += kTraceBlock) top[i] = s.nodes[i];
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
    }
    __syncthreads();
    const unsigned total = remap ? *remapCount : (q.counts ? prefix[kShards] : q.space);
    // explicit address spaces: with two generic pointers the compiler merges the two arms of "LDS or global" into one
    // flat load through a selected pointer, which sends the LDS lanes through the vector-memory path as well
    const __attribute__((address_space(3))) float4v* topLds = (const __attribute__((address_space(3))) float4v*)top;
    const __attribute__((address_space(1))) float4v* nodesGlobal = (const __attribute__((address_space(1))) float4v*)s.nodes;

    // lane state
    bool hasRay = false;
    int node = 0;                     // next node to test; 0 after the walk has run off the tree (root is tested first)
    bool walking = false;             // the box walk of this ray is still going
    // FIFO of leaves whose triangles are still to be tested, in visit order.  slots = parked leaves (0..4); cur =
    // next triangle of the oldest one (valid while slots > 0); p0..p2 = first triangles of the slots-1 younger
    // ones, newest in p0.  The walk does
    // not depend on triangle results (the reference's box test ignores the current best hit, cs488.h:1669), so
    // parking the triangle tests changes nothing but the schedule; the tests themselves run per lane in visit
    // order, which keeps the strict-'<' tie-break.
    int slots = 0, cur = 0, p0 = 0, p1 = 0, p2 = 0;
    unsigned raySlot = 0, queueSlot = 0;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 1, yx = 0, yy = 0, yz = 1;  // ray; y = RN(1/d)
    float tMin = 0.0f, tMax = 0.0f;
    bool fast = false;
    int bestTri = -1;
    float bestT = 3.402823466e+38f, bestB = 0.0f, bestG = 0.0f;
    unsigned nv = 0, nt = 0;
#ifdef PT_TUNE_DUMMY_VALU
    float dummyValu = 0.5f;
#endif
    // wavefront-uniform chunk of ray indices
    unsigned chunkBase = 0, chunkLeft = 0, chunkShard = 0;
    // Rays are handed out in chunks, one atomic on the launch's one cursor per chunk.  Atomics on one address
    // complete about one per 12 ns on this chip: at 64 rays per atomic that alone caps a launch at 5.5 G rays/s
    // (measured with rays that miss the root box: 3.0 ms per 16.8 M rays, whatever the kernel did per ray), which
    // is where both traversal kernels stood.  Guided self-scheduling: a wavefront asks for half of its share of
    // what was left when it last looked (64..1024 rays), so chunks are large while the queue is long and shrink
    // to 64 at the end, where they decide how evenly the wavefronts finish.
    const unsigned nWaves = gridDim.x * (unsigned)(kTraceBlock / 64);
    unsigned lastBase = 0;
    bool exhausted = (total == 0);
    constexpr int kDepth = ANYHIT ? kPendAnyHit : kPendClosest;  // leaves a lane may park (1..4)
    // triangle steps start when this many lanes hold parked leaves and go on while kLeafLanesKeep still do
    constexpr int kLeafLanes = ANYHIT ? PT_TUNE_LEAF_ANYHIT : PT_TUNE_LEAF;
    constexpr int kLeafLanesKeep = ANYHIT ? PT_TUNE_LEAF_KEEP_ANYHIT : PT_TUNE_LEAF_KEEP;

    // (no lambdas here: variables captured by reference end up in a stack closure that the optimiser may fail to
    // scalarise once it has merged "c ? x : y" over two captures into a load through a selected pointer)
// the ray of this lane is done: publish its result, free the lane
#define PT_FINISH()                                                                        \
    do {                                                                                   \
        if (ANYHIT) {                                                                      \
            if (bestTri >= 0) occ[target ? (unsigned)target[queueSlot] : queueSlot] = 1;   \
        } else {                                                                           \
            hits[raySlot] = make_float4(i2f(bestTri), bestT, bestB, bestG);                \
        }                                                                                  \
        hasRay = false;                                                                    \
        walking = false;                                                                   \
        slots = 0;                                                                         \
    } while (0)
#define PT_PEND_FULL() (slots >= kDepth)

    while (true) {
        const unsigned long long mEmpty = __ballot(!hasRay);
        const int nEmpty = __popcll(mEmpty);
        if (nEmpty == 64 && exhausted) break;
        if (!exhausted && nEmpty >= kRefillLanes) {
            // ---- refill: empty lanes take the next rays of the queue ----
            if (chunkLeft == 0) {
                unsigned base = 0;
                unsigned chunkRays = ((total - lastBase) / (nWaves * 2u)) & ~63u;
                chunkRays = chunkRays < 64u ? 64u : (chunkRays > 1024u ? 1024u : chunkRays);
                if (lane == 0) base = atomicAdd(cursor, chunkRays);
                base = __shfl(base, 0);
                if (base >= total) {
                    exhausted = true;
                } else {
                    chunkBase = base;
                    lastBase = base;
                    chunkLeft = (total - base) < chunkRays ? (total - base) : chunkRays;
                    // shard of the chunk's first ray: the number of shards that end at or before it (one vote
                    // instead of a binary search per lane; lanes step on from here, usually zero times)
                    if (q.counts && !remap) chunkShard = (unsigned)__popcll(__ballot(prefix[lane + 1] <= base));
                }
            }
            if (chunkLeft != 0) {
                const unsigned rank = (unsigned)__popcll(mEmpty & ((1ull << lane) - 1ull));
                const unsigned take = chunkLeft < (unsigned)nEmpty ? chunkLeft : (unsigned)nEmpty;
                if (!hasRay && rank < take) {
                    const unsigned j = chunkBase + rank;  // dense ray index
                    unsigned slot = j;
                    if (remap) {
                        slot = (unsigned)remap[j];  // -1: a k_trace_wide wavefront has traced this one already
                        remap[j] = -1;              // the list's resting state
                    } else if (q.counts) {  // dense index -> shard: j < total = prefix[kShards] ends the loop
                        unsigned sh = chunkShard;
                        while (j >= prefix[sh + 1]) ++sh;
                        slot = sh * q.regionSize + (j - prefix[sh]);
                    }
                    if (!(remap && (int)slot < 0)) {
                        queueSlot = slot;
                        raySlot = index ? (unsigned)index[slot] : slot;
                        const size_t rs = ANYHIT ? (size_t)slot : (size_t)raySlot;  // shadow rays live in queue order
                        const float4 r0 = rays[2 * rs], r1 = rays[2 * rs + 1];
                        ox = r0.x; oy = r0.y; oz = r0.z; tMin = r0.w;
                        dx = r1.x; dy = r1.y; dz = r1.z; tMax = r1.w;
                        fast = s.planesTame && fast_div_ok(vec3{ox, oy, oz}, vec3{dx, dy, dz});
                        yx = 1.0f / dx; yy = 1.0f / dy; yz = 1.0f / dz;
                        bestTri = -1; bestT = 3.402823466e+38f; bestB = 0.0f; bestG = 0.0f;
                        node = 0;
                        hasRay = true;
                        walking = true;
                    }
                }
                chunkBase += take;
                chunkLeft -= take;
            }
            if (nEmpty == 64) continue;  // nothing else to do yet
        }
        // ---- box burst: walking lanes with room to park a leaf advance up to kBoxBurst nodes (cs488.h:1669-1680) ----
        // The scalar unit (exec-mask juggling, ~4.4 cycles per instruction and SIMD) limits this loop as much as the
        // vector ALU does (tools/micro/issue_rate.hip), so the step keeps its control flow flat: one region for the
        // node load, one for a parked leaf, one for a finished ray; when every ray of the wavefront qualifies for the
        // reciprocal slab test (the usual case) the burst runs a copy without the IEEE arm.
#define PT_BOX_STEP(FASTONLY)                                                                                         \
    if (walking && !PT_PEND_FULL()) {                                                                                 \
        float4v n0, n1; /* {min.x, max.x, min.y, max.y} {min.z, max.z, idMiss, link} */                               \
        if (node < nTop) { /* top of the tree: from LDS */                                                            \
            n0 = topLds[2 * node];                                                                                    \
            n1 = topLds[2 * node + 1];                                                                                \
        } else {                                                                                                      \
            n0 = nodesGlobal[2 * (size_t)node];                                                                       \
            n1 = nodesGlobal[2 * (size_t)node + 1];                                                                   \
        }                                                                                                             \
        if (COUNT) nv++;                                                                                              \
        bool hit;                                                                                                     \
        if (FASTONLY) hit = slab_hit_fast(n0, n1, ox, oy, oz, dx, dy, dz, yx, yy, yz);                                \
        else hit = fast ? slab_hit_fast(n0, n1, ox, oy, oz, dx, dy, dz, yx, yy, yz)                                   \
                        : slab_hit_ieee(n0, n1, vec3{ox, oy, oz}, vec3{dx, dy, dz});                                  \
        const int link = f2i(n1.w);                                                                                   \
        const bool leafHit = hit && link < 0; /* park its triangles, go on at idHit (= idMiss) */                     \
        const bool none = slots == 0;                                                                                 \
        node = (!hit || leafHit) ? f2i(n1.z) : link; /* idMiss, or idHit of an interior node */                       \
        if (leafHit) {                                                                                                \
            if (none) {                                                                                               \
                cur = ~link;                                                                                          \
            } else {                                                                                                  \
                p2 = p1; p1 = p0; p0 = ~link;                                                                         \
            }                                                                                                         \
            ++slots;                                                                                                  \
        }                                                                                                             \
        const bool ended = node == 0;                                                                                 \
        walking = !ended;                                                                                             \
        if (ended && none && !leafHit) PT_FINISH();                                                                   \
    }
        if (__ballot(hasRay && !fast) == 0ull) {
#pragma unroll
            for (int b = 0; b < kBoxBurst; b++) PT_BOX_STEP(true)
        } else {
#pragma unroll
            for (int b = 0; b < kBoxBurst; b++) PT_BOX_STEP(false)
        }
#undef PT_BOX_STEP
        // ---- triangle steps: when enough lanes hold parked leaves, or nobody can walk on ----
        const unsigned long long mCanWalk = __ballot(walking && !PT_PEND_FULL());
        unsigned long long mPend = __ballot(slots > 0);
        // lanes that cannot walk on before their parked triangles are tested (walk over, or FIFO full)
        const unsigned long long mStuck = mPend & ~mCanWalk;
        if (mPend != 0ull && (__popcll(mPend) >= kLeafLanes || __popcll(mStuck) >= kStuckLanes || mCanWalk == 0ull)) {
            do {
                if (slots > 0) {  // one triangle of the oldest parked leaf (cs488.h:1671-1675 / 1693-1697)
                    const float4 q0 = s.leafTris[3 * (size_t)cur];
                    const float4 q1 = s.leafTris[3 * (size_t)cur + 1];
                    const float4 q2 = s.leafTris[3 * (size_t)cur + 2];
                    if (COUNT) nt++;
                    float t, be, ga;
                    const bool acc = tri_test(vec3{q0.x, q0.y, q0.z}, vec3{q1.x, q1.y, q1.z}, vec3{q2.x, q2.y, q2.z},
                                              vec3{ox, oy, oz}, vec3{dx, dy, dz}, tMin, tMax, t, be, ga);
                    if (acc && (ANYHIT || t < bestT)) {
                        bestTri = f2i(q0.w); bestT = t; bestB = be; bestG = ga;
                    }
                    if (ANYHIT && acc) {
                        PT_FINISH();  // isHit returns at the first accepted triangle (cs488.h:1696)
                    } else if (f2i(q1.w)) {  // last triangle of this leaf: on to the next parked leaf
                        --slots;
                        if (slots > 0) {
                            cur = (slots == 1) ? p0 : (slots == 2) ? p1 : p2;
                        } else if (!walking) {
                            PT_FINISH();
                        }
                    } else {
                        ++cur;
                    }
                }
                mPend = __ballot(slots > 0);
            } while (__popcll(mPend) >= kLeafLanesKeep);
        }
    }
#undef PT_FINISH
#undef PT_PEND_FULL
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
            HitInfo hi = reconstruct_hit(s, tri, h.y, h.z, h.w, vec3{r1.x, r1.y, r1.z});
            hi.T = hit_texcoord(s, tri, h.z, h.w);
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
        const int ed = 127 + kDirLo + (int)((r2 >> 6) % (unsigned)(kDirHi - kDirLo + 1));   // |d| in [2^-40, 2^40]
        const int ea = 127 + kNumLo + (int)((r2 >> 14) % (unsigned)(kNumHi - kNumLo + 1));  // |a| in [2^-84, 2^41]
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

// pt/vec.hpp's primitives over plain arrays, on the device (mipt_selftest_vec_ops): what tests compare with the
// reference's own linalg.h bit for bit.  op codes: see include/mipt.h
__host__ __device__ inline void vec_op_one(int op, const float* a, const float* b, const float* c, const float* d, size_t i, float* out) {
    const vec3 A{a[3 * i], a[3 * i + 1], a[3 * i + 2]};
    const vec3 B = b ? vec3{b[3 * i], b[3 * i + 1], b[3 * i + 2]} : vec3{0, 0, 0};
    const vec3 Cc = c ? vec3{c[3 * i], c[3 * i + 1], c[3 * i + 2]} : vec3{0, 0, 0};
    const vec3 D = d ? vec3{d[3 * i], d[3 * i + 1], d[3 * i + 2]} : vec3{0, 0, 0};
    vec3 r{0, 0, 0};
    switch (op) {
    case 0: out[i] = pt::dot(A, B); return;
    case 1: r = pt::cross(A, B); break;
    case 2: r = pt::normalize(A); break;
    case 3: out[i] = pt::det3(A, B, Cc); return;
    case 4: r = pt::mulcols(A, B, Cc, D); break;
    case 5: out[i] = pt::length2(A); return;
    case 6: { const pt::vec2 q = pt::normalize(pt::vec2{A.x, A.y}); out[2 * i] = q.x; out[2 * i + 1] = q.y; return; }
    case 7: r = vec3{A.x / B.x, A.y / B.y, A.z / B.z}; break;
    default: return;
    }
    out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
}
__global__ __launch_bounds__(kBlock) void k_vec_ops(int op, const float* __restrict__ a, const float* __restrict__ b,
                                                    const float* __restrict__ c, const float* __restrict__ d, unsigned n,
                                                    float* __restrict__ out) {
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) vec_op_one(op, a, b, c, d, i, out);
}
void vec_ops_host(int op, const float* a, const float* b, const float* c, const float* d, size_t n, float* out) {
    for (size_t i = 0; i < n; i++) vec_op_one(op, a, b, c, d, i, out);
}

// The handful of rays per launch that k_trace_wide hands back, a wavefront per ray (slow_walk.hpp), as a launch of its
// own: since round 3 only for contexts that do not let k_trace_wide's own wavefronts trace them (MIPT_DRAIN=0).
// Entries that a k_trace_wide wavefront has already taken read -1.
template <bool ANYHIT, bool COUNT>
__global__ __launch_bounds__(64) void k_trace_slow(DevScene s, const float4* __restrict__ rays, const int* __restrict__ index,
                                                   const int* __restrict__ target, float4* __restrict__ hits,
                                                   unsigned char* __restrict__ occ, unsigned long long* __restrict__ counters,
                                                   int* __restrict__ list, const unsigned* __restrict__ listCount) {
    __shared__ int stack[kSlowStack];
    const unsigned n = *listCount;
    if (n > kSlowMaxRays || !s.slowOk) return;  // k_trace takes long lists, and every list of a tree too deep for this stack
    const int lane = threadIdx.x;
    unsigned nv = 0, nt = 0;
    for (unsigned r = blockIdx.x; r < n; r += gridDim.x) {
        if (list[r] < 0) continue;
        const unsigned slot = (unsigned)list[r];
        const unsigned raySlot = index ? (unsigned)index[slot] : slot;
        const size_t rs = ANYHIT ? (size_t)slot : (size_t)raySlot;  // shadow rays live in queue order
        const float4 r0 = rays[2 * rs], r1 = rays[2 * rs + 1];
        int bestTri, bestSeq;
        float bestT, bestB, bestG;
        slow_walk_wave<ANYHIT, COUNT, 64>(s, vec3{r0.x, r0.y, r0.z}, vec3{r1.x, r1.y, r1.z}, r0.w, r1.w, stack, kSlowStack, lane, bestTri,
                                          bestSeq, bestT, bestB, bestG, nv, nt);
        if (lane == 0) {
            if (ANYHIT) {
                if (bestTri >= 0) occ[target ? (unsigned)target[slot] : slot] = 1;
            } else {
                hits[raySlot] = make_float4(i2f(bestTri), bestT, bestB, bestG);
            }
            list[r] = -1;  // the list's resting state
        }
        wave_lds_sync();
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {
            nv += __shfl_down(nv, off);
            nt += __shfl_down(nt, off);
        }
        if (lane == 0) {
            atomicAdd(&counters[0], (unsigned long long)nv);
            atomicAdd(&counters[1], (unsigned long long)nt);
        }
    }
}

// ---- launchers ----
static inline unsigned grid_for(size_t n) {
    size_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > kMaxGrid) g = kMaxGrid;
    return (unsigned)g;
}

// persistent grid: two blocks per CU (their LDS fills it), never more than the rays need
static inline unsigned trace_grid(size_t nMax) {
    size_t g = (nMax + kTraceBlock - 1) / kTraceBlock;
    if (g < 1) g = 1;
    if (g > (size_t)kTraceGrid) g = kTraceGrid;
    return (unsigned)g;
}
void launch_trace_closest(hipStream_t st, const DevScene& s, const float4* rays, const int* index, QueueView q,
                          unsigned* cursor, float4* hits, unsigned long long* counters, int* remap,
                          const unsigned* remapCount) {
    const unsigned g = trace_grid(q.space);
    if (counters)
        hipLaunchKernelGGL((k_trace<false, true>), dim3(g), dim3(kTraceBlock), 0, st, s, rays, index, nullptr, q, cursor, hits,
                           nullptr, counters, remap, remapCount);
    else
        hipLaunchKernelGGL((k_trace<false, false>), dim3(g), dim3(kTraceBlock), 0, st, s, rays, index, nullptr, q, cursor, hits,
                           nullptr, counters, remap, remapCount);
}
void launch_trace_occluded(hipStream_t st, const DevScene& s, const float4* rays, const int* target, QueueView q,
                           unsigned* cursor, unsigned char* occ, unsigned long long* counters, int* remap,
                           const unsigned* remapCount) {
    const unsigned g = trace_grid(q.space);
    if (counters)
        hipLaunchKernelGGL((k_trace<true, true>), dim3(g), dim3(kTraceBlock), 0, st, s, rays, nullptr, target, q, cursor,
                           nullptr, occ, counters, remap, remapCount);
    else
        hipLaunchKernelGGL((k_trace<true, false>), dim3(g), dim3(kTraceBlock), 0, st, s, rays, nullptr, target, q, cursor,
                           nullptr, occ, counters, remap, remapCount);
}
// the rays list[0 .. *listCount) (queue slots), a wavefront each; a no-op for lists longer than kSlowMaxRays
void launch_trace_slow_closest(hipStream_t st, const DevScene& s, const float4* rays, const int* index, float4* hits,
                               unsigned long long* counters, int* list, const unsigned* listCount) {
    if (counters)
        hipLaunchKernelGGL((k_trace_slow<false, true>), dim3(kSlowGrid), dim3(64), 0, st, s, rays, index, nullptr, hits, nullptr, counters, list, listCount);
    else
        hipLaunchKernelGGL((k_trace_slow<false, false>), dim3(kSlowGrid), dim3(64), 0, st, s, rays, index, nullptr, hits, nullptr, counters, list, listCount);
}
void launch_trace_slow_occluded(hipStream_t st, const DevScene& s, const float4* rays, const int* target, unsigned char* occ,
                                unsigned long long* counters, int* list, const unsigned* listCount) {
    if (counters)
        hipLaunchKernelGGL((k_trace_slow<true, true>), dim3(kSlowGrid), dim3(64), 0, st, s, rays, nullptr, target, nullptr, occ, counters, list, listCount);
    else
        hipLaunchKernelGGL((k_trace_slow<true, false>), dim3(kSlowGrid), dim3(64), 0, st, s, rays, nullptr, target, nullptr, occ, counters, list, listCount);
}
void launch_eye_rays(hipStream_t st, const CameraDev& cam, size_t first, size_t count, float4* rays) {
    hipLaunchKernelGGL(k_eye_rays, dim3(grid_for(count)), dim3(kBlock), 0, st, cam, first, (unsigned)count, rays);
}
void launch_resolve_hits(hipStream_t st, const DevScene& s, const float4* rays, const float4* hits, size_t n,
                         float* out) {
    hipLaunchKernelGGL(k_resolve_hits, dim3(grid_for(n)), dim3(kBlock), 0, st, s, rays, hits, (unsigned)n, out);
}
void launch_vec_ops(hipStream_t st, int op, const float* a, const float* b, const float* c, const float* d, size_t n, float* out) {
    hipLaunchKernelGGL(k_vec_ops, dim3(grid_for(n)), dim3(kBlock), 0, st, op, a, b, c, d, (unsigned)n, out);
}
void launch_background(hipStream_t st, const DevScene& s, const float* dir, size_t n, float* rgb) {
    hipLaunchKernelGGL(k_background, dim3(grid_for(n)), dim3(kBlock), 0, st, s, dir, (unsigned)n, rgb);
}

}  // namespace ptd
