// kernels.hpp -- host-callable launchers of the HIP kernels and the buffers
// the wavefront integrator works on.
#pragma once
#include <hip/hip_runtime.h>

#include "../../../include/mipt.h"
#include "device_scene.hpp"

namespace ptd {

struct CameraDev;

constexpr unsigned kMaxGrid = 8192;  // blocks; larger queues are grid-strided
constexpr int kMaxLights = 16;
constexpr int kMaxIter = 66;         // bounce iterations a frame can enqueue (counter slots)
// Work queues are SHARDED: a returning atomic on one word runs at ~88/us on
// MI355X, which a frame's 32k wavefront appends would saturate (measured:
// 0.37 ms per 2M-item queue).  A queue is kShards regions of regionSize slots,
// each with its own counter on its own 128-B line; the wavefront with index w
// of a kernel's iteration space appends to shard w % kShards, so no region can
// overflow.  Slot i of a queue is live iff (i % regionSize) < count[i / regionSize].
constexpr int kShards = 64;
constexpr int kCntStride = 32;       // unsigned words between two shard counters (128 B)
constexpr int kQueuesPerIter = 5;    // 0 = extension, 1 = waiting Lambertian vertices, 2 = shadow rays,
                                     // 3 / 4 = work cursors of the closest-hit / any-hit launch (shard 0 only)

__host__ __device__ inline size_t counter_index(int it, int queue, int shard) {
    return (((size_t)it * kQueuesPerIter + queue) * kShards + shard) * kCntStride;
}
constexpr size_t kCounterWords = (size_t)kMaxIter * kQueuesPerIter * kShards * kCntStride;

// a sharded queue as a kernel sees it; counts == nullptr means "dense, n items"
struct QueueView {
    const unsigned* counts;  // counter of shard 0; shard s at counts[s * kCntStride]
    unsigned regionSize;     // slots per shard (multiple of 64)
    unsigned space;          // kShards * regionSize, or the dense item count
};

// Per-path state of the wavefront integrator, SoA of float4 (16-B lane loads).
// Index p = slot of the path inside this context's pixel partition.
struct PathBuffers {
    size_t capacity;      // paths
    unsigned regionSize;  // queue slots per shard = 64 * ceil(capacity / 64 / kShards)
    unsigned space;       // kShards * regionSize: iteration space of every integrator kernel
    int nLights;
    float4* rays;         // [2p] = {o.xyz, tmin}, [2p+1] = {d.xyz, tmax}: extension ray of the path
    float4* hit;          // {bits(tri), t, beta, gamma}
    float4* mult;         // {multTerm.xyz, currEta}
    float4* cnst;         // {constTerm.xyz, bits(flags)}    flags bit0: currDir = -ray.d, bits 8..: metal depth (direct)
    float4* pendL;        // {L.xyz, bits(missMode)}         what the vertex that shot the ray still owes
    float4* hitMul;       // {multiplier applied to multTerm on a hit, bits(rng dimension)}
    float4* vP;           // {P.xyz, bits(material)}          vertex cache shadeA -> shadeB
    float4* vN;           // {N.xyz, 0}
    float4* vTex;         // {texture.xyz, 0}   (direct mode: brdf)
    float4* contrib;      // [p*nLights+i] {unshadowed contribution of light i (direct: irradiance), bits(state)}
    unsigned char* occ;   // [p*nLights+i] 1 = occluded
    float4* ksStack;      // [p*6+k] metal Ks chain (direct mode)
    int* pixel;           // pixel index of the path, -1 = outside the image
    float4* res;          // {radiance of the finished path, bits(RES_*)}: resolved into accum/rgb in sample order
    int* qExt[2];         // ping-pong queues of path slots to extend
    int* qAct;            // Lambertian vertices waiting for their shadow rays
    float4* shRays;       // shadow-ray queue, 2 float4 per ray
    int* shTarget;        // occ index each shadow ray reports to
    unsigned* counters;   // kCounterWords, see counter_index()
    unsigned long long* visitCounters;  // [2] closest: node visits, tri tests; [2] occluded
};

struct FrameParams {
    int mode, level, maxBounces, lastIter;
    unsigned seedHash, sample;      // sample = index of the first sample in flight
    int samplesInFlight;            // path mode: samples of every pixel traced together (1 otherwise)
    unsigned nTilePaths;            // path slots per sample = owned tiles x tile pixels
    int width, height;
    int tileW, tileH, tilesX, nOwnedTiles;
    const int* tileList;  // owned tile ids
    float* accum;
    float* rgb;
};

// cursor: one zeroed word per launch (the work counter the persistent wavefronts pull ray chunks from)
// literal threaded walk (kernels_trace.hip).  remap != nullptr: trace only the queue slots remap[0 .. *remapCount)
void launch_trace_closest(hipStream_t st, const DevScene& s, const float4* rays, const int* index, QueueView q,
                          unsigned* cursor, float4* hits, unsigned long long* counters, int* remap = nullptr,
                          const unsigned* remapCount = nullptr);
void launch_trace_occluded(hipStream_t st, const DevScene& s, const float4* rays, const int* target, QueueView q,
                           unsigned* cursor, unsigned char* occ, unsigned long long* counters, int* remap = nullptr,
                           const unsigned* remapCount = nullptr);
// order-free traversal over the 4-wide tree (kernels_trace_wide.hip) for the rays that qualify; the queue slots of
// the others are appended to slowList (count in *slowCount) for a literal walk afterwards
void launch_trace_wide_closest(hipStream_t st, const DevScene& s, const float4* rays, const int* index, QueueView q,
                               unsigned* cursor, float4* hits, unsigned long long* counters, unsigned* slowCount,
                               int* slowList, unsigned* nextHeader = nullptr);
void launch_trace_wide_occluded(hipStream_t st, const DevScene& s, const float4* rays, const int* target, QueueView q,
                                unsigned* cursor, unsigned char* occ, unsigned long long* counters, unsigned* slowCount,
                                int* slowList, unsigned* nextHeader = nullptr);
// path mode: the extension rays (C) and the shadow rays (O) of one bounce in one launch; shadowFirst picks the order
void launch_trace_wide_both(hipStream_t st, const DevScene& s, const float4* raysC, const int* indexC, QueueView qC, unsigned* cursorC,
                            float4* hits, unsigned long long* countersC, unsigned* slowCountC, int* slowListC, const float4* raysO,
                            const int* targetO, QueueView qO, unsigned* cursorO, unsigned char* occ, unsigned long long* countersO,
                            unsigned* slowCountO, int* slowListO, int shadowFirst, unsigned* nextHeaderC, unsigned* nextHeaderO);
void launch_trace_slow_closest(hipStream_t st, const DevScene& s, const float4* rays, const int* index, float4* hits,
                               unsigned long long* counters, int* list, const unsigned* listCount);
void launch_trace_slow_occluded(hipStream_t st, const DevScene& s, const float4* rays, const int* target, unsigned char* occ,
                                unsigned long long* counters, int* list, const unsigned* listCount);
// StacklessBVH2::intersect, literal (kernels_trace_bittrail.hip); needs s.btOk
void launch_trace_bittrail_closest(hipStream_t st, const DevScene& s, const float4* rays, const int* index, QueueView q,
                                   float4* hits, unsigned long long* counters);
void launch_trace_bittrail_occluded(hipStream_t st, const DevScene& s, const float4* rays, const int* target, QueueView q,
                                    unsigned char* occ, unsigned long long* counters);
void launch_eye_rays(hipStream_t st, const CameraDev& cam, size_t first, size_t count, float4* rays);
void launch_resolve_hits(hipStream_t st, const DevScene& s, const float4* rays, const float4* hits, size_t n,
                         float* out);
void launch_background(hipStream_t st, const DevScene& s, const float* dir, size_t n, float* rgb);

// pt/vec.hpp's primitives over arrays of float3, on the device / on the host (same source, same flags)
void launch_vec_ops(hipStream_t st, int op, const float* a, const float* b, const float* c, const float* d, size_t n, float* out);
void vec_ops_host(int op, const float* a, const float* b, const float* c, const float* d, size_t n, float* out);
void launch_divcheck(hipStream_t st, unsigned seed, unsigned blocks, unsigned perThread, unsigned long long* mismatches);

// integrator (kernels_integrate.hip)
void launch_generate(hipStream_t st, const CameraDev& cam, const FrameParams& fp, const PathBuffers& pb);
void launch_shade_a(hipStream_t st, const DevScene& s, const FrameParams& fp, const PathBuffers& pb, int iter);
void launch_shade_b(hipStream_t st, const DevScene& s, const FrameParams& fp, const PathBuffers& pb, int iter);
// path mode: the single shading kernel of bounce `iter` (kernels_integrate.hip); finalsOnly: only resolve the paths
// that ended at a Lambertian vertex of bounce iter-1
void launch_shade_path(hipStream_t st, const DevScene& s, const FrameParams& fp, const PathBuffers& pb, int iter, bool finalsOnly);
// totals[0] += sum of extension-queue lengths, totals[1] += sum of shadow-queue lengths over iterations [0, nIter)
// rgb (16-byte aligned) -> 8-bit channels as Image::save converts them; out 4-byte aligned
void launch_pack_rgb8(hipStream_t st, const float* rgb, size_t nValues, unsigned char* out);
void launch_resolve_samples(hipStream_t st, const FrameParams& fp, const PathBuffers& pb);
void launch_fold_counters(hipStream_t st, unsigned* counters, int nIter, unsigned long long* totals);
// multi-GPU gather (kernels_tiles.hip): owned tiles of a W x H float3 frame <-> a dense buffer of nTiles x tileW x tileH x 3 floats
void launch_pack_tiles(hipStream_t st, const float* rgb, const int* tiles, unsigned nTiles, int tileW, int tileH, int tilesX,
                       int width, int height, float* out);
void launch_unpack_tiles(hipStream_t st, const float* in, const int* tiles, unsigned nTiles, int tileW, int tileH, int tilesX,
                         int width, int height, float* rgb);

}  // namespace ptd
