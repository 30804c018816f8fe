// device_scene.hpp -- layout of the scene arena in HBM and the records the
// kernels exchange.  One hipMalloc holds every section (256-B aligned):
//
//   nodes     2 x float4 per BVH node (32 B), the two planes of an axis adjacent (packed fp32 math):
//               n0 = {min.x, max.x, min.y, max.y}
//               n1 = {min.z, max.z, bits(idMiss), bits(link)}
//                                            link > 0: idHit of an interior node
//                                            link < 0: leaf, first leaf triangle = ~link,
//                                                      and its idHit == idMiss
//             (threaded links of StacklessBVH, reference cs488.h:1637-1659; 0 = done)
//             Device numbering: the root is 0, the kTopNodes nodes with the largest boxes (a connected
//             top of the tree) come first, the rest follow in preorder; the links are renumbered to match.
//   wnodes    (PT_WIDE_K = 8, the default: eight slots per node, 112 B in 128-B records, see kWideNodeF4 below; what
//             follows describes the four-slot layouts PT_WIDE_K = 4 keeps)
//             (PT_WIDE_HALF) 4 x 16 B per 4-wide node: rows x, y, z = 8 halves {lo of slots 0..3, hi of slots
//             0..3}, fp16 rounded OUTWARDS (lo down, hi up: the slot test only has to be conservative), row 3 =
//             {bits(childBase), bits(leafBase), bits(interiorMask | leafMask << 4), 0}; 64 B, never straddling a line.
//             Scenes with a plane beyond the fp16 range take the literal walk.  With PT_WIDE_HALF=0:
//             "wide": 7 x float4 per 4-wide node (112 B) over the SAME leaves, for rays that qualify for the
//             order-free traversal (kernels_trace.hip: k_trace_wide); built on the host from the binary tree
//             (api_device.hip: build_wide), numbered breadth-first:
//               {min.x of slots 0..3} {max.x} {min.y} {max.y} {min.z} {max.z}
//               {bits(childBase), bits(leafBase), bits(interiorMask | leafMask << 4), 0}
//             interior child of slot k = childBase + popcount(interiorMask & ((1 << k) - 1)),
//             leaf of slot k = leafBase + popcount(leafMask & ((1 << k) - 1))
//   leafHead  3 x float4 per leaf (wide leaf numbering): a copy of the leaf's first leafTris record (third float4's
//             .w = bits(its leafTris index)), so that opening a leaf is ONE round trip to memory
//   leafBox   2 x float4 per leaf (wide leaf numbering): {min.x, max.x, min.y, max.y} {min.z, max.z, bits(first
//             leaf triangle), 0} -- the leaf's exact box for the reference's exact predicate; read only for the
//             few triangles that would change a ray's result, hence apart from leafHead
//   nodeRight int per node: the right child of an interior node (device numbering), 0 for a leaf; k_trace_slow
//   btLinks   int4 per node {idRight, idParent, idSibling, 0} (device numbering): the links of StacklessBVH2::build
//             (cs488.h:1713-1730) for k_trace_bittrail; idLeft is the node's link
//   leafTris  3 x float4 per triangle reference (48 B), in leaf (triList) order:
//               {p0.xyz, bits(original triangle id)}
//               {p0-p1,  bits(1 if last triangle of its leaf)}
//               {p0-p2,  0}
//             the two edge vectors are the first two operands of the
//             reference's Cramer test (cs488.h:782-783), rounded once on the
//             host exactly as the reference rounds them per test
//   triAttr   8 x float4 per triangle (128 B = one cache line per shaded hit; 7 used), original order, for shading only:
//               p0 p1 p2 n0 n1 n2 (18 floats) bits(material id) pad | uv0 uv1 uv2 (6) pad: an untextured vertex reads the
//               first five float4 only (shading.hpp: reconstruct_hit, hit_texcoord)
//   materials 4 x float4 per material: {type, eta, Ns, bits(texel offset or -1)}
//               {Kd.xyz, bits(texW)} {Ks.xyz, bits(texH)} {Ka.xyz, 0}
//   texels    8-bit RGB, row 0 = top row of the file
//   env       float3 per pixel, row 0 = bottom (after Image::load's flip)
//   lights    2 x float4 per light: {position, 0} {wattage, 0}
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ptd {

struct DevScene {
    const float4* nodes;
    const float4* leafTris;
    const float4* triAttr;
    const float4* materials;
    const unsigned char* texels;
    const float* env;
    const float4* lights;
    int nNodes, nLeafTris, nTris, nMaterials, envW, envH, nLights;
    const float4* wnodes;   // 4-wide nodes (nWide == 0: the scene does not qualify, every ray takes the literal walk)
    const float4* leafBox;
    const float4* leafHead;
    const int* nodeRight;   // right child of every interior node (device numbering; 0 for leaves), only when nWide > 0
    const int4* btLinks;    // {idRight, idParent, idSibling, 0} per node (device numbering): StacklessBVH2's links, only when btOk
    const int* roots;       // the root of every object's tree (device numbering; Scene::objects, cs488.h:2006): nRoots >= 1.  The
    int nRoots;             // literal walk needs none of it (tree k's exits lead to root k + 1); k_trace_slow and k_trace_bittrail do
    int btOk;               // the tree is a verified binary tree of depth <= 31: k_trace_bittrail may run
    int slowOk;             // ... of depth <= 96: k_trace_slow's node stack holds every state of its walk (deeper trees hand
                            // their few awkward rays to the literal k_trace, a lane per ray)
    int nWide, nWideTop;    // wide nodes [0, nWideTop) are kept in LDS by k_trace_wide
    float wideInvScale;     // the fp16 planes of the wide nodes are the scene's planes times a power of two S (so that any
                            // tame scene fits fp16's range at full relative precision); this is 1 / S
    int nTopNodes;   // nodes [0, nTopNodes) are the top of the tree, numbered first so that k_trace can hold them in LDS
    int drain;       // k_trace_wide's own wavefronts trace the rays the launch hands back (kernels_trace_wide.hip, "Draining the
                     // hand-back list"); 0 (MIPT_DRAIN=0 when the context is made): k_trace_slow does, in a launch of its own
    int handBackAll; // test switch (MIPT_HAND_BACK_ALL=1 when the context is made): k_trace_wide hands EVERY ray to the exact walk kernels
    int planesTame;  // every box plane is 0 or has magnitude in [2^-61, 2^40] and min <= max: the exact reciprocal slab division may be used (traverse.hpp)
};

// ray record, 32 B: {o.xyz, tmin} {d.xyz, tmax}
// hit record, 16 B: {bits(tri_id or -1), t, beta, gamma}

constexpr int kBlock = 256;
#ifndef PT_TUNE_BLOCK
#define PT_TUNE_BLOCK 1024
#endif
constexpr int kTraceBlock = PT_TUNE_BLOCK;  // k_trace: 16 wavefronts per block, two blocks per CU (LDS: 2 x ~64 KB of 160 KB)
#ifndef PT_TUNE_TOP
#define PT_TUNE_TOP 2032
#endif
#ifndef PT_TUNE_GRID
#define PT_TUNE_GRID 512
#endif
constexpr int kTriAttrF4 = 8;       // float4 per triAttr record
#ifndef PT_WIDE_HALF
#define PT_WIDE_HALF 1  // wide nodes with fp16 planes rounded outwards (64 B, four 16-B rows); 0: fp32 planes (112 B, seven rows)
#endif
#ifndef PT_WIDE_DRAIN
#define PT_WIDE_DRAIN 1  // k_trace_wide's wavefronts trace the rays the launch has handed back themselves ("Draining the hand-back list" in kernels_trace_wide.hip); short lists only: list-mode k_trace follows every launch and looks after what is left (api_device.hip)
#endif
#ifndef PT_WIDE_K
#define PT_WIDE_K 8  // slots per wide node: 8 (default since r2: 12 instead of 19 dependent visits per closest-hit ray) or 4; 8 needs PT_WIDE_HALF
#endif
constexpr int kWideK = PT_WIDE_K;
#if PT_WIDE_K == 8
// EIGHT slots per node: per axis 32 B = {lo of slots 0..7 | hi of slots 0..7} as fp16 rounded outwards, at byte 0 / 32 /
// 64; meta {bits(childBase), bits(leafBase), bits(nInterior | nLeaf << 8), 0} at byte 96.  The INTERIOR slots come
// first (slots 0 .. nInterior-1, then the leaves), so slot k's child is childBase + k and its leaf leafBase + k -
// nInterior, without masks and popcounts.  A ray reads its near planes at (row + 16 [d < 0]) and its far planes at (row
// + 16 [d >= 0]): two 16-B loads per axis, no selects.  112 B in LDS, 128-B records in the arena (one cache line).
constexpr int kWideNodeF4 = 7;
constexpr int kWideNodeStrideF4 = 8;
#ifndef PT_TUNE_WIDE_LDS128
#define PT_TUNE_WIDE_LDS128 1  // 1: the copy of the tree's top in LDS keeps the arena's 128-B records, so a visit forms ONE set of piece addresses for both homes of a node (and no multiply by 112); 0: packed, 112 B per node (753 instead of 659 nodes)
#endif
#elif PT_WIDE_HALF
constexpr int kWideNodeF4 = 4;        // 16-B rows per wide node: x, y, z planes as 8 halves {lo of slots 0..3, hi of slots 0..3}, meta
constexpr int kWideNodeStrideF4 = 4;  // 64-B records, 64-B aligned: a node never straddles a cache line
#else
constexpr int kWideNodeF4 = 7;      // float4 per wide node (its stride in LDS)
#ifndef PT_TUNE_WIDE_NODE_STRIDE
#define PT_TUNE_WIDE_NODE_STRIDE 7
#endif
constexpr int kWideNodeStrideF4 = PT_TUNE_WIDE_NODE_STRIDE;  // ... and in the arena (8 = one 128-B line per node)
#endif
#ifndef PT_TUNE_LEAFHEAD_STRIDE
#define PT_TUNE_LEAFHEAD_STRIDE 3
#endif
#if PT_WIDE_K == 8 && PT_TUNE_WIDE_LDS128
constexpr int kWideNodeLdsF4 = kWideNodeStrideF4;  // float4 per wide node in LDS
#else
constexpr int kWideNodeLdsF4 = kWideNodeF4;
#endif
constexpr int kLeafHeadStrideF4 = PT_TUNE_LEAFHEAD_STRIDE;   // float4 per leafHead record (4 = half a line each)
constexpr int kWideMaxDepth = 16;   // levels of the wide tree k_trace_wide's per-lane stack can hold (binary depth 32)
#ifndef PT_TUNE_WIDE_BLOCK
#define PT_TUNE_WIDE_BLOCK 1024
#endif
#ifndef PT_TUNE_WIDE_PER_CU
#define PT_TUNE_WIDE_PER_CU 1
#endif
// LDS of one k_trace_wide block (PT_TUNE_WIDE_PER_CU blocks share a CU's 160 KB): the per-lane stacks (kWideMaxDepth
// words per lane), the shard prefix sums, and the first kWideTopNodes wide nodes (16 * kWideNodeF4 bytes each)
// PT_WIDE_SHARE (round 3): once the queue is empty, idle lanes of a wavefront adopt parts of their neighbours' stacks
// (kernels_trace_wide.hip, "Sharing the tail"); costs 13 bytes of LDS per lane: a byte to pair lanes, an 8-byte merge
// key and a 4-byte count of helpers still out for the lane's ray
#ifndef PT_WIDE_SHARE
#define PT_WIDE_SHARE (PT_WIDE_K == 8)
#endif
#ifndef PT_TUNE_WIDE_TOP
#define PT_TUNE_WIDE_TOP ((163840 / PT_TUNE_WIDE_PER_CU - 640 - (4 * kWideMaxDepth + 13 * PT_WIDE_SHARE) * PT_TUNE_WIDE_BLOCK) / (16 * kWideNodeLdsF4))
#endif
constexpr int kWideBlocksPerCU = PT_TUNE_WIDE_PER_CU;
constexpr int kWideTopNodes = PT_TUNE_WIDE_TOP;
constexpr int kTraceGrid = PT_TUNE_GRID;  // persistent grid of k_trace: blocks per CU x 256 CUs
constexpr int kTopNodes = PT_TUNE_TOP;   // nodes k_trace keeps in LDS (32 B each; with the shard prefix sums just under 64 KB)

}  // namespace ptd
