// kernels_trace_wide.hip -- the order-free traversal over the wide tree (eight slots per node by default, PT_WIDE_K;
// the text below was written for four and holds for any width).
//
// Why the reference's results do not depend on HOW the leaves are found.  The reference walks its binary tree in a
// fixed order and tests a leaf's triangles iff the leaf's box AND the boxes of all its ancestors pass
// AABB::intersect (cs488.h:571-608, :1661-1683).  For a ray with finite, nonzero direction components and finite
// operands every slab quotient RN(RN(plane - o) / d) is a monotone function of the plane, so for nested boxes
// (child inside parent, which holds exactly for this BVH: a node's box is the min/max of its triangles, so the
// parent's is the union of its children's -- build_wide() verifies it edge by edge) the entry distance of the
// parent is <= the child's and its exit distance >= the child's: a leaf whose own box passes the predicate has
// all its ancestors passing too.  Hence
//     the triangles the reference tests  ==  the triangles of the leaves whose OWN box passes the exact predicate,
// and its answer is the smallest accepted t among them, the first one in depth-first (= leafTris) order on a tie
// (strict '<' in visit order, cs488.h:1674).  Any traversal may be used to find those leaves, in any order, with
// any CONSERVATIVE test on the way (one that never rejects a box the exact predicate accepts), provided that
//   * each candidate leaf is admitted by the exact predicate on its exact box (slab_hit_fast), and
//   * ties in t are broken towards the smaller leafTris index.
// isHit (cs488.h:1684-1705) returns "some accepted triangle exists", which is order-free by itself.
//
// This kernel does that for the rays that qualify (fast_div_ok, in a scene whose planes are tame and whose tree
// build_wide() accepted): a wide tree (eight slots per node since round 2, PT_WIDE_K; four before: a node's slots = a
// binary node's children opened largest box first), all boxes of a node tested in straight-line code with a cheap conservative test (plane
// offset times a reciprocal that carries the slack, instead of the exact division), a per-lane stack of one packed
// entry per visited node with hit children, leaves parked and tested later: triangle first, the exact box
// predicate only for a triangle that would change the result.  One visit replaces about four steps of the threaded
// walk (the walk spends as many issue slots on its exec-mask juggling per box as on arithmetic: DESIGN.md).  Rays
// that do not qualify are appended to a list; NaN rays are answered as misses at once; a short list is traced by the
// wavefronts of this launch as they finish (a wavefront per ray: "Draining the hand-back list" below), whatever is
// left by the literal k_trace after it, a lane per ray (kernels_trace.hip).
#include "kernels.hpp"
#include "shading.hpp"
#include "slow_walk.hpp"

namespace ptd {

#ifndef PT_TUNE_WIDE_BLOCK
#define PT_TUNE_WIDE_BLOCK 1024
#endif
#ifndef PT_TUNE_WIDE_WAVES
#define PT_TUNE_WIDE_WAVES 4
#endif
#ifndef PT_TUNE_WIDE_BURST
#if PT_WIDE_K == 8
#define PT_TUNE_WIDE_BURST 10         // eight-slot nodes: 8: 4.22 ms per closest-hit bench launch, 12: 4.26, 16: 4.40; at the end of r3 (108 VGPRs, cheaper visits): 6: 3.89, 8: 3.83, 10: 3.79, 12: 3.80, 14: 3.88
#else
#define PT_TUNE_WIDE_BURST 16         // visits between two votes, closest hit (r2, fp16 nodes: 6: 5.42 ms per bench launch, 8: 5.25, 12: 5.07, 16: 5.09, 24: 5.35; with the rebuilt hierarchy: 8: 4.72, 12: 4.56, 16: 4.53, 20: 4.64)
#endif
#endif
#ifndef PT_TUNE_WIDE_STATIC_SHIFT
#define PT_TUNE_WIDE_STATIC_SHIFT 2   // the static first chunks cover 1 - 2^-shift of the queue
#endif
#ifndef PT_TUNE_WIDE_CHUNK_MIN
#define PT_TUNE_WIDE_CHUNK_MIN 128u   // smallest chunk the cursor hands out (r2: 64 -> 128: a 1 M-ray launch 152 -> 131 us, a 1/8 share of the bench step 6.5 -> 6.3 ms, the whole step unchanged; 256: the same)
#endif
#ifndef PT_TUNE_WIDE_REFILL
#define PT_TUNE_WIDE_REFILL 16
#endif
#ifndef PT_TUNE_WIDE_LEAF
#define PT_TUNE_WIDE_LEAF 40
#endif
#ifndef PT_TUNE_WIDE_KEEP
#define PT_TUNE_WIDE_KEEP 16
#endif
#ifndef PT_TUNE_WIDE_STUCK
#define PT_TUNE_WIDE_STUCK 8
#endif
#ifndef PT_TUNE_WIDE_GREEDY
#define PT_TUNE_WIDE_GREEDY 0
#endif
#if PT_WIDE_HALF
#undef PT_TUNE_WIDE_FMA
#define PT_TUNE_WIDE_FMA 1  // the fp16 layout implies the fma form
#endif
#ifndef PT_TUNE_WIDE_FMA
#define PT_TUNE_WIDE_FMA 1             // slot test with one fma per plane (wide_hit_mask_fma); 0: subtract and multiply for every ray
#endif
#ifndef PT_TUNE_WIDE_REFILL_ANYHIT
#define PT_TUNE_WIDE_REFILL_ANYHIT PT_TUNE_WIDE_REFILL
#endif
#ifndef PT_TUNE_WIDE_BURST_ANYHIT
#if PT_WIDE_K == 8
#define PT_TUNE_WIDE_BURST_ANYHIT 5   // eight-slot nodes: 2: 1.45 ms per any-hit bench launch, 3: 1.37, 4: 1.39, 6: 1.31, 8: 1.35; at the end of r3 (88 VGPRs): 3: 1.08, 4: 1.14, 5: 0.96, 6: 1.05, 8: 1.14 (not monotonic, and the same on every run: profiles/r03/tune_sweep_r3_end.txt)
#else
#define PT_TUNE_WIDE_BURST_ANYHIT 4   // ... any hit: rays end early, shorter bursts waste fewer visits (3: 1.70 ms, 4: 1.62, 6: 1.67)
#endif
#endif
#ifndef PT_TUNE_WIDE_LEAF_ANYHIT
#define PT_TUNE_WIDE_LEAF_ANYHIT 32    // any-hit rays hold fewer leaves: start their leaf phase earlier (r2: 40/8/16 1.50 ms per bench launch, 32/16/8 1.41)
#endif
#ifndef PT_TUNE_WIDE_KEEP_ANYHIT
#define PT_TUNE_WIDE_KEEP_ANYHIT 8
#endif
#ifndef PT_TUNE_WIDE_STUCK_ANYHIT
#define PT_TUNE_WIDE_STUCK_ANYHIT 16
#endif
// The any-hit instantiation runs FIVE wavefronts per SIMD: 256-thread blocks (one wavefront per SIMD), five per CU, each
// with its own 16 KB of stacks and a 216-node top of the tree (30.5 KB of LDS per block; 32.7 KB each did not fit five).
// Measured with the rebuilt hierarchy: 1.375 -> 1.28 ms per any-hit bench launch.  The closest-hit instantiation
// needs 101 VGPRs (5 wavefronts allow 96: three spilled registers) and gains 1.3 % at equal LDS but loses 2.7 % against
// its big-LDS configuration, so it stays at one 1024-thread block per CU, 4 wavefronts per SIMD.
#if PT_WIDE_K == 8
// ... with eight-slot nodes (112 B, 121 VGPRs in the closest-hit and ~100 in the any-hit instantiation) five wavefronts
// spill 23 registers: both instantiations run one 1024-thread block per CU (any-hit 1.38 ms against 1.82 at five).
// (End of round 3: 103 / 92 VGPRs; two 640-thread blocks per CU at five wavefronts still spill 9 to 17 and halve the
// tree top in LDS: 5.3 ms per closest-hit bench launch against 3.8.)
#ifndef PT_TUNE_WIDE_BLOCK_ANYHIT
#define PT_TUNE_WIDE_BLOCK_ANYHIT 1024
#endif
#ifndef PT_TUNE_WIDE_WAVES_ANYHIT
#define PT_TUNE_WIDE_WAVES_ANYHIT 4
#endif
#ifndef PT_TUNE_WIDE_PER_CU_ANYHIT
#define PT_TUNE_WIDE_PER_CU_ANYHIT 1
#endif
#ifndef PT_TUNE_WIDE_TOP_ANYHIT
#define PT_TUNE_WIDE_TOP_ANYHIT kWideTopNodes
#endif
#else
#ifndef PT_TUNE_WIDE_BLOCK_ANYHIT
#define PT_TUNE_WIDE_BLOCK_ANYHIT 256
#endif
#ifndef PT_TUNE_WIDE_WAVES_ANYHIT
#define PT_TUNE_WIDE_WAVES_ANYHIT 5
#endif
#ifndef PT_TUNE_WIDE_PER_CU_ANYHIT
#define PT_TUNE_WIDE_PER_CU_ANYHIT 5
#endif
#ifndef PT_TUNE_WIDE_TOP_ANYHIT
#define PT_TUNE_WIDE_TOP_ANYHIT (13824 / (16 * kWideNodeF4))  // 13.5 KB of nodes beside 16 KB of stacks: 216 64-B nodes
#endif
#endif
template <bool ANYHIT> struct WideCfg {
    static constexpr int kBlock = ANYHIT ? PT_TUNE_WIDE_BLOCK_ANYHIT : PT_TUNE_WIDE_BLOCK;
    static constexpr int kWaves = ANYHIT ? PT_TUNE_WIDE_WAVES_ANYHIT : PT_TUNE_WIDE_WAVES;
    static constexpr int kPerCU = ANYHIT ? PT_TUNE_WIDE_PER_CU_ANYHIT : kWideBlocksPerCU;
    static constexpr int kTop = ANYHIT ? (PT_TUNE_WIDE_TOP_ANYHIT < kWideTopNodes ? PT_TUNE_WIDE_TOP_ANYHIT : kWideTopNodes) : kWideTopNodes;
};
#ifndef PT_TUNE_WIDE_GROUPS
#define PT_TUNE_WIDE_GROUPS 4
#endif
constexpr int kWideGroups = PT_TUNE_WIDE_GROUPS;  // leaf groups a lane may hold (one group = the leaf slots of one wide node)

// The conservative slot test.  With y = RN(1/d): q~ = RN(RN(p - o) * y) and the exact q = RN(RN(p - o) / d)
// satisfy |q~ - q| <= 2^-22 |q| (three roundings of 2^-24 each), q~ and q have the same sign and vanish together.
// L(x) = x - 2^-21|x| and U(x) = x + 2^-21|x| are increasing, L(q~) <= q <= U(q~), and min/max commute with
// increasing maps, so t1 >= L(t1~), t2 <= U(t2~); the exact predicate max(t1, 0) <= t2 therefore implies
// max(L(t1~), 0) <= U(t2~).  The code widens by more than that and does it once per ray instead of once per slot:
// near planes are multiplied by yn = RN(y (1 - 2^-19)), far planes by yf = RN(y (1 + 2^-19)), so a positive near
// quotient comes out below (s/d)(1 - 2^-19)(1 + 2^-24)^3 < (s/d)(1 - 2^-24) <= q and a positive far quotient above
// q; a negative near quotient stays negative (the clamp at 0 makes its size irrelevant) and a negative far
// quotient stays negative (both forms reject).  No product underflows: fast_div_ok bounds the magnitudes.
// nearx .. farz: the node's plane rows already picked by the ray's direction signs (min row where d > 0, max row
// where d < 0 for "near", the other one for "far"): the products are monotone in the plane, so these ARE
// min / max of the two candidates, without computing both.
// (Folding subtraction and multiplication into one fma per plane, fma(p, y, -o*y), needs an absolute slack of
// 2^-24 |o*y| for the cancellation; for rays with one tiny direction component that slack is as large as the
// scene, those rays visit thousands of nodes and the launch waits for them: measured 20 % slower, not kept.)
__device__ __forceinline__ int wide_hit_mask(const float4v nearx, const float4v farx, const float4v neary, const float4v fary,
                                             const float4v nearz, const float4v farz, const float ox, const float oy,
                                             const float oz, const float2v ynx, const float2v yny, const float2v ynz,
                                             const float2v yfx, const float2v yfy, const float2v yfz) {
    // yn* = RN(y (1 - 2^-19)), yf* = RN(y (1 + 2^-19)), both lanes of a pair equal: the factors of L above, folded into
    // the reciprocals once per ray (three roundings of 2^-24 each leave more than the 2^-20 the argument needs)
    const float4v sx0 = nearx - ox, sx1 = farx - ox, sy0 = neary - oy, sy1 = fary - oy, sz0 = nearz - oz, sz1 = farz - oz;
    const float2v nxa = sx0.xy * ynx, nxb = sx0.zw * ynx, fxa = sx1.xy * yfx, fxb = sx1.zw * yfx;
    const float2v nya = sy0.xy * yny, nyb = sy0.zw * yny, fya = sy1.xy * yfy, fyb = sy1.zw * yfy;
    const float2v nza = sz0.xy * ynz, nzb = sz0.zw * ynz, fza = sz1.xy * yfz, fzb = sz1.zw * yfz;
    const float u0 = fmaxf(fmaxf(fmaxf(nxa.x, nya.x), nza.x), 0.0f), v0 = fminf(fminf(fxa.x, fya.x), fza.x);
    const float u1 = fmaxf(fmaxf(fmaxf(nxa.y, nya.y), nza.y), 0.0f), v1 = fminf(fminf(fxa.y, fya.y), fza.y);
    const float u2 = fmaxf(fmaxf(fmaxf(nxb.x, nyb.x), nzb.x), 0.0f), v2 = fminf(fminf(fxb.x, fyb.x), fzb.x);
    const float u3 = fmaxf(fmaxf(fmaxf(nxb.y, nyb.y), nzb.y), 0.0f), v3 = fminf(fminf(fxb.y, fyb.y), fzb.y);
    return (!(u0 > v0) ? 1 : 0) | (!(u1 > v1) ? 2 : 0) | (!(u2 > v2) ? 4 : 0) | (!(u3 > v3) ? 8 : 0);
}

// The same test with ONE fused multiply-add per plane, t~ = fma(p, y', -c), instead of a subtraction and a product:
// 24 v_fma_f32 per visit in place of 24 v_sub_f32 + 12 v_pk_mul_f32, about half the issue cycles of the arithmetic.
// What is lost is the exact cancellation of p - o: the constant c = o y' is rounded, so a plane that coincides with
// the ray's origin no longer gives exactly 0, and the slack has to be absolute.  With u = 2^-24, A = o y' (real),
// c0 = RN(A), near: y' = yn = RN(y (1 - 2^-19)), cn = RN(c0 + 2^-22 |c0|)  ==>  cn - A in [1.5 u |A|, 7 u |A|], hence
//     t~n = RN(p yn - cn) = RN((p - o) yn - D),  D in [1.5, 7] u |A|,   (p - o) yn = T r,  r < 1 - 2^-20,
// where T = (p - o) / d is the real quotient and the reference's q = RN(RN(p - o) / d) lies within 2.0001 u |T| of it
// with the same sign.  T > 0: t~n <= max(T r (1 + u), 0) <= T (1 - 2^-21) <= q.  T <= 0: T r - D <= 0, so t~n <= 0 and
// the clamp at 0 makes its size irrelevant.  Far planes mirror this (yf = RN(y (1 + 2^-19)), cf = RN(c0f - 2^-22 |c0f|)):
// T >= 0 gives t~f >= T (1 + 2^-21) >= q, and T < 0 is rejected by the reference whatever t~f is.  So the exact
// predicate max(t1, 0) <= t2 implies max(t~1, 0) <= t~2: conservative, as the order-free traversal requires.
// The price is a slack of up to 7 u |o| in SPACE along each axis, i.e. 7 u |o y| in t: harmless while |y| is moderate,
// but a ray with a tiny direction component (huge y) that starts ON a box plane -- every secondary ray starts on a
// wall -- would then pass every box that touches that plane within a long stretch of t.  Such rays (a direction
// component below 2^-16, 0.5 in 10,000; round 2: 2^-12) keep the subtract-and-multiply form: `precise` lanes, same kernel.
__device__ __forceinline__ int wide_hit_mask_fma(const float4v nearx, const float4v farx, const float4v neary, const float4v fary,
                                                 const float4v nearz, const float4v farz, const float ynx, const float yny,
                                                 const float ynz, const float yfx, const float yfy, const float yfz,
                                                 const float cnx, const float cny, const float cnz, const float cfx,
                                                 const float cfy, const float cfz) {
#define PT_F(P, Y, C) __builtin_fmaf(P, Y, -(C))
    const float u0 = fmaxf(fmaxf(fmaxf(PT_F(nearx.x, ynx, cnx), PT_F(neary.x, yny, cny)), PT_F(nearz.x, ynz, cnz)), 0.0f);
    const float v0 = fminf(fminf(PT_F(farx.x, yfx, cfx), PT_F(fary.x, yfy, cfy)), PT_F(farz.x, yfz, cfz));
    const float u1 = fmaxf(fmaxf(fmaxf(PT_F(nearx.y, ynx, cnx), PT_F(neary.y, yny, cny)), PT_F(nearz.y, ynz, cnz)), 0.0f);
    const float v1 = fminf(fminf(PT_F(farx.y, yfx, cfx), PT_F(fary.y, yfy, cfy)), PT_F(farz.y, yfz, cfz));
    const float u2 = fmaxf(fmaxf(fmaxf(PT_F(nearx.z, ynx, cnx), PT_F(neary.z, yny, cny)), PT_F(nearz.z, ynz, cnz)), 0.0f);
    const float v2 = fminf(fminf(PT_F(farx.z, yfx, cfx), PT_F(fary.z, yfy, cfy)), PT_F(farz.z, yfz, cfz));
    const float u3 = fmaxf(fmaxf(fmaxf(PT_F(nearx.w, ynx, cnx), PT_F(neary.w, yny, cny)), PT_F(nearz.w, ynz, cnz)), 0.0f);
    const float v3 = fminf(fminf(PT_F(farx.w, yfx, cfx), PT_F(fary.w, yfy, cfy)), PT_F(farz.w, yfz, cfz));
#undef PT_F
    return (!(u0 > v0) ? 1 : 0) | (!(u1 > v1) ? 2 : 0) | (!(u2 > v2) ? 4 : 0) | (!(u3 > v3) ? 8 : 0);
}
#ifndef PT_TUNE_WIDE_PRECISE_EXP
#define PT_TUNE_WIDE_PRECISE_EXP (-20)  // rays with a direction component below 2^this do not take the one-fma slot test (its slack is absolute: such a ray passes every box that touches its start plane): they are handed to the exact walk, or, with PT_TUNE_WIDE_PRECISE_HANDBACK=0, keep the subtract-and-multiply form as `precise` lanes (a wavefront that holds one then runs BOTH forms on every visit: r3, ms per closest-hit bench launch: -10: 4.16, -12: 4.05, -14 .. -20: 4.02, -24: 4.03)
#endif
#ifndef PT_TUNE_WIDE_PRECISE_HANDBACK
#define PT_TUNE_WIDE_PRECISE_HANDBACK 1  // 1: rays with a direction component below 2^PT_TUNE_WIDE_PRECISE_EXP are handed to the exact walk (a few dozen per launch, traced by the wavefronts that finish first) and the kernel has no subtract-and-multiply slot test: 128 -> 108 VGPRs (closest hit), 110 -> 88 (any hit).  0: they stay, as `precise` lanes
#endif
#ifndef PT_TUNE_WIDE_CARRY_ID
#define PT_TUNE_WIDE_CARRY_ID 1  // 1: the winning triangle's id is carried in a register (there is room since the `precise` form left); 0: looked up in leafTris when the ray finishes, a dependent load every wavefront waited for after nearly every burst
#endif
#ifndef PT_TUNE_WIDE_BURST_UNROLL
#define PT_TUNE_WIDE_BURST_UNROLL 1  // visits of a burst per turn of its loop (0: unrolled in full)
#endif
#define PT_PRAGMA_(X) _Pragma(#X)
#define PT_PRAGMA(X) PT_PRAGMA_(X)
#ifndef PT_TUNE_WIDE_GROUPS_ANYHIT
#define PT_TUNE_WIDE_GROUPS_ANYHIT PT_TUNE_WIDE_GROUPS  // 1: an any-hit ray's leaves are opened before it walks on -- the bench scene, where most shadow rays end at a triangle, gains (41.9 instead of 47.3 slot tests per shadow ray, any-hit launch 0.97 -> 0.92 ms, step -0.6 %), scenes whose shadow rays mostly reach the light lose (bunny +1.8 %, direct-lit bunny +2.7 %, testObj +0.9 %): not the default (profiles/r03 r3v10 against r3v9)
#endif
#ifndef PT_TUNE_WIDE_STREAM
#define PT_TUNE_WIDE_STREAM 3  // bit 0: loads, bit 1: stores.  rays, queue slots and hit records pass through once: non-temporal loads and stores, so that they do not push the tree out of L2
#endif
#if PT_TUNE_WIDE_STREAM
__device__ __forceinline__ float4 stream_load(const float4* p) {
    const float4v v = __builtin_nontemporal_load((const float4v*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int stream_load(const int* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stream_store(float4* p, const float4 v) { __builtin_nontemporal_store(float4v{v.x, v.y, v.z, v.w}, (float4v*)p); }
#if PT_TUNE_WIDE_STREAM & 1
#define PT_STREAM_LOAD(P) stream_load(P)
#else
#define PT_STREAM_LOAD(P) (*(P))
#endif
#if PT_TUNE_WIDE_STREAM & 2
#define PT_STREAM_STORE(P, V) stream_store((P), (V))
#else
#define PT_STREAM_STORE(P, V) (*(P) = (V))
#endif
#else
#define PT_STREAM_LOAD(P) (*(P))
#define PT_STREAM_STORE(P, V) (*(P) = (V))
#endif
#ifndef PT_TUNE_WIDE_FLAT_STACK
#define PT_TUNE_WIDE_FLAT_STACK 1  // the stack steps of a visit with selects instead of nested branches (r3: closest-hit launch 3.72 -> 3.71 ms, any-hit 0.945 -> 0.936, four A/B pairs of four)
#endif
#ifndef PT_TUNE_WIDE_SIGNMASK
#define PT_TUNE_WIDE_SIGNMASK 1  // the slot test's hit mask from the sign bits of v - u (wide_hit_mask_fma_h)
#endif
#if PT_WIDE_HALF
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
// wide_hit_mask_fma on fp16 planes: the conversion rides in the instruction (v_fma_mix_f32 takes an fp16 operand), so a
// slot costs six instructions of arithmetic and nothing to unpack.  An fp16 value is an fp32 value: the argument above
// holds with p = the stored plane, and a stored plane lies outside the box's (lo rounded down, hi rounded up), where
// the reference's quotient is monotone in the plane.
__device__ __forceinline__ int wide_hit_mask_fma_h(const half4v nearx, const half4v farx, const half4v neary, const half4v fary,
                                                   const half4v nearz, const half4v farz, const float ynx, const float yny,
                                                   const float ynz, const float yfx, const float yfy, const float yfz,
                                                   const float cnx, const float cny, const float cnz, const float cfx,
                                                   const float cfy, const float cfz) {
#define PT_F(P, Y, C) __builtin_fmaf((float)(P), Y, -(C))
    const float u0 = fmaxf(fmaxf(fmaxf(PT_F(nearx.x, ynx, cnx), PT_F(neary.x, yny, cny)), PT_F(nearz.x, ynz, cnz)), 0.0f);
    const float v0 = fminf(fminf(PT_F(farx.x, yfx, cfx), PT_F(fary.x, yfy, cfy)), PT_F(farz.x, yfz, cfz));
    const float u1 = fmaxf(fmaxf(fmaxf(PT_F(nearx.y, ynx, cnx), PT_F(neary.y, yny, cny)), PT_F(nearz.y, ynz, cnz)), 0.0f);
    const float v1 = fminf(fminf(PT_F(farx.y, yfx, cfx), PT_F(fary.y, yfy, cfy)), PT_F(farz.y, yfz, cfz));
    const float u2 = fmaxf(fmaxf(fmaxf(PT_F(nearx.z, ynx, cnx), PT_F(neary.z, yny, cny)), PT_F(nearz.z, ynz, cnz)), 0.0f);
    const float v2 = fminf(fminf(PT_F(farx.z, yfx, cfx), PT_F(fary.z, yfy, cfy)), PT_F(farz.z, yfz, cfz));
    const float u3 = fmaxf(fmaxf(fmaxf(PT_F(nearx.w, ynx, cnx), PT_F(neary.w, yny, cny)), PT_F(nearz.w, ynz, cnz)), 0.0f);
    const float v3 = fminf(fminf(PT_F(farx.w, yfx, cfx), PT_F(fary.w, yfy, cfy)), PT_F(farz.w, yfz, cfz));
#undef PT_F
#if PT_TUNE_WIDE_SIGNMASK
    // slot k is missed iff u_k > v_k iff the float v_k - u_k is negative (a difference of two finite floats has the sign
    // of the exact difference, and is -0 never: x - x = +0; every t here is finite, see fast_div_ok): its sign bit goes
    // straight into the mask -- v_sub_f32 + v_alignbit_b32 (m << 1 | sign) per slot instead of a compare into vcc, a
    // select on vcc and an or
    unsigned m = __builtin_amdgcn_alignbit(0u, (unsigned)f2i(v3 - u3), 31u);
    m = __builtin_amdgcn_alignbit(m, (unsigned)f2i(v2 - u2), 31u);
    m = __builtin_amdgcn_alignbit(m, (unsigned)f2i(v1 - u1), 31u);
    m = __builtin_amdgcn_alignbit(m, (unsigned)f2i(v0 - u0), 31u);
    return (int)(~m & 15u);
#else
    return (!(u0 > v0) ? 1 : 0) | (!(u1 > v1) ? 2 : 0) | (!(u2 > v2) ? 4 : 0) | (!(u3 > v3) ? 8 : 0);
#endif
}
// fp16 planes back to the scene's scale (k = DevScene::wideInvScale, a power of two: exact)
__device__ __forceinline__ float4v widen(const half4v h, const float k) { return float4v{(float)h.x * k, (float)h.y * k, (float)h.z * k, (float)h.w * k}; }
#endif
// the subtract-and-multiply form on scalar reciprocals (the `precise` lanes of a kernel built with the fma form)
__device__ __forceinline__ int wide_hit_mask_sub(const float4v nearx, const float4v farx, const float4v neary, const float4v fary,
                                                 const float4v nearz, const float4v farz, const float ox, const float oy,
                                                 const float oz, const float ynx, const float yny, const float ynz,
                                                 const float yfx, const float yfy, const float yfz) {
    return wide_hit_mask(nearx, farx, neary, fary, nearz, farz, ox, oy, oz, float2v{ynx, ynx}, float2v{yny, yny}, float2v{ynz, ynz},
                         float2v{yfx, yfx}, float2v{yfy, yfy}, float2v{yfz, yfz});
}

// entry of the per-lane stacks: base << 8 | kindMask << 4 | pendingMask (pending != 0, so an entry is never 0);
// slot k of the group is item  base + popcount(kindMask & ((1 << k) - 1))
__device__ __forceinline__ unsigned pack_group(int base, int kindMask, int pending) {
    return ((unsigned)base << 8) | ((unsigned)kindMask << 4) | (unsigned)pending;
}
__device__ __forceinline__ int group_take(unsigned& e) {  // removes the lowest pending slot, returns its item
    const unsigned pend = e & 15u;
    const unsigned bit = pend & (0u - pend);
    e &= ~bit;
    return (int)(e >> 8) + __popc((e >> 4) & 15u & (bit - 1u));
}

#if PT_WIDE_K == 8
// eight-slot nodes keep interior slots first: an entry is base << 8 | pending, slot k's item is base + k (for leaf groups
// the pending bits are already shifted down by the number of interior slots)
__device__ __forceinline__ int group_take8(unsigned& e) {  // removes the lowest pending slot, returns its item
    const int k = __builtin_ctz(e);  // the pending bits are the lowest of the entry and not all zero
    e &= e - 1u;
    return (int)(e >> 8) + k;
}
constexpr unsigned kPendMask = 255u;
#define PT_GROUP_TAKE group_take8
#else
constexpr unsigned kPendMask = 15u;
#define PT_GROUP_TAKE group_take
#endif

#ifndef PT_WIDE_POOL
#define PT_WIDE_POOL 1  // rays reach the wavefronts through a pool per BLOCK in LDS (below: "The block's pool"); 0: one chunk per wavefront from the global cursor
#endif
#ifndef PT_TUNE_WIDE_POOL_MIN
#define PT_TUNE_WIDE_POOL_MIN 256u    // smallest / largest chunk a block asks the launch's cursor for
#endif
#ifndef PT_TUNE_WIDE_POOL_SHIFT
#define PT_TUNE_WIDE_POOL_SHIFT 1     // a block asks for its share of what is left, shifted right by this
#endif
#ifndef PT_TUNE_WIDE_POOL_MAX
#define PT_TUNE_WIDE_POOL_MAX 8192u
#endif
#ifndef PT_TUNE_WIDE_TAIL_BURST
#define PT_TUNE_WIDE_TAIL_BURST 2   // once the queue is empty: visits between two votes (0: as while it is not).  r3, with the greedy rule below: a 64-ray launch 58 -> 47 us, 262,144 rays 94 -> 85 us, the 1/8 share of the bench frame 4.80 -> 4.77 ms (4: the same)
#endif
#ifndef PT_TUNE_WIDE_TAIL_GREEDY
#define PT_TUNE_WIDE_TAIL_GREEDY 1  // once the queue is empty: the phase with more lanes to serve runs next
#endif
#ifndef PT_TUNE_WIDE_SHARE_MIN
#define PT_TUNE_WIDE_SHARE_MIN 4  // idle lanes before a hand-over round is worth its instructions (tail of a launch only)
#endif
#ifndef PT_DIAG_UTIL
#define PT_DIAG_UTIL 0  // diagnostic builds only: 1 / 2 turn the counters into lanes-active / lanes-issued of the visit / leaf phase
#endif
#define PT_WIDE_ATTR __attribute__((amdgpu_waves_per_eu(WideCfg<ANYHIT>::kWaves, WideCfg<ANYHIT>::kWaves)))

#ifndef PT_DIAG_TAIL
#define PT_DIAG_TAIL 0  // diagnostic builds only (tools/tail_diag.py): wall-clock stamps of every wavefront's start, queue-empty moment and end
#endif
#if PT_DIAG_TAIL
// per wavefront of the last closest-hit launch: {start, queue-empty, end, hand-overs} (plain stores: atomics on a few words
// from 4,096 wavefronts would themselves stretch the launch)
__device__ unsigned long long g_tailDiag[2 * 4 * 4096];  // [any-hit][wavefront][4]
#endif

// LDS of one block, declared by the kernels and handed to the body (a body inlined twice must not declare it twice)
struct WideLds {
    float4* top;                  // the first nTop wide nodes, packed (kWideNodeF4 float4 each)
    unsigned* stack;              // per-lane stacks: entry d of lane t at [d * block + t]
    const unsigned* prefix;       // prefix sums over the queue's 64 shard counts (kShards + 1 words)
    unsigned char* pair;          // hand-over round: rank of a giver -> its lane
    unsigned long long* key;      // per owner lane: min over its helpers of t << 32 | index
    unsigned* out;                // per owner lane: helpers still out | failed << 31
    unsigned long long* pool;     // the block's pool of rays: next dense index << 32 | end of the chunk
    unsigned* poolCtl;            // [0] lock of the one wavefront that refills the pool, [1] the launch's queue is empty
};
#define PT_WIDE_LDS_DECL(KBLK, KTOP)                                                                       \
    __shared__ float4 ldsTop[kWideNodeLdsF4 * (KTOP)];                                                     \
    __shared__ unsigned ldsStack[kWideMaxDepth * (KBLK)];                                                  \
    __shared__ unsigned ldsPrefix[2][kShards + 1];                                                         \
    __shared__ unsigned char ldsPair[(PT_WIDE_SHARE && PT_WIDE_K == 8) ? (KBLK) : 1];                      \
    __shared__ unsigned long long ldsKey[(PT_WIDE_SHARE && PT_WIDE_K == 8) ? (KBLK) : 1];                  \
    __shared__ unsigned ldsOut[(PT_WIDE_SHARE && PT_WIDE_K == 8) ? (KBLK) : 1];                            \
    __shared__ unsigned long long ldsPool[2];                                                              \
    __shared__ unsigned ldsPoolCtl[2][4]
// the top of the tree into LDS (packed there, kWideNodeStrideF4 apart in the arena)
__device__ __forceinline__ void wide_load_top(const DevScene& s, float4* top, int nTop, int kBlk) {
    for (int i = threadIdx.x; i < kWideNodeLdsF4 * nTop; i += kBlk)
        top[i] = s.wnodes[(i / kWideNodeLdsF4) * kWideNodeStrideF4 + i % kWideNodeLdsF4];
}
// static first chunks: a wavefront's share of three quarters of the queue, 64 .. 1024 rays (see "The queue cursor" in the body)
__device__ __forceinline__ unsigned wide_static_chunk(unsigned total, unsigned nWaves) {
    const unsigned c0 = ((total - (total >> PT_TUNE_WIDE_STATIC_SHIFT)) / nWaves) & ~63u;
    return c0 < 64u ? 64u : (c0 > 1024u ? 1024u : c0);
}
// inclusive scan over a sharded queue's 64 counts and the block's pool with its static first chunk, by the block's
// wavefront `wave`
__device__ __forceinline__ void wide_prefix(const QueueView& q, unsigned* prefix, int wave, unsigned long long* pool, unsigned* poolCtl,
                                            int wavesPerBlock) {
    if ((int)(threadIdx.x >> 6) != wave) return;
    const int lane = threadIdx.x & 63;
    unsigned total = q.space;
    if (q.counts) {
        unsigned v = q.counts[(size_t)lane * kCntStride];
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned up = __shfl_up(v, off);
            if (lane >= off) v += up;
        }
        prefix[lane + 1] = v;
        if (lane == 0) prefix[0] = 0;
        total = (unsigned)__shfl((int)v, 63);
    }
    if (lane == 0) {  // block b owns rays [b C0, (b + 1) C0), C0 = its wavefronts' static chunks together
        const unsigned long long C0 = (unsigned long long)wide_static_chunk(total, gridDim.x * (unsigned)wavesPerBlock) * (unsigned)wavesPerBlock;
        const unsigned long long first = (unsigned long long)blockIdx.x * C0;
        const unsigned base = first < (unsigned long long)total ? (unsigned)first : total;
        const unsigned len = (unsigned long long)(total - base) < C0 ? (total - base) : (unsigned)C0;
        *pool = ((unsigned long long)base << 32) | (unsigned long long)(base + len);
        poolCtl[0] = 0u;
        poolCtl[1] = (unsigned long long)gridDim.x * C0 >= (unsigned long long)total ? 1u : 0u;  // nothing behind the static chunks
    }
}

template <bool ANYHIT, bool COUNT>
__device__ __forceinline__ void wide_body(
    const DevScene& s, const float4* __restrict__ rays, const int* __restrict__ index, const int* __restrict__ target,
    const QueueView q, unsigned* __restrict__ cursor, float4* __restrict__ hits, unsigned char* __restrict__ occ,
    unsigned long long* __restrict__ counters, unsigned* __restrict__ slowCount, int* __restrict__ slowList, const WideLds lds) {
    constexpr int kBlk = WideCfg<ANYHIT>::kBlock;  // threads per block; shadows nothing: the launchers use the same WideCfg
    float4* const top = lds.top;
    unsigned* const stackLds = lds.stack;
    const unsigned* const prefix = lds.prefix;
    // Sharing the tail (PT_WIDE_SHARE).  A launch ends with every wavefront draining the rays it still holds: lanes fall
    // idle one by one while the longest ray of the 64 walks on alone -- 0.1 to 0.2 ms per launch in which the chip does
    // next to nothing, ten times per frame.  The traversal is ORDER-FREE (see the top of this file): the answer is the
    // smallest accepted t over the admitted leaves, ties to the smaller leafTris index, however the leaves are found --
    // so they may as well be found by SEVERAL lanes.  Once the queue is empty, an idle lane adopts the bottom entry of
    // a busy lane's stack (the pending children of the shallowest node that lane still owes a visit: its largest piece
    // of work), loads the same ray and walks that part on its own, with its own unverified best and its own verdicts.
    //  * any hit: nothing to merge -- whichever lane finds an admitted, accepted triangle sets the ray's flag.
    //  * closest hit: a helper that has walked its part folds (t, leafTris index) into the owner's 64-bit key in LDS
    //    with an atomic minimum (t > 0, so the float's bits order like the number; the index in the low word is the
    //    reference's tie-break) and counts itself off; the owner writes the hit once its own walk is over and no
    //    helper is out.  If a helper's triangle wins, the owner tests that one triangle again to get beta and gamma
    //    (same operands, same instructions, same bits).  A failed verdict anywhere (deferred admission, below) marks
    //    the ray, and the owner hands it to the exact walk as before.
    // Lanes of one wavefront run in lockstep and its LDS operations complete in program order, so no more than the
    // compiler's ordering is needed.  The COUNT instantiations do not share: their visit counts stay a function of the
    // ray alone.
    // leaf groups a lane may hold before it stops visiting (one group = the hit leaf slots of one wide node); see
    // PT_TUNE_WIDE_GROUPS_ANYHIT for what one group per any-hit ray does
    constexpr int kG = ANYHIT ? PT_TUNE_WIDE_GROUPS_ANYHIT : kWideGroups;
    constexpr bool SHARE = PT_WIDE_SHARE && (PT_WIDE_K == 8) && !COUNT;
    unsigned char* const pairLds = lds.pair;
    unsigned long long* const shareKey = lds.key;
    unsigned* const shareOut = lds.out;
    const int lane = threadIdx.x & 63;
    const unsigned waveBase = threadIdx.x & ~63u;
    const int nTop = s.nWideTop < WideCfg<ANYHIT>::kTop ? s.nWideTop : WideCfg<ANYHIT>::kTop;
    const unsigned total = q.counts ? prefix[kShards] : q.space;
    const __attribute__((address_space(3))) float4v* topLds = (const __attribute__((address_space(3))) float4v*)top;
    const __attribute__((address_space(1))) float4v* nodesGlobal = (const __attribute__((address_space(1))) float4v*)s.wnodes;
    const unsigned headOfs = (unsigned)((const char*)s.leafHead - (const char*)s.leafTris);

    // lane state
    bool hasRay = false;
    int w = -1;             // next wide node to visit (-1: the walk is over)
    unsigned tos = 0;       // top of the stack of interior groups (0: none); older entries in stack[0..sp)
    __attribute__((address_space(3))) unsigned* stack = (__attribute__((address_space(3))) unsigned*)stackLds + threadIdx.x;
    int sp = 0;  // in units of kBlk entries
    int bot = 0;   // bottom of the stack, same units: entries below it were given to helpers (PT_WIDE_SHARE)
    int role = -1; // -1: the lane owns its ray alone | -2: owns it and has given parts away | >= 0: helps that lane of this wavefront
    // leaf groups still to be opened (LIFO, g[0] on top; order is free) and the open leaf's triangle cursor
    int groups = 0;
    unsigned g[kG] = {};  // compile-time indices only: registers
    int cur = -1, curLeaf = 0;
    unsigned raySlot = 0, queueSlot = 0;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 1, yx = 0, yy = 0, yz = 1;  // ray; y = RN(1/d)
    int sx = 0, sy = 0, sz = 0;  // 1 where the direction component is negative
#if PT_WIDE_K == 8
    unsigned oNx = 0, oFx = 16, oNy = 32, oFy = 48, oNz = 64, oFz = 80;  // byte offsets of this ray's near / far half-rows in a node
#endif
#if PT_TUNE_WIDE_FMA
    float ynx = 0, yny = 0, ynz = 0, yfx = 0, yfy = 0, yfz = 0;  // y shrunk / grown, see wide_hit_mask
    float cnx = 0, cny = 0, cnz = 0, cfx = 0, cfy = 0, cfz = 0;  // o y' widened outwards, see wide_hit_mask_fma
    bool precise = false;  // a tiny direction component: this ray keeps the subtract-and-multiply slot test
#else
    float2v ynx = {0, 0}, yny = {0, 0}, ynz = {0, 0}, yfx = {0, 0}, yfy = {0, 0}, yfz = {0, 0};  // y shrunk / grown, see wide_hit_mask
#endif
    float tMin = 0.0f, tMax = 0.0f;
    int bestTri = -1, bestSeq = 0x7fffffff;
    float bestT = 3.402823466e+38f, bestB = 0.0f, bestG = 0.0f;
    // Deferred admission.  A triangle counts only if its leaf's EXACT box passes the reference's predicate; that test
    // (two loads, ~35 instructions) is needed by the one or two lanes of an iteration whose triangle was accepted,
    // and run inside the iteration it executes for the whole wavefront nearly every time.  Instead an accepted
    // triangle becomes the lane's UNVERIFIED best (uLeaf >= 0; best* already hold it) and the box tests of all such
    // lanes run together, once at the end of a leaf phase.  While a lane holds an unverified best, later triangles
    // are filtered against it: if it verifies, that is what the reference's strict '<' would have done; if it does
    // not (the conservative slot test admitted a leaf the reference never opens: a ray within 2^-19 of a box face),
    // the filtering may have dropped a triangle that should have won, so the ray is handed to the exact walk
    // (slowList) and traced again from scratch -- rare, and the result is the reference's either way.  Any-hit rays
    // stop at their first accepted triangle and wait for the verdict: pass = occluded, fail = carry on.
    int uLeaf = -1;
    unsigned nv = 0, nt = 0;
    unsigned chunkBase = 0, chunkLeft = 0, chunkShard = 0;
    // Rays are handed out in chunks, one atomic on the launch's one cursor per chunk.  Atomics on one address
    // complete about one per 12 ns on this chip: at 64 rays per atomic that alone caps a launch at 5.5 G rays/s
    // (measured with rays that miss the root box: 3.0 ms per 16.8 M rays, whatever the kernel did per ray), which
    // is where both traversal kernels stood.  Guided self-scheduling: a wavefront asks for half of its share of
    // what was left when it last looked (64..1024 rays), so chunks are large while the queue is long and shrink
    // to 64 at the end, where they decide how evenly the wavefronts finish.
    // The FIRST chunk of every wavefront is static: wavefront k owns rays [k c0, (k + 1) c0) with c0 = its share of
    // three quarters of the queue (PT_TUNE_WIDE_STATIC_SHIFT; 64..1024), and the cursor hands out what lies behind nWaves c0.  A launch starts
    // with all 4,096 wavefronts asking at once, and one that fits the static part (262,144 rays at c0 = 64: config 1)
    // ended with 4,096 more atomics that only said "empty": 2 x 4,096 x 12 ns = the 100 us such a launch took for 8 us
    // of instructions (SQ_INSTS_VALU / SQ_WAVE_CYCLES of the launch).  A 1 M-ray launch spent 16,384 atomics (all
    // chunks at the 64-ray floor) = 0.2 ms of its 0.34 ms the same way.
    const unsigned nWaves = gridDim.x * (unsigned)(kBlk / 64);
    const unsigned gw = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (unsigned)(kBlk / 64) + (threadIdx.x >> 6)));
    (void)gw;
    const unsigned c0 = wide_static_chunk(total, nWaves);
    const unsigned long long staticAll = (unsigned long long)nWaves * c0;
    const unsigned dynBase = staticAll < (unsigned long long)total ? (unsigned)staticAll : total;  // cursor value 0 = this ray
    // The block's pool (PT_WIDE_POOL, round 3).  With a chunk per WAVEFRONT the end of a launch is ragged: the wavefront
    // that took the last 128-ray chunk works on it for two ray-times after the first wavefront has found the queue empty
    // (measured on a 4 M-ray launch, tools/tail_diag.py: queue-empty moments spread over 75 us, ends over 73 us, the median
    // wavefront done 48 us before the last), and smaller chunks run into the cursor's 12 ns per atomic.  Now the cursor
    // feeds a pool per BLOCK in LDS (one word: next index << 32 | end), a wavefront takes exactly as many rays as it has
    // empty lanes with one LDS atomic, and the ONE wavefront that finds the pool empty and wins the block's lock asks the
    // cursor for the next chunk (guided: half the block's share of what is left, PT_TUNE_WIDE_POOL_MIN .. _MAX) while the
    // others walk on: a sixteenth of the global atomics, chunks down to 16 rays per wavefront at the end.  The lock's
    // holder never waits for anything but its own atomic, so every wavefront reaches the end of the queue.
    unsigned long long* const pool = lds.pool;
    unsigned* const poolCtl = lds.poolCtl;
#if !PT_WIDE_POOL
    {
        const unsigned long long first = (unsigned long long)gw * c0;
        chunkBase = first < (unsigned long long)total ? (unsigned)first : total;
        chunkLeft = (total - chunkBase) < c0 ? (total - chunkBase) : c0;
        if (q.counts && chunkLeft != 0) chunkShard = (unsigned)__popcll(__ballot(prefix[lane + 1] <= chunkBase));
    }
#endif
    unsigned lastBase = dynBase;
    bool exhausted = PT_WIDE_POOL ? (total == 0u) : (chunkLeft == 0 && dynBase >= total);
#if PT_DIAG_TAIL
    const unsigned long long diagStart = wall_clock64();
    unsigned long long diagEmpty = exhausted ? diagStart : 0ull;
    unsigned diagGifts = 0;
#endif

// an entry of the hand-back list: written through to the memory side, where the other XCDs' wavefronts will look for it
#define PT_LIST_PUT(I, SLOT) __hip_atomic_store(&slowList[I], (int)(SLOT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define PT_FINISH()                                                                        \
    do {                                                                                   \
        if (ANYHIT) {                                                                      \
            if (bestTri >= 0) occ[target ? (unsigned)target[queueSlot] : queueSlot] = 1;   \
        } else {                                                                           \
            /* the id of the winning triangle sits in its leafTris record; looking it up here, once per ray, */ \
            /* keeps it out of the leaf loop's registers */                               \
            int id = -1;                                                                   \
            if (PT_TUNE_WIDE_CARRY_ID) { if (bestSeq != 0x7fffffff) id = bestTri; }        \
            else if (bestSeq != 0x7fffffff) id = f2i(s.leafTris[3 * (size_t)bestSeq].w);   \
            PT_STREAM_STORE(&hits[raySlot], make_float4(i2f(id), bestT, bestB, bestG));     \
        }                                                                                  \
        PT_RESET_LANE();                                                                   \
    } while (0)
#define PT_RESET_LANE()                                                                    \
    do {                                                                                   \
        hasRay = false;                                                                    \
        w = -1;                                                                            \
        tos = 0;                                                                           \
        sp = 0;                                                                            \
        bot = 0;                                                                           \
        role = -1;                                                                         \
        groups = 0;                                                                        \
        cur = -1;                                                                          \
        uLeaf = -1;                                                                        \
    } while (0)
// the lane's walk is over: nothing to visit, nothing parked, nothing unverified
#define PT_WALK_OVER() (hasRay && w < 0 && groups == 0 && cur < 0 && uLeaf < 0)

    while (true) {
        const unsigned long long mEmpty = __ballot(!hasRay);
        const int nEmpty = __popcll(mEmpty);
        if (nEmpty == 64 && exhausted) break;
        bool fresh = false, adopt = false;  // this lane takes a ray from the queue / a piece of a neighbour's ray
        unsigned adoptEntry = 0u;
        int adoptMeta = 0;
        if (!exhausted && nEmpty >= (ANYHIT ? PT_TUNE_WIDE_REFILL_ANYHIT : PT_TUNE_WIDE_REFILL)) {
            // ---- refill: empty lanes take the next rays of the queue ----
#if PT_WIDE_POOL
            {
                unsigned long long old = 0ull;
                if (lane == 0) old = __hip_atomic_fetch_add(pool, (unsigned long long)(unsigned)nEmpty << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned cur = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(old >> 32));
                const unsigned end = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)old);
                chunkLeft = 0u;
                if (cur < end) {  // (the index half may have run past the end while the pool stood empty: that is "empty" too)
                    chunkBase = cur;
                    chunkLeft = (end - cur) < (unsigned)nEmpty ? (end - cur) : (unsigned)nEmpty;
                    lastBase = cur > dynBase ? cur : dynBase;  // how far the CURSOR has come, as far as this wavefront knows
                    if (q.counts) chunkShard = (unsigned)__popcll(__ballot(prefix[lane + 1] <= cur));
                } else {
                    int state = 0;  // 1: the launch's queue is empty, 2: this wavefront refills the pool
                    if (lane == 0) {
                        if (__hip_atomic_load(&poolCtl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) state = 1;
                        else if (atomicCAS(&poolCtl[0], 0u, 1u) == 0u) state = 2;
                    }
                    state = __builtin_amdgcn_readfirstlane(state);
                    if (state == 1) {
                        exhausted = true;
#if PT_DIAG_TAIL
                        diagEmpty = wall_clock64();
#endif
                    } else if (state == 2) {
                        if (lane == 0) {
                            const unsigned long long now = __hip_atomic_load(pool, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            // (another wavefront may have refilled it between this one's look and its lock)
                            if ((unsigned)(now >> 32) >= (unsigned)now && __hip_atomic_load(&poolCtl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
                                unsigned chunkRays = (((total - lastBase) / gridDim.x) >> PT_TUNE_WIDE_POOL_SHIFT) & ~63u;
                                chunkRays = chunkRays < PT_TUNE_WIDE_POOL_MIN ? PT_TUNE_WIDE_POOL_MIN : (chunkRays > PT_TUNE_WIDE_POOL_MAX ? PT_TUNE_WIDE_POOL_MAX : chunkRays);
                                const unsigned b = atomicAdd(cursor, chunkRays);
                                if (b >= total - dynBase) {
                                    __hip_atomic_store(&poolCtl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                } else {
                                    const unsigned nb = dynBase + b, len = (total - nb) < chunkRays ? (total - nb) : chunkRays;
                                    __hip_atomic_store(pool, ((unsigned long long)nb << 32) | (unsigned long long)(nb + len), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
                            }
                            __hip_atomic_store(&poolCtl[0], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            }
#else
            if (chunkLeft == 0) {
                unsigned base = total;
                unsigned chunkRays = ((total - lastBase) / (nWaves * 2u)) & ~63u;
                chunkRays = chunkRays < PT_TUNE_WIDE_CHUNK_MIN ? PT_TUNE_WIDE_CHUNK_MIN : (chunkRays > 1024u ? 1024u : chunkRays);
                if (dynBase < total) {  // otherwise the static chunks were the whole queue: nothing to ask for
                    if (lane == 0) base = atomicAdd(cursor, chunkRays);
                    base = __shfl(base, 0);
                    base = base < total - dynBase ? dynBase + base : total;
                }
                if (base >= total) {
                    exhausted = true;
#if PT_DIAG_TAIL
                    diagEmpty = wall_clock64();
#endif
                } else {
                    chunkBase = base;
                    lastBase = base;
                    chunkLeft = (total - base) < chunkRays ? (total - base) : chunkRays;
                    if (q.counts) chunkShard = (unsigned)__popcll(__ballot(prefix[lane + 1] <= base));
                }
            }
#endif
            if (chunkLeft != 0) {
                const unsigned rank = (unsigned)__popcll(mEmpty & ((1ull << lane) - 1ull));
                const unsigned take = chunkLeft < (unsigned)nEmpty ? chunkLeft : (unsigned)nEmpty;
                if (!hasRay && rank < take) {
                    const unsigned j = chunkBase + rank;  // dense ray index
                    unsigned slot = j;
                    if (q.counts) {  // dense index -> shard: j < total = prefix[kShards] ends the loop
                        unsigned sh = chunkShard;
                        while (j >= prefix[sh + 1]) ++sh;
                        slot = sh * q.regionSize + (j - prefix[sh]);
                    }
                    queueSlot = slot;
                    raySlot = index ? (unsigned)PT_STREAM_LOAD(&index[slot]) : slot;
                    fresh = true;
                }
                chunkBase += take;
                chunkLeft -= take;
            }
        } else if (SHARE && exhausted && nEmpty >= PT_TUNE_WIDE_SHARE_MIN && nEmpty < 64) {
            // ---- the tail: idle lanes adopt the bottom stack entries of busy ones ("Sharing the tail" above) ----
            // what a busy lane can spare: the bottom entry of its stack (the pending children of the shallowest node it
            // still owes a visit: its largest piece), else the pending siblings of the node it is about to visit, else
            // its oldest parked leaf group when something else keeps it busy
            int giveKind = 0;  // 1: interior entry, 2: leaf group
            if (hasRay && !(ANYHIT && uLeaf >= 0)) {
                if (sp > bot || (tos != 0u && w >= 0)) giveKind = 1;
                else if (groups >= 2 || (groups == 1 && (w >= 0 || cur >= 0))) giveKind = 2;
            }
            const unsigned long long mGive = __ballot(giveKind != 0);
            if (mGive != 0ull) {
                const unsigned long long below = (1ull << lane) - 1ull;
                const int nGive = __popcll(mGive);
                const int nPairs = nGive < nEmpty ? nGive : nEmpty;
                const int rankG = __popcll(mGive & below), rankE = __popcll(mEmpty & below);
                unsigned gEntry = 0u;
                int gMeta = 0;  // (owner lane + 1) | kind << 8
                if (giveKind != 0 && rankG < nPairs) {
                    if (giveKind == 1) {
                        if (sp > bot) { gEntry = stack[bot]; bot += kBlk; }
                        else { gEntry = tos; tos = 0u; }
                    } else {
                        gEntry = groups >= 4 ? g[kG - 1] : (groups == 3 ? g[kG > 2 ? 2 : kG - 1] : (groups == 2 ? g[kG > 1 ? 1 : 0] : g[0]));
                        --groups;
                    }
                    pairLds[waveBase + (unsigned)rankG] = (unsigned char)lane;
                    int root = -1;
                    if (!ANYHIT) {
                        if (role == -1) {  // first piece this lane gives away: its merge slot starts empty
                            role = -2;
                            shareKey[threadIdx.x] = ~0ull;
                            shareOut[threadIdx.x] = 0u;
                        }
                        root = role >= 0 ? role : lane;  // a helper gives on behalf of the lane it helps
                        __hip_atomic_fetch_add(&shareOut[waveBase + (unsigned)root], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                    gMeta = (root + 1) | (giveKind << 8);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#if PT_DIAG_TAIL
                diagGifts += (unsigned)nPairs;
#endif
                const bool takes = !hasRay && rankE < nPairs;
                int src = lane;
                if (takes) src = (int)pairLds[waveBase + (unsigned)rankE];
                const unsigned gotEntry = (unsigned)__shfl((int)gEntry, src);
                const unsigned gotQueue = (unsigned)__shfl((int)queueSlot, src), gotRay = (unsigned)__shfl((int)raySlot, src);
                const int gotMeta = __shfl(gMeta, src);
                if (takes) {
                    adoptEntry = gotEntry;
                    queueSlot = gotQueue;
                    raySlot = gotRay;
                    adoptMeta = gotMeta;
                    adopt = true;
                }
            }
        }
        if (__ballot(fresh || adopt) != 0ull) {
            {
                bool slow = false;
                if (fresh || adopt) {
                    const size_t rs = ANYHIT ? (size_t)queueSlot : (size_t)raySlot;  // shadow rays live in queue order
                    const float4 r0 = PT_STREAM_LOAD(&rays[2 * rs]), r1 = PT_STREAM_LOAD(&rays[2 * rs + 1]);
                    ox = r0.x; oy = r0.y; oz = r0.z; tMin = r0.w;
                    dx = r1.x; dy = r1.y; dz = r1.z; tMax = r1.w;
                    // (an adopted ray qualified when its owner took it from the queue)
#if PT_TUNE_WIDE_PRECISE_HANDBACK
                    static_assert(PT_TUNE_WIDE_PRECISE_EXP >= kDirLo, "the hand-back bound lies inside fast_div_ok's");
                    const bool qualifies = fast_div_ok_flat(ox, oy, oz, dx, dy, dz, PT_TUNE_WIDE_PRECISE_EXP);
#else
                    const bool qualifies = fast_div_ok(vec3{ox, oy, oz}, vec3{dx, dy, dz});
#endif
                    if (adopt | (!s.handBackAll & qualifies)) {
                        yx = 1.0f / dx; yy = 1.0f / dy; yz = 1.0f / dz;
#if PT_TUNE_WIDE_FMA
                        ynx = yx * 0x1.ffffcp-1f; yny = yy * 0x1.ffffcp-1f; ynz = yz * 0x1.ffffcp-1f;
                        yfx = yx * 0x1.00002p0f; yfy = yy * 0x1.00002p0f; yfz = yz * 0x1.00002p0f;
                        {
                            const float ax = ox * ynx, ay = oy * yny, az = oz * ynz, bx = ox * yfx, by = oy * yfy, bz = oz * yfz;
                            cnx = __builtin_fmaf(__builtin_fabsf(ax), 0x1p-22f, ax); cny = __builtin_fmaf(__builtin_fabsf(ay), 0x1p-22f, ay);
                            cnz = __builtin_fmaf(__builtin_fabsf(az), 0x1p-22f, az);
                            cfx = __builtin_fmaf(__builtin_fabsf(bx), -0x1p-22f, bx); cfy = __builtin_fmaf(__builtin_fabsf(by), -0x1p-22f, by);
                            cfz = __builtin_fmaf(__builtin_fabsf(bz), -0x1p-22f, bz);
                        }
#if PT_TUNE_WIDE_PRECISE_HANDBACK
                        precise = false;
#else
                        precise = !(mag_within(dx, PT_TUNE_WIDE_PRECISE_EXP, kDirHi) && mag_within(dy, PT_TUNE_WIDE_PRECISE_EXP, kDirHi) && mag_within(dz, PT_TUNE_WIDE_PRECISE_EXP, kDirHi));
#endif
#if PT_WIDE_HALF
                        // the stored planes are p S with S a power of two (DevScene::wideInvScale = 1 / S): p S (y' / S) is the
                        // same real product, and y' / S is exact, so the fma form sees no difference; the constants c were
                        // formed from the unscaled y'.  `precise` lanes keep the unscaled y' and unscale the planes instead.
                        if (!precise) {
                            ynx *= s.wideInvScale; yny *= s.wideInvScale; ynz *= s.wideInvScale;
                            yfx *= s.wideInvScale; yfy *= s.wideInvScale; yfz *= s.wideInvScale;
                        }
#endif
#else
                        ynx = (float2v)(yx * 0x1.ffffcp-1f); yny = (float2v)(yy * 0x1.ffffcp-1f); ynz = (float2v)(yz * 0x1.ffffcp-1f);
                        yfx = (float2v)(yx * 0x1.00002p0f); yfy = (float2v)(yy * 0x1.00002p0f); yfz = (float2v)(yz * 0x1.00002p0f);
#endif
                        sx = dx < 0.0f ? 1 : 0; sy = dy < 0.0f ? 1 : 0; sz = dz < 0.0f ? 1 : 0;
#if PT_WIDE_K == 8
                        oNx = (unsigned)sx << 4; oFx = 16u - oNx; oNy = 32u + ((unsigned)sy << 4); oFy = 80u - oNy;
                        oNz = 64u + ((unsigned)sz << 4); oFz = 144u - oNz;
#endif
                        bestTri = -1; bestSeq = 0x7fffffff; bestT = 3.402823466e+38f; bestB = 0.0f; bestG = 0.0f;
                        w = 0;
                        sp = 0; bot = 0; role = -1; tos = 0u;
#if PT_WIDE_K == 8
                        if (SHARE && adopt) {
                            if ((adoptMeta >> 8) == 1) {  // a helper starts at the first pending child of the entry it was given
                                tos = adoptEntry;
                                w = group_take8(tos);
                                if ((tos & 255u) == 0u) tos = 0u;
                            } else {  // ... or with the leaf group it was given
                                w = -1;
                                g[0] = adoptEntry;
                                groups = 1;
                            }
                            if (!ANYHIT) role = (adoptMeta & 255) - 1;
                        }
#endif
                        hasRay = true;
                    } else if (ox != ox || oy != oy || oz != oz || dx != dx || dy != dy || dz != dz) {
                        // A NaN anywhere in the ray: every determinant of raytraceTriangle that decides acceptance
                        // contains every component of d or of p0 - o, so t = C / D is NaN and `t > tMin` fails for
                        // every triangle (cs488.h:782-812): the reference walks the whole tree (a NaN quotient
                        // passes AABB::intersect) and returns a miss.  Same answer, without the walk.
                        if (!ANYHIT) hits[raySlot] = make_float4(i2f(-1), 3.402823466e+38f, 0.0f, 0.0f);
                    } else {
                        slow = true;  // this ray takes the literal walk in a later launch
                    }
                }
                const unsigned long long mSlow = __ballot(slow);
                if (mSlow != 0ull) {
                    unsigned at = 0;
                    if (lane == 0) at = atomicAdd(slowCount, (unsigned)__popcll(mSlow));
                    at = __shfl(at, 0);
                    if (slow) PT_LIST_PUT(at + (unsigned)__popcll(mSlow & ((1ull << lane) - 1ull)), queueSlot);
                }
            }
        }
        if (nEmpty == 64) continue;  // nothing else to do yet (the refill may have found rays: look again)
        // ---- visits: lanes with a node to visit and room for one more leaf group ----
        // (the tail: shorter turns, work spreads sooner)
        const int nBurst = (PT_TUNE_WIDE_TAIL_BURST > 0 && SHARE && exhausted) ? PT_TUNE_WIDE_TAIL_BURST
                                                                               : (ANYHIT ? PT_TUNE_WIDE_BURST_ANYHIT : PT_TUNE_WIDE_BURST);
#if PT_TUNE_WIDE_BURST_UNROLL
        PT_PRAGMA(unroll PT_TUNE_WIDE_BURST_UNROLL)
#else
#pragma unroll
#endif
        for (int b = 0; b < nBurst; b++) {
#if PT_DIAG_UTIL == 1
            if (COUNT) { const unsigned long long ma = __ballot(w >= 0 && groups < kG); if (ma) { nv += (w >= 0 && groups < kG) ? 1u : 0u; nt += lane == 0 ? 64u : 0u; } }
#endif
            if (w >= 0 && groups < kG) {
#if PT_WIDE_K == 8
                // seven 16-B pieces per node from either home: near and far half-rows of x, y, z (picked by address), meta
                half8v nx8, fx8, ny8, fy8, nz8, fz8;
                float4v meta;
#define PT_PIECE(AS, T, BASE, OFS) (*(const __attribute__((address_space(AS))) T*)((const __attribute__((address_space(AS))) char*)(BASE) + (OFS)))
#if PT_TUNE_WIDE_LDS128
                // both homes keep 128-B records: one set of piece addresses
                const unsigned a = (unsigned)w << 7;
                const unsigned aNx = a + oNx, aFx = a + oFx, aNy = a + oNy, aFy = a + oFy, aNz = a + oNz, aFz = a + oFz;
                if (w < nTop) {  // top of the tree: from LDS
                    nx8 = PT_PIECE(3, half8v, topLds, aNx); fx8 = PT_PIECE(3, half8v, topLds, aFx);
                    ny8 = PT_PIECE(3, half8v, topLds, aNy); fy8 = PT_PIECE(3, half8v, topLds, aFy);
                    nz8 = PT_PIECE(3, half8v, topLds, aNz); fz8 = PT_PIECE(3, half8v, topLds, aFz);
                    meta = PT_PIECE(3, float4v, topLds, a + 96u);
                } else {
                    nx8 = PT_PIECE(1, half8v, nodesGlobal, (size_t)aNx); fx8 = PT_PIECE(1, half8v, nodesGlobal, (size_t)aFx);
                    ny8 = PT_PIECE(1, half8v, nodesGlobal, (size_t)aNy); fy8 = PT_PIECE(1, half8v, nodesGlobal, (size_t)aFy);
                    nz8 = PT_PIECE(1, half8v, nodesGlobal, (size_t)aNz); fz8 = PT_PIECE(1, half8v, nodesGlobal, (size_t)aFz);
                    meta = PT_PIECE(1, float4v, nodesGlobal, (size_t)(a + 96u));
                }
#else
                if (w < nTop) {  // top of the tree: from LDS, 112-B records
                    const unsigned a = __umul24((unsigned)w, 112u);
                    nx8 = PT_PIECE(3, half8v, topLds, a + oNx); fx8 = PT_PIECE(3, half8v, topLds, a + oFx);
                    ny8 = PT_PIECE(3, half8v, topLds, a + oNy); fy8 = PT_PIECE(3, half8v, topLds, a + oFy);
                    nz8 = PT_PIECE(3, half8v, topLds, a + oNz); fz8 = PT_PIECE(3, half8v, topLds, a + oFz);
                    meta = PT_PIECE(3, float4v, topLds, a + 96u);
                } else {  // 128-B records
                    const unsigned a = (unsigned)w << 7;
                    nx8 = PT_PIECE(1, half8v, nodesGlobal, (size_t)(a + oNx)); fx8 = PT_PIECE(1, half8v, nodesGlobal, (size_t)(a + oFx));
                    ny8 = PT_PIECE(1, half8v, nodesGlobal, (size_t)(a + oNy)); fy8 = PT_PIECE(1, half8v, nodesGlobal, (size_t)(a + oFy));
                    nz8 = PT_PIECE(1, half8v, nodesGlobal, (size_t)(a + oNz)); fz8 = PT_PIECE(1, half8v, nodesGlobal, (size_t)(a + oFz));
                    meta = PT_PIECE(1, float4v, nodesGlobal, (size_t)(a + 96u));
                }
#endif
#undef PT_PIECE
                const int counts = f2i(meta.z);
                const int nInt = counts & 255;  // interior slots come first, then the leaf slots; the EMPTY slots behind them hold
                                                // the box no ray hits (+inf / -inf planes, build_wide), so no mask of the used slots
#if !PT_DIAG_UTIL
                if (COUNT) nv += (unsigned)(nInt + (counts >> 8));
#endif
                int hm;
                if (!PT_TUNE_WIDE_PRECISE_HANDBACK && precise)
                    hm = wide_hit_mask_sub(widen(nx8.lo, s.wideInvScale), widen(fx8.lo, s.wideInvScale), widen(ny8.lo, s.wideInvScale), widen(fy8.lo, s.wideInvScale), widen(nz8.lo, s.wideInvScale), widen(fz8.lo, s.wideInvScale),
                                           ox, oy, oz, ynx, yny, ynz, yfx, yfy, yfz) |
                         (wide_hit_mask_sub(widen(nx8.hi, s.wideInvScale), widen(fx8.hi, s.wideInvScale), widen(ny8.hi, s.wideInvScale), widen(fy8.hi, s.wideInvScale), widen(nz8.hi, s.wideInvScale), widen(fz8.hi, s.wideInvScale),
                                            ox, oy, oz, ynx, yny, ynz, yfx, yfy, yfz) << 4);
                else
                    hm = wide_hit_mask_fma_h(nx8.lo, fx8.lo, ny8.lo, fy8.lo, nz8.lo, fz8.lo, ynx, yny, ynz, yfx, yfy, yfz, cnx, cny, cnz,
                                             cfx, cfy, cfz) |
                         (wide_hit_mask_fma_h(nx8.hi, fx8.hi, ny8.hi, fy8.hi, nz8.hi, fz8.hi, ynx, yny, ynz, yfx, yfy, yfz, cnx, cny,
                                              cnz, cfx, cfy, cfz) << 4);
                const int lh = hm >> nInt, ih = hm & ((1 << nInt) - 1);  // lh: bit k = leaf slot k of this node
                if (lh != 0) {  // park the candidate leaves of this node as one group
#pragma unroll
                    for (int k = kG - 1; k > 0; k--) g[k] = g[k - 1];
                    g[0] = ((unsigned)f2i(meta.y) << 8) | (unsigned)lh;
                    ++groups;
                }
#if PT_TUNE_WIDE_FLAT_STACK
                {   // the steps below with selects in place of the nested branches (one branch each is left, for the LDS write and
                    // the LDS read)
                    const bool push = (ih != 0) & (tos != 0u);
                    if (push) stack[sp] = tos;
                    sp += push ? kBlk : 0;
                    tos = ih != 0 ? (((unsigned)f2i(meta.x) << 8) | (unsigned)ih) : tos;
                    const unsigned e = tos, rest = e & (e - 1u);
                    w = e != 0u ? (int)(e >> 8) + __builtin_ctz(e | 0x80000000u) : -1;
                    const bool emptied = (rest & 255u) == 0u;  // (also when e is 0)
                    const bool pop = (e != 0u) & emptied & (sp > bot);
                    sp -= pop ? kBlk : 0;
                    unsigned under = 0u;
                    if (pop) under = stack[sp];
                    tos = emptied ? under : rest;
                }
#else
                if (ih != 0) {  // the hit children go on top of the stack as one entry
                    if (tos != 0u) { stack[sp] = tos; sp += kBlk; }
                    tos = ((unsigned)f2i(meta.x) << 8) | (unsigned)ih;
                }
                w = -1;
                if (tos != 0u) {  // next: the lowest pending child of the top entry
                    w = group_take8(tos);
                    if ((tos & 255u) == 0u) {
                        tos = 0u;
                        if (sp > bot) { sp -= kBlk; tos = stack[sp]; }
                    }
                }
#endif
            }
        }
#else
#if PT_WIDE_HALF
                // four 16-B rows per node from either home: x, y, z planes as halves {lo0..3, hi0..3}, meta
                half8v rx, ry, rz;
                float4v meta;
                if (w < nTop) {  // top of the tree: from LDS
                    const __attribute__((address_space(3))) char* at = (const __attribute__((address_space(3))) char*)topLds + ((unsigned)w << 6);
                    rx = *(const __attribute__((address_space(3))) half8v*)at;
                    ry = *(const __attribute__((address_space(3))) half8v*)(at + 16);
                    rz = *(const __attribute__((address_space(3))) half8v*)(at + 32);
                    meta = *(const __attribute__((address_space(3))) float4v*)(at + 48);
                } else {
                    const __attribute__((address_space(1))) char* at = (const __attribute__((address_space(1))) char*)nodesGlobal + ((size_t)(unsigned)w << 6);
                    rx = *(const __attribute__((address_space(1))) half8v*)at;
                    ry = *(const __attribute__((address_space(1))) half8v*)(at + 16);
                    rz = *(const __attribute__((address_space(1))) half8v*)(at + 32);
                    meta = *(const __attribute__((address_space(1))) float4v*)(at + 48);
                }
                const int masks = f2i(meta.z);
                const int imask = masks & 15, lmask = (masks >> 4) & 15;
#if !PT_DIAG_UTIL
                if (COUNT) nv += (unsigned)__popc((unsigned)(imask | lmask));
#endif
                // near = the plane the ray meets first along the axis: lo where d > 0, hi where d < 0
                const half4v nearx = sx ? rx.hi : rx.lo, farx = sx ? rx.lo : rx.hi;
                const half4v neary = sy ? ry.hi : ry.lo, fary = sy ? ry.lo : ry.hi;
                const half4v nearz = sz ? rz.hi : rz.lo, farz = sz ? rz.lo : rz.hi;
                const int hm = precise ? wide_hit_mask_sub(widen(nearx, s.wideInvScale), widen(farx, s.wideInvScale), widen(neary, s.wideInvScale), widen(fary, s.wideInvScale), widen(nearz, s.wideInvScale), widen(farz, s.wideInvScale),
                                                           ox, oy, oz, ynx, yny, ynz, yfx, yfy, yfz)
                                       : wide_hit_mask_fma_h(nearx, farx, neary, fary, nearz, farz, ynx, yny, ynz, yfx, yfy, yfz, cnx, cny,
                                                             cnz, cfx, cfy, cfz);
#else
                float4v nearx, farx, neary, fary, nearz, farz, meta;  // rows picked by the direction signs (sx, sy, sz)
                // byte offsets of the seven rows inside the record, the same for both homes of a node; 32-bit
                // offsets from a uniform base let the loads address with base + offset instead of 64-bit sums
                const unsigned rxn = (unsigned)sx << 4, rxf = 16u - rxn, ryn = 32u + ((unsigned)sy << 4), ryf = 80u - ryn,
                               rzn = 64u + ((unsigned)sz << 4), rzf = 144u - rzn;
#define PT_ROW(AS, BASE, OFS) (*(const __attribute__((address_space(AS))) float4v*)((const __attribute__((address_space(AS))) char*)(BASE) + (OFS)))
                if (w < nTop) {  // top of the tree: from LDS
                    const unsigned a = __umul24((unsigned)w, 16u * kWideNodeF4);
                    nearx = PT_ROW(3, topLds, a + rxn); farx = PT_ROW(3, topLds, a + rxf); neary = PT_ROW(3, topLds, a + ryn);
                    fary = PT_ROW(3, topLds, a + ryf); nearz = PT_ROW(3, topLds, a + rzn); farz = PT_ROW(3, topLds, a + rzf);
                    meta = PT_ROW(3, topLds, a + 96u);
                } else {
                    const unsigned a = __umul24((unsigned)w, 16u * kWideNodeStrideF4);  // < 2^22 nodes x 128 B
                    nearx = PT_ROW(1, nodesGlobal, (size_t)(a + rxn)); farx = PT_ROW(1, nodesGlobal, (size_t)(a + rxf));
                    neary = PT_ROW(1, nodesGlobal, (size_t)(a + ryn)); fary = PT_ROW(1, nodesGlobal, (size_t)(a + ryf));
                    nearz = PT_ROW(1, nodesGlobal, (size_t)(a + rzn)); farz = PT_ROW(1, nodesGlobal, (size_t)(a + rzf));
                    meta = PT_ROW(1, nodesGlobal, (size_t)(a + 96u));
                }
#undef PT_ROW
                const int masks = f2i(meta.z);
                const int imask = masks & 15, lmask = (masks >> 4) & 15;
#if !PT_DIAG_UTIL
                if (COUNT) nv += (unsigned)__popc((unsigned)(imask | lmask));
#endif
#if PT_TUNE_WIDE_FMA
                const int hm = precise ? wide_hit_mask_sub(nearx, farx, neary, fary, nearz, farz, ox, oy, oz, ynx, yny, ynz, yfx, yfy, yfz)
                                       : wide_hit_mask_fma(nearx, farx, neary, fary, nearz, farz, ynx, yny, ynz, yfx, yfy, yfz, cnx, cny, cnz,
                                                           cfx, cfy, cfz);
#else
                const int hm = wide_hit_mask(nearx, farx, neary, fary, nearz, farz, ox, oy, oz, ynx, yny, ynz, yfx, yfy, yfz);
#endif
#endif
                const int lh = hm & lmask, ih = hm & imask;
                if (lh != 0) {  // park the candidate leaves of this node as one group
#pragma unroll
                    for (int k = kG - 1; k > 0; k--) g[k] = g[k - 1];
                    g[0] = pack_group(f2i(meta.y), lmask, lh);
                    ++groups;
                }
                if (ih != 0) {  // the hit children go on top of the stack as one entry
                    if (tos != 0u) { stack[sp] = tos; sp += kBlk; }
                    tos = pack_group(f2i(meta.x), imask, ih);
                }
                w = -1;
                if (tos != 0u) {  // next: the lowest pending child of the top entry
                    w = group_take(tos);
                    if ((tos & 15u) == 0u) {
                        tos = 0u;
                        if (sp > bot) { sp -= kBlk; tos = stack[sp]; }
                    }
                }
            }
        }
#endif  // PT_WIDE_K
        if (PT_WALK_OVER() && (!SHARE || ANYHIT || role == -1)) PT_FINISH();  // walked everything, nothing parked, nothing unverified (lanes that share a ray: at the end of the iteration)
        // ---- leaves: open parked leaves with the exact box test, test their triangles ----
        const unsigned long long mCanVisit = __ballot(w >= 0 && groups < kG);
        unsigned long long mPend = __ballot(cur >= 0 || groups > 0);
        const unsigned long long mStuck = mPend & ~mCanVisit;
#if PT_TUNE_WIDE_GREEDY
        // experiment: whichever phase has more lanes to serve runs next
        if (mPend != 0ull && (__popcll(mPend) >= __popcll(mCanVisit) ||
                              __popcll(mStuck) >= (ANYHIT ? PT_TUNE_WIDE_STUCK_ANYHIT : PT_TUNE_WIDE_STUCK))) {
#else
        const bool tailGreedy = PT_TUNE_WIDE_TAIL_GREEDY && SHARE && exhausted;
        if (mPend != 0ull && (__popcll(mPend) >= (ANYHIT ? PT_TUNE_WIDE_LEAF_ANYHIT : PT_TUNE_WIDE_LEAF) ||
                              __popcll(mStuck) >= (ANYHIT ? PT_TUNE_WIDE_STUCK_ANYHIT : PT_TUNE_WIDE_STUCK) || mCanVisit == 0ull ||
                              (tailGreedy && __popcll(mPend) >= __popcll(mCanVisit)))) {
#endif
            do {
                // One round trip per iteration: a lane either opens the next leaf of its top group (the copy of the
                // leaf's first triangle in its record) or goes on with the next triangle of its open leaf.  A
                // candidate leaf counts only if its exact box passes the reference's predicate; "box passes AND
                // triangle accepted" is evaluated triangle first, so the box test (and the load of the box) is paid
                // only by the few lanes whose triangle would change the result.
                const bool waiting = ANYHIT && uLeaf >= 0;  // an any-hit ray with an accepted triangle waits for its verdict
                const bool opening = !waiting && cur < 0 && groups > 0;
#if PT_DIAG_UTIL == 2
                if (COUNT) { nv += ((cur >= 0 && !waiting) || opening) ? 1u : 0u; nt += lane == 0 ? 64u : 0u; }
#endif
                if ((cur >= 0 && !waiting) || opening) {
                    // leafHead and leafTris lie in one arena less than 4 GB apart (checked when the context is made):
                    // one uniform base, 32-bit offsets
                    unsigned triOfs;
                    if (opening) {
                        curLeaf = PT_GROUP_TAKE(g[0]);
                        triOfs = headOfs + (unsigned)curLeaf * (16u * kLeafHeadStrideF4);
                        if ((g[0] & kPendMask) == 0u) {
#pragma unroll
                            for (int k = 0; k + 1 < kG; k++) g[k] = g[k + 1];
                            --groups;
                        }
                    } else {
                        triOfs = ((unsigned)cur << 5) + ((unsigned)cur << 4);  // 48 B per leafTris record
                    }
                    const __attribute__((address_space(1))) float4v* triAt =
                        (const __attribute__((address_space(1))) float4v*)((const __attribute__((address_space(1))) char*)s.leafTris + (size_t)triOfs);
                    const float4v q0 = triAt[0], q1 = triAt[1], q2 = triAt[2];
                    if (opening) cur = f2i(q2.w);
#if !PT_DIAG_UTIL
                    if (COUNT) nt++;
#endif
                    float t, be, ga;
                    bool acc = tri_test(vec3{q0.x, q0.y, q0.z}, vec3{q1.x, q1.y, q1.z}, vec3{q2.x, q2.y, q2.z},
                                        vec3{ox, oy, oz}, vec3{dx, dy, dz}, tMin, tMax, t, be, ga);
                    // the reference keeps the FIRST of equal t's in depth-first order: smaller leafTris index wins
                    acc = acc & (ANYHIT | (t < bestT) | ((t == bestT) & (cur < bestSeq)));  // (bitwise: compares and scalar ands, no nested branches)
                    if (acc) {  // provisionally the best; its leaf's exact box is tested at the end of this leaf phase
                        // a verified best that an unverified one replaces is lost if the latter fails: that ray is
                        // re-traced (see above), so nothing needs to be kept
                        if (ANYHIT || PT_TUNE_WIDE_CARRY_ID) bestTri = f2i(q0.w);  // any hit: the flag "something accepted"; closest hit: see PT_FINISH
                        bestSeq = cur; bestT = t; bestB = be; bestG = ga;
                        uLeaf = curLeaf;
                    }
                    if (f2i(q1.w)) cur = -1;  // last triangle of this leaf
                    else ++cur;
                }
                mPend = __ballot((cur >= 0 || groups > 0) && !(ANYHIT && uLeaf >= 0));
#if PT_TUNE_WIDE_GREEDY
            } while (mPend != 0ull && __popcll(mPend) >= __popcll(__ballot(w >= 0 && groups < kG)));
#else
            } while (__popcll(mPend) >= (ANYHIT ? PT_TUNE_WIDE_KEEP_ANYHIT : PT_TUNE_WIDE_KEEP) ||
                     (tailGreedy && mPend != 0ull && __popcll(mPend) >= __popcll(__ballot(w >= 0 && groups < kG))));
#endif
            // ---- the verdicts: exact box predicate of every lane that holds an unverified best ----
            if (uLeaf >= 0) {
                const __attribute__((address_space(1))) float4v* rec =
                    (const __attribute__((address_space(1))) float4v*)s.leafBox + 2 * (size_t)uLeaf;
#if !PT_DIAG_UTIL
                if (COUNT) nv++;
#endif
                const bool pass = slab_hit_fast(rec[0], rec[1], ox, oy, oz, dx, dy, dz, yx, yy, yz);  // the reference's predicate
                uLeaf = -1;
                if (ANYHIT) {
                    if (pass) PT_FINISH();  // isHit: any accepted triangle decides
                    else bestTri = -1;      // that triangle does not count; the search goes on
                } else if (!pass) {
                    // the filtering since that candidate may have dropped the true winner: trace this ray again, exactly
                    if (!SHARE || role == -1) {
                        unsigned at = atomicAdd(slowCount, 1u);
                        PT_LIST_PUT(at, queueSlot);
                        PT_RESET_LANE();
                    } else if (role >= 0) {  // a helper: mark the owner's ray, count off, fall idle
                        __hip_atomic_fetch_or(&shareOut[waveBase + (unsigned)role], 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        __hip_atomic_fetch_sub(&shareOut[waveBase + (unsigned)role], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        PT_RESET_LANE();
                    } else {  // the owner of a shared ray: mark it, drop what is left of its own walk, wait for the helpers
                        __hip_atomic_fetch_or(&shareOut[threadIdx.x], 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        w = -1; tos = 0u; sp = bot; groups = 0; cur = -1; bestSeq = 0x7fffffff;
                    }
                }
            }
            if (PT_WALK_OVER() && (!SHARE || ANYHIT || role == -1)) PT_FINISH();
        }
        if (SHARE && !ANYHIT && exhausted && PT_WALK_OVER()) {  // only lanes that share a ray get here with their walk over
            if (role >= 0) {
                // a helper has walked its part: fold its best into the owner's key (its verdicts are in: uLeaf < 0), count off
                if (bestSeq != 0x7fffffff)
                    __hip_atomic_fetch_min(&shareKey[waveBase + (unsigned)role], ((unsigned long long)(unsigned)f2i(bestT) << 32) | (unsigned)bestSeq,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_sub(&shareOut[waveBase + (unsigned)role], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                PT_RESET_LANE();
            } else if (role == -2) {
                const unsigned out = __hip_atomic_load(&shareOut[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if ((out & 0x7fffffffu) == 0u) {  // every helper has reported
                    if (out != 0u) {  // some verdict failed: the exact walk traces this ray again
                        unsigned at = atomicAdd(slowCount, 1u);
                        PT_LIST_PUT(at, queueSlot);
                        PT_RESET_LANE();
                    } else {
                        const unsigned long long theirs = __hip_atomic_load(&shareKey[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        const unsigned long long mine = bestSeq != 0x7fffffff ? (((unsigned long long)(unsigned)f2i(bestT) << 32) | (unsigned)bestSeq) : ~0ull;
                        if (theirs < mine) {
                            // a helper's triangle wins: t and the index are in the key; beta and gamma come from testing that
                            // one triangle again (same operands, same instructions: the bits the helper had)
                            bestSeq = (int)(unsigned)(theirs & 0xffffffffull);
                            const __attribute__((address_space(1))) float4v* triAt =
                                (const __attribute__((address_space(1))) float4v*)s.leafTris + 3 * (size_t)bestSeq;
                            const float4v q0 = triAt[0], q1 = triAt[1], q2 = triAt[2];
                            float t = 0.0f, be = 0.0f, ga = 0.0f;
                            (void)tri_test(vec3{q0.x, q0.y, q0.z}, vec3{q1.x, q1.y, q1.z}, vec3{q2.x, q2.y, q2.z}, vec3{ox, oy, oz},
                                           vec3{dx, dy, dz}, tMin, tMax, t, be, ga);
                            bestT = t; bestB = be; bestG = ga;
                            if (PT_TUNE_WIDE_CARRY_ID) bestTri = f2i(q0.w);
                        }
                        PT_FINISH();
                    }
                }
            }
        }
    }
#undef PT_FINISH
#undef PT_LIST_PUT
#undef PT_RESET_LANE
#undef PT_WALK_OVER
#if PT_WIDE_DRAIN
    // Draining the hand-back list (round 3).  The few rays a launch hands back (slowList: a zero direction component, a
    // failed verdict) used to wait for two more launches (k_trace_slow, list-mode k_trace), 20 to 35 us per traversal
    // on the frame's critical path.  Now every wavefront that leaves the loop above traces its share of the list as it
    // stands, the whole wavefront per ray (slow_walk.hpp) on the LDS its lanes' stacks occupy: entry i belongs to wavefront
    // i mod nWaves -- no claiming, hence no atomics that thousands of wavefronts would fight over.  (Two claiming schemes
    // were tried first.  Scan the list and take entries by exchange: every wavefront that arrives walks over the entries
    // the earlier ones took.  A compare-and-swap cursor: all wavefronts arrive at once, one wins per round, and a list
    // of 1,500 entries took 23 ms.)  The owner puts -1 back, the list's resting state, BEFORE it traces the ray.  An entry
    // may be a moment behind its count -- its producer, a running wavefront that waits for nobody, is between its
    // atomicAdd and its store -- so the owner reads it until it is there (bounded).  A wavefront looks at the count, traces
    // its entries below it, looks again, and leaves when the count has stopped growing under it.  Nobody waits for
    // anything but a store in flight, so blocks that are not resident yet cannot be waited for.  An entry that arrives
    // after its owner has gone (a verdict that failed late in the launch; the rays of a launch so small that most of its
    // wavefronts leave before the first ray is handed back) stays in the list for the launch that follows, list-mode
    // k_trace.  (A version in which the launch's last wavefront traced those itself, one after the other, made a launch
    // of 16 axis-parallel rays take 124 us.)
    // Only SHORT lists are emptied here: up to two entries per wavefront of the launch (the usual few dozen rays of a frame;
    // all of a launch's rays when they are few).  A longer list -- a batch of axis-parallel rays, say, none of which
    // qualifies -- is work for every LANE, not for a wavefront per ray, and wavefronts that leave early would orphan most
    // of it: it is left as it stands (what was taken reads -1) for the launch that follows, list-mode k_trace, which
    // skips what has been taken; the same launch looks after trees too deep for this walk's stack (!slowOk).
    // (Visibility without fences: an agent-scope fence writes the whole L2 of the wavefront's XCD back -- 4,096 of them cost
    // 160 us per launch when tried.  The list is therefore written, read and reset with agent-scope ATOMICS only, which
    // act at the memory side and need no cache maintenance.)
    if (s.drain && s.slowOk) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);  // this wavefront's own entries are out before it looks at the list
        int* const wstack = (int*)stackLds + waveBase;
        const unsigned drainMax = 2u * nWaves;
        // one handed-back ray, by the whole wavefront
        auto traceEntry = [&](const unsigned i) {
            int got = -1;
            if (lane == 0) {
                // (bounded: a store in flight arrives within microseconds; the bound only keeps a wavefront from spinning
                // for ever should an entry never come -- every wavefront must reach its end)
                for (int spin = 0; spin < (1 << 20); spin++) {
                    got = __hip_atomic_load(&slowList[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (got >= 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (got >= 0) __hip_atomic_store(&slowList[i], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            got = __builtin_amdgcn_readfirstlane(got);
            if (got < 0) return;
            const unsigned slot = (unsigned)got;
            const unsigned rsl = index ? (unsigned)index[slot] : slot;
            const size_t rs = ANYHIT ? (size_t)slot : (size_t)rsl;
            const float4 r0 = rays[2 * rs], r1 = rays[2 * rs + 1];
            int bTri, bSeq;
            float bT, bB, bG;
            slow_walk_wave<ANYHIT, COUNT, kBlk>(s, vec3{r0.x, r0.y, r0.z}, vec3{r1.x, r1.y, r1.z}, r0.w, r1.w, wstack, 64 * kWideMaxDepth,
                                                lane, bTri, bSeq, bT, bB, bG, nv, nt);
            if (lane == 0) {
                if (ANYHIT) {
                    if (bTri >= 0) occ[target ? (unsigned)target[slot] : slot] = 1;
                } else {
                    hits[rsl] = make_float4(i2f(bTri), bT, bB, bG);
                }
            }
            wave_lds_sync();
        };
        // this wavefront's share: entries gw, gw + nWaves, ... below the count, until the count stops growing under it
        unsigned seen = 0u;
        for (;;) {
            unsigned n = 0u;
            if (lane == 0) n = __hip_atomic_load(slowCount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            n = (unsigned)__builtin_amdgcn_readfirstlane((int)n);
            if (n == seen || n > drainMax) break;
            unsigned i = seen + (gw + nWaves - seen % nWaves) % nWaves;  // the first index >= seen that is gw mod nWaves
            for (; i < n; i += nWaves) traceEntry(i);
            seen = n;
        }
    }
#endif
#if PT_DIAG_TAIL
    if (lane == 0 && gw < 4096u) {
        unsigned long long* dg = g_tailDiag + (ANYHIT ? 4 * 4096 : 0) + 4 * gw;
        dg[0] = diagStart;
        dg[1] = diagEmpty;
        dg[2] = wall_clock64();
        dg[3] = diagGifts;
    }
#endif
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

template <bool ANYHIT, bool COUNT>
__global__ __launch_bounds__(WideCfg<ANYHIT>::kBlock) PT_WIDE_ATTR void k_trace_wide(
    DevScene s, const float4* __restrict__ rays, const int* __restrict__ index, const int* __restrict__ target,
    QueueView q, unsigned* __restrict__ cursor, float4* __restrict__ hits, unsigned char* __restrict__ occ,
    unsigned long long* __restrict__ counters, unsigned* __restrict__ slowCount, int* __restrict__ slowList,
    unsigned* __restrict__ nextHeader) {
    PT_WIDE_LDS_DECL(WideCfg<ANYHIT>::kBlock, WideCfg<ANYHIT>::kTop);
    // the hand-back list's header for the NEXT launch on this list (two headers used in turn: no memset between launches)
    if (nextHeader && blockIdx.x == 0 && threadIdx.x < 4) nextHeader[threadIdx.x] = 0u;
    const int nTop = s.nWideTop < WideCfg<ANYHIT>::kTop ? s.nWideTop : WideCfg<ANYHIT>::kTop;
    wide_load_top(s, ldsTop, nTop, WideCfg<ANYHIT>::kBlock);
    wide_prefix(q, ldsPrefix[0], 0, &ldsPool[0], ldsPoolCtl[0], WideCfg<ANYHIT>::kBlock / 64);
    __syncthreads();
    wide_body<ANYHIT, COUNT>(s, rays, index, target, q, cursor, hits, occ, counters, slowCount, slowList,
                             WideLds{ldsTop, ldsStack, ldsPrefix[0], ldsPair, ldsKey, ldsOut, &ldsPool[0], ldsPoolCtl[0]});
}

#if PT_WIDE_K == 8
// Both traversals of a bounce in ONE launch (path mode): the extension rays of bounce it + 1 and the shadow rays of
// bounce it depend only on shade(it), and as two launches each ends in its own tail -- a stretch of 0.1 ms in which a
// few wavefronts finish their longest rays while the chip idles (two streams did not help: the blocks of the second
// kernel need whole CUs, which the first gives up only as its own blocks end).  Here a wavefront that has drained the
// first queue goes straight on to the second: same block, same top of the tree in LDS, one tail per bounce instead of
// two.  The bodies are the two instantiations above, unchanged; each has its own queue, cursor and hand-back list.
static_assert(WideCfg<false>::kBlock == WideCfg<true>::kBlock && WideCfg<false>::kTop == WideCfg<true>::kTop &&
                  WideCfg<false>::kWaves == WideCfg<true>::kWaves, "the fused launch needs one block shape");
template <bool COUNT>
__global__ __launch_bounds__(WideCfg<false>::kBlock) __attribute__((amdgpu_waves_per_eu(WideCfg<false>::kWaves, WideCfg<false>::kWaves)))
void k_trace_wide_both(DevScene s, const float4* __restrict__ raysC, const int* __restrict__ indexC, QueueView qC,
                       unsigned* __restrict__ cursorC, float4* __restrict__ hits, unsigned long long* __restrict__ countersC,
                       unsigned* __restrict__ slowCountC, int* __restrict__ slowListC,
                       const float4* __restrict__ raysO, const int* __restrict__ targetO, QueueView qO,
                       unsigned* __restrict__ cursorO, unsigned char* __restrict__ occ, unsigned long long* __restrict__ countersO,
                       unsigned* __restrict__ slowCountO, int* __restrict__ slowListO, int shadowFirst,
                       unsigned* __restrict__ nextHeaderC, unsigned* __restrict__ nextHeaderO) {
    PT_WIDE_LDS_DECL(WideCfg<false>::kBlock, WideCfg<false>::kTop);
    if (blockIdx.x == 0 && threadIdx.x < 4) {
        if (nextHeaderC) nextHeaderC[threadIdx.x] = 0u;
        if (nextHeaderO) nextHeaderO[threadIdx.x] = 0u;
    }
    const int nTop = s.nWideTop < WideCfg<false>::kTop ? s.nWideTop : WideCfg<false>::kTop;
    wide_load_top(s, ldsTop, nTop, WideCfg<false>::kBlock);
    wide_prefix(qC, ldsPrefix[0], 0, &ldsPool[0], ldsPoolCtl[0], WideCfg<false>::kBlock / 64);
    wide_prefix(qO, ldsPrefix[1], 1, &ldsPool[1], ldsPoolCtl[1], WideCfg<false>::kBlock / 64);
    __syncthreads();
    if (shadowFirst)
        wide_body<true, COUNT>(s, raysO, nullptr, targetO, qO, cursorO, nullptr, occ, countersO, slowCountO, slowListO,
                               WideLds{ldsTop, ldsStack, ldsPrefix[1], ldsPair, ldsKey, ldsOut, &ldsPool[1], ldsPoolCtl[1]});
    wide_body<false, COUNT>(s, raysC, indexC, nullptr, qC, cursorC, hits, nullptr, countersC, slowCountC, slowListC,
                            WideLds{ldsTop, ldsStack, ldsPrefix[0], ldsPair, ldsKey, ldsOut, &ldsPool[0], ldsPoolCtl[0]});
    if (!shadowFirst)
        wide_body<true, COUNT>(s, raysO, nullptr, targetO, qO, cursorO, nullptr, occ, countersO, slowCountO, slowListO,
                               WideLds{ldsTop, ldsStack, ldsPrefix[1], ldsPair, ldsKey, ldsOut, &ldsPool[1], ldsPoolCtl[1]});
}
#endif

template <bool ANYHIT>
static inline unsigned wide_grid(size_t nMax) {
    size_t g = (nMax + WideCfg<ANYHIT>::kBlock - 1) / WideCfg<ANYHIT>::kBlock;
    if (g < 1) g = 1;
    const size_t cap = (size_t)256 * WideCfg<ANYHIT>::kPerCU;  // as many blocks as fit the CUs' LDS at once
    if (g > cap) g = cap;
    return (unsigned)g;
}
void launch_trace_wide_closest(hipStream_t st, const DevScene& s, const float4* rays, const int* index, QueueView q,
                               unsigned* cursor, float4* hits, unsigned long long* counters, unsigned* slowCount,
                               int* slowList, unsigned* nextHeader) {
    const unsigned g = wide_grid<false>(q.space);
    if (counters)
        hipLaunchKernelGGL((k_trace_wide<false, true>), dim3(g), dim3(WideCfg<false>::kBlock), 0, st, s, rays, index, nullptr, q, cursor,
                           hits, nullptr, counters, slowCount, slowList, nextHeader);
    else
        hipLaunchKernelGGL((k_trace_wide<false, false>), dim3(g), dim3(WideCfg<false>::kBlock), 0, st, s, rays, index, nullptr, q, cursor,
                           hits, nullptr, counters, slowCount, slowList, nextHeader);
}
void launch_trace_wide_occluded(hipStream_t st, const DevScene& s, const float4* rays, const int* target, QueueView q,
                                unsigned* cursor, unsigned char* occ, unsigned long long* counters, unsigned* slowCount,
                                int* slowList, unsigned* nextHeader) {
    const unsigned g = wide_grid<true>(q.space);
    if (counters)
        hipLaunchKernelGGL((k_trace_wide<true, true>), dim3(g), dim3(WideCfg<true>::kBlock), 0, st, s, rays, nullptr, target, q, cursor,
                           nullptr, occ, counters, slowCount, slowList, nextHeader);
    else
        hipLaunchKernelGGL((k_trace_wide<true, false>), dim3(g), dim3(WideCfg<true>::kBlock), 0, st, s, rays, nullptr, target, q, cursor,
                           nullptr, occ, counters, slowCount, slowList, nextHeader);
}

#if PT_WIDE_K == 8
void launch_trace_wide_both(hipStream_t st, const DevScene& s, const float4* raysC, const int* indexC, QueueView qC, unsigned* cursorC,
                            float4* hits, unsigned long long* countersC, unsigned* slowCountC, int* slowListC, const float4* raysO,
                            const int* targetO, QueueView qO, unsigned* cursorO, unsigned char* occ, unsigned long long* countersO,
                            unsigned* slowCountO, int* slowListO, int shadowFirst, unsigned* nextHeaderC, unsigned* nextHeaderO) {
    const unsigned g = wide_grid<false>((size_t)qC.space + qO.space);
    if (countersC)
        hipLaunchKernelGGL((k_trace_wide_both<true>), dim3(g), dim3(WideCfg<false>::kBlock), 0, st, s, raysC, indexC, qC, cursorC, hits,
                           countersC, slowCountC, slowListC, raysO, targetO, qO, cursorO, occ, countersO, slowCountO, slowListO, shadowFirst, nextHeaderC, nextHeaderO);
    else
        hipLaunchKernelGGL((k_trace_wide_both<false>), dim3(g), dim3(WideCfg<false>::kBlock), 0, st, s, raysC, indexC, qC, cursorC, hits,
                           countersC, slowCountC, slowListC, raysO, targetO, qO, cursorO, occ, countersO, slowCountO, slowListO, shadowFirst, nextHeaderC, nextHeaderO);
}
#endif

}  // namespace ptd

#if PT_DIAG_TAIL
// diagnostic builds only; not part of include/mipt.h
extern "C" int mipt_debug_tail(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(ptd::g_tailDiag), 2 * 4 * 4096 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(ptd::g_tailDiag)) != hipSuccess || hipMemset(p, 0, 2 * 4 * 4096 * sizeof(unsigned long long)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
