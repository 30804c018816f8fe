// kernels_integrate.hip -- wavefront integrator around the traversal kernels.
//
// One bounce iteration `it` (= the reference's currlevel / level) is
//     extend(it) -> shade_a(it) -> occluded(it) -> shade_b(it)
// shade_a resolves the extension ray's result (miss: the path ends with what
// the previous vertex left pending; hit: pending terms are applied), then
// handles the new vertex: metal / glass shoot the next extension ray at once,
// a Lambertian vertex emits its shadow rays and waits in qAct; shade_b
// finishes Lambertian vertices (light sum, Russian roulette, cosine sample).
// Queues are compacted with wavefront ballots: one atomic per 64 lanes.
// Queue lengths stay in device memory, so a frame is enqueued without syncs.
//
// Semantics restated from the reference (cs488.h): Scene::Pathtrace :2230-2261,
// shade_pathtrace :2267-2397, Scene::Raytrace :2200-2229, shade :2402-2496.
// Quirks are kept (see DESIGN.md "integrator quirks").
#include "kernels.hpp"
#include "shading.hpp"

namespace ptd {

enum : int { FLAG_NEGD = 1,        // currDir / viewDir of the vertex is -ray.d
             FLAG_FROM_METAL = 2 };  // direct mode: the ray was shot by a metal vertex
constexpr float kFltMax = 3.402823466e+38f;

// Queue slot for each lane whose pred is set: ballot + popcount, ONE atomicAdd
// per wavefront, on the counter of the shard this wavefront feeds.
// Shard rule (keeps every region within regionSize): a kernel that consumes
// the extension queue (generate, shade_a) feeds shard (wavefront index) %
// kShards; shade_b, which consumes qAct, feeds the shard its slot lies in, so
// the paths that reach next iteration's extension shard s are exactly those
// that shade_a wavefronts = s (mod kShards) handled: at most regionSize.
__device__ __forceinline__ unsigned wave_append(unsigned* shard0, unsigned regionSize, unsigned shard, bool pred) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return 0u;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(shard0 + (size_t)shard * kCntStride, (unsigned)__popcll(mask));
    base = __shfl(base, leader);
    return shard * regionSize + base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
}
// is slot i of a sharded queue live?
__device__ __forceinline__ bool slot_live(const unsigned* shard0, unsigned regionSize, unsigned i) {
    const unsigned sh = i / regionSize;
    return (i - sh * regionSize) < shard0[(size_t)sh * kCntStride];
}

__device__ __forceinline__ vec3 xyz(const float4 v) { return vec3{v.x, v.y, v.z}; }
__device__ __forceinline__ float4 pack(const vec3 v, float w) { return make_float4(v.x, v.y, v.z, w); }

// Path mode keeps several samples of every pixel in flight (slot p = pixel slot * samplesInFlight + sample in flight, slot_of:
// the samples of a pixel sit in adjacent lanes -- they start with the same ray, C:2183-2197 has no jitter), so a
// finished path only RECORDS its radiance; k_resolve_samples adds the records to the AccumulationBuffer in sample
// order afterwards, which keeps the float sums identical to the reference's sample-after-sample loop (C:2240-2241).
#ifndef PT_TUNE_SAMPLE_MINOR
#define PT_TUNE_SAMPLE_MINOR 1  // 1: slot p = pixel slot * samplesInFlight + sample in flight (the samples of a pixel in adjacent lanes); 0: sample * nTilePaths + pixel slot
#endif
__device__ __forceinline__ unsigned slot_of(const FrameParams& fp, unsigned pixelSlot, unsigned sm) {
    return PT_TUNE_SAMPLE_MINOR ? pixelSlot * (unsigned)fp.samplesInFlight + sm : sm * fp.nTilePaths + pixelSlot;
}
__device__ __forceinline__ void slot_split(const FrameParams& fp, unsigned p, unsigned& pixelSlot, unsigned& sm) {
    if (PT_TUNE_SAMPLE_MINOR) { pixelSlot = p / (unsigned)fp.samplesInFlight; sm = p - pixelSlot * (unsigned)fp.samplesInFlight; }
    else { sm = p / fp.nTilePaths; pixelSlot = p - sm * fp.nTilePaths; }
}
enum : int { RES_NONE = 0, RES_ACCUMULATE = 1, RES_BACKGROUND = 2,
             // the path ended at a Lambertian vertex of the LAST bounce (the cap) in a scene with one light: the record holds
             // constTerm, and the vertex' direct light (contrib[] / occ[], FLAG_FOLDED form) is added when the samples are
             // resolved -- the same sum `finish_path(constTerm + L)` would have recorded (C:2316), without a pass of its own
             // over those paths after the last shadow rays
             RES_PENDING = 3 };
__device__ __forceinline__ void finish_path(const PathBuffers& pb, unsigned p, const vec3 result) {
    pb.res[p] = make_float4(result.x, result.y, result.z, i2f(RES_ACCUMULATE));
}
__device__ __forceinline__ void write_pixel(const FrameParams& fp, int pix, const vec3 v) {
    float* q = fp.rgb + (size_t)3 * pix;
    q[0] = v.x; q[1] = v.y; q[2] = v.z;
}

// ---- generate: eye rays for this context's tiles (C:2183-2197) ----
__global__ __launch_bounds__(kBlock) void k_generate(CameraDev cam, FrameParams fp, PathBuffers pb) {
    const unsigned tilePix = (unsigned)(fp.tileW * fp.tileH);
    const unsigned n = fp.nTilePaths * (unsigned)fp.samplesInFlight;
    for (unsigned p = blockIdx.x * kBlock + threadIdx.x; p < pb.space; p += gridDim.x * kBlock) {
        bool valid = p < n;
        int pix = -1;
        if (valid) {
            unsigned j, sm;  // pixel slot, sample in flight
            slot_split(fp, p, j, sm);
            const unsigned k = j / tilePix, r = j % tilePix;
            const unsigned blk = r >> 6, lane = r & 63u;  // 8x8 pixel block per wavefront
            const unsigned bpr = (unsigned)fp.tileW >> 3;
            const int lx = (int)((blk % bpr) * 8u + (lane & 7u));
            const int ly = (int)((blk / bpr) * 8u + (lane >> 3));
            const int t = fp.tileList[k];
            const int x = (t % fp.tilesX) * fp.tileW + lx;
            const int y = (t / fp.tilesX) * fp.tileH + ly;
            valid = (x < fp.width) && (y < fp.height);
            if (valid) {
                pix = y * fp.width + x;
                vec3 o, d;
                eye_ray(cam, x, y, o, d);
                pb.rays[2 * (size_t)p] = pack(o, 0.0f);
                pb.rays[2 * (size_t)p + 1] = pack(d, kFltMax);
            }
            if (sm == 0) pb.pixel[j] = pix;  // one entry per pixel slot: the samples in flight share it
            // (res[p] needs no initial value: every path that starts writes it exactly once -- a miss, a capped or
            // light-hitting vertex, or the pass over the paths that ended at a Lambertian vertex -- before
            // k_resolve_samples reads it, and slots without a pixel are skipped there)
        }
        const unsigned slot = wave_append(pb.counters + counter_index(0, 0, 0), pb.regionSize, (p >> 6) % (unsigned)kShards, valid);
        if (valid) pb.qExt[0][slot] = (int)p;
    }
}

// AccumulationBuffer += result; FrameBuffer = Accum / sampleCount, sample after sample (C:2232, C:2240-2244)
__device__ __forceinline__ float4 settled_record(const PathBuffers& pb, size_t p) {
    float4 r = pb.res[p];
    if (f2i(r.w) == RES_PENDING) {
        const float4 c = pb.contrib[p];  // one light
        vec3 L{0.0f, 0.0f, 0.0f};
        if (!(f2i(c.w) && pb.occ[p])) L = L + xyz(c);  // pending_light(), C:2303-2309
        const vec3 v = xyz(r) + L;                     // C:2316 with L already times multTerm (FLAG_FOLDED)
        r = make_float4(v.x, v.y, v.z, i2f(RES_ACCUMULATE));
    }
    return r;
}
#if PT_TUNE_SAMPLE_MINOR
// The samples in flight of a pixel lie side by side (slot_of), 16 B each, and a pixel's sum runs over them in order: a block
// fetches the records of its 256 pixels, kResChunk samples at a time, with every lane on the next 16 B of memory (a pixel's
// chunk is one 128-B line), settles the pending ones there, and hands them to the pixels' threads through LDS (component
// arrays, kResChunk + 1 floats per pixel: the threads' walks hit distinct banks).  One thread per pixel reading its own
// 16-B records would fetch every line eight times over.
constexpr int kResChunk = 8;
__global__ __launch_bounds__(kBlock) void k_resolve_samples(FrameParams fp, PathBuffers pb) {
    constexpr int kRow = kResChunk + 1;
    __shared__ float lx[kBlock * kRow], ly[kBlock * kRow], lz[kBlock * kRow];
    __shared__ int lk[kBlock * kRow];
    const unsigned S = (unsigned)fp.samplesInFlight, t = threadIdx.x;
    for (unsigned jBase = blockIdx.x * kBlock; jBase < fp.nTilePaths; jBase += gridDim.x * kBlock) {
        const unsigned j = jBase + t;
        const int pix = j < fp.nTilePaths ? pb.pixel[j] : -1;
        float* a = fp.accum + (size_t)3 * (pix < 0 ? 0 : pix);
        float ax = 0.0f, ay = 0.0f, az = 0.0f, qx = 0.0f, qy = 0.0f, qz = 0.0f;
        if (pix >= 0) { ax = a[0]; ay = a[1]; az = a[2]; }
        bool touched = false, shown = false;
        for (unsigned c0 = 0; c0 < S; c0 += kResChunk) {
            const unsigned cc = (S - c0) < (unsigned)kResChunk ? (S - c0) : (unsigned)kResChunk;
            __syncthreads();  // the chunk before has been consumed
            for (unsigned e = t; e < kBlock * cc; e += kBlock) {
                const unsigned jj = e / cc, sl = e - jj * cc;
                float4 r = make_float4(0.0f, 0.0f, 0.0f, i2f(RES_NONE));
                if (jBase + jj < fp.nTilePaths) r = settled_record(pb, (size_t)(jBase + jj) * S + c0 + sl);
                lx[jj * kRow + sl] = r.x; ly[jj * kRow + sl] = r.y; lz[jj * kRow + sl] = r.z; lk[jj * kRow + sl] = f2i(r.w);
            }
            __syncthreads();
            if (pix >= 0) {
                for (unsigned sl = 0; sl < cc; sl++) {
                    const int kind = lk[t * kRow + sl];
                    const float rx = lx[t * kRow + sl], ry = ly[t * kRow + sl], rz = lz[t * kRow + sl];
                    if (kind == RES_ACCUMULATE) {
                        ax += rx; ay += ry; az += rz;
                        const float sc = (float)(fp.sample + c0 + sl + 1u);  // sampleCount++ precedes the sample's loop
                        qx = ax / sc; qy = ay / sc; qz = az / sc;           // (the frame buffer keeps the last one)
                        touched = true; shown = true;
                    } else if (kind == RES_BACKGROUND) {
                        qx = rx; qy = ry; qz = rz;
                        shown = true;
                    }
                }
            }
        }
        if (touched) { a[0] = ax; a[1] = ay; a[2] = az; }
        if (shown) { float* q = fp.rgb + (size_t)3 * pix; q[0] = qx; q[1] = qy; q[2] = qz; }
    }
}
#else
__global__ __launch_bounds__(kBlock) void k_resolve_samples(FrameParams fp, PathBuffers pb) {
    for (unsigned j = blockIdx.x * kBlock + threadIdx.x; j < fp.nTilePaths; j += gridDim.x * kBlock) {
        const int pix = pb.pixel[j];
        if (pix < 0) continue;
        float* a = fp.accum + (size_t)3 * pix;
        float* q = fp.rgb + (size_t)3 * pix;
        float ax = a[0], ay = a[1], az = a[2];
        bool touched = false;
        for (int sm = 0; sm < fp.samplesInFlight; sm++) {
            const float4 r = settled_record(pb, slot_of(fp, j, (unsigned)sm));
            const int kind = f2i(r.w);
            if (kind == RES_ACCUMULATE) {
                ax += r.x; ay += r.y; az += r.z;
                const float sc = (float)(fp.sample + (unsigned)sm + 1u);  // sampleCount++ precedes the sample's loop
                q[0] = ax / sc; q[1] = ay / sc; q[2] = az / sc;
                touched = true;
            } else if (kind == RES_BACKGROUND) {
                q[0] = r.x; q[1] = r.y; q[2] = r.z;
            }
        }
        if (touched) { a[0] = ax; a[1] = ay; a[2] = az; }
    }
}
#endif

// emit the next extension ray of path p
__device__ __forceinline__ void emit_extension(const PathBuffers& pb, unsigned p, const vec3 P, float tmin, const vec3 dir) {
    pb.rays[2 * (size_t)p] = pack(P, tmin);
    pb.rays[2 * (size_t)p + 1] = pack(dir, kFltMax);
}

// ---- path mode ----
// ONE shading kernel per bounce: shade(it) runs after extend(it) and before occluded(it).  A Lambertian vertex
// computes its light terms, emits its shadow rays AND (unless the path ends here) samples and emits the next
// extension ray in the same pass -- the sampled direction does not depend on the shadow rays' outcome.  The
// direct light L of that vertex (C:2303-2309) is summed one bounce later, when the next shade reads contrib[] and
// the occlusion flags: the same additions in the same order, only later.  Paths that end at a Lambertian vertex
// (bounce cap, Russian roulette) wait in the "fin" queue for their shadow rays and are resolved by the next
// shade call; the host issues one last finals-only call after the loop.
enum : int { FLAG_PEND_LAMBERT = 4,   // the previous vertex was Lambertian: its L is still in contrib[] / occ[]
             // One light (every shipped configuration): L = 0 + c0, so L * multTerm (C:2316 / C:2322) is c0 * multTerm and
             // can be formed when c0 is, and the weight the next hit applies to multTerm (C:2323) can be applied at
             // once: contrib[] then holds c0 * multTerm, mult[] the multiplied term, the rng dimension rides in the
             // flags' upper half, and the vertex neither writes hitMul[] nor does the next one read it (32 B per
             // Lambertian bounce of ~260).  Same operands, same products, formed one kernel earlier.
             FLAG_FOLDED = 8 };

__device__ __forceinline__ vec3 pending_light(const PathBuffers& pb, unsigned p, int nl) {
    vec3 L{0.0f, 0.0f, 0.0f};
    for (int k = 0; k < nl; k++) {  // C:2303-2309
        const float4 c = pb.contrib[(size_t)p * nl + k];
        if (f2i(c.w) && pb.occ[(size_t)p * nl + k]) continue;
        L = L + xyz(c);
    }
    return L;
}

#ifndef PT_TUNE_SHADE_WAVES
#define PT_TUNE_SHADE_WAVES 6  // 80 VGPRs, no spills (0 = the compiler's choice: 97 VGPRs, 5 wavefronts per SIMD, 3 % slower; 7: spills, 6 % slower)
#endif
#if PT_TUNE_SHADE_WAVES
__attribute__((amdgpu_waves_per_eu(PT_TUNE_SHADE_WAVES, PT_TUNE_SHADE_WAVES)))
#endif
__global__ __launch_bounds__(kBlock) void k_shade_path(DevScene s, FrameParams fp, PathBuffers pb, int it, int finalsOnly) {
    const int nl = pb.nLights;
    const bool fold = nl == 1;  // see FLAG_FOLDED
    // The "fin" queue of bounce it is written by this call while the one of bounce it-1 is still being read by other
    // blocks of it (Russian roulette ends paths at every bounce after the fifth): two buffers, by parity.  vP is the
    // direct mode's vertex cache, idle in path mode, and larger than a queue.
    int* const qFin[2] = {pb.qAct, (int*)pb.vP};
    // ---- paths that ended at a Lambertian vertex of bounce it-1: their shadow rays are traced now ----
    if (it > 0) {
        const unsigned* cFin = pb.counters + counter_index(it - 1, 1, 0);
        for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < pb.space; i += gridDim.x * kBlock) {
            if (!slot_live(cFin, pb.regionSize, i)) continue;
            const unsigned p = (unsigned)qFin[(it - 1) & 1][i];
            const float4 c4 = pb.cnst[p];
            const vec3 L = pending_light(pb, p, nl);
            finish_path(pb, p, (f2i(c4.w) & FLAG_FOLDED) ? xyz(c4) + L : xyz(c4) + xyz(pb.mult[p]) * L);  // C:2316
        }
    }
    if (finalsOnly) return;

    const unsigned* cCur = pb.counters + counter_index(it, 0, 0);
    const int* q = pb.qExt[it & 1];
    int* qNext = pb.qExt[(it + 1) & 1];
    unsigned* cNext = pb.counters + counter_index(it + 1, 0, 0);
    unsigned* cFinOut = pb.counters + counter_index(it, 1, 0);
    unsigned* cSh = pb.counters + counter_index(it, 2, 0);
    const bool capped = (fp.maxBounces >= 0) && (it >= fp.maxBounces);

    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < pb.space; i += gridDim.x * kBlock) {
        const bool valid = slot_live(cCur, pb.regionSize, i);
        bool toExt = false, toFin = false;
        unsigned p = 0;
        if (valid) {
            p = (unsigned)q[i];
            const vec3 d = xyz(pb.rays[2 * (size_t)p + 1]);
            const float4 h = pb.hit[p];
            const int tri = f2i(h.x);
            vec3 mult{1.0f, 1.0f, 1.0f}, cnst{0.0f, 0.0f, 0.0f};
            float eta = 1.0f;
            int flags = FLAG_NEGD;
            unsigned dim = 0;  // rng dimension
            bool alive = true;
            if (it == 0) {
                if (tri < 0) {  // C:2244: miss pixels show the background, unaccumulated
                    const vec3 bg = get_background(s, -d);
                    pb.res[p] = make_float4(bg.x, bg.y, bg.z, i2f(RES_BACKGROUND));
                    alive = false;
                }
            } else {
                const float4 m4 = pb.mult[p], c4 = pb.cnst[p];
                mult = xyz(m4); eta = m4.w; cnst = xyz(c4); flags = f2i(c4.w);
                const bool pendLambert = (flags & FLAG_PEND_LAMBERT) != 0;
                vec3 L{0.0f, 0.0f, 0.0f};
                if (pendLambert) L = pending_light(pb, p, nl);
                const bool folded = (flags & FLAG_FOLDED) != 0;  // L is already times multTerm, mult already weighted
                if (tri < 0) {
                    const vec3 result = pendLambert ? (folded ? cnst + L : cnst + mult * L) : cnst + mult * get_background(s, d);
                    finish_path(pb, p, result);
                    alive = false;
                } else if (folded) {
                    cnst = cnst + L;                          // C:2322
                    dim = (unsigned)flags >> 16;
                } else {
                    if (pendLambert) cnst = cnst + L * mult;  // C:2322
                    const float4 hm = pb.hitMul[p];
                    mult = mult * xyz(hm);                    // C:2323 / C:2348
                    dim = (unsigned)f2i(hm.w);
                }
            }
            if (alive) {
                const HitInfo hi = reconstruct_hit(s, tri, h.y, h.z, h.w, d);
                const MaterialDev m = load_material(s, hi.material);
                const vec3 currDir = (flags & FLAG_NEGD) ? -d : d;
                const vec3 P = hi.P, N = hi.N;
                if (m.type == MIPT_MAT_LAMBERTIAN) {
                    vec3 texture{1.0f, 1.0f, 1.0f};
                    if (m.texOfs >= 0) texture = texture * fetch_texture(s, m, hit_texcoord(s, tri, h.z, h.w));
                    const vec3 brdf = m.Kd / kPI;
                    for (int k = 0; k < nl; k++) {  // C:2287-2310
                        const float4 lp = s.lights[2 * k], lw = s.lights[2 * k + 1];
                        vec3 l = xyz(lp) - P;
                        const float falloff = pt::length2(l);
                        l = l / sqrtf(falloff);
                        const float ndl = pt::dot(N, l);
                        const vec3 irradiance = light_factor(ndl, falloff) * xyz(lw);
                        const bool need = ndl > 0.0f;
                        const vec3 c = (irradiance * texture) * brdf;
                        pb.contrib[(size_t)p * nl + k] = pack(fold ? c * mult : c, i2f(need ? 1 : 0));
                        pb.occ[(size_t)p * nl + k] = 0;
                        // shadow ray P -> light, (0, dist), C:2303 / C:2428; the lanes here are
                        // exactly the Lambertian vertices, the ballot sees only them
                        const unsigned sl = wave_append(cSh, pb.regionSize * (unsigned)nl, (i >> 6) % (unsigned)kShards, need);
                        if (need) {
                            pb.shRays[2 * (size_t)sl] = pack(P, 0.0f);
                            pb.shRays[2 * (size_t)sl + 1] = pack(pt::normalize(l), sqrtf(falloff));
                            pb.shTarget[sl] = (int)((size_t)p * nl + k);
                        }
                    }
                    unsigned pixelSlot, inFlight;  // inFlight: which of the samples in flight this path is
                    slot_split(fp, p, pixelSlot, inFlight);
                    const int pix = pb.pixel[pixelSlot];
                    const unsigned sample = fp.sample + inFlight;
                    bool done = capped;
                    if (!done && it > 5) {  // C:2311-2314, no weight compensation
                        if (counter_rand(fp.seedHash, (unsigned)pix, sample, dim++) < .25f) done = true;
                    }
                    if (done && fold && capped) {
                        pb.res[p] = pack(cnst, i2f(RES_PENDING));  // settled by k_resolve_samples, after the last shadow rays
                    } else if (done) {
                        if (!fold) pb.mult[p] = pack(mult, eta);
                        pb.cnst[p] = pack(cnst, i2f(fold ? (FLAG_PEND_LAMBERT | FLAG_FOLDED) : FLAG_PEND_LAMBERT));
                        toFin = true;
                    } else {
                        const float u1 = counter_rand(fp.seedHash, (unsigned)pix, sample, dim++);
                        const float u2 = counter_rand(fp.seedHash, (unsigned)pix, sample, dim++);
                        const vec3 smp = sample_lambert(N, u1, u2);
                        const vec3 sample_brdf = m.Kd / kPI;
                        const float pdf = 1.0f / kPI;
                        emit_extension(pb, p, P, 0.001f, smp);
                        const vec3 weight = (texture * sample_brdf) / pdf;
                        if (fold) {
                            pb.mult[p] = pack(mult * weight, eta);
                            pb.cnst[p] = pack(cnst, i2f(FLAG_PEND_LAMBERT | FLAG_FOLDED | (int)(dim << 16)));
                        } else {
                            pb.mult[p] = pack(mult, eta);
                            pb.hitMul[p] = pack(weight, i2f((int)dim));
                            pb.cnst[p] = pack(cnst, i2f(FLAG_PEND_LAMBERT));  // the next vertex reads currDir = ray.d
                        }
                        toExt = true;
                    }
                } else if (m.type == MIPT_MAT_METAL) {
                    if (capped) {
                        finish_path(pb, p, cnst);
                    } else {
                        const vec3 smp = mirror(-currDir, N);  // C:496
                        emit_extension(pb, p, P, 0.001f, smp);
                        pb.hitMul[p] = pack(m.Ks, i2f((int)dim));
                        pb.mult[p] = pack(mult, eta);
                        pb.cnst[p] = pack(cnst, i2f(0));
                        toExt = true;
                    }
                } else if (m.type == MIPT_MAT_GLASS) {
                    if (capped) {
                        finish_path(pb, p, cnst);
                    } else {
                        const float eta2 = (m.eta == eta) ? 1.0f : m.eta;  // C:2359
                        const vec3 w = -currDir;
                        const float ratio = eta / eta2;
                        const float inside = snell_inside(ratio, pt::dot(w, N));
                        pb.hitMul[p] = make_float4(1.0f, 1.0f, 1.0f, i2f((int)dim));
                        if (inside >= 0.0f) {  // C:2366-2377; the ray direction is -refracted
                            emit_extension(pb, p, P, 0.01f, refract_dir(w, N, ratio, inside));
                            pb.mult[p] = pack(mult, eta2);
                            pb.cnst[p] = pack(cnst, i2f(FLAG_NEGD));
                        } else {  // total internal reflection, C:2378-2389
                            emit_extension(pb, p, P, 0.0f, mirror(w, N));
                            pb.mult[p] = pack(mult, eta);
                            pb.cnst[p] = pack(cnst, i2f(0));
                        }
                        toExt = true;
                    }
                } else {
                    finish_path(pb, p, vec3{0.0f, 0.0f, 0.0f});  // C:2391-2394, constTerm is dropped
                }
            }
        }
        const unsigned sExt = wave_append(cNext, pb.regionSize, (i >> 6) % (unsigned)kShards, toExt);
        if (toExt) qNext[sExt] = (int)p;
        const unsigned sFin = wave_append(cFinOut, pb.regionSize, (i >> 6) % (unsigned)kShards, toFin);
        if (toFin) qFin[it & 1][sFin] = (int)p;
    }
}

// ---- direct mode (shade) ----
__device__ __forceinline__ vec3 unwind_metal(const PathBuffers& pb, unsigned p, int depth, vec3 v) {
    for (int k = depth - 1; k >= 0; --k) v = xyz(pb.ksStack[(size_t)p * 6 + k]) * v;  // C:2463
    return v;
}

__global__ __launch_bounds__(kBlock) void k_shade_a_direct(DevScene s, FrameParams fp, PathBuffers pb, int it) {
    const unsigned* cCur = pb.counters + counter_index(it, 0, 0);
    const int* q = pb.qExt[it & 1];
    int* qNext = pb.qExt[(it + 1) & 1];
    unsigned* cNext = pb.counters + counter_index(it + 1, 0, 0);
    unsigned* cAct = pb.counters + counter_index(it, 1, 0);
    unsigned* cSh = pb.counters + counter_index(it, 2, 0);
    const int nl = pb.nLights;
    const int level = it;

    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < pb.space; i += gridDim.x * kBlock) {
        const bool valid = slot_live(cCur, pb.regionSize, i);
        bool toExt = false, toAct = false;
        unsigned p = 0;
        vec3 P{0, 0, 0};
        if (valid) {
            p = (unsigned)q[i];
            const int pix = pb.pixel[p];
            const vec3 d = xyz(pb.rays[2 * (size_t)p + 1]);
            const float4 h = pb.hit[p];
            const int tri = f2i(h.x);
            float eta = 1.0f;
            int flags = FLAG_NEGD, depth = 0;
            if (it > 0) {
                const float4 c4 = pb.cnst[p];
                eta = pb.mult[p].w;
                flags = f2i(c4.w) & 0xff;
                depth = f2i(c4.w) >> 8;
            }
            if (tri < 0) {  // C:2212, C:2462, C:2478, C:2488: always the background at -d; a metal
                            // vertex returns it WITHOUT its own Ks (C:2461-2462)
                const int applied = depth - ((flags & FLAG_FROM_METAL) ? 1 : 0);
                write_pixel(fp, pix, unwind_metal(pb, p, applied, get_background(s, -d)));
            } else if (fp.mode == 0) {  // primary only: (t, triangle id, material)
                const HitInfo hi = reconstruct_hit(s, tri, h.y, h.z, h.w, d);
                write_pixel(fp, pix, vec3{h.y, (float)tri, (float)hi.material});
            } else {
                const HitInfo hi = reconstruct_hit(s, tri, h.y, h.z, h.w, d);
                const MaterialDev m = load_material(s, hi.material);
                const vec3 viewDir = (flags & FLAG_NEGD) ? -d : d;
                const vec3 N = hi.N;
                P = hi.P;
                if (m.type == MIPT_MAT_LAMBERTIAN) {
                    vec3 brdf = m.Kd / kPI;
                    if (m.texOfs >= 0) brdf = brdf * fetch_texture(s, m, hit_texcoord(s, tri, h.z, h.w));
                    for (int k = 0; k < nl; k++) {  // C:2409-2435
                        const float4 lp = s.lights[2 * k], lw = s.lights[2 * k + 1];
                        vec3 l = xyz(lp) - P;
                        const float falloff = pt::length2(l);
                        l = l / sqrtf(falloff);
                        const float ndl = pt::dot(N, l);
                        const vec3 irradiance = light_factor(ndl, falloff) * xyz(lw);
                        const bool need = ndl > 0.0f;
                        pb.contrib[(size_t)p * nl + k] = pack(irradiance, i2f(need ? 1 : 0));
                        pb.occ[(size_t)p * nl + k] = 0;
                        // shadow ray P -> light, (0, dist), C:2303 / C:2428; the lanes here are
                        // exactly the Lambertian vertices, the ballot sees only them
                        const unsigned sl = wave_append(cSh, pb.regionSize * (unsigned)nl, (i >> 6) % (unsigned)kShards, need);
                        if (need) {
                            pb.shRays[2 * (size_t)sl] = pack(P, 0.0f);
                            pb.shRays[2 * (size_t)sl + 1] = pack(pt::normalize(l), sqrtf(falloff));
                            pb.shTarget[sl] = (int)((size_t)p * nl + k);
                        }
                    }
                    pb.vTex[p] = pack(brdf, 0.0f);
                    pb.cnst[p] = make_float4(0.0f, 0.0f, 0.0f, i2f(flags | (depth << 8)));
                    toAct = true;
                } else if (m.type == MIPT_MAT_METAL) {
                    if (level > 5) {
                        write_pixel(fp, pix, unwind_metal(pb, p, depth, vec3{(float).1, (float).1, (float).1}));
                    } else {
                        const vec3 reflected = mirror(-viewDir, N);  // C:2459
                        emit_extension(pb, p, P, 0.001f, reflected);
                        pb.ksStack[(size_t)p * 6 + depth] = pack(m.Ks, 0.0f);
                        pb.cnst[p] = make_float4(0.0f, 0.0f, 0.0f, i2f(FLAG_FROM_METAL | ((depth + 1) << 8)));
                        pb.mult[p] = make_float4(1.0f, 1.0f, 1.0f, eta);
                        toExt = true;
                    }
                } else if (m.type == MIPT_MAT_GLASS) {
                    if (level > 5) {
                        write_pixel(fp, pix, unwind_metal(pb, p, depth, vec3{0.0f, 0.0f, 0.0f}));
                    } else {
                        const float eta2 = (m.eta == eta) ? 1.0f : m.eta;
                        const vec3 w = -viewDir;
                        const float ratio = eta / eta2;
                        const float inside = snell_inside(ratio, pt::dot(w, N));
                        if (inside >= 0.0f) {  // C:2474-2480: next viewDir = -refracted = -ray.d
                            emit_extension(pb, p, P, 0.0f, refract_dir(w, N, ratio, inside));
                            pb.cnst[p] = make_float4(0.0f, 0.0f, 0.0f, i2f(FLAG_NEGD | (depth << 8)));
                            pb.mult[p] = make_float4(1.0f, 1.0f, 1.0f, eta2);
                        } else {  // C:2481-2490
                            emit_extension(pb, p, P, 0.0f, mirror(w, N));
                            pb.cnst[p] = make_float4(0.0f, 0.0f, 0.0f, i2f(0 | (depth << 8)));
                            pb.mult[p] = make_float4(1.0f, 1.0f, 1.0f, eta);
                        }
                        toExt = true;
                    }
                } else {
                    write_pixel(fp, pix, unwind_metal(pb, p, depth, vec3{100.0f, 0.0f, 100.0f}));  // C:2494
                }
            }
        }
        const unsigned sExt = wave_append(cNext, pb.regionSize, (i >> 6) % (unsigned)kShards, toExt);
        if (toExt) qNext[sExt] = (int)p;
        const unsigned sAct = wave_append(cAct, pb.regionSize, (i >> 6) % (unsigned)kShards, toAct);
        if (toAct) pb.qAct[sAct] = (int)p;
    }
}

__global__ __launch_bounds__(kBlock) void k_shade_b_direct(DevScene s, FrameParams fp, PathBuffers pb, int it) {
    const unsigned* cCur = pb.counters + counter_index(it, 1, 0);
    const int nl = pb.nLights;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < pb.space; i += gridDim.x * kBlock) {
        if (!slot_live(cCur, pb.regionSize, i)) continue;
        const unsigned p = (unsigned)pb.qAct[i];
        const int pix = pb.pixel[p];
        const vec3 brdf = xyz(pb.vTex[p]);
        const int depth = f2i(pb.cnst[p].w) >> 8;
        vec3 L{0.0f, 0.0f, 0.0f};
        for (int k = 0; k < nl; k++) {
            const float4 c = pb.contrib[(size_t)p * nl + k];
            vec3 irradiance = xyz(c);
            if (f2i(c.w) && pb.occ[(size_t)p * nl + k]) irradiance = vec3{0.1f, 0.1f, 0.1f} * irradiance;  // C:2428-2431
            L = L + irradiance * brdf;
        }
        write_pixel(fp, pix, unwind_metal(pb, p, depth, L));
    }
}

// one wavefront: lane = shard
// ... which also leaves the counters and cursors of the iterations the frame used at zero for the next frame (no 1.6-MB
// memset per frame)
__global__ __launch_bounds__(64) void k_fold_counters(unsigned* __restrict__ counters, int nIter, unsigned long long* totals) {
    const int s = threadIdx.x;
    unsigned long long e = 0, sh = 0;
    for (int it = 0; it < nIter; it++) {
        e += counters[counter_index(it, 0, s)];
        sh += counters[counter_index(it, 2, s)];
    }
    const int nUsed = nIter + 2 < kMaxIter ? nIter + 2 : kMaxIter;  // shade(it) appends to the extension queue of it + 1
    for (int it = 0; it < nUsed; it++)
        for (int qu = 0; qu < kQueuesPerIter; qu++) counters[counter_index(it, qu, s)] = 0u;
    for (int off = 32; off > 0; off >>= 1) {
        e += __shfl_down(e, off);
        sh += __shfl_down(sh, off);
    }
    if (s == 0) {
        totals[0] += e;
        totals[1] += sh;
    }
}

// ---- 8-bit frame: Image::save's per-channel conversion (C:159-168, C:228) on the device, four channels per lane ----
// (unsigned char)(255.0f * gammaCorrection(toneMapping(v))) with gamma 1.0: pow(x, 1) is x, fmin/fmax are IEEE
// minNum/maxNum on both sides, the conversion truncates; same bytes as pt::packChannel8 on the host.
__device__ __forceinline__ unsigned pack_channel8(float r) {
    const float tm = fmaxf(fminf(1.0f, r), 0.0f);
    return (unsigned)(unsigned char)(255.0f * tm);
}
__global__ __launch_bounds__(kBlock) void k_pack_rgb8(const float* __restrict__ rgb, size_t nValues, unsigned char* __restrict__ out) {
    const size_t nQuads = nValues >> 2;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nQuads; i += (size_t)gridDim.x * kBlock) {
        const float4 v = ((const float4*)rgb)[i];
        ((unsigned*)out)[i] = pack_channel8(v.x) | (pack_channel8(v.y) << 8) | (pack_channel8(v.z) << 16) | (pack_channel8(v.w) << 24);
    }
    if (blockIdx.x == 0 && threadIdx.x < (nValues & 3)) {
        const size_t i = (nQuads << 2) + threadIdx.x;
        out[i] = (unsigned char)pack_channel8(rgb[i]);
    }
}

// ---- launchers ----
#ifndef PT_TUNE_INTEGRATE_GRID
#define PT_TUNE_INTEGRATE_GRID 65536  // r1, 5 wavefronts: 8192: 10.30 ms of shading per bench step, 16384: 10.23, 32768: 10.13, one item per thread: 10.42; r2, 6 wavefronts: 32768: 9.79, 65536: 9.6, 131072: 9.7
#endif
constexpr size_t kIntegrateGrid = PT_TUNE_INTEGRATE_GRID;  // blocks of the grid-strided integrator kernels
static inline unsigned grid_for(size_t n) {
    size_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > kIntegrateGrid) g = kIntegrateGrid;
    return (unsigned)g;
}

void launch_generate(hipStream_t st, const CameraDev& cam, const FrameParams& fp, const PathBuffers& pb) {
    hipLaunchKernelGGL(k_generate, dim3(grid_for(pb.space)), dim3(kBlock), 0, st, cam, fp, pb);
}
void launch_shade_a(hipStream_t st, const DevScene& s, const FrameParams& fp, const PathBuffers& pb, int iter) {
    const unsigned g = grid_for(pb.space);
    hipLaunchKernelGGL(k_shade_a_direct, dim3(g), dim3(kBlock), 0, st, s, fp, pb, iter);
}
void launch_shade_b(hipStream_t st, const DevScene& s, const FrameParams& fp, const PathBuffers& pb, int iter) {
    const unsigned g = grid_for(pb.space);
    hipLaunchKernelGGL(k_shade_b_direct, dim3(g), dim3(kBlock), 0, st, s, fp, pb, iter);
}
void launch_shade_path(hipStream_t st, const DevScene& s, const FrameParams& fp, const PathBuffers& pb, int iter, bool finalsOnly) {
    hipLaunchKernelGGL(k_shade_path, dim3(grid_for(pb.space)), dim3(kBlock), 0, st, s, fp, pb, iter, finalsOnly ? 1 : 0);
}

void launch_pack_rgb8(hipStream_t st, const float* rgb, size_t nValues, unsigned char* out) {
    hipLaunchKernelGGL(k_pack_rgb8, dim3(grid_for((nValues >> 2) + 1)), dim3(kBlock), 0, st, rgb, nValues, out);
}
void launch_resolve_samples(hipStream_t st, const FrameParams& fp, const PathBuffers& pb) {
    hipLaunchKernelGGL(k_resolve_samples, dim3(grid_for(fp.nTilePaths)), dim3(kBlock), 0, st, fp, pb);
}
void launch_fold_counters(hipStream_t st, unsigned* counters, int nIter, unsigned long long* totals) {
    hipLaunchKernelGGL(k_fold_counters, dim3(1), dim3(64), 0, st, counters, nIter, totals);
}

}  // namespace ptd
