// kernels_trace.hip -- the traversal kernels (closest-hit "extend", any-hit
// "occluded") and the small kernels around the parity entry points: eye-ray
// generation, HitInfo resolution, environment lookup.
//
// Launch shape: 256-thread blocks (4 wavefronts of 64 lanes), one ray per
// lane, grid-stride over the ray queue.  The queue length may live in device
// memory (countPtr) so that a whole frame is enqueued without host syncs.
#include "kernels.hpp"
#include "shading.hpp"

namespace ptd {

// rays: 2 float4 per ray; index (nullable) maps queue slot -> ray/hit slot.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace_closest(DevScene s, const float4* __restrict__ rays,
                                                          const int* __restrict__ index, QueueView q,
                                                          float4* __restrict__ hits,
                                                          unsigned long long* __restrict__ counters) {
    unsigned nv = 0, nt = 0;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < q.space; i += gridDim.x * kBlock) {
        if (q.counts) {  // sharded queue: slot i is live iff it lies below its shard's counter
            const unsigned sh = i / q.regionSize;
            if (i - sh * q.regionSize >= q.counts[(size_t)sh * kCntStride]) continue;
        }
        const unsigned p = index ? (unsigned)index[i] : i;
        const float4 r0 = rays[2 * (size_t)p], r1 = rays[2 * (size_t)p + 1];
        const Hit h = walk<false, COUNT>(s.nodes, s.leafTris, vec3{r0.x, r0.y, r0.z}, vec3{r1.x, r1.y, r1.z}, r0.w,
                                         r1.w, nv, nt);
        hits[p] = make_float4(i2f(h.tri), h.t, h.beta, h.gamma);
    }
    if (COUNT) {
        // wave-level reduction, one atomic pair per wavefront
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

// any-hit: occ[target[i]] = 1 when occluded (never writes 0: the caller clears)
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace_occluded(DevScene s, const float4* __restrict__ rays,
                                                           const int* __restrict__ target, QueueView q,
                                                           unsigned char* __restrict__ occ,
                                                           unsigned long long* __restrict__ counters) {
    unsigned nv = 0, nt = 0;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < q.space; i += gridDim.x * kBlock) {
        if (q.counts) {
            const unsigned sh = i / q.regionSize;
            if (i - sh * q.regionSize >= q.counts[(size_t)sh * kCntStride]) continue;
        }
        const float4 r0 = rays[2 * (size_t)i], r1 = rays[2 * (size_t)i + 1];
        const Hit h = walk<true, COUNT>(s.nodes, s.leafTris, vec3{r0.x, r0.y, r0.z}, vec3{r1.x, r1.y, r1.z}, r0.w, r1.w,
                                        nv, nt);
        const unsigned t = target ? (unsigned)target[i] : i;
        if (h.tri >= 0) occ[t] = 1;
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

// ---- launchers ----
static inline unsigned grid_for(size_t n) {
    size_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > kMaxGrid) g = kMaxGrid;
    return (unsigned)g;
}

void launch_trace_closest(hipStream_t st, const DevScene& s, const float4* rays, const int* index, QueueView q,
                          float4* hits, unsigned long long* counters) {
    const unsigned g = grid_for(q.space);
    if (counters)
        hipLaunchKernelGGL(k_trace_closest<true>, dim3(g), dim3(kBlock), 0, st, s, rays, index, q, hits, counters);
    else
        hipLaunchKernelGGL(k_trace_closest<false>, dim3(g), dim3(kBlock), 0, st, s, rays, index, q, hits, counters);
}
void launch_trace_occluded(hipStream_t st, const DevScene& s, const float4* rays, const int* target, QueueView q,
                           unsigned char* occ, unsigned long long* counters) {
    const unsigned g = grid_for(q.space);
    if (counters)
        hipLaunchKernelGGL(k_trace_occluded<true>, dim3(g), dim3(kBlock), 0, st, s, rays, target, q, occ, counters);
    else
        hipLaunchKernelGGL(k_trace_occluded<false>, dim3(g), dim3(kBlock), 0, st, s, rays, target, q, occ, counters);
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
