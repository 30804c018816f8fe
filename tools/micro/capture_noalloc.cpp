// capture_noalloc.cpp -- mipt_trace_device_noalloc under HIP graph capture, with the HIP API alone (no torch): the call is
// recorded into a graph on a capturing stream (hipStreamCaptureModeGlobal: an allocation, a free or a synchronisation
// by any thread would fail the capture), the graph is launched twice on fresh rays, and the hits are compared with
// mipt_trace_device's.  Prints "capture ok" and returns 0 when they agree bit for bit.
//   hipcc -O2 -std=c++17 tools/micro/capture_noalloc.cpp -o tools/micro/capture_noalloc -Lpath-tracer_amd -lmipt -Wl,-rpath,'$ORIGIN/../../path-tracer_amd'
//   tools/micro/capture_noalloc <scene.miptsc>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/mipt.h"

#define CHECK(expr)                                                                                   \
    do {                                                                                              \
        if ((expr) != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #expr, hipGetErrorString(hipGetLastError())); return 1; } \
    } while (0)
#define MCHECK(expr)                                                                   \
    do {                                                                               \
        if ((expr) != MIPT_OK) { fprintf(stderr, "%s failed: %s\n", #expr, mipt_last_error()); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: capture_noalloc scene.miptsc   (a flat scene file, mipt_scene_cache_save)\n"); return 2; }
    mipt_mesh* mesh = nullptr;
    mipt_bvh* bvh = nullptr;
    MCHECK(mipt_scene_cache_load(argv[1], &mesh, &bvh));
    mipt_light light{{3.0f, 3.0f, 3.0f}, {1000.0f, 1000.0f, 1000.0f}};
    mipt_scene_desc desc;
    MCHECK(mipt_scene_desc_fill(mesh, bvh, nullptr, &light, 1, &desc));
    mipt_scene* scene = nullptr;
    MCHECK(mipt_scene_create(&desc, &scene));
    hipStream_t stream;
    CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    mipt_ctx* ctx = nullptr;
    MCHECK(mipt_context_create(scene, 0, stream, &ctx));
    const size_t n = 20000;
    float *dRays = nullptr, *dHits = nullptr, *dRef = nullptr;
    CHECK(hipMalloc((void**)&dRays, n * 32));
    CHECK(hipMalloc((void**)&dHits, n * 16));
    CHECK(hipMalloc((void**)&dRef, n * 16));
    CHECK(hipMemset(dRays, 0, n * 32));
    MCHECK(mipt_trace_device_noalloc(ctx, dRays, n, dHits));  // (the runtime loads the kernels' code on their first launch)
    CHECK(hipStreamSynchronize(stream));
    size_t free0 = 0, free1 = 0, total = 0;
    CHECK(hipMemGetInfo(&free0, &total));
    MCHECK(mipt_trace_device_noalloc(ctx, dRays, n, dHits));
    CHECK(hipStreamSynchronize(stream));
    CHECK(hipMemGetInfo(&free1, &total));
    if (free0 != free1) { fprintf(stderr, "a plain call changed free device memory: %zu -> %zu\n", free0, free1); return 1; }
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
    MCHECK(mipt_trace_device_noalloc(ctx, dRays, n, dHits));
    CHECK(hipStreamEndCapture(stream, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    std::vector<float> rays(n * 8), got(n * 4), ref(n * 4);
    for (int rep = 0; rep < 2; rep++) {
        for (size_t i = 0; i < n; i++) {
            float* r = &rays[8 * i];
            r[0] = 0.5f * u(rng); r[1] = 0.5f * u(rng) + 0.1f; r[2] = 0.5f * u(rng); r[3] = 1e-4f;
            r[4] = u(rng); r[5] = u(rng); r[6] = u(rng); r[7] = 3.402823466e+38f;
            if (i < 400) r[4 + i % 3] = 0.0f;  // handed back: the list in the static scratch is used
        }
        CHECK(hipMemcpy(dRays, rays.data(), n * 32, hipMemcpyHostToDevice));
        CHECK(hipMemset(dHits, 0xee, n * 16));
        CHECK(hipGraphLaunch(exec, stream));
        CHECK(hipStreamSynchronize(stream));
        CHECK(hipMemcpy(got.data(), dHits, n * 16, hipMemcpyDeviceToHost));
        MCHECK(mipt_trace_device(ctx, dRays, n, dRef));
        CHECK(hipStreamSynchronize(stream));
        CHECK(hipMemcpy(ref.data(), dRef, n * 16, hipMemcpyDeviceToHost));
        if (memcmp(got.data(), ref.data(), n * 16) != 0) { fprintf(stderr, "launch %d: hits differ from mipt_trace_device\n", rep); return 1; }
    }
    printf("capture ok\n");
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    mipt_context_destroy(ctx);
    mipt_scene_destroy(scene);
    return 0;
}
