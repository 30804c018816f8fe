#!/usr/bin/env python3
"""Would two half-frames in flight hide the shading kernels (HBM-bound) behind the traversal kernels (issue / latency-
bound)?  Zero-code-change experiment: two contexts on one GPU, each with its own stream and path memory, each rendering
half of the 16 samples of the bench frame, enqueued back to back; against one context rendering all 16."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import scene_fixtures as sf
B = importlib.import_module("path-tracer_amd").binding
W, H = 1920, 1080
scene = sf.product_scene_from_mesh_arrays(*sf.sponza_standin(1), env=sf.procedural_env(1024), lights=sf.SPONZA_LIGHT)
dev = torch.device("cuda", 0)
cam = B.make_camera(W, H, sf.SPONZA_EYE, sf.SPONZA_LOOKAT)
streams = [torch.cuda.Stream(dev) for _ in range(2)]
ctxs = [B.Context(scene, device=0, stream=s.cuda_stream) for s in streams]
bufs = [(torch.zeros(H * W * 3, device=dev), torch.zeros(H * W * 3, device=dev)) for _ in range(2)]
def run(parts, steps=5):
    t0 = None
    for k in range(steps + 1):
        if k == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(parts):
            spp = 16 // parts
            ctxs[i].render_device(cam, B.RenderParams(B.MODE_PATH, spp, k * 16 + i * spp, 4, 1, 0, 1, 64, 8), bufs[i][0].data_ptr(), bufs[i][1].data_ptr())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for overlap in (True, False):
    for c in ctxs: c.set_overlap(overlap)
    print(f"side stream {overlap}: one context, 16 spp: {run(1):.2f} ms per step; two contexts, 8 spp each, concurrent: {run(2):.2f} ms per step", flush=True)
