#!/usr/bin/env python3
"""Would two half-shares in flight hide the launch tails?  Rank r of N owns the tiles with (tx + 3 ty) mod N == r, which
are exactly parts r and r + N of a 2N-way split, so the same pixels can be rendered as two sub-partitions by two
contexts on two streams (same accumulation buffers: the pixel sets are disjoint, every pixel keeps its sample order, the
frame stays bit-identical).  Prints ms per 16-spp step for one context rendering part r / N against two contexts
rendering parts r and r + N of 2N concurrently, for N = 1, 2, 4, 8."""
import hashlib, importlib, os, sys, time
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
accum = torch.zeros(H * W * 3, device=dev)
rgb = torch.zeros(H * W * 3, device=dev)


def run(n, split, steps=5, r=0):
    accum.zero_(); rgb.zero_()
    torch.cuda.synchronize()
    t0 = None
    for k in range(steps + 2):
        if k == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        if split:
            for i in range(2):
                ctxs[i].render_device(cam, B.RenderParams(B.MODE_PATH, 16, k * 16, 4, 1, r + i * n, 2 * n, 64, 8), accum.data_ptr(), rgb.data_ptr())
        else:
            ctxs[0].render_device(cam, B.RenderParams(B.MODE_PATH, 16, k * 16, 4, 1, r, n, 64, 8), accum.data_ptr(), rgb.data_ptr())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return ms, hashlib.sha256(rgb.cpu().numpy().tobytes()).hexdigest()[:12]


for n in (1, 2, 4, 8):
    a, ha = run(n, False)
    b, hb = run(n, True)
    print(f"share 1/{n}: one context {a:.2f} ms per step, two half-shares on two streams {b:.2f} ms per step; frames {'identical' if ha == hb else 'DIFFER'}", flush=True)
