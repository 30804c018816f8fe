#!/usr/bin/env python3
"""bench frames rendered with the bounces overlapped on two streams and serialised: ray counts and frame hashes must agree"""
import hashlib, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scene_fixtures as sf
B = importlib.import_module("path-tracer_amd").binding
W, H = 1920, 1080
scene = sf.product_scene_from_mesh_arrays(*sf.sponza_standin(1), env=sf.procedural_env(1024), lights=sf.SPONZA_LIGHT)
dev = torch.device("cuda", 0)
ctx = B.Context(scene, device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
cam = B.make_camera(W, H, sf.SPONZA_EYE, sf.SPONZA_LOOKAT)
for mode in (1, 0, 1, 0):
    ctx.set_overlap(mode)
    accum = torch.zeros(H * W * 3, dtype=torch.float32, device=dev); rgb = torch.zeros_like(accum)
    ctx.reset_stats()
    for k in range(2):
        ctx.render_device(cam, B.RenderParams(B.MODE_PATH, 16, 16 * k, 4, 1, 0, 1, 64, 8), accum.data_ptr(), rgb.data_ptr())
    torch.cuda.synchronize()
    st = ctx.stats()
    print("overlap", mode, st["rays_closest"], st["rays_occluded"], hashlib.sha1(rgb.cpu().numpy().tobytes()).hexdigest()[:12], flush=True)
