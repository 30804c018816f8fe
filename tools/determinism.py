#!/usr/bin/env python3
"""renders the same frame several times and compares bit patterns (a wavefront integrator must not depend on the
order in which atomics complete)"""
import hashlib, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import scene_fixtures as sf
B = importlib.import_module("path-tracer_amd").binding
env = sf.procedural_env(256)
scene = sf.product_scene_from_mesh_arrays(*sf.load_scene_arrays("cornellbox"), env=env)
ctx = B.Context(scene, device=0)
cam = B.default_camera(256, 256)
for bounces in (4, 8, 12, 16, 32, 60, -1):
    hs = []
    imgs = []
    for rep in range(3):
        img = ctx.render(cam, mode=B.MODE_PATH, spp=8, max_bounces=bounces, seed=1)
        imgs.append(img.copy())
        hs.append(hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest()[:12])
    nd = int((imgs[0].view(np.uint32) != imgs[1].view(np.uint32)).any(axis=-1).sum())
    print("bounces", bounces, hs, "pixels differing between run 0 and 1:", nd, flush=True)
    if nd:
        ys, xs = np.nonzero((imgs[0].view(np.uint32) != imgs[1].view(np.uint32)).any(axis=-1))
        for y, x in list(zip(ys, xs))[:5]:
            print("   pixel", x, y, imgs[0][y, x], imgs[1][y, x], flush=True)
