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
def probe(name, w, h, mode, spp, bounce_list):
    arrays = sf.sponza_standin(1) if name == "sponza" else sf.load_scene_arrays(name)
    scene = sf.product_scene_from_mesh_arrays(*arrays, env=env, lights=sf.SPONZA_LIGHT if name == "sponza" else sf.DEFAULT_LIGHT)
    ctx = B.Context(scene, device=0)
    cam = B.make_camera(w, h, sf.SPONZA_EYE, sf.SPONZA_LOOKAT) if name == "sponza" else B.default_camera(w, h)
    for bounces in bounce_list:
        imgs = [ctx.render(cam, mode=mode, spp=spp, max_bounces=bounces, seed=1).copy() for _ in range(3)]
        hs = [hashlib.sha256(np.ascontiguousarray(i).tobytes()).hexdigest()[:12] for i in imgs]
        nd = int((imgs[0].view(np.uint32) != imgs[1].view(np.uint32)).any(axis=-1).sum())
        print(name, "mode", mode, "bounces", bounces, hs, "pixels differing between run 0 and 1:", nd, flush=True)


probe("cornellbox", 256, 256, B.MODE_PATH, 8, (4, 8, 12, 32, -1))
probe("testObj", 256, 192, B.MODE_PATH, 4, (4, 12, -1))
probe("teapot-glass", 256, 192, B.MODE_PATH, 4, (8, 20))
probe("sponza", 320, 180, B.MODE_PATH, 8, (4, 12))
probe("cornellbox-metal", 256, 192, B.MODE_PATH, 4, (12,))
probe("cornellbox-metal", 256, 192, B.MODE_DIRECT, 1, (6,))
probe("bunny", 256, 192, B.MODE_DIRECT, 1, (6,))
