#!/usr/bin/env python3
"""How much would ray ordering buy the traversal kernels?  Secondary-like rays (origins on surfaces of the sponza
stand-in, cosine-ish random directions) traced in random order, binned by direction octant + a coarse origin grid
(what a cheap binning at queue-append time could give), and fully sorted by octant + Morton code of the origin."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import scene_fixtures as sf
B = importlib.import_module("path-tracer_amd").binding
arrays = sf.sponza_standin(1)
scene = sf.product_scene_from_mesh_arrays(*arrays, lights=sf.SPONZA_LIGHT)
ctx = B.Context(scene, device=0)
tri = np.asarray(arrays[0], np.float32).reshape(-1, 3, 3)
lo, hi = tri.reshape(-1, 3).min(0), tri.reshape(-1, 3).max(0)
rng = np.random.default_rng(1)
n = 1 << 24
# primary-like hits: points on random triangles (area-weighted), pushed off the surface a little
e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
area = np.linalg.norm(np.cross(e1, e2), axis=1)
t_idx = rng.choice(len(tri), n, p=area / area.sum())
w = rng.dirichlet((1, 1, 1), n).astype(np.float32)
o = (tri[t_idx] * w[:, :, None]).sum(1).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
o = (o + d * np.float32(1e-3)).astype(np.float32)

def timed(o, d, label):
    for name, fn, key, lk in (("closest", lambda: ctx.trace(o, d), "ms_closest", "launches_closest"),
                              ("any-hit", lambda: ctx.occluded(o, d, 0.0, 3.0), "ms_occluded", "launches_occluded")):
        fn(); ctx.reset_stats(); ctx.set_profiling(True)
        for _ in range(3): fn()
        st = ctx.stats(); ctx.set_profiling(False)
        print(f"{label:34s} {name:8s} {st[key] / st[lk]:7.3f} ms per {n} rays", flush=True)

timed(o, d, "random order")
octant = (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)
cell = np.clip(((o - lo) / (hi - lo) * 2).astype(np.int64), 0, 1)
coarse = octant * 8 + cell[:, 0] + 2 * cell[:, 1] + 4 * cell[:, 2]
k = np.argsort(coarse, kind="stable")
timed(o[k], d[k], "64 bins: octant x 2x2x2 cells")
q = np.clip(((o - lo) / (hi - lo) * 1024).astype(np.int64), 0, 1023)
def spread(v):
    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
    return v
morton = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
k = np.argsort(octant * (1 << 30) + morton, kind="stable")
timed(o[k], d[k], "sorted: octant, Morton(origin)")

# the same sorted rays, but the 64-ray blocks (what one wavefront takes at a time) in random order: coherent inside
# a wavefront, while the wavefronts that run at the same time work in unrelated parts of the scene
os_, ds_ = o[k], d[k]
blocks = rng.permutation(n // 64)
idx = (blocks[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)
timed(os_[idx], ds_[idx], "sorted inside shuffled 64-ray blocks")
blocks = rng.permutation(n // 1024)
idx = (blocks[:, None] * 1024 + np.arange(1024)[None, :]).reshape(-1)
timed(os_[idx], ds_[idx], "sorted inside shuffled 1024-ray blocks")

# primary rays of the bench camera (8 frames' worth), in pixel order, in shuffled 64-ray blocks, fully shuffled
cam = B.make_camera(1920, 1080, sf.SPONZA_EYE, sf.SPONZA_LOOKAT)
po, pd = ctx.eye_rays(cam)
po, pd = np.tile(po, (8, 1)), np.tile(pd, (8, 1))
n = len(po)
timed(po, pd, "primary rays, pixel order")
blocks = rng.permutation(n // 64)
idx = (blocks[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)
timed(po[idx], pd[idx], "primary rays, shuffled 64-ray blocks")
idx = rng.permutation(n)
timed(po[idx], pd[idx], "primary rays, fully shuffled")
