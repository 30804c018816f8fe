#!/usr/bin/env python3
"""Kernel time of one closest-hit / any-hit launch against its ray count (incoherent rays inside the sponza
stand-in): the intercept of the line is what a launch costs beyond its rays (ramp-up, the tail while the last
wavefronts drain)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import scene_fixtures as sf
B = importlib.import_module("path-tracer_amd").binding
arrays = sf.sponza_standin(1)
scene = sf.product_scene_from_mesh_arrays(*arrays, lights=sf.SPONZA_LIGHT)
ctx = B.Context(scene, device=0)
pos = np.asarray(arrays[0], np.float32).reshape(-1, 3)
lo, hi = pos.min(0), pos.max(0)
rng = np.random.default_rng(1)
nmax = 1 << 25
o = (lo + (hi - lo) * rng.uniform(0.05, 0.95, (nmax, 3))).astype(np.float32)
d = rng.normal(size=(nmax, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
for n in (1 << 21, 1 << 22, 1 << 23, 1 << 24, 1 << 25):
    for name, fn, key, lk in (("closest", lambda: ctx.trace(o[:n], d[:n]), "ms_closest", "launches_closest"),
                              ("any-hit", lambda: ctx.occluded(o[:n], d[:n], 0.0, 3.0), "ms_occluded", "launches_occluded")):
        fn(); ctx.reset_stats(); ctx.set_profiling(True)
        for _ in range(3): fn()
        st = ctx.stats(); ctx.set_profiling(False)
        ms = st[key] / st[lk]
        print(f"{name:8s} {n:9d} rays  {ms:7.3f} ms  {n / ms / 1e3:7.1f} Mrays/s", flush=True)
