import sys, numpy as np, importlib
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import scene_fixtures as sf
B = importlib.import_module('path-tracer_amd').binding
name = sys.argv[1] if len(sys.argv) > 1 else 'sponza'
arrays = sf.sponza_standin(1) if name == 'sponza' else sf.load_scene_arrays(name)
scene = sf.product_scene_from_mesh_arrays(*arrays, lights=sf.SPONZA_LIGHT)
ctx = B.Context(scene, device=0)
pos = np.asarray(arrays[0], np.float32).reshape(-1, 3); lo, hi = pos.min(0), pos.max(0)
rng = np.random.default_rng(1)
for n in (64, 4096, 65536, 262144, 1048576):
    o = (lo + (hi - lo) * rng.uniform(0.05, 0.95, (n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    for name, fn, key, lk in (("closest", lambda: ctx.trace(o, d), "ms_closest", "launches_closest"), ("any-hit", lambda: ctx.occluded(o, d, 0.0, 3.0), "ms_occluded", "launches_occluded")):
        fn(); ctx.reset_stats(); ctx.set_profiling(True)
        for _ in range(5): fn()
        st = ctx.stats(); ctx.set_profiling(False)
        print(f"{name:8s} {n:8d} rays {st[key]/st[lk]*1e3:8.1f} us", flush=True)
