import sys, numpy as np, importlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import scene_fixtures as sf
B = importlib.import_module('path-tracer_amd').binding
scene = sf.product_scene_from_mesh_arrays(*sf.sponza_standin(1), lights=sf.SPONZA_LIGHT)
ctx = B.Context(scene, device=0)
n = 1 << 24
rng = np.random.default_rng(0)
o = np.full((n,3), 100.0, np.float32) + rng.uniform(0,1,(n,3)).astype(np.float32)
d = np.full((n,3), 0.57735026, np.float32) + rng.uniform(0,0.01,(n,3)).astype(np.float32)
for name, fn in (("occluded", lambda: ctx.occluded(o, d, 0.0, 10.0)), ("closest", lambda: ctx.trace(o, d))):
    fn(); ctx.reset_stats(); ctx.set_profiling(True)
    for _ in range(3): fn()
    st = ctx.stats(); ctx.set_profiling(False)
    key = "ms_occluded" if name == "occluded" else "ms_closest"
    lk = "launches_occluded" if name == "occluded" else "launches_closest"
    print(name, "root-miss rays: %.3f ms per launch of %d rays" % (st[key]/st[lk], n), flush=True)
