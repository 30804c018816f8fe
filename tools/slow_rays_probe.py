#!/usr/bin/env python3
"""How long the exact walk takes for rays the order-free traversal hands back: n rays with a zero direction component (they
never qualify) through mipt_trace, in-kernel drain (default) against MIPT_DRAIN=0 (k_trace_slow up to 4,096 rays, list-mode
k_trace beyond).  python tools/slow_rays_probe.py [n ...]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import scene_fixtures as sf
B = importlib.import_module("path-tracer_amd").binding
arrays = sf.sponza_standin(1)
scene = sf.product_scene_from_mesh_arrays(*arrays, lights=sf.SPONZA_LIGHT)
ctx = B.Context(scene, device=0)
pos = np.asarray(arrays[0], np.float32).reshape(-1, 3); lo, hi = pos.min(0), pos.max(0)
rng = np.random.default_rng(1)
for n in [int(a) for a in sys.argv[1:]] or [16, 256, 1500, 4096, 22000, 200000]:
    o = (lo + (hi - lo) * rng.uniform(0.05, 0.95, (n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    d[:, 0] = 0.0
    ctx.trace(o, d); ctx.reset_stats(); ctx.set_profiling(True)
    for _ in range(3): ctx.trace(o, d)
    st = ctx.stats(); ctx.set_profiling(False)
    print(f"{n:8d} handed-back rays: {st['ms_closest'] / st['launches_closest'] * 1e3:10.1f} us per launch", flush=True)
