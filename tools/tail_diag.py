#!/usr/bin/env python3
"""Where a closest-hit launch spends its end (a -DPT_DIAG_TAIL=1 variant, MIPT_LIB_PATH): wall-clock stamps of every
wavefront's start, of the moment it finds the queue empty, and of its end.  Prints, per launch size: launch span, the time
from the FIRST wavefront finding the queue empty to the last wavefront's end (the tail), the span over which wavefronts
end, the mean drain time of a wavefront (queue empty -> its end) and the number of hand-overs (PT_WIDE_SHARE)."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import scene_fixtures as sf
B = importlib.import_module("path-tracer_amd").binding
arrays = sf.sponza_standin(1)
scene = sf.product_scene_from_mesh_arrays(*arrays, lights=sf.SPONZA_LIGHT)
ctx = B.Context(scene, device=0)
L = B.lib()
pos = np.asarray(arrays[0], np.float32).reshape(-1, 3); lo, hi = pos.min(0), pos.max(0)
rng = np.random.default_rng(1)
buf = (C.c_ulonglong * (2 * 4 * 4096))()
for n in (64, 65536, 262144, 1 << 20, 1 << 22, 1 << 24):
    o = (lo + (hi - lo) * rng.uniform(0.05, 0.95, (n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    ctx.trace(o, d)
    for rep in range(2):
        L.mipt_debug_tail(None, 1)
        ctx.trace(o, d)
        L.mipt_debug_tail(buf, 0)
        a = np.frombuffer(buf, np.uint64).reshape(2, 4096, 4)[0].astype(np.float64)
        a = a[a[:, 2] > 0]
        start, empty, end, gifts = a[:, 0] / 100.0, a[:, 1] / 100.0, a[:, 2] / 100.0, a[:, 3]  # 100 MHz -> microseconds
        t0 = start.min()
        pct = lambda x, q: float(np.percentile(x, q))
        print(f"{n:9d} rays, {len(a)} waves: span {end.max() - t0:8.1f} us | queue-empty at {empty.min() - t0:7.1f} .. {empty.max() - t0:7.1f} | "
              f"ends at p10 {pct(end, 10) - t0:7.1f} p50 {pct(end, 50) - t0:7.1f} p90 {pct(end, 90) - t0:7.1f} p99 {pct(end, 99) - t0:7.1f} max {end.max() - t0:7.1f} | "
              f"drain mean {np.mean(end - empty):6.1f} max {np.max(end - empty):6.1f} | hand-overs {int(gifts.sum())}", flush=True)
