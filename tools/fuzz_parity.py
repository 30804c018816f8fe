#!/usr/bin/env python3
"""Parity fuzz outside the test-suite: per scene, a few million rays of several adversarial families (from
vertices and edge midpoints, grazing along triangle planes, axis-parallel, tiny and huge direction components,
origins on box planes) through the three GPU traversals, bit for bit (hit triangle, t, barycentrics, any-hit answer): the default and
the threaded traversal against the oracle's threaded walk, the bit-trail kernel against the oracle's bit-trail
variant.  The reference's two traversals are NOT the same function on such rays (near-first order picks another
triangle of an exact tie in t, and culls): the last columns count where the two oracles differ."""
import importlib, os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import scene_fixtures as sf
import oracle_binding as ob
B = importlib.import_module("path-tracer_amd").binding
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
total_bad = 0
for name in ("cornellbox", "teapot-glass", "testObj", "bunny", "sponza", "room (5 objects)"):
    if name.startswith("room"):  # several objects per scene (mipt_scene_create_multi), one of them twice: exact ties between objects
        objs = sf.room_objects()
        scene = sf.product_scene_from_objects(objs)
        orc = sf.oracle_scene_from_objects(objs)
        tri = np.concatenate([np.asarray(a[0], np.float32).reshape(-1, 3, 3) for a in objs])
    else:
        arrays = sf.sponza_standin(1) if name == "sponza" else sf.load_scene_arrays(name)
        scene = sf.product_scene_from_mesh_arrays(*arrays)
        orc = sf.oracle_scene_from_mesh_arrays(*arrays)
        tri = np.asarray(arrays[0], np.float32).reshape(-1, 3, 3)
    ctx = B.Context(scene, device=0)
    lo, hi = tri.reshape(-1, 3).min(0), tri.reshape(-1, 3).max(0)
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000)  # the same rays in every run
    def unit(v):
        return (v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-30)).astype(np.float32)
    t_idx = rng.integers(0, len(tri), N)
    inside = (lo + (hi - lo) * rng.uniform(0.02, 0.98, (N, 3))).astype(np.float32)
    w = rng.dirichlet((1, 1, 1), N).astype(np.float32)
    on_tri = (tri[t_idx] * w[:, :, None]).sum(1).astype(np.float32)
    e1, e2 = tri[t_idx, 1] - tri[t_idx, 0], tri[t_idx, 2] - tri[t_idx, 0]
    axes = np.eye(3, dtype=np.float32)
    fam = {
        "towards points on triangles": (inside, unit(on_tri - inside)),
        "from vertices, random": (tri[t_idx, rng.integers(0, 3, N)].copy(), unit(rng.normal(size=(N, 3)))),
        "from edge midpoints towards vertices": (((tri[t_idx, 0] + tri[t_idx, 1]) * np.float32(0.5)).astype(np.float32), unit(tri[rng.integers(0, len(tri), N), 2] - (tri[t_idx, 0] + tri[t_idx, 1]) * np.float32(0.5))),
        "grazing in triangle planes": (on_tri - unit(e1 * rng.normal(size=(N, 1)) + e2 * rng.normal(size=(N, 1))) * np.float32(0.5), unit(e1 * rng.normal(size=(N, 1)) + e2 * rng.normal(size=(N, 1)))),
        "axis-parallel from vertices": (tri[t_idx, 0].copy(), (axes[rng.integers(0, 3, N)] * rng.choice(np.array([-1, 1], np.float32), N)[:, None]).astype(np.float32)),
        "one tiny / huge direction component": (inside, None),
    }
    d = unit(rng.normal(size=(N, 3)))
    d[np.arange(N), rng.integers(0, 3, N)] = rng.choice(np.array([1e-38, 1e-30, 1e-12, 2.0 ** -40, 0.0, -0.0, 1e12, 3e38, -1e-20], np.float32), N)
    fam["one tiny / huge direction component"] = (inside, d)
    for label, (o, dd) in fam.items():
        o = np.ascontiguousarray(o, np.float32); dd = np.ascontiguousarray(dd, np.float32)
        t0 = time.time()
        ref = orc.intersect(o, dd)
        far = float(np.linalg.norm(hi - lo)) * 0.3
        ref_occ = orc.occluded(o, dd, 1e-4, far)
        ref_bt = orc.intersect(o, dd, variant=ob.BITTRAIL)  # the reference's OTHER traversal: its own order and culling
        ref_bt_occ = orc.occluded(o, dd, 1e-4, far, variant=ob.BITTRAIL)
        two_refs_t = int((bits(ref[1]) != bits(ref_bt[1])).sum())
        two_refs_id = int((ref[0] != ref_bt[0]).sum())
        bad = {}
        for mode, mname in ((B.TRAVERSAL_DEFAULT, "wide"), (B.TRAVERSAL_THREADED, "walk"), (B.TRAVERSAL_BITTRAIL, "bittrail")):
            try:
                ctx.set_traversal(mode)
            except B.MiptError:
                continue
            got = ctx.trace(o, dd)
            want, want_occ = (ref_bt, ref_bt_occ) if mode == B.TRAVERSAL_BITTRAIL else (ref, ref_occ)
            nb = int((~((got[0] == want[0]) & (bits(got[1]) == bits(want[1])) & (bits(got[2]) == bits(want[2])).all(axis=1))).sum())
            nb += int((ctx.occluded(o, dd, 1e-4, far) != want_occ).sum())
            bad[mname] = nb
        ctx.set_traversal(B.TRAVERSAL_DEFAULT)
        total_bad += sum(bad.values())
        print(f"{name:16s} {label:40s} {N} rays  hits {int((ref[0] >= 0).sum()):8d}  mismatches {bad}  oracle threaded vs bit-trail: t differs {two_refs_t}, id differs {two_refs_id}  ({time.time() - t0:.1f} s)", flush=True)
print("TOTAL MISMATCHES", total_bad, flush=True)
