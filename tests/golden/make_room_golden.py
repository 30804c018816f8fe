"""Generates tests/golden/golden_room.npz: closest-hit records of the CPU oracle for the scene of SEVERAL OBJECTS of
tests/scene_fixtures.room_objects() (Scene::addObject, cs488.h:2020; the loops of Scene::intersect / isHit,
cs488.h:2056-2085), default camera at 512x384, for both of the reference's traversals: FNV-1a hashes over scene-wide
triangle ids / t bits / barycentric bits, hit count, hits per object, every 7th record in full, and any-hit answers for a
fixed tMax.  Built from the committed scene_*.npz fixtures only (runs anywhere the oracle builds); data, no source text."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle_binding as ob  # noqa: E402
import scene_fixtures as sf  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    objs = sf.room_objects()
    s = sf.oracle_scene_from_objects(objs)
    cam = ob.make_camera(512, 384)
    o, d = s.eye_rays(cam)
    base = np.cumsum([0] + [len(a[3]) for a in objs]).astype(np.int32)
    g = dict(width=512, height=384, stride=7, tri_base=base)
    for label, variant in (("threaded", ob.THREADED), ("bittrail", ob.BITTRAIL)):
        tri, t, bary = s.intersect(o, d, variant=variant)
        hit = tri >= 0
        g[f"{label}_hits"] = int(hit.sum())
        g[f"{label}_fnv_t"] = ob.fnv1a32_of_bits(t[hit].view(np.uint32))
        g[f"{label}_fnv_tri"] = ob.fnv1a32_of_bits(tri.view(np.uint32))
        g[f"{label}_fnv_bary"] = ob.fnv1a32_of_bits(bary[hit].reshape(-1).view(np.uint32))
        g[f"{label}_hits_per_object"] = np.array([int(((tri >= base[k]) & (tri < base[k + 1])).sum()) for k in range(len(objs))], np.int64)
        g[f"{label}_sub_tri"], g[f"{label}_sub_t"], g[f"{label}_sub_bary"] = tri[::7], t[::7], bary[::7]
        occ = s.occluded(o, d, 0.0, 1.4, variant=variant)
        g[f"{label}_occluded"] = int(occ.sum())
        g[f"{label}_fnv_occ"] = ob.fnv1a32_of_bits(occ.astype(np.uint32))
        print(label, "hits", g[f"{label}_hits"], "per object", g[f"{label}_hits_per_object"], "fnv_t %08x" % g[f"{label}_fnv_t"],
              "occluded", g[f"{label}_occluded"])
    np.savez_compressed(os.path.join(OUT, "golden_room.npz"), **g)
    print(os.path.getsize(os.path.join(OUT, "golden_room.npz")), "bytes")


if __name__ == "__main__":
    main()
