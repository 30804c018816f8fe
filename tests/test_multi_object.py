"""Several objects per scene (Scene::addObject cs488.h:2020, one BVH per object :2027-2035, the loops of Scene::intersect
and Scene::isHit over them :2056-2085): mipt_scene_create_multi against the oracle's scene of several objects.
CPU part: the oracle's loop against a merge of its own single-object answers, and the host side of the library (object
bases, the hierarchy over all objects' leaves, refusals).  GPU part (-m gpu): hits, any-hit answers and images through
the C-ABI, bit for bit / within the integrators' tolerances, with every traversal the library has."""
import os

import numpy as np
import pytest

_cache = {}


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def room(sf, product):
    key = ("room", product)
    if key not in _cache:
        objs = sf.room_objects()
        env = sf.procedural_env(128)
        _cache[key] = sf.product_scene_from_objects(objs, env=env) if product else sf.oracle_scene_from_objects(objs, env=env)
    return _cache[key]


def rays_in_the_room(ob, orc, n=30000, seed=11):
    """primary rays of the default camera + incoherent rays from points inside the box, some axis-parallel or with a zero
    component (those take the exact walk kernels)"""
    rng = np.random.default_rng(seed)
    o0, d0 = orc.eye_rays(ob.make_camera(160, 120))
    o1 = rng.uniform(-0.35, 0.35, (n, 3)).astype(np.float32)
    d1 = rng.normal(size=(n, 3)).astype(np.float32)
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True).astype(np.float32)
    d1[:300] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 300)] * rng.choice([-1.0, 1.0], (300, 1)).astype(np.float32)
    d1[300:600, 1] = 0.0
    return np.concatenate([o0, o1]), np.concatenate([d0, d1])


def test_oracle_scene_of_several_objects_is_the_merge_of_its_objects(ob, sf):
    """Scene::intersect (cs488.h:2070-2085) restated: smallest t over the objects, strict '<', so an exact tie stays with
    the earlier object -- checked against single-object oracle scenes merged in numpy, for both of the reference's
    traversals; ids count through the objects."""
    objs = sf.room_objects()
    multi = room(sf, False)
    singles = [sf.oracle_scene_from_mesh_arrays(*a) for a in objs]
    o, d = rays_in_the_room(ob, multi, n=4000)
    base = np.cumsum([0] + [len(a[3]) for a in objs])
    for variant in (ob.THREADED, ob.BITTRAIL):
        tri = np.full(o.shape[0], -1, np.int32)
        t = np.full(o.shape[0], np.float32(ob.FLT_MAX), np.float32)
        for k, s in enumerate(singles):
            tk, tt, _ = s.intersect(o, d, variant=variant)
            win = (tk >= 0) & (tt < t)  # strict: the earlier object keeps a tie
            tri[win] = tk[win] + base[k]
            t[win] = tt[win]
        mt, mtt, _ = multi.intersect(o, d, variant=variant)
        assert np.array_equal(mt, tri) and np.array_equal(bits(mtt), bits(t))
        occ = multi.occluded(o, d, 0.0, 0.7, variant=variant)
        any_single = np.zeros(o.shape[0], np.uint8)
        for s in singles:
            any_single |= s.occluded(o, d, 0.0, 0.7, variant=variant)
        assert np.array_equal(occ, any_single)
    # the duplicated bunny (objects 1 and 3) never wins: every hit on it is an exact tie with object 1
    mt, _, _ = multi.intersect(o, d)
    assert ((mt >= base[1]) & (mt < base[2])).sum() > 100
    assert ((mt >= base[3]) & (mt < base[4])).sum() == 0


def test_host_side_of_a_scene_of_several_objects(B, sf):
    objs = sf.room_objects()
    scene = room(sf, True)
    assert scene.object_count == len(objs)
    base = np.cumsum([0] + [len(a[3]) for a in objs])
    mbase = np.cumsum([0] + [len(a[4]) for a in objs])
    for k in range(len(objs)):
        assert scene.object_tri_base(k) == base[k] and scene.object_material_base(k) == mbase[k]
    assert scene.object_tri_base(len(objs)) == -1
    assert scene.wide_nodes > 0, "the hierarchy over all objects' leaves was not built: every ray would take the literal walk"
    leaves, depth = scene.selfcheck_wide()
    assert leaves == sum(b.num_leaves for b in scene.bvhs) and depth > 0
    one = sf.product_scene_from_objects(objs[:1])
    assert one.object_count == 1 and one.object_tri_base(0) == 0 and one.object_tri_base(1) == -1


def test_cached_hierarchy_of_several_objects(B, sf, tmp_path):
    objs = sf.room_objects()[:3]
    path = os.path.join(tmp_path, "room.wide")
    a = sf.product_scene_from_objects(objs, wide_cache=path)
    assert os.path.getsize(path) > 0
    b = sf.product_scene_from_objects(objs, wide_cache=path)
    c = sf.product_scene_from_objects(objs)
    assert a.wide_hash == b.wide_hash == c.wide_hash
    # another order of the same objects is another scene (other leafTris order, other tie-breaks): the key must differ
    d = sf.product_scene_from_objects([objs[1], objs[0], objs[2]], wide_cache=path)
    assert d.wide_hash != a.wide_hash
    d.selfcheck_wide()


@pytest.mark.gpu
@pytest.mark.parametrize("traversal", ["default", "walk", "bittrail", "hand_back_all", "no_drain"])
def test_hits_in_a_scene_of_several_objects(B, ob, sf, traversal, monkeypatch):
    """ids, t, barycentrics, HitInfo and the any-hit answer against the oracle's loop over the objects, through the
    default traversal (one hierarchy over all objects' leaves), the literal walk (a forest: tree k's exits lead to root
    k + 1), the bit-trail kernel (object after object, as the reference) and the exact walk kernels on every ray"""
    scene, orc = room(sf, True), room(sf, False)
    if traversal == "hand_back_all":
        monkeypatch.setenv("MIPT_HAND_BACK_ALL", "1")
    if traversal == "no_drain":
        monkeypatch.setenv("MIPT_HAND_BACK_ALL", "1")
        monkeypatch.setenv("MIPT_DRAIN", "0")
    ctx = B.Context(scene, device=0)
    variant = ob.THREADED
    if traversal == "walk":
        ctx.set_traversal(B.TRAVERSAL_THREADED)
    elif traversal == "bittrail":
        ctx.set_traversal(B.TRAVERSAL_BITTRAIL)
        variant = ob.BITTRAIL
    o, d = rays_in_the_room(ob, orc)
    for tmin, tmax in ((0.0, B.FLT_MAX), (0.001, 0.6)):
        tri, t, bary, rec = ctx.trace(o, d, tmin, tmax, want_hitrec=True)
        otri, ot, obary, orec = orc.intersect(o, d, tmin, tmax, variant=variant, want_hitrec=True)
        assert np.array_equal(tri, otri)
        assert np.array_equal(bits(t), bits(ot)) and np.array_equal(bits(bary), bits(obary))
        assert np.array_equal(bits(rec), bits(orec)), "HitInfo (t, P, N, T, scene-wide material id) differs"
    tmax = np.random.default_rng(3).uniform(0.05, 1.5, o.shape[0]).astype(np.float32)
    assert np.array_equal(ctx.occluded(o, d, 0.0, tmax), orc.occluded(o, d, 0.0, tmax, variant=variant))
    # every object is hit somewhere except the duplicate, whose hits all tie with the earlier copy
    per_object = [int(((tri >= scene.object_tri_base(k)) & (tri < scene.object_tri_base(k) + m.num_triangles)).sum())
                  for k, m in enumerate(scene.meshes)]
    assert per_object[3] == 0 and min(per_object[0], per_object[1], per_object[2], per_object[4]) > 0, per_object


@pytest.mark.gpu
def test_images_of_a_scene_of_several_objects(B, ob, sf):
    """both integrators over several objects (metal box, Lambertian bunny, textured glass teapot, textured cube):
    materials and textures are looked up per object (cs488.h:772, the hit's material pointer)"""
    scene, orc = room(sf, True), room(sf, False)
    ctx = B.Context(scene, device=0)
    cam, ocam = B.default_camera(192, 144), ob.make_camera(192, 144)
    direct, ref = ctx.render(cam, mode=B.MODE_DIRECT), orc.raytrace(ocam)
    differing = int((bits(direct) != bits(ref)).any(axis=2).sum())
    assert differing <= 2, (differing, "direct-mode pixels differ from the oracle")
    img = ctx.render(cam, mode=B.MODE_PATH, spp=2, max_bounces=6, seed=9)
    refp = orc.pathtrace(ocam, spp=2, max_bounces=6, rng_mode=1, seed=9)
    mean = float(refp.astype(np.float64).mean())
    rmse = float(np.sqrt(np.mean((img.astype(np.float64) - refp) ** 2)))
    assert rmse <= 5e-3 * mean and abs(float(img.astype(np.float64).mean()) - mean) <= 1e-4 * mean, (rmse / mean, mean)
    # partitions of the image assemble the same frame (what the multi-GPU hosts rely on)
    parts = [ctx.render(cam, mode=B.MODE_PATH, spp=2, max_bounces=6, seed=9, part_index=k, part_count=3) for k in range(3)]
    assert np.array_equal(bits(sum(parts)), bits(img))


@pytest.mark.gpu
def test_headless_cli_adds_objects(B, tmp_path):
    """mipt_render --scene a.obj --add-scene b.obj: the frame equals the library's render of the two meshes as one scene of
    two objects, and differs from the frame of the first object alone"""
    import subprocess
    a, b = tmp_path / "floor.obj", tmp_path / "blocker.obj"
    a.write_text("v -1 -0.5 -1\nv 1 -0.5 -1\nv 1 -0.5 1\nv -1 -0.5 1\nvn 0 1 0\nf 1//1 3//1 2//1\nf 1//1 4//1 3//1\n")
    b.write_text("v -0.3 0.0 -0.3\nv 0.3 0.0 -0.3\nv 0.0 0.3 0.3\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    exe = os.path.join(os.path.dirname(B.LIB_PATH), "mipt_render")
    pfm = str(tmp_path / "out.pfm")
    run = subprocess.run([exe, "--scene", str(a), "--add-scene", str(b), "--width", "120", "--height", "90", "--mode", "direct",
                          "--out-pfm", pfm], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    assert "Added" in run.stdout
    with open(pfm, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(v) for v in f.readline().split())
        scale = float(f.readline())
        img = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4").reshape(h, w, 3).astype(np.float32)
    cam = B.default_camera(120, 90)
    both = B.Context(B.Scene([B.Mesh.load(str(a)), B.Mesh.load(str(b))]), device=0).render(cam, mode=B.MODE_DIRECT)
    alone = B.Context(B.Scene(B.Mesh.load(str(a))), device=0).render(cam, mode=B.MODE_DIRECT)
    assert np.array_equal(bits(img), bits(both))
    assert not np.array_equal(bits(img), bits(alone))


def _check_against_room_golden(g, label, tri, t, bary, occ, ob):
    hit = tri >= 0
    assert int(hit.sum()) == int(g[f"{label}_hits"])
    assert ob.fnv1a32_of_bits(bits(t[hit])) == int(g[f"{label}_fnv_t"])
    assert ob.fnv1a32_of_bits(bits(tri)) == int(g[f"{label}_fnv_tri"])
    assert ob.fnv1a32_of_bits(bits(bary[hit]).reshape(-1)) == int(g[f"{label}_fnv_bary"])
    s = int(g["stride"])
    assert np.array_equal(tri[::s], g[f"{label}_sub_tri"]) and np.array_equal(bits(t[::s]), bits(g[f"{label}_sub_t"]))
    assert np.array_equal(bits(bary[::s]), bits(g[f"{label}_sub_bary"]))
    base = g["tri_base"]
    per_object = [int(((tri >= base[k]) & (tri < base[k + 1])).sum()) for k in range(len(base) - 1)]
    assert per_object == g[f"{label}_hits_per_object"].tolist() and per_object[3] == 0
    assert int(occ.sum()) == int(g[f"{label}_occluded"]) and ob.fnv1a32_of_bits(occ.astype(np.uint32)) == int(g[f"{label}_fnv_occ"])


def test_oracle_against_the_committed_records_of_the_room(ob, sf):
    """tests/golden/golden_room.npz (make_room_golden.py): the oracle's scene of several objects keeps returning what it
    returned when the fixture was made, for both of the reference's traversals"""
    g = np.load(os.path.join(sf.GOLDEN, "golden_room.npz"))
    orc = room(sf, False)
    o, d = orc.eye_rays(ob.make_camera(int(g["width"]), int(g["height"])))
    for label, variant in (("threaded", ob.THREADED), ("bittrail", ob.BITTRAIL)):
        tri, t, bary = orc.intersect(o, d, variant=variant)
        _check_against_room_golden(g, label, tri, t, bary, orc.occluded(o, d, 0.0, 1.4, variant=variant), ob)


@pytest.mark.gpu
def test_product_against_the_committed_records_of_the_room(B, ob, sf):
    """the same committed records through the C-ABI: the default traversal and the literal walk against the threaded
    records, the bit-trail kernel against the bit-trail records"""
    g = np.load(os.path.join(sf.GOLDEN, "golden_room.npz"))
    ctx = B.Context(room(sf, True), device=0)
    o, d = ctx.eye_rays(B.default_camera(int(g["width"]), int(g["height"])))
    for label, mode in (("threaded", B.TRAVERSAL_DEFAULT), ("threaded", B.TRAVERSAL_THREADED), ("bittrail", B.TRAVERSAL_BITTRAIL)):
        ctx.set_traversal(mode)
        tri, t, bary = ctx.trace(o, d)
        _check_against_room_golden(g, label, tri, t, bary, ctx.occluded(o, d, 0.0, 1.4), ob)
