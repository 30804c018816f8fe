"""GPU parity tests proper: the HIP traversal through the C-ABI against the
CPU oracle (bit-exact: triangle ids, t, barycentrics, HitInfo) on the same
rays, against the committed golden fixtures, and size-independent properties
at the BASELINE resolutions."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCENES = ["cornellbox", "cube", "textured", "teapot-glass", "testObj", "bunny"]

_cache = {}


def setup_scene(B, sf, name, env=None):
    key = (name, env is not None)
    if key not in _cache:
        arrays = sf.sponza_standin(1) if name == "sponza-standin-trace" else sf.load_scene_arrays(name)
        scene = sf.product_scene_from_mesh_arrays(*arrays, env=env)
        ctx = B.Context(scene, device=0)
        orc = sf.oracle_scene_from_mesh_arrays(*arrays, env=env)
        _cache[key] = (scene, ctx, orc)
    return _cache[key]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("name", SCENES)
def test_primary_hits_bit_exact_vs_oracle_and_golden(B, ob, sf, name):
    scene, ctx, orc = setup_scene(B, sf, name)
    g = sf.load_golden(name)
    w, h = int(g["width"]), int(g["height"])
    cam = B.default_camera(w, h)
    o, d = ctx.eye_rays(cam)
    oo, od = orc.eye_rays(ob.make_camera(w, h))
    assert np.array_equal(bits(o), bits(oo)) and np.array_equal(bits(d), bits(od)), "eye rays differ bitwise"
    tri, t, bary, rec = ctx.trace(o, d, want_hitrec=True)
    otri, ot, obary, orec = orc.intersect(oo, od, want_hitrec=True)
    assert np.array_equal(tri, otri)
    assert np.array_equal(bits(t), bits(ot))
    assert np.array_equal(bits(bary), bits(obary))
    assert np.array_equal(bits(rec), bits(orec)), "HitInfo (t,P,N,T,material) differs bitwise"
    hit = tri >= 0
    assert int(hit.sum()) == int(g["hits"])
    assert ob.fnv1a32_of_bits(bits(t[hit])) == int(g["fnv_t"])
    assert ob.fnv1a32_of_bits(bits(tri)) == int(g["fnv_tri"])
    assert ob.fnv1a32_of_bits(bits(bary[hit]).reshape(-1)) == int(g["fnv_bary"])
    s = int(g["stride"])
    assert np.array_equal(tri[::s], g["sub_tri"]) and np.array_equal(bits(t[::s]), bits(g["sub_t"]))


@pytest.mark.parametrize("name", ["cornellbox", "bunny", "testObj"])
def test_secondary_rays_and_windows(B, ob, sf, name):
    """incoherent rays from surface points, with tMin/tMax windows, plus degenerate directions"""
    scene, ctx, orc = setup_scene(B, sf, name)
    rng = np.random.default_rng(7)
    n = 40000
    cam = B.default_camera(256, 192)
    o, d = ctx.eye_rays(cam)
    tri, t, _ = ctx.trace(o, d)
    hit = np.flatnonzero(tri >= 0)
    pick = rng.choice(hit, n)
    P = (o[pick] + d[pick] * t[pick, None]).astype(np.float32)
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True).astype(np.float32)
    # axis-aligned and zero-component directions: divisions by zero in the slab test
    dirs[:200] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 200)] * rng.choice([-1.0, 1.0], (200, 1)).astype(np.float32)
    dirs[200:400, rng.integers(0, 3)] = 0.0
    for tmin, tmax in ((0.001, B.FLT_MAX), (0.01, 0.5), (0.0, B.FLT_MAX)):
        a = ctx.trace(P, dirs, tmin, tmax)
        b = orc.intersect(P, dirs, tmin, tmax)
        assert np.array_equal(a[0], b[0])
        assert np.array_equal(bits(a[1]), bits(b[1]))
        assert np.array_equal(bits(a[2]), bits(b[2]))
    tmax = rng.uniform(0.05, 3.0, n).astype(np.float32)
    occ = ctx.occluded(P, dirs, 0.0, tmax)
    assert np.array_equal(occ, orc.occluded(P, dirs, 0.0, tmax))
    # any-hit == "closest-hit found something" (Scene::isHit, cs488.h:2056-2068)
    assert np.array_equal(occ, orc.occluded(P, dirs, 0.0, tmax, variant=ob.BITTRAIL))


def test_edge_cases(B, ob, sf):
    scene, ctx, orc = setup_scene(B, sf, "cornellbox")
    # empty batch
    tri, t, bary = ctx.trace(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert tri.size == 0
    # NaN / inf / zero directions, origins inside and far outside
    o = np.array([[0, 0, 1.5], [0, 0, 1.5], [0, 0, 1.5], [1e30, 0, 0], [0, 0, 0], [0, 0, 0]], np.float32)
    d = np.array([[0, 0, 0], [np.nan, 0, -1], [np.inf, 0, -1], [-1, 0, 0], [0, 1, 0], [0, 0, -1]], np.float32)
    a = ctx.trace(o, d)
    b = orc.intersect(o, d)
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))


def test_single_triangle_known_answer(B, ob):
    """createSingleTriangle (cs488.h:949-970): ray down -z through the centroid hits at t = 1.5"""
    mesh = B.Mesh.single_triangle()
    scene = B.Scene(mesh)
    ctx = B.Context(scene)
    o = np.array([[0.0, -1.0 / 6.0, 1.5], [2.0, 2.0, 1.5]], np.float32)
    d = np.array([[0.0, 0.0, -1.0], [0.0, 0.0, -1.0]], np.float32)
    tri, t, bary, rec = ctx.trace(o, d, want_hitrec=True)
    assert tri.tolist() == [0, -1]
    assert t[0] == np.float32(1.5)
    assert np.allclose(rec[0, 4:7], [0, 0, 1])  # normal +z, facing the ray origin


@pytest.mark.parametrize("name,w,h", [("cornellbox", 512, 512), ("bunny", 1024, 1024), ("teapot-glass", 1920, 1080)])
def test_full_size_properties(B, ob, sf, name, w, h):
    """BASELINE resolutions (configs[0] is "bit-exact primary hit IDs" at 512x512): ids, t and barycentrics of the
    WHOLE frame bit-exact against the oracle (every ray, not a subsample: the oracle needs well under a second per
    frame); hits lie on their triangles; tracing again returns the same bits"""
    scene, ctx, orc = setup_scene(B, sf, name)
    cam = B.default_camera(w, h)
    o, d = ctx.eye_rays(cam)
    tri, t, bary = ctx.trace(o, d)
    assert tri.size == w * h
    otri, ot, obary = orc.intersect(o, d)
    assert np.array_equal(tri, otri) and np.array_equal(bits(t), bits(ot)) and np.array_equal(bits(bary), bits(obary))
    # barycentric validity and t window on every hit
    hit = tri >= 0
    assert np.all(t[hit] > 1e-5) and np.all(bary[hit] >= 0) and np.all(bary[hit].sum(axis=1) <= 1.0 + 1e-6)
    # idempotence: tracing the same rays again gives the same bits
    tri2, t2, _ = ctx.trace(o, d)
    assert np.array_equal(tri, tri2) and np.array_equal(bits(t), bits(t2))


def test_fast_division_is_ieee(B, sf):
    """the slab test's reciprocal + residual-correction division (traverse.hpp: div_by_recip) rounds exactly like
    IEEE `/` on 2^33 operand pairs inside its precondition range, incl. all-ones / single-bit mantissas and a = 0"""
    scene, ctx, orc = setup_scene(B, sf, "cube")
    for seed in (1, 2):
        assert ctx.selftest_fast_division(1 << 32, seed) == 0


def test_scene_outside_reciprocal_preconditions_takes_ieee_path(B, ob, sf):
    """one extra triangle with a vertex coordinate of 1e-25 puts box planes below 2^-61: the library must fall
    back to the literal IEEE slab test for every ray (exact_reciprocal False) and still match the oracle bit for
    bit; the same holds for the scene scaled by 2^43 (planes above 2^40)"""
    base = sf.load_scene_arrays("cornellbox")
    for variant in ("tiny-plane", "huge"):
        arrays = [np.array(a) if isinstance(a, np.ndarray) else a for a in base]
        cam = B.default_camera(96, 64)
        k = np.float32(1.0)
        if variant == "tiny-plane":
            extra = np.array([[1e-25, 0.5, 0.5], [0.1, 0.5, 0.5], [0.1, 0.6, 0.5]], np.float32)
            arrays[0] = np.concatenate([np.asarray(arrays[0], np.float32).reshape(-1, 3, 3), extra[None]]).reshape(-1)
            arrays[1] = np.concatenate([np.asarray(arrays[1], np.float32).reshape(-1, 3, 3),
                                        np.tile(np.array([0, 0, 1], np.float32), (1, 3, 1))]).reshape(-1)
            arrays[2] = np.concatenate([np.asarray(arrays[2], np.float32).reshape(-1, 3, 2),
                                        np.zeros((1, 3, 2), np.float32)]).reshape(-1)
            arrays[3] = np.concatenate([np.asarray(arrays[3], np.int32), np.zeros(1, np.int32)])
        else:
            k = np.float32(2.0 ** 43)
            arrays[0] = (np.asarray(arrays[0], np.float32) * k).astype(np.float32)
        scene = sf.product_scene_from_mesh_arrays(*arrays)
        assert not scene.exact_reciprocal, variant
        ctx = B.Context(scene, device=0)
        orc = sf.oracle_scene_from_mesh_arrays(*arrays)
        o, d = ctx.eye_rays(cam)
        o = (o * k).astype(np.float32)
        a = ctx.trace(o, d)
        b = orc.intersect(o, d)
        assert (b[0] >= 0).sum() > 1000, variant
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and \
            np.array_equal(bits(a[2]), bits(b[2])), variant
        # the bit-trail kernel divides the IEEE way and needs a binary tree, not tame planes
        ctx.set_traversal(B.TRAVERSAL_BITTRAIL)
        c = ctx.trace(o, d)
        bb = orc.intersect(o, d, variant=ob.BITTRAIL)
        assert np.array_equal(c[0], bb[0]) and np.array_equal(bits(c[1]), bits(bb[1])), variant


@pytest.mark.parametrize("n", [1 << 13, 1 << 16])
def test_rays_straddling_reciprocal_preconditions(B, ob, sf, n):
    """per-ray choice between the reciprocal and the IEEE slab division: directions and origins with components
    at, just inside and just outside the precondition bounds (2^-40 / 2^40 for d, 2^-61 / 2^40 for o), zeros,
    denormals -- all in one batch so that lanes of one wavefront mix both code paths.  The rays the wide traversal
    hands back go to k_trace_slow (a wavefront per ray) when they are few (n = 2^13: about 3,000) and to the literal
    k_trace (a lane per ray) when they are more than 4,096 (n = 2^16)"""
    scene, ctx, orc = setup_scene(B, sf, "sponza-standin-trace")
    assert scene.exact_reciprocal
    rng = np.random.default_rng(11)
    o = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32) * np.float32(0.5) + np.array([0, 1.0, 0], np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    mags_d = np.array([2.0 ** -41, 2.0 ** -40, 2.0 ** -39, 1e-30, 1e-42, 0.0, 2.0 ** 40, 2.0 ** 41], np.float32)
    mags_o = np.array([2.0 ** -62, 2.0 ** -61, 2.0 ** -60, 1e-40, 0.0, -(2.0 ** -61), 8.7e-18], np.float32)
    sel = rng.integers(0, 4, n)
    ax = rng.integers(0, 3, n)
    rows = np.arange(n)
    md = mags_d[rng.integers(0, len(mags_d), n)] * rng.choice(np.array([-1, 1], np.float32), n)
    mo = mags_o[rng.integers(0, len(mags_o), n)]
    d[rows[sel == 1], ax[sel == 1]] = md[sel == 1]
    o[rows[sel == 2], ax[sel == 2]] = mo[sel == 2]
    d[rows[sel == 3], ax[sel == 3]] = md[sel == 3]
    o[rows[sel == 3], (ax[sel == 3] + 1) % 3] = mo[sel == 3]
    a = ctx.trace(o, d)
    b = orc.intersect(o, d)
    assert (b[0] >= 0).sum() > n // 4
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(bits(a[2]), bits(b[2]))
    occ = ctx.occluded(o, d, 1e-4, 2.5)
    assert np.array_equal(occ, orc.occluded(o, d, 1e-4, 2.5))


@pytest.mark.parametrize("name", ["cornellbox", "sponza-standin-trace"])
def test_axis_parallel_rays_from_vertices(B, ob, sf, name):
    """rays along the axes that start AT mesh vertices: two direction components are exactly zero and the origin
    lies in box planes, so AABB::intersect (cs488.h:571-608) sees (p - o) / 0 = NaN and +-inf quotients -- a child box
    can pass where its parent fails.  These rays leave the order-free traversal and take k_trace_slow, which
    must follow the reference's parent-hit rule literally"""
    scene, ctx, orc = setup_scene(B, sf, name)
    pos = (sf.sponza_standin(1) if name.startswith("sponza") else sf.load_scene_arrays(name))[0]
    verts = np.asarray(pos, np.float32).reshape(-1, 3)
    rng = np.random.default_rng(5)
    n = 3000  # all of them are handed back: more than the launch drains itself (two entries per wavefront), so list-mode k_trace traces most of them
    o = verts[rng.integers(0, len(verts), n)].copy()
    axes = np.eye(3, dtype=np.float32)
    d = axes[rng.integers(0, 3, n)] * rng.choice(np.array([-1, 1], np.float32), n)[:, None]
    a = ctx.trace(o, d)
    b = orc.intersect(o, d)
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(bits(a[2]), bits(b[2]))
    for tmin, tmax in ((0.0, 3.0e38), (1e-4, 2.5)):
        assert np.array_equal(ctx.occluded(o, d, tmin, tmax), orc.occluded(o, d, tmin, tmax))


@pytest.mark.parametrize("n_axis,n_other", [(1, 0), (16, 0), (70, 0), (16, 5000), (300, 200000), (20000, 100000)])
def test_handed_back_rays_of_every_list_length(B, ob, sf, n_axis, n_other):
    """the rays k_trace_wide hands back are traced by their owner wavefronts inside the launch when the list is short,
    by list-mode k_trace after it when it is long or when an entry arrives after its owner has gone (a launch of a
    few rays: most wavefronts leave before the first one is handed back): every split of the work gives the oracle's
    bits, for lists of 1 to 20,000 entries alone and among ordinary rays"""
    scene, ctx, orc = setup_scene(B, sf, "sponza-standin-trace")
    pos = np.asarray(sf.sponza_standin(1)[0], np.float32).reshape(-1, 3)
    lo, hi = pos.min(0), pos.max(0)
    rng = np.random.default_rng(n_axis * 7 + n_other)
    n = n_axis + n_other
    o = (lo + (hi - lo) * rng.uniform(0.05, 0.95, (n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    which = rng.permutation(n)[:n_axis]
    d[which, rng.integers(0, 3, n_axis)] = 0.0  # a zero component: never qualifies
    for _ in range(2):  # twice: the list headers are used in turn
        a = ctx.trace(o, d)
        b = orc.intersect(o, d)
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(bits(a[2]), bits(b[2]))
        assert np.array_equal(ctx.occluded(o, d, 1e-4, 2.5), orc.occluded(o, d, 1e-4, 2.5))


@pytest.mark.parametrize("name", ["cornellbox", "teapot-glass", "testObj"])
def test_literal_walk_and_wide_traversal_agree(B, ob, sf, name, monkeypatch):
    """MIPT_TRAVERSAL=walk forces the literal threaded walk (k_trace) for a context; both it and the default
    order-free traversal over the 4-wide tree (k_trace_wide) must reproduce the oracle bit for bit, on primary
    rays and on incoherent rays from random points (the wide kernel's stack and leaf groups in all states)"""
    arrays = sf.load_scene_arrays(name)
    scene = sf.product_scene_from_mesh_arrays(*arrays)
    assert scene.wide_nodes > 0
    orc = sf.oracle_scene_from_mesh_arrays(*arrays)
    wide_ctx = B.Context(scene, device=0)
    monkeypatch.setenv("MIPT_TRAVERSAL", "walk")
    walk_ctx = B.Context(scene, device=0)
    monkeypatch.delenv("MIPT_TRAVERSAL")
    cam = B.default_camera(160, 120)
    o, d = wide_ctx.eye_rays(cam)
    rng = np.random.default_rng(3)
    o2 = rng.uniform(-1.2, 1.2, (30000, 3)).astype(np.float32)
    d2 = rng.normal(size=(30000, 3)).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True).astype(np.float32)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    ref = orc.intersect(o, d)
    for ctx in (wide_ctx, walk_ctx):
        got = ctx.trace(o, d)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(bits(got[1]), bits(ref[1])) and \
            np.array_equal(bits(got[2]), bits(ref[2]))
        assert np.array_equal(ctx.occluded(o, d, 1e-4, 1.5), orc.occluded(o, d, 1e-4, 1.5))
    # the two kernels do different amounts of work for the same answer: the wide one tests the triangles of every
    # CANDIDATE leaf (conservative test) and asks the exact box predicate only for triangles that would change the
    # result, so it tests a few more triangles than the reference and far fewer boxes exactly
    wide_ctx.reset_stats(); wide_ctx.set_counting(True); wide_ctx.trace(o, d); a = wide_ctx.stats()
    walk_ctx.reset_stats(); walk_ctx.set_counting(True); walk_ctx.trace(o, d); b = walk_ctx.stats()
    assert b["tri_tests_closest"] <= a["tri_tests_closest"] <= 1.5 * b["tri_tests_closest"] + 16
    wide_ctx.set_counting(False); walk_ctx.set_counting(False)


@pytest.mark.parametrize("name", ["cornellbox", "bunny", "teapot-glass", "testObj", "sponza-standin-trace"])
def test_bittrail_kernel_is_the_references_scene_traversal(B, ob, sf, name):
    """StacklessBVH2::intersect (cs488.h:1732-1803), what Scene instantiates, as a literal GPU
    kernel (mipt_context_set_traversal BITTRAIL).  Hits bit for bit against the oracle's bit-trail variant, AND the
    same number of box tests and triangle tests (the traversal is ray-dependent: equal counts mean the same
    culling decisions); the default traversal gives the same hits on the same rays; Scene::isHit through it"""
    arrays = sf.sponza_standin(1) if name == "sponza-standin-trace" else sf.load_scene_arrays(name)
    scene = sf.product_scene_from_mesh_arrays(*arrays)
    orc = sf.oracle_scene_from_mesh_arrays(*arrays)
    ctx = B.Context(scene, device=0)
    cam = B.default_camera(200, 150)
    o, d = ctx.eye_rays(cam)
    rng = np.random.default_rng(9)
    n2 = 20000  # incoherent rays from inside the scene's bounds
    lo, hi = np.asarray(arrays[0], np.float32).reshape(-1, 3).min(0), np.asarray(arrays[0], np.float32).reshape(-1, 3).max(0)
    o2 = (lo + (hi - lo) * rng.uniform(0.05, 0.95, (n2, 3))).astype(np.float32)
    d2 = rng.normal(size=(n2, 3)).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True).astype(np.float32)
    o = np.concatenate([o, o2]); d = np.concatenate([d, d2])
    default_hits = ctx.trace(o, d)
    ctx.set_traversal(B.TRAVERSAL_BITTRAIL)
    ctx.reset_stats()
    ctx.set_counting(True)
    a = ctx.trace(o, d)
    st = ctx.stats()
    ctx.set_counting(False)
    b = orc.intersect(o, d, variant=ob.BITTRAIL)
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(bits(a[2]), bits(b[2]))
    assert (st["node_visits_closest"], st["tri_tests_closest"]) == orc.last_counts
    assert all(np.array_equal(bits(x), bits(y)) for x, y in zip(a, default_hits))
    far = float(np.linalg.norm(hi - lo)) * 0.25
    assert np.array_equal(ctx.occluded(o, d, 0.0, far), orc.occluded(o, d, 0.0, far, variant=ob.BITTRAIL))
    ctx.set_traversal(B.TRAVERSAL_THREADED)
    c = ctx.trace(o, d)
    assert all(np.array_equal(bits(x), bits(y)) for x, y in zip(c, default_hits))
    ctx.set_traversal(B.TRAVERSAL_DEFAULT)


@pytest.mark.parametrize("name", ["cornellbox", "teapot-glass", "sponza-standin-trace"])
def test_all_axes_bvh_gives_the_same_hits(B, ob, sf, name):
    """the three-axis SAH tree (mipt_bvh_build_all_axes) against the ORACLE on the same rays (not
    against another product context).  The oracle walks the reference's tree; its brute-force variant
    (bruteforceIntersect, cs488.h:931-947) tests every triangle and knows no tree at all.  Same t bit for bit as the
    threaded oracle except where a triangle's own leaf box decides (the reference's walk only tests a triangle whose
    leaf box passes AABB::intersect; another tree has other leaf boxes) -- at most 1 ray in 10,000, and every such ray
    must then agree with brute force or with the threaded walk; ids may differ on exact ties only; fewer box tests"""
    arrays = sf.sponza_standin(1) if name == "sponza-standin-trace" else sf.load_scene_arrays(name)
    ref_ctx = B.Context(sf.product_scene_from_mesh_arrays(*arrays), device=0)
    alt_ctx = B.Context(sf.product_scene_from_mesh_arrays(*arrays, all_axes=True), device=0)
    orc = sf.oracle_scene_from_mesh_arrays(*arrays)
    cam = B.default_camera(320, 240)
    o, d = ref_ctx.eye_rays(cam)
    rng = np.random.default_rng(21)
    pos = np.asarray(arrays[0], np.float32).reshape(-1, 3)
    lo, hi = pos.min(0), pos.max(0)
    n2 = 60000
    o = np.concatenate([o, (lo + (hi - lo) * rng.uniform(0.05, 0.95, (n2, 3))).astype(np.float32)])
    d2 = rng.normal(size=(n2, 3)).astype(np.float32)
    d = np.concatenate([d, d2 / np.linalg.norm(d2, axis=1, keepdims=True).astype(np.float32)])
    counts = []
    res = []
    for ctx in (ref_ctx, alt_ctx):
        ctx.reset_stats(); ctx.set_counting(True)
        res.append(ctx.trace(o, d))
        counts.append(ctx.stats()["node_visits_closest"])
        ctx.set_counting(False)
    threaded = orc.intersect(o, d)
    b = res[1]
    assert np.array_equal(res[0][0], threaded[0]) and np.array_equal(bits(res[0][1]), bits(threaded[1]))  # the reference's tree: exact
    same_t = bits(b[1]) == bits(threaded[1])
    assert (~same_t).sum() <= 1e-4 * len(o), (~same_t).sum()
    assert ((b[0] != threaded[0]) & same_t).sum() <= 1e-3 * len(o)  # exact ties only
    assert (b[0] >= 0).sum() > 0.2 * len(o)
    odd = np.flatnonzero(~same_t)
    if odd.size:  # the few rays a leaf box decides: brute force (no boxes at all) is the tie-breaker
        brute = orc.intersect(o[odd], d[odd], variant=ob.BRUTEFORCE)
        assert np.all((bits(b[1][odd]) == bits(brute[1])) | (bits(threaded[1][odd]) == bits(brute[1])))
    sub = rng.choice(len(o), 3000, replace=False)  # and on a subsample the all-axes tree agrees with brute force outright
    brute = orc.intersect(o[sub], d[sub], variant=ob.BRUTEFORCE)
    assert (bits(b[1][sub]) != bits(brute[1])).sum() <= 2
    if name != "cornellbox":  # 36 triangles leave nothing to gain
        assert counts[1] < counts[0], counts


@pytest.mark.parametrize("name", ["cornellbox", "sponza-standin-trace"])
def test_rays_through_vertices_and_the_exact_walk_kernels_on_ordinary_rays(B, ob, sf, name, monkeypatch):
    """rays from edge midpoints towards mesh vertices: exact ties in t between the triangles that share the vertex (the
    reference keeps the first in depth-first order, cs488.h:1674), and leaf boxes the exact predicate rejects although
    one of their triangles accepts the ray -- the cases the order-free traversal's deferred admission must get right
    (an unverified best whose leaf box fails sends the ray to the exact walk kernels).  Second half: with
    MIPT_HAND_BACK_ALL=1 every ray takes the exact walk (a wavefront per ray for short lists, list-mode k_trace for long
    ones), which since deferred admission also sees rays with ordinary directions"""
    arrays = sf.sponza_standin(1) if name == "sponza-standin-trace" else sf.load_scene_arrays(name)
    scene = sf.product_scene_from_mesh_arrays(*arrays)
    orc = sf.oracle_scene_from_mesh_arrays(*arrays)
    tri = np.asarray(arrays[0], np.float32).reshape(-1, 3, 3)
    rng = np.random.default_rng(5)
    for n in (3000, 30000):
        k, k2 = rng.integers(0, len(tri), n), rng.integers(0, len(tri), n)
        o = ((tri[k, 0] + tri[k, 1]) * np.float32(0.5)).astype(np.float32)
        d = tri[k2, 2] - o
        d = (d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)).astype(np.float32)
        ref = orc.intersect(o, d)
        oref = orc.occluded(o, d, 1e-4, 2.0)
        assert (ref[0] >= 0).mean() > 0.5
        # MIPT_DRAIN=1 (the default): short lists (two entries per wavefront of the launch) are traced by k_trace_wide's own
        # wavefronts as they finish, the rest by list-mode k_trace after the launch (round 3); =0: by k_trace_slow in a launch
        # of its own; twice per context, since the list must be back at rest
        # (-1 everywhere) after every call
        for hand_back, drain in (("0", "1"), ("1", "1"), ("1", "0")):
            monkeypatch.setenv("MIPT_HAND_BACK_ALL", hand_back)
            monkeypatch.setenv("MIPT_DRAIN", drain)
            ctx = B.Context(scene, device=0)
            for rep in range(2):
                got = ctx.trace(o, d)
                assert np.array_equal(got[0], ref[0]) and np.array_equal(bits(got[1]), bits(ref[1])) and \
                    np.array_equal(bits(got[2]), bits(ref[2])), (name, n, hand_back, drain, rep)
                assert np.array_equal(ctx.occluded(o, d, 1e-4, 2.0), oref), (name, n, hand_back, drain, rep)
        monkeypatch.delenv("MIPT_HAND_BACK_ALL")
        monkeypatch.delenv("MIPT_DRAIN")


@pytest.mark.parametrize("name", ["teapot-glass", "testObj", "sponza"])
def test_every_hierarchy_over_the_references_leaves_gives_the_oracles_hits(B, ob, sf, name, monkeypatch):
    """k_trace_wide walks a hierarchy of the library's own over the reference's leaves (default: all-axes SAH over the
    leaf boxes, insertion-optimised, collapsed four wide with the least expected visits).  The hits must not depend on
    which hierarchy that is: MIPT_WIDE_TOP = reference | rebuilt | optimised and MIPT_WIDE_COLLAPSE = pairs | greedy |
    optimal all have to return the oracle's bits (threaded walk over the reference's own tree) on primary and on
    incoherent rays, closest hit and any hit -- and the default must visit fewer nodes than round 1's tree did."""
    arrays = sf.sponza_standin(1) if name == "sponza" else sf.load_scene_arrays(name)
    orc = sf.oracle_scene_from_mesh_arrays(*arrays)
    pos = np.asarray(arrays[0], np.float32).reshape(-1, 3)
    lo, hi = pos.min(0), pos.max(0)
    rng = np.random.default_rng(11)
    n = 40000
    o = (lo + (hi - lo) * rng.uniform(0.02, 0.98, (n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    ref = orc.intersect(o, d)
    ref_occ = orc.occluded(o, d, 1e-4, 0.6 * float(np.linalg.norm(hi - lo)))
    visits = {}
    for top, collapse in (("optimised", "optimal"), ("rebuilt", "optimal"), ("rebuilt", "greedy"), ("reference", "optimal"),
                          ("reference", "greedy"), ("reference", "pairs")):
        monkeypatch.setenv("MIPT_WIDE_TOP", top)
        monkeypatch.setenv("MIPT_WIDE_COLLAPSE", collapse)
        scene = sf.product_scene_from_mesh_arrays(*arrays)
        assert scene.wide_nodes > 0
        scene.selfcheck_wide()
        ctx = B.Context(scene, device=0)
        ctx.reset_stats(); ctx.set_counting(True)
        got = ctx.trace(o, d)
        visits[(top, collapse)] = ctx.stats()["node_visits_closest"]
        ctx.set_counting(False)
        assert np.array_equal(got[0], ref[0]), (top, collapse, "triangle ids")
        assert np.array_equal(bits(got[1]), bits(ref[1])) and np.array_equal(bits(got[2]), bits(ref[2])), (top, collapse)
        assert np.array_equal(ctx.occluded(o, d, 1e-4, 0.6 * float(np.linalg.norm(hi - lo))), ref_occ), (top, collapse, "any hit")
    assert visits[("optimised", "optimal")] < visits[("reference", "greedy")]


@pytest.mark.parametrize("scale,expect_wide", [(3.0e5, True), (3.0e-5, True), (700.0, True), (2.0e11, True)])
def test_scenes_at_the_edges_of_the_fp16_node_planes(B, ob, sf, scale, expect_wide):
    """The wide nodes store their planes as fp16 rounded outwards, times a power of two chosen per scene so that the
    largest plane lands in [16384, 32768): a scene scaled by 3e5 or 2e11 (planes far beyond fp16's 65504) or by 3e-5
    (planes in what would be fp16's subnormal range; smaller scenes fall under the reference's own t > 0.00001,
    cs488.h:812) must qualify for the wide traversal like any other and return the oracle's hits, barycentrics and
    any-hit answers bit for bit, on primary and incoherent rays -- admission rests on the exact fp32 leaf box."""
    base = sf.load_scene_arrays("cornellbox")
    arrays = [np.array(a) if isinstance(a, np.ndarray) else a for a in base]
    k = np.float32(scale)
    arrays[0] = (np.asarray(arrays[0], np.float32) * k).astype(np.float32)
    scene = sf.product_scene_from_mesh_arrays(*arrays)
    assert scene.exact_reciprocal
    assert (scene.wide_nodes > 0) == expect_wide
    scene.selfcheck_wide()
    ctx = B.Context(scene, device=0)
    orc = sf.oracle_scene_from_mesh_arrays(*arrays)
    o, d = ctx.eye_rays(B.default_camera(128, 96))
    o = (o * k).astype(np.float32)
    rng = np.random.default_rng(5)
    pos = np.asarray(arrays[0], np.float32).reshape(-1, 3)
    lo, hi = pos.min(0), pos.max(0)
    o2 = (lo + (hi - lo) * rng.uniform(0.02, 0.98, (20000, 3))).astype(np.float32)
    d2 = rng.normal(size=(20000, 3)).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True).astype(np.float32)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    a, b = ctx.trace(o, d), orc.intersect(o, d)
    assert (b[0] >= 0).sum() > 5000
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(bits(a[2]), bits(b[2]))
    tmax = float(0.5 * np.linalg.norm(hi - lo))
    assert np.array_equal(ctx.occluded(o, d, 0.0, tmax), orc.occluded(o, d, 0.0, tmax))


def test_a_given_tree_three_hundred_levels_deep(B, ob, sf):
    """Since round 2 the wide traversal walks a hierarchy of the library's own over the GIVEN tree's leaves, so the given
    tree's depth no longer limits it -- but k_trace_slow, which re-traces the few awkward rays on the given tree with a
    node stack in LDS, has a limit of its own (96 levels; deeper trees hand those rays to the stackless literal walk).
    A hand-made chain BVH over 300 separate triangles (every interior node = one leaf + the rest: 299 levels; the
    reference's builder would never make it, the C-ABI accepts any proper tree): the scene must qualify for the wide
    traversal, and generic rays, axis-parallel rays (handed back: zero direction components) and a batch long enough
    for the list-mode kernel must all return the tree-independent brute-force oracle's hits bit for bit."""
    import ctypes as C
    n = 300
    i = np.arange(n, dtype=np.float32)
    p0 = np.stack([i, 0 * i, 0.05 * (i % 3)], 1); p1 = p0 + np.float32([0.8, 0.0, 0.0]); p2 = p0 + np.float32([0.4, 0.8, 0.02])
    pos = np.stack([p0, p1, p2], 1).astype(np.float32)                       # (n, 3, 3)
    nrm = np.tile(np.float32([0, 0, 1]), (n, 3, 1)); uv = np.zeros((n, 3, 2), np.float32); mid = np.zeros(n, np.int32)
    base = sf.load_scene_arrays("cornellbox")
    mats, texs = base[4][:1], []
    arrays = (pos.reshape(-1), nrm.reshape(-1), uv.reshape(-1), mid, mats, texs)
    mesh = B.Mesh.from_arrays(*arrays)
    scene_ref = B.Scene(mesh)                                                 # the library's own (reference) build: for the desc
    lo, hi = pos.min(1), pos.max(1)                                           # per triangle
    nn = 2 * n - 1
    node_min, node_max = np.zeros((nn, 3), np.float32), np.zeros((nn, 3), np.float32)
    id_hit, id_miss = np.zeros(nn, np.int32), np.zeros(nn, np.int32)
    leaf_first, leaf_count = np.zeros(nn, np.int32), np.zeros(nn, np.int32)
    suffix_lo = np.minimum.accumulate(lo[::-1], 0)[::-1]; suffix_hi = np.maximum.accumulate(hi[::-1], 0)[::-1]
    for k in range(n - 1):                                                    # interior 2k: leaf k | everything after it
        node_min[2 * k], node_max[2 * k] = suffix_lo[k], suffix_hi[k]
        id_hit[2 * k], id_miss[2 * k] = 2 * k + 1, 0
        node_min[2 * k + 1], node_max[2 * k + 1] = lo[k], hi[k]
        id_hit[2 * k + 1] = id_miss[2 * k + 1] = 2 * k + 2
        leaf_first[2 * k + 1], leaf_count[2 * k + 1] = k, 1
    node_min[nn - 1], node_max[nn - 1] = lo[n - 1], hi[n - 1]
    leaf_first[nn - 1], leaf_count[nn - 1] = n - 1, 1
    tri_index = np.arange(n, dtype=np.int32)
    d = B.SceneDesc()
    C.pointer(d)[0] = scene_ref.desc
    keep = [np.ascontiguousarray(a) for a in (node_min.reshape(-1), node_max.reshape(-1), id_hit, id_miss, leaf_first, leaf_count, tri_index)]
    d.n_nodes = nn
    for name, arr in zip(("node_min", "node_max", "id_hit", "id_miss", "leaf_first", "leaf_count", "tri_index"), keep):
        setattr(d, name, arr.ctypes.data_as(type(getattr(d, name))))
    d.n_tri_index = n
    h = C.c_void_p()
    B.check(B.lib().mipt_scene_create(C.byref(d), C.byref(h)))
    deep = B.Scene.__new__(B.Scene)
    deep.h, deep.mesh, deep.bvh, deep.env, deep.lights, deep.desc, deep._keep = h, mesh, scene_ref.bvh, None, scene_ref.lights, d, keep
    assert deep.wide_nodes > 0
    leaves, depth = deep.selfcheck_wide()
    assert leaves == n and depth < 16
    ctx = B.Context(deep, device=0)
    orc = sf.oracle_scene_from_mesh_arrays(*arrays)
    rng = np.random.default_rng(9)
    m = 20000
    o = np.stack([rng.uniform(-1, n + 1, m), rng.uniform(-0.5, 1.3, m), np.full(m, 3.0)], 1).astype(np.float32)
    dgen = rng.normal(size=(m, 3)).astype(np.float32); dgen[:, 2] = -np.abs(dgen[:, 2]) - 0.5
    dgen /= np.linalg.norm(dgen, axis=1, keepdims=True).astype(np.float32)
    down = np.tile(np.float32([0, 0, -1]), (m, 1))                            # two zero components: every one handed back
    for dirs, label in ((dgen, "generic"), (down[:3000], "axis-parallel, short list"), (down, "axis-parallel, long list")):
        oo = o[: len(dirs)]
        got = ctx.trace(oo, dirs)
        ref = orc.intersect(oo, dirs, variant=ob.BRUTEFORCE)
        assert (ref[0] >= 0).sum() > len(dirs) // 20, label
        assert np.array_equal(got[0], ref[0]) and np.array_equal(bits(got[1]), bits(ref[1])) and np.array_equal(bits(got[2]), bits(ref[2])), label
        assert np.array_equal(ctx.occluded(oo, dirs, 0.0, 10.0), orc.occluded(oo, dirs, 0.0, 10.0)), label
