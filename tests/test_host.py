"""Host logic behind the C-ABI (no GPU): library loads and exports every
declared symbol, loader / BVH build / image IO, scene validation errors."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEDIA = "/root/reference/media"  # only present in the build container
needs_media = pytest.mark.skipif(not os.path.isdir(MEDIA), reason="reference media not on this machine")


def test_library_exports_every_declared_symbol(B):
    header = open(os.path.join(ROOT, "include", "mipt.h")).read()
    declared = set(re.findall(r"\b(mipt_[a-z0-9_]+)\s*\(", header))
    assert declared == set(B.SYMBOLS), declared ^ set(B.SYMBOLS)
    lib = B.lib()
    for s in declared:
        assert hasattr(lib, s), s
    assert b"gfx950" in lib.mipt_version()


def test_no_gpu_means_error_not_fallback(B, sf):
    """on a machine without a GPU every compute entry point must fail loudly"""
    import subprocess
    if subprocess.run(["bash", "-c", "ls /dev/kfd"], capture_output=True).returncode == 0:
        pytest.skip("a GPU is present")
    scene = sf.product_scene_from_mesh_arrays(*sf.load_scene_arrays("cornellbox"))
    with pytest.raises(B.MiptError):
        B.Context(scene)


def test_bvh_build_equals_oracle_build(B, ob, sf):
    """the product's builder (iterative, in place) and the oracle's (recursive restatement of
    cs488.h:1412-1595, 1649-1659, 1718-1728) give identical trees: boxes, links, leaf order"""
    for name in ("cornellbox", "cube", "teapot-glass", "testObj", "bunny"):
        arrays = sf.load_scene_arrays(name)
        mesh = B.Mesh.from_arrays(*arrays)
        bvh = B.BVH(mesh)
        a = bvh.arrays()
        o = sf.oracle_scene_from_mesh_arrays(*arrays)
        b = o.export_bvh()
        st = o.stats()
        assert (bvh.num_nodes, bvh.num_leaves, bvh.depth, bvh.max_leaf_size) == (st["nodes"], st["leaves"], st["depth"], st["max_leaf"])
        for k in ("node_min", "node_max"):
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), (name, k)
        for k in ("id_hit", "id_miss", "leaf_first", "leaf_count", "id_parent", "id_sibling", "tri_index"):
            assert np.array_equal(a[k], b[k]), (name, k)
        interior = a["leaf_count"] == 0
        assert np.array_equal(a["id_left"][interior], b["id_left"][interior])
        assert np.array_equal(a["id_right"][interior], b["id_right"][interior])
        # preorder: idLeft == id + 1; threading points forward
        ids = np.arange(bvh.num_nodes)
        assert np.all(a["id_left"][interior] == ids[interior] + 1)
        assert np.all((a["id_miss"] == 0) | (a["id_miss"] > ids))


def test_bvh_degenerate_inputs(B):
    # one triangle; identical triangles (all SAH costs tie); zero-area parent box (area 0 -> NaN costs -> leaf)
    tri = np.array([0, 0, 0, 1, 0, 0, 0, 1, 0], np.float32)
    nrm = np.array([0, 0, 1] * 3, np.float32)
    for n in (1, 2, 7):
        mesh = B.Mesh.from_arrays(np.tile(tri, n), np.tile(nrm, n))
        bvh = B.BVH(mesh)
        a = bvh.arrays()
        assert bvh.num_leaves >= 1 and a["leaf_count"].sum() == n
        assert sorted(a["tri_index"].tolist()) == list(range(n))
    seg = np.array([0, 0, 0, 1, 0, 0, 2, 0, 0], np.float32)  # degenerate: all points on a line
    bvh = B.BVH(B.Mesh.from_arrays(np.tile(seg, 5), np.tile(nrm, 5)))
    assert bvh.arrays()["leaf_count"].sum() == 5


def test_single_triangle_mesh(B):
    m = B.Mesh.single_triangle()  # cs488.h:949-970
    assert m.num_triangles == 1
    assert np.allclose(m.positions(), [-0.5, -0.5, 0, 0.5, -0.5, 0, 0, 0.5, 0])
    assert np.allclose(m.normals(), [0, 0, 1] * 3)
    assert np.allclose(m.texcoords(), [0, 0, 0, 1, 1, 0])


@needs_media
def test_obj_loader_counts_and_material_typing(B):
    """TriangleMesh::load (cs488.h:835-923): triangle counts and the typing rules C:849-868"""
    expect = {"cornellbox": (36, [0, 0, 0, 3, 0]), "cornellbox-metal": (36, None), "cornellbox-glass": (36, None),
              "bunny": (69451, [0]), "teapot-glass": (15704, [2]), "testObj": (11950, [0] * 5), "cube": (12, None),
              "textured": (2, None)}
    for name, (ntri, types) in expect.items():
        m = B.Mesh.load(os.path.join(MEDIA, name + ".obj"))
        assert m.num_triangles == ntri
        if types is not None:
            assert [x["type"] for x in m.materials()] == types
    glass = B.Mesh.load(os.path.join(MEDIA, "teapot-glass.obj")).materials()[0]
    assert glass["eta"] == 1.5 and glass["tex_id"] == 0  # name "glass*" -> eta 1.5; map_Kd default.png
    metal = B.Mesh.load(os.path.join(MEDIA, "cornellbox-metal.obj")).materials()
    assert 1 in [x["type"] for x in metal]  # Ns == 100 -> MAT_METAL
    bunny = B.Mesh.load(os.path.join(MEDIA, "bunny.obj"))
    assert abs(bunny.materials()[0]["Kd"][0] - 0.9) < 1e-7  # no mtl -> one default Lambertian
    n = bunny.normals().reshape(-1, 3, 3)
    assert np.array_equal(n[:, 0], n[:, 1]) and np.array_equal(n[:, 0], n[:, 2])  # face normals (C:884-892)


@needs_media
def test_committed_scene_fixtures_equal_the_loader(B, sf):
    for name in ("cornellbox", "teapot-glass", "textured"):
        m = B.Mesh.load(os.path.join(MEDIA, name + ".obj"))
        pos, nrm, uv, mid, mats, texs = sf.load_scene_arrays(name)
        assert np.array_equal(pos.view(np.uint32), m.positions().view(np.uint32))
        assert np.array_equal(nrm.view(np.uint32), m.normals().view(np.uint32))
        assert np.array_equal(uv.view(np.uint32), m.texcoords().view(np.uint32))
        assert np.array_equal(mid, m.material_ids())
        assert all(np.array_equal(a, b) for a, b in zip(texs, m.textures()))


def test_loader_errors(B, tmp_path):
    with pytest.raises(B.MiptError):
        B.Mesh.load(str(tmp_path / "missing.obj"))
    p = tmp_path / "empty.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\n")
    with pytest.raises(B.MiptError):
        B.Mesh.load(str(p))
    p = tmp_path / "badidx.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n")
    with pytest.raises(B.MiptError):
        B.Mesh.load(str(p))
    p = tmp_path / "nomtl.obj"
    p.write_text("mtllib nothere.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    with pytest.raises(B.MiptError):
        B.Mesh.load(str(p))


def test_loader_small_obj_with_mtl(B, tmp_path):
    (tmp_path / "m.mtl").write_text("newmtl lightA\nKd 1 1 1\nNs 5\nnewmtl mirror\nKs .9 .8 .7\nNs 100\nnewmtl glassy\nNs 1\n")
    (tmp_path / "s.obj").write_text(
        "mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nvn 0 0 2\nusemtl mirror\nf 1//1 2//1 3//1\nusemtl glassy\n"
        "f 1//1 3//1 4//1 2//1\nusemtl lightA\nf 2//1 3//1 4//1\n")
    m = B.Mesh.load(str(tmp_path / "s.obj"))
    assert m.num_triangles == 3  # the quad contributes its first three vertices only (C:1177)
    assert m.material_ids().tolist() == [1, 2, 0]
    assert [x["type"] for x in m.materials()] == [B.MAT_LIGHT, B.MAT_METAL, B.MAT_GLASS]
    assert np.allclose(m.normals()[:3], [0, 0, 1])  # vn normalised at parse time (C:1133-1137)
    assert m.materials()[2]["Ks"] == pytest.approx((0.9, 0.8, 0.7))  # Ks carries over to the next material


def test_image_hdr_png_roundtrip(B, sf, tmp_path):
    env = sf.procedural_env(64)
    p = str(tmp_path / "e.hdr")
    B.save_hdr(p, env)
    back = B.Image.load(p).pixels()
    assert back.shape == env.shape
    # RGBE keeps 8 mantissa bits of the largest channel
    assert np.all(np.abs(back - env) <= env.max(axis=2, keepdims=True) / 128.0 + 1e-6)
    # orientation: row 0 is the bottom row on both sides
    assert np.argmax(back.sum(axis=(1, 2))) == np.argmax(env.sum(axis=(1, 2)))
    png = str(tmp_path / "o.png")
    ramp = np.zeros((4, 8, 3), np.float32)
    ramp[..., 0] = np.linspace(-0.5, 1.5, 8)[None, :]  # clamp both ends (Image::save, C:223-236)
    ramp[3, :, 1] = 1.0  # top row green
    B.save_png(png, ramp)
    img = B.Image.load(png).pixels()  # PNG -> float goes through pow(v/255, 2.2)
    assert img.shape == (4, 8, 3)
    assert np.all(img[3, :, 1] == 1.0) and np.all(img[:3, :, 1] == 0.0)
    expect = np.power(np.floor(255.0 * np.clip(ramp[0, :, 0], 0, 1)) / 255.0, 2.2)
    assert np.allclose(img[0, :, 0], expect, atol=1e-6)
    B.save_pfm(str(tmp_path / "o.pfm"), ramp)
    raw = open(tmp_path / "o.pfm", "rb").read()
    assert raw.startswith(b"PF\n8 4\n-1.0\n") and len(raw) == len(b"PF\n8 4\n-1.0\n") + ramp.nbytes


@needs_media
def test_png_decoder_on_reference_textures(B):
    teapot = B.Mesh.load(os.path.join(MEDIA, "teapot-glass.obj"))
    tex = teapot.textures()[0]  # default.png: 128x128, 1-bit palette
    assert tex.shape == (128, 128, 3) and len(np.unique(tex.reshape(-1, 3), axis=0)) <= 2
    tex2 = B.Mesh.load(os.path.join(MEDIA, "textured.obj")).textures()[0]  # test.png: 1044x1059 RGB
    assert tex2.shape == (1059, 1044, 3) and tex2.std() > 10


def test_scene_validation_rejects_bad_links(B, sf):
    """the device walk trusts the links; mipt_scene_create must refuse anything that could loop or overrun"""
    arrays = sf.load_scene_arrays("cube")
    mesh = B.Mesh.from_arrays(*arrays)
    bvh = B.BVH(mesh)
    lights = (B.Light * 1)()
    good = B.SceneDesc()
    B.check(B.lib().mipt_scene_desc_fill(mesh.h, bvh.h, None, lights, 1, C.byref(good)))
    h = C.c_void_p()
    B.check(B.lib().mipt_scene_create(C.byref(good), C.byref(h)))
    B.lib().mipt_scene_destroy(h)
    a = bvh.arrays()

    def with_patch(field, arr):
        d = B.SceneDesc()
        C.pointer(d)[0] = good
        keep = np.ascontiguousarray(arr)
        setattr(d, field, keep.ctypes.data_as(type(getattr(d, field))))
        out = C.c_void_p()
        rc = B.lib().mipt_scene_create(C.byref(d), C.byref(out))
        if rc == 0:
            B.lib().mipt_scene_destroy(out)
        return rc

    bad = a["id_miss"].copy(); bad[3] = 1  # backward link -> possible cycle
    assert with_patch("id_miss", bad) == -1
    bad = a["id_hit"].copy(); bad[0] = 0  # interior node whose hit link is the sentinel
    assert with_patch("id_hit", bad) == -1
    bad = a["leaf_first"].copy(); bad[np.flatnonzero(a["leaf_count"])[0]] = 10 ** 6
    assert with_patch("leaf_first", bad) == -1
    bad = a["tri_index"].copy(); bad[0] = 999
    assert with_patch("tri_index", bad) == -1
    d = B.SceneDesc()
    C.pointer(d)[0] = good
    d.n_lights = 17
    assert B.lib().mipt_scene_create(C.byref(d), C.byref(h)) == -5  # MIPT_ERR_LIMIT
    assert b"16" in B.lib().mipt_last_error()


def test_bench_scene_qualifies_for_exact_reciprocal(B, sf):
    """the bench workload must run the traversal kernel's reciprocal slab division (a scene with one plane outside
    [2^-61, 2^40] silently falls back to IEEE division for every ray); a scaled-down copy must not qualify"""
    arrays = list(sf.sponza_standin(1))
    assert sf.product_scene_from_mesh_arrays(*arrays, lights=sf.SPONZA_LIGHT).exact_reciprocal
    arrays[0] = (np.asarray(arrays[0], np.float32) * np.float32(2.0 ** -70)).astype(np.float32)
    assert not sf.product_scene_from_mesh_arrays(*arrays, lights=sf.SPONZA_LIGHT).exact_reciprocal


def test_wide_tree_eligibility(B, sf):
    """the order-free traversal is only used where its proof holds: boxes nested child-inside-parent, a proper
    binary tree, tame planes.  The bench scene and every fixture qualify; a tree with one child box poking out of
    its parent, or a threaded graph that is not a tree, must fall back to the literal walk (wide_nodes == 0) while
    still being accepted (the links point forward, so the literal walk terminates)"""
    assert sf.product_scene_from_mesh_arrays(*sf.sponza_standin(1), lights=sf.SPONZA_LIGHT).wide_nodes > 6600  # 46,251 leaves, at most 8 slots per node
    arrays = sf.load_scene_arrays("cube")
    mesh = B.Mesh.from_arrays(*arrays)
    bvh = B.BVH(mesh)
    lights = (B.Light * 1)()
    good = B.SceneDesc()
    B.check(B.lib().mipt_scene_desc_fill(mesh.h, bvh.h, None, lights, 1, C.byref(good)))
    a = bvh.arrays()

    def wide_nodes_with(field, arr):
        d = B.SceneDesc()
        C.pointer(d)[0] = good
        keep = np.ascontiguousarray(arr)
        setattr(d, field, keep.ctypes.data_as(type(getattr(d, field))))
        out = C.c_void_p()
        B.check(B.lib().mipt_scene_create(C.byref(d), C.byref(out)))
        n = B.lib().mipt_scene_wide_nodes(out)
        B.lib().mipt_scene_destroy(out)
        return n

    assert wide_nodes_with("node_min", a["node_min"]) > 0  # unchanged: qualifies
    interior = np.flatnonzero(a["leaf_count"] == 0)
    child = int(a["id_hit"][interior[0]])
    bad = a["node_min"].copy(); bad[3 * child] -= 1.0  # the child's box now sticks out of its parent's
    assert wide_nodes_with("node_min", bad) == 0
    bad = a["node_max"].copy(); bad[3 * child + 1] += 0.5
    assert wide_nodes_with("node_max", bad) == 0
    # a forward-pointing threaded graph that is not a tree: a leaf's miss link skips its sibling subtree
    leaf = int(np.flatnonzero(a["leaf_count"] > 0)[0])
    if a["id_miss"][leaf] != 0 and a["id_miss"][a["id_miss"][leaf]] != 0:
        bad_miss, bad_hit = a["id_miss"].copy(), a["id_hit"].copy()
        bad_miss[leaf] = bad_hit[leaf] = a["id_miss"][a["id_miss"][leaf]]
        d = B.SceneDesc()
        C.pointer(d)[0] = good
        k1, k2 = np.ascontiguousarray(bad_miss), np.ascontiguousarray(bad_hit)
        d.id_miss = k1.ctypes.data_as(type(d.id_miss))
        d.id_hit = k2.ctypes.data_as(type(d.id_hit))
        out = C.c_void_p()
        if B.lib().mipt_scene_create(C.byref(d), C.byref(out)) == 0:
            assert B.lib().mipt_scene_wide_nodes(out) == 0
            B.lib().mipt_scene_destroy(out)


def test_scene_cache_round_trip_and_rejects_damage(B, tmp_path):
    """the flat cache returns the mesh and the BVH array for array (so the scene built from it is the
    one built from the .obj), and a truncated, bit-flipped or foreign file is refused"""
    mesh = B.Mesh.load(os.path.join(MEDIA, "cornellbox.obj")) if os.path.isdir(MEDIA) else B.Mesh.single_triangle()
    bvh = B.BVH(mesh)
    path = str(tmp_path / "scene.miptsc")
    B.save_scene_cache(path, mesh, bvh)
    mesh2, bvh2 = B.load_scene_cache(path)
    for f in ("positions", "normals", "texcoords", "material_ids"):
        assert np.array_equal(getattr(mesh, f)().view(np.uint32), getattr(mesh2, f)().view(np.uint32)), f
    assert mesh.materials() == mesh2.materials()
    assert all(np.array_equal(a, b) for a, b in zip(mesh.textures(), mesh2.textures())) and len(mesh.textures()) == len(mesh2.textures())
    a, b = bvh.arrays(), bvh2.arrays()
    assert a.keys() == b.keys() and all(np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)) for k in a)
    assert (bvh.num_nodes, bvh.num_leaves, bvh.depth, bvh.max_leaf_size) == (bvh2.num_nodes, bvh2.num_leaves, bvh2.depth, bvh2.max_leaf_size)
    blob = open(path, "rb").read()
    bad = str(tmp_path / "bad.miptsc")
    for damaged in (blob[:len(blob) // 2], blob[:100] + bytes([blob[100] ^ 1]) + blob[101:], b"OBJ " + blob[4:], b""):
        open(bad, "wb").write(damaged)
        with pytest.raises(B.MiptError):
            B.load_scene_cache(bad)
    with pytest.raises(B.MiptError):
        B.load_scene_cache(str(tmp_path / "missing.miptsc"))


def test_pack_rgb8_and_compare_images(B):
    """Image::save's channel conversion (C:159-168, C:228) and the RMSE helper, against numpy"""
    rng = np.random.default_rng(2)
    v = rng.uniform(-0.5, 1.5, 4099).astype(np.float32)
    v[:8] = [0.0, -0.0, 1.0, 0.999999, 1.0 / 255, np.inf, -np.inf, 1e-40]
    want = (np.float32(255.0) * np.clip(v, 0.0, 1.0).astype(np.float32)).astype(np.uint8)
    assert np.array_equal(B.pack_rgb8(v), want)
    assert B.pack_rgb8(np.array([np.nan], np.float32))[0] == 255  # std::min(1.0f, NaN) is 1.0f
    a = rng.uniform(0, 4, (7, 5, 3)).astype(np.float32)
    b = (a + rng.normal(0, 0.1, a.shape)).astype(np.float32)
    rmse, ma, mb = B.compare_images(a, b)
    assert abs(rmse - np.sqrt(np.mean((a.astype(np.float64) - b) ** 2))) < 1e-12
    assert abs(ma - a.astype(np.float64).mean()) < 1e-12 and abs(mb - b.astype(np.float64).mean()) < 1e-12


def test_all_axes_bvh_is_a_valid_tree_and_cheaper(B, sf):
    """the three-axis SAH build is not the reference's tree, but it must be a tree of the same kind
    -- every triangle in exactly one leaf, every box the tight union of what lies below it, preorder links -- so
    that everything downstream (wide tree, bit-trail links) accepts it, and it should be the better tree"""
    mesh = B.Mesh.from_arrays(*sf.sponza_standin(1))
    ref, alt = B.BVH(mesh), B.BVH(mesh, all_axes=True)
    a = alt.arrays()
    n = alt.num_nodes
    pos = mesh.positions().reshape(-1, 3, 3)
    leaf = a["leaf_count"] > 0
    assert sorted(a["tri_index"].tolist()) == list(range(mesh.num_triangles))
    lo, hi = a["node_min"].reshape(-1, 3), a["node_max"].reshape(-1, 3)
    for i in np.flatnonzero(leaf)[::97]:  # a leaf's box is the tight box of its triangles
        t = pos[a["tri_index"][a["leaf_first"][i]:a["leaf_first"][i] + a["leaf_count"][i]]].reshape(-1, 3)
        assert np.array_equal(lo[i], t.min(0)) and np.array_equal(hi[i], t.max(0))
    inner = np.flatnonzero(~leaf)
    l, r = a["id_left"][inner], a["id_right"][inner]
    assert np.array_equal(l, inner + 1) and (r > l).all() and (r < n).all()  # preorder numbering
    assert np.array_equal(lo[inner], np.minimum(lo[l], lo[r])) and np.array_equal(hi[inner], np.maximum(hi[l], hi[r]))
    assert alt.depth <= 31 and alt.num_nodes < ref.num_nodes
    scene = B.Scene(mesh, bvh=alt, lights=sf.SPONZA_LIGHT)
    assert scene.wide_nodes > 0 and scene.exact_reciprocal


def test_scene_cache_source_tag(B, tmp_path):
    """a cache records what it was built from; a load that names another source is refused (the CLI then
    rebuilds), a load without a tag takes any cache"""
    mesh = B.Mesh.single_triangle()
    bvh = B.BVH(mesh)
    path = str(tmp_path / "tagged.miptsc")
    src = tmp_path / "a.obj"
    src.write_text("v 0 0 0\n")
    tag = "obj " + B.file_fingerprint(str(src)) + "; bvh reference"
    assert B.file_fingerprint(str(src)).startswith("a.obj:8:") and B.file_fingerprint(str(tmp_path / "none.obj")) == "none.obj:absent"
    B.save_scene_cache(path, mesh, bvh, tag=tag)
    B.load_scene_cache(path, expect_tag=tag)
    B.load_scene_cache(path)
    src.write_text("v 0 0 1\n")  # same size, other content
    tag2 = "obj " + B.file_fingerprint(str(src)) + "; bvh reference"
    assert tag2 != tag
    with pytest.raises(B.MiptError, match="another source"):
        B.load_scene_cache(path, expect_tag=tag2)
    with pytest.raises(B.MiptError, match="another source"):
        B.load_scene_cache(path, expect_tag=tag.replace("reference", "all-axes"))


def test_hdr_header_larger_than_file_is_refused(B, tmp_path):
    """an .hdr whose header announces more pixels than the file can hold must fail before allocating"""
    p = tmp_path / "huge.hdr"
    p.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 30000 +X 30000\n" + bytes(64))
    with pytest.raises(B.MiptError, match="more pixels"):
        B.Image.load(str(p))
    p.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2000000000 +X 2000000000\n" + bytes(64))
    with pytest.raises(B.MiptError):
        B.Image.load(str(p))


@pytest.mark.parametrize("top", ["optimised", "rebuilt", "reference"])
@pytest.mark.parametrize("name", ["cornellbox", "teapot-glass", "testObj", "sponza", "single"])
def test_wide_tree_over_the_references_leaves(B, sf, name, top, monkeypatch):
    """The order-free traversal walks a hierarchy of the library's own over the REFERENCE's leaves (an all-axes SAH build
    over the leaf boxes, insertion-optimised, collapsed four wide; MIPT_WIDE_TOP selects the plainer ones).  Whatever
    the hierarchy, the hits are the reference's only if it reaches every reference leaf exactly once and every stored
    (fp16, outward-rounded) box contains everything below it down to the leaf's exact box: mipt_scene_selfcheck_wide
    checks exactly that on the arrays the GPU will read."""
    monkeypatch.setenv("MIPT_WIDE_TOP", top)
    if name == "single":
        mesh = B.Mesh.single_triangle() if hasattr(B.Mesh, "single_triangle") else None
        if mesh is None:
            pytest.skip("no single-triangle helper")
        scene = B.Scene(mesh)
        n_leaves_ref = 1
    else:
        arrays = sf.sponza_standin(1) if name == "sponza" else sf.load_scene_arrays(name)
        scene = sf.product_scene_from_mesh_arrays(*arrays, lights=sf.DEFAULT_LIGHT)
        n_leaves_ref = None
    assert scene.wide_nodes > 0
    leaves, depth = scene.selfcheck_wide()
    assert leaves > 0 and 0 <= depth < 16
    if n_leaves_ref is not None:
        assert leaves == n_leaves_ref


def test_wide_hierarchy_cache_and_thread_independence(B, sf, tmp_path, monkeypatch):
    """mipt_scene_create_cached: the second creation loads the hierarchy the first one built -- the same arrays bit for bit
    (mipt_scene_wide_hash) and faster; a build on one thread and on all of them give the same tree; a damaged or foreign
    file is ignored and rewritten; other lights (the slot order depends on them) are another key"""
    import time
    arrays = sf.sponza_standin(1)
    path = str(tmp_path / "atrium.wide")
    t0 = time.perf_counter()
    a = sf.product_scene_from_mesh_arrays(*arrays, lights=sf.SPONZA_LIGHT, wide_cache=path)
    t_build = time.perf_counter() - t0
    assert os.path.getsize(path) > 1000 and a.wide_nodes > 0
    a.selfcheck_wide()
    t0 = time.perf_counter()
    b = sf.product_scene_from_mesh_arrays(*arrays, lights=sf.SPONZA_LIGHT, wide_cache=path)
    t_load = time.perf_counter() - t0
    assert b.wide_hash == a.wide_hash and b.wide_nodes == a.wide_nodes
    b.selfcheck_wide()
    assert t_load < t_build  # (the mesh upload and the reference BVH are in both)
    plain = sf.product_scene_from_mesh_arrays(*arrays, lights=sf.SPONZA_LIGHT)
    assert plain.wide_hash == a.wide_hash
    monkeypatch.setenv("MIPT_BUILD_THREADS", "1")
    one = sf.product_scene_from_mesh_arrays(*arrays, lights=sf.SPONZA_LIGHT)
    monkeypatch.setenv("MIPT_BUILD_THREADS", "5")
    five = sf.product_scene_from_mesh_arrays(*arrays, lights=sf.SPONZA_LIGHT)
    monkeypatch.delenv("MIPT_BUILD_THREADS")
    assert one.wide_hash == a.wide_hash and five.wide_hash == a.wide_hash
    # damage: a flipped byte in the payload, a truncated file, a foreign file -> rebuilt, same tree, file whole again
    good = open(path, "rb").read()
    for bad in (good[:200] + bytes([good[200] ^ 1]) + good[201:], good[: len(good) // 2], b"not a cache"):
        open(path, "wb").write(bad)
        c = sf.product_scene_from_mesh_arrays(*arrays, lights=sf.SPONZA_LIGHT, wide_cache=path)
        assert c.wide_hash == a.wide_hash
        assert open(path, "rb").read() == good
    other = sf.product_scene_from_mesh_arrays(*arrays, lights=((0.0, 1.0, 0.0, 10.0, 10.0, 10.0),), wide_cache=path)
    assert other.wide_nodes > 0 and open(path, "rb").read() != good  # another key: rebuilt and overwritten
    other.selfcheck_wide()
