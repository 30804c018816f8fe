"""Scene fixtures shared by the tests, smoke() and bench.py: committed meshes
(tests/golden/scene_*.npz), the procedural stand-ins for the assets the
reference's media directory lacks (Sponza, uffizi_probe.hdr), and helpers
that put the same scene into the product (libmipt.so) and into the oracle."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

DEFAULT_LIGHT = ((3.0, 3.0, 3.0, 1000.0, 1000.0, 1000.0),)  # reference main.cpp:26-27


def encode_rows(rows):
    """lossless: unique rows (bitwise) + index"""
    rows = np.ascontiguousarray(rows, np.float32)
    bits = rows.view(np.uint32)
    table, idx = np.unique(bits, axis=0, return_inverse=True)
    return table.view(np.float32), idx.reshape(-1).astype(np.int32)


def load_scene_arrays(name):
    """-> pos(n*9) nrm(n*9) uv(n*6) mat_id(n) materials(list of dict) textures(list of arrays)"""
    z = np.load(os.path.join(GOLDEN, f"scene_{name}.npz"))
    pos = z["pos_table"][z["pos_index"]].reshape(-1)
    nrm = z["nrm_table"][z["nrm_index"]].reshape(-1)
    uv = z["uv_table"][z["uv_index"]].reshape(-1)
    mats = []
    for r in z["materials"]:
        mats.append(dict(type=int(r[0]), eta=float(r[1]), Ka=tuple(map(float, r[2:5])), Kd=tuple(map(float, r[5:8])),
                         Ks=tuple(map(float, r[8:11])), Ns=float(r[11]), tex_id=int(r[12])))
    texs = []
    while f"tex{len(texs)}" in z.files:
        texs.append(z[f"tex{len(texs)}"])
    return pos, nrm, uv, z["mat_id"], mats, texs


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"golden_{name}.npz"))


def oracle_scene_from_mesh_arrays(pos, nrm, uv, mid, mats, texs, env=None, lights=DEFAULT_LIGHT):
    import oracle_binding as ob
    return ob.OracleScene(pos, nrm, uv, mid, mats, texs, env=env, lights=lights)


def product_scene_from_mesh_arrays(pos, nrm, uv, mid, mats, texs, env=None, lights=DEFAULT_LIGHT, all_axes=False, wide_cache=None):
    """all_axes: the three-axis SAH build (mipt_bvh_build_all_axes) instead of the reference's tree;
    wide_cache: file that keeps the library's hierarchy above the leaves (mipt_scene_create_cached)"""
    B = importlib.import_module("path-tracer_amd.binding")
    mesh = B.Mesh.from_arrays(pos, nrm, uv, mid, mats, texs)
    img = B.Image.from_array(env) if env is not None else None
    return B.Scene(mesh, bvh=B.BVH(mesh, all_axes=True) if all_axes else None, env=img, lights=lights, wide_cache=wide_cache)


def placed(arrays, scale=1.0, offset=(0.0, 0.0, 0.0)):
    """a mesh's arrays with its positions scaled and moved (float32, computed once: the oracle and the product get the
    same bits); normals, texture coordinates and materials unchanged"""
    pos, nrm, uv, mid, mats, texs = arrays
    p = np.asarray(pos, np.float32).reshape(-1, 3) * np.float32(scale) + np.asarray(offset, np.float32)
    return np.ascontiguousarray(p, np.float32).reshape(-1), nrm, uv, mid, mats, texs


def room_objects():
    """Several objects in one scene (Scene::addObject, cs488.h:2020): the metal Cornell box, a bunny and a glass teapot
    inside it, the SAME bunny once more (every hit on it is an exact tie between two objects: the earlier one must win,
    cs488.h:2078) and a textured cube.  A list of mesh-array tuples, in object order."""
    bunny = placed(load_scene_arrays("bunny"), 0.35, (-0.12, -0.22, -0.05))
    return [load_scene_arrays("cornellbox-metal"), bunny, placed(load_scene_arrays("teapot-glass"), 0.3, (0.16, 0.05, 0.12)),
            bunny, placed(load_scene_arrays("cube"), 0.12, (0.2, -0.3, 0.2))]


def oracle_scene_from_objects(objects, env=None, lights=DEFAULT_LIGHT):
    """the oracle's scene of several objects: object 0 carries environment and lights, the others are added in order"""
    import oracle_binding as ob
    head = ob.OracleScene(*objects[0], env=env, lights=lights)
    for arrays in objects[1:]:
        head.add_object(ob.OracleScene(*arrays, env=None, lights=lights))
    return head


def product_scene_from_objects(objects, env=None, lights=DEFAULT_LIGHT, wide_cache=None):
    B = importlib.import_module("path-tracer_amd.binding")
    meshes = [B.Mesh.from_arrays(*arrays) for arrays in objects]
    img = B.Image.from_array(env) if env is not None else None
    return B.Scene(meshes, env=img, lights=lights, wide_cache=wide_cache)


def procedural_env(size=256):
    """Deterministic stand-in for uffizi_probe.hdr (absent from the reference
    tree): an angular-map probe with a sky gradient and three Gaussian
    'windows'. Pure formula, no RNG. Returned bottom-row-first float32 (h,w,3)."""
    j, i = np.meshgrid(np.arange(size, dtype=np.float64), np.arange(size, dtype=np.float64), indexing="ij")
    u = (i + 0.5) / size * 2.0 - 1.0
    v = (j + 0.5) / size * 2.0 - 1.0
    r = np.sqrt(u * u + v * v)
    sky = 0.25 + 0.55 * np.clip(0.5 + 0.5 * v, 0.0, 1.0)
    img = np.stack([0.9 * sky, 1.0 * sky, 1.25 * sky], axis=-1)
    for cx, cy, sig, amp, col in ((0.35, 0.45, 0.10, 18.0, (1.0, 0.92, 0.75)), (-0.5, 0.2, 0.07, 9.0, (0.8, 0.9, 1.0)),
                                  (0.0, -0.6, 0.2, 1.5, (1.0, 0.7, 0.5))):
        g = amp * np.exp(-((u - cx) ** 2 + (v - cy) ** 2) / (2.0 * sig * sig))
        img += g[..., None] * np.asarray(col)
    img *= np.clip(1.15 - 0.3 * r, 0.2, 1.2)[..., None]
    return np.ascontiguousarray(img, np.float32)


def sponza_standin(detail=1):
    """Procedural stand-in for sponza.obj (absent, .MISSING_LARGE_BLOBS): a
    closed atrium -- floor, four walls, ceiling with a skylight opening, two
    storeys of colonnades with arches, hanging drapes.  Deterministic.
    detail=1 gives ~66k triangles, detail=2 ~262k.
    Returns pos, nrm, uv, mat_id, materials, textures (all Lambertian, 6 materials)."""
    tris = []  # (p0,p1,p2, mat)

    def quad(a, b, c, d, mat, nu=1, nv=1):
        a, b, c, d = (np.asarray(x, np.float64) for x in (a, b, c, d))
        for iu in range(nu):
            for iv in range(nv):
                u0, u1, v0, v1 = iu / nu, (iu + 1) / nu, iv / nv, (iv + 1) / nv

                def P(u, v):
                    return (a * (1 - u) + b * u) * (1 - v) + (d * (1 - u) + c * u) * v
                tris.append((P(u0, v0), P(u1, v0), P(u1, v1), mat))
                tris.append((P(u0, v0), P(u1, v1), P(u0, v1), mat))

    def column(cx, cz, y0, y1, rad, seg, rings, mat):
        for k in range(seg):
            a0, a1 = 2 * np.pi * k / seg, 2 * np.pi * (k + 1) / seg
            for r_ in range(rings):
                ya, yb = y0 + (y1 - y0) * r_ / rings, y0 + (y1 - y0) * (r_ + 1) / rings
                # slight entasis
                ra = rad * (1.0 - 0.15 * (r_ / rings))
                rb = rad * (1.0 - 0.15 * ((r_ + 1) / rings))
                p00 = (cx + ra * np.cos(a0), ya, cz + ra * np.sin(a0))
                p10 = (cx + ra * np.cos(a1), ya, cz + ra * np.sin(a1))
                p11 = (cx + rb * np.cos(a1), yb, cz + rb * np.sin(a1))
                p01 = (cx + rb * np.cos(a0), yb, cz + rb * np.sin(a0))
                quad(p00, p10, p11, p01, mat)

    def arch(x0, x1, z, ybase, height, thick, seg, mat):
        # semicircular arch band between two columns, in the plane z = const
        cxm, rad = 0.5 * (x0 + x1), 0.5 * (x1 - x0)
        for k in range(seg):
            a0, a1 = np.pi * k / seg, np.pi * (k + 1) / seg
            i0 = (cxm - rad * np.cos(a0), ybase + rad * np.sin(a0) * height, z)
            i1 = (cxm - rad * np.cos(a1), ybase + rad * np.sin(a1) * height, z)
            o0 = (i0[0], ybase + rad * height + thick, z)
            o1 = (i1[0], ybase + rad * height + thick, z)
            quad(i0, i1, o1, o0, mat)
            quad((i0[0], i0[1], z - thick), (i1[0], i1[1], z - thick), i1, i0, mat)

    X, Y, Z = 1.6, 1.3, 0.9  # half extents; atrium spans [-X,X] x [-Y/2.., ] ...
    y0, y1 = -0.6, 0.9
    n = 32 * detail
    quad((-X, y0, Z), (X, y0, Z), (X, y0, -Z), (-X, y0, -Z), 0, n, n)              # floor
    quad((-X, y0, -Z), (X, y0, -Z), (X, y1, -Z), (-X, y1, -Z), 1, n, n // 2)       # back wall
    quad((X, y0, Z), (-X, y0, Z), (-X, y1, Z), (X, y1, Z), 1, n, n // 2)           # front wall (behind camera)
    quad((-X, y0, Z), (-X, y0, -Z), (-X, y1, -Z), (-X, y1, Z), 2, n, n // 2)       # left wall
    quad((X, y0, -Z), (X, y0, Z), (X, y1, Z), (X, y1, -Z), 3, n, n // 2)           # right wall
    # ceiling ring with a skylight opening in the middle
    sx, sz = 0.7, 0.35
    quad((-X, y1, -Z), (X, y1, -Z), (X, y1, -sz), (-X, y1, -sz), 1, n, n // 4)
    quad((-X, y1, sz), (X, y1, sz), (X, y1, Z), (-X, y1, Z), 1, n, n // 4)
    quad((-X, y1, -sz), (-sx, y1, -sz), (-sx, y1, sz), (-X, y1, sz), 1, n // 4, n // 4)
    quad((sx, y1, -sz), (X, y1, -sz), (X, y1, sz), (sx, y1, sz), 1, n // 4, n // 4)
    # two storeys of colonnades along both long sides
    ncol = 8
    seg, rings = 24 * detail, 24 * detail
    for storey, (ya, yb) in enumerate(((y0, 0.1), (0.18, 0.8))):
        for side, zc in enumerate((-0.55, 0.55)):
            xs = np.linspace(-1.35, 1.35, ncol)
            for cx in xs:
                column(cx, zc, ya, yb, 0.055 - 0.012 * storey, seg, rings, 4)
            for k in range(ncol - 1):
                arch(xs[k] + 0.05, xs[k + 1] - 0.05, zc, yb - 0.12, 0.6, 0.06, 24 * detail, 1)
        # gallery floor slab of the upper storey
        if storey == 0:
            quad((-X, 0.1, -Z), (X, 0.1, -Z), (X, 0.1, -0.5), (-X, 0.1, -0.5), 1, n, n // 6)
            quad((-X, 0.1, 0.5), (X, 0.1, 0.5), (X, 0.1, Z), (-X, 0.1, Z), 1, n, n // 6)
            quad((-X, 0.18, -0.5), (X, 0.18, -0.5), (X, 0.18, -Z), (-X, 0.18, -Z), 1, n, n // 6)
            quad((-X, 0.18, Z), (X, 0.18, Z), (X, 0.18, 0.5), (-X, 0.18, 0.5), 1, n, n // 6)
    # drapes: sinusoidally folded cloth hanging from the gallery
    for dk, (xc, col) in enumerate(((-0.8, 5), (0.0, 2), (0.8, 3))):
        nu, nv = 72 * detail, 40 * detail
        for iu in range(nu):
            for iv in range(nv):
                def D(u, v):
                    x = xc - 0.3 + 0.6 * u
                    y = 0.1 - 0.55 * v
                    z = -0.45 + 0.04 * np.sin(u * 9 * np.pi + dk) * (0.3 + v)
                    return (x, y, z)
                u0, u1, v0, v1 = iu / nu, (iu + 1) / nu, iv / nv, (iv + 1) / nv
                tris.append((D(u0, v0), D(u1, v0), D(u1, v1), col))
                tris.append((D(u0, v0), D(u1, v1), D(u0, v1), col))

    m = len(tris)
    pos = np.zeros((m, 9), np.float32)
    mat = np.zeros(m, np.int32)
    for k, (a, b, c, mm) in enumerate(tris):
        pos[k, 0:3], pos[k, 3:6], pos[k, 6:9] = a, b, c
        mat[k] = mm
    # face normals with the loader's formula (cs488.h:885-887), in float32
    p0, p1, p2 = pos[:, 0:3], pos[:, 3:6], pos[:, 6:9]
    e0, e1 = p1 - p0, p2 - p0
    cr = np.stack([e0[:, 1] * e1[:, 2] - e0[:, 2] * e1[:, 1], e0[:, 2] * e1[:, 0] - e0[:, 0] * e1[:, 2],
                   e0[:, 0] * e1[:, 1] - e0[:, 1] * e1[:, 0]], axis=1).astype(np.float32)
    ln = np.sqrt(((np.float32(0) + cr[:, 0] * cr[:, 0]) + cr[:, 1] * cr[:, 1]) + cr[:, 2] * cr[:, 2]).astype(np.float32)
    ok = ln > 0
    pos, mat, cr, ln = pos[ok], mat[ok], cr[ok], ln[ok]
    fn = (cr / ln[:, None]).astype(np.float32)
    nrm = np.concatenate([fn, fn, fn], axis=1)
    uv = np.zeros((pos.shape[0], 6), np.float32)
    kd = [(0.55, 0.5, 0.42), (0.72, 0.68, 0.6), (0.6, 0.12, 0.1), (0.12, 0.35, 0.14), (0.7, 0.7, 0.66), (0.15, 0.2, 0.55)]
    mats = [dict(type=0, eta=1.0, Ka=(0, 0, 0), Kd=k, Ks=(0, 0, 0), Ns=5.0, tex_id=-1) for k in kd]
    return pos.reshape(-1), nrm.reshape(-1), uv.reshape(-1), mat, mats, []


SPONZA_LIGHT = ((0.5, 4.0, 1.0, 1000.0, 1000.0, 1000.0),)  # reference main.cpp:24 ("use this for sponza.obj")
SPONZA_EYE, SPONZA_LOOKAT = (1.45, -0.15, 0.0), (-1.0, 0.05, 0.0)  # inside the atrium, down its long axis
