"""ctypes binding of oracle/liboracle.so (pt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, as the checker / the reported CPU baseline.
The product (path-tracer_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
SRC_PATH = os.path.join(_HERE, "pt_oracle.c")

THREADED, BITTRAIL, RECURSIVE, BRUTEFORCE = 0, 1, 2, 3
FLT_MAX = float(np.finfo(np.float32).max)


def build(force=False):
    """gcc -O2 -ffp-contract=off: no FMA contraction, IEEE divide/sqrt, baseline x86-64."""
    if force or not os.path.exists(LIB_PATH) or (
            os.path.exists(SRC_PATH) and os.path.getmtime(SRC_PATH) > os.path.getmtime(LIB_PATH)):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", LIB_PATH, SRC_PATH, "-lm"])
    return LIB_PATH


def build_ref():
    """oracle/_ref: the reference's linalg.h and stb_image.h, unmodified, by include path (oracle/Makefile `ref`); a
    no-op where /root/reference is absent (the GPU box uses the prebuilt files, or the committed vectors)"""
    subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


class Camera(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("eye", C.c_float * 3), ("up", C.c_float * 3),
                ("viewDir", C.c_float * 3), ("aspect", C.c_float), ("filmSize", C.c_float),
                ("distanceToFilm", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.orc_scene_create.restype = vp
        L.orc_scene_create.argtypes = [C.c_int, fp, fp, fp, ip]
        L.orc_scene_set_materials.argtypes = [vp, C.c_int, fp, C.POINTER(C.c_void_p), ip, ip]
        L.orc_scene_set_env.argtypes = [vp, C.c_int, C.c_int, fp]
        L.orc_scene_set_lights.argtypes = [vp, C.c_int, fp]
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_scene_add_object.argtypes = [vp, vp]
        L.orc_build.argtypes = [vp]
        L.orc_bvh_stats.argtypes = [vp, ip]
        L.orc_bvh_export.argtypes = [vp, fp, fp, ip, ip, ip, ip, ip, ip, ip, ip, ip]
        L.orc_make_camera.argtypes = [C.POINTER(Camera), C.c_int, C.c_int, fp, fp, fp, C.c_float, C.c_float]
        L.orc_eye_rays.argtypes = [C.POINTER(Camera), C.c_longlong, C.c_longlong, fp, fp]
        L.orc_intersect.argtypes = [vp, C.c_int, C.c_longlong, fp, fp, C.c_float, C.c_float, ip, fp, fp, fp,
                                    C.POINTER(C.c_longlong)]
        L.orc_occluded.argtypes = [vp, C.c_int, C.c_longlong, fp, fp, fp, fp, C.POINTER(C.c_ubyte)]
        L.orc_background.argtypes = [vp, C.c_longlong, fp, fp]
        L.orc_vec_ops.argtypes = [C.c_int, fp, fp, fp, fp, C.c_longlong, fp]
        L.orc_pcg32_rand.restype = C.c_float
        L.orc_counter_u32.restype = C.c_uint32
        L.orc_counter_u32.argtypes = [C.c_uint32] * 4
        L.orc_raytrace.restype = C.c_longlong
        L.orc_raytrace.argtypes = [vp, C.POINTER(Camera), C.c_int, C.c_int, C.c_int, fp]
        L.orc_pathtrace.restype = C.c_longlong
        L.orc_pathtrace.argtypes = [vp, C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_uint, fp, fp]
        assert L.orc_sizeof_camera() == C.sizeof(Camera)
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int)) if a is not None else None


def make_camera(width, height, eye=(0.0, 0.0, 1.5), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov_deg=45.0,
                film=0.032):
    cam = Camera()
    e = np.asarray(eye, np.float32)
    l = np.asarray(lookat, np.float32)
    u = np.asarray(up, np.float32)
    lib().orc_make_camera(C.byref(cam), width, height, _fp(e), _fp(l), _fp(u), fov_deg, film)
    return cam


class OracleScene:
    """materials: list of dicts (type, eta, Ka, Kd, Ks, Ns, tex_id); textures: list of (h,w,3) uint8"""

    def __init__(self, pos, nrm, uv, mat_id, materials, textures=(), env=None, lights=((3, 3, 3, 1000, 1000, 1000),)):
        L = lib()
        self.pos = np.ascontiguousarray(pos, np.float32).reshape(-1)
        self.nrm = np.ascontiguousarray(nrm, np.float32).reshape(-1)
        self.uv = np.ascontiguousarray(uv, np.float32).reshape(-1)
        self.mat_id = np.ascontiguousarray(mat_id, np.int32)
        self.ntris = self.pos.size // 9
        self.h = C.c_void_p(L.orc_scene_create(self.ntris, _fp(self.pos), _fp(self.nrm), _fp(self.uv),
                                               _ip(self.mat_id)))
        nm = len(materials)
        params = np.zeros((nm, 12), np.float32)
        texs = [np.ascontiguousarray(t, np.uint8) for t in textures]
        tp = (C.c_void_p * nm)()
        tw = np.zeros(nm, np.int32)
        th = np.zeros(nm, np.int32)
        for i, m in enumerate(materials):
            params[i, 0] = m["type"]
            params[i, 1] = m["eta"]
            params[i, 2:5] = m["Ka"]
            params[i, 5:8] = m["Kd"]
            params[i, 8:11] = m["Ks"]
            params[i, 11] = m["Ns"]
            tid = m.get("tex_id", -1)
            if tid >= 0:
                tp[i] = texs[tid].ctypes.data
                th[i], tw[i] = texs[tid].shape[0], texs[tid].shape[1]
        L.orc_scene_set_materials(self.h, nm, _fp(params), tp, _ip(tw), _ip(th))
        if env is not None:
            self.set_env(env)
        self.set_lights(lights)
        L.orc_build(self.h)

    def add_object(self, other):
        """Scene::addObject (cs488.h:2020): `other` (an OracleScene of its own mesh, built) becomes the next object of
        this scene; triangle and material ids reported by intersect() then count through all objects in order."""
        self.objects = getattr(self, "objects", [])
        self.objects.append(other)  # keeps it alive: the scene does not own it
        lib().orc_scene_add_object(self.h, other.h)
        return self

    def set_env(self, env):
        env = np.ascontiguousarray(env, np.float32)
        lib().orc_scene_set_env(self.h, env.shape[1], env.shape[0], _fp(env))

    def set_lights(self, lights):
        a = np.ascontiguousarray(np.asarray(lights, np.float32).reshape(-1, 6))
        lib().orc_scene_set_lights(self.h, a.shape[0], _fp(a))

    def stats(self):
        s = np.zeros(5, np.int32)
        lib().orc_bvh_stats(self.h, _ip(s))
        return dict(nodes=int(s[0]), leaves=int(s[1]), depth=int(s[2]), max_leaf=int(s[3]), tri_refs=int(s[4]))

    def export_bvh(self):
        n = self.stats()["nodes"]
        d = dict(node_min=np.zeros(3 * n, np.float32), node_max=np.zeros(3 * n, np.float32))
        for k in ("id_hit", "id_miss", "leaf_first", "leaf_count", "id_left", "id_right", "id_parent", "id_sibling"):
            d[k] = np.zeros(n, np.int32)
        d["tri_index"] = np.zeros(self.ntris, np.int32)
        lib().orc_bvh_export(self.h, _fp(d["node_min"]), _fp(d["node_max"]), _ip(d["id_hit"]), _ip(d["id_miss"]),
                             _ip(d["leaf_first"]), _ip(d["leaf_count"]), _ip(d["id_left"]), _ip(d["id_right"]),
                             _ip(d["id_parent"]), _ip(d["id_sibling"]), _ip(d["tri_index"]))
        return d

    def eye_rays(self, cam, first=0, count=None):
        if count is None:
            count = cam.width * cam.height - first
        o = np.empty((count, 3), np.float32)
        d = np.empty((count, 3), np.float32)
        lib().orc_eye_rays(C.byref(cam), first, count, _fp(o), _fp(d))
        return o, d

    def intersect(self, o, d, tmin=0.0, tmax=FLT_MAX, variant=THREADED, want_hitrec=False):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = o.shape[0]
        tri = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        bary = np.empty((n, 2), np.float32)
        rec = np.empty((n, 10), np.float32) if want_hitrec else None
        counts = (C.c_longlong * 2)(0, 0)
        lib().orc_intersect(self.h, variant, n, _fp(o), _fp(d), tmin, tmax, _ip(tri), _fp(t), _fp(bary), _fp(rec),
                            counts)
        self.last_counts = (counts[0], counts[1])
        return (tri, t, bary, rec) if want_hitrec else (tri, t, bary)

    def occluded(self, o, d, tmin, tmax, variant=THREADED):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = o.shape[0]
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, np.float32), (n,)))
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)))
        occ = np.empty(n, np.uint8)
        lib().orc_occluded(self.h, variant, n, _fp(o), _fp(d), _fp(tmin), _fp(tmax),
                           occ.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return occ

    def background(self, dirs):
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        rgb = np.empty_like(d)
        lib().orc_background(self.h, d.shape[0], _fp(d), _fp(rgb))
        return rgb

    def raytrace(self, cam, variant=THREADED, rows=None):
        rgb = np.zeros((cam.height, cam.width, 3), np.float32)
        r0, r1 = rows if rows else (0, cam.height)
        self.last_rays = lib().orc_raytrace(self.h, C.byref(cam), variant, r0, r1, _fp(rgb))
        return rgb

    def pathtrace(self, cam, spp=1, sample_offset=0, max_bounces=-1, rng_mode=1, seed=1, variant=THREADED, rows=None,
                  accum=None):
        """rng_mode 0 = the reference's global PCG32 stream (call pcg32_reset() first), 1 = counter RNG"""
        if accum is None:
            accum = np.zeros((cam.height, cam.width, 3), np.float32)
        rgb = np.zeros((cam.height, cam.width, 3), np.float32)
        r0, r1 = rows if rows else (0, cam.height)
        self.last_rays = lib().orc_pathtrace(self.h, C.byref(cam), variant, r0, r1, sample_offset, spp, max_bounces,
                                             rng_mode, seed, _fp(accum), _fp(rgb))
        self.last_accum = accum
        return rgb

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.orc_scene_destroy(self.h)
            self.h = None


def pcg32_reset():
    lib().orc_pcg32_reset()


def fnv1a32_of_bits(values_u32):
    """FNV-1a-32 over 32-bit words as SURVEY 8(c) defines it: h = (h ^ bits) * 16777619, init 2166136261"""
    h = 2166136261
    for b in values_u32.tolist():
        h = ((h ^ b) * 16777619) & 0xFFFFFFFF
    return h


VEC_OUT_WIDTH = (1, 3, 3, 1, 3, 1, 2, 3)  # floats per element: dot cross normalize det3 mul length2 normalize2 div


def vec_ops(op, a, b=None, c=None, d=None, fn=None):
    """orc_vec_ops (the linalg restatement at the top of pt_oracle.c) over (n, 3) float32 arrays; fn: another library's
    function of the same signature (oracle/_ref/libref_linalg.so: ref_vec_ops)"""
    arrs = [None if x is None else np.ascontiguousarray(x, np.float32).reshape(-1, 3) for x in (a, b, c, d)]
    n = arrs[0].shape[0]
    out = np.empty((n, VEC_OUT_WIDTH[op]), np.float32)
    fp = C.POINTER(C.c_float)
    ptrs = [x.ctypes.data_as(fp) if x is not None else None for x in arrs]
    rc = (fn or lib().orc_vec_ops)(op, *ptrs, n, out.ctypes.data_as(fp))
    assert rc == 0
    return out[:, 0] if out.shape[1] == 1 else out


REF_DIR = os.path.join(_HERE, "_ref")


def ref_linalg():
    """oracle/_ref/libref_linalg.so: the reference's unmodified linalg.h behind ref_vec_ops (oracle/Makefile builds it
    where /root/reference is present); None when it is not on this machine"""
    p = os.path.join(REF_DIR, "libref_linalg.so")
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    fp = C.POINTER(C.c_float)
    L.ref_vec_ops.argtypes = [C.c_int, fp, fp, fp, fp, C.c_longlong, fp]
    return L


def ref_stb():
    """oracle/_ref/libref_stb.so: the reference's unmodified stb_image.h (stbi_loadf / stbi_load with 3 components)"""
    p = os.path.join(REF_DIR, "libref_stb.so")
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    ip = C.POINTER(C.c_int)
    L.ref_stbi_loadf_rgb.restype = C.POINTER(C.c_float)
    L.ref_stbi_loadf_rgb.argtypes = [C.c_char_p, ip, ip, ip]
    L.ref_stbi_load_rgb.restype = C.POINTER(C.c_ubyte)
    L.ref_stbi_load_rgb.argtypes = [C.c_char_p, ip, ip, ip]
    L.ref_stbi_free.argtypes = [C.c_void_p]
    return L


def ref_stb_loadf(L, path):
    """stbi_loadf(path, &w, &h, &comp, 3) -> (h, w, 3) float32, rows in FILE order (top row first)"""
    w, h, comp = C.c_int(), C.c_int(), C.c_int()
    p = L.ref_stbi_loadf_rgb(path.encode(), C.byref(w), C.byref(h), C.byref(comp))
    if not p:
        return None
    out = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    L.ref_stbi_free(p)
    return out


def ref_stb_load8(L, path):
    w, h, comp = C.c_int(), C.c_int(), C.c_int()
    p = L.ref_stbi_load_rgb(path.encode(), C.byref(w), C.byref(h), C.byref(comp))
    if not p:
        return None
    out = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    L.ref_stbi_free(p)
    return out
