"""ctypes binding of libmipt.so (include/mipt.h) -- the test/bench driver side.

Thin by design: every call goes straight through the C-ABI; nothing here
computes.  A missing or unloadable library raises -- there is no fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIPT_LIB_PATH: developer override (tools/tune_trace.sh benches variant builds); unset = the in-tree library
LIB_PATH = os.environ.get("MIPT_LIB_PATH") or os.path.join(_HERE, "libmipt.so")

MAT_LAMBERTIAN, MAT_METAL, MAT_GLASS, MAT_LIGHT = 0, 1, 2, 3
MODE_PRIMARY, MODE_DIRECT, MODE_PATH = 0, 1, 2
FLT_MAX = float(np.finfo(np.float32).max)


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("eta", C.c_float), ("Ka", C.c_float * 3), ("Kd", C.c_float * 3),
                ("Ks", C.c_float * 3), ("Ns", C.c_float), ("tex_id", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("rgb", C.POINTER(C.c_uint8)), ("width", C.c_int32), ("height", C.c_int32)]


class Light(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("wattage", C.c_float * 3)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32), ("node_min", C.POINTER(C.c_float)), ("node_max", C.POINTER(C.c_float)),
        ("id_hit", C.POINTER(C.c_int32)), ("id_miss", C.POINTER(C.c_int32)),
        ("leaf_first", C.POINTER(C.c_int32)), ("leaf_count", C.POINTER(C.c_int32)),
        ("n_tri_index", C.c_int32), ("tri_index", C.POINTER(C.c_int32)),
        ("n_tris", C.c_int32), ("pos", C.POINTER(C.c_float)), ("nrm", C.POINTER(C.c_float)),
        ("uv", C.POINTER(C.c_float)), ("mat_id", C.POINTER(C.c_int32)),
        ("n_materials", C.c_int32), ("materials", C.POINTER(Material)),
        ("n_textures", C.c_int32), ("textures", C.POINTER(Texture)),
        ("env_width", C.c_int32), ("env_height", C.c_int32), ("env_rgb", C.POINTER(C.c_float)),
        ("n_lights", C.c_int32), ("lights", C.POINTER(Light)),
    ]


class Camera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("eye", C.c_float * 3), ("lookat", C.c_float * 3),
                ("up", C.c_float * 3), ("fov_deg", C.c_float), ("film_size", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("spp", C.c_int32), ("sample_offset", C.c_int32), ("max_bounces", C.c_int32),
                ("seed", C.c_uint32), ("part_index", C.c_int32), ("part_count", C.c_int32),
                ("tile_w", C.c_int32), ("tile_h", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_occluded", C.c_uint64),
                ("launches_closest", C.c_uint64), ("launches_occluded", C.c_uint64), ("launches_other", C.c_uint64),
                ("ms_closest", C.c_double), ("ms_occluded", C.c_double), ("ms_other", C.c_double),
                ("node_visits_closest", C.c_uint64), ("tri_tests_closest", C.c_uint64),
                ("node_visits_occluded", C.c_uint64), ("tri_tests_occluded", C.c_uint64),
                ("launches_shade", C.c_uint64), ("ms_shade", C.c_double)]


# every symbol include/mipt.h declares (tests check the library exports all of them)
SYMBOLS = [
    "mipt_last_error", "mipt_version",
    "mipt_mesh_load_obj", "mipt_mesh_create_single_triangle", "mipt_mesh_create", "mipt_mesh_num_triangles",
    "mipt_mesh_num_materials", "mipt_mesh_num_textures", "mipt_mesh_positions", "mipt_mesh_normals",
    "mipt_mesh_texcoords", "mipt_mesh_material_ids", "mipt_mesh_materials", "mipt_mesh_textures", "mipt_mesh_destroy",
    "mipt_image_load", "mipt_image_create", "mipt_image_width", "mipt_image_height", "mipt_image_pixels",
    "mipt_image_save_png", "mipt_image_save_hdr", "mipt_image_save_pfm", "mipt_image_destroy",
    "mipt_scene_cache_save", "mipt_scene_cache_load", "mipt_scene_cache_save_tagged", "mipt_scene_cache_load_tagged", "mipt_file_fingerprint",
    "mipt_image_pack_rgb8", "mipt_image_compare", "mipt_pack_rgb8_device",
    "mipt_bvh_build", "mipt_bvh_build_all_axes", "mipt_bvh_num_nodes", "mipt_bvh_num_leaves", "mipt_bvh_depth", "mipt_bvh_max_leaf_size",
    "mipt_bvh_node_min", "mipt_bvh_node_max", "mipt_bvh_id_hit", "mipt_bvh_id_miss", "mipt_bvh_leaf_first",
    "mipt_bvh_leaf_count", "mipt_bvh_id_left", "mipt_bvh_id_right", "mipt_bvh_id_parent", "mipt_bvh_id_sibling",
    "mipt_bvh_tri_index", "mipt_bvh_destroy",
    "mipt_scene_desc_fill", "mipt_scene_create", "mipt_scene_create_cached", "mipt_scene_create_multi", "mipt_scene_object_count", "mipt_scene_object_tri_base", "mipt_scene_object_material_base", "mipt_scene_destroy", "mipt_scene_exact_reciprocal", "mipt_scene_wide_nodes", "mipt_scene_wide_hash", "mipt_scene_selfcheck_wide",
    "mipt_context_create", "mipt_context_destroy", "mipt_context_arena_bytes",
    "mipt_trace", "mipt_trace_device", "mipt_trace_device_noalloc", "mipt_occluded", "mipt_eye_rays", "mipt_background",
    "mipt_selftest_fast_division", "mipt_selftest_vec_ops", "mipt_buffer_alloc", "mipt_buffer_free", "mipt_buffer_zero", "mipt_buffer_upload", "mipt_buffer_download",
    "mipt_render_device", "mipt_render", "mipt_context_synchronize", "mipt_context_set_profiling",
    "mipt_context_set_counting", "mipt_context_set_overlap", "mipt_context_get_stats", "mipt_context_reset_stats", "mipt_context_set_traversal",
    "mipt_node_create", "mipt_node_destroy", "mipt_node_num_gpus", "mipt_node_uses_rccl", "mipt_node_render", "mipt_node_synchronize",
    "mipt_node_get_stats", "mipt_render_node",
]

_lib = None


TRAVERSAL_DEFAULT, TRAVERSAL_THREADED, TRAVERSAL_BITTRAIL = 0, 1, 2


class MiptError(RuntimeError):
    pass


def lib():
    """Load libmipt.so (built in-tree by __graft_entry__.build()). Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MiptError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first")
    L = C.CDLL(LIB_PATH)
    vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)
    L.mipt_last_error.restype = C.c_char_p
    L.mipt_version.restype = C.c_char_p
    L.mipt_mesh_load_obj.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.mipt_mesh_create_single_triangle.argtypes = [C.POINTER(vp)]
    L.mipt_mesh_create.argtypes = [C.c_int32, fp, fp, fp, ip, C.c_int32, C.POINTER(Material), C.c_int32,
                                   C.POINTER(Texture), C.POINTER(vp)]
    for n in ("mipt_mesh_num_triangles", "mipt_mesh_num_materials", "mipt_mesh_num_textures", "mipt_image_width",
              "mipt_image_height", "mipt_bvh_num_nodes", "mipt_bvh_num_leaves", "mipt_bvh_depth",
              "mipt_bvh_max_leaf_size"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = C.c_int32
    for n in ("mipt_mesh_positions", "mipt_mesh_normals", "mipt_mesh_texcoords", "mipt_image_pixels",
              "mipt_bvh_node_min", "mipt_bvh_node_max"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = fp
    for n in ("mipt_mesh_material_ids", "mipt_bvh_id_hit", "mipt_bvh_id_miss", "mipt_bvh_leaf_first",
              "mipt_bvh_leaf_count", "mipt_bvh_id_left", "mipt_bvh_id_right", "mipt_bvh_id_parent",
              "mipt_bvh_id_sibling", "mipt_bvh_tri_index"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = ip
    L.mipt_mesh_materials.argtypes = [vp]
    L.mipt_mesh_materials.restype = C.POINTER(Material)
    L.mipt_mesh_textures.argtypes = [vp]
    L.mipt_mesh_textures.restype = C.POINTER(Texture)
    for n in ("mipt_mesh_destroy", "mipt_image_destroy", "mipt_bvh_destroy", "mipt_scene_destroy",
              "mipt_context_destroy"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = None
    L.mipt_image_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.mipt_image_create.argtypes = [C.c_int32, C.c_int32, fp, C.POINTER(vp)]
    for n in ("mipt_image_save_png", "mipt_image_save_hdr", "mipt_image_save_pfm"):
        getattr(L, n).argtypes = [C.c_char_p, C.c_int32, C.c_int32, fp]
    L.mipt_bvh_build.argtypes = [vp, C.POINTER(vp)]
    L.mipt_bvh_build_all_axes.argtypes = [vp, C.POINTER(vp)]
    L.mipt_scene_cache_save.argtypes = [C.c_char_p, vp, vp]
    L.mipt_image_pack_rgb8.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_uint8)]
    L.mipt_image_compare.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t] + [C.POINTER(C.c_double)] * 3
    L.mipt_pack_rgb8_device.argtypes = [vp, vp, C.c_size_t, vp]
    L.mipt_scene_cache_load.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(vp)]
    L.mipt_scene_cache_save_tagged.argtypes = [C.c_char_p, vp, vp, C.c_char_p]
    L.mipt_scene_cache_load_tagged.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp), C.POINTER(vp)]
    L.mipt_file_fingerprint.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.mipt_scene_exact_reciprocal.argtypes = [vp]
    L.mipt_scene_exact_reciprocal.restype = C.c_int32
    L.mipt_scene_wide_nodes.argtypes = [vp]
    L.mipt_scene_wide_nodes.restype = C.c_int32
    L.mipt_scene_selfcheck_wide.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.mipt_scene_selfcheck_wide.restype = C.c_int32
    L.mipt_scene_desc_fill.argtypes = [vp, vp, vp, C.POINTER(Light), C.c_int32, C.POINTER(SceneDesc)]
    L.mipt_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(vp)]
    L.mipt_scene_create_cached.argtypes = [C.POINTER(SceneDesc), C.c_char_p, C.POINTER(vp)]
    L.mipt_scene_wide_hash.argtypes = [vp]
    L.mipt_scene_create_multi.argtypes = [C.POINTER(SceneDesc), C.c_int32, C.c_char_p, C.POINTER(vp)]
    L.mipt_scene_object_count.argtypes = [vp]
    L.mipt_scene_object_tri_base.argtypes = [vp, C.c_int32]
    L.mipt_scene_object_material_base.argtypes = [vp, C.c_int32]
    L.mipt_scene_wide_hash.restype = C.c_uint64
    L.mipt_context_create.argtypes = [vp, C.c_int, vp, C.POINTER(vp)]
    L.mipt_context_arena_bytes.argtypes = [vp]
    L.mipt_context_arena_bytes.restype = C.c_size_t
    L.mipt_trace.argtypes = [vp, fp, fp, C.c_size_t, C.c_float, C.c_float, ip, fp, fp, fp]
    L.mipt_trace_device.argtypes = [vp, vp, C.c_size_t, vp]
    L.mipt_trace_device_noalloc.argtypes = [vp, vp, C.c_size_t, vp]
    L.mipt_occluded.argtypes = [vp, fp, fp, fp, fp, C.c_size_t, C.POINTER(C.c_uint8)]
    L.mipt_eye_rays.argtypes = [vp, C.POINTER(Camera), C.c_size_t, C.c_size_t, fp, fp]
    L.mipt_background.argtypes = [vp, fp, C.c_size_t, fp]
    L.mipt_render_device.argtypes = [vp, C.POINTER(Camera), C.POINTER(RenderParams), vp, vp]
    L.mipt_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(RenderParams), fp]
    L.mipt_context_synchronize.argtypes = [vp]
    L.mipt_selftest_fast_division.argtypes = [vp, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
    L.mipt_selftest_vec_ops.argtypes = [vp, C.c_int, fp, fp, fp, fp, C.c_size_t, fp]
    L.mipt_buffer_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.mipt_buffer_free.argtypes = [vp, vp]
    L.mipt_buffer_zero.argtypes = [vp, vp, C.c_size_t]
    L.mipt_buffer_upload.argtypes = [vp, vp, vp, C.c_size_t]
    L.mipt_buffer_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.mipt_context_set_profiling.argtypes = [vp, C.c_int]
    L.mipt_context_set_counting.argtypes = [vp, C.c_int]
    L.mipt_context_set_overlap.argtypes = [vp, C.c_int]
    L.mipt_context_set_traversal.argtypes = [vp, C.c_int]
    L.mipt_context_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.mipt_context_reset_stats.argtypes = [vp]
    L.mipt_node_create.argtypes = [vp, C.c_int, ip, C.POINTER(vp)]
    L.mipt_node_destroy.argtypes = [vp]
    L.mipt_node_destroy.restype = None
    L.mipt_node_num_gpus.argtypes = [vp]
    L.mipt_node_uses_rccl.argtypes = [vp]
    L.mipt_node_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(RenderParams), fp]
    L.mipt_node_synchronize.argtypes = [vp]
    L.mipt_node_get_stats.argtypes = [vp, C.POINTER(Stats), C.POINTER(C.c_double)]
    L.mipt_render_node.argtypes = [vp, C.POINTER(Camera), C.POINTER(RenderParams), C.c_int, fp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise MiptError(f"mipt error {rc}: {lib().mipt_last_error().decode(errors='replace')}")


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def _view(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


def default_camera(width, height):
    """camera globals of the reference, C:66-69, C:81-83"""
    cam = Camera()
    cam.width, cam.height = width, height
    cam.eye[:] = (0.0, 0.0, 1.5)
    cam.lookat[:] = (0.0, 0.0, 0.0)
    cam.up[:] = (0.0, 1.0, 0.0)
    cam.fov_deg = 45.0
    cam.film_size = 0.032
    return cam


def make_camera(width, height, eye, lookat, up=(0.0, 1.0, 0.0), fov_deg=45.0, film_size=0.032):
    cam = Camera()
    cam.width, cam.height = width, height
    cam.eye[:] = eye
    cam.lookat[:] = lookat
    cam.up[:] = up
    cam.fov_deg = fov_deg
    cam.film_size = film_size
    return cam


class Mesh:
    """TriangleMesh of the reference (C:629-633) behind mipt_mesh_*"""

    def __init__(self, handle):
        self.h = handle

    @staticmethod
    def load(path):
        h = C.c_void_p()
        check(lib().mipt_mesh_load_obj(os.fsencode(path), C.byref(h)))
        return Mesh(h)

    @staticmethod
    def single_triangle():
        h = C.c_void_p()
        check(lib().mipt_mesh_create_single_triangle(C.byref(h)))
        return Mesh(h)

    @staticmethod
    def from_arrays(pos, nrm, uv=None, mat_id=None, materials=None, textures=None):
        """materials: list of dict(type, eta, Ka, Kd, Ks, Ns, tex_id); textures: list of (h,w,3) uint8 arrays"""
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1)
        nrm = np.ascontiguousarray(nrm, dtype=np.float32).reshape(-1)
        n = pos.size // 9
        uv = None if uv is None else np.ascontiguousarray(uv, dtype=np.float32).reshape(-1)
        mat_id = None if mat_id is None else np.ascontiguousarray(mat_id, dtype=np.int32)
        materials = materials or []
        textures = [np.ascontiguousarray(t, dtype=np.uint8) for t in (textures or [])]
        cm = (Material * max(1, len(materials)))()
        for i, m in enumerate(materials):
            cm[i].type = int(m.get("type", 0))
            cm[i].eta = float(m.get("eta", 1.0))
            cm[i].Ka[:] = m.get("Ka", (0, 0, 0))
            cm[i].Kd[:] = m.get("Kd", (0.9, 0.9, 0.9))
            cm[i].Ks[:] = m.get("Ks", (0, 0, 0))
            cm[i].Ns = float(m.get("Ns", 0.0))
            cm[i].tex_id = int(m.get("tex_id", -1))
        ct = (Texture * max(1, len(textures)))()
        for i, t in enumerate(textures):
            ct[i].rgb = t.ctypes.data_as(C.POINTER(C.c_uint8))
            ct[i].height, ct[i].width = t.shape[0], t.shape[1]
        h = C.c_void_p()
        check(lib().mipt_mesh_create(n, _fp(pos), _fp(nrm), _fp(uv), _ip(mat_id), len(materials), cm, len(textures),
                                     ct, C.byref(h)))
        return Mesh(h)

    @property
    def num_triangles(self):
        return lib().mipt_mesh_num_triangles(self.h)

    @property
    def num_materials(self):
        return lib().mipt_mesh_num_materials(self.h)

    def positions(self):
        return _view(lib().mipt_mesh_positions(self.h), 9 * self.num_triangles, np.float32)

    def normals(self):
        return _view(lib().mipt_mesh_normals(self.h), 9 * self.num_triangles, np.float32)

    def texcoords(self):
        return _view(lib().mipt_mesh_texcoords(self.h), 6 * self.num_triangles, np.float32)

    def material_ids(self):
        return _view(lib().mipt_mesh_material_ids(self.h), self.num_triangles, np.int32)

    def materials(self):
        p = lib().mipt_mesh_materials(self.h)
        out = []
        for i in range(self.num_materials):
            m = p[i]
            out.append(dict(type=m.type, eta=m.eta, Ka=tuple(m.Ka), Kd=tuple(m.Kd), Ks=tuple(m.Ks), Ns=m.Ns,
                            tex_id=m.tex_id))
        return out

    def textures(self):
        p = lib().mipt_mesh_textures(self.h)
        out = []
        for i in range(lib().mipt_mesh_num_textures(self.h)):
            t = p[i]
            out.append(np.ctypeslib.as_array(t.rgb, shape=(t.height, t.width, 3)).copy())
        return out

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mipt_mesh_destroy(self.h)
            self.h = None


class Image:
    def __init__(self, handle):
        self.h = handle

    @staticmethod
    def load(path):
        h = C.c_void_p()
        check(lib().mipt_image_load(os.fsencode(path), C.byref(h)))
        return Image(h)

    @staticmethod
    def from_array(rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        hh, ww = rgb.shape[0], rgb.shape[1]
        h = C.c_void_p()
        check(lib().mipt_image_create(ww, hh, _fp(rgb), C.byref(h)))
        return Image(h)

    @property
    def width(self):
        return lib().mipt_image_width(self.h)

    @property
    def height(self):
        return lib().mipt_image_height(self.h)

    def pixels(self):
        return _view(lib().mipt_image_pixels(self.h), 3 * self.width * self.height, np.float32).reshape(
            self.height, self.width, 3)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mipt_image_destroy(self.h)
            self.h = None


def save_png(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    check(lib().mipt_image_save_png(os.fsencode(path), rgb.shape[1], rgb.shape[0], _fp(rgb)))


def pack_rgb8(rgb):
    """float image -> uint8 as Image::save converts it (no flip)"""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.empty(rgb.shape, np.uint8)
    check(lib().mipt_image_pack_rgb8(_fp(rgb), rgb.size, out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


def compare_images(a, b):
    """-> (rmse, mean_a, mean_b) over all channels, accumulated in double"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape
    r, ma, mb = C.c_double(), C.c_double(), C.c_double()
    check(lib().mipt_image_compare(_fp(a), _fp(b), a.size, C.byref(r), C.byref(ma), C.byref(mb)))
    return r.value, ma.value, mb.value


def save_hdr(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    check(lib().mipt_image_save_hdr(os.fsencode(path), rgb.shape[1], rgb.shape[0], _fp(rgb)))


def save_pfm(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    check(lib().mipt_image_save_pfm(os.fsencode(path), rgb.shape[1], rgb.shape[0], _fp(rgb)))


class BVH:
    """StacklessBVH of the reference (C:1641-1660) behind mipt_bvh_*"""

    def __init__(self, mesh, handle=None, all_axes=False):
        self.mesh = mesh
        if handle is None:
            handle = C.c_void_p()
            check((lib().mipt_bvh_build_all_axes if all_axes else lib().mipt_bvh_build)(mesh.h, C.byref(handle)))
        self.h = handle

    @property
    def num_nodes(self):
        return lib().mipt_bvh_num_nodes(self.h)

    @property
    def num_leaves(self):
        return lib().mipt_bvh_num_leaves(self.h)

    @property
    def depth(self):
        return lib().mipt_bvh_depth(self.h)

    @property
    def max_leaf_size(self):
        return lib().mipt_bvh_max_leaf_size(self.h)

    def arrays(self):
        L, n = lib(), self.num_nodes
        d = dict(node_min=_view(L.mipt_bvh_node_min(self.h), 3 * n, np.float32),
                 node_max=_view(L.mipt_bvh_node_max(self.h), 3 * n, np.float32))
        for k in ("id_hit", "id_miss", "leaf_first", "leaf_count", "id_left", "id_right", "id_parent", "id_sibling"):
            d[k] = _view(getattr(L, "mipt_bvh_" + k)(self.h), n, np.int32)
        d["tri_index"] = _view(L.mipt_bvh_tri_index(self.h), self.mesh.num_triangles, np.int32)
        return d

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mipt_bvh_destroy(self.h)
            self.h = None


def save_scene_cache(path, mesh, bvh, tag=None):
    """flat binary cache of a loaded mesh and its built BVH (mipt_scene_cache_save[_tagged])"""
    if tag is None:
        check(lib().mipt_scene_cache_save(os.fsencode(path), mesh.h, bvh.h))
    else:
        check(lib().mipt_scene_cache_save_tagged(os.fsencode(path), mesh.h, bvh.h, tag.encode()))


def load_scene_cache(path, expect_tag=None):
    """-> (Mesh, BVH) as they were saved; raises on a missing, truncated or corrupted file, and -- when expect_tag is
    given -- on a cache that was written with another source tag"""
    mh, bh = C.c_void_p(), C.c_void_p()
    if expect_tag is None:
        check(lib().mipt_scene_cache_load(os.fsencode(path), C.byref(mh), C.byref(bh)))
    else:
        check(lib().mipt_scene_cache_load_tagged(os.fsencode(path), expect_tag.encode(), C.byref(mh), C.byref(bh)))
    mesh = Mesh(mh)
    return mesh, BVH(mesh, bh)


def file_fingerprint(path):
    buf = C.create_string_buffer(512)
    check(lib().mipt_file_fingerprint(os.fsencode(path), buf, 512))
    return buf.value.decode()


class Scene:
    """Scene of the reference (C:2004-2036): objects, lights, background."""

    def __init__(self, mesh, bvh=None, env=None, lights=((3.0, 3.0, 3.0, 1000.0, 1000.0, 1000.0),), wide_cache=None):
        """mesh: a Mesh, or a list of them = Scene::addObject once per mesh (cs488.h:2020; bvh then None or a list);
        wide_cache: path of a file that keeps the library's own hierarchy above the leaves (mipt_scene_create_cached)"""
        meshes = list(mesh) if isinstance(mesh, (list, tuple)) else [mesh]
        bvhs = list(bvh) if isinstance(bvh, (list, tuple)) else [bvh] + [None] * (len(meshes) - 1)
        self.meshes = meshes
        self.bvhs = [b if b is not None else BVH(m) for m, b in zip(meshes, bvhs)]
        self.mesh, self.bvh = self.meshes[0], self.bvhs[0]
        self.env = env
        self.lights = (Light * max(1, len(lights)))()
        for i, l in enumerate(lights):
            self.lights[i].position[:] = l[0:3]
            self.lights[i].wattage[:] = l[3:6]
        self.descs = (SceneDesc * len(meshes))()
        for k, (m, b) in enumerate(zip(self.meshes, self.bvhs)):
            check(lib().mipt_scene_desc_fill(m.h, b.h, env.h if env is not None else None, self.lights, len(lights),
                                             C.byref(self.descs[k])))
        self.desc = self.descs[0]
        h = C.c_void_p()
        if len(meshes) > 1:
            check(lib().mipt_scene_create_multi(self.descs, len(meshes), os.fsencode(wide_cache) if wide_cache else None, C.byref(h)))
        elif wide_cache:
            check(lib().mipt_scene_create_cached(C.byref(self.desc), os.fsencode(wide_cache), C.byref(h)))
        else:
            check(lib().mipt_scene_create(C.byref(self.desc), C.byref(h)))
        self.h = h

    @property
    def object_count(self):
        return int(lib().mipt_scene_object_count(self.h))

    def object_tri_base(self, k):
        """scene-wide id of object k's triangle 0 (mipt_scene_object_tri_base)"""
        return int(lib().mipt_scene_object_tri_base(self.h, k))

    def object_material_base(self, k):
        return int(lib().mipt_scene_object_material_base(self.h, k))

    @property
    def wide_hash(self):
        """FNV-1a of the wide tree's arrays as the GPU reads them (mipt_scene_wide_hash)"""
        return int(lib().mipt_scene_wide_hash(self.h))

    @property
    def exact_reciprocal(self):
        """True when the scene's box planes allow the traversal kernel's exact reciprocal slab division"""
        return bool(lib().mipt_scene_exact_reciprocal(self.h))

    @property
    def wide_nodes(self):
        """nodes of the 4-wide tree of the order-free traversal kernel; 0 = this scene takes the literal walk"""
        return int(lib().mipt_scene_wide_nodes(self.h))

    def selfcheck_wide(self):
        """(leaves, depth) of the 4-wide tree after checking that it reaches every leaf once and that every stored box
        contains what lies below it (mipt_scene_selfcheck_wide; host only); raises MiptError otherwise"""
        leaves, depth = C.c_int32(), C.c_int32()
        check(lib().mipt_scene_selfcheck_wide(self.h, C.byref(leaves), C.byref(depth)))
        return leaves.value, depth.value

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mipt_scene_destroy(self.h)
            self.h = None


class Context:
    """One GPU context: the uploaded arena + the kernels (mipt_context_*, mipt_trace, mipt_render...)."""

    def __init__(self, scene, device=0, stream=None):
        self.scene = scene
        h = C.c_void_p()
        check(lib().mipt_context_create(scene.h, int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h

    @property
    def arena_bytes(self):
        return lib().mipt_context_arena_bytes(self.h)

    def trace(self, ray_o, ray_d, tmin=0.0, tmax=FLT_MAX, want_hitrec=False):
        o = np.ascontiguousarray(ray_o, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(ray_d, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        tri = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        bary = np.empty((n, 2), np.float32)
        rec = np.empty((n, 10), np.float32) if want_hitrec else None
        check(lib().mipt_trace(self.h, _fp(o), _fp(d), n, tmin, tmax, _ip(tri), _fp(t), _fp(bary), _fp(rec)))
        return (tri, t, bary, rec) if want_hitrec else (tri, t, bary)

    def trace_device(self, rays_ptr, n, hits_ptr, noalloc=False):
        """device pointers, asynchronous on the context's stream: rays n x {o.xyz, tmin, d.xyz, tmax}, hits n x
        {tri_id, t, beta, gamma}; noalloc: the entry point that takes its scratch from the library's static array"""
        f = lib().mipt_trace_device_noalloc if noalloc else lib().mipt_trace_device
        check(f(self.h, C.c_void_p(rays_ptr), n, C.c_void_p(hits_ptr)))

    def occluded(self, ray_o, ray_d, tmin, tmax):
        o = np.ascontiguousarray(ray_o, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(ray_d, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, np.float32), (n,)))
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)))
        occ = np.empty(n, np.uint8)
        check(lib().mipt_occluded(self.h, _fp(o), _fp(d), _fp(tmin), _fp(tmax), n,
                                  occ.ctypes.data_as(C.POINTER(C.c_uint8))))
        return occ

    def eye_rays(self, cam, first=0, count=None):
        if count is None:
            count = cam.width * cam.height - first
        o = np.empty((count, 3), np.float32)
        d = np.empty((count, 3), np.float32)
        check(lib().mipt_eye_rays(self.h, C.byref(cam), first, count, _fp(o), _fp(d)))
        return o, d

    def background(self, dirs):
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        rgb = np.empty_like(d)
        check(lib().mipt_background(self.h, _fp(d), d.shape[0], _fp(rgb)))
        return rgb

    def render(self, cam, mode=MODE_PATH, spp=1, sample_offset=0, max_bounces=-1, seed=1, part_index=0, part_count=1,
               tile_w=0, tile_h=0):
        p = RenderParams(mode, spp, sample_offset, max_bounces, seed, part_index, part_count, tile_w, tile_h)
        rgb = np.zeros((cam.height, cam.width, 3), np.float32)
        check(lib().mipt_render(self.h, C.byref(cam), C.byref(p), _fp(rgb)))
        return rgb

    def render_device(self, cam, params, accum_ptr, rgb_ptr):
        check(lib().mipt_render_device(self.h, C.byref(cam), C.byref(params), C.c_void_p(accum_ptr),
                                       C.c_void_p(rgb_ptr)))

    def synchronize(self):
        check(lib().mipt_context_synchronize(self.h))

    def selftest_fast_division(self, n_pairs, seed=1):
        bad = C.c_uint64(0)
        check(lib().mipt_selftest_fast_division(self.h, n_pairs, seed, C.byref(bad)))
        return bad.value

    def vec_ops(self, op, a, b=None, c=None, d=None):
        """pt/vec.hpp's primitives on this context's GPU (see vec_ops below)"""
        return vec_ops(op, a, b, c, d, ctx=self)

    def buffer_alloc(self, nbytes, zero=True):
        p = C.c_void_p()
        check(lib().mipt_buffer_alloc(self.h, nbytes, C.byref(p)))
        if zero:
            check(lib().mipt_buffer_zero(self.h, p, nbytes))
        return p.value

    def buffer_free(self, ptr):
        check(lib().mipt_buffer_free(self.h, C.c_void_p(ptr)))

    def buffer_upload(self, ptr, array):
        a = np.ascontiguousarray(array)
        check(lib().mipt_buffer_upload(self.h, C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def buffer_download(self, ptr, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        check(lib().mipt_buffer_download(self.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes))
        return out

    def set_profiling(self, on):
        check(lib().mipt_context_set_profiling(self.h, int(on)))

    def pack_rgb8_device(self, rgb_ptr, n_values, out_ptr):
        """device float rgb -> device uint8, asynchronous on the context's stream (mipt_pack_rgb8_device)"""
        check(lib().mipt_pack_rgb8_device(self.h, C.c_void_p(rgb_ptr), n_values, C.c_void_p(out_ptr)))

    def set_traversal(self, traversal):
        """TRAVERSAL_DEFAULT (order-free wide tree), TRAVERSAL_THREADED (literal idHit/idMiss walk) or
        TRAVERSAL_BITTRAIL (StacklessBVH2::intersect, literal)"""
        check(lib().mipt_context_set_traversal(self.h, int(traversal)))

    def set_overlap(self, on):
        """path mode: shadow rays of bounce k beside the extension rays of bounce k+1 on a side stream (default on)"""
        check(lib().mipt_context_set_overlap(self.h, int(on)))

    def set_counting(self, on):
        check(lib().mipt_context_set_counting(self.h, int(on)))

    def stats(self):
        s = Stats()
        check(lib().mipt_context_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def reset_stats(self):
        check(lib().mipt_context_reset_stats(self.h))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mipt_context_destroy(self.h)
            self.h = None


class Node:
    """One frame over the GPUs of a node (mipt_node_*): a context, stream and host thread per GPU, tiles gathered to
    GPU 0 with one RCCL collective per call.  devices: list of device ids (repeats = ranks sharing a GPU)."""

    def __init__(self, scene, ngpus=None, devices=None):
        self.scene = scene
        if devices is not None:
            devs = np.ascontiguousarray(devices, dtype=np.int32)
            ngpus = int(devs.size)
        else:
            devs = None
        h = C.c_void_p()
        check(lib().mipt_node_create(scene.h, int(ngpus), _ip(devs), C.byref(h)))
        self.h = h

    @property
    def num_gpus(self):
        return lib().mipt_node_num_gpus(self.h)

    @property
    def uses_rccl(self):
        return bool(lib().mipt_node_uses_rccl(self.h))

    def render(self, cam, mode=MODE_PATH, spp=1, sample_offset=0, max_bounces=-1, seed=1, tile_w=0, tile_h=0, readback=True):
        p = RenderParams(mode, spp, sample_offset, max_bounces, seed, 0, 1, tile_w, tile_h)
        rgb = np.zeros((cam.height, cam.width, 3), np.float32) if readback else None
        check(lib().mipt_node_render(self.h, C.byref(cam), C.byref(p), _fp(rgb)))
        return rgb

    def synchronize(self):
        """waits for what render(readback=False) left enqueued"""
        check(lib().mipt_node_synchronize(self.h))

    def stats(self):
        s, ms = Stats(), C.c_double()
        check(lib().mipt_node_get_stats(self.h, C.byref(s), C.byref(ms)))
        d = {k: getattr(s, k) for k, _ in Stats._fields_}
        d["last_frame_ms"] = ms.value
        return d

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mipt_node_destroy(self.h)
            self.h = None


def render_node(scene, cam, ngpus, mode=MODE_PATH, spp=1, sample_offset=0, max_bounces=-1, seed=1, tile_w=0, tile_h=0):
    """mipt_render_node: node over GPUs 0..ngpus-1, one frame, host image"""
    p = RenderParams(mode, spp, sample_offset, max_bounces, seed, 0, 1, tile_w, tile_h)
    rgb = np.zeros((cam.height, cam.width, 3), np.float32)
    check(lib().mipt_render_node(scene.h, C.byref(cam), C.byref(p), int(ngpus), _fp(rgb)))
    return rgb


VEC_OUT_WIDTH = (1, 3, 3, 1, 3, 1, 2, 3)  # floats per element: dot cross normalize det3 mul length2 normalize2 div


def vec_ops(op, a, b=None, c=None, d=None, ctx=None):
    """mipt_selftest_vec_ops: pt/vec.hpp's primitives over (n, 3) float32 arrays; ctx None = the host build"""
    arrs = [None if x is None else np.ascontiguousarray(x, np.float32).reshape(-1, 3) for x in (a, b, c, d)]
    n = arrs[0].shape[0]
    out = np.empty((n, VEC_OUT_WIDTH[op]), np.float32)
    fp = C.POINTER(C.c_float)
    ptrs = [x.ctypes.data_as(fp) if x is not None else None for x in arrs]
    check(lib().mipt_selftest_vec_ops(ctx.h if ctx is not None else None, op, *ptrs, n, out.ctypes.data_as(fp)))
    return out[:, 0] if out.shape[1] == 1 else out
