"""path-tracer_amd -- MI355X-native ray/scene hot path of flamebrawler/Path-tracer.

The product is csrc/ (HIP kernels for gfx950 + C++ host library) behind the
C-ABI of include/mipt.h, built in-tree as libmipt.so; `binding` is the thin
ctypes driver the tests and bench.py use.  The directory name has a hyphen:
import it with importlib.import_module("path-tracer_amd").
"""
from . import binding, tiles  # noqa: F401
from .binding import (BVH, Camera, Context, Image, Mesh, MiptError, RenderParams, Scene,  # noqa: F401
                      default_camera, make_camera, save_hdr, save_pfm, save_png)
