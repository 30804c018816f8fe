#!/usr/bin/env python3
"""tools/tail_diag.py for a FRAME (a -DPT_DIAG_TAIL=1 variant, MIPT_LIB_PATH): renders one rank's share of the bench frame
and prints, for the last extension-ray phase and the last shadow-ray phase that ran, when the wavefronts found their
queue empty and when they ended (both relative to the first start of the extension-ray phase of that launch)."""
import argparse, ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
ap = argparse.ArgumentParser(); ap.add_argument("--parts", type=int, default=8); ap.add_argument("--bounces", type=int, default=1)
a = ap.parse_args()
import torch
import scene_fixtures as sf
B = importlib.import_module("path-tracer_amd").binding
scene = sf.product_scene_from_mesh_arrays(*sf.sponza_standin(1), env=sf.procedural_env(1024), lights=sf.SPONZA_LIGHT)
dev = torch.device("cuda", 0)
ctx = B.Context(scene, device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
cam = B.make_camera(1920, 1080, sf.SPONZA_EYE, sf.SPONZA_LOOKAT)
accum = torch.zeros(1080 * 1920 * 3, device=dev); rgb = torch.zeros_like(accum)
L = B.lib()
buf = (C.c_ulonglong * (2 * 4 * 4096))()
for rep in range(3):
    L.mipt_debug_tail(None, 1)
    # bounces = 1: generate, closest(0), shade(0), ONE fused launch {closest(1), occluded(0)}, shade(1), occluded(1) alone
    ctx.render_device(cam, B.RenderParams(B.MODE_PATH, 16, rep * 16, a.bounces, 1, 0, a.parts, 64, 8), accum.data_ptr(), rgb.data_ptr())
    torch.cuda.synchronize()
    L.mipt_debug_tail(buf, 0)
    d = np.frombuffer(buf, np.uint64).reshape(2, 4096, 4).astype(np.float64)
    t0 = None
    for k, name in ((0, "extension rays"), (1, "shadow rays (last launch: alone)")):
        x = d[k][d[k][:, 2] > 0]
        start, empty, end = x[:, 0] / 100.0, x[:, 1] / 100.0, x[:, 2] / 100.0
        t0 = start.min()
        pct = lambda v, q: float(np.percentile(v, q))
        print(f"{name:34s} {len(x)} waves: starts {start.min() - t0:6.1f}..{start.max() - t0:6.1f} | queue-empty {empty.min() - t0:7.1f}..{empty.max() - t0:7.1f} | "
              f"ends p10 {pct(end, 10) - t0:7.1f} p50 {pct(end, 50) - t0:7.1f} p90 {pct(end, 90) - t0:7.1f} max {end.max() - t0:7.1f} | hand-overs {int(x[:, 3].sum())}", flush=True)
