#!/usr/bin/env python3
"""One rank's share of an N-GPU bench step, on one GPU: renders part `--part` of `--parts` of the 1080p frame with
16 * parts samples per step (what bench.py gives every rank) and prints the time per step, to be compared with the
N = 1 step of bench.py.  No collective; for checking that per-rank work does not grow with N."""
import argparse, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
ap = argparse.ArgumentParser()
ap.add_argument("--parts", type=int, default=8)
ap.add_argument("--part", type=int, default=3)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--strong", action="store_true", help="16 samples per step whatever --parts (one rank's share of ONE 16-spp frame: "
                                                     "bench.py's headline for N > 1) instead of 16 * parts (its weak figure)")
a = ap.parse_args()
import torch
import scene_fixtures as sf
B = importlib.import_module("path-tracer_amd").binding
W, H = 1920, 1080
scene = sf.product_scene_from_mesh_arrays(*sf.sponza_standin(1), env=sf.procedural_env(1024), lights=sf.SPONZA_LIGHT)
dev = torch.device("cuda", 0)
ctx = B.Context(scene, device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
cam = B.make_camera(W, H, sf.SPONZA_EYE, sf.SPONZA_LOOKAT)
accum = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
rgb = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
spp, sample = (16 if a.strong else 16 * a.parts), 0
for k in range(a.steps + 2):
    if k == 2:
        torch.cuda.synchronize(); ctx.reset_stats(); t0 = time.perf_counter()
    ctx.render_device(cam, B.RenderParams(B.MODE_PATH, spp, sample, 4, 1, a.part, a.parts, 64, 8), accum.data_ptr(), rgb.data_ptr())
    sample += spp
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
st = ctx.stats()
rays = (st["rays_closest"] + st["rays_occluded"]) / a.steps
print(f"part {a.part}/{a.parts}: {dt * 1e3:.2f} ms per step, {rays / 1e6:.1f} M rays per step, {rays / dt / 1e6:.0f} Mrays/s", flush=True)
