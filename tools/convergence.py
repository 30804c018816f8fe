#!/usr/bin/env python3
"""BASELINE.json configs[4]: sponza-standin 3840x2160 progressive path tracing to 1024 spp (4 bounces, seed 1) on
ONE GPU: sustained Mrays/s over the whole run and the RMSE of the frame after 1, 2, 4, ... samples against the
1024-sample frame (mipt_image_compare).  The accumulation buffer is carried on the device between steps of 16
samples, as bench.py does; frames are read back only at the powers of two."""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=1024)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
a = ap.parse_args()
import numpy as np
import torch
import scene_fixtures as sf
B = importlib.import_module("path-tracer_amd").binding
W, H = a.width, a.height
dev = torch.device("cuda", 0)
scene = sf.product_scene_from_mesh_arrays(*sf.sponza_standin(1), env=sf.procedural_env(1024), lights=sf.SPONZA_LIGHT)
ctx = B.Context(scene, device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
cam = B.make_camera(W, H, sf.SPONZA_EYE, sf.SPONZA_LOOKAT)
accum = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
rgb = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
snaps, done, t_render = {}, 0, 0.0
marks = [1 << k for k in range(20) if (1 << k) <= a.spp]
ctx.reset_stats()
while done < a.spp:
    nxt = min([m for m in marks if m > done] + [a.spp])
    n = min(nxt - done, 16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.render_device(cam, B.RenderParams(B.MODE_PATH, n, done, 4, 1, 0, 1, 64, 8), accum.data_ptr(), rgb.data_ptr())
    torch.cuda.synchronize(); t_render += time.perf_counter() - t0
    done += n
    if done in marks or done == a.spp:
        snaps[done] = rgb.cpu().numpy().copy()
    if done % 128 == 0:
        print(f"{done} spp", file=sys.stderr, flush=True)
st = ctx.stats()
rays = st["rays_closest"] + st["rays_occluded"]
ref = snaps[a.spp]
curve = []
for k in sorted(snaps):
    rmse, mean_k, mean_ref = B.compare_images(snaps[k], ref)
    curve.append({"spp": k, "rmse": rmse, "rmse_over_mean": rmse / mean_ref, "mean": mean_k})
print(json.dumps({"config": f"sponza-standin {W}x{H} progressive path, {a.spp} spp, 4 bounces, seed 1, one MI355X",
                  "render_seconds": t_render, "rays": int(rays), "Mrays_per_s": rays / t_render / 1e6,
                  "reference": f"the {a.spp}-spp frame of the same run", "curve": curve}), flush=True)
