#!/usr/bin/env python3
"""Configs 1 and 2 of BASELINE.json alone (latency-bound frames), for `rocprofv3 --kernel-trace --stats`: which launches
a 0.1-0.6 ms frame is made of.  usage: small_configs.py 1|2 [frames]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import scene_fixtures as sf
B = importlib.import_module("path-tracer_amd").binding
which = sys.argv[1] if len(sys.argv) > 1 else "1"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
name, w, h, mode, bounces = ("cornellbox", 512, 512, B.MODE_PRIMARY, 0) if which == "1" else ("bunny", 1024, 1024, B.MODE_DIRECT, 6)
sc = sf.product_scene_from_mesh_arrays(*sf.load_scene_arrays(name), env=sf.procedural_env(1024), lights=sf.DEFAULT_LIGHT)
ctx = B.Context(sc, device=0, stream=stream)
cam = B.default_camera(w, h)
accum = torch.zeros(h * w * 3, dtype=torch.float32, device=dev)
rgb = torch.zeros(h * w * 3, dtype=torch.float32, device=dev)
for k in range(frames + 5):
    if k == 5:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.render_device(cam, B.RenderParams(mode, 1, k, bounces, 1, 0, 1, 64, 8), accum.data_ptr(), rgb.data_ptr())
torch.cuda.synchronize()
print(json.dumps({"config": which, "frames": frames, "ms_per_frame": round((time.perf_counter() - t0) / frames * 1e3, 4)}))
